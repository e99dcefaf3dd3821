#!/bin/bash
# usage: tools/fft_variants.sh ["ENV=.. ENV=.." ...]  -- FFTGS leg of bench.py under A/B switches of the fused passes
if [ $# -eq 0 ]; then set -- "GSS_FFTGS_SLAB=0" "GSS_FFTGS_OVERLAP=0" "GSS_FFTGS_OVERLAP=1"; fi
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 1 --warmup 0 --no-cpu-baseline --lugs 0 --npoints 100000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['fftgs']
print(d['value'], 'per s;', d['ms_per_realisation'], 'ms wall;', d['roofline']['avg_ms'], 'ms events;', d['roofline']['sequential_ms'], 'ms sequential kernels;', d['kernel_ms'], d['sample_variance'], d['end_to_end_32_per_gpu']['value'])"
done
