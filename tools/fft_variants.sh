#!/bin/bash
# usage: fft_variants.sh  (runs the FFTGS leg of bench.py under the A/B switches of the strided passes)
for v in "GSS_FFTGS_AXIS=1" "GSS_FFTGS_AXIS=2" "GSS_FFTGS_TXY=2" "GSS_FFTGS_TXY=2 GSS_FFTGS_TXZ=2" "GSS_FFTGS_TXZ=2"; do
  echo "== $v"
  env $v python bench.py --steps 1 --warmup 0 --no-cpu-baseline --lugs 0 --npoints 100000 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['fftgs']
print(d['value'], d['ms_per_realisation'], d['kernel_ms'], d['sample_variance'])"
done
