#!/bin/bash
# Collects rocprofv3 PMC passes for the kriging bench (one pass per counter group, as the TCC/SQ slot
# limits require).  Usage: tools/pmc_krig.sh <outdir> [extra bench args]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
shift
mkdir -p "$OUT"
i=0
for p in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $p --output-format csv -d "$OUT/pass$i" -o k -- \
    python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --fftgs 0 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
find "$OUT" -name "*.csv" | head -20
