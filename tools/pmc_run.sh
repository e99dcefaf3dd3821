#!/bin/bash
# Generic rocprofv3 PMC collection: tools/pmc_run.sh <outdir> -- <python script and args>
export TMPDIR=/tmp
OUT=$1; shift; shift
mkdir -p "$OUT"
i=0
for p in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d "$OUT/pass$i" -o k -- python3 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
