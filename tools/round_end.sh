#!/bin/bash
# End-of-round measurements on the GPU box (outputs under gpurun_out/round_end/):
#   bash tools/round_end.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round_end
mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
tail -1 $O/bench_n1.json | cut -c1-400
timeout -k 10 600 python3 bench_configs.py --configs 1h,2,3,4,cond_fftgs,idw,lwr,sgs > $O/bench_configs.jsonl 2> $O/bench_configs.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -o k -- python3 bench_configs.py --configs 4,idw,lwr > $O/cfg4_under_rocprof.jsonl 2> $O/rocprof4.err || exit 1
echo done
