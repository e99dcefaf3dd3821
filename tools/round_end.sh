#!/bin/bash
# End-of-round measurements on the GPU box (outputs under gpurun_out/round_end/):
#   bash tools/round_end.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round_end
rm -rf $O; mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
tail -1 $O/bench_n1.json | cut -c1-300
timeout -k 10 1100 python3 bench_configs.py --configs 1h,2,2h,3,4,4full,api,bigk,lu,cond_fftgs,fftgs_gen,idw,lwr,sgs,sgs_bigk > $O/bench_configs.jsonl 2> $O/bench_configs.err || exit 1
echo configs done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats4 -o k -- python3 bench_configs.py --configs 4,bigk,idw,lwr > $O/cfg4_under_rocprof.jsonl 2> $O/rocprof4.err || exit 1
echo stats done
# PMC passes (each group in a run of its own, --kernel-trace only beside --pmc): K5 / K4 on configs[4], the five FFTGS
# passes at 512^3, the kriging step of configs[1]
bash tools/pmc_run.sh $O/pmc_k5 -- bench_configs.py --configs 4 > $O/pmc_k5.log 2>&1
python3 tools/pmc_summary.py $O/pmc_k5 krig_local > $O/pmc_k5_summary.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_k5 knn_pruned >> $O/pmc_k5_summary.txt 2>&1
echo pmc k5 done
bash tools/pmc_run.sh $O/pmc_fft -- tools/fftgs_one.py 512 4 > $O/pmc_fft.log 2>&1
python3 tools/pmc_traffic_fftgs.py $O/pmc_fft $O/fftgs_512_pmc_traffic.json > $O/pmc_fft_traffic.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fft ff_ > $O/fftgs_512_pmc_summary.txt 2>&1
echo pmc fft done
bash tools/pmc_krig.sh $O/pmc_krig --lugs 0 > $O/pmc_krig.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_krig krig_quadform $O/krig_cfg2_pmc_traffic.json > $O/pmc_krig_traffic.log 2>&1
python3 tools/pmc_summary.py $O/pmc_krig krig_ > $O/krig_cfg2_pmc_summary.txt 2>&1
echo pmc krig done
# keep what is small: the stats CSVs and the summaries (the traces and counter dumps are tens of MB)
find $O -name "*kernel_stats.csv" | while read f; do cp "$f" "$O/$(echo $f | sed 's#/#_#g' | sed 's#.*round_end_##')"; done
rm -rf $O/stats $O/stats4 $O/pmc_k5 $O/pmc_fft $O/pmc_krig
ls -la $O
echo done
