#!/bin/bash
# Round-3 evidence run, third edition (after the shade-path round trips and the one-record finish): same contents as tools/run_r03_final.sh.
cd "$(dirname "$0")/.."; ulimit -c 0
O=$PWD/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/r03_gputest_final.txt 2>&1; echo "GPUTEST rc=$? $(tail -1 $O/r03_gputest_final.txt)"
for w in helmet atrium; do
  rm -rf $O/prof_r03f_$w
  tools/profile.sh r03f_$w --workload $w --steps 3 --warmup 1 > /dev/null 2>&1
  python tools/summarize_pmc.py $O/prof_r03f_$w $O/prof_r03f_$w/summary.json > /dev/null
  python tools/make_pmc_latest.py $O/prof_r03f_$w/summary.json $w 128 3 profiles/r03_fetch_calibration.json > $O/pmc_latest_$w.json
  cp $O/pmc_latest_$w.json profiles/pmc_latest_$w.json
  python -c "
import json; j=json.load(open('$O/pmc_latest_$w.json')); print('PMC $w', {k:(round(v['hbm_bytes_per_launch']/1e9,2), v['avg_us']) for k,v in j['kernels'].items()})"
done
timeout 600 python bench.py > $O/r03_bench_helmet.json 2> $O/r03_bench_helmet.err; echo "BENCH rc=$?"; cut -c1-300 $O/r03_bench_helmet.json
timeout 400 python bench.py --workload street --steps 3 --warmup 1 > $O/r03_bench_street.json 2> $O/r03_bench_street.err; cut -c1-200 $O/r03_bench_street.json
timeout 300 python bench.py --workload glass --steps 3 --warmup 1 > $O/r03_bench_glass.json 2> $O/r03_bench_glass.err; cut -c1-200 $O/r03_bench_glass.json
timeout 300 python bench.py --workload glass --denoise --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_glass_denoise.json 2> $O/r03_bench_glass_denoise.err; cut -c1-200 $O/r03_bench_glass_denoise.json
timeout 200 python bench.py --workload helmet --in-flight 1 --frames-per-step 64 --steps 3 --warmup 1 --no-cpu-baseline --also none > $O/r03_bench_helmet_f1.json 2>/dev/null; cut -c1-200 $O/r03_bench_helmet_f1.json
timeout 200 python bench.py --workload helmet --in-flight 8 --frames-per-step 64 --steps 3 --warmup 1 --no-cpu-baseline --also none > $O/r03_bench_helmet_f8.json 2>/dev/null; cut -c1-200 $O/r03_bench_helmet_f8.json
