#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): GPU tests, the default bench line, the rocprofv3 kernel-trace + PMC
# passes of the same workload, and the bench lines of the other workloads.  Outputs under gpurun_out/ (copied to profiles/ by hand).
set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 280 python bench.py > gpurun_out/bench_helmet.json 2> gpurun_out/bench_helmet.err; tail -c 1500 gpurun_out/bench_helmet.json
tools/profile.sh r01_helmet --workload helmet --steps 3 --warmup 1 > /dev/null 2>&1
python tools/summarize_pmc.py gpurun_out/prof_r01_helmet gpurun_out/prof_r01_helmet/summary.json > /dev/null
ls gpurun_out/prof_r01_helmet
for w in atrium glass street; do
  timeout 200 python bench.py --no-cpu-baseline --steps 8 --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; cut -c1-200 gpurun_out/bench_$w.json
done
timeout 200 python bench.py --no-cpu-baseline --steps 8 --width 3840 --height 2160 > gpurun_out/bench_helmet_4k.json 2>/dev/null; cut -c1-200 gpurun_out/bench_helmet_4k.json
timeout 200 python bench.py --no-cpu-baseline --steps 64 --in-flight 1 > gpurun_out/bench_helmet_f1.json 2>/dev/null; cut -c1-200 gpurun_out/bench_helmet_f1.json
timeout 200 python bench.py --no-cpu-baseline --steps 32 --in-flight 8 > gpurun_out/bench_helmet_f8.json 2>/dev/null; cut -c1-200 gpurun_out/bench_helmet_f8.json
export TMPDIR=/tmp; out=$PWD/gpurun_out/prof_r01_atrium; mkdir -p $out
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $GRAFT_REPO_ROOT/bench.py --workload atrium --steps 3 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1)
