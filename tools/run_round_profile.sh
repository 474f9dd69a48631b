#!/bin/bash
# Round evidence run on the GPU box (one gpurun call): the default bench line, the rocprofv3 kernel-trace + PMC passes of the same
# workload, kernel-trace stats and bench lines of the other workloads.  Outputs under gpurun_out/ (copied to profiles/ by hand).
# usage: tools/run_round_profile.sh <round tag, e.g. r02>
set -x
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > gpurun_out/${tag}_bench_helmet.json 2> gpurun_out/${tag}_bench_helmet.err; cut -c1-300 gpurun_out/${tag}_bench_helmet.json
tools/profile.sh ${tag}_helmet --workload helmet --steps 3 --warmup 1 > /dev/null 2>&1
python tools/summarize_pmc.py gpurun_out/prof_${tag}_helmet gpurun_out/prof_${tag}_helmet/summary.json > /dev/null
ls gpurun_out/prof_${tag}_helmet
timeout 300 python bench.py --workload atrium --steps 4 --warmup 1 > gpurun_out/${tag}_bench_atrium.json 2> gpurun_out/${tag}_bench_atrium.err; cut -c1-300 gpurun_out/${tag}_bench_atrium.json
for w in glass street; do
  timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $w > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err; cut -c1-300 gpurun_out/${tag}_bench_$w.json
done
tools/kstats.sh ${tag}_atrium --workload atrium --steps 1 --warmup 1
tools/kstats.sh ${tag}_glass --workload glass --steps 1 --warmup 1
