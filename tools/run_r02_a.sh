#!/bin/bash
# Round-2 evidence run A (one gpurun call): GPU tests, default bench line (with cpu baseline + parity), atrium line, kernel-trace stats of both.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r02_gputest.txt
timeout 400 python bench.py > gpurun_out/r02_bench_helmet.json 2> gpurun_out/r02_bench_helmet.err; tail -c 3000 gpurun_out/r02_bench_helmet.json; tail -3 gpurun_out/r02_bench_helmet.err
timeout 400 python bench.py --workload atrium --steps 4 --warmup 1 > gpurun_out/r02_bench_atrium.json 2> gpurun_out/r02_bench_atrium.err; tail -c 3000 gpurun_out/r02_bench_atrium.json; tail -3 gpurun_out/r02_bench_atrium.err
tools/kstats.sh r02_helmet --workload helmet --steps 3 --warmup 1
tools/kstats.sh r02_atrium --workload atrium --steps 1 --warmup 1
