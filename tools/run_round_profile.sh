set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 280 python bench.py > gpurun_out/bench_helmet.json 2> gpurun_out/bench_helmet.err; tail -c 1500 gpurun_out/bench_helmet.json
tools/profile.sh r01_helmet --workload helmet --steps 3 --warmup 1 > /dev/null 2>&1
python tools/summarize_pmc.py gpurun_out/prof_r01_helmet gpurun_out/prof_r01_helmet/summary.json > /dev/null
ls gpurun_out/prof_r01_helmet
