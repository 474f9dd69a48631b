timeout 800 python -m pytest tests/test_gpu_multirank.py -m gpu -q 2>&1 | tail -15
timeout 300 python bench.py --workload glass --denoise --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_glass_denoise.json 2> gpurun_out/r02_bench_glass_denoise.err; tail -c 1500 gpurun_out/r02_bench_glass_denoise.json; tail -3 gpurun_out/r02_bench_glass_denoise.err
timeout 300 python bench.py --workload glass --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_glass.json 2> gpurun_out/r02_bench_glass.err; cut -c1-300 gpurun_out/r02_bench_glass.json
