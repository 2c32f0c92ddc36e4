mkdir -p gpurun_out/r03p
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03p/pytest_gpu.txt 2>&1; grep "bf16 training mode\|passed\|failed" gpurun_out/r03p/pytest_gpu.txt | cut -c1-400
timeout -k 10 900 python bench.py > gpurun_out/r03p/bench_default.json 2> gpurun_out/r03p/bench_default.err; tail -c 2500 gpurun_out/r03p/bench_default.json; tail -3 gpurun_out/r03p/bench_default.err
