#!/bin/bash
# Round verification on the GPU box: GPU tests, default bench, multi-stream bench, rocprofv3 kernel-trace summaries.
set -o pipefail
R=$(pwd); mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 40 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail gpurun_out/bench_default.err; exit 1; }
timeout -k 10 200 python bench.py --steps 40 --streams 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_s1.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --steps 40 --batch 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_b1.json 2>/dev/null || exit 1
for i in 1 2 3; do STREAMS=4 SHAPE=16,7,1,180,320 timeout -k 10 200 python scripts/replay_determinism.py; done > gpurun_out/determinism_s4.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v10e -o v10e -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --graph 0 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_v10e.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v10d -o v10d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_v10d.log 2>&1 || exit 1
cd $R
python scripts/rocprof_summary.py $(find gpurun_out/prof_v10e -name "*results.db" | head -1) > gpurun_out/v10_eager1stream_kernel_stats.txt
python scripts/rocprof_summary.py $(find gpurun_out/prof_v10d -name "*results.db" | head -1) > gpurun_out/v10_default_kernel_stats.txt
find gpurun_out/prof_v10e gpurun_out/prof_v10d -name "*.db" -delete
echo done
