mkdir -p gpurun_out/r03e
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -s > gpurun_out/r03e/pytest_train.txt 2>&1; grep "bf16 training mode\|passed\|failed\|Error" gpurun_out/r03e/pytest_train.txt | cut -c1-300
GRAPH=1 timeout -k 10 300 python scripts/one_train.py 5 2>&1 | tail -1
export LONG_NAMES=1
bash scripts/prof_script.sh r03e_prof scripts/one_train.py GRAPH=0 > gpurun_out/r03e/prof.txt 2>&1; head -24 gpurun_out/r03e_prof/kernel_stats.txt | cut -c1-170
