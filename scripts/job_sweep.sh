mkdir -p gpurun_out/r03s
for cfg in "16 2" "32 2" "32 4" "48 3" "64 4" "64 2"; do set -- $cfg; echo "batch $1 streams $2"; timeout -k 10 300 python bench.py --batch $1 --streams $2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -k "rgb_twin" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -k "follows_the_weight" 2>&1 | tail -3
