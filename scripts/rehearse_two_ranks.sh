mkdir -p gpurun_out/r03g2
timeout -k 10 900 python bench.py --gpus 2 --device 0 --backend gloo --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/r03g2/bench_g2.json 2> gpurun_out/r03g2/bench_g2.err; echo rc=$?; tail -c 1800 gpurun_out/r03g2/bench_g2.json; tail -4 gpurun_out/r03g2/bench_g2.err | cut -c1-300
FCVSR_BENCH_TRAIN_FAIL_RANK=1 timeout -k 10 900 python bench.py --gpus 2 --device 0 --backend gloo --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/r03g2/bench_g2_fail.json 2> gpurun_out/r03g2/bench_g2_fail.err; echo rc=$?; python -c "
import json; d=json.loads(open('gpurun_out/r03g2/bench_g2_fail.json').read().strip().splitlines()[-1]); print(d['value'], d['train'])"
