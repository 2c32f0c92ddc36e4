#!/bin/bash
# HBM traffic of the dominant kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), then the JSON bench.py reads.
set -e
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --streams 1 --graph 0 > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_traffic_json.py $(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) "${1:-void fcvsr::conv3_res_kernel<true, true, 1>}" gpurun_out/pmc_traffic.json 16 | tail -12
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +20M -delete
