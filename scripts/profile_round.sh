#!/bin/bash
# Round profiles on the GPU box (outputs under gpurun_out/prof/, summaries only): rocprofv3 kernel-trace stats of the bench command
# (eager single stream, and the default 4-stream hipGraph replay), HBM-traffic PMC passes of the dominant kernel, SQ counters.
set -o pipefail
R=$(pwd); O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_e /tmp/pr_d
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_e -o e -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras --no-train > $O/prof_e.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/pr_d -o d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-train > $O/prof_d.log 2>&1 || exit 1
cd $R
python scripts/rocprof_summary.py $(find /tmp/pr_e -name "*results.db" | head -1) > $O/bf16_b16_eager1stream_kernel_stats.txt
python scripts/rocprof_summary.py $(find /tmp/pr_d -name "*results.db" | head -1) > $O/bf16_b16_default_kernel_stats.txt
KN="${1:-void fcvsr::conv3_res_kernel<true, 2, 1, 1>}"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-train --streams 1 --graph 0 > $O/pmc_$c.log 2>&1 || exit 1
done
cd $R
python scripts/pmc_traffic_json.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) "$KN" $O/pmc_traffic.json 16 | tail -14
python scripts/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/bf16_b16_pmc_hbm_traffic.txt > /dev/null
head -32 $O/bf16_b16_eager1stream_kernel_stats.txt | cut -c1-150
echo done
