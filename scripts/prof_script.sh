#!/bin/bash
# kernel-trace stats of one micro-benchmark script:  scripts/prof_script.sh <out-tag> <script.py> [ENV=VAL ...]
TAG=$1; S=$2; shift 2
R=$(pwd); mkdir -p $R/gpurun_out/$TAG
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ps_$TAG -o p -- python3 $R/$S > $R/gpurun_out/$TAG/run.log 2>&1 || exit 1
cd $R
LONG_NAMES=${LONG_NAMES:-} python scripts/rocprof_summary.py $(find /tmp/ps_$TAG -name "*results.db" | head -1) > gpurun_out/$TAG/kernel_stats.txt
head -12 gpurun_out/$TAG/kernel_stats.txt | cut -c1-160
