#!/bin/bash
# Counter passes over one micro-benchmark, summarised for the kernels whose name contains $1.
#   scripts/pmc_kernel.sh <name-substring> <out-tag> <script.py> [ENV=VAL ...]
# rocprofv3 --pmc only (never combined with the sys / runtime traces); one pass per counter group (SQ groups only: the TCC / TCP
# groups did not finish within 5 minutes on this pool).
F=$1; TAG=$2; S=$3; shift 3
R=$(pwd); mkdir -p $R/gpurun_out/$TAG; cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$TAG/g$i -o g -- python3 $R/$S > $R/gpurun_out/$TAG/run$i.log 2>&1 || echo "group $i failed"
done
cd $R
python3 - "$F" /tmp/pmc_$TAG > gpurun_out/$TAG/summary.txt <<'PY'
import csv, glob, collections, sys
filt, root = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:80]
        if filt not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    print(k)
    for c, v in sorted(d.items()):
        per = v / max(1, n[k][c])
        print(f"   {c:32s} {per:16.0f} per launch" + (f"  {100 * v / wc:6.1f} % of wave cycles" if wc and c.startswith("SQ_") else ""))
PY
cat gpurun_out/$TAG/summary.txt
