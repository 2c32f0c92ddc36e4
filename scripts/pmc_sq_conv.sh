#!/bin/bash
# SQ counter pass over scripts/one_conv.py (3x3 MFMA convolution, 64 -> 64, three pyramid levels of B clips; FCVSR_MFMA_RES=0 selects
# the lean kernel, default the LDS-resident-weight kernel)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_conv -o sq -- python3 $R/scripts/one_conv.py > $R/gpurun_out/pmc_conv.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_conv2 -o sq -- python3 $R/scripts/one_conv.py >> $R/gpurun_out/pmc_conv.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.Counter()
for path in glob.glob("gpurun_out/pmc_conv*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "conv3_" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {v:18.0f}  {100 * v / wc:7.1f} % of SQ_WAVE_CYCLES")
PY
