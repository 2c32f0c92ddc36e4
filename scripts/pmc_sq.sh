#!/bin/bash
# SQ counter pass over scripts/bench_fft.py (wave-cycle breakdown of the FFT kernels): rocprofv3 --pmc only, one pass.
R=$(pwd); cd /tmp && export TMPDIR=/tmp
B=16 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_fft -o sq -- python3 $R/scripts/bench_fft.py > $R/gpurun_out/pmc_fft.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_fft/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if "fft" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v:16.0f}  {100 * v / wc:6.1f} % of wave cycles")
PY
