import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
from bench_ws import make, run, timeit, L3
wp, bias, groups, flops = make(64, 64, 16, L3[:1], True, 0)
os.environ["FCVSR_MFMA_WS"] = "1"
for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
    os.environ["FCVSR_WS_DBG"] = str(dbg)
    us = timeit(lambda: run(groups, wp, bias, 64, 0))
    print(f"dbg {dbg} (1 no staging, 2 no MFMA, 4 no stores): {us:8.1f} us", flush=True)
