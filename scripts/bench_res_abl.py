"""Ablation of the LDS-resident-weight 3x3 kernel (FCVSR_RES_DBG bits: 1 skip the halo copies, 2 skip the MFMA loop, 4 skip the stores)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bench_res import make, run, timeit, L3

def main():
    os.environ["FCVSR_MFMA_RES"] = "1"
    for cin, cout, B, levels in ((64, 64, 16, L3[:1]), (64, 128, 16, L3), (128, 64, 16, L3)):
        wp, bias, groups, flops = make(cin, cout, B, levels, True, 0)
        for dbg in (0, 32, 64, 96, 0, 32, 64, 96, 2, 13):
            os.environ["FCVSR_RES_DBG"] = str(dbg)
            us = timeit(lambda: run(groups, wp, bias, cout, 0))
            print(f"{cin}->{cout} B={B} levels={len(levels)} dbg={dbg}: {us:8.1f} us ({flops/us/1e6:7.1f} TF/s nominal)", flush=True)
    os.environ["FCVSR_RES_DBG"] = "0"

if __name__ == "__main__":
    main()
