"""Ablation of the three-group resident-weight kernel (FCVSR_RES_V=3/4; FCVSR_RES_DBG bits: 1 skip the halo copies, 2 skip the
MFMA loop, 4 no stores, 8 skip the store role, 16 copy role at priority 0, 32 store role at priority 1, 64 multiply at priority 3)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from bench_res_v import make, run, L3
from fcvsr_amd import hip

def main():
    os.environ["FCVSR_MFMA_RES"] = "1"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for cin, cout, B, levels, act in ((64, 64, 16, L3, hip.ACT_LEAKY), (64, 128, 16, L3, hip.ACT_LEAKY), (128, 64, 16, L3, hip.ACT_NONE)):
        wp, bias, groups, flops = make(cin, cout, B, levels, act)
        cfgs = [("1", 0), ("3", 0), ("4", 0), ("3", 16), ("3", 32), ("3", 64), ("3", 48), ("3", 9), ("3", 8), ("3", 1), ("3", 2), ("3", 11), ("1", 2), ("1", 13)]
        ts = {c: [] for c in cfgs}
        for rnd in range(5):
            for c in cfgs:
                os.environ["FCVSR_RES_V"], os.environ["FCVSR_RES_DBG"] = c[0], str(c[1])
                for _ in range(2): run(groups, wp, bias, cout, act)
                e0.record()
                for _ in range(10): run(groups, wp, bias, cout, act)
                e1.record(); torch.cuda.synchronize()
                ts[c].append(e0.elapsed_time(e1) * 100.0)
        for c in cfgs:
            t = sorted(ts[c])
            print(f"{cin}->{cout} v{c[0]} dbg={c[1]:3d}: med {t[len(t)//2]:7.1f} us min {t[0]:7.1f} ({flops/t[len(t)//2]/1e6:6.1f} TF/s nominal)", flush=True)
    os.environ["FCVSR_RES_DBG"] = "0"

if __name__ == "__main__":
    main()
