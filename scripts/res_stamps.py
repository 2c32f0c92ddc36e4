"""Where a phase of the LDS-resident-weight 3x3 kernel spends its cycles: in-kernel s_memtime stamps of one workgroup
(FCVSR_RES_STAMPS=1; a diagnostic path, no stamp executes otherwise).  Slots: 0 phase start, 1 MFMA loop done, 2 multiply role
done (bias/act tail), 3 LDS-DMA issued, 4 stores issued, 5 copies landed, 6 at the barrier."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ["FCVSR_RES_STAMPS"] = "1"
os.environ["FCVSR_MFMA_RES"] = "1"
import numpy as np
import torch
from fcvsr_amd import hip
from bench_res import make, run, L3

def main():
    cin, cout = int(os.environ.get("CIN", "64")), int(os.environ.get("COUT", "64"))
    wp, bias, groups, flops = make(cin, cout, 16, L3, True, 0)
    for _ in range(3):
        run(groups, wp, bias, cout, 0)
    torch.cuda.synchronize()
    buf = np.zeros((8, 64, 8), dtype=np.uint64)
    hip.check(hip.lib().fcvsr_debug_res_stamps(buf.ctypes.data_as(C.c_void_p), buf.nbytes), "stamps")
    st = buf.astype(np.int64)
    for w in (0, 4):
        print(f"wave {w}: phase | role | start->loop | loop->tail | start->dma | dma->stores | stores->landed | role end->barrier | phase length")
        for p in range(2, 14):
            s = st[w, p]
            nxt = st[w, p + 1, 0]
            if s[1]:
                print(f"  {p:3d}  mul   {s[1]-s[0]:7d} {s[2]-s[1]:7d} {'':7s} {'':7s} {'':7s} {s[6]-s[2]:7d} {nxt-s[0]:7d}")
            else:
                print(f"  {p:3d}  store {'':7s} {'':7s} {s[3]-s[0]:7d} {s[4]-s[3]:7d} {s[5]-s[4]:7d} {s[6]-s[5]:7d} {nxt-s[0]:7d}")

if __name__ == "__main__":
    main()
