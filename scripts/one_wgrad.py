"""Matrix-core weight gradient micro-benchmark / form comparison.
  FCVSR_WGRAD_FORM=1 python scripts/one_wgrad.py save /tmp/w1.pt ; FCVSR_WGRAD_FORM=2 python scripts/one_wgrad.py cmp /tmp/w1.pt"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
path = sys.argv[2] if len(sys.argv) > 2 else None
outs = {}
def run(B, H, W, cin, cout, k, iters=0):
    g = torch.Generator().manual_seed(B * 7 + H + cin + k)
    x = torch.randn(B, H, W, cin, generator=g).cuda()
    gy = torch.randn(B, H, W, cout, generator=g).cuda()
    n = L.fcvsr_conv2d_wgrad_mfma_scratch_elems(B, H, W, cin, cout, k, k)
    scratch = torch.empty(n, device="cuda")
    dw = torch.zeros(cout, cin, k, k, device="cuda")
    xv, gv = hip.view(x), hip.view(gy)
    def f():
        hip.check(L.fcvsr_conv2d_wgrad_mfma(C.byref(xv), C.byref(gv), B, H, W, k, k, 1, k // 2, dw.data_ptr(), scratch.data_ptr(), n, hip.stream_ptr()), "wgrad")
    f(); torch.cuda.synchronize()
    us = None
    if iters:
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
    return dw.cpu(), us
if mode in ("save", "cmp"):
    for cfg in ((2, 9, 37, 64, 64, 3), (1, 4, 32, 64, 128, 3), (3, 17, 33, 128, 64, 1), (2, 64, 64, 64, 64, 3), (1, 5, 3, 64, 64, 3)):
        outs[cfg] = run(*cfg)[0]
    if mode == "save":
        torch.save(outs, path)
    else:
        ref = torch.load(path)
        for k_, v in outs.items():
            d = (ref[k_] - v).abs()
            print(k_, "max abs diff %.4g (ref max %.3g), differing %.3f %%" % (float(d.max()), float(ref[k_].abs().max()), 100 * float((d > 0).float().mean())))
else:
    for cfg in ((4, 128, 128, 64, 64, 3), (4, 64, 64, 64, 64, 3), (4, 32, 32, 64, 64, 3), (4, 128, 128, 64, 128, 3), (4, 128, 128, 128, 64, 3), (4, 128, 128, 64, 64, 1), (8, 128, 128, 64, 64, 3)):
        _, us = run(*cfg, iters=20)
        fl = 2.0 * cfg[0] * cfg[1] * cfg[2] * cfg[3] * cfg[4] * cfg[5] * cfg[5]
        print("B=%d %dx%d %d->%d k%d: %.1f us (%.0f TFLOP/s) incl. reduction" % (*cfg, us, fl / us * 1e-6))
