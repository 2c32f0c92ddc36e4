"""IAC step micro-benchmark: single direction, both directions (shared K1), both directions with F[1] folded in."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
B, H, W, n = int(os.environ.get("B", "8")), 180, 320, 64
dt = torch.bfloat16
prev = [torch.randn(B, H, W, n, device="cuda").to(dt) for _ in range(2)]
fin = [torch.randn(B, H, W, n, device="cuda").to(dt) for _ in range(2)]
off = torch.randn(B, H, W, 4, device="cuda")
K = torch.randn(B, H, W, 9 * n, device="cuda").to(dt)
k0 = torch.randn(B, H, W, n, device="cuda").to(dt)
w = torch.randn(9 * n, n, 1, 1, device="cuda") / 8; bias = torch.randn(9 * n, device="cuda") * 0.1
wp = hip.pack_conv_weight_mfma(w, dt)
dst = [torch.empty_like(prev[0]), torch.empty_like(prev[1])]
V2 = hip.View * 2
kv = hip.view(K[..., 192:384]); k0v = hip.view(k0)
pv, ov = V2(hip.view(prev[0]), hip.view(prev[1])), V2(hip.view(off[..., 0:2]), hip.view(off[..., 2:4]))
fv, dv = V2(hip.view(fin[0]), hip.view(fin[1])), V2(hip.view(dst[0]), hip.view(dst[1]))
st = hip.stream_ptr()
def single():
    for d in range(2):
        hip.check(L.fcvsr_iac_step(C.byref(pv[d]), C.byref(ov[d]), C.byref(kv), C.byref(fv[d]), 0.1, B, H, W, C.byref(dv[d]), st), "iac")
def both():
    hip.check(L.fcvsr_iac_step2(pv, ov, C.byref(kv), fv, 0.1, B, H, W, dv, st), "iac2")
def fused():
    hip.check(L.fcvsr_iac_step2_fused(pv, ov, C.byref(k0v), wp.data_ptr() + 192 * 64 * 2, bias.data_ptr() + 192 * 4, fv, 0.1, B, H, W, dv, st), "iac2f")
def f1():
    hip.conv2d_mfma([dict(srcs=[k0], dst=K)], wp, 1, 9 * n, hip.BF16, bias=bias)
def t(f, iters=20):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print(f"B={B}: 2 x iac_step {t(single):.1f} us | iac_step2 {t(both):.1f} us | iac_step2_fused {t(fused):.1f} us | F.1 conv (all 3 iterations) {t(f1):.1f} us")
