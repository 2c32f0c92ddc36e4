import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
B, H, W, n = 8, 180, 320, 64
dt = torch.bfloat16
prev = torch.randn(B, H, W, n, device="cuda").to(dt); fin = torch.randn(B, H, W, n, device="cuda").to(dt)
off = torch.randn(B, H, W, 2, device="cuda"); K = torch.randn(B, H, W, 3 * n, device="cuda").to(dt)
dst = torch.empty_like(prev)
pv, ov, kv, fv, dv = (hip.view(t) for t in (prev, off, K, fin, dst))
def f():
    hip.check(L.fcvsr_iac_step(C.byref(pv), C.byref(ov), C.byref(kv), C.byref(fv), 0.1, B, H, W, C.byref(dv), hip.stream_ptr()), "iac")
for _ in range(5): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
byt = prev.numel() * 2 * 3 + K.numel() * 2 + off.numel() * 4
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"iac_step B={B}: {us:.1f} us, {byt/us/1e6:.2f} TB/s (algorithmic)")
