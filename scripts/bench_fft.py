import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
B, H, W, n = int(os.environ.get("B", "8")), 180, 320, 64
Wf = W // 2 + 1
src = torch.rand(B, H, W, n, device="cuda")
if os.environ.get("SRC", "f32") == "bf16": src = src.to(torch.bfloat16)
spec = torch.empty(B, H, Wf, 2 * n, device="cuda"); work = torch.empty_like(spec)
dst = torch.empty(B, H, W, n, device="cuda"); mask = torch.rand(H, Wf, device="cuda")
sv, dv = hip.view(src), hip.view(dst)
def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
f = lambda: hip.check(L.fcvsr_rfft2(C.byref(sv), B, H, W, n, spec.data_ptr(), 2 * n, 0, n, hip.stream_ptr()), "r")
g = lambda: hip.check(L.fcvsr_irfft2(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, mask.data_ptr(), work.data_ptr(), C.byref(dv), hip.stream_ptr()), "i")
byts = src.numel() * 4 + 3 * spec.numel() * 4
print(f"LDS budget {os.environ.get('FCVSR_FFT_LDS_KB','128')} KB: rfft2 {t(f):8.1f} us  irfft2(mask) {t(g):8.1f} us   (ideal @4.5TB/s: {byts/4.5e6:.0f} us each)")
