"""Fused convfuse stack vs the three 1x1 launches it replaces."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
B, H, Wf, n = int(os.environ.get("B", "8")), 180, 161, 64
spec = torch.randn(B, H, Wf, 6 * n, device="cuda")
x1f, x2f, x3f = spec[..., :2 * n], spec[..., 2 * n:4 * n], spec[..., 4 * n:]
dt = torch.bfloat16
p0 = hip.pack_conv_weight_mfma(torch.randn(128, 256, 1, 1, device="cuda") / 16, dt)
p2 = hip.pack_conv_weight_mfma(torch.randn(128, 128, 1, 1, device="cuda") / 11, dt)
p4 = hip.pack_conv_weight_mfma(torch.randn(128, 128, 1, 1, device="cuda") / 11, dt)
off = torch.empty(2 * B, H, Wf, 2 * n, device="cuda", dtype=dt); t0 = torch.empty_like(off); t1 = torch.empty_like(off)
dirs = list(enumerate((x1f, x3f)))
def split():
    hip.conv2d_mfma([dict(srcs=[xa, x2f], dst=t0[d * B:(d + 1) * B]) for d, xa in dirs], p0, 1, 128, hip.BF16, act=hip.ACT_RELU)
    hip.conv2d_mfma([dict(srcs=[t0], dst=t1)], p2, 1, 128, hip.BF16, act=hip.ACT_RELU)
    hip.conv2d_mfma([dict(srcs=[t1[d * B:(d + 1) * B]], dst=off[d * B:(d + 1) * B], res=[xa, x2f]) for d, xa in dirs], p4, 1, 128, hip.BF16, res_scale=[1.0, -1.0])
P2 = C.c_void_p * 2
def fused():
    hip.check(L.fcvsr_freq_mlp3(P2(x1f.data_ptr(), x3f.data_ptr()), P2(x2f.data_ptr(), x2f.data_ptr()), 2, 6 * n, B * H * Wf, p0.data_ptr(), p2.data_ptr(), p4.data_ptr(), P2(off[:B].data_ptr(), off[B:].data_ptr()), 2 * n, hip.stream_ptr()), "f")
def t(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
fl = 2.0 * 2 * B * H * Wf * (256 * 128 + 2 * 128 * 128)
a, b = t(fused), t(split)
print(f"B={B}: fused {a:.1f} us ({fl/a/1e6:.0f} TF/s) | three launches {b:.1f} us ({fl/b/1e6:.0f} TF/s)")
