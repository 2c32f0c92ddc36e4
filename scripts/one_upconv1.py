"""upconv1 (1x1, 64 -> 256, PixelShuffle store, PReLU) at the bench shape vs the same GEMM without the shuffle."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt = torch.bfloat16
B, H, W = int(os.environ.get("B", "16")), 180, 320
x = torch.randn(B, H, W, 64, device="cuda").to(dt)
w = torch.randn(256, 64, 1, 1, device="cuda") / 8
b = torch.randn(256, device="cuda") * 0.1
slope = torch.tensor([0.25], device="cuda")
def t(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for ps in (True, False):
    wp = hip.pack_conv_weight_mfma(w, dt, ps=ps)
    bb = b[hip.ps_order(256).cuda()].contiguous() if ps else b
    y = torch.empty(B, 2 * H, 2 * W, 64, device="cuda", dtype=dt) if ps else torch.empty(B, H, W, 256, device="cuda", dtype=dt)
    f = lambda: hip.conv2d_mfma([dict(srcs=[x], dst=y)], wp, 1, 256, hip.BF16, bias=bb, act=hip.ACT_PRELU, slope_t=slope, pixel_shuffle=ps)
    us = t(f)
    byt = x.numel() * 2 + y.numel() * 2
    print(f"ps={ps}: {us:.1f} us, {byt / us / 1e6:.2f} TB/s, kernel {hip.lib().fcvsr_last_conv_kernel().decode()}")
