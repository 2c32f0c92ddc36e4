"""Lean 3x3 MFMA conv, 16-bit source and destination (the bench's dominant variant): three layer shapes, HIP events."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt, mdt = torch.bfloat16, hip.BF16
L3 = [(180, 320), (90, 160), (45, 80)]
def run(cin, cout, B=16, iters=30):
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    wp = hip.pack_conv_weight_mfma(w, dt)
    groups = []; flops = 0
    for (H, W) in L3:
        x = torch.randn(B, H, W, cin, device="cuda").to(dt)
        y = torch.empty(B, H, W, cout, device="cuda", dtype=dt)
        groups.append(dict(srcs=[x], dst=y)); flops += 2.0 * B * H * W * cin * cout * 9
    for _ in range(5): hip.conv2d_mfma(groups, wp, 3, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): hip.conv2d_mfma(groups, wp, 3, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{cin:3d}->{cout:3d} B={B}: {us:8.1f} us {flops/us/1e6:7.1f} TF/s", flush=True)
for cin, cout in ((64, 64), (64, 128), (128, 64)):
    run(cin, cout)
