"""conv3_res on fewer CUs (FCVSR_RES_CUS): does a power-limited MFMA kernel keep its rate when part of the chip is left free?"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt, mdt = torch.bfloat16, hip.BF16
B = 16
for cin, cout in ((64, 64), (64, 128), (128, 64)):
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    bias = torch.randn(cout, device="cuda")
    wp = hip.pack_conv_weight_mfma(w, dt)
    groups = []; flops = 0
    for (H, W) in ((180, 320), (90, 160), (45, 80)):
        x = torch.randn(B, H, W, cin, device="cuda").to(dt)
        y = torch.empty(B, H, W, cout, device="cuda", dtype=dt)
        groups.append(dict(srcs=[x], dst=y)); flops += 2.0 * B * H * W * cin * cout * 9
    f = lambda: hip.conv2d_mfma(groups, wp, 3, cout, mdt, bias=bias, act=hip.ACT_LEAKY, slope=0.2)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"FCVSR_RES_CUS={os.environ.get('FCVSR_RES_CUS', 'all')}: {cin}->{cout}: {us:.1f} us, {flops / us * 1e-6:.0f} TFLOP/s, kernel {hip.lib().fcvsr_last_conv_kernel().decode()}")
