"""3x3 MFMA conv: effect of 16-bit source / destination storage (same-process A/B, HIP events)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt, mdt = torch.bfloat16, hip.BF16
L3 = [(180, 320), (90, 160), (45, 80)]
def run(cin, cout, s16, d16, B=8, iters=20):
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    wp = hip.pack_conv_weight_mfma(w, dt)
    groups = []; flops = 0; byt = 0
    for (H, W) in L3:
        x = torch.randn(B, H, W, cin, device="cuda").to(dt if s16 else torch.float32)
        y = torch.empty(B, H, W, cout, device="cuda", dtype=dt if d16 else torch.float32)
        groups.append(dict(srcs=[x], dst=y)); flops += 2.0 * B * H * W * cin * cout * 9
        byt += x.numel() * x.element_size() + y.numel() * y.element_size()
    for _ in range(3): hip.conv2d_mfma(groups, wp, 3, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): hip.conv2d_mfma(groups, wp, 3, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{cin:3d}->{cout:3d} src {'16' if s16 else '32'} dst {'16' if d16 else '32'}: {us:8.1f} us {flops/us/1e6:7.1f} TF/s  {byt/us/1e6:5.2f} TB/s(alg)", flush=True)
for ntmax in ("128", "64"):
    os.environ["FCVSR_MFMA_NTMAX"] = ntmax
    print("NT max =", ntmax)
    for cin, cout, s16, d16 in ((64, 128, False, True), (64, 128, True, True), (64, 256, False, True)):
        run(cin, cout, s16, d16)
