"""Micro-benchmark of fcvsr_conv2d_mfma on the layer shapes of the path (HIP events, same-process A/B)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip

torch.manual_seed(0)
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float16
mdt = hip.BF16 if dt == torch.bfloat16 else hip.F16


def run(name, B, levels, cin, cout, k, iters=30):
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    wp = hip.pack_conv_weight_mfma(w, dt)
    groups = []
    flops = 0
    for (H, W) in levels:
        x = torch.randn(B, H, W, cin, device="cuda")
        y = torch.empty(B, H, W, cout, device="cuda")
        groups.append(dict(srcs=[x], dst=y))
        flops += 2.0 * B * H * W * cin * cout * k * k
    for _ in range(5):
        hip.conv2d_mfma(groups, wp, k, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.conv2d_mfma(groups, wp, k, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:34s} B={B} {us:9.1f} us  {flops/us/1e6:8.1f} TFLOP/s", flush=True)


L3 = [(180, 320), (90, 160), (45, 80)]
if os.environ.get("ABL2"):
    def run2(name, B, levels, cin, cout, k, dst16, iters=20):
        w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        wp = hip.pack_conv_weight_mfma(w, dt)
        groups = []; flops = 0
        for (H, W) in levels:
            x = torch.randn(B, H, W, cin, device="cuda")
            y = torch.empty(B, H, W, cout, device="cuda", dtype=dt if dst16 else torch.float32)
            groups.append(dict(srcs=[x], dst=y)); flops += 2.0 * B * H * W * cin * cout * k * k
        for _ in range(3): hip.conv2d_mfma(groups, wp, k, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): hip.conv2d_mfma(groups, wp, k, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"  {name:30s} {us:8.1f} us {flops/us/1e6:7.1f} TF/s", flush=True)
    for dbg in (0, 8, 1, 9, 2, 4, 13, 15):
        os.environ["FCVSR_MFMA_DBG"] = str(dbg)
        print("dbg", dbg, "(1 staging, 2 MFMA, 4 stores, 8 weights skipped)")
        run2("64->128 bf16out L0-2 B=4", 4, L3, 64, 128, 3, True)
        run2("64->64 f32out L0-2 B=4", 4, L3, 64, 64, 3, False)
    sys.exit(0)
if os.environ.get("MWTEST"):
    for mw in ("2", "1"):
        os.environ["FCVSR_MFMA_MW"] = mw
        print("MW", mw)
        for B in (1, 4):
            run("3x3 64->128 L0+L1+L2", B, L3, 64, 128, 3)
            run("3x3 128->64 L0+L1+L2", B, L3, 128, 64, 3)
            run("3x3 64->64 L0+L1+L2", B, L3, 64, 64, 3)
            run("1x1 64->576 L0", B, L3[:1], 64, 576, 1)
            run("1x1 256->128 freq(180x161)", B, [(180, 161)], 256, 128, 1)
    sys.exit(0)
if os.environ.get("ABLATE"):
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        os.environ["FCVSR_MFMA_DBG"] = str(dbg)
        print("dbg", dbg, "(1 skip staging, 2 skip MFMA, 4 skip stores)")
        for B in (1, 4):
            run("3x3 64->128 L0", B, L3[:1], 64, 128, 3)
            run("3x3 128->64 L0", B, L3[:1], 128, 64, 3)
    sys.exit(0)
for B in (1, 4, 8):
    run("3x3 64->128 L0", B, L3[:1], 64, 128, 3)
    run("3x3 64->128 L0+L1+L2", B, L3, 64, 128, 3)
    run("3x3 128->64 L0+L1+L2", B, L3, 128, 64, 3)
    run("3x3 64->64 L0+L1+L2", B, L3, 64, 64, 3)
    run("1x1 64->576 L0", B, L3[:1], 64, 576, 1)
    run("1x1 256->128 freq(180x161)", B, [(180, 161)], 256, 128, 1)
    run("3x3 64->1 720x1280", B, [(720, 1280)], 64, 1, 3, iters=10)
