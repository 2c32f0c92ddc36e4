"""Micro-benchmark of the 1x1 (flat) MFMA conv path with the in-model tensor dtypes / layouts."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip

dt = torch.bfloat16
mdt = hip.BF16


def run(name, B, H, W, cin, cout, src16, dst16, ps=False, pix_stride=None, iters=20):
    w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    wp = hip.pack_conv_weight_mfma(w, dt, ps=ps)
    sd = dt if src16 else torch.float32
    if pix_stride:
        buf = torch.randn(B, H, W, pix_stride, device="cuda").to(sd)
        x = buf[..., :cin]
    else:
        x = torch.randn(B, H, W, cin, device="cuda").to(sd)
    dd = dt if dst16 else torch.float32
    y = torch.empty(B, 2 * H, 2 * W, cout // 4, device="cuda", dtype=dd) if ps else torch.empty(B, H, W, cout, device="cuda", dtype=dd)
    g = [dict(srcs=[x], dst=y)]
    for _ in range(3):
        hip.conv2d_mfma(g, wp, 1, cout, mdt, pixel_shuffle=ps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.conv2d_mfma(g, wp, 1, cout, mdt, pixel_shuffle=ps)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    byt = x.numel() * x.element_size() + y.numel() * y.element_size()
    print(f"{name:40s} {us:9.1f} us  {byt/us/1e6:6.2f} TB/s  {2.0*B*H*W*cin*cout/us/1e6:7.1f} TF/s", flush=True)


B = 4
for dbg in os.environ.get("DBGS", "0").split(","):
    os.environ["FCVSR_MFMA_DBG"] = dbg
    for mw in os.environ.get("MWS", "2,1").split(","):
        os.environ["FCVSR_MFMA_MW"] = mw
        print(f"--- dbg={dbg} MW={mw}")
        run("F1 64->576 src f32 dst f32", B, 180, 320, 64, 576, False, False)
        run("F1 64->576 src bf16 dst f32", B, 180, 320, 64, 576, True, False)
        run("F1 64->576 src bf16 dst bf16", B, 180, 320, 64, 576, True, True)
        run("upconv2 64->256 PS f32->f32", B, 360, 640, 64, 256, False, False, ps=True, iters=5)
        run("upconv2 64->256 PS bf16->bf16", B, 360, 640, 64, 256, True, True, ps=True, iters=5)
        run("upconv2-like 64->256 noPS bf16->bf16", B, 360, 640, 64, 256, True, True, ps=False, iters=5)
        run("convfuse.0-like 128->128 strided f32", B, 180, 161, 128, 128, False, True, pix_stride=384)
        run("convfuse.0-like 128->128 dense f32", B, 180, 161, 128, 128, False, True)
        run("down.0 64->64 f32->f32", B, 180, 320, 64, 64, False, False)
