"""Fused up-sampler tail vs the two stand-alone launches it replaces (HIP events, same process)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
B, H2, W2, n = int(os.environ.get("B", "4")), 360, 640, 64
dt = torch.bfloat16
u1 = torch.randn(B, H2, W2, n, device="cuda").to(dt)
w2 = torch.randn(256, 64, 1, 1, device="cuda") / 8; b2 = torch.randn(256, device="cuda") * 0.1
wl = torch.randn(1, 64, 3, 3, device="cuda") / 24; bl = torch.randn(1, device="cuda")
slope = torch.tensor([0.25], device="cuda")
w2p = hip.pack_conv_weight_mfma(w2, dt, ps=True); b2p = b2[hip.ps_order(256).cuda()].contiguous()
wlp = hip.pack_conv_weight_mfma(wl, dt)
tab = torch.zeros(16, 64, device="cuda"); tab[:9] = wl[0].permute(1, 2, 0).reshape(9, 64); tab = tab.to(dt).contiguous()
out = torch.zeros(B, 1, 2 * H2, 2 * W2, device="cuda"); out_v = out.permute(0, 2, 3, 1)
u2 = torch.empty(B, 2 * H2, 2 * W2, n, device="cuda", dtype=dt)
u1v, ov = hip.view(u1), hip.view(out_v)
def fused():
    hip.check(L.fcvsr_tail_fused(C.byref(u1v), w2p.data_ptr(), b2p.data_ptr(), slope.data_ptr(), tab.data_ptr(), bl.data_ptr(), B, H2, W2, C.byref(ov), hip.stream_ptr()), "tail")
def split():
    hip.conv2d_mfma([dict(srcs=[u1], dst=u2)], w2p, 1, 256, hip.BF16, bias=b2p, act=hip.ACT_PRELU, slope_t=slope, pixel_shuffle=True)
    hip.conv2d_mfma([dict(srcs=[u2], dst=out_v, res=[out_v])], wlp, 3, 1, hip.BF16, bias=bl, res_scale=[1.0])
def t(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print(f"B={B}: fused tail {t(fused):.1f} us | upconv2 + conv_last0 {t(split):.1f} us")
