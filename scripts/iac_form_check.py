"""Second form of the fused IAC step (FCVSR_IAC_FORM=2) against the first on the same inputs: run once per form, compare the saved outputs.
usage: FCVSR_IAC_FORM=1 python scripts/iac_form_check.py save /tmp/f1.pt; FCVSR_IAC_FORM=2 python scripts/iac_form_check.py cmp /tmp/f1.pt"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
L = hip.lib()
mode, path = sys.argv[1], sys.argv[2]
outs = {}
for (B, H, W) in ((2, 37, 45), (1, 180, 320), (3, 8, 14), (2, 5, 3)):
    g = torch.Generator().manual_seed(B * 1000 + H)
    n, dt = 64, torch.bfloat16
    prev = [torch.randn(B, H, W, n, generator=g).cuda().to(dt) for _ in range(2)]
    fin = [torch.randn(B, H, W, n, generator=g).cuda().to(dt) for _ in range(2)]
    off = (torch.randn(B, H, W, 4, generator=g) * 2.5).cuda()
    k0 = torch.randn(B, H, W, n, generator=g).cuda().to(dt)
    w = (torch.randn(9 * n, n, 1, 1, generator=g) / 8).cuda(); bias = (torch.randn(9 * n, generator=g) * 0.1).cuda()
    wp = hip.pack_conv_weight_mfma(w, dt)
    dst = [torch.zeros_like(prev[0]), torch.zeros_like(prev[1])]
    V2 = hip.View * 2
    k0v = hip.view(k0)
    pv, ov = V2(hip.view(prev[0]), hip.view(prev[1])), V2(hip.view(off[..., 0:2]), hip.view(off[..., 2:4]))
    fv, dv = V2(hip.view(fin[0]), hip.view(fin[1])), V2(hip.view(dst[0]), hip.view(dst[1]))
    hip.check(L.fcvsr_iac_step2_fused(pv, ov, C.byref(k0v), wp.data_ptr() + 192 * 64 * 2, bias.data_ptr() + 192 * 4, fv, 0.1, B, H, W, dv, hip.stream_ptr()), "iac2f")
    torch.cuda.synchronize()
    outs[(B, H, W)] = [d.float().cpu() for d in dst]
if mode == "save":
    torch.save(outs, path)
else:
    ref = torch.load(path)
    for k, v in outs.items():
        for d in range(2):
            a, b = ref[k][d], v[d]
            diff = (a - b).abs()
            print(k, "dir", d, "max abs diff %.4g, differing %.4f %%, ref max %.3g, finite %s" % (float(diff.max()), 100.0 * float((diff > 0).float().mean()), float(a.abs().max()), bool(torch.isfinite(b).all())))
