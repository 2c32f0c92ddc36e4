"""LDS-resident-weight 3x3 kernel (conv_res.hip) vs the lean kernel: exactness (same accumulation order -> bit-identical) and speed."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt, mdt = torch.bfloat16, hip.BF16
L3 = [(180, 320), (90, 160), (45, 80)]

def make(cin, cout, B, levels, d16, nres):
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    bias = torch.randn(cout, device="cuda")
    wp = hip.pack_conv_weight_mfma(w, dt)
    groups = []; flops = 0
    for (H, W) in levels:
        x = torch.randn(B, H, W, cin, device="cuda").to(dt)
        y = torch.empty(B, H, W, cout, device="cuda", dtype=dt if d16 else torch.float32)
        res = [torch.randn(B, H, W, cout, device="cuda").to(dt) for _ in range(nres)]
        groups.append(dict(srcs=[x], dst=y, res=res)); flops += 2.0 * B * H * W * cin * cout * 9
    return wp, bias, groups, flops

def run(groups, wp, bias, cout, nres):
    hip.conv2d_mfma(groups, wp, 3, cout, mdt, bias=bias, act=hip.ACT_LEAKY, slope=0.1, res_scale=[1.0, -0.5][:nres])

def timeit(f, iters=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def main():
    import inspect
    print(inspect.signature(hip.conv2d_mfma))
    for cin, cout, B, levels, d16, nres in ((64, 64, 2, [(21, 37)], True, 0), (64, 64, 2, [(21, 37), (11, 19), (6, 10)], False, 2),
                                            (64, 128, 1, [(40, 70)], True, 1), (64, 256, 1, [(17, 33)], True, 0),
                                            (128, 64, 2, [(40, 70), (20, 35)], True, 1)):
        wp, bias, groups, _ = make(cin, cout, B, levels, d16, nres)
        os.environ["FCVSR_MFMA_RES"] = "0"; run(groups, wp, bias, cout, nres); torch.cuda.synchronize()
        ref = [g["dst"].clone() for g in groups]
        for g in groups: g["dst"].zero_()
        os.environ["FCVSR_MFMA_RES"] = "1"; run(groups, wp, bias, cout, nres); torch.cuda.synchronize()
        err = max(float((g["dst"].float() - r.float()).abs().max()) for g, r in zip(groups, ref))
        print(f"exactness {cin}->{cout} B={B} levels={levels} d16={d16} nres={nres}: max |res - lean| = {err:g}", flush=True)
    
    for cin, cout, B, levels, d16 in ((64, 64, 4, L3[:1], True), (64, 64, 16, L3[:1], True), (64, 64, 16, L3, True), (64, 64, 16, L3, False),
                                      (64, 128, 4, L3, True), (64, 128, 16, L3, True), (64, 256, 8, L3[:1], True),
                                      (128, 64, 4, L3, True), (128, 64, 16, L3, True)):
        wp, bias, groups, flops = make(cin, cout, B, levels, d16, 0)
        out = []
        for ws in ("0", "1"):
            os.environ["FCVSR_MFMA_RES"] = ws
            us = timeit(lambda: run(groups, wp, bias, cout, 0))
            out.append(f"{'res' if ws == '1' else 'lean'} {us:8.1f} us {flops/us/1e6:7.1f} TF/s")
        print(f"{cin}->{cout} B={B} levels={len(levels)} d16={d16}: " + " | ".join(out), flush=True)

if __name__ == '__main__':
    main()
