"""A/B of the resident-weight 3x3 kernel variants (FCVSR_RES_V=1: round 2, =2: round 3) on the path's layer shapes:
bit-equality with the lean kernel and interleaved timings in one process (median / min over rounds)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt, mdt = torch.bfloat16, hip.BF16
L3 = [(180, 320), (90, 160), (45, 80)]
VARIANTS = os.environ.get("VARIANTS", "1,3").split(",")


def make(cin, cout, B, levels, act):
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    bias = torch.randn(cout, device="cuda")
    wp = hip.pack_conv_weight_mfma(w, dt)
    groups = []; flops = 0
    for (H, W) in levels:
        x = torch.randn(B, H, W, cin, device="cuda").to(dt)
        y = torch.empty(B, H, W, cout, device="cuda", dtype=dt)
        groups.append(dict(srcs=[x], dst=y, res=[])); flops += 2.0 * B * H * W * cin * cout * 9
    return wp, bias, groups, flops


def run(groups, wp, bias, cout, act):
    hip.conv2d_mfma(groups, wp, 3, cout, mdt, bias=bias, act=act, slope=0.1)


def main():
    for cin, cout, B, levels, act in ((64, 64, 2, [(21, 37)], hip.ACT_LEAKY), (64, 64, 3, [(21, 37), (11, 19), (6, 10)], hip.ACT_NONE),
                                      (64, 128, 1, [(40, 70)], hip.ACT_LEAKY), (128, 64, 2, [(40, 70), (20, 35)], hip.ACT_NONE),
                                      (64, 64, 16, L3, hip.ACT_LEAKY), (128, 64, 16, L3, hip.ACT_NONE)):
        wp, bias, groups, _ = make(cin, cout, B, levels, act)
        os.environ["FCVSR_MFMA_RES"] = "0"; run(groups, wp, bias, cout, act); torch.cuda.synchronize()
        ref = [g["dst"].clone() for g in groups]
        os.environ["FCVSR_MFMA_RES"] = "1"
        for v in VARIANTS:
            for g in groups: g["dst"].zero_()
            os.environ["FCVSR_RES_V"] = v; run(groups, wp, bias, cout, act); torch.cuda.synchronize()
            eq = all(torch.equal(g["dst"], r) for g, r in zip(groups, ref))
            err = max(float((g["dst"].float() - r.float()).abs().max()) for g, r in zip(groups, ref))
            print(f"exact {cin}->{cout} B={B} L={len(levels)} act={act} v{v} [{hip.lib().fcvsr_last_conv_kernel().decode()}]: bit-equal to lean {eq} (max diff {err:g})", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for cin, cout, B, levels, act in ((64, 64, 16, L3, hip.ACT_LEAKY), (64, 64, 16, L3, hip.ACT_NONE), (64, 128, 16, L3, hip.ACT_LEAKY),
                                      (128, 64, 16, L3, hip.ACT_NONE), (64, 64, 8, L3, hip.ACT_LEAKY), (64, 128, 8, L3, hip.ACT_LEAKY)):
        wp, bias, groups, flops = make(cin, cout, B, levels, act)
        ts = {v: [] for v in VARIANTS}
        for rnd in range(7):
            for v in VARIANTS:
                os.environ["FCVSR_RES_V"] = v
                for _ in range(2): run(groups, wp, bias, cout, act)
                e0.record()
                for _ in range(10): run(groups, wp, bias, cout, act)
                e1.record(); torch.cuda.synchronize()
                ts[v].append(e0.elapsed_time(e1) * 100.0)
        out = []
        for v in VARIANTS:
            t = sorted(ts[v]); med, mn = t[len(t) // 2], t[0]
            out.append(f"v{v} med {med:7.1f} us ({flops/med/1e6:6.1f} TF/s) min {mn:7.1f}")
        print(f"{cin}->{cout} B={B} act={act}: " + " | ".join(out), flush=True)


if __name__ == '__main__':
    main()
