"""Per-layer conv timings (HIP events around every conv launch) for one forward at the bench shape."""
import os, sys, re, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from fcvsr_amd import hip
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict

B = int(os.environ.get("B", "4"))
prec = os.environ.get("FCVSR_PRECISION", "bf16")
MN = os.environ.get("MODEL", "GShiftNet_S"); m = getattr(A, MN)(); m.load_state_dict(synthetic_state_dict(state_dict_shapes(MN))); m = m.cuda(); m.precision = prec
x = torch.rand(B, 7, 1, 180, 320, device="cuda")
with torch.no_grad():
    for _ in range(3): m(x)
    hip.PROFILE = []
    m(x); torch.cuda.synchronize()
recs = hip.PROFILE; hip.PROFILE = None
agg = collections.OrderedDict()
for e0, e1, fl, kind, name, _nb, _var in recs:
    key = re.sub(r"body\.\d+\.body\.\d+", "body.G.body.K", name); key = re.sub(r"body\.\d+\.conv", "body.G.conv", key)
    key = re.sub(r"MConvB\.\d+", "MConvB.I", key)
    a = agg.setdefault((key, kind + ":" + str(_var)), [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"B={B} prec={prec}: total conv time {tot/1e3:.2f} ms over {len(recs)} launches")
for (k, kind), (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} {kind:20s} n={n:3d} {us:9.1f} us  avg {us/n:8.1f} us  {fl/us/1e6 if us else 0:7.1f} TF/s")
