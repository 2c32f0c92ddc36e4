"""feat_extract: dedicated single-K-step kernel vs the three generic MFMA launches."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
B = int(os.environ.get("B", "4"))
m = A.GShiftNet_S(); m.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"))); m = m.cuda(); m.precision = "bf16"; m.use_graph = False
x = torch.rand(B, 7, 1, 180, 320, device="cuda")
for flag in (True, False):
    m.fast_feat = flag
    with torch.no_grad():
        for _ in range(2): m(x)
        hip.PROFILE = []           # eager, per-launch events are only recorded for convs; time the whole forward instead
        hip.PROFILE = None
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"B={B} fast_feat={flag}: forward {dt*1e3:.3f} ms")
