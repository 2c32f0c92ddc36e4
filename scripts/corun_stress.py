"""Whole-network victim test for the packed-FP32 / LDS hazard: a single-stream graph replay of the
model runs on one stream while another stream keeps every CU's LDS busy with the lean 3x3 MFMA convolution.  Every
output must be bit-identical to the solo result."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from fcvsr_amd import hip
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
m = A.GShiftNet_S(); m.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"))); m = m.cuda()
m.precision = os.environ.get("PREC", "bf16"); m.streams = 1; m.use_graph = True
w = torch.randn(64, 64, 3, 3, device="cuda") / 24
wp = hip.pack_conv_weight_mfma(w, torch.bfloat16)
big = torch.randn(16, 180, 320, 64, device="cuda").to(torch.bfloat16); bigd = torch.empty_like(big)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
iters = int(os.environ.get("ITERS", "60"))
for shape in ((2, 7, 1, 36, 68), (4, 7, 1, 64, 96), (2, 7, 1, 180, 320)):
    x = torch.from_numpy(np.random.RandomState(sum(shape)).rand(*shape).astype(np.float32)).cuda()
    with torch.no_grad():
        ref = m(x).clone(); ref2 = m(x).clone()
        assert torch.equal(ref, ref2)
        bad = 0
        for it in range(iters):
            torch.cuda.synchronize()
            with torch.cuda.stream(sb):
                for _ in range(12):
                    hip.conv2d_mfma([dict(srcs=[big], dst=bigd)], wp, 3, 64, hip.BF16, act=hip.ACT_LEAKY, slope=0.1)
            with torch.cuda.stream(sa):
                y = m(x)
            torch.cuda.synchronize()
            bad += int(not torch.equal(y, ref))
    print(f"shape {shape}: outputs differing from the solo result under a co-running conv: {bad} of {iters}", flush=True)
