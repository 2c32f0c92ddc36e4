import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd import hip
dt, mdt = torch.bfloat16, hip.BF16
L3 = [(180, 320), (90, 160), (45, 80)]
B, cin, cout = int(os.environ.get("B", "16")), int(os.environ.get("CIN", "64")), int(os.environ.get("COUT", "64"))
w = torch.randn(cout, cin, 3, 3, device="cuda") / 24
wp = hip.pack_conv_weight_mfma(w, dt)
io = torch.bfloat16 if os.environ.get("IO16", "1") == "1" else torch.float32       # default: the bench's dominant variant
groups = [dict(srcs=[torch.randn(B, H, W, cin, device="cuda").to(io)], dst=torch.empty(B, H, W, cout, device="cuda", dtype=io)) for H, W in L3]
for _ in range(5):
    hip.conv2d_mfma(groups, wp, 3, cout, mdt, act=hip.ACT_LEAKY, slope=0.1)
torch.cuda.synchronize()
