"""Training step alone (the bench's `train` sub-record workload), for kernel traces: python scripts/one_train.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
from fcvsr_amd.train import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
tm = A.GShiftNet_S()
tm.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"), gain=0.5), strict=True)
tm = tm.to(dev)
tm.train_precision = os.environ.get("PREC", "bf16")
g = torch.Generator().manual_seed(300)
tx = torch.rand(4, 7, 1, 128, 128, generator=g).to(dev)
th = torch.rand(4, 1, 512, 512, generator=g).to(dev)
step = TrainStep(tm, lr=1e-4, weight_decay=1e-5, use_graph=os.environ.get("GRAPH", "0") == "1")
step(tx, th); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): lv = step(tx, th)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.1f} ms per step, loss {lv:.3f}")
