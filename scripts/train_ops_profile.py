"""Which torch operators are left in the training step: one eager step under torch.profiler, aten ops grouped by name + input shapes,
sorted by device time (python scripts/train_ops_profile.py [rows])."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from torch.profiler import profile, ProfilerActivity
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
from fcvsr_amd.train import TrainStep
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
tm = A.GShiftNet_S()
tm.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"), gain=0.5), strict=True)
tm = tm.to(dev)
tm.train_precision = os.environ.get("PREC", "bf16")
g = torch.Generator().manual_seed(300)
tx = torch.rand(4, 7, 1, 128, 128, generator=g).to(dev)
th = torch.rand(4, 1, 512, 512, generator=g).to(dev)
step = TrainStep(tm, lr=1e-4, weight_decay=1e-5, use_graph=False)
step(tx, th); step(tx, th); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=os.environ.get("STACK", "0") == "1") as prof:
    step(tx, th)
    torch.cuda.synchronize()
def dev_us(e):
    return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
stack = os.environ.get("STACK", "0") == "1"
evs = prof.key_averages(group_by_stack_n=8) if stack else prof.key_averages(group_by_input_shape=True)
evs = sorted((e for e in evs if e.key.startswith("aten::") and dev_us(e) > 0), key=dev_us, reverse=True)
print(f"aten operators with device time: {sum(dev_us(e) for e in evs) / 1e3:.2f} ms in {sum(e.count for e in evs)} calls")
for e in evs[:rows]:
    if stack:
        fr = [f for f in e.stack if "fcvsr_amd" in f or "bench.py" in f][:3]
        where = " <- ".join(f.split("fcvsr_amd/")[-1] for f in fr)
    else:
        where = str(e.input_shapes)[:150]
    print(f"{e.key:28s} {dev_us(e):9.1f} us {e.count:5d}  {where}")
