"""Which torch operators / runtime copies are left in the inference step: one eager single-stream forward under torch.profiler,
aten ops grouped by name + input shapes, sorted by device time (python scripts/infer_ops_profile.py [rows])."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from torch.profiler import profile, ProfilerActivity
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
m = A.GShiftNet_S()
m.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"), gain=0.5), strict=True)
m = m.to(dev)
m.precision, m.streams, m.use_graph = "bf16", 1, False
x = torch.rand(16, 7, 1, 180, 320, generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    m(x); m(x); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m(x)
        torch.cuda.synchronize()
def dev_us(e):
    return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
evs = sorted((e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and dev_us(e) > 0), key=dev_us, reverse=True)
print(f"aten operators with device time: {sum(dev_us(e) for e in evs) / 1e3:.3f} ms in {sum(e.count for e in evs)} calls")
for e in evs[:rows]:
    print(f"{e.key:28s} {dev_us(e):9.1f} us {e.count:5d}  {str(e.input_shapes)[:150]}")
