"""Determinism probe: repeated forwards (eager and graph replay) must be bit-identical."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
shape = tuple(int(v) for v in os.environ.get("SHAPE", "2,7,1,36,68").split(","))
m = A.GShiftNet_S(); m.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"))); m = m.cuda()
m.precision = os.environ.get("PREC", "bf16"); m.streams = int(os.environ.get("STREAMS", "2"))
x = torch.from_numpy(np.random.RandomState(sum(shape)).rand(*shape).astype(np.float32)).cuda()
for graph in (False, True):
    m.use_graph = graph
    with torch.no_grad():
        ys = [m(x).clone() for _ in range(5)]
    torch.cuda.synchronize()
    d = [float((y - ys[-1]).abs().max()) for y in ys]
    print(f"graph={graph}: max |y_i - y_last| = {d}")
    if graph is False: ref = ys[-1]
print("eager vs graph:", float((ref - ys[-1]).abs().max()))
