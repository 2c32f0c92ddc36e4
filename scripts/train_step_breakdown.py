"""Wall-clock split of one TrainStep call (hipGraph mode): graph replay of forward + backward, gradient all-reduce (world 1: none),
optimizer.step(), with a device synchronisation after each part."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from fcvsr_amd.arch import CVSR_freq as A
from fcvsr_amd.arch.schema import state_dict_shapes
from fcvsr_amd.weights import synthetic_state_dict
from fcvsr_amd.train import TrainStep
dev = torch.device("cuda:0")
tm = A.GShiftNet_S()
tm.load_state_dict(synthetic_state_dict(state_dict_shapes("GShiftNet_S"), gain=0.5), strict=True)
tm = tm.to(dev); tm.train_precision = "bf16"
g = torch.Generator().manual_seed(300)
tx = torch.rand(4, 7, 1, 128, 128, generator=g).to(dev); th = torch.rand(4, 1, 512, 512, generator=g).to(dev)
step = TrainStep(tm, lr=1e-4, weight_decay=1e-5, use_graph=True)
for _ in range(3): step(tx, th)
torch.cuda.synchronize()
ta = tb = tc = 0.0; n = 10
for _ in range(n):
    t0 = time.perf_counter(); loss = step.local_backward(tx, th); torch.cuda.synchronize()
    t1 = time.perf_counter(); step.allreduce(); torch.cuda.synchronize()
    t2 = time.perf_counter(); step.optimizer.step(); torch.cuda.synchronize()
    t3 = time.perf_counter(); ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
print(f"graph replay fwd+bwd {ta / n * 1e3:.2f} ms | all-reduce {tb / n * 1e3:.2f} ms | optimizer.step {tc / n * 1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(n): step(tx, th)
torch.cuda.synchronize()
print(f"whole step {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
