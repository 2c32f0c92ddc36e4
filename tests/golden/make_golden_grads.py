#!/usr/bin/env python3
"""Gradient goldens: parameter gradients of the reference's training objective, computed by the REFERENCE model itself
(build container only; same read-only import and stubs as make_golden.py).

Objective = the Charbonnier SUM loss of CVSR_train/opt/loss.py:20-31 (sum(sqrt(diff^2 + 1e-4)); restated here in one line
because `opt.loss` imports the absent `pytorch_wavelets` at module level) between `model(x)` and a seeded target, exactly the
`sr = model(frames); loss = CharbonnierLoss(sr, hr); loss.backward()` of train_LD_freqCVSR_S_22.py:247-250.

Outputs (data only):
  grad_Sreduced_24x16.npz - reduced config GShiftNet_S(n_features=32, ACNum=2, Freq_Inv=2, SCGroupN=1), batch 2:
                            x, target, loss and the FULL gradient of every parameter (347 780 values).
  grad_S_16x20.npz        - default GShiftNet_S, batch 1: x, target, loss and per-parameter signatures
                            (L2 norm, sum, 32 sampled entries at key-seeded positions) - the full gradient would be 15 MB.
"""
import json
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from make_golden import import_reference                      # noqa: E402
from fcvsr_amd.weights import synthetic_state_dict            # noqa: E402


def sample_positions(key: str, numel: int, n: int = 32) -> np.ndarray:
    rs = np.random.RandomState(zlib.crc32(("grad:" + key).encode()) & 0x7fffffff)
    return rs.randint(0, numel, size=min(n, numel)).astype(np.int64)


def run(ref, name, ctor, kwargs, x, target, full):
    torch.manual_seed(0)
    model = getattr(ref, ctor)(**kwargs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synthetic_state_dict(shapes, gain=0.5), strict=True)
    model.train()
    sr = model(torch.from_numpy(x))
    diff = sr - torch.from_numpy(target)
    loss = torch.sum(torch.sqrt(diff * diff + 1e-4))
    loss.backward()
    out = {"x": x, "target": target, "loss": np.float64(loss.item())}
    named = dict(model.state_dict(keep_vars=True))
    n_none = 0
    for k, p in named.items():
        if not isinstance(p, torch.nn.Parameter):
            continue
        if p.grad is None:
            n_none += 1
            out["none:" + k] = np.zeros(0, dtype=np.float32)
            continue
        g = p.grad.detach().numpy().astype(np.float32)
        if full:
            out["grad:" + k] = g
        else:
            flat = g.reshape(-1).astype(np.float64)
            out["sig:" + k] = np.concatenate([[np.sqrt((flat ** 2).sum()), flat.sum()], flat[sample_positions(k, flat.size)]])
    meta = dict(ctor=ctor, kwargs=kwargs, gain=0.5, full=bool(full))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss), "params without grad:", n_none)


def main():
    ref = import_reference()
    rs = np.random.RandomState(4321)
    x = rs.rand(2, 7, 1, 24, 16).astype(np.float32)
    t = rs.rand(2, 1, 96, 64).astype(np.float32)
    run(ref, "grad_Sreduced_24x16", "GShiftNet_S", dict(n_features=32, ACNum=2, Freq_Inv=2, SCGroupN=1), x, t, True)
    x = rs.rand(1, 7, 1, 16, 20).astype(np.float32)
    t = rs.rand(1, 1, 64, 80).astype(np.float32)
    run(ref, "grad_S_16x20", "GShiftNet_S", {}, x, t, False)


if __name__ == "__main__":
    main()
