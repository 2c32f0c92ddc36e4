"""One data-parallel training step and the epoch loop around it (counterpart of reference
CVSR_train/train_LD_freqCVSR_S_22.py:183-266 on one process per GPU).

Frame windows are independent samples, so the batch shards over ranks (clip data parallel); the only collective of a step is
ONE all-reduce of the flat f32 gradient buffer (3.70 M / 8.81 M elements = 14.8 / 35.2 MB for S / full) over RCCL - a single
large message per step suits xGMI's point-to-point links better than per-layer buckets at these sizes.  The CVSR_train
loss is a SUM over the batch (opt/loss.py:20-31), so ranks add their gradients (op SUM); mmedit's mean loss averages.
Parameters that never receive a gradient (the never-called `DivEnh.Conv`) are left out of the buffer, so no
"unused parameter" machinery is needed (the reference's mmedit configs set find_unused_parameters=True for them).
"""
from __future__ import annotations

import os
import random
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .loss import charbonnier_loss
from .ops import accumulate_into_grad


def trainable_parameters(model: torch.nn.Module) -> List[Tuple[str, torch.nn.Parameter]]:
    """Unique parameters that take part in the forward, in registration order (aliases `body.3.*` / `RCB.*` counted once)."""
    out = []
    for name, p in model.named_parameters():                     # named_parameters de-duplicates shared Parameters
        if p.requires_grad and ".DivEnh_block." in name and ".Conv." in name:
            continue                                             # reference CVSR_freq.py:2108: constructed, never called
        if p.requires_grad:
            out.append((name, p))
    return out


class FlatGradAllReduce:
    """Every gradient lives in ONE contiguous f32 buffer: `param.grad` of every trainable parameter is a view of it, so zeroing the
    gradients is one fill, the all-reduce needs no packing, and nothing is unpacked (round 2 copied 303 tensors in and out per step)."""

    def __init__(self, params: Sequence[torch.nn.Parameter], op: str = "sum", group=None):
        if op not in ("sum", "mean"):
            raise ValueError("op must be 'sum' (CVSR_train sum loss) or 'mean' (mmedit mean loss)")
        self.params = list(params)
        self.op, self.group = op, group
        self.sizes = [p.numel() for p in self.params]
        self.numel = int(sum(self.sizes))
        self.flat: Optional[torch.Tensor] = None

    def bind(self) -> torch.Tensor:
        """(Re)attach `.grad` of every parameter to its slice of the flat buffer (idempotent; follows `model.to(device)`)."""
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        for v, p in zip(self.flat.split(self.sizes), self.params):
            if p.grad is None:
                p.grad = v.view(p.shape)
            elif p.grad.data_ptr() != v.data_ptr():              # a gradient produced elsewhere (plain autograd): adopt its values
                v.copy_(p.grad.reshape(-1))
                p.grad = v.view(p.shape)
        return self.flat

    def zero(self) -> None:
        self.bind().zero_()

    def __call__(self) -> torch.Tensor:
        flat = self.bind()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if flat.is_cuda and dist.get_backend(self.group) == "gloo":         # CPU rehearsals of the N>1 path
                host = flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(host)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)   # backend "nccl" is RCCL on ROCm
            if self.op == "mean":
                flat.div_(dist.get_world_size(self.group))
        return flat


class TrainStep:
    """optimizer.zero_grad(); sr = model(lr); loss = Charbonnier(sr, hr); loss.backward(); all-reduce; optimizer.step()."""

    def __init__(self, model: torch.nn.Module, *, lr: float = 1e-4, weight_decay: float = 1e-5,
                 loss_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor] = charbonnier_loss, reduce_op: str = "sum",
                 optimizer: Optional[torch.optim.Optimizer] = None, group=None, use_graph: bool = False):
        """use_graph: capture forward + loss + backward of the first batch shape in a hipGraph and replay it (the step is ~4 k
        small launches; replaying removes their host cost).  Gradients then live in static buffers; the all-reduce and the
        optimizer step stay eager.  Batches must keep the captured shape (a different shape is captured anew)."""
        self.model = model
        self.use_graph = use_graph
        self._graphs = {}
        named = trainable_parameters(model)
        self.names = [n for n, _ in named]
        params = [p for _, p in named]
        # reference defaults: Adam(lr=1e-4, weight_decay=1e-5) (train_LD_freqCVSR_S_22.py:35,42,204)
        # (torch's fused=True Adam does not advance the parameters' version counters, which the inference engine's packed-weight
        # cache is keyed on: the default multi-tensor implementation stays)
        self.optimizer = optimizer or torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
        self.loss_fn = loss_fn
        self.allreduce = FlatGradAllReduce(params, reduce_op, group)

    def local_backward(self, lr_frames: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
        """This rank's forward + loss + backward (no collective): gradients end up in `.grad`.  Split from the collective
        half so that a caller can agree on success across ranks BEFORE anybody enters the all-reduce (bench.py does)."""
        if self.use_graph and lr_frames.is_cuda:
            return self._graphed(lr_frames, hr)
        self.allreduce.zero()                              # one fill: every .grad is a view of the flat buffer
        sr = self.model(lr_frames)
        loss = self.loss_fn(sr, hr)
        with accumulate_into_grad():                       # the HIP reductions add weight / bias gradients straight into it
            loss.backward()
        return loss

    def reduce_and_update(self) -> None:
        """The collective half: ONE all-reduce of the flat gradient buffer, then the (replicated) optimizer step."""
        self.allreduce()
        self.optimizer.step()

    def __call__(self, lr_frames: torch.Tensor, hr: torch.Tensor) -> float:
        """lr_frames: (b, 7, C, h, w), hr: (b, C, 4h, 4w) - this rank's share of the batch."""
        loss = self.local_backward(lr_frames, hr)
        self.reduce_and_update()
        return float(loss.detach())

    def _graphed(self, lr_frames: torch.Tensor, hr: torch.Tensor) -> torch.Tensor:
        key = (tuple(lr_frames.shape), tuple(hr.shape), str(lr_frames.device))
        ent = self._graphs.get(key)
        if ent is None:
            sx, sh = lr_frames.clone(), hr.clone()
            params = self.allreduce.params
            # warm-up on a side stream (allocations, weight packing, per-kernel attributes, per-device zero page), as
            # torch.cuda.graphs requires; leaves .grad tensors allocated so the capture accumulates into static buffers
            side = torch.cuda.Stream(lr_frames.device)
            side.wait_stream(torch.cuda.current_stream(lr_frames.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.allreduce.zero()
                    with accumulate_into_grad():
                        self.loss_fn(self.model(sx), sh).backward()
            torch.cuda.current_stream(lr_frames.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.allreduce.flat.zero_()                  # gradients accumulate into the static flat buffer
                loss = self.loss_fn(self.model(sx), sh)
                with accumulate_into_grad():
                    loss.backward()
            ent = self._graphs[key] = (graph, sx, sh, loss)
        graph, sx, sh, loss = ent
        sx.copy_(lr_frames)
        sh.copy_(hr)
        graph.replay()
        return loss


# ---- data transforms of the reference loader (CVSR_train/opt/data_LD_LR.py:248-344), on {lr_imgs (f,h,w), hr_imgs (f',4h,4w)} ----
def random_crop(sample: Dict[str, np.ndarray], size: int = 128, rng=np.random) -> Dict[str, np.ndarray]:
    lr, hr = sample["lr_imgs"], sample["hr_imgs"]
    h, w = lr.shape[1:]
    top, left = rng.randint(0, h - size), rng.randint(0, w - size)           # high end exclusive, as in the reference
    return dict(sample, lr_imgs=lr[:, top:top + size, left:left + size],
                hr_imgs=hr[:, top * 4:(top + size) * 4, left * 4:(left + size) * 4])


def augment(sample: Dict[str, np.ndarray], rng=random) -> Dict[str, np.ndarray]:
    lr, hr = sample["lr_imgs"], sample["hr_imgs"]
    hflip, vflip, rot90 = rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.5
    if hflip:
        lr, hr = lr[:, :, ::-1], hr[:, :, ::-1]
    if vflip:
        lr, hr = lr[:, ::-1, :], hr[:, ::-1, :]
    if rot90:
        lr, hr = lr.transpose(0, 2, 1), hr.transpose(0, 2, 1)
    return dict(sample, lr_imgs=lr.copy(), hr_imgs=hr.copy())


def to_tensor(sample: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    """uint8 (f,h,w) -> float (1,f,h,w) / 255 (channel axis first, as the reference's ToTensor)."""
    return {"lr_imgs": torch.from_numpy(sample["lr_imgs"][np.newaxis]).float() / 255.0,
            "hr_imgs": torch.from_numpy(sample["hr_imgs"][np.newaxis]).float() / 255.0}


def fit(model: torch.nn.Module, batches: Callable[[int], Iterable[Dict[str, torch.Tensor]]], *, epochs: int, device,
        lr: float = 1e-4, weight_decay: float = 1e-5, milestones: Sequence[int] = (2000, 8000, 12000, 20000), gamma: float = 0.5,
        val_itv: int = 1, ckpt_dir: Optional[str] = None, warm_start_epoch: int = 0, log: Callable[[str], None] = print) -> List[float]:
    """Epoch loop of the reference (train_LD_freqCVSR_S_22.py:239-266): MultiStepLR stepped at the START of every epoch,
    Charbonnier-sum loss, Adam, `epoch-%d.pth` state_dict checkpoints every `val_itv` epochs (rank 0).
    `batches(epoch)` yields {'lr_imgs': (b,C,7,h,w), 'hr_imgs': (b,C,f',4h,4w)} like the reference DataLoader."""
    step = TrainStep(model, lr=lr, weight_decay=weight_decay)
    sched = torch.optim.lr_scheduler.MultiStepLR(step.optimizer, milestones=list(milestones), gamma=gamma)
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    model.train()
    history: List[float] = []
    for epoch in range(warm_start_epoch, epochs):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                      # the reference steps the scheduler before the optimizer
            sched.step()
        losses = []
        for data in batches(epoch):
            frames = data["lr_imgs"].permute(0, 2, 1, 3, 4).to(device)      # (b, frames, chn, h, w)
            hr = data["hr_imgs"].to(device)[:, :, 0]
            losses.append(step(frames, hr))
        avg = round(sum(losses) / max(1, len(losses)), 5)
        history.append(avg)
        log("Epoch: %d/%d | average epoch loss: %f" % (epoch + 1, epochs, avg))
        if (epoch + 1) % val_itv == 0 and ckpt_dir is not None and rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            torch.save(model.state_dict(), os.path.join(ckpt_dir, "epoch-%d.pth" % (epoch + 1 + warm_start_epoch)))
    return history
