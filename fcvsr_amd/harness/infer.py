"""Sequence inference harness: the counterpart of the reference's evaluation loop
(CVSR_train/test_LD_freqCVSR_S_22.py:48-123) around the drop-in model.

For every output frame i of a sequence: take the 7 LR frames window_indices(i) (edge replicate by default), pad rows to a
multiple of 4 the way the reference pads 270 -> 272 (zero rows appended at the bottom, :24-26), run the model on batches
of windows, crop the padding off the SR frame (:81-86), clamp to [0,1], scale by 255 and TRUNCATE to uint8 (:88-89;
mmedit's tensor2img rounds instead - `quantise="round"`).  Frames are independent, so windows are batched and, across
GPUs, sharded by `fcvsr_amd.harness.sharding`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import numpy as np
import torch

from .metrics import psnr
from .windows import window_indices


def pad_to_multiple(frames: torch.Tensor, mult: int = 4) -> torch.Tensor:
    """(N,C,H,W) -> zero-padded at the bottom/right so that H, W are multiples of `mult`."""
    H, W = frames.shape[-2:]
    ph, pw = (-H) % mult, (-W) % mult
    if ph == 0 and pw == 0:
        return frames
    return torch.nn.functional.pad(frames, (0, pw, 0, ph))


@torch.no_grad()
def super_resolve_sequence(model, lr: torch.Tensor, *, num_frames: int = 7, padding: str = "replicate", batch: int = 8,
                           centres: Optional[Iterable[int]] = None, quantise: str = "truncate") -> np.ndarray:
    """lr: (N,C,H,W) float in [0,1] (host or device).  Returns uint8 (len(centres),C,4H,4W) SR frames."""
    N, C, H, W = lr.shape
    dev = next(model.parameters()).device
    x = pad_to_multiple(lr.float(), 4).to(dev)
    centres = list(range(N)) if centres is None else list(centres)
    out: List[np.ndarray] = []
    for s in range(0, len(centres), batch):
        idx = [window_indices(i, num_frames, N, padding) for i in centres[s:s + batch]]
        win = torch.stack([x[j] for j in idx], 0)                 # (b, 7, C, Hp, Wp)
        sr = model(win)[:, :, :4 * H, :4 * W]
        sr = sr.clamp(0, 1) * 255.0
        sr = sr.round() if quantise == "round" else sr            # uint8 cast truncates
        out.append(sr.to(torch.uint8).cpu().numpy())
    return np.concatenate(out, 0)


def sequence_psnr(sr_u8: np.ndarray, hr_u8: np.ndarray, crop_border: int = 4) -> float:
    """Mean per-frame PSNR over a sequence, first channel, borders cropped (reference metric/psnr_ssim.py:447-485)."""
    vals = [psnr(a[0], b[0], crop_border) for a, b in zip(sr_u8, hr_u8)]
    return float(np.mean(vals))


def sequence_ssim(sr_u8: np.ndarray, hr_u8: np.ndarray, crop_border: int = 4) -> float:
    """Mean per-frame SSIM over a sequence, first channel, borders cropped (reference test_LD_freqCVSR_S_22.py:109-118)."""
    from .metrics import ssim
    vals = [ssim(a[0], b[0], crop_border) for a, b in zip(sr_u8, hr_u8)]
    return float(np.mean(vals))


class StreamedSuperResolver:
    """Streams several LR sequences through one GPU in fixed-size batches of 7-frame windows (BASELINE config 5: REDS4-shaped
    100-frame sequences, clip-parallel over ranks; counterpart of the per-frame loop of reference
    CVSR_train/test_LD_freqCVSR_S_22.py:66-91 with the sequence list of anna_file/REDS4_GT.txt).

    * this rank's share of the flattened frame list comes from `harness.sharding.shard_sequences` (contiguous ranges; no
      data-path collective), so a rank reads only its ranges plus a 3-frame halo;
    * frames stay on the host (pinned); each batch of windows is gathered on the host, copied on a side stream while the
      previous batch is in the model (double buffering: two pinned staging buffers, two device buffers, two events), and the
      quantised SR frames come back into a preallocated pinned uint8 buffer - the 288 GB of HBM are not needed for the
      sequences themselves, only two batches are resident;
    * the last, partial batch is padded to the full batch size so that every call has the same shape (one hipGraph / one
      set of cached buffers in the engine), padded outputs are dropped.
    """

    def __init__(self, model, *, num_frames: int = 7, padding: str = "replicate", batch: int = 8, quantise: str = "truncate"):
        self.model, self.num_frames, self.padding, self.batch, self.quantise = model, num_frames, padding, batch, quantise
        self.device = next(model.parameters()).device

    def plan(self, seq_lens, rank: int = 0, world: int = 1):
        """[(seq, centre)] of this rank, in order."""
        from .sharding import shard_sequences
        return [(s, i) for (s, a, b) in shard_sequences(list(seq_lens), rank, world) for i in range(a, b)]

    @torch.no_grad()
    def run(self, sequences, rank: int = 0, world: int = 1):
        """sequences: list of (N_s, C, H, W) float tensors in [0,1] (host; all of one frame size).
        Returns {seq: (first_centre, uint8 array (n, C, 4H, 4W))} for the frames this rank owns."""
        seq_lens = [int(s.shape[0]) for s in sequences]
        work = self.plan(seq_lens, rank, world)
        if not work:
            return {}
        C, H, W = sequences[0].shape[1:]
        for s in sequences:
            if tuple(s.shape[1:]) != (C, H, W):
                raise ValueError("all sequences of one run must share the frame size")
        padded = [pad_to_multiple(s.float(), 4) for s in sequences]
        Hp, Wp = padded[0].shape[-2:]
        B, T = self.batch, self.num_frames
        on_gpu = self.device.type == "cuda"
        pin = dict(pin_memory=True) if on_gpu else {}
        stage = [torch.empty((B, T, C, Hp, Wp), dtype=torch.float32, **pin) for _ in range(2)]
        dev_in = [torch.empty((B, T, C, Hp, Wp), dtype=torch.float32, device=self.device) for _ in range(2)]
        out_host = torch.empty((len(work), C, 4 * H, 4 * W), dtype=torch.uint8, **pin)
        copy_stream = torch.cuda.Stream(self.device) if on_gpu else None
        ready = [torch.cuda.Event() for _ in range(2)] if on_gpu else None
        consumed = [torch.cuda.Event() for _ in range(2)] if on_gpu else None
        batches = [work[i:i + B] for i in range(0, len(work), B)]

        def fill(bi):
            buf = stage[bi & 1]
            for k in range(B):
                s, c = batches[bi][min(k, len(batches[bi]) - 1)]          # pad the last batch with its own last window
                idx = window_indices(c, T, seq_lens[s], self.padding)
                for t, j in enumerate(idx):
                    buf[k, t].copy_(padded[s][j])
            if on_gpu:
                with torch.cuda.stream(copy_stream):
                    if bi >= 2:
                        copy_stream.wait_event(consumed[bi & 1])          # the model has read this device buffer
                    dev_in[bi & 1].copy_(buf, non_blocking=True)
                    ready[bi & 1].record(copy_stream)
            else:
                dev_in[bi & 1].copy_(buf)

        fill(0)
        pos = 0
        for bi, items in enumerate(batches):
            if on_gpu:
                torch.cuda.current_stream(self.device).wait_event(ready[bi & 1])
            if bi + 1 < len(batches):
                if on_gpu and bi >= 1:
                    ready[(bi + 1) & 1].synchronize()                     # host staging buffer of batch bi-1 has been copied
                fill(bi + 1)                                              # overlaps the model call below
            sr = self.model(dev_in[bi & 1])[:, :, :4 * H, :4 * W]
            if on_gpu:
                consumed[bi & 1].record(torch.cuda.current_stream(self.device))
            sr = sr.clamp(0, 1) * 255.0
            sr = sr.round() if self.quantise == "round" else sr
            n = len(items)
            out_host[pos:pos + n].copy_(sr[:n].to(torch.uint8), non_blocking=on_gpu)
            pos += n
        if on_gpu:
            torch.cuda.synchronize(self.device)
        res, arr, k = {}, out_host.numpy(), 0
        for (s, a, b) in __import__("fcvsr_amd.harness.sharding", fromlist=["shard_sequences"]).shard_sequences(seq_lens, rank, world):
            res[s] = (a, arr[k:k + (b - a)].copy())
            k += b - a
        return res
