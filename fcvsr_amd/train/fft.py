"""Frequency transforms of the training path on the library's own NHWC kernels (fcvsr_rfft2 / fcvsr_irfft2 / fcvsr_irfft2_bands),
differentiable through their ADJOINTS - which are the same kernels (reference: the torch.fft calls of CVSR_freq.py:1452-1465,
:1497-1505, :2082-2090 under `loss.backward()`):

  r2c (unnormalised) of a real image keeps the half spectrum kx <= W/2.  Its adjoint applied to a gradient G on that half spectrum is
      gx = Re sum_{k in half} G_k e^{+i theta} = N * c2r(G / w),   w = 1 on the self-conjugate columns kx in {0, W/2}, 2 elsewhere
  (c2r = irfft2 with norm 1/N, which counts the interior columns twice and reads only the real part of the edge columns after its
  column pass); the adjoint of c2r is (w / N) * r2c.  The band operator x -> c2r(M * r2c(x)) with a real mask M is self-adjoint.

torch.fft works on NCHW-contiguous tensors: around every transform the channels_last activations were transposed (copies of up to
100 MB) and the [imag | real] packing was a strided cat.  These Functions read and write NHWC directly; the packed spectrum falls out
of the kernel's layout (channel c: imaginary part at im_off + c, real part at re_off + c)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from .. import hip


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.permute(0, 2, 3, 1).contiguous()           # no copy when t is channels_last


_COLW: Dict[Tuple, torch.Tensor] = {}


def _col_weights(H: int, W: int, dev, kind: str) -> torch.Tensor:
    """(H, Wf) or (Wf,) weights of the adjoint identities, cached per size and device (created outside any stream capture:
    the first forward of a shape runs eagerly, as the engine requires for its own tables)."""
    key = (H, W, str(dev), kind)
    t = _COLW.get(key)
    if t is None:
        Wf = W // 2 + 1
        w = torch.full((Wf,), 2.0, dtype=torch.float32)
        w[0] = 1.0
        if W % 2 == 0:
            w[Wf - 1] = 1.0
        if kind == "r2c_adjoint_mask":                   # N / w as the (H, Wf) mask of fcvsr_irfft2
            t = (float(H * W) / w).reshape(1, Wf).expand(H, Wf).contiguous().to(dev)
        else:                                            # w / N per column
            t = (w / float(H * W)).to(dev)
        _COLW[key] = t
    return t


def _rfft2(xv: torch.Tensor, n: int, im_off: int, re_off: int) -> torch.Tensor:
    """xv: (B,H,W,>=n) channel-contiguous view -> dense (B,H,Wf,2n) spectrum with the given imaginary / real channel offsets."""
    B, H, W, _ = xv.shape
    Wf = W // 2 + 1
    spec = torch.empty((B, H, Wf, 2 * n), dtype=torch.float32, device=xv.device)
    v = hip.view(xv)
    v.c = n
    hip.check(hip.lib().fcvsr_rfft2(C.byref(v), B, H, W, n, spec.data_ptr(), 2 * n, im_off, re_off, hip.stream_ptr()), "fcvsr_rfft2")
    return spec


def _irfft2(spec: torch.Tensor, n: int, im_off: int, re_off: int, H: int, W: int, mask=None) -> torch.Tensor:
    """spec: dense (B,H,Wf,2n) -> dense (B,H,W,n), scaled 1/(H*W) (times `mask` (H,Wf) in the column pass)."""
    B = spec.shape[0]
    out = torch.empty((B, H, W, n), dtype=torch.float32, device=spec.device)
    work = torch.empty_like(spec)
    ov = hip.view(out)
    hip.check(hip.lib().fcvsr_irfft2(spec.data_ptr(), 2 * n, im_off, re_off, B, H, W, n, hip.ptr(mask), work.data_ptr(), C.byref(ov),
                                     hip.stream_ptr()), "fcvsr_irfft2")
    return out


class _SpecPackFn(torch.autograd.Function):
    """x (B,n,H,W) -> cat[rfft2(x).imag, rfft2(x).real] (B,2n,H,Wf)  (reference :1452-1465, imaginary parts first)."""

    @staticmethod
    def forward(ctx, x):
        xv = x.float().permute(0, 2, 3, 1)               # channel-contiguous view (a channel slice of a channels_last tensor qualifies)
        if xv.stride(3) != 1:
            xv = xv.contiguous()
        n = xv.shape[3]
        ctx.hw = (xv.shape[1], xv.shape[2], n)
        return _rfft2(xv, n, 0, n).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        H, W, n = ctx.hw
        gv = _nhwc(g.float())
        gx = _irfft2(gv, n, 0, n, H, W, _col_weights(H, W, gv.device, "r2c_adjoint_mask"))
        return gx.permute(0, 3, 1, 2)


class _IrfftPairFn(torch.autograd.Function):
    """o (B,2m,H,Wf) = [real parts (m) | imaginary parts (m)] -> irfft2(real + i imag, s=(H,W)) (B,m,H,W)  (reference :1497-1505)."""

    @staticmethod
    def forward(ctx, o, H, W):
        ov = _nhwc(o.float())
        m = ov.shape[3] // 2
        ctx.hw = (H, W, m)
        return _irfft2(ov, m, m, 0, H, W).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        H, W, m = ctx.hw
        gv = _nhwc(g.float())
        spec = _rfft2(gv, m, m, 0)                       # (B,H,Wf,2m): real parts at 0.., imaginary at m..
        spec = spec * _col_weights(H, W, gv.device, "c2r_adjoint_cols").reshape(1, 1, -1, 1)
        return spec.permute(0, 3, 1, 2), None, None


class _SplitBandsFn(torch.autograd.Function):
    """x (B,n,H,W), masks (Q,H,Wf) real -> Q tensors irfft2(rfft2(x) * masks[q])  (MultiFreq_Refinment's Split_freq, reference
    :2082-2090 with the symmetrised half-spectrum masks): one forward transform, the Q masked inverse transforms in one call; the
    operator is self-adjoint, so the backward is the same pair of kernels per band."""

    @staticmethod
    def forward(ctx, x, masks):
        xv = _nhwc(x.float())
        B, H, W, n = xv.shape
        Q = masks.shape[0]
        Wf = W // 2 + 1
        spec = _rfft2(xv, n, 0, n)
        work = torch.empty((Q, B, H, Wf, 2 * n), dtype=torch.float32, device=xv.device)
        outs = [torch.empty((B, H, W, n), dtype=torch.float32, device=xv.device) for _ in range(Q)]
        bvs = (hip.View * Q)(*[hip.view(o) for o in outs])
        mk = masks.float().contiguous()
        hip.check(hip.lib().fcvsr_irfft2_bands(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, mk.data_ptr(), Q, work.data_ptr(), bvs,
                                               hip.stream_ptr()), "fcvsr_irfft2_bands")
        ctx.save_for_backward(mk)
        ctx.dims = (B, H, W, n, Q)
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        (mk,) = ctx.saved_tensors
        B, H, W, n, Q = ctx.dims
        acc = None
        for q, g in enumerate(gs):
            gv = _nhwc(g.float())
            t = _irfft2(_rfft2(gv, n, 0, n), n, 0, n, H, W, mk[q])
            acc = t if acc is None else acc + t
        return acc.permute(0, 3, 1, 2), None


def spec_pack(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train.fft needs device tensors (the HIP path has no CPU fallback)")
    return _SpecPackFn.apply(x)


def irfft_pair(o: torch.Tensor, H: int, W: int) -> torch.Tensor:
    if not o.is_cuda:
        raise RuntimeError("fcvsr_amd.train.fft needs device tensors (the HIP path has no CPU fallback)")
    return _IrfftPairFn.apply(o, H, W)


def split_bands(x: torch.Tensor, masks: torch.Tensor) -> List[torch.Tensor]:
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train.fft needs device tensors (the HIP path has no CPU fallback)")
    return list(_SplitBandsFn.apply(x, masks))
