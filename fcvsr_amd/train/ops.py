"""Convolution with HIP kernels in all three directions (forward, input gradient, weight gradient) as a torch.autograd
Function.  Replaces nn.Conv2d's forward AND backward on the training path (reference: every nn.Conv2d of
CVSR_train/arch/CVSR_freq.py under `loss.backward()`, train_LD_freqCVSR_S_22.py:244-251).

Tensors are (B,C,H,W) in torch's channels_last memory format, i.e. NHWC in memory, the layout of the HIP kernels: the
(B,H,W,C) permutation handed to the C ABI is a zero-copy view.

precision "f32": exact-f32 direct kernels in all directions (the mode the gradient goldens are checked in).
precision "bf16"/"f16": forward and input gradient on the matrix cores (fcvsr_conv2d_mfma, f32 accumulate) when the layer is
eligible (1x1 / 3x3, channel counts the MFMA path takes); weight gradient with bf16 products on the matrix cores for the
3x3 / 1x1 layers with multiples of 64 channels (fcvsr_conv2d_wgrad_mfma), exact f32 for the rest.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import hip

_MMA = {"bf16": (hip.BF16, torch.bfloat16), "f16": (hip.F16, torch.float16)}


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) any layout -> contiguous (B,H,W,C) view of a channels_last tensor."""
    return t.permute(0, 2, 3, 1).contiguous()           # no copy when t is already channels_last


def _run_conv(x_nhwc: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int, precision: str) -> torch.Tensor:
    """x (B,H,W,Cin) f32, w (Cout,Cin,k,k) -> (B,Ho,Wo,Cout) f32 ("same" padding k//2)."""
    cout, cin, k, _ = w.shape
    B, H, W, _ = x_nhwc.shape
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x_nhwc.device)
    b = None if bias is None else bias.detach().float().contiguous()
    if precision in _MMA and k in (1, 3) and stride == 1 and cin % 4 == 0 and cout % 4 == 0:
        mdt, tdt = _MMA[precision]
        groups = [dict(srcs=[x_nhwc], dst=out)]
        if hip.mfma_eligible(k, stride, groups):
            hip.conv2d_mfma(groups, hip.pack_conv_weight_mfma(w, tdt), k, cout, mdt, bias=b)
            return out
    wm = hip.pack_conv_weight_f32mfma(w) if (k in (1, 3) and stride == 1 and cin % 32 == 0 and cout % 4 == 0) else None
    hip.conv2d([x_nhwc], hip.pack_conv_weight(w), k, cout, out, bias=b, stride=stride, w_f32mfma=wm)
    return out


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, precision):
        xv = _nhwc(x.float())
        ctx.save_for_backward(xv, w)
        ctx.stride, ctx.precision, ctx.has_bias = stride, precision, bias is not None
        out = _run_conv(xv, w.detach(), bias, stride, precision)
        return out.permute(0, 3, 1, 2)                    # (B,Cout,Ho,Wo), channels_last in memory

    @staticmethod
    def backward(ctx, gy):
        xv, w = ctx.saved_tensors
        stride, precision = ctx.stride, ctx.precision
        cout, cin, k, _ = w.shape
        B, H, W, _ = xv.shape
        gyv = _nhwc(gy.float())
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dL/dx = "same" stride-1 convolution of dL/dy (zero-inserted for stride 2) with the transposed, tap-flipped weight
            wt = w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous()
            g_in = gyv
            if stride != 1:
                g_in = torch.zeros((B, H, W, cout), dtype=torch.float32, device=gyv.device)
                g_in[:, ::stride, ::stride, :][:, :gyv.shape[1], :gyv.shape[2]] = gyv
            gx = _run_conv(g_in, wt, None, 1, precision).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            L = hip.lib()
            Ho, Wo = gyv.shape[1], gyv.shape[2]
            # 16-bit modes: products on the matrix cores for the 3x3 / 1x1 layers with multiples of 64 channels; exact f32 otherwise
            mm = precision in _MMA and L.fcvsr_conv2d_wgrad_mfma_eligible(cin, cout, k, k, stride, k // 2)
            n = (L.fcvsr_conv2d_wgrad_mfma_scratch_elems if mm else L.fcvsr_conv2d_wgrad_scratch_elems)(B, Ho, Wo, cin, cout, k, k)
            scratch = torch.empty(n, dtype=torch.float32, device=xv.device)
            gw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=xv.device)
            import ctypes as C
            xd, gd = hip.view(xv), hip.view(gyv)
            fn = L.fcvsr_conv2d_wgrad_mfma if mm else L.fcvsr_conv2d_wgrad
            hip.check(fn(C.addressof(xd), C.addressof(gd), B, H, W, k, k, stride, k // 2, gw.data_ptr(), scratch.data_ptr(), n,
                         hip.stream_ptr()), "fcvsr_conv2d_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gyv.sum(dim=(0, 1, 2))
        return gx, gw, gb, None, None


def conv2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, precision: str = "f32") -> torch.Tensor:
    """nn.Conv2d(k, stride, padding=k//2) with HIP forward / input-gradient / weight-gradient kernels.  CUDA (HIP) tensors only:
    there is no CPU fallback."""
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train.conv2d needs device tensors (the HIP path has no CPU fallback)")
    return _Conv2dFn.apply(x, w, bias, stride, precision)
