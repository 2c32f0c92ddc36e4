"""Convolution with HIP kernels in all three directions (forward, input gradient, weight gradient) as a torch.autograd
Function.  Replaces nn.Conv2d's forward AND backward on the training path (reference: every nn.Conv2d of
CVSR_train/arch/CVSR_freq.py under `loss.backward()`, train_LD_freqCVSR_S_22.py:244-251).

Tensors are (B,C,H,W) in torch's channels_last memory format, i.e. NHWC in memory, the layout of the HIP kernels: the
(B,H,W,C) permutation handed to the C ABI is a zero-copy view.

precision "f32": exact-f32 kernels in all directions (the mode the gradient goldens are checked in).
precision "bf16"/"f16": forward and input gradient on the matrix cores (fcvsr_conv2d_mfma, f32 accumulate) when the layer is
eligible (1x1 / 3x3, channel counts the MFMA path takes); weight gradient with bf16 products on the matrix cores for the
3x3 / 1x1 layers with multiples of 64 channels (fcvsr_conv2d_wgrad_mfma), exact f32 for the rest.

Round 3: a layer is ONE forward launch and at most five backward launches.  Bias and LeakyReLU / ReLU ride in the forward
kernel's epilogue (`act`, `slope`); the activation's backward is one elementwise launch on the saved OUTPUT; the bias gradient is
a two-stage column sum; the 16-bit operand packings of the weight (forward and transposed + tap-flipped for the input
gradient) are one launch each, cached per parameter version - they were chains of 4-8 small torch kernels per layer and step.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch

from .. import hip

_MMA = {"bf16": (hip.BF16, torch.bfloat16), "f16": (hip.F16, torch.float16)}
_ACT = {None: hip.ACT_NONE, "none": hip.ACT_NONE, "relu": hip.ACT_RELU, "lrelu": hip.ACT_LEAKY}

# packed 16-bit weights per (storage pointer, parameter version, dtype, transposed), valid for ONE forward + backward pass
# (graph.forward_train clears it): within a pass a layer that is applied several times packs once
_PACKED: Dict[Tuple, torch.Tensor] = {}
_PACKED_MAX = 4096


# While True (TrainStep sets it around loss.backward()), the weight / bias gradient of a layer whose parameter already owns a dense f32
# `.grad` (a view of the step's flat, pre-zeroed gradient buffer) is ADDED into that tensor by the reduction kernel itself and the
# autograd Function returns None for it: no AccumulateGrad addition per parameter and pass.  Off by default: plain
# `loss.backward()` / `torch.autograd.grad` see ordinary gradients.
ACCUMULATE_INTO_GRAD = False


class accumulate_into_grad:
    """Context manager: let the HIP reductions add parameter gradients straight into existing `.grad` tensors."""

    def __enter__(self):
        global ACCUMULATE_INTO_GRAD
        self._old, ACCUMULATE_INTO_GRAD = ACCUMULATE_INTO_GRAD, True

    def __exit__(self, *exc):
        global ACCUMULATE_INTO_GRAD
        ACCUMULATE_INTO_GRAD = self._old


def _grad_sink(p: Optional[torch.Tensor]):
    """The parameter's `.grad` if the reductions may add into it in place, else None."""
    if not ACCUMULATE_INTO_GRAD or p is None or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or g.device != p.device:
        return None
    return g


# The packings one pass asked for (leaf parameters only: their storage is stable), recorded during a pass and replayed as ONE launch
# at the start of the next ones (fcvsr_pack_weights_mfma_multi): items = [(weight, dtype, transposed, packed tensor)], table on the device.
_PLAN = None                         # dict(items=..., tabs={dtype: (table tensor, n_items, total_blocks)}, device=...)
_RECORD = []                         # requests of the pass in progress (while no plan exists)
_PLAN_ENABLED = os.environ.get("FCVSR_PACK_PLAN", "1") == "1"


def _build_plan(items):
    dev = items[0][0].device
    be = hip.lib().fcvsr_pack_weights_multi_block_elems()
    tabs = {}
    for tdt in set(it[1] for it in items):
        rows, blk = [], 0
        for (w, t, transposed, out) in items:
            if t != tdt:
                continue
            cout, cin, kh, kw = w.shape
            kk, rp, cp = out.shape
            rows.append([w.data_ptr(), out.data_ptr(), cout, cin, kk, rp, cp, int(transposed), blk])
            blk += (kk * rp * cp + be - 1) // be
        tabs[tdt] = (torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), blk)
    return dict(items=items, tabs=tabs, device=dev, ptrs=[it[0].data_ptr() for it in items])


_OWNER = None                        # whose passes _RECORD / _PLAN belong to (forward_train passes a key of its parameter set)


def clear_packed_weights(owner=None) -> None:
    """Start of a differentiable pass (graph.forward_train): forget the cached operand packings; when the previous pass of the SAME
    parameter set left a plan (same storage), re-pack all of its weights from their current values in one launch."""
    global _PLAN, _RECORD, _OWNER
    _PACKED.clear()
    if not _PLAN_ENABLED:
        return
    if owner != _OWNER:                                   # another model: drop the plan (and its references), record afresh
        _PLAN, _RECORD, _OWNER = None, [], owner
        return
    if _PLAN is None and _RECORD and not torch.cuda.is_current_stream_capturing():
        _PLAN = _build_plan(_RECORD)                      # (uploads the table: never inside a capture)
    _RECORD = []
    if _PLAN is None:
        return
    items = _PLAN["items"]
    if any(w.data_ptr() != ptr or w.dtype != torch.float32 or not w.is_contiguous() or w.device != _PLAN["device"]
           for (w, _, _, _), ptr in zip(items, _PLAN["ptrs"])):
        _PLAN = None                                      # parameters moved: record again during this pass
        return
    L = hip.lib()
    for tdt, (tab, n, blocks) in _PLAN["tabs"].items():
        hip.check(L.fcvsr_pack_weights_mfma_multi(tab.data_ptr(), n, blocks, hip._DT[tdt], hip.stream_ptr()), "fcvsr_pack_weights_mfma_multi")
    for (w, tdt, transposed, out) in items:
        _PACKED[(w.data_ptr(), w._version, tdt, transposed, tuple(w.shape), str(w.device))] = out


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) any layout -> contiguous (B,H,W,C) view of a channels_last tensor."""
    return t.permute(0, 2, 3, 1).contiguous()           # no copy when t is already channels_last


def packed_weight_mfma(w: torch.Tensor, tdt: torch.dtype, transposed: bool) -> torch.Tensor:
    """16-bit operand layout of fcvsr_conv2d_mfma for `w` (or for the input-gradient convolution), one HIP launch, cached."""
    key = (w.data_ptr(), w._version, tdt, transposed, tuple(w.shape), str(w.device))
    got = _PACKED.get(key)
    if got is not None:
        return got
    if len(_PACKED) >= _PACKED_MAX:
        _PACKED.clear()
    cout, cin, kh, kw = w.shape
    rows, cols = (cin, cout) if transposed else (cout, cin)
    rp, cp = (rows + 127) // 128 * 128, (cols + 63) // 64 * 64
    out = torch.empty((kh * kw, rp, cp), dtype=tdt, device=w.device)
    wd = w.detach()
    if wd.dtype != torch.float32 or not wd.is_contiguous():
        wd = wd.float().contiguous()
    hip.check(hip.lib().fcvsr_pack_weight_mfma(wd.data_ptr(), cout, cin, kh, kw, out.data_ptr(), rp, cp, hip._DT[tdt], int(transposed),
                                               hip.stream_ptr()), "fcvsr_pack_weight_mfma")
    _PACKED[key] = out
    if _PLAN_ENABLED and _PLAN is None and w.is_leaf and wd.data_ptr() == w.data_ptr():
        _RECORD.append((w, tdt, transposed, out))         # a parameter in stable storage: part of the next passes' one-launch packing
    return out


def _run_conv(x_nhwc: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int, precision: str, *,
              transposed: bool = False, act: int = hip.ACT_NONE, slope: float = 0.0) -> torch.Tensor:
    """x (B,H,W,Cin) f32, w (Cout,Cin,k,k) [transposed: the input-gradient convolution with w^T, taps flipped] -> (B,Ho,Wo,Cout) f32
    ("same" padding k//2), bias and activation in the kernel's epilogue."""
    cout, cin, k, _ = w.shape
    if transposed:
        cout, cin = cin, cout
    B, H, W, _ = x_nhwc.shape
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x_nhwc.device)
    b = None if bias is None else bias.detach()
    if b is not None and (b.dtype != torch.float32 or not b.is_contiguous()):
        b = b.float().contiguous()
    if precision in _MMA and k in (1, 3) and stride == 1 and cin % 4 == 0 and (cout % 4 == 0 or cout < 4):
        mdt, tdt = _MMA[precision]
        co4, dst, b4 = cout, out, b
        if cout < 4:             # conv_last0 (64 -> 1 at 4H x 4W): run as a 4-channel layer, the packed weight rows past cout are zero
            co4 = 4
            dst = torch.empty((B, Ho, Wo, 4), dtype=torch.float32, device=x_nhwc.device)
            b4 = None if b is None else torch.nn.functional.pad(b, (0, 4 - cout))
        groups = [dict(srcs=[x_nhwc], dst=dst)]
        if hip.mfma_eligible(k, stride, groups):
            hip.conv2d_mfma(groups, packed_weight_mfma(w, tdt, transposed), k, co4, mdt, bias=b4, act=act, slope=slope)
            return dst if co4 == cout else dst[..., :cout]
    wl = w.detach()
    if transposed:
        wl = wl.permute(1, 0, 2, 3).flip(2, 3).contiguous()
    wm = hip.pack_conv_weight_f32mfma(wl) if (k in (1, 3) and stride == 1 and cin % 32 == 0 and cout % 4 == 0) else None
    hip.conv2d([x_nhwc], hip.pack_conv_weight(wl), k, cout, out, bias=b, stride=stride, w_f32mfma=wm, act=act, slope=slope)
    return out


def _colsum(g_nhwc: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sum over (b, y, x) of a dense (B,H,W,C) f32 tensor: the bias gradient (written, or - in accumulate mode - added, to `out`)."""
    Cn = g_nhwc.shape[3]
    npix = g_nhwc.numel() // Cn
    L = hip.lib()
    n = L.fcvsr_colsum_scratch_elems(npix, Cn)
    scratch = torch.empty(n, dtype=torch.float32, device=g_nhwc.device)
    if out is None:
        out = torch.empty(Cn, dtype=torch.float32, device=g_nhwc.device)
    hip.check(L.fcvsr_colsum(g_nhwc.data_ptr(), npix, Cn, out.data_ptr(), scratch.data_ptr(), n, hip.stream_ptr()), "fcvsr_colsum")
    return out


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, precision, act, slope):
        xv = _nhwc(x.float())
        out = _run_conv(xv, w, bias, stride, precision, act=act, slope=slope)
        ctx.save_for_backward(xv, w, out if act != hip.ACT_NONE else None, bias)
        ctx.stride, ctx.precision, ctx.has_bias, ctx.act, ctx.slope = stride, precision, bias is not None, act, slope
        return out.permute(0, 3, 1, 2)                    # (B,Cout,Ho,Wo), channels_last in memory

    @staticmethod
    def backward(ctx, gy):
        xv, w, y, bias = ctx.saved_tensors
        stride, precision = ctx.stride, ctx.precision
        cout, cin, k, _ = w.shape
        B, H, W, _ = xv.shape
        gyv = _nhwc(gy.float())
        L = hip.lib()
        if ctx.act != hip.ACT_NONE:                       # gradient at the pre-activation, from the saved output
            gp = torch.empty_like(gyv)
            hip.check(L.fcvsr_act_bwd(gyv.data_ptr(), y.data_ptr(), gp.data_ptr(), ctx.slope if ctx.act == hip.ACT_LEAKY else 0.0,
                                      gyv.numel(), hip.stream_ptr()), "fcvsr_act_bwd")
            gyv = gp
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dL/dx = "same" stride-1 convolution of dL/dy (zero-inserted for stride 2) with the transposed, tap-flipped weight
            g_in = gyv
            if stride != 1:
                g_in = torch.zeros((B, H, W, cout), dtype=torch.float32, device=gyv.device)
                g_in[:, ::stride, ::stride, :][:, :gyv.shape[1], :gyv.shape[2]] = gyv
            gx = _run_conv(g_in, w, None, 1, precision, transposed=True).permute(0, 3, 1, 2)
        sink = _grad_sink(w) if ctx.needs_input_grad[1] else None      # add into the flat gradient buffer in place?
        fused_bias, gb_f, bsink_f = False, None, None
        if sink is not None:
            L.fcvsr_wgrad_set_accumulate(1)
            L.fcvsr_colsum_set_accumulate(1)
        try:
            if ctx.needs_input_grad[1] and cout == 1 and k == 3 and stride == 1 and cin in (16, 32, 64):
                # one output channel (conv_last0): x is read once, 9 x cin accumulators per thread
                n = L.fcvsr_wgrad_cout1_scratch_elems(B, H, cin)
                scratch = torch.empty(n, dtype=torch.float32, device=xv.device)
                gw = sink if sink is not None else torch.empty((1, cin, 3, 3), dtype=torch.float32, device=xv.device)
                hip.check(L.fcvsr_wgrad_cout1(xv.data_ptr(), gyv.data_ptr(), B, H, W, cin, gw.data_ptr(), scratch.data_ptr(), n, hip.stream_ptr()),
                          "fcvsr_wgrad_cout1")
            elif ctx.needs_input_grad[1]:
                Ho, Wo = gyv.shape[1], gyv.shape[2]
                # 16-bit modes: products on the matrix cores for the 3x3 / 1x1 layers with multiples of 64 channels; exact f32 otherwise
                mm = precision in _MMA and L.fcvsr_conv2d_wgrad_mfma_eligible(cin, cout, k, k, stride, k // 2)
                n = (L.fcvsr_conv2d_wgrad_mfma_scratch_elems if mm else L.fcvsr_conv2d_wgrad_scratch_elems)(B, Ho, Wo, cin, cout, k, k)
                scratch = torch.empty(n, dtype=torch.float32, device=xv.device)
                gw = sink if sink is not None else torch.empty((cout, cin, k, k), dtype=torch.float32, device=xv.device)
                xd, gd = hip.view(xv), hip.view(gyv)
                fn = L.fcvsr_conv2d_wgrad_mfma if mm else L.fcvsr_conv2d_wgrad
                if mm and ctx.has_bias and ctx.needs_input_grad[2]:
                    # the matrix-core kernel has every gy tile in registers: it also sums gy's columns (the bias gradient)
                    bsink_f = _grad_sink(bias)
                    gb_f = bsink_f if bsink_f is not None else torch.empty(cout, dtype=torch.float32, device=xv.device)
                    fused_bias = bool(L.fcvsr_wgrad_set_bias_out(gb_f.data_ptr(), 1 if bsink_f is not None else 0))
                hip.check(fn(C.addressof(xd), C.addressof(gd), B, H, W, k, k, stride, k // 2, gw.data_ptr(), scratch.data_ptr(), n,
                             hip.stream_ptr()), "fcvsr_conv2d_wgrad")
        finally:
            if sink is not None:
                L.fcvsr_wgrad_set_accumulate(0)
                L.fcvsr_colsum_set_accumulate(0)
                gw = None                                             # already in w.grad
        if fused_bias:
            gb = None if bsink_f is not None else gb_f
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            bsink = _grad_sink(bias)
            if bsink is not None:
                L.fcvsr_colsum_set_accumulate(1)
                try:
                    _colsum(gyv, out=bsink)
                finally:
                    L.fcvsr_colsum_set_accumulate(0)
            else:
                gb = _colsum(gyv)
        return gx, gw, gb, None, None, None, None


def conv2d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, precision: str = "f32",
           act: Optional[str] = None, slope: float = 0.0) -> torch.Tensor:
    """act(nn.Conv2d(k, stride, padding=k//2)(x)) with HIP forward / input-gradient / weight-gradient kernels; `act` in
    {None, "relu", "lrelu"} (slope) is evaluated in the forward kernel's epilogue and differentiated from the saved output.
    CUDA (HIP) tensors only: there is no CPU fallback."""
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train.conv2d needs device tensors (the HIP path has no CPU fallback)")
    return _Conv2dFn.apply(x, w, bias, stride, precision, _ACT[act], float(slope))


class _ConvLevelsFn(torch.autograd.Function):
    """One nn.Conv2d applied to several tensors of different sizes (the pyramid levels of a BlockRCB layer, reference
    CVSR_freq.py:766-777) as ONE forward launch, ONE input-gradient launch, one matrix-core weight-gradient launch per level with a
    single ordered reduction, and one bias-gradient reduction - instead of three independent layers whose gradients autograd adds.
    16-bit modes, stride 1, cin and cout multiples of 64 (the wrapper checks)."""

    @staticmethod
    def forward(ctx, w, bias, precision, act, slope, *xs):
        mdt, tdt = _MMA[precision]
        cout, cin, k, _ = w.shape
        xvs = [_nhwc(x.float()) for x in xs]
        outs = [torch.empty((xv.shape[0], xv.shape[1], xv.shape[2], cout), dtype=torch.float32, device=xv.device) for xv in xvs]
        b = None if bias is None else bias.detach().float().contiguous()
        hip.conv2d_mfma([dict(srcs=[xv], dst=o) for xv, o in zip(xvs, outs)], packed_weight_mfma(w, tdt, False), k, cout, mdt, bias=b,
                        act=act, slope=slope)
        ctx.save_for_backward(w, bias, *xvs, *(outs if act != hip.ACT_NONE else []))
        ctx.n, ctx.precision, ctx.has_bias, ctx.act, ctx.slope = len(xs), precision, bias is not None, act, slope
        return tuple(o.permute(0, 3, 1, 2) for o in outs)

    @staticmethod
    def backward(ctx, *gys):
        n = ctx.n
        w, bias = ctx.saved_tensors[0], ctx.saved_tensors[1]
        xvs = ctx.saved_tensors[2:2 + n]
        ys = ctx.saved_tensors[2 + n:]
        mdt, tdt = _MMA[ctx.precision]
        cout, cin, k, _ = w.shape
        L = hip.lib()
        st = hip.stream_ptr()
        gvs = []
        for i, gy in enumerate(gys):
            gv = _nhwc(gy.float())
            if ctx.act != hip.ACT_NONE:
                gp = torch.empty_like(gv)
                hip.check(L.fcvsr_act_bwd(gv.data_ptr(), ys[i].data_ptr(), gp.data_ptr(), ctx.slope if ctx.act == hip.ACT_LEAKY else 0.0,
                                          gv.numel(), st), "fcvsr_act_bwd")
                gv = gp
            gvs.append(gv)
        gxs = [None] * n
        if any(ctx.needs_input_grad[5:]):
            gx = [torch.empty_like(xv) for xv in xvs]
            hip.conv2d_mfma([dict(srcs=[gv], dst=o) for gv, o in zip(gvs, gx)], packed_weight_mfma(w, tdt, True), k, cin, mdt)
            gxs = [o.permute(0, 3, 1, 2) for o in gx]
        gw = gb = None
        fused_bias, gb_f, bsink_f = False, None, None
        if ctx.needs_input_grad[0]:
            Bs = (C.c_int * n)(*[xv.shape[0] for xv in xvs])
            Hs = (C.c_int * n)(*[xv.shape[1] for xv in xvs])
            Ws = (C.c_int * n)(*[xv.shape[2] for xv in xvs])
            ne = L.fcvsr_conv2d_wgrad_mfma_groups_scratch_elems(Bs, Hs, Ws, n, cin, cout, k, k)
            scratch = torch.empty(ne, dtype=torch.float32, device=w.device)
            sink = _grad_sink(w)
            gw = sink if sink is not None else torch.empty((cout, cin, k, k), dtype=torch.float32, device=w.device)
            xd = (hip.View * n)(*[hip.view(xv) for xv in xvs])
            gd = (hip.View * n)(*[hip.view(gv) for gv in gvs])
            L.fcvsr_wgrad_set_accumulate(1 if sink is not None else 0)
            if ctx.has_bias and ctx.needs_input_grad[1]:
                bsink_f = _grad_sink(bias)
                gb_f = bsink_f if bsink_f is not None else torch.empty(cout, dtype=torch.float32, device=w.device)
                fused_bias = bool(L.fcvsr_wgrad_set_bias_out(gb_f.data_ptr(), 1 if bsink_f is not None else 0))
            try:
                hip.check(L.fcvsr_conv2d_wgrad_mfma_groups(xd, gd, Bs, Hs, Ws, n, k, k, k // 2, gw.data_ptr(), scratch.data_ptr(), ne, st),
                          "fcvsr_conv2d_wgrad_mfma_groups")
            finally:
                L.fcvsr_wgrad_set_accumulate(0)
            if sink is not None:
                gw = None
        if fused_bias:
            gb = None if bsink_f is not None else gb_f
        elif ctx.has_bias and ctx.needs_input_grad[1]:
            ptrs = (C.c_void_p * n)(*[gv.data_ptr() for gv in gvs])
            npx = (C.c_longlong * n)(*[gv.numel() // cout for gv in gvs])
            ne = L.fcvsr_colsum_groups_scratch_elems(npx, n, cout)
            scratch = torch.empty(ne, dtype=torch.float32, device=w.device)
            bsink = _grad_sink(bias)
            gb = bsink if bsink is not None else torch.empty(cout, dtype=torch.float32, device=w.device)
            L.fcvsr_colsum_set_accumulate(1 if bsink is not None else 0)
            try:
                hip.check(L.fcvsr_colsum_groups(ptrs, npx, n, cout, gb.data_ptr(), scratch.data_ptr(), ne, st), "fcvsr_colsum_groups")
            finally:
                L.fcvsr_colsum_set_accumulate(0)
            if bsink is not None:
                gb = None
        return (gw, gb, None, None, None, *gxs)


def conv2d_levels(xs, w: torch.Tensor, bias: Optional[torch.Tensor] = None, precision: str = "f32", act: Optional[str] = None,
                  slope: float = 0.0):
    """[act(conv(x)) for x in xs] for ONE stride-1 layer applied to up to three tensors (pyramid levels): grouped launches in the 16-bit
    modes when the layer takes the matrix-core path in every direction; otherwise a plain loop over `conv2d`."""
    xs = list(xs)
    cout, cin, k, _ = w.shape
    ok = (precision in _MMA and 1 < len(xs) <= 3 and k in (1, 3) and cin % 64 == 0 and cout % 64 == 0 and all(x.is_cuda for x in xs))
    if ok:
        ok = hip.mfma_eligible(k, 1, [dict(srcs=[_nhwc(x.float())], dst=_nhwc(x.float())) for x in xs])
    if not ok:
        return [conv2d(x, w, bias, 1, precision, act, slope) for x in xs]
    return list(_ConvLevelsFn.apply(w, bias, precision, _ACT[act], float(slope), *xs))
