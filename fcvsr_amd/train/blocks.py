"""Fused training blocks: sub-graphs of the forward whose forward AND backward are hand-written HIP kernels wrapped as
torch.autograd Functions (the convolutions live in `ops.py`).  Each replaces a chain of device-side torch operators of
`graph.py` and is tested against that chain (tests/test_train_gpu.py)."""
from __future__ import annotations

import torch

from .. import hip


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    return t.permute(0, 2, 3, 1).contiguous()           # no copy when t is channels_last


class _RcbTailFn(torch.autograd.Function):
    """R = LeakyReLU_0.2(ContextBlock(r)) + z (reference CVSR_freq.py:657-701, :722-725): three forward and four backward launches."""

    @staticmethod
    def forward(ctx, r, z, wmask, w1, w2, slope):
        rv, zv = _nhwc(r.float()), _nhwc(z.float())
        B, H, W, Cn = rv.shape
        L = hip.lib()
        nblk = L.fcvsr_rcbt_nblk(H * W)
        out = torch.empty_like(rv)
        stats = torch.empty((B, L.fcvsr_rcbt_stat_elems()), dtype=torch.float32, device=rv.device)
        scratch = torch.empty(B * nblk * 66, dtype=torch.float32, device=rv.device)
        wm, a1, a2 = (t.detach().float().contiguous() for t in (wmask, w1, w2))
        hip.check(L.fcvsr_rcbt_forward(rv.data_ptr(), zv.data_ptr(), wm.data_ptr(), a1.data_ptr(), a2.data_ptr(), slope, B, H * W, Cn,
                                       out.data_ptr(), stats.data_ptr(), scratch.data_ptr(), scratch.numel(), hip.stream_ptr()),
                  "fcvsr_rcbt_forward")
        ctx.save_for_backward(rv, wm, a1, a2, stats, wmask, w1, w2)
        ctx.slope, ctx.shapes = slope, (wmask.shape, w1.shape, w2.shape)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        rv, wm, a1, a2, stats, p_wm, p_w1, p_w2 = ctx.saved_tensors
        gv = _nhwc(g.float())
        B, H, W, Cn = rv.shape
        L = hip.lib()
        nblk = L.fcvsr_rcbt_nblk(H * W)
        gr = torch.empty_like(rv)
        from .ops import _grad_sink
        sinks = [_grad_sink(p) for p in (p_wm, p_w1, p_w2)]
        inplace = all(s_ is not None for s_ in sinks)                      # add into the flat gradient buffer (ops.accumulate_into_grad)
        if inplace:
            dwm, dw1, dw2 = sinks
        else:
            dwm = torch.empty(Cn, dtype=torch.float32, device=rv.device)
            dw1 = torch.empty(Cn * Cn, dtype=torch.float32, device=rv.device)
            dw2 = torch.empty(Cn * Cn, dtype=torch.float32, device=rv.device)
        n = B * nblk * Cn + B * (Cn + 1) + 4 + 2 * B * Cn * Cn
        scratch = torch.empty(n, dtype=torch.float32, device=rv.device)
        hip.check(L.fcvsr_rcbt_backward(rv.data_ptr(), gv.data_ptr(), wm.data_ptr(), a1.data_ptr(), a2.data_ptr(), stats.data_ptr(), ctx.slope,
                                        B, H * W, Cn, gr.data_ptr(), dwm.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), scratch.data_ptr(), n,
                                        int(inplace), hip.stream_ptr()), "fcvsr_rcbt_backward")
        if inplace:
            return gr.permute(0, 3, 1, 2), g, None, None, None, None
        s0, s1, s2 = ctx.shapes
        return gr.permute(0, 3, 1, 2), g, dwm.reshape(s0), dw1.reshape(s1), dw2.reshape(s2), None


def rcb_tail(r: torch.Tensor, z: torch.Tensor, wmask: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, slope: float = 0.2) -> torch.Tensor:
    """LeakyReLU_slope(r + W2 . LeakyReLU_slope(W1 . softmaxpool(r))) + z for 64-channel device tensors (B,64,H,W)."""
    if not r.is_cuda:
        raise RuntimeError("fcvsr_amd.train.blocks needs device tensors (the HIP path has no CPU fallback)")
    return _RcbTailFn.apply(r, z, wmask, w1, w2, float(slope))


class _DivEnhBandFn(torch.autograd.Function):
    """One DivEnh band (i >= 1) of MultiFreq_Refinment and its running sums (reference CVSR_freq.py:2104-2133 at :2201-2254):
    (Sf, So) -> (Sf + f, So + e1 CA(e1) + e2 CA(e2)), three forward and four backward launches (train_mffr.hip)."""

    @staticmethod
    def forward(ctx, f, sf, so, a, b, w1, w2):
        fv, sfv, sov = (_nhwc(t.float()) for t in (f, sf, so))
        B, H, W, Cn = fv.shape
        L = hip.lib()
        nblk = L.fcvsr_divenh_band_nblk(H * W)
        sf_out, so_out = torch.empty_like(fv), torch.empty_like(fv)
        stats = torch.empty((B, L.fcvsr_divenh_band_stat_elems(Cn)), dtype=torch.float32, device=fv.device)
        scratch = torch.empty(B * nblk * 2 * Cn, dtype=torch.float32, device=fv.device)
        av, bv, a1, a2 = (t.detach().float().contiguous() for t in (a, b, w1, w2))
        hip.check(L.fcvsr_divenh_band_forward(fv.data_ptr(), sfv.data_ptr(), sov.data_ptr(), av.data_ptr(), bv.data_ptr(), a1.data_ptr(),
                                              a2.data_ptr(), B, H * W, Cn, sf_out.data_ptr(), so_out.data_ptr(), stats.data_ptr(),
                                              scratch.data_ptr(), scratch.numel(), hip.stream_ptr()), "fcvsr_divenh_band_forward")
        ctx.save_for_backward(fv, sfv, sov, av, bv, a1, a2, stats, a, b, w1, w2)
        return sf_out.permute(0, 3, 1, 2), so_out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gsf, gso):
        fv, sfv, sov, av, bv, a1, a2, stats, p_a, p_b, p_w1, p_w2 = ctx.saved_tensors
        B, H, W, Cn = fv.shape
        CR = Cn // 16
        hv = _nhwc(gsf.float()) if gsf is not None else torch.zeros_like(fv)
        ov = _nhwc(gso.float()) if gso is not None else torch.zeros_like(fv)
        L = hip.lib()
        nblk = L.fcvsr_divenh_band_nblk(H * W)
        gf, gsf_o, gso_o = torch.empty_like(fv), torch.empty_like(fv), torch.empty_like(fv)
        from .ops import _grad_sink
        sinks = [_grad_sink(p) for p in (p_a, p_b, p_w1, p_w2)]
        inplace = all(s_ is not None for s_ in sinks)
        if inplace:
            ga, gb, dw1, dw2 = sinks
        else:
            ga = torch.empty(Cn, dtype=torch.float32, device=fv.device)
            gb = torch.empty(Cn, dtype=torch.float32, device=fv.device)
            dw1 = torch.empty(CR * Cn, dtype=torch.float32, device=fv.device)
            dw2 = torch.empty(Cn * CR, dtype=torch.float32, device=fv.device)
        n = B * nblk * 2 * Cn + 2 * B * Cn + 2 * B * Cn * CR
        scratch = torch.empty(n, dtype=torch.float32, device=fv.device)
        hip.check(L.fcvsr_divenh_band_backward(fv.data_ptr(), sfv.data_ptr(), sov.data_ptr(), av.data_ptr(), bv.data_ptr(), a1.data_ptr(),
                                               a2.data_ptr(), stats.data_ptr(), hv.data_ptr(), ov.data_ptr(), B, H * W, Cn, gf.data_ptr(),
                                               gsf_o.data_ptr(), gso_o.data_ptr(), ga.data_ptr(), gb.data_ptr(), dw1.data_ptr(),
                                               dw2.data_ptr(), scratch.data_ptr(), n, int(inplace), hip.stream_ptr()),
                  "fcvsr_divenh_band_backward")
        outs = tuple(t.permute(0, 3, 1, 2) for t in (gf, gsf_o, gso_o))
        if inplace:
            return outs + (None, None, None, None)
        return outs + (ga.reshape(p_a.shape), gb.reshape(p_b.shape), dw1.reshape(p_w1.shape), dw2.reshape(p_w2.shape))


def divenh_band(f, sf, so, a, b, w1, w2):
    """(Sf', So') of one DivEnh band i >= 1 for device tensors (B,C,H,W), C in {32, 64}; a, b: C values; w1 (C/16,C,1,1), w2 (C,C/16,1,1)."""
    if not f.is_cuda:
        raise RuntimeError("fcvsr_amd.train.blocks needs device tensors (the HIP path has no CPU fallback)")
    return _DivEnhBandFn.apply(f, sf, so, a, b, w1, w2)


class _IacFn(torch.autograd.Function):
    """IAC of BOTH alignment directions (reference CVSR_freq.py:1230-1250 called at :1524-1545): A iterations of
    LeakyReLU(SAC_h(SAC_v(flow_warp(feat, off_i), K1_i), K1_i) + feat_in) per direction, the two directions sharing the predicted
    kernels K.  Forward: fcvsr_warp / fcvsr_sac_v / fcvsr_sac_h (3 launches per iteration and direction); backward: two launches per
    iteration and direction (fcvsr_iac_bwd_sac, fcvsr_iac_bwd_warp) + one fill.  inputs: feat_f, feat_b (B,C,H,W), K (B,A*6C,H,W),
    then the A forward and the A backward offset fields (B,2,H,W)."""

    @staticmethod
    def forward(ctx, feat_f, feat_b, K, A, slope, *offs):
        import ctypes as C_
        L = hip.lib()
        st = hip.stream_ptr()
        Kv = _nhwc(K.float())
        B, H, W, KC = Kv.shape
        Cn = KC // (6 * A)
        saved, outs = [], []
        offv = [o.float().permute(0, 2, 3, 1) for o in offs]               # views (any strides)
        for d, feat in enumerate((feat_f, feat_b)):
            fin = _nhwc(feat.float())
            prev = fin
            fv = hip.view(fin)
            for i in range(A):
                k1 = hip.view(Kv[..., i * 6 * Cn: i * 6 * Cn + 3 * Cn])
                ov = hip.view(offv[d * A + i])
                s, v, y = torch.empty_like(fin), torch.empty_like(fin), torch.empty_like(fin)
                pv, sv, vv, yv = hip.view(prev), hip.view(s), hip.view(v), hip.view(y)
                hip.check(L.fcvsr_warp(C_.byref(pv), C_.byref(ov), B, H, W, C_.byref(sv), st), "fcvsr_warp")
                hip.check(L.fcvsr_sac_v(C_.byref(sv), C_.byref(k1), B, H, W, C_.byref(vv), st), "fcvsr_sac_v")
                hip.check(L.fcvsr_sac_h(C_.byref(vv), C_.byref(k1), C_.byref(fv), slope, B, H, W, C_.byref(yv), st), "fcvsr_sac_h")
                saved += [prev, s, v, y]
                prev = y
            outs.append(prev.permute(0, 3, 1, 2))
        ctx.save_for_backward(Kv, *offv, *saved)
        ctx.A, ctx.slope, ctx.Cn = A, slope, Cn
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, g_f, g_b):
        import ctypes as C_
        L = hip.lib()
        st = hip.stream_ptr()
        A, slope, Cn = ctx.A, ctx.slope, ctx.Cn
        Kv = ctx.saved_tensors[0]
        offv = ctx.saved_tensors[1:1 + 2 * A]
        saved = ctx.saved_tensors[1 + 2 * A:]
        B, H, W, KC = Kv.shape
        gK = torch.zeros_like(Kv)                                           # the never-read F2 halves keep a zero gradient
        g_feats, g_offs = [], [None] * (2 * A)
        for d, g_out in enumerate((g_f, g_b)):
            g = _nhwc(g_out.float())
            gfin = torch.empty_like(g)
            for i in reversed(range(A)):
                prev, s, v, y = saved[(d * A + i) * 4:(d * A + i) * 4 + 4]
                k1 = hip.view(Kv[..., i * 6 * Cn: i * 6 * Cn + 3 * Cn])
                gk1 = hip.view(gK[..., i * 6 * Cn: i * 6 * Cn + 3 * Cn])
                gv = torch.empty_like(g)
                hip.check(L.fcvsr_iac_bwd_sac(g.data_ptr(), y.data_ptr(), v.data_ptr(), s.data_ptr(), C_.byref(k1), slope, B, H, W, Cn,
                                              gfin.data_ptr(), int(i != A - 1), gv.data_ptr(), C_.byref(gk1), int(d == 1), st), "fcvsr_iac_bwd_sac")
                gprev = torch.zeros_like(g)
                goff = torch.empty((B, H, W, 2), dtype=torch.float32, device=g.device)
                ov = hip.view(offv[d * A + i])
                hip.check(L.fcvsr_iac_bwd_warp(gv.data_ptr(), C_.byref(k1), prev.data_ptr(), C_.byref(ov), B, H, W, Cn, gprev.data_ptr(),
                                               goff.data_ptr(), st), "fcvsr_iac_bwd_warp")
                g_offs[d * A + i] = goff.permute(0, 3, 1, 2)
                g = gprev
            g_feats.append((gfin + g).permute(0, 3, 1, 2))                 # prev of iteration 0 is feat_in itself
        return (g_feats[0], g_feats[1], gK.permute(0, 3, 1, 2), None, None, *g_offs)


def iac_both(feat_f: torch.Tensor, feat_b: torch.Tensor, K: torch.Tensor, offs_f, offs_b, slope: float = 0.1):
    """Both alignment directions of IAC (A = len(offs_f) iterations) with the kernel predictor output K shared; C in {32, 64}."""
    if not feat_f.is_cuda:
        raise RuntimeError("fcvsr_amd.train.blocks needs device tensors (the HIP path has no CPU fallback)")
    A = len(offs_f)
    return _IacFn.apply(feat_f, feat_b, K, A, float(slope), *offs_f, *offs_b)


class _PReluFn(torch.autograd.Function):
    """nn.PReLU() with one shared slope (reference CVSR_freq.py:2590, ConvBlk :349) on dense tensors: one forward launch, two backward."""

    @staticmethod
    def forward(ctx, x, slope):
        xc = x.float()
        if not (xc.is_contiguous() or xc.is_contiguous(memory_format=torch.channels_last)):
            xc = xc.contiguous()
        y = torch.empty_like(xc)                                  # same (dense) strides: an elementwise map of the buffer
        a = slope.detach().float().reshape(-1)
        hip.check(hip.lib().fcvsr_prelu_fwd(xc.data_ptr(), a.data_ptr(), y.data_ptr(), xc.numel(), hip.stream_ptr()), "fcvsr_prelu_fwd")
        ctx.save_for_backward(xc, a)
        ctx.sshape = slope.shape
        return y

    @staticmethod
    def backward(ctx, g):
        xc, a = ctx.saved_tensors
        gc = g.float()
        if gc.stride() != xc.stride():
            gc = torch.empty_like(xc).copy_(gc)                    # the same memory order as x
        gx = torch.empty_like(xc)
        ga = torch.empty(1, dtype=torch.float32, device=xc.device)
        scratch = torch.empty(2048, dtype=torch.float32, device=xc.device)
        hip.check(hip.lib().fcvsr_prelu_bwd(gc.data_ptr(), xc.data_ptr(), a.data_ptr(), gx.data_ptr(), ga.data_ptr(), scratch.data_ptr(),
                                            xc.numel(), hip.stream_ptr()), "fcvsr_prelu_bwd")
        return gx, ga.reshape(ctx.sshape)


def prelu(x: torch.Tensor, slope: torch.Tensor) -> torch.Tensor:
    """PReLU with a single learnable slope (numel % 4 == 0, 16-byte aligned dense tensor); device tensors only."""
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train.blocks needs device tensors (the HIP path has no CPU fallback)")
    if slope.numel() != 1 or x.numel() % 4:
        return torch.nn.functional.prelu(x, slope.reshape(-1))
    return _PReluFn.apply(x, slope)


class _XscaleFn(torch.autograd.Function):
    """One level of BlockRCB's cross-scale sum (reference CVSR_freq.py:766-777): out = x + r_scale * R + avgpool2(dn) + bilinear_up2(up)
    (dn at twice, up at half the resolution; either may be absent) - fcvsr_xscale forward, the two resampling adjoints backward."""

    @staticmethod
    def forward(ctx, x, R, r_scale, dn, up):
        xv, Rv = _nhwc(x.float()), _nhwc(R.float())
        dv = None if dn is None else _nhwc(dn.float())
        uv = None if up is None else _nhwc(up.float())
        B, H, W, Cn = xv.shape
        out = torch.empty_like(xv)
        hip.check(hip.lib().fcvsr_xscale(xv.data_ptr(), Rv.data_ptr(), r_scale, hip.ptr(dv), hip.ptr(uv), out.data_ptr(), hip.F32, B, H, W, Cn,
                                         hip.stream_ptr()), "fcvsr_xscale")
        ctx.r_scale, ctx.has = r_scale, (dn is not None, up is not None)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        gv = _nhwc(g.float())
        B, H, W, Cn = gv.shape
        L = hip.lib()
        gR = g if ctx.r_scale == 1.0 else g * ctx.r_scale
        gdn = gup = None
        if ctx.has[0]:
            o = torch.empty((B, 2 * H, 2 * W, Cn), dtype=torch.float32, device=gv.device)
            hip.check(L.fcvsr_pool2_adjoint(gv.data_ptr(), o.data_ptr(), B, H, W, Cn, hip.stream_ptr()), "fcvsr_pool2_adjoint")
            gdn = o.permute(0, 3, 1, 2)
        if ctx.has[1]:
            o = torch.empty((B, H // 2, W // 2, Cn), dtype=torch.float32, device=gv.device)
            hip.check(L.fcvsr_up2_adjoint(gv.data_ptr(), o.data_ptr(), B, H // 2, W // 2, Cn, hip.stream_ptr()), "fcvsr_up2_adjoint")
            gup = o.permute(0, 3, 1, 2)
        return g, gR, None, gdn, gup


def xscale(x: torch.Tensor, R: torch.Tensor, r_scale: float, dn=None, up=None) -> torch.Tensor:
    """x + r_scale * R + avgpool2(dn) + bilinear_up2(up) on (B,C,H,W) device tensors, H and W even, C % 4 == 0."""
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train.blocks needs device tensors (the HIP path has no CPU fallback)")
    return _XscaleFn.apply(x, R, float(r_scale), dn, up)


class _CorrLookupFn(torch.autograd.Function):
    """CorrBlock lookup on the un-updated integer grid (reference :1279-1337): a bounds-checked gather from the raw-viewed
    product x1f * x2f / sqrt(C), non-zero only on the columns x <= radius + 1.  fcvsr_corr_lookup on the strip forward, one scatter-free
    kernel backward (was: product, NCHW copy, index gather, mask multiply and an index_put backward with a device sort)."""

    @staticmethod
    def forward(ctx, x1f, x2f, radius):
        import ctypes as C_
        a, b = _nhwc(x1f.float()), _nhwc(x2f.float())
        B, H, Wf, Cn = a.shape
        n = 2 * radius + 1
        out = torch.zeros((B, H, Wf, n * n), dtype=torch.float32, device=a.device)
        xw = min(Wf, radius + 2)
        ov = hip.view(out)
        hip.check(hip.lib().fcvsr_corr_lookup(a.data_ptr(), b.data_ptr(), Cn, B, H, Wf, Cn, radius, xw, C_.byref(ov), hip.stream_ptr()),
                  "fcvsr_corr_lookup")
        ctx.save_for_backward(a, b)
        ctx.radius, ctx.xw = radius, xw
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        import ctypes as C_
        a, b = ctx.saved_tensors
        B, H, Wf, Cn = a.shape
        gv = g.float().permute(0, 2, 3, 1)
        if gv.stride(3) != 1:
            gv = gv.contiguous()
        ga, gb = torch.zeros_like(a), torch.zeros_like(b)
        gview = hip.view(gv)
        hip.check(hip.lib().fcvsr_corr_lookup_bwd(a.data_ptr(), b.data_ptr(), Cn, B, H, Wf, Cn, ctx.radius, ctx.xw, C_.byref(gview), ga.data_ptr(),
                                                  gb.data_ptr(), hip.stream_ptr()), "fcvsr_corr_lookup_bwd")
        return ga.permute(0, 3, 1, 2), gb.permute(0, 3, 1, 2), None


def corr_lookup(x1f: torch.Tensor, x2f: torch.Tensor, radius: int = 4) -> torch.Tensor:
    if not x1f.is_cuda:
        raise RuntimeError("fcvsr_amd.train.blocks needs device tensors (the HIP path has no CPU fallback)")
    return _CorrLookupFn.apply(x1f, x2f, radius)
