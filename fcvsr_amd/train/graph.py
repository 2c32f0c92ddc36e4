"""Differentiable forward of GShiftNet_S / GShiftNet (and the RGB twins) for training: the function the reference trains
through (`sr = model(frames)`; `loss.backward()`, CVSR_train/train_LD_freqCVSR_S_22.py:244-251; forward definition
CVSR_train/arch/CVSR_freq.py:2611-2646 / :2688-2756, specification SURVEY.md Appendix A).

Every nn.Conv2d of the path (98 % of the FLOPs) runs on the HIP kernels in all three directions through
`fcvsr_amd.train.ops.conv2d`; the remaining operators (FFTs, CorrBlock lookup, bilinear warp, separable adaptive 3-tap
convolution, ContextBlock softmax pool, bilinear resampling) are device-side torch operators whose backward torch derives.
This is the training counterpart of `fcvsr_amd.engine` (inference: every operator a hand-written kernel, no autograd); it is
selected automatically by the drop-in modules when gradients are required.  Device tensors only - no CPU fallback.

`p` maps the reference's state_dict keys to the live nn.Parameters of the drop-in module.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from ..engine import band_masks_half
from .blocks import corr_lookup, divenh_band, iac_both, prelu, rcb_tail, xscale
from .fft import irfft_pair, spec_pack, split_bands
from .ops import clear_packed_weights, conv2d, conv2d_levels

Tensor = torch.Tensor


class _Ctx:
    def __init__(self, p: Dict[str, Tensor], precision: str, fused_blocks: bool = True):
        self.p, self.precision = p, precision
        self.fused_blocks = fused_blocks                # False: every non-convolution operator as torch ops (the test reference)

    def conv(self, key: str, x: Tensor, stride: int = 1, act=None, slope: float = 0.0) -> Tensor:
        """nn.Conv2d + optional LeakyReLU / ReLU evaluated in the HIP kernel's epilogue (one launch)."""
        return conv2d(x, self.p[key + ".weight"], self.p.get(key + ".bias"), stride, self.precision, act, slope)

    def conv_padded(self, key: str, parts: List[Tensor], act=None, slope: float = 0.0) -> Tensor:
        """conv(cat(parts)) for a layer whose input-channel count is not a multiple of 64 (upconv_fuse 84, upconv1_L2_2 80, convcorr.0
        211): in the 16-bit modes the concatenation is completed with a zero block and the weight with zero input columns, which makes the
        layer eligible for the matrix-core kernels in all three directions (the exact-f32 VALU weight gradient of such a layer cost
        0.1-0.3 ms).  Same values: the extra products are zeros."""
        w, b = self.p[key + ".weight"], self.p.get(key + ".bias")
        cin = w.shape[1]
        cpad = (-cin) % 64
        if self.precision == "f32" or cpad == 0 or w.shape[2] not in (1, 3):
            return conv2d(torch.cat(parts, 1), w, b, 1, self.precision, act, slope)
        ref = parts[0]
        z = torch.zeros(ref.shape[0], cpad, ref.shape[2], ref.shape[3], dtype=ref.dtype, device=ref.device)
        return conv2d(torch.cat(list(parts) + [z], 1), F.pad(w, (0, 0, 0, 0, 0, cpad)), b, 1, self.precision, act, slope)

    def conv_s2(self, key: str, x: Tensor) -> Tensor:
        """3x3 stride-2 convolution (rconcat1/2, :2594-2595).  16-bit modes: the stride-1 layer on the matrix cores, sub-sampled - 4x
        the (cheap) products, but forward, input gradient and weight gradient all leave the f32 VALU kernels."""
        if self.precision == "f32":
            return self.conv(key, x, stride=2)
        return self.conv(key, x)[:, :, ::2, ::2]

    def chain(self, key: str, t: Tensor, n: int) -> Tensor:
        """n bias-free 1x1 convolutions with ReLU in between (convfuse / convcrt / convcorr, CVSR_freq.py:1371-1396)."""
        for li in range(n):
            t = conv2d(t, self.p[f"{key}.{2 * li}.weight"], None, 1, self.precision, "relu" if li < n - 1 else None)
        return t

    def ca(self, key: str, z: Tensor) -> Tensor:
        """CALayer (:1812-1828): z * sigmoid(W2 relu(W1 mean_HW(z))); the two 1x1 convolutions act on a (B,C) vector."""
        y = z.mean(dim=(2, 3))
        y = F.relu(y @ self.p[key + ".conv_du.0.weight"].flatten(1).t())
        y = torch.sigmoid(y @ self.p[key + ".conv_du.2.weight"].flatten(1).t())
        return z * y[:, :, None, None]


def _lrelu(x: Tensor, s: float) -> Tensor:
    return F.leaky_relu(x, s)                                       # one kernel each way (was a where / mul / compare chain)


def _prelu(x: Tensor, a: Tensor) -> Tensor:
    return prelu(x, a) if _FUSED_PRELU else F.prelu(x, a.reshape(-1))


_FUSED_PRELU = True


def _ps2(x: Tensor) -> Tensor:
    """PixelShuffle(2) (out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]) on the NHWC buffer: ONE copy with contiguous channel runs whose result
    is channels_last again.  F.pixel_shuffle returns an NCHW-contiguous tensor, which cost a 268 MB transposing copy (4.2 ms) in front
    of the next convolution and another one in the backward."""
    B, C4, H, W = x.shape
    c = C4 // 4
    xv = x.permute(0, 2, 3, 1).reshape(B, H, W, c, 2, 2)
    return xv.permute(0, 1, 4, 2, 5, 3).reshape(B, 2 * H, 2 * W, c).permute(0, 3, 1, 2)


def _spec(x: Tensor) -> Tensor:
    X = torch.fft.rfft2(x.contiguous(), norm="backward")
    return torch.cat([X.imag, X.real], dim=1)                       # imag first (:1456-1465)


def _corr_lookup(x1f: Tensor, x2f: Tensor, radius: int = 4) -> Tensor:
    """CorrBlock on the un-updated integer grid = bounds-checked gather from the raw-viewed product (:1279-1337);
    non-zero only for x <= 5, y <= 67."""
    B, C, H, Wf = x1f.shape
    n = 2 * radius + 1
    dev = x1f.device
    img = ((x1f * x2f).contiguous().reshape(B, C * H * Wf) / math.sqrt(float(C))).reshape(B, H * Wf, C // 2, 2)
    ys = torch.arange(H, device=dev).view(1, 1, H, 1)
    xs = torch.arange(Wf, device=dev).view(1, 1, 1, Wf)
    pix = (ys * Wf + xs).expand(n, n, H, Wf)
    d = torch.arange(n, device=dev) - radius
    col = (xs + d.view(n, 1, 1, 1)).expand(n, n, H, Wf)             # output channel c = i * n + j: i shifts the column ...
    row = (ys + d.view(1, n, 1, 1)).expand(n, n, H, Wf)             # ... j the row (meshgrid(dy, dx) added to (x, y), :1303-1309)
    ok = (col >= 0) & (col <= 1) & (row >= 0) & (row <= C // 2 - 1)
    v = img[:, pix, row.clamp(0, C // 2 - 1), col.clamp(0, 1)]      # ONE gather for all 81 planes: (B, n, n, H, Wf)
    return (v * ok.to(v.dtype)).reshape(B, n * n, H, Wf)


def _warp(x: Tensor, off: Tensor) -> Tensor:
    """flow_warp (:1188-1227): bilinear sample of x at (col + off[:,0], row + off[:,1]), zeros outside; differentiable in both."""
    B, C, H, W = x.shape
    dev = x.device
    sx = torch.arange(W, dtype=x.dtype, device=dev).view(1, 1, W) + off[:, 0]
    sy = torch.arange(H, dtype=x.dtype, device=dev).view(1, H, 1) + off[:, 1]
    x0, y0 = torch.floor(sx), torch.floor(sy)
    wx1, wy1 = sx - x0, sy - y0
    flat = x.contiguous().reshape(B, C, H * W)
    out = None
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            xi, yi = (x0 + dx).long(), (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).view(B, 1, H * W).expand(B, C, H * W)
            v = torch.gather(flat, 2, idx).view(B, C, H, W) * (wy * wx * ok.to(x.dtype)).unsqueeze(1)
            out = v if out is None else out + v
    return out


def _sac(s: Tensor, k1: Tensor) -> Tensor:
    """SAC (:1253-1276): vertical then horizontal adaptive 3-tap, replicate padding, kernel1 in BOTH passes (:1272-1273)."""
    B, C, H, W = s.shape
    k = k1.reshape(B, C, 3, H, W)
    sp = F.pad(s, (0, 0, 1, 1), mode="replicate")
    v = sp[:, :, 0:H] * k[:, :, 0] + sp[:, :, 1:H + 1] * k[:, :, 1] + sp[:, :, 2:H + 2] * k[:, :, 2]
    vp = F.pad(v, (1, 1, 0, 0), mode="replicate")
    return vp[..., 0:W] * k[:, :, 0] + vp[..., 1:W + 1] * k[:, :, 1] + vp[..., 2:W + 2] * k[:, :, 2]


def _mgaa(c: _Ctx, key: str, x: Tensor, A: int) -> Tensor:
    """MGAAbk.forward (:1442-1547).  The forward and the backward alignment direction share every weight, so they run STACKED on the
    batch axis (2B) through convfuse / convcorr / the ConvBlk heads / the inverse transforms: half the launches, identical values."""
    B, C3, H, W = x.shape
    d = C3 // 3
    x1, x2, x3 = torch.split(x, d, dim=1)                            # (one cat in the backward instead of three zero-filled slices)
    if c.fused_blocks:
        # the library's NHWC rfft2: the [imag | real] packing of :1456-1465 is the kernel's output layout, no transposes, no cat
        x1f, x2f, x3f = spec_pack(x1), spec_pack(x2), spec_pack(x3)
    else:
        X = torch.fft.rfft2(x.contiguous(), norm="backward")         # one transform for the three groups
        Xi, Xr = X.imag, X.real
        x1f, x2f, x3f = [torch.cat([Xi[:, g * d:(g + 1) * d], Xr[:, g * d:(g + 1) * d]], dim=1) for g in range(3)]   # imag first
    side = torch.cat([x1f, x3f], 0)                                  # (2B, 2d, H, Wf): forward direction, then backward
    mid = torch.cat([x2f, x2f], 0)
    off = (side - mid) + c.chain(key + ".convfuse", torch.cat([side, mid], 1), 3)
    sim = c.chain(key + ".convcrt", x2f, 2)
    corr = corr_lookup(x1f, x2f) if c.fused_blocks else _corr_lookup(x1f, x2f)   # forward pair only, reused for both directions (:1487-1488)
    flow0 = torch.zeros(2 * B, 2, H, x1f.shape[-1], dtype=x.dtype, device=x.device)
    t = c.conv_padded(key + ".convcorr.0", [off, torch.cat([corr, corr], 0), flow0], act="relu")       # 211 -> 64 (:1379-1385)
    t = conv2d(t, c.p[key + ".convcorr.2.weight"], None, 1, c.precision, "relu")
    off = conv2d(t, c.p[key + ".convcorr.4.weight"], None, 1, c.precision)
    sim2 = torch.cat([sim, sim], 0)
    offs: List[List[Tensor]] = [[], []]
    for i in range(A):
        blk = f"{key}.MConvB.{i}"
        t = _prelu(c.conv(blk + ".conv1", off), c.p[blk + ".relu.weight"])
        u = c.conv(blk + ".conv2", t)
        o = (c.ca(blk + ".CA", u) + u) * sim2
        if c.fused_blocks:
            fld = irfft_pair(o, H, W)                                # channels [re0, re1 | im0, im1] -> two offset planes (:1497-1505)
        else:
            fld = torch.fft.irfft2(torch.complex(o[:, :2].contiguous(), o[:, 2:].contiguous()), s=(H, W), norm="backward")
        offs[0].append(fld[:B])
        offs[1].append(fld[B:])
    K = c.conv(key + ".F.1", c.conv(key + ".F.0", c.conv(key + ".conv_KP", x2)))
    if c.fused_blocks and d in (32, 64):
        a_f, a_b = iac_both(x1, x3, K, offs[0], offs[1], 0.1)
        return c.conv(key + ".conv3", torch.cat([a_f, a_b], 1)) + x2
    al = []
    for feat_in, ofs in ((x1, offs[0]), (x3, offs[1])):
        feat = feat_in
        for i in range(A):
            k1 = K[:, i * 6 * d: i * 6 * d + 3 * d]                # F1 half of iteration i (the F2 half is never read)
            feat = _lrelu(_sac(_warp(feat, ofs[i]), k1) + feat_in, 0.1)
        al.append(feat)
    return c.conv(key + ".conv3", torch.cat(al, 1)) + x2


_MASKS_DEV: Dict[tuple, Tensor] = {}


def _dev_masks(Q: int, H: int, W: int, dev) -> Tensor:
    """Band masks on the device, cached (no host-to-device copy inside a forward: the step can be captured in a hipGraph)."""
    key = (Q, H, W, str(dev))
    if key not in _MASKS_DEV:
        _MASKS_DEV[key] = band_masks_half(Q, H, W).to(dev)
    return _MASKS_DEV[key]


def _mffr(c: _Ctx, key: str, x: Tensor, Q: int) -> Tensor:
    B, C, H, W = x.shape
    M = _dev_masks(Q, H, W, x.device)
    if c.fused_blocks:
        freq = split_bands(x, M)[::-1]                               # one forward transform, the Q masked inverse transforms in one call
    else:
        X = torch.fft.rfft2(x.contiguous())
        freq = [torch.fft.irfft2(X * M[n], s=(H, W)) for n in range(Q)][::-1]
    s_f = torch.zeros_like(x)
    s_o = torch.zeros_like(x)
    for i in range(Q):
        blk = f"{key}.DivEnh_block.{i}"
        a, b = c.p[blk + ".a"].reshape(1, -1, 1, 1), c.p[blk + ".b"].reshape(1, -1, 1, 1)
        f = freq[i]
        if i == 0:
            o = c.ca(blk + ".ca", 0.2 * a * (f - f.mean(dim=(2, 3), keepdim=True)) * f + b * f)
        elif c.fused_blocks and C in (32, 64):
            s_f, s_o = divenh_band(f, s_f, s_o, c.p[blk + ".a"], c.p[blk + ".b"], c.p[blk + ".ca.conv_du.0.weight"],
                                   c.p[blk + ".ca.conv_du.2.weight"])
            continue
        else:
            t = f - s_f + 0.2 * s_o
            o = c.ca(blk + ".ca", 0.2 * a * t * f + b * f) + c.ca(blk + ".ca", 0.2 * a * s_o * f + b * f)
        s_f = s_f + f
        s_o = s_o + o
    return c.ca(key + ".ca", s_o) + x


def _context_block(c: _Ctx, key: str, r: Tensor) -> Tensor:
    """ContextBlock (:657-701): softmax-pooled context vector through a two-layer bottleneck, added to every pixel.  The two
    reductions over channels / pixels are batched matrix products on the NHWC buffer (no conv wrapper, no (B,C,HW) product)."""
    B, C, H, W = r.shape
    rm = r.permute(0, 2, 3, 1).reshape(B, H * W, C)                  # zero-copy view of the channels_last tensor
    logits = rm @ c.p[key + ".conv_mask.weight"].reshape(C, 1)       # (B, HW, 1)
    m = torch.softmax(logits, dim=1)
    ctx = (m.transpose(1, 2) @ rm).reshape(B, C)                     # (B,1,HW) x (B,HW,C)
    t = F.leaky_relu(ctx @ c.p[key + ".channel_add_conv.0.weight"].flatten(1).t(), 0.2)
    return r + (t @ c.p[key + ".channel_add_conv.2.weight"].flatten(1).t())[:, :, None, None]


def _block_rcb(c: _Ctx, key: str, xs: List[Tensor]) -> List[Tensor]:
    def lv(name, ts, act=None, slope=0.0):
        """one nn.Conv2d on every pyramid level (shared weights): grouped launches where the layer allows"""
        return conv2d_levels(ts, c.p[name + ".weight"], c.p.get(name + ".bias"), c.precision, act, slope)

    zs = lv(key + ".body.2", lv(key + ".body.0", xs, "lrelu", 0.1))
    rs = lv(key + ".RCB.body.2", lv(key + ".RCB.body.0", zs, "lrelu", 0.2))
    R = []
    for r, z in zip(rs, zs):
        if r.shape[1] == 64 and c.fused_blocks:
            g = key + ".RCB.gcnet"
            R.append(rcb_tail(r, z, c.p[g + ".conv_mask.weight"], c.p[g + ".channel_add_conv.0.weight"], c.p[g + ".channel_add_conv.2.weight"], 0.2))
        else:
            R.append(_lrelu(_context_block(c, key + ".RCB.gcnet", r), 0.2) + z)
    d0, d1 = lv(key + ".down.0", [R[0], R[1]])
    u1, u2 = lv(key + ".up.0", [R[1], R[2]])
    if c.fused_blocks and all(t.shape[2] % 2 == 0 and t.shape[3] % 2 == 0 for t in xs[:2]) and xs[0].shape[1] % 4 == 0:
        # (for even sizes the x0.5 bilinear down-sampling is the 2x2 mean; the doubled R0 / R2 of :771-776 are r_scale = 2)
        return [xscale(xs[0], R[0], 2.0, None, u1), xscale(xs[1], R[1], 1.0, d0, u2), xscale(xs[2], R[2], 2.0, d1, None)]
    dn = [F.interpolate(t, scale_factor=0.5, mode="bilinear", align_corners=False) for t in (d0, d1)]
    up = [F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=False) for t in (u1, u2)]
    return [xs[0] + R[0] + R[0] + up[0], xs[1] + R[1] + dn[0] + up[1], xs[2] + R[2] + dn[1] + R[2]]


def _scnet(c: _Ctx, key: str, xs: List[Tensor], G: int) -> List[Tensor]:
    cur = xs
    for g in range(G):
        t = cur
        for k in range(3):
            t = _block_rcb(c, f"{key}.body.{g}.body.{k}", t)
        cv = conv2d_levels(t, c.p[f"{key}.body.{g}.conv.weight"], c.p.get(f"{key}.body.{g}.conv.bias"), c.precision)
        cur = [a + r for a, r in zip(cur, cv)]
    return [x + r for x, r in zip(xs, cur)]


def forward_train(p: Dict[str, Tensor], x: Tensor, *, precision: str = "f32", fused_blocks: bool = True) -> Tensor:
    """x: (B,7,C,H,W) device tensor in [0,1] -> (B,C,4H,4W), differentiable w.r.t. every live parameter in `p`."""
    if not x.is_cuda:
        raise RuntimeError("fcvsr_amd.train needs device tensors (the HIP path has no CPU fallback)")
    n = p["conv_last0.weight"].shape[1]
    A = p["MGAA.F.1.weight"].shape[0] // (6 * n)
    Q = sum(1 for k in p if k.startswith("MFFRblock.DivEnh_block.") and k.endswith(".a"))
    G = sum(1 for k in p if k.startswith("recorb1.body.") and k.endswith(".conv.weight") and k.count(".") == 4)
    B, T, C, H, W = x.shape
    if H % 4 or W % 4:
        raise ValueError("H and W must be multiples of 4 (3-level pyramid, reference BlockRCB :766-777)")
    # The packed 16-bit weight operands are cached per (storage pointer, version) for the duration of ONE forward + backward pass (a
    # layer applied several times packs once).  Across passes the cache must not survive: the optimizer rewrites every weight, and a
    # freed model's storage can be handed to another parameter of the same shape and version.  Clearing here also guarantees that a
    # hipGraph capture records the packing kernels of its pass.
    first, last = next(iter(p.values())), p["conv_last0.weight"]
    clear_packed_weights(owner=(first.data_ptr(), last.data_ptr(), len(p), precision, str(first.device)))
    c = _Ctx(p, precision, fused_blocks)
    x7 = x.reshape(B, T * C, H, W).float()
    if precision == "f32" or (T * C) % 64 == 0:
        feat = conv2d(x7, p["feat_extract.0.weight"], p["feat_extract.0.bias"], 1, "f32")
    else:
        # 16-bit modes: the frame stack completed with zero channels to 64 and the weight with zero input columns, products in f16
        # (8-bit pixel values stay exact, as in the inference engine): forward and weight gradient on the matrix cores - the exact-f32
        # VALU weight gradient of this 7 -> 448 layer alone cost 1.1 ms per step
        cpad = (-(T * C)) % 64
        xz = torch.cat([x7, torch.zeros(B, cpad, H, W, dtype=x7.dtype, device=x7.device)], 1).contiguous(memory_format=torch.channels_last)
        feat = conv2d(xz, F.pad(p["feat_extract.0.weight"], (0, 0, 0, 0, 0, cpad)), p["feat_extract.0.bias"], 1, "f16")
    f1, f2, f3 = feat[:, :3 * n], feat[:, 3 * n:4 * n], feat[:, 4 * n:]
    a13 = _mgaa(c, "MGAA", torch.cat([f1, f3], 0), A)               # the two outer calls share the weights: one call on 2B samples
    a1, a3 = a13[:B], a13[B:]
    a2 = _mgaa(c, "MGAA", torch.cat([a1, f2, a3], 1), A)
    d0 = _mffr(c, "MFFRblock", a2, Q)
    d1 = c.conv_s2("rconcat1", d0)
    d2 = c.conv_s2("rconcat2", d1)
    o0, o1, o2 = _scnet(c, "recorb1", [d0, d1, d2], G)
    a = p["lrelu.weight"]
    l3 = _prelu(c.conv("upconv1_L3", o2), a)
    l3_1 = _ps2(l3)
    l3_2 = _ps2(l3_1)
    l2 = _prelu(c.conv("upconv1_L2", o1), a)
    l2 = _ps2(l2 + c.conv_padded("upconv1_L2_2", [l2, l3_1]))
    fz = c.conv("recorb0", c.conv_padded("upconv_fuse", [o0, l2, l3_2]))
    u = _ps2(_prelu(c.conv("upconv1", fz), a))                      # PReLU (one shared slope) commutes with the shuffle: same values,
    u = _ps2(_prelu(c.conv("upconv2", u), a))                       # and the activation then runs on the dense conv output
    out = conv2d(u, p["conv_last0.weight"], p["conv_last0.bias"], 1, precision)   # (16-bit modes: a 4-channel MFMA layer, rows 1..3 zero)
    base = F.interpolate(x[:, T // 2].float(), scale_factor=4, mode="bilinear", align_corners=False)
    return out + base
