"""CPU oracle for the FCVSR per-frame forward hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``fcvsr_amd``) never imports anything from ``oracle/`` and has no CPU fallback.

It is a functional (parameter-dict in, tensor out) fp32 restatement, in plain
PyTorch CPU ops, of the reference's ``GShiftNet_S.forward`` / ``GShiftNet.forward``
(reference: CVSR_train/arch/CVSR_freq.py:2611-2646 and :2688-2756), written from the
functional specification in SURVEY.md Appendix A using *clean* math (gather-form
CorrBlock, rfft-symmetrised band masks, explicit bilinear warp, two-pass separable
adaptive conv) rather than the reference's op sequence.

Parity pinning: ``tests/golden/*.npz`` hold inputs, per-block taps and outputs produced
by importing the reference model itself in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks this file
against every one of them (max-abs <= 2e-5 on all taps).  The reference has no golden
vectors of its own for this path, so those fixtures *are* the pin.

All functions take ``p``: a ``dict[str, Tensor]`` keyed exactly like the reference
``state_dict``, and a key prefix.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def _conv(p: Params, key: str, x: Tensor, stride: int = 1) -> Tensor:
    """``nn.Conv2d`` with "same" zero padding (k//2); bias if ``key.bias`` exists."""
    w = p[key + ".weight"]
    b = p.get(key + ".bias")
    return F.conv2d(x, w, b, stride=stride, padding=w.shape[-1] // 2)


def _lrelu(x: Tensor, slope: float) -> Tensor:
    return torch.where(x >= 0, x, x * slope)


def _prelu(x: Tensor, a: Tensor) -> Tensor:
    """Scalar-slope PReLU (reference: nn.PReLU() default num_parameters=1)."""
    return torch.where(x >= 0, x, x * a.reshape(1, -1, 1, 1))


def _ca(p: Params, key: str, z: Tensor) -> Tensor:
    """CALayer (CVSR_freq.py:1812-1828): z * sigmoid(W2 relu(W1 mean_HW(z))), no bias."""
    y = z.mean(dim=(2, 3), keepdim=True)
    y = F.relu(F.conv2d(y, p[key + ".conv_du.0.weight"]))
    y = torch.sigmoid(F.conv2d(y, p[key + ".conv_du.2.weight"]))
    return z * y


def pixel_shuffle2(x: Tensor) -> Tensor:
    """out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w]  (nn.PixelShuffle(2))."""
    B, C, H, W = x.shape
    x = x.reshape(B, C // 4, 2, 2, H, W).permute(0, 1, 4, 2, 5, 3)
    return x.reshape(B, C // 4, 2 * H, 2 * W)


# ----------------------------------------------------------------------------------------------
# MGAAbk (CVSR_freq.py:1365-1547)
# ----------------------------------------------------------------------------------------------
def spec_pack(x: Tensor) -> Tensor:
    """rfft2 (unnormalised forward) packed as [imag, real] on the channel axis (:1452-1465)."""
    X = torch.fft.rfft2(x, norm="backward")
    return torch.cat([X.imag, X.real], dim=1)


def corr_lookup(x1f: Tensor, x2f: Tensor, radius: int = 4) -> Tensor:
    """Gather form of CorrBlock (:1279-1337) evaluated on the integer grid.

    P = x1f*x2f/sqrt(C) is a contiguous (B, C, H*Wf) buffer; pixel p=y*Wf+x owns the C consecutive
    floats at flat offset p*C, viewed as an image I_p of C/2 rows x 2 cols.  Output channel
    c=i*(2r+1)+j is I_p[y+j-r, x+i-r] when that index is in range, else 0.
    """
    B, C, H, Wf = x1f.shape
    n = 2 * radius + 1
    P = (x1f * x2f).reshape(B, C * H * Wf) / torch.sqrt(torch.tensor(float(C)))
    img = P.reshape(B, H * Wf, C // 2, 2)
    ys = torch.arange(H).view(H, 1).expand(H, Wf)
    xs = torch.arange(Wf).view(1, Wf).expand(H, Wf)
    pix = ys * Wf + xs
    out = torch.zeros(B, n * n, H, Wf, dtype=x1f.dtype)
    for i in range(n):
        col = xs + (i - radius)
        for j in range(n):
            row = ys + (j - radius)
            ok = (col >= 0) & (col <= 1) & (row >= 0) & (row <= C // 2 - 1)
            v = img[:, pix, row.clamp(0, C // 2 - 1), col.clamp(0, 1)]
            out[:, i * n + j] = v * ok.to(v.dtype)
    return out


def conv_blk(p: Params, key: str, x: Tensor) -> Tensor:
    """ConvBlk (:344-357): conv k -> PReLU -> conv k -> CA_1(u) + u  (k = 2*index+1, no bias)."""
    t = _prelu(_conv(p, key + ".conv1", x), p[key + ".relu.weight"])
    u = _conv(p, key + ".conv2", t)
    return _ca(p, key + ".CA", u) + u


def warp_bilinear(x: Tensor, off: Tensor) -> Tensor:
    """flow_warp (:1188-1227): sample x at (col+off[:,0], row+off[:,1]), bilinear, zeros outside."""
    B, C, H, W = x.shape
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    sx = xs + off[:, 0]
    sy = ys + off[:, 1]
    x0 = torch.floor(sx)
    y0 = torch.floor(sy)
    wx1 = sx - x0
    wy1 = sy - y0
    out = torch.zeros_like(x)
    flat = x.reshape(B, C, H * W)
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            xi = (x0 + dx).long()
            yi = (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).view(B, 1, H * W).expand(B, C, H * W)
            v = torch.gather(flat, 2, idx).view(B, C, H, W)
            out = out + v * (wy * wx * ok.to(x.dtype)).unsqueeze(1)
    return out


def sac_kernel1_twice(s: Tensor, k1: Tensor) -> Tensor:
    """SAC (:1253-1276): vertical 3-tap then horizontal 3-tap, replicate padding, *both* with kernel1.

    k1: (B, C*3, H, W) with channel index c*3+t.
    """
    B, C, H, W = s.shape
    k = k1.reshape(B, C, 3, H, W)
    sp = F.pad(s, (0, 0, 1, 1), mode="replicate")
    v = sum(sp[:, :, t:t + H, :] * k[:, :, t] for t in range(3))
    vp = F.pad(v, (1, 1, 0, 0), mode="replicate")
    h = sum(vp[:, :, :, t:t + W] * k[:, :, t] for t in range(3))
    return h


def iac(feat_in: Tensor, K: Tensor, offsets: List[Tensor], A: int) -> Tensor:
    """IAC (:1230-1250): A x { warp -> SAC(kernel1 twice) -> + feat_in -> LeakyReLU(0.1) }."""
    C = feat_in.shape[1]
    feat = feat_in
    for i in range(A):
        k1 = K[:, i * 6 * C: i * 6 * C + 3 * C]        # F1 half of iteration i; the F2 half is never read
        s = warp_bilinear(feat, offsets[i])
        feat = _lrelu(sac_kernel1_twice(s, k1) + feat_in, 0.1)
    return feat


def mgaa(p: Params, key: str, x: Tensor, A: int, taps: Optional[dict] = None, tag: str = "") -> Tensor:
    B, C3, H, W = x.shape
    d = C3 // 3
    x1, x2, x3 = x[:, :d], x[:, d:2 * d], x[:, 2 * d:]
    x1f, x2f, x3f = spec_pack(x1), spec_pack(x2), spec_pack(x3)

    def chain(prefix: str, t: Tensor, n: int) -> Tensor:
        for li in range(n):
            t = F.conv2d(t, p[f"{key}.{prefix}.{2 * li}.weight"])
            if li < n - 1:
                t = F.relu(t)
        return t

    off_f = (x1f - x2f) + chain("convfuse", torch.cat([x1f, x2f], 1), 3)
    off_b = (x3f - x2f) + chain("convfuse", torch.cat([x3f, x2f], 1), 3)
    sim = chain("convcrt", x2f, 2)
    corr = corr_lookup(x1f, x2f)                         # forward pair only; reused for both directions
    zero_flow = torch.zeros(B, 2, H, x1f.shape[-1], dtype=x.dtype)
    off_f = chain("convcorr", torch.cat([off_f, corr, zero_flow], 1), 3)
    off_b = chain("convcorr", torch.cat([off_b, corr, zero_flow], 1), 3)

    offs_f, offs_b = [], []
    for i in range(A):
        for src, dst in ((off_f, offs_f), (off_b, offs_b)):
            o = conv_blk(p, f"{key}.MConvB.{i}", src) * sim
            re, im = o[:, :2], o[:, 2:]                  # (real, imag) order on the way out
            dst.append(torch.fft.irfft2(torch.complex(re, im), s=(H, W), norm="backward"))

    K = _conv(p, key + ".F.1", _conv(p, key + ".F.0", _conv(p, key + ".conv_KP", x2)))
    al_f = iac(x1, K, offs_f, A)
    al_b = iac(x3, K, offs_b, A)
    out = _conv(p, key + ".conv3", torch.cat([al_f, al_b], 1)) + x2
    if taps is not None:
        taps[f"mgaa{tag}.off_f"] = off_f
        taps[f"mgaa{tag}.off_b"] = off_b
        taps[f"mgaa{tag}.sim"] = sim
        taps[f"mgaa{tag}.offsets_f"] = torch.stack(offs_f, 1)
        taps[f"mgaa{tag}.offsets_b"] = torch.stack(offs_b, 1)
        taps[f"mgaa{tag}.al_f"] = al_f
        taps[f"mgaa{tag}.al_b"] = al_b
        taps[f"mgaa{tag}.out"] = out
    return out


# ----------------------------------------------------------------------------------------------
# MultiFreq_Refinment (CVSR_freq.py:2183-2254) with Split_freq (:2008-2101), DivEnh (:2104-2133)
# ----------------------------------------------------------------------------------------------
_MASK_CACHE: Dict[Tuple[int, int, int], Tensor] = {}


def gaussian_band_masks_1024(Q: int) -> Tensor:
    """(Q,1024,1024) fp32 telescoping Gaussian masks, centred (fftshift) layout (:2031-2049)."""
    Hm = Wm = 1024
    length = math.sqrt((Hm / 2) ** 2 + (Wm / 2) ** 2)
    step = length / Q
    h2 = (np.arange(-(Hm // 2), Hm - Hm // 2, 1) ** 2).astype(np.float64)
    w2 = (np.arange(-(Wm // 2), Wm - Wm // 2, 1) ** 2).astype(np.float64)
    r2 = np.power(np.sqrt(h2[:, None] + w2[None, :]), 2)
    chunks: List[Tensor] = []
    for n in range(Q):
        g = torch.from_numpy(np.exp(-r2 / (2 * ((step * (n + 1)) ** 2)))).float()
        for prev in chunks:
            g = g - prev
        chunks.append(g)
    return torch.stack(chunks, 0)


def band_masks_half(Q: int, H: int, W: int) -> Tensor:
    """Symmetrised, un-shifted, half-spectrum masks M_sym (Q,H,Wf) such that
    Re(ifft2(ifftshift(fftshift(fft2 x) * mask))) == irfft2(rfft2(x) * M_sym)   (SURVEY A.4).

    The 1024^2 masks are bicubic point-resized to (H,W) (torchvision-0.14 tensor Resize ==
    F.interpolate(bicubic, align_corners=False, antialias=False); SURVEY 8c assumption).
    """
    ck = (Q, H, W)
    if ck not in _MASK_CACHE:
        m = gaussian_band_masks_1024(Q)
        m = F.interpolate(m[None], size=[H, W], mode="bicubic", align_corners=False, antialias=False)[0]
        M = torch.fft.ifftshift(m, dim=(1, 2))
        Mneg = torch.roll(torch.flip(M, dims=(1, 2)), shifts=(1, 1), dims=(1, 2))  # M(-k)
        _MASK_CACHE[ck] = (0.5 * (M + Mneg))[:, :, : W // 2 + 1].contiguous()
    return _MASK_CACHE[ck]


def split_bands(x: Tensor, Q: int) -> List[Tensor]:
    B, C, H, W = x.shape
    M = band_masks_half(Q, H, W)
    X = torch.fft.rfft2(x)
    return [torch.fft.irfft2(X * M[n], s=(H, W)) for n in range(Q)]


def mffr(p: Params, key: str, x: Tensor, Q: int, taps: Optional[dict] = None) -> Tensor:
    freq = split_bands(x, Q)[::-1]                        # "l2h" => reversed list (:2204-2205)
    if taps is not None:
        taps["mffr.bands"] = torch.stack(freq, 1)
    outs: List[Tensor] = []
    s_f = torch.zeros_like(x)
    s_o = torch.zeros_like(x)
    for i in range(Q):
        a = p[f"{key}.DivEnh_block.{i}.a"].reshape(1, -1, 1, 1)
        b = p[f"{key}.DivEnh_block.{i}.b"].reshape(1, -1, 1, 1)
        ca = f"{key}.DivEnh_block.{i}.ca"
        f = freq[i]
        if i == 0:
            t = f - f.mean(dim=(2, 3), keepdim=True)
            o = _ca(p, ca, 0.2 * a * t * f + b * f)
        else:
            t = f - s_f + 0.2 * s_o
            o = _ca(p, ca, 0.2 * a * t * f + b * f) + _ca(p, ca, 0.2 * a * s_o * f + b * f)
        outs.append(o)
        s_f = s_f + f
        s_o = s_o + o
    out = _ca(p, key + ".ca", s_o) + x
    if taps is not None:
        taps["mffr.out"] = out
    return out


# ----------------------------------------------------------------------------------------------
# SCNetbk (CVSR_freq.py:807-822) and children
# ----------------------------------------------------------------------------------------------
def context_block(p: Params, key: str, r: Tensor) -> Tensor:
    """ContextBlock (:657-701): softmax-pooled global context + 2-layer 1x1 MLP, added to r."""
    B, C, H, W = r.shape
    logits = F.conv2d(r, p[key + ".conv_mask.weight"]).reshape(B, 1, H * W)
    m = torch.softmax(logits, dim=2)
    ctx = (r.reshape(B, C, H * W) * m).sum(dim=2).reshape(B, C, 1, 1)
    t = _lrelu(F.conv2d(ctx, p[key + ".channel_add_conv.0.weight"]), 0.2)
    return r + F.conv2d(t, p[key + ".channel_add_conv.2.weight"])


def rcb(p: Params, key: str, z: Tensor) -> Tensor:
    r = _conv(p, key + ".body.2", _lrelu(_conv(p, key + ".body.0", z), 0.2))
    return _lrelu(context_block(p, key + ".gcnet", r), 0.2) + z


def block_rcb(p: Params, key: str, xs: List[Tensor]) -> List[Tensor]:
    def body(z: Tensor) -> Tensor:
        z = _conv(p, key + ".body.2", _lrelu(_conv(p, key + ".body.0", z), 0.1))
        return rcb(p, key + ".RCB", z)

    def dn(z: Tensor) -> Tensor:
        return F.interpolate(_conv(p, key + ".down.0", z), scale_factor=0.5, mode="bilinear", align_corners=False)

    def up(z: Tensor) -> Tensor:
        return F.interpolate(_conv(p, key + ".up.0", z), scale_factor=2.0, mode="bilinear", align_corners=False)

    R = [body(z) for z in xs]
    return [
        xs[0] + R[0] + R[0] + up(R[1]),
        xs[1] + R[1] + dn(R[0]) + up(R[2]),
        xs[2] + R[2] + dn(R[1]) + R[2],
    ]


def scnet(p: Params, key: str, xs: List[Tensor], G: int) -> List[Tensor]:
    cur = xs
    for g in range(G):
        t = cur
        for k in range(3):
            t = block_rcb(p, f"{key}.body.{g}.body.{k}", t)
        cur = [c + _conv(p, f"{key}.body.{g}.conv", r) for c, r in zip(cur, t)]
    return [x + r for x, r in zip(xs, cur)]


# ----------------------------------------------------------------------------------------------
# top level (CVSR_freq.py:2611-2646 / :2688-2756; RGB twins fcvsr.py:74-142, fcvsr_s.py:76-)
# ----------------------------------------------------------------------------------------------
def infer_config(p: Params) -> dict:
    n = p["conv_last0.weight"].shape[1]
    A = p["MGAA.F.1.weight"].shape[0] // (6 * n)
    Q = 0
    while f"MFFRblock.DivEnh_block.{Q}.a" in p:
        Q += 1
    G = 0
    while f"recorb1.body.{G}.conv.weight" in p:
        G += 1
    return dict(n=n, A=A, Q=Q, G=G, in_ch=p["feat_extract.0.weight"].shape[1],
                out_ch=p["conv_last0.weight"].shape[0])


def forward(p: Params, x: Tensor, taps: Optional[dict] = None) -> Tensor:
    """x: (B,7,C,H,W) float32 in [0,1] -> (B,C,4H,4W).  C=1 (Y model) or 3 (RGB twin)."""
    cfg = infer_config(p)
    n, A, Q, G = cfg["n"], cfg["A"], cfg["Q"], cfg["G"]
    B, T, C, H, W = x.shape
    if H % 4 or W % 4:
        raise ValueError("H and W must be multiples of 4 (SURVEY A.7)")
    feat = _conv(p, "feat_extract.0", x.reshape(B, T * C, H, W))
    f1, f2, f3 = feat[:, :3 * n], feat[:, 3 * n:4 * n], feat[:, 4 * n:]
    a1 = mgaa(p, "MGAA", f1, A, taps, "1")
    a3 = mgaa(p, "MGAA", f3, A, taps, "3")
    a2 = mgaa(p, "MGAA", torch.cat([a1, f2, a3], 1), A, taps, "2")
    d0 = mffr(p, "MFFRblock", a2, Q, taps)
    d1 = _conv(p, "rconcat1", d0, stride=2)
    d2 = _conv(p, "rconcat2", d1, stride=2)
    o0, o1, o2 = scnet(p, "recorb1", [d0, d1, d2], G)
    a = p["lrelu.weight"]
    l3 = _prelu(_conv(p, "upconv1_L3", o2), a)
    l3_1 = pixel_shuffle2(l3)
    l3_2 = pixel_shuffle2(l3_1)
    l2 = _prelu(_conv(p, "upconv1_L2", o1), a)
    l2 = pixel_shuffle2(l2 + _conv(p, "upconv1_L2_2", torch.cat([l2, l3_1], 1)))
    fz = _conv(p, "recorb0", _conv(p, "upconv_fuse", torch.cat([o0, l2, l3_2], 1)))
    u = _prelu(pixel_shuffle2(_conv(p, "upconv1", fz)), a)
    u = _prelu(pixel_shuffle2(_conv(p, "upconv2", u)), a)
    out = _conv(p, "conv_last0", u)
    base = F.interpolate(x[:, T // 2], scale_factor=4, mode="bilinear", align_corners=False)
    if taps is not None:
        taps["feat"] = feat
        taps["sc.o0"], taps["sc.o1"], taps["sc.o2"] = o0, o1, o2
        taps["fz"] = fz
        taps["out"] = out + base
    return out + base


def forward_etc(p: Params, x: Tensor) -> Tuple[Tensor, Tensor]:
    """GShiftNet_ETC (CVSR_freq.py:2760-2843): x (B,13,C,H,W) -> (out_seq, x_up), both (B,7,C,4H,4W): the network of
    ``forward`` on the 7 windows x[:, i:i+7] and the bilinear x4 bases of their centre frames."""
    outs, ups = [], []
    for i in range(7):
        sub = x[:, i:i + 7]
        outs.append(forward(p, sub))
        ups.append(F.interpolate(sub[:, 3], scale_factor=4, mode="bilinear", align_corners=False))
    return torch.stack(outs, 1), torch.stack(ups, 1)
