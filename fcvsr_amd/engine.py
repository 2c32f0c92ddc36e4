"""Host-side orchestration of the FCVSR forward on one MI355X: the sequence of libfcvsr_hip.so calls that realises
``GShiftNet_S.forward`` / ``GShiftNet.forward`` (reference CVSR_train/arch/CVSR_freq.py:2611-2646, :2688-2756).

Layout in HBM: every internal activation is NHWC f32 (``(B,H,W,C)`` contiguous, channels innermost); channel slices and
concatenations are expressed as strided views (no copies).  The NCHW boundary tensors are read / written in place through
strided views as well (first conv reads the caller's ``(B,7,C,H,W)`` frames directly, the last conv writes ``(B,C,4H,4W)``).
torch is used for device memory (caching allocator), streams and the one-time weight re-packing; no torch arithmetic
runs in the per-frame path.

Dead compute of the reference that cannot affect outputs is skipped: ``corrb``, the ``F2`` half of
``MGAA.F[1]``, the zero flow channels, ``DivEnh.Conv``, and every visualisation / host sync inside forward.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import hip
from .hip import ACT_LEAKY, ACT_NONE, ACT_PRELU, ACT_RELU, check, lib, ptr, stream_ptr, view


# --------------------------------------------------------------------------------------------------------------
# Band masks (host-side constant tables, cached per (Q,H,W); reference Split_freq.generate_freq_mask :2016-2049 +
# bicubic Resize :2078, folded into the symmetrised half-spectrum form of SURVEY A.4)
# --------------------------------------------------------------------------------------------------------------
_MASKS: Dict[Tuple[int, int, int], torch.Tensor] = {}


def band_masks_half(Q: int, H: int, W: int) -> torch.Tensor:
    """(Q,H,Wf) f32 CPU tensor M_sym with  Re(ifft2(ifftshift(fftshift(fft2 x) * mask_q))) == irfft2(rfft2(x) * M_sym[q])."""
    key = (Q, H, W)
    if key not in _MASKS:
        n = 1024
        step = math.sqrt((n / 2) ** 2 + (n / 2) ** 2) / Q
        d2 = (np.arange(-(n // 2), n - n // 2, 1) ** 2).astype(np.float64)
        r2 = np.power(np.sqrt(d2[:, None] + d2[None, :]), 2)
        bands: List[torch.Tensor] = []
        for q in range(Q):
            g = torch.from_numpy(np.exp(-r2 / (2 * ((step * (q + 1)) ** 2)))).float()
            for prev in bands:
                g = g - prev
            bands.append(g)
        m = torch.stack(bands, 0)
        m = torch.nn.functional.interpolate(m[None], size=[H, W], mode="bicubic", align_corners=False,
                                            antialias=False)[0]
        M = torch.fft.ifftshift(m, dim=(1, 2))
        Mneg = torch.roll(torch.flip(M, dims=(1, 2)), shifts=(1, 1), dims=(1, 2))
        _MASKS[key] = (0.5 * (M + Mneg))[:, :, : W // 2 + 1].contiguous()
    return _MASKS[key]


class Engine:
    def __init__(self, model):
        self._model = weakref.ref(model)
        self._packed: Dict[str, torch.Tensor] = {}
        self._versions: Optional[Tuple] = None
        self._dev_masks: Dict[Tuple, torch.Tensor] = {}
        self._par: Dict[str, torch.Tensor] = {}
        self._streams: List[torch.cuda.Stream] = []
        self._graphs: "OrderedDict[Tuple, Tuple]" = OrderedDict()
        self._warm: set = set()                # configurations whose cached tensors exist (built on the caller's stream)
        self._pack_epoch = 0
        self._warned = False
        self.taps: Optional[dict] = None       # set to a dict to record NCHW copies of intermediate results (tests)

    @property
    def precision(self) -> str:
        p = getattr(self._model(), "precision", "f32")
        if p not in ("f32", "bf16", "f16"):
            raise ValueError(f"precision must be 'f32', 'bf16' or 'f16', got {p!r}")
        return p

    # ---------------------------------------------------------------------------------------------- weights
    def invalidate(self):
        """Drop every cached re-packed weight and captured hipGraph.  The caches are keyed on (Parameter._version, data_ptr),
        which in-place edits through ``param.data`` (EMA hooks, ``w.data *= s``) do not change: call this (or
        ``model.invalidate()``) after such an edit."""
        self._versions = None
        self._packed = {}
        self._graphs.clear()
        self._warm.clear()
        self._pack_epoch += 1

    def _refresh(self, dev):
        m = self._model()
        sd = {k: v for k, v in m.named_parameters()}
        ver = tuple((k, v._version, v.data_ptr()) for k, v in sd.items()) + (str(dev),)
        if ver == self._versions:
            return
        for k, v in sd.items():
            if v.device != dev:
                raise RuntimeError(f"parameter {k} is on {v.device}, input on {dev}: call model.to(device) first")
            if v.dtype != torch.float32:
                raise RuntimeError(f"parameter {k} is {v.dtype}: keep the parameters in float32 and select the arithmetic "
                                   "with model.precision = 'bf16' | 'f16' instead of casting the module")
        self._packed = {}
        self._graphs.clear()
        self._warm.clear()
        self._pack_epoch += 1
        self._par = sd
        self._versions = ver

    def _logical_weight(self, name):
        """(Cout,Cin,kh,kw) f32 weight (+bias) as the kernels see it: dead rows/columns removed, concat padding added."""
        m = self._model()
        n, A = m.n_feats, m.ACNum
        w = self._par[name + ".weight"].detach()
        b = self._par.get(name + ".bias")
        if name == "MGAA.F.1":          # only the F1 half of every iteration's 2*3n kernel channels is ever read
            rows = torch.cat([torch.arange(i * 6 * n, i * 6 * n + 3 * n) for i in range(A)]).to(w.device)
            return w[rows], b.detach()[rows].contiguous()
        if name == "MGAA.convcorr.0":   # drop the 2 zero-flow inputs, pad corr 81 -> 84 channels (16-byte pixels)
            wp = torch.zeros(w.shape[0], 2 * n + 84, 1, 1, device=w.device, dtype=torch.float32)
            wp[:, : 2 * n + 81] = w[:, : 2 * n + 81]
            return wp, None
        return w, (b.detach() if b is not None else None)

    def _weights(self, name, kind, ps=False, rows=None, cols=None):
        """kind: 'direct' (f32 [taps][cin][cout16]) | torch.bfloat16 | torch.float16 (MFMA layout; with ps=True the rows
        are in the sub-pixel-major order the MFMA kernel's pixel-shuffle epilogue expects)."""
        key = (name, kind, ps, rows, cols)
        if key not in self._packed:
            w, b = self._logical_weight(name)
            if cols is not None:                      # a slice of the input channels (a layer applied to part of a concat)
                w = w[:, cols[0]:cols[1]]
            if rows is not None:                      # a slice of the output channels (feat_extract is run per frame group)
                w = w[rows[0]:rows[1]]
                b = b[rows[0]:rows[1]].contiguous() if b is not None else None
            if kind == "direct":
                pk = hip.pack_conv_weight(w)
            elif kind == "f32mfma":
                pk = hip.pack_conv_weight_f32mfma(w, ps=ps)
                if ps and b is not None:
                    b = b[hip.ps_order(w.shape[0]).to(b.device)].contiguous()
            else:
                pk = hip.pack_conv_weight_mfma(w, kind, ps=ps)
                if ps and b is not None:
                    b = b[hip.ps_order(w.shape[0]).to(b.device)].contiguous()
            self._packed[key] = (pk, b, w.shape[0], w.shape[-1])
        return self._packed[key]

    def _feat_weights(self):
        """feat_extract.0 as a [cout][64] f16 matrix, column k = tap*Cin + c (the im2col order of fcvsr_feat_extract)."""
        key = ("feat_extract.0", "im2col")
        if key not in self._packed:
            w = self._par["feat_extract.0.weight"].detach()      # (cout, cin, 3, 3)
            b = self._par.get("feat_extract.0.bias")
            mat = torch.zeros(w.shape[0], 64, device=w.device, dtype=torch.float32)
            mat[:, :9 * w.shape[1]] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
            self._packed[key] = (mat.to(torch.float16).contiguous(), b.detach().contiguous() if b is not None else None)
        return self._packed[key]

    def _tap_weights(self, name, dt):
        """A 3x3 layer with one output channel as a [16][cin] table in dtype dt: row = tap ky*3+kx, rows 9..15 zero (the
        B operand of the 'taps are output columns' GEMM of the fused tail kernel)."""
        key = (name, "taps", dt)
        if key not in self._packed:
            w = self._par[name + ".weight"].detach()             # (1, cin, 3, 3)
            tab = torch.zeros(16, w.shape[1], device=w.device, dtype=torch.float32)
            tab[:9] = w[0].permute(1, 2, 0).reshape(9, w.shape[1])
            self._packed[key] = tab.to(dt).contiguous()
        return self._packed[key]

    def _last_weights(self, name, dt, cimg):
        """conv_last0 (cimg, cin, 3, 3) as the [16 or 32][cin] table of fcvsr_conv_last: row = tap * cimg + c, zero rows past 9 cimg."""
        key = (name, "last", dt)
        if key not in self._packed:
            w = self._par[name + ".weight"].detach()             # (cimg, cin, 3, 3)
            tab = torch.zeros(16 if cimg == 1 else 32, w.shape[1], device=w.device, dtype=torch.float32)
            tab[:9 * cimg] = w.permute(2, 3, 0, 1).reshape(9 * cimg, w.shape[1])      # (ky, kx, c, cin)
            self._packed[key] = tab.to(dt).contiguous()
        return self._packed[key]

    def _tdt(self):
        """Storage dtype of the SCNetbk trunk (x, t2, R, cross-scale terms, block outputs): f32, or - with
        model.trunk16 in the 16-bit modes - the MFMA operand dtype (halves the bytes and the staging instructions of
        every SCNet kernel; f16 recommended: each block rounds the trunk once)."""
        if self.precision == "f32" or not getattr(self._model(), "trunk16", False):
            return torch.float32
        return self._adt()

    @staticmethod
    def _code(dt):
        return {torch.float32: hip.F32, torch.bfloat16: hip.BF16, torch.float16: hip.F16}[dt]

    def _adt(self, freq=False):
        """Storage dtype for tensors whose only consumers are MFMA convolutions: the MFMA operand dtype itself
        (the consumer would round to it anyway, so results are bit-identical and the HBM bytes halve)."""
        if self.precision == "f32":
            return torch.float32
        return torch.bfloat16 if (self.precision == "bf16" or freq) else torch.float16

    def _mask(self, Q, H, W, dev):
        key = (Q, H, W, str(dev))
        if key not in self._dev_masks:
            self._dev_masks[key] = band_masks_half(Q, H, W).to(dev)
        return self._dev_masks[key]

    # ---------------------------------------------------------------------------------------------- helpers
    def _conv(self, name, srcs, dst, *, stride=1, act=ACT_NONE, slope=0.0, slope_t=None, res=(), res_scale=(),
              ps=False, freq=False, direct=False, force_f16=False, rows=None, cols=None):
        self._convg(name, [dict(srcs=srcs, dst=dst, res=res, ps=ps)], stride=stride, act=act, slope=slope,
                    slope_t=slope_t, res_scale=res_scale, ps=ps, freq=freq, direct=direct, force_f16=force_f16, rows=rows,
                    cols=cols)
        return dst

    def _convg(self, name, groups, *, stride=1, act=ACT_NONE, slope=0.0, slope_t=None, res_scale=(), ps=False,
               freq=False, direct=False, force_f16=False, gc_wmask=None, rows=None, cols=None):
        """One conv layer applied to 1..3 tensors that share its weights (pyramid levels): a single grouped MFMA launch in
        the 16-bit modes, per-tensor exact-f32 direct launches otherwise."""
        ksz = self._par[name + ".weight"].shape[-1]
        any16 = any(t.dtype != torch.float32 for g in groups for t in list(g["srcs"]) + [g["dst"]])
        elig = self.precision != "f32" and not direct and hip.mfma_eligible(ksz, stride, groups)
        if any16 and not elig:
            raise RuntimeError(f"{name}: 16-bit activations require the MFMA path (ineligible shape/alignment)")
        if elig:
            # spectra are unnormalised (|DC| ~ H*W*mean can exceed the f16 range): frequency-domain layers use bf16
            dt = torch.bfloat16 if (self.precision == "bf16" or freq) else torch.float16
            if force_f16:      # image-domain input in [0,1]: f16's 11-bit significand keeps 8-bit pixels exact
                dt = torch.float16
            w, b, cout, _ = self._weights(name, dt, ps, rows, cols)
            hip.conv2d_mfma(groups, w, ksz, cout, hip.BF16 if dt == torch.bfloat16 else hip.F16, stride=stride, bias=b, act=act,
                            slope=slope, slope_t=slope_t, res_scale=res_scale, pixel_shuffle=ps, gc_wmask=gc_wmask,
                            name=name)
            return True
        w, b, cout, _ = self._weights(name, "direct", False, rows, cols)
        # exact f32 on the matrix cores (v_mfma_f32_32x32x2_f32) for the 3x3 / 1x1 stride-1 layers with a dense source of a
        # multiple of 32 channels; the direct VALU kernel for the rest (skinny / strided / concatenated / pixel-shuffled layers)
        wm = bm = None
        if (ksz in (1, 3) and stride == 1 and not direct and all(len(g["srcs"]) == 1 and g["srcs"][0].shape[3] % 32 == 0
                                                                  for g in groups) and (not ps or cout % 16 == 0)):
            wm, bm = self._weights(name, "f32mfma", ps, rows, cols)[:2]
        for g in groups:
            hip.conv2d(g["srcs"], w, ksz, cout, g["dst"], bias=b, stride=stride, act=act, slope=slope, slope_t=slope_t,
                       res=g.get("res", ()), res_scale=res_scale, pixel_shuffle=ps, name=name, w_f32mfma=wm, bias_f32mfma=bm)
        return False

    def _tap(self, name, t_nhwc):
        if self.taps is not None:
            self.taps[name] = t_nhwc.float().permute(0, 3, 1, 2).contiguous().clone()

    @staticmethod
    def _new(dev, *shape, dtype=torch.float32):
        return torch.empty(*shape, device=dev, dtype=dtype)

    def _channel_sum(self, t):
        B, H, W, Cc = t.shape
        nblk = (H * W + 255) // 256
        out = self._new(t.device, B, Cc)
        scratch = self._new(t.device, B * nblk * Cc)
        v = view(t)
        check(lib().fcvsr_channel_sum(C.byref(v), B, H, W, out.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                      stream_ptr()), "fcvsr_channel_sum")
        return out

    def _ca_gate(self, sums, inv_hw, prefix, Bn, c):
        par = self._par
        w1, w2 = par[prefix + ".conv_du.0.weight"], par[prefix + ".conv_du.2.weight"]
        gate = self._new(sums.device, Bn, c)
        check(lib().fcvsr_ca_gate(sums.data_ptr(), inv_hw, w1.data_ptr(), w2.data_ptr(), Bn, c, w1.shape[0],
                                  gate.data_ptr(), stream_ptr()), "fcvsr_ca_gate")
        return gate

    # ---------------------------------------------------------------------------------------------- MGAAbk
    def _mgaa(self, x1, x2, x3, tag):
        """x1,x2,x3: (B,H,W,n) views (reference MGAAbk.forward, CVSR_freq.py:1442-1547)."""
        m = self._model()
        n, A = m.n_feats, m.ACNum
        dev = x1.device
        B, H, W, _ = x1.shape
        Wf = W // 2 + 1
        L = lib()
        st = stream_ptr()
        par = self._par

        # three spectra, each [imag(n), real(n)] per pixel.  Separate tensors, not one 6n-channel record: the FFT passes
        # then write whole 512-byte pixel records (two channel chunks side by side) instead of a third of a 1536-byte one
        spec = self._new(dev, 3, B, H, Wf, 2 * n)
        fs = 2 * n                                               # pixel stride of a spectrum, in floats
        for i, xi in enumerate((x1, x2, x3)):
            v = view(xi)
            check(L.fcvsr_rfft2(C.byref(v), B, H, W, n, spec[i].data_ptr(), fs, 0, n, st), "fcvsr_rfft2")
        x1f, x2f, x3f = spec[0], spec[1], spec[2]

        # offset spectra: (x?f - x2f) + convfuse(cat[x?f, x2f]); batch index = dir*B + b
        fdt = self._adt(freq=True)
        # the offset spectra feed only convcorr.0: stored in its operand dtype (bit-identical results, half the bytes)
        off = self._new(dev, 2 * B, H, Wf, 2 * n, dtype=fdt)
        fuse_mlp = fdt != torch.float32 and n == 64 and getattr(m, "fuse_freq_mlp", True)
        if fuse_mlp:
            # the whole convfuse stack for both directions in one kernel: hidden tensors never leave the CU
            ws = [self._weights(f"MGAA.convfuse.{i}", torch.bfloat16)[0] for i in (0, 2, 4)]
            P2 = C.c_void_p * 2
            check(L.fcvsr_freq_mlp3(P2(x1f.data_ptr(), x3f.data_ptr()), P2(x2f.data_ptr(), x2f.data_ptr()), 2, fs,
                                    B * H * Wf, ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                    P2(off[:B].data_ptr(), off[B:].data_ptr()), 2 * n, st), "fcvsr_freq_mlp3")
        else:
            # both directions per launch: two problem groups sharing the weights (forward: x1f vs x2f, backward: x3f vs x2f)
            t0 = self._new(dev, 2 * B, H, Wf, 2 * n, dtype=fdt)
            t1 = self._new(dev, 2 * B, H, Wf, 2 * n, dtype=fdt)
            dirs = list(enumerate((x1f, x3f)))
            self._convg("MGAA.convfuse.0", [dict(srcs=[xa, x2f], dst=t0[d * B:(d + 1) * B]) for d, xa in dirs],
                        act=ACT_RELU, freq=True)
            self._convg("MGAA.convfuse.2", [dict(srcs=[t0], dst=t1)], act=ACT_RELU, freq=True)
            self._convg("MGAA.convfuse.4", [dict(srcs=[t1[d * B:(d + 1) * B]], dst=off[d * B:(d + 1) * B], res=[xa, x2f])
                                            for d, xa in dirs], res_scale=[1.0, -1.0], freq=True)
        sim = self._new(dev, B, H, Wf, 4)
        off4 = self._new(dev, 2 * B, H, Wf, 4)
        fuse_head = fuse_mlp and getattr(m, "fuse_freq_head", True)
        if fuse_head:
            # convcrt (128 -> 64 -> 4 on the centre spectrum) as one launch
            check(L.fcvsr_freq_head(x2f.data_ptr(), hip.F32, fs, B * H * Wf,
                                    self._weights("MGAA.convcrt.0", torch.bfloat16)[0].data_ptr(), None,
                                    self._weights("MGAA.convcrt.2", torch.bfloat16)[0].data_ptr(), sim.data_ptr(), st),
                  "fcvsr_freq_head(convcrt)")
        else:
            s0 = self._new(dev, B, H, Wf, n, dtype=fdt)
            self._conv("MGAA.convcrt.0", [x2f], s0, act=ACT_RELU, freq=True)
            self._conv("MGAA.convcrt.2", [s0], sim, freq=True)

        if fdt == torch.float32:
            c0 = self._new(dev, 2 * B, H, Wf, n, dtype=fdt)
            c1 = self._new(dev, 2 * B, H, Wf, n, dtype=fdt)
            corr = self._new(dev, B, H, Wf, 84)                  # 81 live channels + 3 zero pad (16-byte pixels)
            cv = view(corr)
            check(L.fcvsr_corr_lookup(x1f.data_ptr(), x2f.data_ptr(), fs, B, H, Wf, 2 * n, 4, Wf, C.byref(cv), st),
                  "fcvsr_corr_lookup")
            for d in range(2):
                self._conv("MGAA.convcorr.0", [off[d * B:(d + 1) * B], corr], c0[d * B:(d + 1) * B], act=ACT_RELU, freq=True)
            self._conv("MGAA.convcorr.2", [c0], c1, act=ACT_RELU, freq=True)
            self._conv("MGAA.convcorr.4", [c1], off4, freq=True)
        else:
            # The CorrBlock lookup is identically zero beyond column radius+1 = 5 (it samples a 2-pixel-wide image,
            # CVSR_freq.py:1318-1337), so its 81 input channels contribute exact zeros to convcorr.0 everywhere else:
            # the stack runs over both directions on the offset spectra alone, then the narrow strip x < 8 is recomputed
            # with the lookup channels and pasted over - the same sums as the full concat (zeros add nothing to an f32 chain).
            if fuse_head:
                check(L.fcvsr_freq_head(off.data_ptr(), hip.BF16, 2 * n, 2 * B * H * Wf,
                                        self._weights("MGAA.convcorr.0", torch.bfloat16, cols=(0, 2 * n))[0].data_ptr(),
                                        self._weights("MGAA.convcorr.2", torch.bfloat16)[0].data_ptr(),
                                        self._weights("MGAA.convcorr.4", torch.bfloat16)[0].data_ptr(), off4.data_ptr(), st),
                      "fcvsr_freq_head(convcorr)")
            else:
                c0 = self._new(dev, 2 * B, H, Wf, n, dtype=fdt)
                c1 = self._new(dev, 2 * B, H, Wf, n, dtype=fdt)
                self._conv("MGAA.convcorr.0", [off], c0, act=ACT_RELU, freq=True, cols=(0, 2 * n))
            xs = min(Wf, 8)
            corr = self._new(dev, B, H, xs, 84)
            cv = view(corr)
            check(L.fcvsr_corr_lookup(x1f.data_ptr(), x2f.data_ptr(), fs, B, H, Wf, 2 * n, 4, xs, C.byref(cv), st),
                  "fcvsr_corr_lookup")
            off_s = off[:, :, :xs].to(torch.float32, memory_format=torch.contiguous_format)   # (2B,H,xs,2n) strip copy
            c0_s = self._new(dev, 2 * B, H, xs, n, dtype=fdt)
            for d in range(2):
                self._conv("MGAA.convcorr.0", [off_s[d * B:(d + 1) * B], corr], c0_s[d * B:(d + 1) * B], act=ACT_RELU,
                           freq=True)
            if fuse_head:
                c1_s = self._new(dev, 2 * B, H, xs, n, dtype=fdt)
                off4_s = self._new(dev, 2 * B, H, xs, 4)
                self._conv("MGAA.convcorr.2", [c0_s], c1_s, act=ACT_RELU, freq=True)
                self._conv("MGAA.convcorr.4", [c1_s], off4_s, freq=True)
                off4[:, :, :xs].copy_(off4_s)
            else:
                c0[:, :, :xs].copy_(c0_s)
                self._conv("MGAA.convcorr.2", [c0], c1, act=ACT_RELU, freq=True)
                self._conv("MGAA.convcorr.4", [c1], off4, freq=True)

        # A multi-scale ConvBlk heads -> (real, imag) planes -> irfft2 -> pixel offsets
        ospec = self._new(dev, B, H, Wf, 8 * A)                  # re: [0,4A), im: [4A,8A); channel = (dir*A+i)*2 + j
        u = self._new(dev, 2 * B, H, Wf, 4)
        ntile = ((H + 15) // 16) * ((Wf + 15) // 16)
        partial = self._new(dev, 2 * B * ntile * 4)
        for i in range(A):
            pre = f"MGAA.MConvB.{i}"
            # conv1 + PReLU + conv2 + channel sums in one launch, CALayer gate + (. * sim) + plane split in a second
            w1 = self._weights(pre + ".conv1", "direct")[0]
            w2 = self._weights(pre + ".conv2", "direct")[0]
            check(L.fcvsr_convblk(off4.data_ptr(), w1.data_ptr(), w2.data_ptr(), par[pre + ".relu.weight"].data_ptr(),
                                  par[pre + ".conv1.weight"].shape[-1], par[pre + ".CA.conv_du.0.weight"].data_ptr(),
                                  par[pre + ".CA.conv_du.2.weight"].data_ptr(), sim.data_ptr(), B, 2, H, Wf, u.data_ptr(),
                                  partial.data_ptr(), partial.numel(), ospec.data_ptr(), 8 * A, 0, 4 * A, A, i, st),
                  "fcvsr_convblk")
        offsets = self._new(dev, B, H, W, 4 * A)
        ov = view(offsets)
        check(L.fcvsr_irfft2(ospec.data_ptr(), 8 * A, 4 * A, 0, B, H, W, 4 * A, None, None, C.byref(ov), st),
              "fcvsr_irfft2")

        # kernel predictor (only the F1 half of F[1] is ever read)
        kp = self._new(dev, B, H, W, n, dtype=self._adt())
        k0 = self._new(dev, B, H, W, n, dtype=self._adt())
        # the adaptive kernels feed only the IAC kernel: in the 16-bit modes they are stored in the MFMA dtype
        fused_iac = n % 32 == 0                       # fused kernel works on 32-channel slabs; else 3-kernel f32 path
        self._conv("MGAA.conv_KP", [x2], kp)
        self._conv("MGAA.F.0", [kp], k0)
        # 16-bit modes, n = 64: F[1] (a 1x1 convolution) is folded into the IAC kernel - the A*3n adaptive-kernel channels
        # (1152 bytes per pixel) are computed per tile and never stored
        fold_f1 = fused_iac and n == 64 and self.precision != "f32" and getattr(self._model(), "fold_f1", True)
        if not fold_f1:
            K = self._new(dev, B, H, W, A * 3 * n, dtype=self._adt() if fused_iac else torch.float32)
            self._conv("MGAA.F.1", [k0], K)

        # iterative alignment: warp -> SAC(kernel1 twice) -> + feat_in -> LeakyReLU(0.1)
        fdt_act = x1.dtype                               # f32, or the 16-bit activation dtype (trunk16)
        al = self._new(dev, B, H, W, 2 * n, dtype=fdt_act)
        ping = [self._new(dev, B, H, W, n, dtype=fdt_act), self._new(dev, B, H, W, n, dtype=fdt_act)]
        if not fused_iac:
            if fdt_act != torch.float32:
                raise RuntimeError("16-bit activations need the fused IAC kernel (n_features % 32 == 0)")
            s = self._new(dev, B, H, W, n)
            vbuf = self._new(dev, B, H, W, n)
        if fused_iac and n % 64 == 0:
            # both directions of an iteration in one launch: they share the adaptive kernels, which are read once
            ping2 = [self._new(dev, B, H, W, n, dtype=fdt_act), self._new(dev, B, H, W, n, dtype=fdt_act)]
            V2 = hip.View * 2
            cur = [x1, x3]
            fins = V2(view(x1), view(x3))
            if fold_f1:
                wk, bk, _, _ = self._weights("MGAA.F.1", self._adt())
                k0_v = view(k0)
            for i in range(A):
                if not fold_f1:
                    k_v = view(K[..., i * 3 * n:(i + 1) * 3 * n])
                if i == A - 1:
                    dsts = [al[..., :n], al[..., n:]]
                else:
                    dsts = [ping[i % 2], ping2[i % 2]]
                prevs = V2(view(cur[0]), view(cur[1]))
                offs = V2(view(offsets[..., 2 * i:2 * i + 2]), view(offsets[..., 2 * (A + i):2 * (A + i) + 2]))
                dv2 = V2(view(dsts[0]), view(dsts[1]))
                if fold_f1:
                    check(L.fcvsr_iac_step2_fused(prevs, offs, C.byref(k0_v), wk.data_ptr() + i * 3 * n * 64 * 2,
                                                  bk.data_ptr() + i * 3 * n * 4, fins, 0.1, B, H, W, dv2, st),
                          "fcvsr_iac_step2_fused")
                else:
                    check(L.fcvsr_iac_step2(prevs, offs, C.byref(k_v), fins, 0.1, B, H, W, dv2, st), "fcvsr_iac_step2")
                cur = dsts
        else:
            for d, fin in enumerate((x1, x3)):
                cur = fin
                fv = view(fin)
                for i in range(A):
                    g = d * A + i
                    o_v = view(offsets[..., 2 * g:2 * g + 2])
                    k_v = view(K[..., i * 3 * n:(i + 1) * 3 * n])
                    dst = al[..., d * n:(d + 1) * n] if i == A - 1 else ping[i % 2]
                    cur_v, d_v = view(cur), view(dst)
                    if fused_iac:
                        check(L.fcvsr_iac_step(C.byref(cur_v), C.byref(o_v), C.byref(k_v), C.byref(fv), 0.1, B, H, W,
                                               C.byref(d_v), st), "fcvsr_iac_step")
                    else:
                        s_v, v_v = view(s), view(vbuf)
                        check(L.fcvsr_warp(C.byref(cur_v), C.byref(o_v), B, H, W, C.byref(s_v), st), "fcvsr_warp")
                        check(L.fcvsr_sac_v(C.byref(s_v), C.byref(k_v), B, H, W, C.byref(v_v), st), "fcvsr_sac_v")
                        check(L.fcvsr_sac_h(C.byref(v_v), C.byref(k_v), C.byref(fv), 0.1, B, H, W, C.byref(d_v), st),
                              "fcvsr_sac_h")
                    cur = dst
        out = self._new(dev, B, H, W, n, dtype=fdt_act)
        self._conv("MGAA.conv3", [al], out, res=[x2])
        if self.taps is not None:
            self._tap(f"mgaa{tag}.off_f", off4[:B])
            self._tap(f"mgaa{tag}.off_b", off4[B:])
            self._tap(f"mgaa{tag}.sim", sim)
            o = offsets.permute(0, 3, 1, 2).reshape(B, 2, A, 2, H, W)
            self.taps[f"mgaa{tag}.offsets_f"] = o[:, 0].contiguous().clone()
            self.taps[f"mgaa{tag}.offsets_b"] = o[:, 1].contiguous().clone()
            self._tap(f"mgaa{tag}.al_f", al[..., :n])
            self._tap(f"mgaa{tag}.al_b", al[..., n:])
            self._tap(f"mgaa{tag}.out", out)
        return out

    # ---------------------------------------------------------------------------------------------- MFFR
    def _mffr(self, x, out_dtype=torch.float32):
        """MultiFreq_Refinment.forward (CVSR_freq.py:2201-2254) on dense NHWC x."""
        m = self._model()
        n, Q = m.n_feats, m.Freq_Inv
        dev = x.device
        B, H, W, _ = x.shape
        Wf = W // 2 + 1
        L = lib()
        st = stream_ptr()
        par = self._par
        masks = self._mask(Q, H, W, dev)
        spec = self._new(dev, B, H, Wf, 2 * n)
        work = self._new(dev, B, H, Wf, 2 * n)                   # (band-by-band path)
        xv = view(x)
        check(L.fcvsr_rfft2(C.byref(xv), B, H, W, n, spec.data_ptr(), 2 * n, 0, n, st), "fcvsr_rfft2")
        bands = self._new(dev, Q, B, H, W, n)
        if masks.is_contiguous() and getattr(m, "fuse_bands", True):
            # all Q masked inverse transforms in one call: the spectrum columns are read once (fcvsr_irfft2_bands)
            work = self._new(dev, Q, B, H, Wf, 2 * n)
            bvs = (hip.View * Q)(*[view(bands[q]) for q in range(Q)])
            check(L.fcvsr_irfft2_bands(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, masks.data_ptr(), Q, work.data_ptr(), bvs, st),
                  "fcvsr_irfft2_bands")
        else:
            for q in range(Q):
                bv = view(bands[q])
                check(L.fcvsr_irfft2(spec.data_ptr(), 2 * n, 0, n, B, H, W, n, masks[q].data_ptr(), work.data_ptr(),
                                     C.byref(bv), st), "fcvsr_irfft2")
        freq = [bands[Q - 1 - i] for i in range(Q)]               # 'l2h' => reversed band list (:2204-2205)
        s_f = self._new(dev, B, H, W, n)
        s_o = self._new(dev, B, H, W, n)
        nblk = (H * W + 255) // 256
        scratch = self._new(dev, 2 * B * nblk * n)
        sums = self._new(dev, 2, B, n)
        inv_hw = 1.0 / (H * W)
        mean_sum = self._channel_sum(freq[0])
        # reduce(0); then per band: gates from the current sums, apply fused with the reduction the NEXT step needs (the
        # next band's e1/e2 sums, or the channel sums of s_o for the final CALayer) - s_f, s_o make one round trip per band
        ab = [(par[f"MFFRblock.DivEnh_block.{i}.a"], par[f"MFFRblock.DivEnh_block.{i}.b"]) for i in range(Q)]
        check(L.fcvsr_divenh(0, 1, freq[0].data_ptr(), s_f.data_ptr(), s_o.data_ptr(), ab[0][0].data_ptr(),
                             ab[0][1].data_ptr(), mean_sum.data_ptr(), inv_hw, None, None, sums.data_ptr(),
                             scratch.data_ptr(), scratch.numel(), B, H, W, n, st), "fcvsr_divenh(reduce)")
        for i in range(Q):
            pre = f"MFFRblock.DivEnh_block.{i}"
            a, b = ab[i]
            first = 1 if i == 0 else 0
            g1 = self._ca_gate(sums[0], inv_hw, pre + ".ca", B, n)
            g2 = self._ca_gate(sums[1], inv_hw, pre + ".ca", B, n) if i > 0 else None
            sums = self._new(dev, 2, B, n)                        # the gates above were computed from the previous buffer
            nxt = i + 1 < Q
            check(L.fcvsr_divenh_apply_next(first, freq[i].data_ptr(), s_f.data_ptr(), s_o.data_ptr(), a.data_ptr(),
                                            b.data_ptr(), mean_sum.data_ptr(), inv_hw, g1.data_ptr(), ptr(g2),
                                            freq[i + 1].data_ptr() if nxt else None,
                                            ab[i + 1][0].data_ptr() if nxt else None,
                                            ab[i + 1][1].data_ptr() if nxt else None, sums.data_ptr(),
                                            scratch.data_ptr(), scratch.numel(), B, H, W, n, st),
                  "fcvsr_divenh_apply_next")
        g = self._ca_gate(sums[0], inv_hw, "MFFRblock.ca", B, n)
        out = self._new(dev, B, H, W, n, dtype=out_dtype)
        check(L.fcvsr_scale_add(s_o.data_ptr(), g.data_ptr(), x.data_ptr(), self._code(x.dtype), out.data_ptr(),
                                self._code(out_dtype), B, H, W, n, st), "fcvsr_scale_add")
        if self.taps is not None:
            self.taps["mffr.bands"] = torch.stack([f.float().permute(0, 3, 1, 2) for f in freq], 1).contiguous().clone()
            self._tap("mffr.out", out.float())
        return out

    # ---------------------------------------------------------------------------------------------- SCNetbk
    def _block_rcb(self, pre, xs):
        """BlockRCB.forward (CVSR_freq.py:766-777) on the 3-level pyramid xs (dense NHWC)."""
        m = self._model()
        n = m.n_feats
        L = lib()
        st = stream_ptr()
        par = self._par
        def like(x, c, dtype=torch.float32):
            return self._new(x.device, x.shape[0], x.shape[1], x.shape[2], c, dtype=dtype)

        # the four 3x3 convs of the block run once per layer over all three pyramid levels (shared weights, one launch)
        t1 = [like(x, 2 * n, self._adt()) for x in xs]      # consumed only by the next conv -> MFMA operand dtype
        tdt = self._tdt()
        t2 = [like(x, n, tdt) for x in xs]
        r1 = [like(x, n, self._adt()) for x in xs]
        # RCB.body.2's output r is read once more (gc_apply): with the level-grouped tail it is stored like the trunk
        r16 = self.precision != "f32" and tdt != torch.float32 and getattr(m, "pool_first", True)
        rr = [like(x, n, tdt if r16 else torch.float32) for x in xs]
        self._convg(pre + ".body.0", [dict(srcs=[x], dst=t) for x, t in zip(xs, t1)], act=ACT_LEAKY, slope=0.1)
        self._convg(pre + ".body.2", [dict(srcs=[a], dst=t) for a, t in zip(t1, t2)])
        self._convg(pre + ".RCB.body.0", [dict(srcs=[a], dst=t) for a, t in zip(t2, r1)], act=ACT_LEAKY, slope=0.2)
        # ContextBlock: in the MFMA modes the softmax-pooling partials come out of the conv epilogue (no extra pass over r)
        wmask = par[pre + ".RCB.gcnet.conv_mask.weight"]
        w1g, w2g = par[pre + ".RCB.gcnet.channel_add_conv.0.weight"], par[pre + ".RCB.gcnet.channel_add_conv.2.weight"]
        nparts = [((x.shape[1] + 3) // 4) * ((x.shape[2] + 31) // 32) for x in xs]
        parts = [self._new(x.device, x.shape[0], npt, n + 2) for x, npt in zip(xs, nparts)]
        if r16 and n == 64 and getattr(m, "gc_separate", True):
            # r is stored in 16 bit anyway: run the layer without the ContextBlock epilogue (it then goes to the resident-weight
            # kernel: 102 vs 185 us) and compute the softmax-pool partials from the stored r in one launch for all levels
            self._convg(pre + ".RCB.body.2", [dict(srcs=[a], dst=t) for a, t in zip(r1, rr)])
            pl = (hip.GcPartialLevel * 3)()
            for l, x in enumerate(xs):
                pl[l].r, pl[l].partial = rr[l].data_ptr(), parts[l].data_ptr()
                pl[l].B, pl[l].H, pl[l].W = x.shape[0], x.shape[1], x.shape[2]
            check(L.fcvsr_gc_partial_levels(pl, 3, self._code(rr[0].dtype), wmask.data_ptr(), n, st), "fcvsr_gc_partial_levels")
            fused_gc = True
        else:
            fused_gc = self._convg(pre + ".RCB.body.2", [dict(srcs=[a], dst=t, gc_partial=pt) for a, t, pt in zip(r1, rr, parts)],
                                   gc_wmask=wmask if self.precision != "f32" else None)
        if fused_gc and getattr(m, "pool_first", True):
            return self._block_rcb_tail_levels(pre, xs, t2, rr, parts, nparts, w1g, w2g, tdt)
        R = []
        for l, x in enumerate(xs):
            dev = x.device
            B, H, W, _ = x.shape
            r = rr[l]
            add = self._new(dev, B, n)
            if fused_gc:
                check(L.fcvsr_gc_finish(parts[l].data_ptr(), nparts[l], w1g.data_ptr(), w2g.data_ptr(), B, n,
                                        add.data_ptr(), st), "fcvsr_gc_finish")
            else:
                nblk = (H * W + 255) // 256
                scratch = self._new(dev, B * nblk * (n + 2))
                check(L.fcvsr_gc_context(r.data_ptr(), wmask.data_ptr(), w1g.data_ptr(), w2g.data_ptr(), B, H, W, n,
                                         add.data_ptr(), scratch.data_ptr(), scratch.numel(), st), "fcvsr_gc_context")
            Rl = self._new(dev, B, H, W, n, dtype=tdt)
            check(L.fcvsr_gc_apply(r.data_ptr(), add.data_ptr(), t2[l].data_ptr(), Rl.data_ptr(), self._code(tdt), 0.2, B,
                                   H, W, n, st), "fcvsr_gc_apply")
            R.append(Rl)
        dn = [torch.empty_like(R[l]) for l in (0, 1)]
        up = [torch.empty_like(R[l]) for l in (1, 2)]
        self._convg(pre + ".down.0", [dict(srcs=[R[l]], dst=dn[l]) for l in (0, 1)])
        self._convg(pre + ".up.0", [dict(srcs=[R[l]], dst=up[l - 1]) for l in (1, 2)])
        outs = []
        for l, x in enumerate(xs):
            B, H, W, _ = x.shape
            y = torch.empty_like(x)
            rs = 2.0 if l in (0, 2) else 1.0
            d = dn[l - 1] if l >= 1 else None
            u = up[l] if l <= 1 else None
            check(L.fcvsr_xscale(x.data_ptr(), R[l].data_ptr(), rs, ptr(d), ptr(u), y.data_ptr(), self._code(tdt), B, H, W,
                                 n, st), "fcvsr_xscale")
            outs.append(y)
        return outs

    def _block_rcb_tail_levels(self, pre, xs, t2, rr, parts, nparts, w1g, w2g, tdt):
        """Second half of BlockRCB (:722-725, :766-777) with one launch per step for all three pyramid levels, and the
        down path evaluated as conv1x1(avgpool2(R)) instead of avgpool2(conv1x1(R)): the two commute (both are linear, the
        averaging weights sum to one so the bias passes through) and the convolution then runs on a quarter of the pixels."""
        m = self._model()
        n = m.n_feats
        L = lib()
        st = stream_ptr()
        dev = xs[0].device
        B = xs[0].shape[0]
        code = self._code(tdt)
        adds = [self._new(dev, B, n) for _ in xs]
        fl = (hip.GcFinishLevel * 3)()
        for l in range(3):
            fl[l].partial, fl[l].add, fl[l].nparts = parts[l].data_ptr(), adds[l].data_ptr(), nparts[l]
        check(L.fcvsr_gc_finish_levels(fl, 3, w1g.data_ptr(), w2g.data_ptr(), B, n, st), "fcvsr_gc_finish_levels")
        P = [self._new(dev, B, xs[l].shape[1] // 2, xs[l].shape[2] // 2, n, dtype=tdt) for l in (0, 1)]
        if (getattr(m, "fuse_rcb_l0", True) and tdt != torch.float32 and rr[0].dtype == tdt and xs[0].dtype == tdt
                and n % 8 == 0):
            return self._block_rcb_tail_l0(pre, xs, t2, rr, adds, P, tdt)
        R = [torch.empty_like(t) for t in t2]
        al = (hip.GcApplyLevel * 3)()
        for l in range(3):
            al[l].r, al[l].add, al[l].z, al[l].out = rr[l].data_ptr(), adds[l].data_ptr(), t2[l].data_ptr(), R[l].data_ptr()
            al[l].pool = P[l].data_ptr() if l < 2 else None
            al[l].B, al[l].H, al[l].W = B, xs[l].shape[1], xs[l].shape[2]
        check(L.fcvsr_gc_apply_levels(al, 3, code, self._code(rr[0].dtype), 0.2, n, st), "fcvsr_gc_apply_levels")
        dn = [torch.empty_like(P[l]) for l in (0, 1)]           # dn[l] lives at level l+1's resolution
        up = [torch.empty_like(R[l]) for l in (1, 2)]
        self._convg(pre + ".down.0", [dict(srcs=[P[l]], dst=dn[l]) for l in (0, 1)])
        self._convg(pre + ".up.0", [dict(srcs=[R[l]], dst=up[l - 1]) for l in (1, 2)])
        outs = [torch.empty_like(x) for x in xs]
        xl = (hip.XscaleLevel * 3)()
        for l in range(3):
            xl[l].x, xl[l].r, xl[l].out = xs[l].data_ptr(), R[l].data_ptr(), outs[l].data_ptr()
            xl[l].dn = dn[l - 1].data_ptr() if l >= 1 else None
            xl[l].up = up[l].data_ptr() if l <= 1 else None
            xl[l].r_scale = 2.0 if l in (0, 2) else 1.0
            xl[l].dn_pooled = 1
            xl[l].B, xl[l].H, xl[l].W = B, xs[l].shape[1], xs[l].shape[2]
        check(L.fcvsr_xscale_levels(xl, 3, code, n, st), "fcvsr_xscale_levels")
        return outs

    def _block_rcb_tail_l0(self, pre, xs, t2, rr, adds, P, tdt):
        """Same result as the generic sequence above (bit for bit), with level 0's R never stored: levels 1 and 2 go through
        gc_apply / xscale as before, level 0 through fcvsr_rcb_level0 once up.0(R1) exists (832 -> 576 bytes per level-0
        pixel at 64 channels)."""
        m = self._model()
        n = m.n_feats
        L = lib()
        st = stream_ptr()
        B = xs[0].shape[0]
        code = self._code(tdt)
        R = [None, torch.empty_like(t2[1]), torch.empty_like(t2[2])]
        al = (hip.GcApplyLevel * 3)()
        for i, l in enumerate((1, 2)):
            al[i].r, al[i].add, al[i].z, al[i].out = rr[l].data_ptr(), adds[l].data_ptr(), t2[l].data_ptr(), R[l].data_ptr()
            al[i].pool = P[1].data_ptr() if l == 1 else None
            al[i].B, al[i].H, al[i].W = B, xs[l].shape[1], xs[l].shape[2]
        check(L.fcvsr_gc_apply_levels(al, 2, code, code, 0.2, n, st), "fcvsr_gc_apply_levels")
        up = [torch.empty_like(R[l]) for l in (1, 2)]
        self._convg(pre + ".up.0", [dict(srcs=[R[l]], dst=up[l - 1]) for l in (1, 2)])
        outs = [torch.empty_like(x) for x in xs]
        check(L.fcvsr_rcb_level0(xs[0].data_ptr(), rr[0].data_ptr(), adds[0].data_ptr(), t2[0].data_ptr(), up[0].data_ptr(),
                                 outs[0].data_ptr(), P[0].data_ptr(), 0.2, 2.0, code, B, xs[0].shape[1], xs[0].shape[2], n, st),
              "fcvsr_rcb_level0")
        dn = [torch.empty_like(P[l]) for l in (0, 1)]
        self._convg(pre + ".down.0", [dict(srcs=[P[l]], dst=dn[l]) for l in (0, 1)])
        xl = (hip.XscaleLevel * 3)()
        for i, l in enumerate((1, 2)):
            xl[i].x, xl[i].r, xl[i].out = xs[l].data_ptr(), R[l].data_ptr(), outs[l].data_ptr()
            xl[i].dn = dn[l - 1].data_ptr()
            xl[i].up = up[1].data_ptr() if l == 1 else None
            xl[i].r_scale = 2.0 if l == 2 else 1.0
            xl[i].dn_pooled = 1
            xl[i].B, xl[i].H, xl[i].W = B, xs[l].shape[1], xs[l].shape[2]
        check(L.fcvsr_xscale_levels(xl, 2, code, n, st), "fcvsr_xscale_levels")
        return outs

    def _scnet(self, xs):
        m = self._model()
        cur = xs
        for g in range(m.SCGroupN):
            t = cur
            for k in range(3):
                t = self._block_rcb(f"recorb1.body.{g}.body.{k}", t)
            last = g == m.SCGroupN - 1
            # the trunk may be 16-bit (trunk16); SCNetbk's outputs leave in f32
            # (the net outputs stay f32: upconv_fuse concatenates o0 with two narrow f32 tensors and a launch takes sources
            # of one dtype)
            nxt = [torch.empty_like(c, dtype=torch.float32 if last else c.dtype) for c in cur]
            # fold SCNetbk's outer skip (x + body(x), :817-821) into the last group conv's epilogue
            grp = [dict(srcs=[t[l]], dst=nxt[l], res=([cur[l], xs[l]] if last else [cur[l]])) for l in range(3)]
            self._convg(f"recorb1.body.{g}.conv", grp)
            cur = nxt
        return cur

    # ---------------------------------------------------------------------------------------------- top level
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        m = self._model()
        with torch.no_grad():
            return self._forward_checked(x, m)

    def _forward_checked(self, x, m):
        if x.dim() != 5:
            raise ValueError(f"expected (B,T,C,H,W) input, got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("fcvsr_amd runs on MI355X only: move the model and the input to a HIP device "
                               "(there is no CPU fallback)")
        B, T, Cimg, H, W = x.shape
        if T != m._in_frames or Cimg != m._img_ch:
            raise ValueError(f"expected {m._in_frames} frames of {m._img_ch} channel(s), got T={T}, C={Cimg}")
        if H % 4 or W % 4:
            raise ValueError("H and W must be multiples of 4 (3-level pyramid, reference BlockRCB :766-777)")
        dev = x.device
        with torch.cuda.device(dev):
            x = x.contiguous().float()
            self._refresh(dev)
            ns = max(1, min(int(getattr(m, "streams", 1)), B))
            flags = tuple(bool(getattr(m, f, True)) for f in ("trunk16", "fold_f1", "fuse_tail", "pool_first", "fuse_freq_mlp",
                                                              "fuse_freq_head", "fast_feat", "fuse_rcb_tail", "gc_separate", "fast_last",
                                                              "fuse_rcb_l0", "fuse_bands"))
            cfg = (tuple(x.shape[1:]), self.precision, str(dev), self._pack_epoch, flags)
            if ns > 1 and cfg not in self._warm:
                # First pass of a configuration: re-packed weights, band masks and per-kernel attributes are created lazily
                # inside the forward.  They must be built on the CALLER's stream, before the fan-out: side streams are only
                # ordered after the caller's stream, not after each other, so a tensor first written on side stream 0
                # would be read by the other side streams with no dependency (and would live in stream 0's allocator pool).
                out = self._run(x, m, 1, dev)
                self._warm.add(cfg)
                if not getattr(m, "use_graph", False) or self.taps is not None or hip.PROFILE is not None:
                    return out
            self._warm.add(cfg)
            if not getattr(m, "use_graph", False) or self.taps is not None or hip.PROFILE is not None:
                return self._run(x, m, ns, dev)
            # hipGraph mode: the ~650 launches of one forward are captured once per (shape, precision, streams, weights
            # version) and replayed, which removes the host launch cost (~9 us per ctypes launch) from the critical path.
            key = (tuple(x.shape), ns) + cfg[1:]
            ent = self._graphs.get(key)
            if ent is None:
                sx = x.clone()
                for _ in range(2):                       # eager warm-up on the capture configuration
                    self._run(sx, m, ns, dev)
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    so = self._run(sx, m, ns, dev)
                # bounded cache: each entry pins a captured graph, its private memory pool and static in/out tensors
                # (ragged last batches and new resolutions would otherwise grow it without bound)
                while len(self._graphs) >= max(1, int(getattr(m, "graph_cache_size", 4))):
                    self._graphs.popitem(last=False)
                ent = self._graphs[key] = (graph, sx, so)
            else:
                self._graphs.move_to_end(key)
            graph, sx, so = ent
            sx.copy_(x)
            graph.replay()
            return so.clone()

    def _run(self, x, m, ns, dev):
        B, T, Cimg, H, W = x.shape
        if ns == 1:
            return self._forward(x, m, B, T, Cimg, H, W, dev)
        # Clips are independent: run sub-batches on separate HIP streams.  Every kernel of the path has serial phases
        # (stage -> MFMA -> store); with several forwards in flight the hardware interleaves workgroups of different
        # kernels, so HBM-bound and MFMA-bound phases of different sub-batches overlap and launch tails are filled.
        # Worth +4-5 % at B=16.  (Kernels of several HW queues then share the CUs: see DESIGN.md "Multi-stream replays" for
        # the packed-FP32 / LDS hazard this exposed and how the build avoids it.)
        out = self._new(dev, B, Cimg, 4 * H, 4 * W)
        cur = torch.cuda.current_stream(dev)
        while len(self._streams) < ns:
            self._streams.append(torch.cuda.Stream(device=dev))
        bounds = [round(i * B / ns) for i in range(ns + 1)]
        for i in range(ns):
            st = self._streams[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                xs = x[bounds[i]:bounds[i + 1]]
                self._forward(xs, m, xs.shape[0], T, Cimg, H, W, dev, out=out[bounds[i]:bounds[i + 1]])
        for i in range(ns):
            cur.wait_stream(self._streams[i])
        return out

    def _forward(self, x, m, B, T, Cimg, H, W, dev, out=None):
        n = m.n_feats
        L = lib()
        st = stream_ptr()
        a_t = m.lrelu.weight
        xin = x.view(B, T * Cimg, H, W).permute(0, 2, 3, 1)      # (b,y,x,c) strided view of the NCHW frames
        # 16-bit activation storage with trunk16 (feat_extract multiplies in f16 - 8-bit pixels stay exact - and stores its
        # output in the mode's activation dtype)
        # feat_extract is launched per frame group so that the two outer groups land in ONE batch-stacked tensor: the first
        # two MGAA calls of the reference (same weights, independent inputs f1 and f3, :2623-2624) run as one call on 2B.
        adt = self._tdt()
        f2 = self._new(dev, B, H, W, n, dtype=adt)
        cin = T * Cimg
        if adt != torch.float32 and cin == 7 and n == 64 and getattr(m, "fast_feat", True):
            # the whole 3x3 patch fits one K = 64 GEMM step: dedicated kernel, the 7 output blocks go straight to their tensors.
            # The three inputs of the stacked MGAA call are three DENSE tensors (2B,H,W,n) = [clip's group 1 | clip's group 3],
            # not channel slices of one 3n-channel tensor: every later reader (FFT rows, IAC taps, conv_KP, conv3's residual)
            # then streams whole DRAM pages instead of a third of each.
            p13 = self._new(dev, 3, 2 * B, H, W, n, dtype=adt)
            wf, bf = self._feat_weights()
            xv = view(xin)
            nb = 7 * n // 64
            P = C.c_void_p * nb
            ptrs = [p13[k, :B].data_ptr() for k in range(3)] + [f2.data_ptr()] + [p13[k, B:].data_ptr() for k in range(3)]
            check(L.fcvsr_feat_extract(C.byref(xv), B, H, W, wf.data_ptr(), ptr(bf), nb, P(*ptrs),
                                       (C.c_int64 * nb)(*([n] * nb)), (C.c_int32 * nb)(*([0] * nb)), self._code(adt), st),
                  "fcvsr_feat_extract")
            x1s, x2s, x3s = p13[0], p13[1], p13[2]
        else:
            f13 = self._new(dev, 2 * B, H, W, 3 * n, dtype=adt)  # [f1 of every clip | f3 of every clip]
            self._conv("feat_extract.0", [xin], f13[:B], force_f16=True, rows=(0, 3 * n))
            self._conv("feat_extract.0", [xin], f13[B:], force_f16=True, rows=(4 * n, 7 * n))
            self._conv("feat_extract.0", [xin], f2, force_f16=True, rows=(3 * n, 4 * n))
            x1s, x2s, x3s = f13[..., :n], f13[..., n:2 * n], f13[..., 2 * n:]
        if self.taps is not None:
            self._tap("feat", torch.cat([x1s[:B].float(), x2s[:B].float(), x3s[:B].float(), f2.float(),
                                         x1s[B:].float(), x2s[B:].float(), x3s[B:].float()], dim=3))
        a13 = self._mgaa(x1s, x2s, x3s, "13")
        if self.taps is not None:                                  # split the stacked taps back into calls "1" and "3"
            for k in [k for k in self.taps if k.startswith("mgaa13.")]:
                v = self.taps.pop(k)
                self.taps["mgaa1." + k[7:]], self.taps["mgaa3." + k[7:]] = v[:B].contiguous(), v[B:].contiguous()
        a2 = self._mgaa(a13[:B], f2, a13[B:], "2")
        tdt = self._tdt()
        d0 = self._mffr(a2, out_dtype=tdt)
        d1 = self._new(dev, B, H // 2, W // 2, n, dtype=tdt)
        d2 = self._new(dev, B, H // 4, W // 4, n, dtype=tdt)
        self._conv("rconcat1", [d0], d1, stride=2)
        self._conv("rconcat2", [d1], d2, stride=2)
        o0, o1, o2 = self._scnet([d0, d1, d2])
        self._tap("sc.o0", o0), self._tap("sc.o1", o1), self._tap("sc.o2", o2)

        # pyramid fuse (:2633-2639); PReLU commutes with PixelShuffle (one shared scalar slope)
        l3_1 = self._new(dev, B, H // 2, W // 2, n // 4)
        self._conv("upconv1_L3", [o2], l3_1, act=ACT_PRELU, slope_t=a_t, ps=True)
        l3_2 = self._new(dev, B, H, W, n // 16)
        check(L.fcvsr_pixel_shuffle(l3_1.data_ptr(), l3_2.data_ptr(), B, H // 2, W // 2, n // 4, st),
              "fcvsr_pixel_shuffle")
        l2 = self._new(dev, B, H // 2, W // 2, n)
        self._conv("upconv1_L2", [o1], l2, act=ACT_PRELU, slope_t=a_t)
        l2p = self._new(dev, B, H, W, n // 4)
        self._conv("upconv1_L2_2", [l2, l3_1], l2p, res=[l2], ps=True)
        fz0 = self._new(dev, B, H, W, n, dtype=self._adt())
        fz = self._new(dev, B, H, W, n, dtype=self._adt())          # read only by upconv1 (MFMA)
        self._conv("upconv_fuse", [o0, l2p, l3_2], fz0)
        self._conv("recorb0", [fz0], fz)
        self._tap("fz", fz)

        # up-sampler (:2641-2645)
        u1 = self._new(dev, B, 2 * H, 2 * W, n, dtype=self._adt())
        self._conv("upconv1", [fz], u1, act=ACT_PRELU, slope_t=a_t, ps=True)
        if out is None:
            out = self._new(dev, B, Cimg, 4 * H, 4 * W)           # NCHW boundary tensor
        out_v = out.permute(0, 2, 3, 1)
        centre = x[:, T // 2].permute(0, 2, 3, 1)                 # (B,H,W,Cimg) view of the centre LR frame
        cv, ov = view(centre), view(out_v)
        check(L.fcvsr_bilinear_up4(C.byref(cv), B, H, W, C.byref(ov), st), "fcvsr_bilinear_up4")
        fuse_tail = (self.precision != "f32" and n == 64 and Cimg == 1 and self._par["upconv2.weight"].shape[-1] == 1
                     and getattr(m, "fuse_tail", True))
        if fuse_tail:
            # upconv2 (1x1) + PixelShuffle + PReLU + conv_last0 in one kernel: the 64-channel tensor at 4H x 4W is never stored
            dt = self._adt()
            w2, b2, _, _ = self._weights("upconv2", dt, True)
            wl = self._tap_weights("conv_last0", dt)
            bl = self._par.get("conv_last0.bias")
            u1v = view(u1)
            check(L.fcvsr_tail_fused(C.byref(u1v), w2.data_ptr(), ptr(b2), a_t.data_ptr(), wl.data_ptr(), ptr(bl), B,
                                     2 * H, 2 * W, C.byref(ov), st), "fcvsr_tail_fused")
        else:
            u2 = self._new(dev, B, 4 * H, 4 * W, n, dtype=self._adt())
            self._conv("upconv2", [u1], u2, act=ACT_PRELU, slope_t=a_t, ps=True)
            if self.precision != "f32" and n == 64 and Cimg <= 3 and getattr(m, "fast_last", True):
                # 3x3 up-convs (full / RGB models): conv_last0 as a memory-bound "taps are MFMA columns" pass over u2
                wl = self._last_weights("conv_last0", self._adt(), Cimg)
                u2v = view(u2)
                check(L.fcvsr_conv_last(C.byref(u2v), wl.data_ptr(), ptr(self._par.get("conv_last0.bias")), B, 4 * H, 4 * W, Cimg,
                                        C.byref(ov), st), "fcvsr_conv_last")
            else:
                self._conv("conv_last0", [u2], out_v, res=[out_v])
        if self.taps is not None:
            self.taps["out"] = out.clone()
        return out
