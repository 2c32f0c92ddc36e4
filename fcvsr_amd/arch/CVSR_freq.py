"""Drop-in boundary: ``from fcvsr_amd.arch.CVSR_freq import GShiftNet, GShiftNet_S``.

Mirrors the constructor signature, ``forward`` signature and ``state_dict`` schema (303 tensors for S, 678 for the
full model, including the aliased ``...body.3.*`` / ``...RCB.*`` duplicates and the never-used ``DivEnh.Conv``) of the
reference classes (reference: CVSR_train/arch/CVSR_freq.py:2577-2646 ``GShiftNet_S``, :2653-2756 ``GShiftNet``), so that
``model.load_state_dict(torch.load(ckpt))``, ``model.to(device)``, ``model.parameters()`` and ``model(lrs)`` in the
reference's train/test scripts (test_LD_freqCVSR_S_22.py:51-79) work unchanged.

The modules below are *parameter containers only* (same names, shapes, default initialisation and construction order
as the reference, so a seeded from-scratch init consumes the RNG identically).  No arithmetic happens in them:
``forward`` hands raw device pointers to the hand-written HIP kernels in libfcvsr_hip.so via ``fcvsr_amd.engine``.
There is no CPU / eager fallback - a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.init as init


class _Holder(nn.Module):
    """Base for parameter containers: calling one directly is a bug (the engine owns the arithmetic)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter container; run the top-level model instead")


def _kaiming_scaled(mods, scale):
    """reference initialize_weights (CVSR_freq.py:635-652): Kaiming-normal(fan_in) * scale, zero bias."""
    for net in mods:
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                init.kaiming_normal_(m.weight, a=0, mode="fan_in")
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.zero_()


class CALayer(_Holder):  # :1812-1828
    def __init__(self, channel, reduction=16, bias=False):
        super().__init__()
        self.conv_du = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=bias), nn.ReLU(inplace=True),
            nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=bias), nn.Sigmoid())


class ConvBlk(_Holder):  # :344-357
    def __init__(self, dim=64, index=1):
        super().__init__()
        self.k_size = 2 * index + 1
        self.conv1 = nn.Conv2d(dim, dim, self.k_size, stride=1, padding=self.k_size // 2, bias=False)
        self.conv2 = nn.Conv2d(dim, dim, self.k_size, stride=1, padding=self.k_size // 2, bias=False)
        self.relu = nn.PReLU()
        self.CA = CALayer(dim, 1, bias=False)


class MGAAbk(_Holder):  # :1365-1430
    def __init__(self, dim, wiF=1.5, AC_Ks=3, ACNum=6, bias=False):
        super().__init__()
        self.dim, self.wiF, self.AC_Ks, self.ACNum = dim, wiF, AC_Ks, ACNum
        self.convfuse = nn.Sequential(
            nn.Conv2d(4 * dim, 2 * dim, 1, bias=bias), nn.ReLU(inplace=True),
            nn.Conv2d(2 * dim, 2 * dim, 1, bias=bias), nn.ReLU(inplace=True),
            nn.Conv2d(2 * dim, 2 * dim, 1, bias=bias))
        self.convcorr = nn.Sequential(
            nn.Conv2d(2 * dim + 83, dim, 1, bias=bias), nn.ReLU(inplace=True),
            nn.Conv2d(dim, dim, 1, bias=bias), nn.ReLU(inplace=True),
            nn.Conv2d(dim, 4, 1, bias=bias))
        self.MConvB = nn.ModuleList([ConvBlk(dim=4, index=i) for i in range(ACNum)])
        self.convcrt = nn.Sequential(
            nn.Conv2d(2 * dim, dim, 1, bias=bias), nn.ReLU(inplace=True), nn.Conv2d(dim, 4, 1, bias=bias))
        self.conv_KP = nn.Conv2d(dim, dim, 3, padding=1)
        self.kernel_dim = ACNum * (dim * AC_Ks * 2)
        self.F = nn.Sequential(nn.Conv2d(dim, dim, 3, padding=1), nn.Conv2d(dim, self.kernel_dim, 1, padding=0))
        self.conv3 = nn.Conv2d(2 * dim, dim, 3, padding=1, bias=bias)


class ContextBlock(_Holder):  # :657-701
    def __init__(self, n_feat, bias=False):
        super().__init__()
        self.conv_mask = nn.Conv2d(n_feat, 1, 1, bias=bias)
        self.channel_add_conv = nn.Sequential(
            nn.Conv2d(n_feat, n_feat, 1, bias=bias), nn.LeakyReLU(0.2), nn.Conv2d(n_feat, n_feat, 1, bias=bias))


class RCB(_Holder):  # :705-725
    def __init__(self, n_feat, bias=False):
        super().__init__()
        act = nn.LeakyReLU(0.2)
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat, 3, 1, 1, bias=bias), act,
                                  nn.Conv2d(n_feat, n_feat, 3, 1, 1, bias=bias))
        self.act = act
        self.gcnet = ContextBlock(n_feat, bias=bias)


class _Interp(nn.Module):  # placeholder keeping Sequential indices of the reference's Interpolate (:623-632)
    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor


class BlockRCB(_Holder):  # :729-777
    def __init__(self, nf, kernel_size=3, width_multiplier=1):
        super().__init__()
        self.RCB = RCB(nf)
        body = [nn.Conv2d(nf, int(nf * width_multiplier), kernel_size, padding=kernel_size // 2),
                nn.LeakyReLU(negative_slope=0.1, inplace=True),
                nn.Conv2d(int(nf * width_multiplier), nf, kernel_size, padding=kernel_size // 2),
                self.RCB]                      # same module registered twice -> aliased state_dict keys (:736,751)
        _kaiming_scaled(body, 0.1)
        self.body = nn.Sequential(*body)
        self.down = nn.Sequential(nn.Conv2d(nf, nf, 1), _Interp(0.5))
        self.up = nn.Sequential(nn.Conv2d(nf, nf, 1), _Interp(2.0))
        _kaiming_scaled([self.up, self.down], 0.1)


class SCGroupbk(_Holder):  # :781-803
    def __init__(self, nf=64, back_RBs=3):
        super().__init__()
        self.conv = nn.Conv2d(nf, nf, 3, padding=1)
        self.body = nn.Sequential(*[BlockRCB(nf, kernel_size=3, width_multiplier=2) for _ in range(back_RBs)])


class SCNetbk(_Holder):  # :807-822
    def __init__(self, nf=64, SCGroupN=4):
        super().__init__()
        self.body = nn.Sequential(*[SCGroupbk(nf=nf) for _ in range(SCGroupN)])


class DivEnh(_Holder):  # :2104-2133
    def __init__(self, channel):
        super().__init__()
        self.Conv = nn.Conv2d(channel, channel, 3, stride=1, padding=1)   # never used by forward
        self.a = nn.Parameter(torch.zeros(channel, 1, 1))
        self.b = nn.Parameter(torch.ones(channel, 1, 1))
        self.ca = CALayer(channel)


class _Split(nn.Module):  # Split_freq has no parameters or buffers (mask is a plain attribute, :2014)
    pass


class MultiFreq_Refinment(_Holder):  # :2183-2199
    def __init__(self, dim, Freq_Inv=8, mode="gaussian", freq_order="l2h"):
        super().__init__()
        if mode != "gaussian" or freq_order != "l2h":
            raise ValueError("only the configuration used by GShiftNet(_S) is built: gaussian masks, 'l2h' order")
        self.split = _Split()
        self.Freq_Inv = Freq_Inv
        self.DivEnh_block = nn.ModuleList([DivEnh(channel=dim) for _ in range(Freq_Inv)])
        self.ca = CALayer(dim)


class _GShiftBase(nn.Module):
    _up_k = 1
    _in_frames = 7
    _img_ch = 1

    def __init__(self, n_features, wiF, AC_Ks, ACNum, Freq_Inv, SCGroupN):
        super().__init__()
        if n_features % 16:
            raise ValueError("n_features must be a multiple of 16 (CALayer r=16, PixelShuffle splits)")
        if AC_Ks != 3:
            raise ValueError("AC_Ks must be 3 (the separable adaptive kernels are 3-tap, CVSR_freq.py:1253)")
        n, k = n_features, self._up_k
        self.n_feats = n
        self.device = torch.device("cuda")
        self.wiF, self.AC_Ks, self.ACNum, self.Freq_Inv, self.SCGroupN = wiF, AC_Ks, ACNum, Freq_Inv, SCGroupN
        cin = self._in_frames * self._img_ch
        self.feat_extract = nn.Sequential(nn.Conv2d(cin, 7 * n, 3, 1, 1))
        self.lrelu = nn.PReLU()
        self.MGAA = MGAAbk(dim=n, wiF=wiF, AC_Ks=AC_Ks, ACNum=ACNum)
        self.rconcat1 = nn.Conv2d(n, n, 3, stride=2, padding=1, bias=True)
        self.rconcat2 = nn.Conv2d(n, n, 3, stride=2, padding=1, bias=True)
        self.recorb1 = SCNetbk(nf=n, SCGroupN=SCGroupN)
        self.recorb0 = nn.Conv2d(n, n, 3, 1, 1, bias=True)
        self.upconv1_L2 = nn.Conv2d(n, n, k, 1, k // 2, bias=True)
        self.upconv1_L2_2 = nn.Conv2d(n + n // 4, n, k, 1, k // 2, bias=True)
        self.upconv1_L3 = nn.Conv2d(n, n, k, 1, k // 2, bias=True)
        self.upconv1 = nn.Conv2d(n, n * 4, k, 1, k // 2, bias=True)
        self.upconv2 = nn.Conv2d(n, n * 4, k, 1, k // 2, bias=True)
        self.pixel_shuffle = nn.PixelShuffle(2)
        self.conv_last0 = nn.Conv2d(n, self._img_ch, 3, 1, 1, bias=True)
        self.MFFRblock = MultiFreq_Refinment(dim=n, Freq_Inv=Freq_Inv, mode="gaussian")
        self.upconv_fuse = nn.Conv2d(n + n // 4 + n // 16, n, 3, 1, 1, bias=True)
        self._engine = None
        # arithmetic of the conv layers: "f32" = exact f32 (parity mode, default); "bf16" / "f16" = matrix cores with
        # 16-bit operands and f32 accumulation (activations stay f32 in HBM).  Not part of state_dict.
        self.precision = os.environ.get("FCVSR_PRECISION", "f32")
        # number of HIP streams a batch is split over (independent clips; overlaps memory- and MFMA-bound phases)
        self.streams = int(os.environ.get("FCVSR_STREAMS", "1"))
        # 16-bit modes only: store the spatial activations (extracted features, aligned features, SCNetbk trunk) in the
        # MFMA dtype instead of f32 - half the HBM bytes and staging instructions.  Spectra, offsets, MultiFreq_Refinment
        # internals, ContextBlock statistics and all accumulation stay f32.
        self.trunk16 = os.environ.get("FCVSR_TRUNK16", "1") == "1"
        # 16-bit modes, n_features == 64: compute the adaptive kernels (F[1]) inside the IAC kernel instead of storing them
        self.fold_f1 = os.environ.get("FCVSR_FOLD_F1", "1") == "1"
        # 16-bit modes, S model (1x1 up-convs), one image channel: upconv2 + PixelShuffle + PReLU + conv_last0 in one kernel
        self.fuse_tail = os.environ.get("FCVSR_FUSE_TAIL", "1") == "1"
        # 16-bit modes: BlockRCB's down path as conv1x1(avgpool2(R)) (they commute) and level-grouped elementwise launches
        self.pool_first = os.environ.get("FCVSR_POOL_FIRST", "1") == "1"
        # 16-bit modes, n_features == 64: the convfuse 1x1 stack of MGAAbk as one kernel (hidden tensors stay on chip)
        self.fuse_freq_mlp = os.environ.get("FCVSR_FUSE_FREQ_MLP", "1") == "1"
        # 16-bit modes, 9*Cin <= 64 (the Y models): feat_extract as a single-K-step GEMM kernel
        self.fast_feat = os.environ.get("FCVSR_FAST_FEAT", "1") == "1"
        # 16-bit modes, n_features == 64: convcrt and (away from the CorrBlock strip) convcorr as one launch each
        self.fuse_freq_head = os.environ.get("FCVSR_FUSE_FREQ_HEAD", "1") == "1"
        # 16-bit modes, 16-bit trunk: ContextBlock partials from the stored r in their own launch (the conv then runs on the
        # resident-weight kernel) instead of from the lean kernel's epilogue
        self.gc_separate = os.environ.get("FCVSR_GC_SEPARATE", "1") == "1"
        # MultiFreq_Refinment band split: all masked inverse transforms in one call (spectrum columns read once)
        self.fuse_bands = os.environ.get("FCVSR_FUSE_BANDS", "1") == "1"
        # 16-bit modes: BlockRCB's full-resolution level in one pass (R0 is never stored)
        self.fuse_rcb_l0 = os.environ.get("FCVSR_FUSE_RCB_L0", "1") == "1"
        # capture the launch sequence of a forward in a hipGraph (per input shape) and replay it
        self.use_graph = os.environ.get("FCVSR_GRAPH", "0") == "1"
        # captured graphs kept per (shape, precision, streams, flags): least-recently-used entries beyond this are freed
        self.graph_cache_size = 4

    def invalidate(self) -> None:
        """Forget every cached re-packed weight / captured hipGraph.  Needed only after in-place edits through
        ``param.data`` (EMA hooks, ``w.data.mul_()``): those do not bump ``Parameter._version``, which - with the
        storage pointer - keys the caches.  ``load_state_dict``, optimizer steps and ``.to()`` are detected."""
        if self._engine is not None:
            self._engine.invalidate()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: float (B, 7, C, H, W) in [0,1] on a HIP device -> (B, C, 4H, 4W).

        Under autograd (gradients enabled and a parameter or the input requires one - the reference's training loop,
        train_LD_freqCVSR_S_22.py:244-251) the differentiable graph of ``fcvsr_amd.train`` runs: HIP convolution kernels in
        the forward, input-gradient and weight-gradient directions, torch operators for the rest.  Otherwise (inference,
        ``torch.no_grad()``) every operator is a hand-written kernel driven by ``fcvsr_amd.engine``."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..train.graph import forward_train
            if x.dim() != 5 or x.shape[1] != self._in_frames or x.shape[2] != self._img_ch:
                raise ValueError(f"expected (B,{self._in_frames},{self._img_ch},H,W) input, got {tuple(x.shape)}")
            return forward_train(self.state_dict(keep_vars=True), x, precision=getattr(self, "train_precision", self.precision))
        from ..engine import Engine
        if self._engine is None or self._engine._model() is not self:      # (a deepcopy carries the source's engine)
            object.__setattr__(self, "_engine", Engine(self))
        return self._engine.forward(x)


class GShiftNet_S(_GShiftBase):
    """FCVSR-S, Y channel (reference CVSR_freq.py:2577-2646): 1x1 up-convs."""
    _up_k = 1

    def __init__(self, n_features=64, wiF=1.5, AC_Ks=3, ACNum=3, Freq_Inv=4, SCGroupN=4):
        super().__init__(n_features, wiF, AC_Ks, ACNum, Freq_Inv, SCGroupN)


class GShiftNet(_GShiftBase):
    """FCVSR, Y channel (reference CVSR_freq.py:2653-2756): 3x3 up-convs."""
    _up_k = 3

    def __init__(self, n_features=64, wiF=1.5, AC_Ks=3, ACNum=6, Freq_Inv=8, SCGroupN=10):
        super().__init__(n_features, wiF, AC_Ks, ACNum, Freq_Inv, SCGroupN)


class GShiftNet_ETC(GShiftNet):
    """Multi-window variant (reference CVSR_freq.py:2760-2843): the input holds 13 frames, the network of ``GShiftNet`` (same
    parameters and state_dict keys) super-resolves the 7 windows ``x[:, i:i+7]`` and returns
    ``(out_seq, x_up)``, both ``(B, 7, C, 4H, 4W)``: the SR frames and the bilinear x4 bases of the window centres.
    The windows are independent, so they are stacked on the batch axis and run as ONE forward instead of the reference's loop."""
    _windows = 7

    def forward(self, x: torch.Tensor):
        if x.dim() != 5 or x.shape[1] != self._in_frames + self._windows - 1 or x.shape[2] != self._img_ch:
            raise ValueError(f"expected (B,{self._in_frames + self._windows - 1},{self._img_ch},H,W) input, got {tuple(x.shape)}")
        B, _, C, H, W = x.shape
        T, Wn = self._in_frames, self._windows
        win = torch.stack([x[:, i:i + T] for i in range(Wn)], 1).reshape(B * Wn, T, C, H, W)
        out = super().forward(win).reshape(B, Wn, C, 4 * H, 4 * W)
        if not x.is_cuda:
            raise RuntimeError("fcvsr_amd runs on MI355X only (there is no CPU fallback)")
        # x_up: bilinear x4 (align_corners=False) of the window centres, frames 3..9, by the same kernel as the global skip
        import ctypes as C_
        from .. import hip
        centres = x[:, T // 2:T // 2 + Wn].reshape(B * Wn, C, H, W).float().contiguous()
        base = torch.empty((B * Wn, C, 4 * H, 4 * W), dtype=torch.float32, device=x.device)
        cv, ov = hip.view(centres.permute(0, 2, 3, 1)), hip.view(base.permute(0, 2, 3, 1))
        hip.check(hip.lib().fcvsr_bilinear_up4(C_.byref(cv), B * Wn, H, W, C_.byref(ov), hip.stream_ptr()), "fcvsr_bilinear_up4")
        return out, base.reshape(B, Wn, C, 4 * H, 4 * W)
