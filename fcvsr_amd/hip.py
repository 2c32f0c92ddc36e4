"""ctypes binding of libfcvsr_hip.so (C ABI declared in include/fcvsr_hip.h).

PyTorch is used only as the owner of device memory and streams: tensors are handed to the library as raw device
pointers + strides (``fcvsr_view``) and every call enqueues on ``torch.cuda.current_stream()``.
There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfcvsr_hip.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_PRELU = 0, 1, 2, 3
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


class View(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("sb", C.c_int64), ("sy", C.c_int64), ("sx", C.c_int64), ("sc", C.c_int64),
                ("c", C.c_int32), ("dtype", C.c_int32)]


class GcFinishLevel(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("add", C.c_void_p), ("nparts", C.c_int32)]


class GcPartialLevel(C.Structure):
    _fields_ = [("r", C.c_void_p), ("partial", C.c_void_p), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


class GcApplyLevel(C.Structure):
    _fields_ = [("r", C.c_void_p), ("add", C.c_void_p), ("z", C.c_void_p), ("out", C.c_void_p), ("pool", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


class XscaleLevel(C.Structure):
    _fields_ = [("x", C.c_void_p), ("r", C.c_void_p), ("dn", C.c_void_p), ("up", C.c_void_p), ("out", C.c_void_p),
                ("r_scale", C.c_float), ("dn_pooled", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("n_src", C.c_int32), ("src", View * 3), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("cout", C.c_int32), ("weight", C.c_void_p), ("cout_pad", C.c_int32), ("bias", C.c_void_p),
                ("act", C.c_int32), ("slope", C.c_float), ("slope_ptr", C.c_void_p), ("n_res", C.c_int32),
                ("res", View * 2), ("res_scale", C.c_float * 2), ("dst", View), ("pixel_shuffle", C.c_int32),
                ("gc_wmask", C.c_void_p), ("gc_partial", C.c_void_p)]


# bench.py instrumentation: when PROFILE is a list, every conv launch is bracketed by HIP events on the launch stream and
# (start, stop, algorithmic_flops) is appended.  None (default) = no instrumentation.
PROFILE = None
COMPUTE_DTYPE = "f32"
DOMINANT_KERNEL = "conv_direct_kernel"


class HipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_VP, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_PV = C.POINTER(View)
SIGNATURES = {
    "fcvsr_last_error": [],
    "fcvsr_last_conv_kernel": [],
    "fcvsr_debug_res_stamps": [C.c_void_p, C.c_size_t],
    "fcvsr_conv2d_wgrad_scratch_elems": [C.c_int] * 7,
    "fcvsr_conv2d_wgrad": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                           C.c_longlong, C.c_void_p],
    "fcvsr_conv2d_wgrad_mfma_eligible": [C.c_int] * 6,
    "fcvsr_conv2d_wgrad_mfma_scratch_elems": [C.c_int] * 7,
    "fcvsr_conv2d_wgrad_mfma": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_longlong, C.c_void_p],
    "fcvsr_abi_version": [],
    "fcvsr_device_count": [],
    "fcvsr_conv2d": [C.POINTER(ConvDesc), _VP],
    "fcvsr_conv2d_f32mfma": [C.POINTER(ConvDesc), _VP],
    "fcvsr_conv2d_f32mfma_eligible": [C.POINTER(ConvDesc)],
    "fcvsr_conv2d_mfma": [C.POINTER(ConvDesc), _I, _I, _VP],
    "fcvsr_rfft2": [_PV, _I, _I, _I, _I, _VP, _I64, _I, _I, _VP],
    "fcvsr_irfft2": [_VP, _I64, _I, _I, _I, _I, _I, _I, _VP, _VP, _PV, _VP],
    "fcvsr_irfft2_bands": [_VP, _I64, _I, _I, _I, _I, _I, _I, _VP, _I, _VP, _PV, _VP],
    "fcvsr_corr_lookup": [_VP, _VP, _I64, _I, _I, _I, _I, _I, _I, _PV, _VP],
    "fcvsr_channel_sum": [_PV, _I, _I, _I, _VP, _VP, _I64, _VP],
    "fcvsr_ca_gate": [_VP, _F, _VP, _VP, _I, _I, _I, _VP, _VP],
    "fcvsr_convblk_tail": [_VP, _VP, _VP, _I, _I, _I, _I, _VP, _I64, _I, _I, _I, _I, _VP],
    "fcvsr_convblk": [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP],
    "fcvsr_warp": [_PV, _PV, _I, _I, _I, _PV, _VP],
    "fcvsr_sac_v": [_PV, _PV, _I, _I, _I, _PV, _VP],
    "fcvsr_sac_h": [_PV, _PV, _PV, _F, _I, _I, _I, _PV, _VP],
    "fcvsr_iac_step": [_PV, _PV, _PV, _PV, _F, _I, _I, _I, _PV, _VP],
    "fcvsr_feat_extract": [_PV, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP],
    "fcvsr_freq_head": [_VP, _I, _I64, _I64, _VP, _VP, _VP, _VP, _VP],
    "fcvsr_freq_mlp3": [_VP, _VP, _I, _I64, _I64, _VP, _VP, _VP, _VP, _I64, _VP],
    "fcvsr_iac_step2": [_PV, _PV, _PV, _PV, _F, _I, _I, _I, _PV, _VP],
    "fcvsr_iac_step2_fused": [_PV, _PV, _PV, _VP, _VP, _PV, _F, _I, _I, _I, _PV, _VP],
    "fcvsr_divenh": [_I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _VP, _VP, _VP, _I64, _I, _I, _I, _I, _VP],
    "fcvsr_divenh_apply_next": [_I, _VP, _VP, _VP, _VP, _VP, _VP, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _I, _I, _I, _I, _VP],
    "fcvsr_scale_add": [_VP, _VP, _VP, _I, _VP, _I, _I, _I, _I, _I, _VP],
    "fcvsr_gc_context": [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _I64, _VP],
    "fcvsr_gc_finish": [_VP, _I, _VP, _VP, _I, _I, _VP, _VP],
    "fcvsr_gc_apply": [_VP, _VP, _VP, _VP, _I, _F, _I, _I, _I, _I, _VP],
    "fcvsr_xscale": [_VP, _VP, _F, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP],
    "fcvsr_gc_finish_levels": [C.POINTER(GcFinishLevel), _I, _VP, _VP, _I, _I, _VP],
    "fcvsr_gc_partial_levels": [C.POINTER(GcPartialLevel), _I, _I, _VP, _I, _VP],
    "fcvsr_gc_apply_levels": [C.POINTER(GcApplyLevel), _I, _I, _I, _F, _I, _VP],
    "fcvsr_xscale_levels": [C.POINTER(XscaleLevel), _I, _I, _I, _VP],
    "fcvsr_rcb_level0": [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _F, _F, _I, _I, _I, _I, _I, _VP],
    "fcvsr_pixel_shuffle": [_VP, _VP, _I, _I, _I, _I, _VP],
    "fcvsr_bilinear_up4": [_PV, _I, _I, _I, _PV, _VP],
    "fcvsr_tail_fused": [_PV, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _PV, _VP],
    "fcvsr_conv_last": [_PV, _VP, _VP, _I, _I, _I, _I, _PV, _VP],
    "fcvsr_pack_weight_mfma": [_VP, _I, _I, _I, _I, _VP, _I, _I, _I, _I, _VP],
    "fcvsr_pack_weights_multi_block_elems": [],
    "fcvsr_pack_weights_mfma_multi": [_VP, _I, _I, _I, _VP],
    "fcvsr_act_bwd": [_VP, _VP, _VP, _F, C.c_longlong, _VP],
    "fcvsr_colsum_scratch_elems": [C.c_longlong, _I],
    "fcvsr_colsum": [_VP, C.c_longlong, _I, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_iac_bwd_sac": [_VP, _VP, _VP, _VP, _PV, _F, _I, _I, _I, _I, _VP, _I, _VP, _PV, _I, _VP],
    "fcvsr_iac_bwd_warp": [_VP, _PV, _VP, _PV, _I, _I, _I, _I, _VP, _VP, _VP],
    "fcvsr_prelu_fwd": [_VP, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_prelu_bwd": [_VP, _VP, _VP, _VP, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_wgrad_cout1_scratch_elems": [_I, _I, _I],
    "fcvsr_wgrad_cout1": [_VP, _VP, _I, _I, _I, _I, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_wgrad_set_bias_out": [_VP, _I],
    "fcvsr_conv2d_wgrad_mfma_groups_scratch_elems": [_VP, _VP, _VP, _I, _I, _I, _I, _I],
    "fcvsr_conv2d_wgrad_mfma_groups": [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_colsum_groups_scratch_elems": [_VP, _I, _I],
    "fcvsr_colsum_groups": [_VP, _VP, _I, _I, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_up2_adjoint": [_VP, _VP, _I, _I, _I, _I, _VP],
    "fcvsr_pool2_adjoint": [_VP, _VP, _I, _I, _I, _I, _VP],
    "fcvsr_wgrad_set_accumulate": [_I],
    "fcvsr_wgrad_get_accumulate": [],
    "fcvsr_colsum_set_accumulate": [_I],
    "fcvsr_rcbt_nblk": [_I],
    "fcvsr_rcbt_stat_elems": [],
    "fcvsr_rcbt_forward": [_VP, _VP, _VP, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_rcbt_backward": [_VP, _VP, _VP, _VP, _VP, _VP, _F, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, C.c_longlong, _I, _VP],
    "fcvsr_divenh_band_nblk": [_I],
    "fcvsr_divenh_band_stat_elems": [_I],
    "fcvsr_divenh_band_forward": [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, C.c_longlong, _VP],
    "fcvsr_divenh_band_backward": [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP,
                                   C.c_longlong, _I, _VP],
    "fcvsr_corr_lookup_bwd": [_VP, _VP, _I64, _I, _I, _I, _I, _I, _I, _PV, _VP, _VP, _VP],
}
_RESTYPES = {"fcvsr_wgrad_set_accumulate": None, "fcvsr_colsum_set_accumulate": None, "fcvsr_last_error": C.c_char_p, "fcvsr_last_conv_kernel": C.c_char_p, "fcvsr_conv2d_wgrad_scratch_elems": C.c_longlong,
             "fcvsr_conv2d_wgrad_mfma_scratch_elems": C.c_longlong, "fcvsr_colsum_scratch_elems": C.c_longlong,
             "fcvsr_wgrad_cout1_scratch_elems": C.c_longlong, "fcvsr_conv2d_wgrad_mfma_groups_scratch_elems": C.c_longlong,
             "fcvsr_colsum_groups_scratch_elems": C.c_longlong}


def lib() -> C.CDLL:
    """Load the shared library (built by ``fcvsr_amd.build``).  Raises if it is absent: there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} not found: run `python -m fcvsr_amd.build` (hipcc, gfx950). "
                           "fcvsr_amd has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name, None)
            if fn is None:                      # a library older than this binding: the call site raises AttributeError when reached
                continue                        # (tests/test_host_logic.py checks that the built library exports every declared symbol)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fcvsr_last_error()
        raise HipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def view(t: torch.Tensor) -> View:
    """View of a 4-D tensor whose logical dims are (b, y, x, c) with arbitrary strides (zero-copy)."""
    assert t.dim() == 4, t.shape
    sb, sy, sx, sc = t.stride()
    if t.shape[3] == 1:
        sc = 1                                   # the stride of a size-1 dimension is arbitrary in torch (channels_last with C = 1)
    return View(t.data_ptr(), sb, sy, sx, sc, t.shape[3], _DT[t.dtype])


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, kh, kw) -> [kh*kw][Cin][cout_pad] f32, cout padded to a multiple of 16 (zero rows)."""
    cout, cin, kh, kw = w.shape
    cp = (cout + 15) // 16 * 16
    out = torch.zeros(kh * kw, cin, cp, dtype=torch.float32, device=w.device)
    out[:, :, :cout] = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    return out.contiguous()


def pack_conv_weight_f32mfma(w: torch.Tensor, ps: bool = False) -> torch.Tensor:
    """(Cout, Cin, kh, kw) -> f32 [kh*kw][ceil64(Cout)][Cin] (layout of fcvsr_conv2d_f32mfma), zero rows past Cout; with
    ps=True the rows are in the sub-pixel-major order of the pixel-shuffle epilogue (`ps_order`)."""
    if ps:
        w = w[ps_order(w.shape[0]).to(w.device)]
    cout, cin, kh, kw = w.shape
    cop = (cout + 63) // 64 * 64
    out = torch.zeros(kh * kw, cop, cin, dtype=torch.float32, device=w.device)
    out[:, :cout] = w.detach().float().permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
    return out.contiguous()


def ps_order(cout: int) -> torch.Tensor:
    """Row permutation for pixel-shuffled MFMA layers: new row (2i+j)*(cout/4)+c <- original channel 4c+2i+j."""
    q = cout // 4
    return torch.tensor([4 * c + sp for sp in range(4) for c in range(q)], dtype=torch.long)


def pack_conv_weight_mfma(w: torch.Tensor, dtype: torch.dtype, ps: bool = False) -> torch.Tensor:
    """(Cout, Cin, kh, kw) -> 16-bit [kh*kw][ceil128(Cout)][ceil64(Cin)], zero padded (layout of fcvsr_conv2d_mfma)."""
    if ps:
        w = w[ps_order(w.shape[0]).to(w.device)]
    cout, cin, kh, kw = w.shape
    cop, cip = (cout + 127) // 128 * 128, (cin + 63) // 64 * 64
    out = torch.zeros(kh * kw, cop, cip, dtype=dtype, device=w.device)
    out[:, :cout, :cin] = w.detach().float().permute(2, 3, 0, 1).reshape(kh * kw, cout, cin).to(dtype)
    return out.contiguous()


def _fill_desc(d: "ConvDesc", srcs, wpacked, ksize, cout, cout_pad, dst, bias, stride, act, slope, slope_t, res,
               res_scale, pixel_shuffle) -> int:
    d.n_src = len(srcs)
    for i, s in enumerate(srcs):
        d.src[i] = view(s)
    d.B, d.H, d.W = srcs[0].shape[0], srcs[0].shape[1], srcs[0].shape[2]
    d.kh = d.kw = ksize
    d.stride = stride
    d.pad = ksize // 2
    d.cout = cout
    d.weight = wpacked.data_ptr()
    d.cout_pad = cout_pad
    d.bias = ptr(bias)
    d.act = act
    d.slope = slope
    d.slope_ptr = ptr(slope_t)
    d.n_res = len(res)
    for i, r in enumerate(res):
        d.res[i] = view(r)
        d.res_scale[i] = res_scale[i] if i < len(res_scale) else 1.0
    d.dst = view(dst)
    d.pixel_shuffle = int(pixel_shuffle)
    return sum(s.shape[3] for s in srcs)


def mfma_eligible(ksize: int, stride: int, groups) -> bool:
    """Can fcvsr_conv2d_mfma take this problem? (1x1/3x3, stride 1, channel-contiguous 16-byte-aligned f32 inputs)"""
    if ksize not in (1, 3) or stride not in (1, 2) or (stride == 2 and ksize != 3):
        return False
    for g in groups:
        if len(g["srcs"]) == 1 and g["srcs"][0].stride(3) != 1:       # planar (NCHW) source, e.g. feat_extract
            s0 = g["srcs"][0]
            if s0.dtype != torch.float32 or s0.shape[3] > 32 or ksize != 3 or stride != 1:
                return False
            continue
        for s in g["srcs"]:
            sb, sy, sx, sc = s.stride()
            gran = 4 if s.dtype == torch.float32 else 8
            if sc != 1 or s.shape[3] % gran or sx % gran or sy % gran or sb % gran or s.data_ptr() % 16:
                return False
            if ksize == 1 and (sy != sx * s.shape[2] or (s.shape[0] > 1 and sb != sy * s.shape[1])):
                return False
        for t in list(g.get("res", ())) + ([] if g.get("ps") else [g["dst"]]):
            sb, sy, sx, sc = t.stride()
            if ksize == 1 and (sy != sx * t.shape[2] or (t.shape[0] > 1 and sb != sy * t.shape[1])):
                return False
    return True


def conv2d_mfma(groups, wpacked: torch.Tensor, ksize: int, cout: int, mma_dtype: int, *, stride: int = 1,
                bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: float = 0.0,
                slope_t: Optional[torch.Tensor] = None, res_scale: Sequence[float] = (), pixel_shuffle: bool = False,
                gc_wmask: Optional[torch.Tensor] = None, name: str = ""):
    """groups: 1..3 dicts {srcs: [..], dst: t, res: [..]} sharing weights / epilogue (one launch)."""
    n = len(groups)
    descs = (ConvDesc * n)()
    flops = 0.0
    for i, g in enumerate(groups):
        cin = _fill_desc(descs[i], g["srcs"], wpacked, ksize, cout, wpacked.shape[1], g["dst"], bias, stride, act, slope,
                         slope_t, g.get("res", ()), res_scale, pixel_shuffle)
        assert wpacked.shape[0] == ksize * ksize and wpacked.shape[2] >= cin, (wpacked.shape, ksize, cin)
        if gc_wmask is not None:
            descs[i].gc_wmask = gc_wmask.data_ptr()
            descs[i].gc_partial = g["gc_partial"].data_ptr()
        flops += 2.0 * descs[i].B * descs[i].H * descs[i].W * cout * cin * ksize * ksize / (stride * stride)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().fcvsr_conv2d_mfma(descs, n, mma_dtype, stream_ptr()), "fcvsr_conv2d_mfma")
        e1.record()
        nbytes = sum(t.numel() * t.element_size() for g in groups for t in list(g["srcs"]) + list(g.get("res", ())) + [g["dst"]])
        # the kernel the dispatcher actually launched (fcvsr_last_conv_kernel): bench.py groups the timings by it
        kname = lib().fcvsr_last_conv_kernel().decode()
        PROFILE.append((e0, e1, flops, "mfma", name, nbytes, kname))
        return
    check(lib().fcvsr_conv2d_mfma(descs, n, mma_dtype, stream_ptr()), "fcvsr_conv2d_mfma")


F32_MFMA = os.environ.get("FCVSR_F32_MFMA", "1") == "1"     # exact-f32 layers on the matrix cores where eligible


def conv2d(srcs: Sequence[torch.Tensor], wpacked: torch.Tensor, ksize: int, cout: int, dst: torch.Tensor, *,
           bias: Optional[torch.Tensor] = None, stride: int = 1, act: int = ACT_NONE, slope: float = 0.0,
           slope_t: Optional[torch.Tensor] = None, res: Sequence[torch.Tensor] = (),
           res_scale: Sequence[float] = (), pixel_shuffle: bool = False, name: str = "",
           w_f32mfma: Optional[torch.Tensor] = None, bias_f32mfma: Optional[torch.Tensor] = None) -> torch.Tensor:
    """srcs / res / dst are (b,y,x,c)-ordered tensors (any strides).  Exact f32: the direct VALU kernel, or - when
    `w_f32mfma` (pack_conv_weight_f32mfma) is given and the layer qualifies - the f32-operand matrix-core kernel."""
    if w_f32mfma is not None and F32_MFMA and len(srcs) == 1 and stride == 1 and PROFILE is None:
        dm = ConvDesc()              # (pixel-shuffled layers: w_f32mfma / bias_f32mfma are in sub-pixel-major row order)
        _fill_desc(dm, srcs, w_f32mfma, ksize, cout, w_f32mfma.shape[1], dst, bias_f32mfma if pixel_shuffle else bias, stride, act,
                   slope, slope_t, res, res_scale, pixel_shuffle)
        if lib().fcvsr_conv2d_f32mfma_eligible(C.byref(dm)):
            check(lib().fcvsr_conv2d_f32mfma(C.byref(dm), stream_ptr()), "fcvsr_conv2d_f32mfma")
            return dst
    d = ConvDesc()
    cin = _fill_desc(d, srcs, wpacked, ksize, cout, wpacked.shape[-1], dst, bias, stride, act, slope, slope_t, res,
                     res_scale, pixel_shuffle)
    assert wpacked.shape[0] == ksize * ksize and wpacked.shape[1] == cin, (wpacked.shape, ksize, cin)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().fcvsr_conv2d(C.byref(d), stream_ptr()), "fcvsr_conv2d")
        e1.record()
        ho = (d.H + 2 * d.pad - d.kh) // stride + 1
        wo = (d.W + 2 * d.pad - d.kw) // stride + 1
        PROFILE.append((e0, e1, 2.0 * d.B * ho * wo * cout * cin * ksize * ksize, "direct", name, 0, "direct"))
        return dst
    check(lib().fcvsr_conv2d(C.byref(d), stream_ptr()), "fcvsr_conv2d")
    return dst
