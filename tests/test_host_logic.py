"""CPU-only checks: schema parity with the reference state_dict, synthetic weights, host-side tables, and that the C-ABI
library loads and exports every symbol include/fcvsr_hip.h declares (no compute without a GPU)."""
import ctypes
import json
import os
import re

import pytest
import torch

from helpers import GOLDEN_DIR, load_schema

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("ctor", ["GShiftNet_S", "GShiftNet", "GShiftNet_ETC", "FCVSR_SNet", "FCVSRNet"])
def test_state_dict_schema_matches_reference(ctor):
    from fcvsr_amd.arch.schema import state_dict_shapes
    ref = {k: tuple(v) for k, v in load_schema()[ctor].items()}
    mine = state_dict_shapes(ctor)
    assert list(mine) == list(ref)          # same keys, same order (incl. aliased RCB/body.3 duplicates)
    assert mine == ref


def test_param_counts_match_survey():
    from fcvsr_amd.arch import CVSR_freq as A
    assert sum(p.numel() for p in A.GShiftNet_S().parameters()) == 3704709
    with torch.device("meta"):
        assert sum(p.numel() for p in A.GShiftNet().parameters()) == 8811336


def test_rgb_twin_param_counts_and_hooks():
    from fcvsr_amd.arch import fcvsr_rgb as R
    with torch.device("meta"):
        assert sum(p.numel() for p in R.FCVSR_SNet().parameters()) == 4130951
        assert sum(p.numel() for p in R.FCVSRNet().parameters()) == 8868938
        m = R.FCVSR_SNet()
    m.init_weights(None)
    with pytest.raises(TypeError):
        m.init_weights(123)


def test_ctor_signature_defaults():
    import inspect
    from fcvsr_amd.arch import CVSR_freq as A
    s = inspect.signature(A.GShiftNet_S.__init__)
    assert [(k, v.default) for k, v in list(s.parameters.items())[1:]] == [
        ("n_features", 64), ("wiF", 1.5), ("AC_Ks", 3), ("ACNum", 3), ("Freq_Inv", 4), ("SCGroupN", 4)]
    s = inspect.signature(A.GShiftNet.__init__)
    assert [(k, v.default) for k, v in list(s.parameters.items())[1:]] == [
        ("n_features", 64), ("wiF", 1.5), ("AC_Ks", 3), ("ACNum", 6), ("Freq_Inv", 8), ("SCGroupN", 10)]


def test_synthetic_weights_deterministic_and_alias_consistent():
    from fcvsr_amd.weights import synthetic_tensor
    a = synthetic_tensor("recorb1.body.0.body.1.body.3.body.0.weight", (64, 64, 3, 3))
    b = synthetic_tensor("recorb1.body.0.body.1.RCB.body.0.weight", (64, 64, 3, 3))
    assert torch.equal(a, b)
    assert torch.equal(a, synthetic_tensor("recorb1.body.0.body.1.RCB.body.0.weight", (64, 64, 3, 3)))
    assert abs(float(synthetic_tensor("lrelu.weight", (1,))) - 0.25) < 0.3


def test_band_masks_match_oracle():
    from fcvsr_amd.engine import band_masks_half
    from oracle import fcvsr_oracle as O
    for Q, H, W in [(4, 16, 20), (8, 20, 24), (2, 24, 16)]:
        assert torch.allclose(band_masks_half(Q, H, W), O.band_masks_half(Q, H, W), atol=0, rtol=0)


def test_library_exports_every_declared_symbol():
    from fcvsr_amd import hip
    from fcvsr_amd.build import build
    path = build()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "fcvsr_hip.h")).read()
    declared = set(re.findall(r"\b(fcvsr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fcvsr_hip.h but not exported"
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    assert hip.lib().fcvsr_abi_version() == 1


def test_ctypes_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof as seen by a C compiler reading include/fcvsr_hip.h == the ctypes mirror in fcvsr_amd/hip.py."""
    import subprocess
    from fcvsr_amd import hip
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "fcvsr_hip.h"\n'
        'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(fcvsr_view), sizeof(fcvsr_conv_desc),'
        ' offsetof(fcvsr_conv_desc, weight), offsetof(fcvsr_conv_desc, res), offsetof(fcvsr_conv_desc, dst),'
        ' offsetof(fcvsr_conv_desc, pixel_shuffle), sizeof(fcvsr_gc_finish_level), sizeof(fcvsr_gc_apply_level),'
        ' offsetof(fcvsr_gc_apply_level, B), sizeof(fcvsr_xscale_level), offsetof(fcvsr_xscale_level, r_scale),'
        ' offsetof(fcvsr_xscale_level, W)); return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    D = hip.ConvDesc
    assert got == [ctypes.sizeof(hip.View), ctypes.sizeof(D), D.weight.offset, D.res.offset, D.dst.offset,
                   D.pixel_shuffle.offset, ctypes.sizeof(hip.GcFinishLevel), ctypes.sizeof(hip.GcApplyLevel),
                   hip.GcApplyLevel.B.offset, ctypes.sizeof(hip.XscaleLevel), hip.XscaleLevel.r_scale.offset,
                   hip.XscaleLevel.W.offset]


def test_cpu_input_raises_no_fallback():
    from fcvsr_amd.arch import CVSR_freq as A
    m = A.GShiftNet_S()
    with pytest.raises(RuntimeError):
        m(torch.rand(1, 7, 1, 16, 16))


def test_sliding_window_indices():
    """reference generate_input_index (test_LD_freqCVSR_S_22.py:13-16): clip to [0, n-1] => edge replicate."""
    from fcvsr_amd.harness.windows import window_indices
    assert window_indices(0, 7, 10, "replicate") == [0, 0, 0, 0, 1, 2, 3]
    assert window_indices(9, 7, 10, "replicate") == [6, 7, 8, 9, 9, 9, 9]
    assert window_indices(5, 7, 10, "replicate") == [2, 3, 4, 5, 6, 7, 8]
    # mmedit GenerateFrameIndiceswithPadding (augmentation.py:856-877)
    assert window_indices(0, 5, 100, "reflection") == [2, 1, 0, 1, 2]
    assert window_indices(0, 5, 100, "reflection_circle") == [4, 3, 0, 1, 2]
    assert window_indices(99, 5, 100, "reflection") == [97, 98, 99, 98, 97]
    assert window_indices(99, 5, 100, "reflection_circle") == [97, 98, 99, 96, 95]


def test_build_keeps_packed_fp32_out_of_every_code_object(tmp_path):
    """The packed-FP32 / LDS hazard: every file is compiled without the SLP vectoriser, and no code
    object of the BUILT library contains a v_pk_{add,mul,fma}_f32 instruction (checked by disassembling the .so)."""
    import shutil
    import subprocess
    from fcvsr_amd import build as B
    for f in os.listdir(B.CSRC):
        if f.endswith(".hip"):
            assert "-fno-slp-vectorize" in B.flags_for(os.path.join(B.CSRC, f)), f
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if shutil.which(B.HIPCC) is None:
        pytest.skip("no hipcc on this machine: nothing was built here")
    # a machine that can BUILD the library must also be able to check it: a missing disassembler is a failure, not a skip
    assert os.path.exists(objdump), f"{objdump} is missing although hipcc exists: the packed-FP32 guard cannot run"
    so = tmp_path / "libfcvsr_hip.so"
    shutil.copy(B.build(), so)
    subprocess.check_call([objdump, "--offloading", str(so)], stdout=subprocess.DEVNULL, cwd=tmp_path)
    cos = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert len(cos) >= 10, cos
    n_lds = 0
    for co in cos:
        text = subprocess.check_output([objdump, "-d", str(tmp_path / co)]).decode()
        n_lds += "ds_read_b128" in text
        import re
        # every packed-FP32 VALU mnemonic of gfx950 (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32, v_pk_mov_b32 is integer/moves
        # and is not matched): any v_pk_*_f32
        m = re.search(r"\bv_pk_[a-z0-9_]*_f32\b", text)
        assert m is None, f"{m.group(0)} found in code object {co}"
    assert n_lds >= 5
