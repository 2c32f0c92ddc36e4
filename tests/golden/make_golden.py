#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE model itself (build container only).

Usage:  python tests/golden/make_golden.py            (needs /root/reference; CPU, ~1 min)

The reference (CVSR_train/arch/CVSR_freq.py) is imported read-only from /root/reference with two
in-memory stubs for modules that are absent from the image:
  * ``cv2``            - only imported on the live path, never called;
  * ``torchvision.transforms.Resize`` - restated as torchvision-0.14.1's tensor path
    ``F.interpolate(img[None], size, mode='bicubic', align_corners=False, antialias=False)[0]``
    (the reference pins torchvision 0.14.1, README.md:14).  This is a recorded assumption.
Nothing from the reference is written into the repo: the outputs are data only
(inputs, per-block taps, outputs, and the state_dict key->shape schema).

Weights are the key-seeded synthetic fill of ``fcvsr_amd.weights`` (gain 0.5), so fixtures need
not carry them.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from fcvsr_amd.weights import synthetic_state_dict  # noqa: E402

REF_ROOT = "/root/reference/CVSR_train"


def import_reference():
    sys.dont_write_bytecode = True
    sys.modules["cv2"] = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class _IM:
        BICUBIC = "bicubic"

    def Resize(size, interpolation=None):
        return lambda img: F.interpolate(img[None], size=list(size), mode="bicubic",
                                         align_corners=False, antialias=False)[0]

    tvt.Resize = Resize
    tvt.functional = types.SimpleNamespace(InterpolationMode=_IM)
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF_ROOT)
    from arch import CVSR_freq as ref
    return ref


def import_reference_rgb():
    """mmedit twins: loaded by file path with stub modules for mmcv / mmedit.models.registry."""
    import importlib.util
    sys.modules["cv2"].imwrite = lambda *a, **k: False
    sys.modules.setdefault("mmcv", types.ModuleType("mmcv"))
    reg = types.ModuleType("mmedit.models.registry")

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    reg.BACKBONES = _Reg()
    for name in ("mmedit", "mmedit.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mmedit.models.registry"] = reg
    mods = {}
    base = "/root/reference/mmedit_train/mmedit/models/backbones/sr_backbones/"
    for fn in ("fcvsr.py", "fcvsr_s.py"):
        spec = importlib.util.spec_from_file_location("ref_" + fn[:-3], base + fn)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[fn] = m
    return mods


def run_case(ref, name, ctor, kwargs, x, tap_filter=None):
    torch.manual_seed(0)
    model = getattr(ref, ctor)(**kwargs)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synthetic_state_dict(shapes, gain=0.5)
    model.load_state_dict(sd, strict=True)
    taps = {}
    call = {"mgaa": 0}
    order = ["1", "3", "2"]

    def tag():
        return order[call["mgaa"]]

    hooks = []
    hooks.append(model.feat_extract.register_forward_hook(lambda m, i, o: taps.__setitem__("feat", o)))
    cc = {"n": 0}

    def h_convcorr(m, i, o):
        taps[f"mgaa{tag()}.off_{'fb'[cc['n'] % 2]}"] = o
        cc["n"] += 1

    hooks.append(model.MGAA.convcorr.register_forward_hook(h_convcorr))
    hooks.append(model.MGAA.convcrt.register_forward_hook(lambda m, i, o: taps.__setitem__(f"mgaa{tag()}.sim", o)))

    def h_mgaa(m, i, o):
        taps[f"mgaa{tag()}.out"] = o[0]
        call["mgaa"] += 1

    hooks.append(model.MGAA.register_forward_hook(h_mgaa))
    hooks.append(model.MFFRblock.split.register_forward_hook(
        lambda m, i, o: taps.__setitem__("mffr.bands", torch.stack(list(o[0])[::-1], 1))))
    hooks.append(model.MFFRblock.register_forward_hook(lambda m, i, o: taps.__setitem__("mffr.out", o)))

    def h_sc(m, i, o):
        taps["sc.o0"], taps["sc.o1"], taps["sc.o2"] = o

    hooks.append(model.recorb1.register_forward_hook(h_sc))
    hooks.append(model.recorb0.register_forward_hook(lambda m, i, o: taps.__setitem__("fz", o)))

    iac_n = {"n": 0}
    orig_iac = ref.IAC

    def iac_spy(feat_in, Pred_K, offsets_list, *a, **k):
        out = orig_iac(feat_in, Pred_K, offsets_list, *a, **k)
        d = "fb"[iac_n["n"] % 2]
        taps[f"mgaa{tag()}.offsets_{d}"] = torch.stack(list(offsets_list), 1)
        taps[f"mgaa{tag()}.al_{d}"] = out
        iac_n["n"] += 1
        return out

    ref.IAC = iac_spy
    try:
        with torch.no_grad():
            y = model(torch.from_numpy(x))
    finally:
        ref.IAC = orig_iac
        for h in hooks:
            h.remove()
    taps["out"] = y
    out = {"x": x}
    for k, v in taps.items():
        if tap_filter is None or any(k == s or (s.startswith(".") and k.endswith(s)) for s in tap_filter):
            out["tap:" + k] = v.detach().numpy().astype(np.float32)
    meta = dict(ctor=ctor, kwargs=kwargs, gain=0.5)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "taps:", len(out) - 2, "out mean/std", float(y.mean()), float(y.std()))
    return shapes


def run_etc(ref, name, x):
    """GShiftNet_ETC (CVSR_freq.py:2760-2843): 13 frames in, (out_seq, x_up) out; default full-size configuration."""
    torch.manual_seed(0)
    model = ref.GShiftNet_ETC()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(synthetic_state_dict(shapes, gain=0.5), strict=True)
    with torch.no_grad():
        out_seq, x_up = model(torch.from_numpy(x))
    meta = dict(ctor="GShiftNet_ETC", kwargs={}, gain=0.5)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x, **{"tap:out_seq": out_seq.numpy().astype(np.float32),
                        "tap:x_up": x_up.numpy().astype(np.float32)}, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(name, "out_seq", tuple(out_seq.shape), "mean/std", float(out_seq.mean()), float(out_seq.std()))
    return shapes


def main():
    ref = import_reference()
    rs = np.random.RandomState(1234)
    small = ("out", "mffr.out", ".sim", ".off_f", ".off_b", ".offsets_f", ".offsets_b", "mgaa2.out", "sc.o2", "fz")
    sS = run_case(ref, "S_16x20", "GShiftNet_S", {}, rs.rand(1, 7, 1, 16, 20).astype(np.float32))
    run_case(ref, "S_b2_72x36", "GShiftNet_S", {}, rs.rand(2, 7, 1, 72, 36).astype(np.float32),
             ("out", "mgaa2.out", ".offsets_f", ".offsets_b", "sc.o2"))
    sF = run_case(ref, "full_20x24", "GShiftNet", {}, rs.rand(1, 7, 1, 20, 24).astype(np.float32), small)
    run_case(ref, "S_cfg1_64x64", "GShiftNet_S", {},
             np.random.RandomState(0).rand(1, 7, 1, 64, 64).astype(np.float32), ("out", "mgaa2.out"))
    run_case(ref, "Sreduced_24x16", "GShiftNet_S", dict(n_features=32, ACNum=2, Freq_Inv=2, SCGroupN=1),
             rs.rand(1, 7, 1, 24, 16).astype(np.float32), small)
    sE = run_etc(ref, "etc_12x16", np.random.RandomState(77).rand(1, 13, 1, 12, 16).astype(np.float32))
    rgb = import_reference_rgb()
    sRS = run_case(rgb["fcvsr_s.py"], "rgbS_16x20", "FCVSR_SNet", {}, rs.rand(1, 7, 3, 16, 20).astype(np.float32),
                   ("out", "mgaa2.out", "sc.o2", "fz", "feat"))
    sRF = run_case(rgb["fcvsr.py"], "rgbfull_12x16", "FCVSRNet", {}, rs.rand(1, 7, 3, 12, 16).astype(np.float32),
                   ("out", "mgaa2.out"))
    with open(os.path.join(HERE, "schema.json"), "w") as f:
        json.dump({"GShiftNet_S": {k: list(v) for k, v in sS.items()},
                   "GShiftNet": {k: list(v) for k, v in sF.items()},
                   "GShiftNet_ETC": {k: list(v) for k, v in sE.items()},
                   "FCVSR_SNet": {k: list(v) for k, v in sRS.items()},
                   "FCVSRNet": {k: list(v) for k, v in sRF.items()}}, f, indent=0)


if __name__ == "__main__":
    main()
