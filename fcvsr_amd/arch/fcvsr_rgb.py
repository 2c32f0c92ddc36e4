"""RGB twins of the FCVSR models as registered in the reference's MMEditing fork:
``FCVSRNet`` (reference mmedit_train/mmedit/models/backbones/sr_backbones/fcvsr.py:38-158) and
``FCVSR_SNet`` (.../fcvsr_s.py:41-158).  Same math as the Y models with 21 input channels (7 RGB frames), 3 output
channels and 3x3 up-convs in both sizes; same constructor defaults, ``forward(x[B,7,3,H,W]) -> [B,3,4H,4W]``,
``init_weights(pretrained=None, strict=True)`` and ``state_dict`` schema.

If an installed ``mmedit`` is importable the classes are registered in its ``BACKBONES`` registry so that the reference
configs (``generator=dict(type='FCVSR_SNet')``, configs/restorers/fcvsr/*.py) resolve to this implementation.
"""
from __future__ import annotations

import torch

from .CVSR_freq import _GShiftBase


class _RGBBase(_GShiftBase):
    _up_k = 3
    _img_ch = 3

    def init_weights(self, pretrained=None, strict=True):
        """mmedit backbone hook.  ``pretrained``: checkpoint path or None (reference fcvsr.py:145-158)."""
        if isinstance(pretrained, str):
            ckpt = torch.load(pretrained, map_location="cpu", weights_only=True)
            sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
            sd = {k[len("generator."):] if k.startswith("generator.") else k: v for k, v in sd.items()}
            self.load_state_dict(sd, strict=strict)
        elif pretrained is not None:
            raise TypeError(f'"pretrained" must be a str or None. But received {type(pretrained)}.')


class FCVSR_SNet(_RGBBase):
    def __init__(self, n_features=64, wiF=1.5, AC_Ks=3, ACNum=3, Freq_Inv=4, SCGroupN=4):
        super().__init__(n_features, wiF, AC_Ks, ACNum, Freq_Inv, SCGroupN)


class FCVSRNet(_RGBBase):
    def __init__(self, n_features=64, wiF=1.5, AC_Ks=3, ACNum=6, Freq_Inv=8, SCGroupN=10):
        super().__init__(n_features, wiF, AC_Ks, ACNum, Freq_Inv, SCGroupN)


def register_in_mmedit() -> bool:
    """Register both classes in mmedit's BACKBONES registry (no-op when mmedit is not installed)."""
    try:
        from mmedit.models.registry import BACKBONES  # type: ignore
    except Exception:
        return False
    for cls in (FCVSRNet, FCVSR_SNet):
        BACKBONES.register_module(module=cls, force=True)
    return True


register_in_mmedit()
