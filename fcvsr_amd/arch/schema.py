"""state_dict schema (key -> shape) of the drop-in classes without allocating weights (meta device)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def state_dict_shapes(ctor: str, **kwargs) -> Dict[str, Tuple[int, ...]]:
    from . import CVSR_freq, fcvsr_rgb
    mod = CVSR_freq if hasattr(CVSR_freq, ctor) else fcvsr_rgb
    with torch.device("meta"):
        m = getattr(mod, ctor)(**kwargs)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}
