"""state_dict schema (key -> shape) of the drop-in classes without allocating weights (meta device)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def state_dict_shapes(ctor: str, **kwargs) -> Dict[str, Tuple[int, ...]]:
    from . import CVSR_freq
    with torch.device("meta"):
        m = getattr(CVSR_freq, ctor)(**kwargs)
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}
