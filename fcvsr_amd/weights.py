"""Deterministic synthetic weights keyed by ``state_dict`` key name.

No trained FCVSR checkpoint exists offline (reference README.md:29-32 points at Baidu links), so
parity tests, golden fixtures and the benchmark all use the same reproducible fill: every tensor
is generated from a RandomState seeded by crc32 of its (canonical) key, so 15-35 MB of weights
never need to be shipped (protocol: SURVEY.md section 8c).
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, Mapping, Sequence

import numpy as np
import torch

_ALIAS = re.compile(r"\.body\.3\.(body|gcnet)\.")


def canonical_key(key: str) -> str:
    """``BlockRCB`` registers its RCB twice (reference CVSR_freq.py:736,751): ``...body.3.*`` and
    ``...RCB.*`` are the same tensor.  Map both spellings to the ``RCB`` one."""
    return _ALIAS.sub(r".RCB.\1.", key)


def synthetic_tensor(key: str, shape: Sequence[int], gain: float = 0.5) -> torch.Tensor:
    key = canonical_key(key)
    rs = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        if fan_in > 1:
            return torch.from_numpy((rs.standard_normal(shape) * gain / np.sqrt(fan_in)).astype(np.float32))
    if key.endswith("relu.weight"):      # nn.PReLU slopes (``lrelu.weight``, ``MConvB.i.relu.weight``)
        default = 0.25
    elif key.endswith(".b"):             # DivEnh.b
        default = 1.0
    else:                                # biases, DivEnh.a
        default = 0.0
    return torch.from_numpy((default + 0.05 * rs.standard_normal(shape)).astype(np.float32))


def synthetic_state_dict(shapes: Mapping[str, Sequence[int]], gain: float = 0.5) -> Dict[str, torch.Tensor]:
    return {k: synthetic_tensor(k, s, gain) for k, s in shapes.items()}
