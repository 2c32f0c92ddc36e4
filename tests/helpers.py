"""Shared test helpers (load golden fixtures, build key-seeded weights)."""
import json
import os

import numpy as np
import torch

from fcvsr_amd.weights import synthetic_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["S_16x20", "S_b2_72x36", "full_20x24", "S_cfg1_64x64", "Sreduced_24x16", "rgbS_16x20", "rgbfull_12x16"]


def get_ctor(name):
    """Drop-in class by reference name: Y models (CVSR_train) or RGB twins (mmedit fork)."""
    from fcvsr_amd.arch import CVSR_freq, fcvsr_rgb
    return getattr(CVSR_freq, name) if hasattr(CVSR_freq, name) else getattr(fcvsr_rgb, name)


def load_schema():
    with open(os.path.join(GOLDEN_DIR, "schema.json")) as f:
        return json.load(f)


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    taps = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("tap:")}
    return torch.from_numpy(z["x"]), taps, meta


def shapes_for(meta):
    """state_dict key->shape for a golden case (from the committed schema, scaled for reduced configs)."""
    from fcvsr_amd.arch.schema import state_dict_shapes
    kw = dict(meta["kwargs"])
    return state_dict_shapes(meta["ctor"], **kw)


def weights_for(meta):
    return synthetic_state_dict(shapes_for(meta), gain=meta["gain"])
