"""Shared helpers for the tests: deterministic weights/inputs and fixture loading."""
import os
import warnings

import numpy as np
import torch

import recipe
from oracle import s2former as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fx(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tin(name, shape, kind="normal"):
    return torch.from_numpy(recipe.det_input(name, shape, kind))


def det_params(spec_items, prefix="", requires_grad=True, index_for=None):
    """{key: tensor} from the recipe.  `spec_items`: iterable of (key, (shape, dtype));
    int64 entries (relative_position_index) come from `index_for(key)`."""
    p = {}
    for k, (shape, dt) in spec_items:
        if dt == "float32":
            t = torch.from_numpy(recipe.det_weight(prefix + k, shape))
            if requires_grad:
                t.requires_grad_(True)
            p[k] = t
        else:
            p[k] = torch.from_numpy(index_for(k))
    return p


def model_params(cfg, requires_grad=True):
    def index_for(k):
        stage = int(k[len("blocks")])
        return O.relpos_index(cfg.cube(stage))
    return det_params(O.param_spec(cfg).items(), "", requires_grad, index_for)


def model_inputs(tag, scales, in_c, B=4):
    """Same construction as tests/golden/make_golden.py:model_inputs."""
    left = [tin(f"{tag}.left{i}", (B, in_c, s, s), "unit") for i, s in enumerate(scales)]
    right = [tin(f"{tag}.right{i}", (B, in_c, s, s), "unit") for i, s in enumerate(scales)]
    ld = tin(tag + ".ld", (B, 1, 19), "designed")
    rd = tin(tag + ".rd", (B, 1, 19), "designed")
    eps = 0.2
    for i in range(len(scales)):
        for b in (1, 2):
            right[i][b] = left[i][b] * (1 - eps) + eps * right[i][b]
    for b in (1, 2):
        rd[b] = ld[b] * (1 + eps)
    flag = torch.tensor([1, 0, 1, 0], dtype=torch.int64)
    return left, ld, right, rd, flag


MODEL_CASES = {
    "v3_3s3c_642": O.S2Config(scales=(32, 64, 128), in_c=3, depth=(6, 4, 2)),
    "v3_4s4c_321": O.S2Config(scales=(32, 64, 128, 256), in_c=4, depth=(3, 2, 1)),
    "v3_3s3c_111": O.S2Config(scales=(32, 64, 128), in_c=3, depth=(1, 1, 1)),
    "v3_4s4c_642": O.S2Config(scales=(32, 64, 128, 256), in_c=4, depth=(6, 4, 2)),
}
