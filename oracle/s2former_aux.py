"""Oracle (test infrastructure): functional PyTorch-CPU restatement of the auxiliary-head variants of the
multi-scale encoder -- `ShfitScaleFormer_v4` (nets/ShfitScaleFormer.py:1013-1261, `AuxBolck` :329-368) and
`ShfitScaleFormer_v5` (:1264-1503, `CrossScaleAttention_v5` :187-296, `AuxBolck_v5` :370-415).

Both are fixed to three input scales and 3 channels upstream (the aux heads hard-code cube [3,8,8] / [3,4,4]
and the 4x4 / 2x2 slices).  `Dropout2d(p=0.3)` in the aux heads draws from torch's RNG stream, which no
independent implementation reproduces: this oracle (and the golden vectors, which set that module's p to 0 on
the reference instance) pin everything else -- conv, BatchNorm2d batch / running statistics, ReLU, 1x1 conv,
pooling, LayerNorm, output Linear.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .s2former import (Params, S2Config, cross_scale_block, embed_tokens, feature_embed, param_spec, relpos_index,
                       relpos_table_rows, token_pool2x2)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ---- v5 relative position index (:217-263) ----------------------------------------------------------
def relpos_index_v5(cube: Sequence[int]) -> np.ndarray:
    """int64 [N+1, N+1]: the cube index, one extra column (ids base..base+N-1), one extra row
    (ids base+N .. base+2N), and the corner [-1,-1] aliased to [0,0] (:241-261)."""
    base = relpos_index(cube)
    n = base.shape[0]
    bins = relpos_table_rows(cube)                     # max id + 1
    col = bins + np.arange(n, dtype=np.int64)[:, None]
    idx = np.concatenate([base, col], axis=1)
    row = (bins + n) + np.arange(n + 1, dtype=np.int64)[None, :]
    idx = np.concatenate([idx, row], axis=0)
    idx[-1, -1] = idx[0, 0]
    return idx


def relpos_table_rows_v5(cube: Sequence[int]) -> int:
    s, h, w = (int(c) for c in cube)
    return relpos_table_rows(cube) + 2 * s * h * w     # :205-209


# ---- parameter manifests -----------------------------------------------------------------------------
def _aux_spec(spec, pre: str, scales: int, C: int, out_dim: int, v5: bool):
    spec[pre + "aux.0.weight"] = ((C, C, 2, 2), "float32")
    spec[pre + "aux.1.weight"] = ((C,), "float32")
    spec[pre + "aux.1.bias"] = ((C,), "float32")
    spec[pre + "aux.1.running_mean"] = ((C,), "float32")
    spec[pre + "aux.1.running_var"] = ((C,), "float32")
    spec[pre + "aux.1.num_batches_tracked"] = ((), "int64")
    spec[pre + "aux.4.weight"] = ((C // scales, C, 1, 1), "float32")
    spec[pre + "aux.4.bias"] = ((C // scales,), "float32")
    width = 2 * C if v5 else C
    spec[pre + "norm.weight"] = ((width,), "float32")
    spec[pre + "norm.bias"] = ((width,), "float32")
    spec[pre + "out_features.weight"] = ((out_dim, width), "float32")
    spec[pre + "out_features.bias"] = ((out_dim,), "float32")


def v4_config(depth=(3, 2, 1), designed: bool = True) -> S2Config:
    return S2Config(scales=(32, 64, 128), in_c=3, depth=tuple(depth), designed=designed)


def v4_param_spec(cfg: S2Config) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """v3's manifest followed by aux0.*, aux1.* (:1097-1098)."""
    spec = param_spec(cfg)
    _aux_spec(spec, "aux0.", 3, cfg.dim, cfg.out_dim, False)
    _aux_spec(spec, "aux1.", 3, cfg.dim, cfg.out_dim, False)
    return spec


def v4_model_name(cfg: S2Config) -> str:
    """:1033-1037."""
    name = "S2Former_v4-3CH" + ("-SFE" if cfg.designed else "")
    return f"{name}-{cfg.depth[0]}{cfg.depth[1]}{cfg.depth[2]}"


def v5_param_spec(cfg: S2Config) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """v5 always embeds the designed features (:1296); block tables / indices are the extended ones."""
    C = cfg.dim
    base = param_spec(cfg)
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for k, v in base.items():
        if k.endswith("attn.relative_position_bias_table"):
            stage = int(k[len("blocks")])
            spec[k] = ((relpos_table_rows_v5(cfg.cube(stage)), cfg.heads), "float32")
        elif k.endswith("attn.relative_position_index"):
            n = cfg.tokens(int(k[len("blocks")])) + 1
            spec[k] = ((n, n), "int64")
        elif k == "final_features_with_design.weight":
            spec[k] = ((cfg.out_dim, 2 * C), "float32")                      # :1350
        elif k == "head.weight":
            spec["last_block_features.weight"] = ((C, (cfg.n_scales + 1) * C), "float32")   # :1351
            spec["last_block_features.bias"] = ((C,), "float32")
            spec[k] = v
        else:
            spec[k] = v
    _aux_spec(spec, "aux0.", 3, C, cfg.out_dim, True)
    _aux_spec(spec, "aux1.", 3, C, cfg.out_dim, True)
    return spec


# ---- aux heads ---------------------------------------------------------------------------------------
def batch_norm2d(p: Params, pre: str, x: torch.Tensor, training: bool, stats_out: dict | None = None) -> torch.Tensor:
    """nn.BatchNorm2d: training normalises with the batch mean / biased variance and moves the running
    statistics by momentum 0.1 with the UNBIASED variance; eval uses the running statistics."""
    w, b = p[pre + "weight"], p[pre + "bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            # the head runs once per scale through the SAME BatchNorm module (:354-361): the running statistics move
            # once per call, so chain from the latest value
            n = x.numel() // x.shape[1]
            rm = stats_out.get(pre + "running_mean", p[pre + "running_mean"])
            rv = stats_out.get(pre + "running_var", p[pre + "running_var"])
            stats_out[pre + "running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            stats_out[pre + "running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * n / (n - 1)
            stats_out[pre + "num_batches_tracked"] = stats_out.get(pre + "num_batches_tracked", p[pre + "num_batches_tracked"]) + 1
    else:
        mean, var = p[pre + "running_mean"], p[pre + "running_var"]
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def aux_block(p: Params, pre: str, x: torch.Tensor, cube: Sequence[int], training: bool, v5: bool,
              stats_out: dict | None = None) -> torch.Tensor:
    """AuxBolck.forward (:349-368) / AuxBolck_v5.forward (:391-415); Dropout2d taken as identity (header)."""
    S, side = int(cube[0]), int(cube[1])
    B, _, C = x.shape
    ys = []
    for i in range(S):
        t = x[:, side * side * i: side * side * (i + 1), :].transpose(1, 2).reshape(B, C, side, side)
        t = F.conv2d(t, p[pre + "aux.0.weight"])
        t = torch.relu(batch_norm2d(p, pre + "aux.1.", t, training, stats_out))
        t = F.conv2d(t, p[pre + "aux.4.weight"], p[pre + "aux.4.bias"])
        ys.append(t.flatten(2).mean(dim=2))
    y = torch.cat(ys, 1)
    if v5:
        last = x[:, side * side * S:, :].transpose(1, 2).flatten(1)          # the designed-feature token (:403-404)
        y = torch.cat([y, last], 1)                                           # self.norm is NOT applied (:412 commented)
    else:
        y = F.layer_norm(y, (y.shape[1],), p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-5)
    return F.linear(y, p[pre + "out_features.weight"], p[pre + "out_features.bias"])


# ---- v4 ----------------------------------------------------------------------------------------------
def v4_forward_once(p: Params, patches, designed, cfg: S2Config, training: bool, stats_out: dict | None = None):
    """forward_once_design_feature (:1186-1201) / forward_once (:1205-1215): (x, aux0, aux1); callers drop
    the aux outputs in eval mode."""
    C = cfg.dim
    ln = lambda t: F.layer_norm(t, (C,), p["norm.weight"], p["norm.bias"], cfg.ln_eps)
    x = embed_tokens(p, patches, cfg)
    for j in range(cfg.depth[0]):
        x = cross_scale_block(p, f"blocks0.{j}.", x, cfg.heads, cfg.ln_eps)
    aux0 = aux_block(p, "aux0.", x, (3, 8, 8), training, False, stats_out)
    x = ln(token_pool2x2(x, cfg.n_scales, cfg.grid))
    for j in range(cfg.depth[1]):
        x = cross_scale_block(p, f"blocks1.{j}.", x, cfg.heads, cfg.ln_eps)
    aux1 = aux_block(p, "aux1.", x, (3, 4, 4), training, False, stats_out)
    x = ln(token_pool2x2(x, cfg.n_scales, cfg.grid // 2))
    for j in range(cfg.depth[2]):
        x = cross_scale_block(p, f"blocks2.{j}.", x, cfg.heads, cfg.ln_eps)
    x = ln(x)
    B = x.shape[0]
    x = x.reshape(B, cfg.n_scales, -1, C).mean(dim=2).reshape(B, cfg.n_scales * C)
    if cfg.designed:
        f = ln(feature_embed(p, "feature_embed.", designed).squeeze(1))
        x = F.linear(torch.cat((x, f), 1), p["final_features_with_design.weight"], p["final_features_with_design.bias"])
    else:
        x = F.linear(x, p["final_features.weight"], p["final_features.bias"])
    return x, aux0, aux1


# ---- v5 ----------------------------------------------------------------------------------------------
def _pool_keep_last(x: torch.Tensor, n_scales: int, side: int) -> torch.Tensor:
    """2x2 pooling of the cube tokens, the extra token carried through (:1389-1400, :1406-1417)."""
    n = n_scales * side * side
    return torch.cat([token_pool2x2(x[:, :n], n_scales, side), x[:, n:]], 1)


def v5_forward_once(p: Params, patches, designed, cfg: S2Config, training: bool, stats_out: dict | None = None):
    """forward_once_design_feature (:1444-1462) -> (x, aux0, aux1)."""
    C = cfg.dim
    ln = lambda t: F.layer_norm(t, (C,), p["norm.weight"], p["norm.bias"], cfg.ln_eps)
    f = ln(feature_embed(p, "feature_embed.", designed).squeeze(1))           # designed_feature_embed (:1375-1379)
    x = torch.cat((embed_tokens(p, patches, cfg), f.unsqueeze(1)), 1)        # N + 1 tokens
    for j in range(cfg.depth[0]):
        x = cross_scale_block(p, f"blocks0.{j}.", x, cfg.heads, cfg.ln_eps)
    aux0 = aux_block(p, "aux0.", x, (3, 8, 8), training, True, stats_out)
    x = ln(_pool_keep_last(x, cfg.n_scales, cfg.grid))
    for j in range(cfg.depth[1]):
        x = cross_scale_block(p, f"blocks1.{j}.", x, cfg.heads, cfg.ln_eps)
    aux1 = aux_block(p, "aux1.", x, (3, 4, 4), training, True, stats_out)
    x = ln(_pool_keep_last(x, cfg.n_scales, cfg.grid // 2))
    for j in range(cfg.depth[2]):
        x = cross_scale_block(p, f"blocks2.{j}.", x, cfg.heads, cfg.ln_eps)
    x = ln(x)
    B = x.shape[0]
    n = cfg.n_scales * 4
    y = torch.cat([x[:, :n].reshape(B, cfg.n_scales, 4, C).mean(dim=2).reshape(B, -1), x[:, n:].mean(dim=1)], 1)
    y = F.linear(y, p["last_block_features.weight"], p["last_block_features.bias"])          # :1437
    y = F.linear(torch.cat((y, f), 1), p["final_features_with_design.weight"], p["final_features_with_design.bias"])
    return y, aux0, aux1
