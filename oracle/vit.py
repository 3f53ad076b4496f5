"""Oracle (test infrastructure): functional PyTorch-CPU restatement of the reference's `vit_model.py`
pair encoders -- `VisionTransformer` (:188-317) and `ScaleEmbedTransformer` (:321-549) -- over a flat
{state_dict key -> tensor} mapping, fp32 on CPU.

Reference locations:
  PatchEmbed   vit_model.py:43-68      FeatureEmbed  :70-92     Attention  :95-135 (scale AFTER q@k^T, no bias table)
  Mlp          :138-157                 Block         :160-185   LayerNorm eps = 1e-6 (:218, :355)
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

from .s2former import feature_embed, mlp

Params = Dict[str, torch.Tensor]
EPS = 1e-6


@dataclass(frozen=True)
class VitConfig:
    img_size: int = 224
    patch: int = 16
    in_c: int = 3
    dim: int = 768
    depth: int = 12
    heads: int = 12
    hidden: int = 3072
    num_classes: int = 100
    representation_size: int = 0          # 0 -> has_logits False (pre_logits = Identity)

    @property
    def n_patches(self) -> int:
        return (self.img_size // self.patch) ** 2


def _block_spec(spec, pre, C, Hd):
    spec[pre + "norm1.weight"] = ((C,), "float32"); spec[pre + "norm1.bias"] = ((C,), "float32")
    spec[pre + "attn.qkv.weight"] = ((3 * C, C), "float32"); spec[pre + "attn.qkv.bias"] = ((3 * C,), "float32")
    spec[pre + "attn.proj.weight"] = ((C, C), "float32"); spec[pre + "attn.proj.bias"] = ((C,), "float32")
    spec[pre + "norm2.weight"] = ((C,), "float32"); spec[pre + "norm2.bias"] = ((C,), "float32")
    spec[pre + "mlp.fc1.weight"] = ((Hd, C), "float32"); spec[pre + "mlp.fc1.bias"] = ((Hd,), "float32")
    spec[pre + "mlp.fc2.weight"] = ((C, Hd), "float32"); spec[pre + "mlp.fc2.bias"] = ((C,), "float32")


def vit_param_spec(cfg: VitConfig, distilled: bool = False) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """state_dict manifest of VisionTransformer (vit_model.py:214-262), registration order; distilled: DeiT token + head_dist
    (:225, :250-253; no pre_logits then, :239)."""
    C = cfg.dim
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    spec["cls_token"] = ((1, 1, C), "float32")
    if distilled:
        spec["dist_token"] = ((1, 1, C), "float32")
    spec["pos_embed"] = ((1, cfg.n_patches + (2 if distilled else 1), C), "float32")
    spec["patch_embed.proj.weight"] = ((C, cfg.in_c, cfg.patch, cfg.patch), "float32")
    spec["patch_embed.proj.bias"] = ((C,), "float32")
    for j in range(cfg.depth):
        _block_spec(spec, f"blocks.{j}.", C, cfg.hidden)
    spec["norm.weight"] = ((C,), "float32"); spec["norm.bias"] = ((C,), "float32")
    feat = C
    if cfg.representation_size:
        spec["pre_logits.fc.weight"] = ((cfg.representation_size, C), "float32")
        spec["pre_logits.fc.bias"] = ((cfg.representation_size,), "float32")
        feat = cfg.representation_size
    if cfg.num_classes > 0:
        spec["head.weight"] = ((cfg.num_classes, feat), "float32"); spec["head.bias"] = ((cfg.num_classes,), "float32")
        if distilled:
            spec["head_dist.weight"] = ((cfg.num_classes, C), "float32"); spec["head_dist.bias"] = ((cfg.num_classes,), "float32")
    return spec


def attention(p: Params, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """vit_model.py:112-135: (q @ k^T) * scale, softmax, @ v, proj."""
    B, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, p[pre + "qkv.weight"], p.get(pre + "qkv.bias")).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    return F.linear((attn @ v).transpose(1, 2).reshape(B, N, C), p[pre + "proj.weight"], p[pre + "proj.bias"])


def block(p: Params, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """vit_model.py:182-185."""
    C = x.shape[-1]
    x = x + attention(p, pre + "attn.", F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], EPS), heads)
    return x + mlp(p, pre + "mlp.", F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], EPS))


def _patch_tokens(p: Params, pre: str, x: torch.Tensor, patch: int) -> torch.Tensor:
    return F.conv2d(x, p[pre + "proj.weight"], p[pre + "proj.bias"], stride=patch).flatten(2).transpose(1, 2)


def vit_forward_once(p: Params, x: torch.Tensor, cfg: VitConfig) -> torch.Tensor:
    """VisionTransformer.forward_once (:264-306) without distillation."""
    C = cfg.dim
    t = _patch_tokens(p, "patch_embed.", x, cfg.patch)
    t = torch.cat((p["cls_token"].expand(t.shape[0], -1, -1), t), dim=1) + p["pos_embed"]
    for j in range(cfg.depth):
        t = block(p, f"blocks.{j}.", t, cfg.heads)
    t = F.layer_norm(t, (C,), p["norm.weight"], p["norm.bias"], EPS)[:, 0]
    if cfg.representation_size:
        t = torch.tanh(F.linear(t, p["pre_logits.fc.weight"], p["pre_logits.fc.bias"]))
    if cfg.num_classes > 0:
        t = F.linear(t, p["head.weight"], p["head.bias"])
    return t


def vit_forward_once_distilled(p: Params, x: torch.Tensor, cfg: VitConfig, training: bool = True):
    """VisionTransformer.forward_once with distilled=True (:264-291): (head(x[:,0]), head_dist(x[:,1])) in training mode, their
    average otherwise."""
    C = cfg.dim
    t = _patch_tokens(p, "patch_embed.", x, cfg.patch)
    t = torch.cat((p["cls_token"].expand(t.shape[0], -1, -1), p["dist_token"].expand(t.shape[0], -1, -1), t), dim=1) + p["pos_embed"]
    for j in range(cfg.depth):
        t = block(p, f"blocks.{j}.", t, cfg.heads)
    t = F.layer_norm(t, (C,), p["norm.weight"], p["norm.bias"], EPS)
    a = F.linear(t[:, 0], p["head.weight"], p["head.bias"])
    b = F.linear(t[:, 1], p["head_dist.weight"], p["head_dist.bias"])
    return (a, b) if training else (a + b) / 2


def vit_forward_pair(p: Params, x1: torch.Tensor, x2: torch.Tensor, cfg: VitConfig):
    """forward(x1, x2) -> forward_twice (:296-299)."""
    return vit_forward_once(p, x1, cfg), vit_forward_once(p, x2, cfg)


# ---- ScaleEmbedTransformer (multi-scale + designed-feature token), vit_model.py:321-549 -------------
SCALE_EMBEDS = ((28, 4), (56, 8), (112, 16), (224, 32))       # (img_size, patch) of patch_embed0..3 (:353-356)


def scale_param_spec(depth: int = 12, dim: int = 768, hidden: int = 3072, in_c: int = 3, num_classes: int = 512,
                     representation_size: int = 0, label: bool = False) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """state_dict manifest for is_multiscale_embed=True, is_feature_embed=True; label = is_label_embed (:369-371, :408-432)."""
    C = dim
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    spec["cls_token"] = ((1, 1, C), "float32")
    if label:
        spec["label_token"] = ((1, 1, C), "float32")
    for i in range(4):
        spec[f"pos_embed{i}"] = ((1, 49, C), "float32")
    spec["pos_embed_non_multiscale"] = ((1, 196, C), "float32")
    for i, (_, ps) in enumerate(SCALE_EMBEDS):
        spec[f"patch_embed{i}.proj.weight"] = ((C, in_c, ps, ps), "float32")
        spec[f"patch_embed{i}.proj.bias"] = ((C,), "float32")
    spec["feature_embed.proj0.weight"] = ((C, 19, 1), "float32"); spec["feature_embed.proj0.bias"] = ((C,), "float32")
    for j in (1, 2):
        spec[f"feature_embed.proj{j}.weight"] = ((C, C, 1), "float32"); spec[f"feature_embed.proj{j}.bias"] = ((C,), "float32")
    for j in range(depth):
        _block_spec(spec, f"blocks.{j}.", C, hidden)
    spec["norm.weight"] = ((C,), "float32"); spec["norm.bias"] = ((C,), "float32")
    feat = C
    if representation_size:
        spec["pre_logits.fc.weight"] = ((representation_size, C), "float32")
        spec["pre_logits.fc.bias"] = ((representation_size,), "float32")
        feat = representation_size
    if label:
        spec["class_logits.weight"] = ((11, 100), "float32"); spec["class_logits.bias"] = ((11,), "float32")
    spec["head.weight"] = ((num_classes, feat), "float32"); spec["head.bias"] = ((num_classes,), "float32")
    spec["my_head.weight"] = ((100, 768), "float32"); spec["my_head.bias"] = ((100,), "float32")
    if label:
        spec["my_class_head.0.weight"] = ((100, C), "float32"); spec["my_class_head.0.bias"] = ((100,), "float32")
        spec["my_class_head.3.weight"] = ((100, 100), "float32"); spec["my_class_head.3.bias"] = ((100,), "float32")
    return spec


def scale_forward_features(p: Params, patches: Sequence[torch.Tensor], designed: torch.Tensor, depth: int = 12, heads: int = 12,
                           scales=(1, 1, 1, 1), representation_size: int = 0, label: bool = False):
    """forward_features (:448-511): 4 scaled patch embeds + cls + designed-feature token -> blocks -> norm -> my_head(x[:,0]).
    label (is_label_embed, :480-483, :503-506): the sequence becomes cls, label, designed, cls, designed, patches and the
    result (embedding, class_logits(x_class), x_class) with x_class = my_class_head(x[:,1]); Dropout(0.3) is the identity here
    (the fixture sets p = 0 on the reference instance: its RNG stream is not reproducible)."""
    C = p["cls_token"].shape[-1]
    xs = [(_patch_tokens(p, f"patch_embed{i}.", patches[i], ps) + p[f"pos_embed{i}"]) * scales[i]
          for i, (_, ps) in enumerate(SCALE_EMBEDS)]
    x = torch.cat(xs, 1)
    f = feature_embed(p, "feature_embed.", designed)
    x = torch.cat((p["cls_token"].expand(x.shape[0], -1, -1), f, x), dim=1)
    if label:
        x = torch.cat((p["cls_token"].expand(x.shape[0], -1, -1), p["label_token"].expand(x.shape[0], -1, -1), f, x), dim=1)
    for j in range(depth):
        x = block(p, f"blocks.{j}.", x, heads)
    x = F.layer_norm(x, (C,), p["norm.weight"], p["norm.bias"], EPS)
    y = F.linear(x[:, 0], p["my_head.weight"], p["my_head.bias"])
    if representation_size:
        y = torch.tanh(F.linear(y, p["pre_logits.fc.weight"], p["pre_logits.fc.bias"]))
    if label:
        c = F.gelu(F.linear(x[:, 1], p["my_class_head.0.weight"], p["my_class_head.0.bias"]))
        c = F.linear(c, p["my_class_head.3.weight"], p["my_class_head.3.bias"])
        return y, F.linear(c, p["class_logits.weight"], p["class_logits.bias"]), c
    return y


def scale_forward_pair(p: Params, x1, f1, x2, f2, **kw):
    """forward(x1, f1, x2, f2) -> forward_forice (:534-537)."""
    return scale_forward_features(p, x1, f1, **kw), scale_forward_features(p, x2, f2, **kw)
