"""Oracle (test infrastructure): functional PyTorch-CPU restatement of the reference's
multi-scale Siamese transformer `ShfitScaleFormer_v3` and its building blocks.

Written as pure functions over a flat {state_dict key -> tensor} mapping so that the same
deterministic weights (tests/golden/recipe.py) can be fed to the reference modules, to this
oracle and to the HIP path.  All arithmetic is fp32 on CPU.

Reference locations (relative to the reference tree):
  PatchEmbed               nets/ShfitScaleFormer.py:12-37
  Mlp                      nets/ShfitScaleFormer.py:39-58
  FeatureEmbed             nets/ShfitScaleFormer.py:60-82
  CrossScaleAttention      nets/ShfitScaleFormer.py:84-156
  CrossScaleBlock          nets/ShfitScaleFormer.py:158-184
  ShfitScaleFormer_v3      nets/ShfitScaleFormer.py:772-1010
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


@dataclass(frozen=True)
class S2Config:
    """Static shape description of a v3 encoder (ctor args of nets/ShfitScaleFormer.py:773-789)."""
    scales: Tuple[int, ...] = (32, 64, 128)   # input_image_scales
    in_c: int = 3                              # hard-coded 3 upstream (:809); parametrised here
    depth: Tuple[int, int, int] = (6, 4, 2)
    grid: int = 8                              # cube_size=[8,8]: tokens per side at stage 0
    dim: int = 768
    heads: int = 12
    hidden: int = 3072                         # mlp_ratio 4.0
    designed: bool = True                      # is_designed_feature_embedding
    n_designed: int = 19
    out_dim: int = 100
    num_classes: int = 11
    ln_eps: float = 1e-5                       # nn.LayerNorm default (:787)

    @property
    def n_scales(self) -> int:
        return len(self.scales)

    def cube(self, stage: int) -> Tuple[int, int, int]:
        g = self.grid >> stage
        return (self.n_scales, g, g)

    def tokens(self, stage: int) -> int:
        s, h, w = self.cube(stage)
        return s * h * w

    def patch_size(self, i: int) -> int:
        return int(self.scales[i] / self.grid)   # :809  int(scale / cube_size[1])


# ----------------------------------------------------------------------------------------
# relative position index (nets/ShfitScaleFormer.py:139-156)
# ----------------------------------------------------------------------------------------
def relpos_index(cube: Sequence[int]) -> np.ndarray:
    """int64 [N,N] index into the bias table for a (scales, rows, cols) token cube.

    Closed form of the meshgrid/broadcast construction at :141-155: tokens are ordered
    scale-major then row-major, and
        idx[i,j] = (zi-zj+S-1)*(2H-1)*(2W-1) + (yi-yj+H-1)*(2W-1) + (xi-xj+W-1).
    """
    S, H, W = (int(c) for c in cube)
    z, y, x = np.meshgrid(np.arange(S), np.arange(H), np.arange(W), indexing="ij")
    z, y, x = z.reshape(-1), y.reshape(-1), x.reshape(-1)
    dz = z[:, None] - z[None, :] + (S - 1)
    dy = y[:, None] - y[None, :] + (H - 1)
    dx = x[:, None] - x[None, :] + (W - 1)
    return (dz * (2 * H - 1) * (2 * W - 1) + dy * (2 * W - 1) + dx).astype(np.int64)


def relpos_table_rows(cube: Sequence[int]) -> int:
    S, H, W = (int(c) for c in cube)
    return (2 * S - 1) * (2 * H - 1) * (2 * W - 1)   # :104-106


# ----------------------------------------------------------------------------------------
# parameter manifest (state_dict keys/shapes; SURVEY 8b, nets/ShfitScaleFormer.py:807-866)
# ----------------------------------------------------------------------------------------
def param_spec(cfg: S2Config) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """Ordered {key: (shape, dtype)} exactly as the reference's state_dict() lists them."""
    C, Hd = cfg.dim, cfg.hidden
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for i in range(cfg.n_scales):
        p = cfg.patch_size(i)
        spec[f"patch_embed_blocks.{i}.proj.weight"] = ((C, cfg.in_c, p, p), "float32")
        spec[f"patch_embed_blocks.{i}.proj.bias"] = ((C,), "float32")
    if cfg.designed:
        spec["feature_embed.proj0.weight"] = ((C, cfg.n_designed, 1), "float32")
        spec["feature_embed.proj0.bias"] = ((C,), "float32")
        for j in (1, 2):
            spec[f"feature_embed.proj{j}.weight"] = ((C, C, 1), "float32")
            spec[f"feature_embed.proj{j}.bias"] = ((C,), "float32")
    for stage in range(3):
        cube = cfg.cube(stage)
        n = cfg.tokens(stage)
        for j in range(cfg.depth[stage]):
            pre = f"blocks{stage}.{j}."
            spec[pre + "norm1.weight"] = ((C,), "float32")
            spec[pre + "norm1.bias"] = ((C,), "float32")
            spec[pre + "attn.relative_position_bias_table"] = ((relpos_table_rows(cube), cfg.heads), "float32")
            spec[pre + "attn.relative_position_index"] = ((n, n), "int64")
            spec[pre + "attn.qkv.weight"] = ((3 * C, C), "float32")
            spec[pre + "attn.qkv.bias"] = ((3 * C,), "float32")
            spec[pre + "attn.proj.weight"] = ((C, C), "float32")
            spec[pre + "attn.proj.bias"] = ((C,), "float32")
            spec[pre + "norm2.weight"] = ((C,), "float32")
            spec[pre + "norm2.bias"] = ((C,), "float32")
            spec[pre + "mlp.fc1.weight"] = ((Hd, C), "float32")
            spec[pre + "mlp.fc1.bias"] = ((Hd,), "float32")
            spec[pre + "mlp.fc2.weight"] = ((C, Hd), "float32")
            spec[pre + "mlp.fc2.bias"] = ((C,), "float32")
    spec["norm.weight"] = ((C,), "float32")
    spec["norm.bias"] = ((C,), "float32")
    spec["final_features.weight"] = ((cfg.out_dim, cfg.n_scales * C), "float32")
    spec["final_features.bias"] = ((cfg.out_dim,), "float32")
    spec["final_features_with_design.weight"] = ((cfg.out_dim, (cfg.n_scales + 1) * C), "float32")
    spec["final_features_with_design.bias"] = ((cfg.out_dim,), "float32")
    spec["head.weight"] = ((cfg.num_classes, cfg.out_dim), "float32")
    spec["head.bias"] = ((cfg.num_classes,), "float32")
    return spec


def model_name(cfg: S2Config) -> str:
    """`net.name` string (nets/ShfitScaleFormer.py:791-795)."""
    name = "S2Former_v3-3CH"
    if cfg.designed:
        name += "-3DP-SEF"
    return f"{name}-{cfg.depth[0]}{cfg.depth[1]}{cfg.depth[2]}"


# ----------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------
def patch_embed(p: Params, pre: str, x: torch.Tensor, patch: int) -> torch.Tensor:
    """Conv2d(k=s=patch) -> [B, HW, C]  (:28-37).  Asserts the image size like :30-31."""
    w = p[pre + "proj.weight"]
    B, C, H, W = x.shape
    assert H == W and H % patch == 0, f"Input image size ({H}*{W}) doesn't match model"
    y = F.conv2d(x, w, p[pre + "proj.bias"], stride=patch)
    return y.flatten(2).transpose(1, 2)


def mlp(p: Params, pre: str, x: torch.Tensor) -> torch.Tensor:
    """fc2(GELU_erf(fc1(x)))  (:52-58; dropout p=0)."""
    h = F.gelu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"]))
    return F.linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def feature_embed(p: Params, pre: str, d: torch.Tensor) -> torch.Tensor:
    """[B,1,F] -> [B,1,C]: three k=1 Conv1d, GELU only after the first (:69-82)."""
    x = d.reshape(d.shape[0], -1)
    x = F.gelu(F.linear(x, p[pre + "proj0.weight"].squeeze(-1), p[pre + "proj0.bias"]))
    x = F.linear(x, p[pre + "proj1.weight"].squeeze(-1), p[pre + "proj1.bias"])
    x = F.linear(x, p[pre + "proj2.weight"].squeeze(-1), p[pre + "proj2.bias"])
    return x.unsqueeze(1)


def relpos_bias(p: Params, pre: str, n: int) -> torch.Tensor:
    """Dense [heads, N, N] bias = table[index.view(-1)].view(N,N,h).permute(2,0,1)  (:123-128)."""
    table = p[pre + "relative_position_bias_table"]
    index = p[pre + "relative_position_index"]
    return table[index.reshape(-1)].reshape(n, n, -1).permute(2, 0, 1).contiguous()


def cross_scale_attention(p: Params, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """Global attention over the token cube with 3-D relative-position bias (:113-136)."""
    B, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (d ** -0.5)                      # scale applied to q BEFORE q @ k^T (:121)
    attn = q @ k.transpose(-2, -1)
    attn = attn + relpos_bias(p, pre, N).unsqueeze(0)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(out, p[pre + "proj.weight"], p[pre + "proj.bias"])


def cross_scale_block(p: Params, pre: str, x: torch.Tensor, heads: int, eps: float) -> torch.Tensor:
    """Pre-norm transformer block (:181-184; DropPath ratio 0 -> identity)."""
    C = x.shape[-1]
    x = x + cross_scale_attention(
        p, pre + "attn.", F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps), heads)
    x = x + mlp(p, pre + "mlp.", F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps))
    return x


def token_pool2x2(x: torch.Tensor, n_scales: int, side: int) -> torch.Tensor:
    """Per scale: [B, side*side, C] -> AvgPool2d(2,2) over the token grid -> [B, (side/2)^2, C]
    (:892-901 and :905-914)."""
    B, N, C = x.shape
    x = x.reshape(B, n_scales, side // 2, 2, side // 2, 2, C)
    return x.mean(dim=(3, 5)).reshape(B, n_scales * (side // 2) ** 2, C)


# ----------------------------------------------------------------------------------------
# whole encoder
# ----------------------------------------------------------------------------------------
def embed_tokens(p: Params, patches: Sequence[torch.Tensor], cfg: S2Config) -> torch.Tensor:
    """v3.patch_embed (:869-882): per-scale embed, concatenated along tokens."""
    ys = [patch_embed(p, f"patch_embed_blocks.{i}.", patches[i], cfg.patch_size(i)) for i in range(cfg.n_scales)]
    return torch.cat(ys, dim=1)


def backbone(p: Params, x: torch.Tensor, cfg: S2Config) -> torch.Tensor:
    """v3.backbone (:888-919): blocks0 -> pool+norm -> blocks1 -> pool+norm -> blocks2."""
    C = cfg.dim
    for stage in range(3):
        for j in range(cfg.depth[stage]):
            x = cross_scale_block(p, f"blocks{stage}.{j}.", x, cfg.heads, cfg.ln_eps)
        if stage < 2:
            x = token_pool2x2(x, cfg.n_scales, cfg.grid >> stage)
            x = F.layer_norm(x, (C,), p["norm.weight"], p["norm.bias"], cfg.ln_eps)   # shared norm (:902, :915)
    return x


def forward_once(p: Params, patches: Sequence[torch.Tensor], designed: torch.Tensor | None,
                 cfg: S2Config) -> torch.Tensor:
    """forward_once_design_feature (:922-950) / forward_once (:954-968) -> [B, 100]."""
    C = cfg.dim
    x = backbone(p, embed_tokens(p, patches, cfg), cfg)
    x = F.layer_norm(x, (C,), p["norm.weight"], p["norm.bias"], cfg.ln_eps)           # :926
    B = x.shape[0]
    x = x.reshape(B, cfg.n_scales, -1, C).mean(dim=2).reshape(B, cfg.n_scales * C)    # :930-938
    if cfg.designed:
        f = feature_embed(p, "feature_embed.", designed).squeeze(1)                   # :939-940
        f = F.layer_norm(f, (C,), p["norm.weight"], p["norm.bias"], cfg.ln_eps)       # :941
        x = torch.cat((x, f), dim=1)                                                  # :945
        return F.linear(x, p["final_features_with_design.weight"], p["final_features_with_design.bias"])
    return F.linear(x, p["final_features.weight"], p["final_features.bias"])


def forward_pair(p: Params, left: Sequence[torch.Tensor], left_designed, right: Sequence[torch.Tensor],
                 right_designed, cfg: S2Config):
    """Training-mode forward (:980-991): two independent passes, returns (f1, f2)."""
    return forward_once(p, left, left_designed, cfg), forward_once(p, right, right_designed, cfg)


def flops_forward_per_sample(cfg: S2Config) -> float:
    """Algorithmic forward FLOPs per encoder sample (SURVEY 8d / BASELINE.md section 3)."""
    C = cfg.dim
    f = 0.0
    for i in range(cfg.n_scales):
        f += 2.0 * cfg.grid * cfg.grid * (cfg.in_c * cfg.patch_size(i) ** 2) * C
    for s in range(3):
        n = cfg.tokens(s)
        f += cfg.depth[s] * (24.0 * n * C * C + 4.0 * n * n * C)
    f += 2.0 * (cfg.n_designed * C + 2 * C * C)
    f += 2.0 * (cfg.n_scales + 1) * C * cfg.out_dim
    return f


# ----------------------------------------------------------------------------------------
# single-stage variants v1 / v2 (nets/ShfitScaleFormer.py:417-607, :610-769) and v6 (:1506-1569)
# ----------------------------------------------------------------------------------------
V12_SCALES = (28, 56, 112, 224)      # cube [4,7,7]: 4 scales x 49 tokens = 196, patch sizes 4 / 8 / 16 / 32


def v12_param_spec(variant: str, depth: int = 12, num_classes: int = 11) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """state_dict manifest of `ShfitScaleFormer` ("v1") / `ShfitScaleFormer_v2` ("v2") in registration order.
    v2 ignores its `depth` argument upstream and always builds 12 blocks (:657)."""
    C, Hd, S = 768, 3072, 4
    cube = (S, 7, 7)
    if variant == "v2":
        depth = 12
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for i, s in enumerate(V12_SCALES):
        p = s // 7
        pre = f"patch_embed_scale{i}." if variant == "v1" else f"patch_embed_blocks.{i}."
        spec[pre + "proj.weight"] = ((C, 3, p, p), "float32"); spec[pre + "proj.bias"] = ((C,), "float32")
    spec["feature_embed.proj0.weight"] = ((C, 19, 1), "float32"); spec["feature_embed.proj0.bias"] = ((C,), "float32")
    for j in (1, 2):
        spec[f"feature_embed.proj{j}.weight"] = ((C, C, 1), "float32"); spec[f"feature_embed.proj{j}.bias"] = ((C,), "float32")
    for j in range(depth):
        pre = f"blocks.{j}."
        spec[pre + "norm1.weight"] = ((C,), "float32"); spec[pre + "norm1.bias"] = ((C,), "float32")
        spec[pre + "attn.relative_position_bias_table"] = ((relpos_table_rows(cube), 12), "float32")
        spec[pre + "attn.relative_position_index"] = ((196, 196), "int64")
        spec[pre + "attn.qkv.weight"] = ((3 * C, C), "float32"); spec[pre + "attn.qkv.bias"] = ((3 * C,), "float32")
        spec[pre + "attn.proj.weight"] = ((C, C), "float32"); spec[pre + "attn.proj.bias"] = ((C,), "float32")
        spec[pre + "norm2.weight"] = ((C,), "float32"); spec[pre + "norm2.bias"] = ((C,), "float32")
        spec[pre + "mlp.fc1.weight"] = ((Hd, C), "float32"); spec[pre + "mlp.fc1.bias"] = ((Hd,), "float32")
        spec[pre + "mlp.fc2.weight"] = ((C, Hd), "float32"); spec[pre + "mlp.fc2.bias"] = ((C,), "float32")
    spec["norm.weight"] = ((C,), "float32"); spec["norm.bias"] = ((C,), "float32")
    spec["final_features.weight"] = ((100, S * C), "float32"); spec["final_features.bias"] = ((100,), "float32")
    spec["final_features_with_design.weight"] = ((100, (S + 1) * C), "float32")
    spec["final_features_with_design.bias"] = ((100,), "float32")
    spec["head.weight"] = ((num_classes, 100), "float32"); spec["head.bias"] = ((num_classes,), "float32")
    return spec


def v12_forward_once(p: Params, variant: str, patches: Sequence[torch.Tensor], designed, depth: int = 12) -> torch.Tensor:
    """forward_once_design_feature / forward_once of v1 (:505-569) and v2 (:689-728): 4 patch embeds -> `depth`
    blocks on the [4,7,7] cube -> norm -> per-scale mean over 49 tokens -> (+ normed designed embedding) -> Linear."""
    C, S = 768, 4
    if variant == "v2":
        depth = 12
    toks = []
    for i, s in enumerate(V12_SCALES):
        pre = f"patch_embed_scale{i}." if variant == "v1" else f"patch_embed_blocks.{i}."
        toks.append(patch_embed(p, pre, patches[i], s // 7))
    x = torch.cat(toks, 1)
    for j in range(depth):
        x = cross_scale_block(p, f"blocks.{j}.", x, 12, 1e-5)
    x = F.layer_norm(x, (C,), p["norm.weight"], p["norm.bias"], 1e-5)
    B = x.shape[0]
    x = x.reshape(B, S, 49, C).mean(dim=2).reshape(B, S * C)
    if designed is None:
        return F.linear(x, p["final_features.weight"], p["final_features.bias"])
    f = feature_embed(p, "feature_embed.", designed).squeeze(1)
    f = F.layer_norm(f, (C,), p["norm.weight"], p["norm.bias"], 1e-5)
    return F.linear(torch.cat((x, f), 1), p["final_features_with_design.weight"], p["final_features_with_design.bias"])


def v6_param_spec(num_classes: int = 11) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    C = 768
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    spec["feature_embed.proj0.weight"] = ((C, 19, 1), "float32"); spec["feature_embed.proj0.bias"] = ((C,), "float32")
    for j in (1, 2):
        spec[f"feature_embed.proj{j}.weight"] = ((C, C, 1), "float32"); spec[f"feature_embed.proj{j}.bias"] = ((C,), "float32")
    spec["norm.weight"] = ((C,), "float32"); spec["norm.bias"] = ((C,), "float32")
    spec["final_features_with_design.weight"] = ((100, C), "float32"); spec["final_features_with_design.bias"] = ((100,), "float32")
    spec["head.weight"] = ((num_classes, 100), "float32"); spec["head.bias"] = ((num_classes,), "float32")
    return spec


def v6_forward_once(p: Params, designed: torch.Tensor) -> torch.Tensor:
    """ShfitScaleFormer_v6 (:1529-1537): FeatureEmbed -> squeeze -> norm -> Linear(768, 100)."""
    f = feature_embed(p, "feature_embed.", designed).squeeze(1)
    f = F.layer_norm(f, (768,), p["norm.weight"], p["norm.bias"], 1e-5)
    return F.linear(f, p["final_features_with_design.weight"], p["final_features_with_design.bias"])
