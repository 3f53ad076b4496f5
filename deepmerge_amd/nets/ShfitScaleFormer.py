"""MI355X-native drop-in for the reference module `nets/ShfitScaleFormer.py` (spelling is API).

Same class names, constructor keywords, forward signatures, attributes (`name`, `depth`,
`input_image_scales`, `cube_size`) and state_dict keys/shapes/dtypes as the reference, so reference
checkpoints load strictly and `Train_SMT.py` / `ExtractFeatures.py` style callers work unchanged.
All arithmetic runs in libdeepmerge_hip (hand-written gfx950 kernels) through `deepmerge_amd.ops`;
the torch.nn.Linear / Conv / LayerNorm objects below are parameter containers only (they give the
reference's key names and default initialisation) and their own forward() is never called.

Additive keyword arguments (defaults keep reference behaviour): `in_c` (the reference hard-codes 3,
nets/ShfitScaleFormer.py:809) and `numerics` ("bf16" throughput mode / "fp32" parity mode; default
`deepmerge_amd.get_numerics()`).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from .. import ops

__all__ = ["PatchEmbed", "Mlp", "FeatureEmbed", "CrossScaleAttention", "CrossScaleBlock", "ShfitScaleFormer",
           "ShfitScaleFormer_v2", "ShfitScaleFormer_v3", "ShfitScaleFormer_v6"]


def drop_path(x, drop_prob: float = 0., training: bool = False):
    """Stochastic depth per sample, as the reference draws it (vit_model.py:12-28, used by nets/ShfitScaleFormer.py:170-183):
    mask = floor(keep + U[0, 1)) per sample, output = x / keep * mask.  Identity for ratio 0 or in eval mode."""
    if drop_prob == 0. or not training:
        return x
    keep = 1 - drop_prob
    mask = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
    mask.floor_()
    return x.div(keep) * mask


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


def stochastic_block_forward(block, x, table_args):
    """The pre-norm block with stochastic depth (training mode, ratio > 0): the two residual branches are separate autograd nodes so a
    per-sample mask can sit between a branch and its residual add; LayerNorm, the attention module and the Mlp are the same kernels
    the fused node uses.  x: [B, N, C] fp32."""
    x = x.float()
    dt = ops.act_dtype(block.numerics)
    y = ops.LayerNormFn.apply(x, block.norm1.weight, block.norm1.bias, block.norm1.eps, dt)
    x = x + block.drop_path(block.attn(y))
    y = ops.LayerNormFn.apply(x, block.norm2.weight, block.norm2.bias, block.norm2.eps, dt)
    return x + block.drop_path(block.mlp(y).float())


def _mode(numerics: Optional[str], module=None) -> str:
    """Validated numerics mode of a module; a "bf16x3" module gets its product scope bound to its calls (ops.bind_numerics)."""
    return ops.bind_numerics(module, numerics or ops.get_numerics())


def relative_position_index(cube: Sequence[int]) -> torch.Tensor:
    """int64 [N,N] bias-table index of a (scales, rows, cols) token cube, tokens scale-major then
    row-major: idx[i,j] = (zi-zj+S-1)(2H-1)(2W-1) + (yi-yj+H-1)(2W-1) + (xi-xj+W-1).
    Same values as the buffer built at nets/ShfitScaleFormer.py:139-156 (pinned by tests/golden)."""
    S, H, W = (int(c) for c in cube)
    t = torch.arange(S * H * W)
    z, y, x = t // (H * W), (t // W) % H, t % W
    d = lambda a, n: a[:, None] - a[None, :] + (n - 1)
    return (d(z, S) * ((2 * H - 1) * (2 * W - 1)) + d(y, H) * (2 * W - 1) + d(x, W)).to(torch.int64)


class PatchEmbed(nn.Module):
    """2-D image to patch tokens: Conv2d(k = stride = patch) -> [B, HW/p^2, out_c]
    (reference :12-37).  Runs as patch extraction (dm_patchify) + one MFMA GEMM."""

    def __init__(self, img_size=224, patch_size=16, in_c=3, out_c=768, norm_layer=None, numerics=None):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_c, out_c, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(out_c) if norm_layer else nn.Identity()
        self.numerics = _mode(numerics, self)

    def operand_rows(self, x):
        """The patch-embed GEMM's A rows [B * num_patches, in_c * p * p] of an image batch (or of pre-gathered ops.PatchCols)."""
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        if isinstance(x, ops.PatchCols):          # operand rows gathered straight from a tile (patches.point_batch_cols)
            if x.patch != self.patch_size[0] or x.cols.dtype != ops.act_dtype(self.numerics):
                raise ValueError("PatchCols were built for another patch size / numerics mode")
            return x.cols
        return ops.patchify(x.float(), self.patch_size[0], ops.act_dtype(self.numerics))

    def forward(self, x):
        B = x.shape[0]
        cols = self.operand_rows(x)
        y = ops.LinearFn.apply(cols, self.proj.weight, self.proj.bias, None, torch.float32)
        return self.norm(y.view(B, self.num_patches, -1))


class Mlp(nn.Module):
    """fc2(GELU(fc1(x))) (reference :39-58); dropout is identity at the reference's drop=0."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., numerics=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:
            raise ValueError("the fused MLP kernel implements nn.GELU (erf) only")
        if drop != 0.:
            raise ValueError("dropout > 0 is not part of the accelerated path (reference uses drop=0)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self.numerics = _mode(numerics, self)

    def _run(self, x2d, residual2d, out_dtype):
        return ops.MlpFn.apply(x2d, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual2d, out_dtype)

    def forward(self, x):
        shp = x.shape
        xin = _CastFn.apply(x.reshape(-1, shp[-1]), ops.act_dtype(self.numerics))
        y = self._run(xin, None, torch.float32)
        return y.view(*shp[:-1], y.shape[-1])


class _CastFn(torch.autograd.Function):
    """fp32 -> operand dtype with a pass-through gradient (module boundaries of standalone blocks)."""

    @staticmethod
    def forward(ctx, x, dtype):
        return ops._as_operand(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return g.float(), None


class FeatureEmbed(nn.Module):
    """Designed-feature embedding: three k=1 Conv1d with GELU after the first only (reference :60-82).
    Always fp32: the 19 inputs are un-normalised physical quantities."""

    def __init__(self, feature_size=19, embed_dim=768, act_layer=nn.GELU, norm_layer=None, numerics=None):
        super().__init__()
        self.proj0 = nn.Conv1d(feature_size, embed_dim, kernel_size=1, stride=1)
        self.proj1 = nn.Conv1d(embed_dim, embed_dim, kernel_size=1, stride=1)
        self.proj2 = nn.Conv1d(embed_dim, embed_dim, kernel_size=1, stride=1)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        self.act = act_layer()
        self.numerics = _mode(numerics, self)

    def forward(self, x):
        B, T, F = x.shape                                    # [B, 1, 19]
        h = ops.MlpFn.apply(x.reshape(B * T, F).float(), self.proj0.weight, self.proj0.bias,
                            self.proj1.weight, self.proj1.bias, None, torch.float32)
        y = ops.LinearFn.apply(h, self.proj2.weight, self.proj2.bias, None, torch.float32)
        return self.norm(y.view(B, T, -1))


class CrossScaleAttention(nn.Module):
    """Global attention over the 3-D token cube with a relative-position bias table (reference :84-156).
    qkv and proj are MFMA GEMMs; the core (scale, QK^T, bias gather, softmax, PV) is one fused kernel."""

    def __init__(self, dim, num_heads, cube_size, qkv_bias=False, qk_scale=None, attn_drop_ratio=0., proj_drop_ratio=0,
                 numerics=None):
        super().__init__()
        if attn_drop_ratio != 0. or proj_drop_ratio != 0:
            raise ValueError("dropout > 0 is not part of the accelerated path (reference uses 0)")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop_ratio)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop_ratio)
        self.cube_size = cube_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self._table_rows(cube_size), num_heads))
        self.register_buffer("relative_position_index", self._make_index(cube_size))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.numerics = _mode(numerics, self)
        self._idx32 = None

    @staticmethod
    def _table_rows(cube):
        return (2 * cube[0] - 1) * (2 * cube[1] - 1) * (2 * cube[2] - 1)

    @staticmethod
    def _make_index(cube):
        return relative_position_index(cube)

    def _index32(self) -> torch.Tensor:
        src = self.relative_position_index
        tag = (src.data_ptr(), src._version, src.device)
        if self._idx32 is None or self._idx32[0] != tag:
            idx = src.to(torch.int32).contiguous()
            # The attention kernels can form the bias from the table themselves when the index is the closed form of a token cube
            # (reference :139-156); vouch for that only after comparing (a loaded checkpoint could carry any buffer).
            cube = tuple(int(c) for c in self.cube_size)
            if len(cube) == 3 and tuple(src.shape) == (cube[0] * cube[1] * cube[2],) * 2 and \
                    bool((src == relative_position_index(cube).to(src.device)).all()):
                idx._dm_cube = cube
            self._idx32 = (tag, idx)
        return self._idx32[1]

    def _run(self, y, residual2d, out_dtype):
        """y: [B, N, C] in the operand dtype.  Returns [B*N, C] = proj(attn(y)) (+ residual)."""
        B, N, Cc = y.shape
        H = self.num_heads
        qkv = ops.LinearFn.apply(y.reshape(B * N, Cc), self.qkv.weight, self.qkv.bias, None, y.dtype)
        o = ops.AttentionFn.apply(qkv, self.relative_position_bias_table, self._index32(), B, N, H, Cc // H, float(self.scale))
        return ops.LinearFn.apply(o.reshape(B * N, Cc), self.proj.weight, self.proj.bias, residual2d, out_dtype)

    def forward(self, x):
        B_, N, Cc = x.shape
        y = _CastFn.apply(x, ops.act_dtype(self.numerics))
        return self._run(y, None, torch.float32).view(B_, N, Cc)


class CrossScaleBlock(nn.Module):
    """Pre-norm block x += attn(LN(x)); x += mlp(LN(x)) (reference :158-184).  The residual stream stays
    fp32; both residual additions are fused into the proj / fc2 GEMM epilogues."""
    _dm_fused_block = True      # every parameter gets its gradient from ops.BlockFn.backward only (see trainer.FlatParams: first-write sinks)

    _attention_cls = CrossScaleAttention

    def __init__(self, dim, num_heads, cube_size, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_ratio=0.,
                 attn_drop_ratio=0., drop_path_ratio=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, numerics=None):
        super().__init__()
        self.numerics = _mode(numerics, self)
        self.norm1 = norm_layer(dim)
        self.attn = self._attention_cls(dim=dim, num_heads=num_heads, cube_size=cube_size, qkv_bias=qkv_bias,
                                        qk_scale=qk_scale, attn_drop_ratio=attn_drop_ratio, proj_drop_ratio=drop_ratio,
                                        numerics=self.numerics)
        # stochastic depth (reference :171): ratio 0 everywhere upstream -> the fused node; a ratio > 0 runs the block as separate
        # nodes in training mode (stochastic_block_forward), and its parameters then take gradients from several writers
        self.drop_path = DropPath(drop_path_ratio) if drop_path_ratio > 0. else nn.Identity()
        if drop_path_ratio > 0.:
            self._dm_fused_block = False
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop_ratio,
                       numerics=self.numerics)

    def forward(self, x):
        a, m = self.attn, self.mlp
        if self.training and isinstance(self.drop_path, DropPath):
            return stochastic_block_forward(self, x, None)
        if a.qkv.bias is None:
            raise ValueError("the fused block kernel path expects qkv_bias=True (the reference default, :165)")
        return ops.BlockFn.apply(x.float(), self.norm1.weight, self.norm1.bias, a.relative_position_bias_table, a._index32(),
                                 a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias,
                                 m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                 a.num_heads, self.norm1.eps, float(a.scale), ops.act_dtype(self.numerics))


class ShfitScaleFormer_v3(nn.Module):
    """Multi-scale Siamese encoder, the variant the reference trains and serves (reference :772-1010)."""
    _dm_first_write_blocks = True     # the blocks are only ever run through their fused forward: FlatParams need not zero their gradients

    def __init__(self, num_classes=11, is_designed_feature_embedding=True, FeatureEmbed=FeatureEmbed, PatchEmbed=PatchEmbed,
                 cube_size=[8, 8], input_image_scales=[32, 64, 128], embed_dim=768, depth=[6, 4, 2], num_heads=12,
                 mlp_ratio=4.0, drop_path_ratio=0., drop_ratio=0., attn_drop_ratio=0., norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, cuda=True, in_c=3, numerics=None):
        super().__init__()
        self.numerics = _mode(numerics, self)
        self.name = "S2Former_v3-3CH" + ("-3DP-SEF" if is_designed_feature_embedding else "")
        self.name = "{0}-{1}{2}{3}".format(self.name, depth[0], depth[1], depth[2])
        self.num_classes = num_classes
        self.is_designed_feature_embedding = is_designed_feature_embedding
        self.patch_embed_layer, self.feature_embed_layer = PatchEmbed, FeatureEmbed
        self.input_image_scales = input_image_scales
        self.input_scales_num = S = len(input_image_scales)
        self.cube_size = cube_size
        self.cube_size.insert(0, S)          # the reference mutates the caller's list the same way (:804)
        grid = self.cube_size[1]
        self.num_features = int(S * embed_dim)
        self.depth = depth
        self.in_c = in_c
        kw = {"numerics": self.numerics} if PatchEmbed is globals()["PatchEmbed"] else {}
        self.patch_embed_blocks = nn.ModuleList(
            [PatchEmbed(img_size=s, patch_size=int(s / grid), in_c=in_c, out_c=embed_dim, **kw) for s in input_image_scales])
        self.feature_embed = FeatureEmbed(feature_size=19, embed_dim=embed_dim) if is_designed_feature_embedding else None

        def stage(cube, n):
            return nn.Sequential(*[getattr(self, "_block_cls", CrossScaleBlock)(dim=embed_dim, num_heads=num_heads, cube_size=cube, mlp_ratio=mlp_ratio,
                                                   drop_ratio=drop_ratio, attn_drop_ratio=attn_drop_ratio, drop_path_ratio=0,
                                                   norm_layer=norm_layer, act_layer=act_layer, numerics=self.numerics)
                                   for _ in range(n)])
        self.blocks0 = stage(self.cube_size, depth[0])
        self.blocks1 = stage([S, grid // 2, grid // 2], depth[1])
        self.blocks2 = stage([S, grid // 4, grid // 4], depth[2])
        self.norm = norm_layer(embed_dim)
        self.pos_drop = nn.Dropout(p=drop_ratio)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.final_features = nn.Linear(int(S * embed_dim), 100)
        self.final_features_with_design = nn.Linear(int((S + 1) * embed_dim), 100)
        self.head = nn.Linear(100, num_classes) if num_classes > 0 else nn.Identity()
        self.avgpool2D = nn.AvgPool2d(kernel_size=2, stride=2)
        self.apply(self._init_weights)

    # -- pieces (public names as in the reference) ------------------------------------------------
    def patch_embed(self, x: List[torch.Tensor]):
        layers = list(self.patch_embed_blocks)
        if not all(type(l) is PatchEmbed and isinstance(l.norm, nn.Identity) and l.num_patches == layers[0].num_patches for l in layers):
            return torch.cat([layer(x[i]) for i, layer in enumerate(layers)], 1)          # caller-supplied embed class: generic form
        # every scale's GEMM writes its tokens at their offset of the token cube: no cat (reference :869-882)
        cols = [l.operand_rows(x[i]) for i, l in enumerate(layers)]
        return ops.PatchEmbedCatFn.apply(len(layers), layers[0].num_patches, *cols, *[l.proj.weight for l in layers], *[l.proj.bias for l in layers])

    def designed_feature_embed(self, x):
        return self.feature_embed(x)

    def _ln(self, x, out_dtype=torch.float32):
        return ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype)

    _dp_cut = None      # set by PairTrainer for one forward pass: autograd cut points for the segmented (data-parallel) backward

    def backbone(self, x):
        S, side = self.input_scales_num, self.cube_size[1]
        if self._dp_cut is None:
            x = self.blocks0(x)
        else:                                    # stage 0 holds ~55 % of the step: one backward segment per block
            for blk in self.blocks0:
                x = self._dp_cut(blk(x), blk)
        x = self._ln(ops.TokenPoolFn.apply(x, S, side))
        x = self.blocks1(x)
        x = self._ln(ops.TokenPoolFn.apply(x, S, side // 2))
        return self.blocks2(x)

    def _pooled_tokens(self, x):
        B = x[0].shape[0]
        x = self.pos_drop(self.patch_embed(x))
        if self._dp_cut is not None:             # the patch embeds' gradients (17 MB at 4 scales x 4 ch) get a bucket of their own: the
            x = self._dp_cut(x, self.patch_embed_blocks)   # exchange of stage-0 block 0 then starts before their backward, not after it
        x = self._ln(self.backbone(x))
        g = x.shape[1] // self.input_scales_num
        return ops.GroupMeanFn.apply(x, g).view(B, -1)

    def forward_once_design_feature(self, x, designed_features):
        x = self._pooled_tokens(x)
        d = torch.squeeze(self.designed_feature_embed(designed_features), dim=1)
        x = torch.cat((x, self._ln(d)), 1)
        return ops.LinearFn.apply(x, self.final_features_with_design.weight, self.final_features_with_design.bias, None, torch.float32)

    def forward_once(self, x):
        x = self._pooled_tokens(x)
        return ops.LinearFn.apply(x, self.final_features.weight, self.final_features.bias, None, torch.float32)

    def extract_features_with_design_features(self, x_path, x_designed_features):
        return self.forward_once_design_feature(x_path, x_designed_features)

    def extract_features(self, x_path):
        return self.forward_once(x_path)

    def forward(self, x1_patches, x1_designed_features, x2_patches=None, x2_designed_features=None):
        """Training mode returns (feature1, feature2); eval mode returns feature1 (reference :977-999).
        Both sides share every weight and no op mixes samples, so they run as ONE batch of 2B."""
        S = self.input_scales_num
        if self.training:
            B = x1_patches[0].shape[0]
            both = [ops.cat_batch(x1_patches[i], x2_patches[i]) for i in range(S)]
            if self.is_designed_feature_embedding:
                f = self.forward_once_design_feature(both, torch.cat((x1_designed_features, x2_designed_features), 0))
            else:
                f = self.forward_once(both)
            return ops.split_halves(f)
        if self.is_designed_feature_embedding:
            return self.forward_once_design_feature(x1_patches, x1_designed_features)
        return self.forward_once(x1_patches)

    def forward_pair_batched(self, both_patches, both_designed):
        """The training forward on inputs that already hold both sides of every pair stacked along the batch
        ([left; right], 2B samples): what forward() builds with torch.cat.  The trainer's graph replay copies the
        step's inputs straight into such buffers.  Returns (feature1, feature2)."""
        B = both_patches[0].shape[0] // 2
        if self.is_designed_feature_embedding:
            f = self.forward_once_design_feature(both_patches, both_designed)
        else:
            f = self.forward_once(both_patches)
        return ops.split_halves(f)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


def relative_position_index_v5(cube: Sequence[int]) -> torch.Tensor:
    """[N+1, N+1] index of the v5 attention: the cube index plus one column / one row of fresh ids for the
    designed-feature token, the corner aliased to entry [0,0] (reference :217-263)."""
    base = relative_position_index(cube)
    n = base.shape[0]
    bins = CrossScaleAttention._table_rows(cube)
    col = bins + torch.arange(n, dtype=torch.int64)[:, None]
    idx = torch.cat([base, col], 1)
    row = bins + n + torch.arange(n + 1, dtype=torch.int64)[None, :]
    idx = torch.cat([idx, row], 0)
    idx[-1, -1] = idx[0, 0]
    return idx


class CrossScaleAttention_v5(CrossScaleAttention):
    """Attention over the token cube plus ONE extra token (the embedded designed features); the bias table
    grows by 2 * S*H*W rows (reference :187-296).  Same fused kernels: only the index / table differ."""

    @staticmethod
    def _table_rows(cube):
        return CrossScaleAttention._table_rows(cube) + 2 * cube[0] * cube[1] * cube[2]

    @staticmethod
    def _make_index(cube):
        return relative_position_index_v5(cube)


class CrossScaleBlock_v5(CrossScaleBlock):
    """Reference :298-327."""
    _attention_cls = CrossScaleAttention_v5


class AuxBolck(nn.Module):
    """Auxiliary head on an intermediate stage (reference :329-368): per scale Conv2d(k=2, no bias) ->
    BatchNorm2d -> ReLU -> Dropout2d(0.3) -> Conv2d(1x1, C/S) -> mean over positions; concatenated ->
    LayerNorm -> Linear(C, 100).  Both convolutions run as MFMA GEMMs (the 2x2 one over gathered windows);
    BatchNorm2d + ReLU + the Dropout2d mask are one HIP kernel family (dm_batchnorm_fwd / _bwd) on the channels-last
    GEMM output; the torch modules only own the parameters, the running statistics and the dropout probability.
    Data parallel: statistics are PER RANK, like the reference under plain DataParallel-free training on one GPU per
    process (no SyncBN: the aux heads see >= 120 x 49 positions per rank at the reference batch size, and SyncBN would
    add two collectives per head per step); gamma / beta gradients are averaged with all the others."""
    _v5 = False

    def __init__(self, in_c=768, out_c=100, cube_size=[3, 8, 8], norm_layer=nn.LayerNorm, numerics=None):
        super().__init__()
        self.cube_size = cube_size
        self.aux = nn.Sequential(
            nn.Conv2d(in_c, in_c, kernel_size=2, padding=0, bias=False),
            nn.BatchNorm2d(in_c),
            nn.ReLU(inplace=True),
            nn.Dropout2d(p=0.3),
            nn.Conv2d(in_c, int(in_c / cube_size[0]), kernel_size=1))
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        width = in_c * 2 if self._v5 else in_c
        self.norm = norm_layer(width)
        self.out_features = nn.Linear(width, out_c)
        self.numerics = _mode(numerics, self)

    def _scale_head(self, t):
        """t: [B, side*side, C] fp32 tokens of one scale -> [B, C/S]."""
        B, n, Cc = t.shape
        side = int(math.isqrt(n))
        o = side - 1
        t4 = t.reshape(B, side, side, Cc)
        # window gather for the 2x2 / stride-1 convolution; element order (c, ky, kx) = the weight's own layout
        cols = torch.stack([t4[:, ky:ky + o, kx:kx + o, :] for ky in (0, 1) for kx in (0, 1)], dim=-1).reshape(B * o * o, Cc * 4)
        dt = ops.act_dtype(self.numerics)
        y = ops.LinearFn.apply(_CastFn.apply(cols, dt), self.aux[0].weight, None, None, torch.float32)
        # BatchNorm2d -> ReLU -> Dropout2d in one HIP pass over the channels-last GEMM output (no NCHW round trip).  The
        # dropout mask is drawn the way torch's feature dropout draws it (bernoulli(1 - p) per (sample, channel), / (1 - p))
        bn, drop = self.aux[1], self.aux[3]
        if bn.momentum is None or not bn.track_running_stats or not bn.affine:
            raise ValueError("the HIP BatchNorm path implements the reference's default BatchNorm2d (affine, momentum 0.1, running stats)")
        training = self.training
        mask = None
        if training and drop.p > 0:
            mask = torch.empty((B, Cc), dtype=torch.float32, device=y.device).bernoulli_(1.0 - drop.p).div_(1.0 - drop.p)
        y = ops.BatchNormReluFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, mask, o * o, bn.eps, bn.momentum, training, True)
        if training:
            bn.num_batches_tracked += 1
        y = ops.LinearFn.apply(_CastFn.apply(y, dt), self.aux[4].weight, self.aux[4].bias, None, torch.float32)
        return ops.GroupMeanFn.apply(y.view(B, o * o, -1), o * o).view(B, -1)

    def forward(self, x):
        S, side = self.cube_size[0], self.cube_size[1]
        n = side * side
        y = torch.cat([self._scale_head(x[:, n * i:n * (i + 1), :]) for i in range(S)], 1)
        if self._v5:
            y = torch.cat([y, x[:, n * S:, :].transpose(1, 2).flatten(1)], 1)      # + the designed-feature token; no norm (:412)
        else:
            y = ops.LayerNormFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)
        return ops.LinearFn.apply(y, self.out_features.weight, self.out_features.bias, None, torch.float32)


class AuxBolck_v5(AuxBolck):
    """Reference :370-415 (LayerNorm(2C) is registered but never applied upstream; kept for state_dict parity)."""
    _v5 = True


class ShfitScaleFormer_v4(ShfitScaleFormer_v3):
    """v3 backbone + two auxiliary heads after blocks0 / blocks1 (reference :1013-1261).  Training returns
    ((x, aux0, aux1), (x, aux0, aux1)); eval returns x.  Three scales / 3 channels, as upstream.

    The two sides run as one batch of 2B through the backbone, but through the aux heads side by side:
    BatchNorm2d statistics (and their running updates) are per forward_once call upstream."""
    _dm_first_write_blocks = False    # (not inherited: the auxiliary heads are separate autograd nodes; plain zero + accumulate)

    def __init__(self, num_classes=11, is_designed_feature_embedding=True, FeatureEmbed=FeatureEmbed, PatchEmbed=PatchEmbed,
                 cube_size=[8, 8], input_image_scales=[32, 64, 128], embed_dim=768, depth=[3, 2, 1], num_heads=12,
                 mlp_ratio=4.0, drop_path_ratio=0., drop_ratio=0., attn_drop_ratio=0., norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, cuda=True, numerics=None):
        if len(input_image_scales) != 3:
            raise ValueError("v4's auxiliary heads are built for exactly three input scales (reference :1097-1098, :1152)")
        super().__init__(num_classes=num_classes, is_designed_feature_embedding=is_designed_feature_embedding,
                         FeatureEmbed=FeatureEmbed, PatchEmbed=PatchEmbed, cube_size=cube_size,
                         input_image_scales=input_image_scales, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                         mlp_ratio=mlp_ratio, drop_path_ratio=drop_path_ratio, drop_ratio=drop_ratio,
                         attn_drop_ratio=attn_drop_ratio, norm_layer=norm_layer, act_layer=act_layer, cuda=cuda, in_c=3,
                         numerics=numerics)
        self.name = "S2Former_v4-3CH" + ("-SFE" if is_designed_feature_embedding else "")
        self.name = "{0}-{1}{2}{3}".format(self.name, depth[0], depth[1], depth[2])
        self._build_aux()
        self.apply(self._init_weights)

    def _build_aux(self):
        self.aux0 = AuxBolck(numerics=self.numerics)
        self.aux1 = AuxBolck(cube_size=[3, 4, 4], numerics=self.numerics)

    def _aux(self, head, x, sides):
        """Aux head per side (None in eval, where upstream discards it and BatchNorm has no side effect)."""
        if not self.training:
            return None
        B = x.shape[0] // sides
        return [head(x[i * B:(i + 1) * B]) for i in range(sides)]

    def backbone(self, x, sides=1):
        S, side = self.input_scales_num, self.cube_size[1]
        x = self.blocks0(x)
        aux0 = self._aux(self.aux0, x, sides)
        x = self._ln(ops.TokenPoolFn.apply(x, S, side))
        x = self.blocks1(x)
        aux1 = self._aux(self.aux1, x, sides)
        x = self._ln(ops.TokenPoolFn.apply(x, S, side // 2))
        return self.blocks2(x), aux0, aux1

    def _encode(self, patches, designed, sides):
        B2 = patches[0].shape[0]
        x, aux0, aux1 = self.backbone(self.pos_drop(self.patch_embed(patches)), sides)
        x = self._ln(x)
        x = ops.GroupMeanFn.apply(x, x.shape[1] // self.input_scales_num).view(B2, -1)
        if self.is_designed_feature_embedding:
            d = torch.squeeze(self.designed_feature_embed(designed), dim=1)
            x = torch.cat((x, self._ln(d)), 1)
            w = self.final_features_with_design
        else:
            w = self.final_features
        return ops.LinearFn.apply(x, w.weight, w.bias, None, torch.float32), aux0, aux1

    def forward_once_design_feature(self, x, designed_features):
        y, aux0, aux1 = self._encode(x, designed_features, 1)
        return (y, aux0[0], aux1[0]) if self.training else y

    def forward_once(self, x):
        return self.forward_once_design_feature(x, None)

    def forward(self, x1_patches, x1_designed_features, x2_patches=None, x2_designed_features=None):
        if not self.training:
            return self._encode(x1_patches, x1_designed_features, 1)[0]
        B = x1_patches[0].shape[0]
        both = [ops.cat_batch(x1_patches[i], x2_patches[i]) for i in range(self.input_scales_num)]
        d = torch.cat((x1_designed_features, x2_designed_features), 0) if self.is_designed_feature_embedding else None
        y, aux0, aux1 = self._encode(both, d, 2)
        return (y[:B], aux0[0], aux1[0]), (y[B:], aux0[1], aux1[1])


class ShfitScaleFormer_v5(ShfitScaleFormer_v4):
    """v4 with the embedded designed features appended to the token cube as one extra TOKEN (N + 1 = 193 at
    stage 0), carried through both poolings, then `last_block_features` Linear((S+1)C, C) and the final
    Linear(2C, 100) (reference :1264-1503)."""

    def __init__(self, num_classes=11, FeatureEmbed=FeatureEmbed, PatchEmbed=PatchEmbed, CrossScaleBlock=CrossScaleBlock_v5,
                 AuxBolck=AuxBolck_v5, cube_size=[8, 8], input_image_scales=[32, 64, 128], embed_dim=768, depth=[3, 2, 1],
                 num_heads=12, mlp_ratio=4.0, drop_path_ratio=0., drop_ratio=0., attn_drop_ratio=0., norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, cuda=True, numerics=None):
        self._block_cls, self._aux_cls = CrossScaleBlock, AuxBolck
        super().__init__(num_classes=num_classes, is_designed_feature_embedding=True, FeatureEmbed=FeatureEmbed,
                         PatchEmbed=PatchEmbed, cube_size=cube_size, input_image_scales=input_image_scales,
                         embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                         drop_path_ratio=drop_path_ratio, drop_ratio=drop_ratio, attn_drop_ratio=attn_drop_ratio,
                         norm_layer=norm_layer, act_layer=act_layer, cuda=cuda, numerics=numerics)
        del self.name                                     # upstream v5 has no name attribute

    def _build_aux(self):
        S, C = self.input_scales_num, self.norm.normalized_shape[0]
        self.final_features_with_design = nn.Linear(2 * C, 100)
        self.last_block_features = nn.Linear(int((S + 1) * C), C)
        head = self.head                                  # keep upstream registration order: ..., last_block_features, head, aux
        del self.head
        self.head = head
        self.aux0 = self._aux_cls(numerics=self.numerics)
        self.aux1 = self._aux_cls(cube_size=[3, 4, 4], numerics=self.numerics)

    def _pool_keep_last(self, x, side):
        S = self.input_scales_num
        n = S * side * side
        return torch.cat([ops.TokenPoolFn.apply(x[:, :n].contiguous(), S, side), x[:, n:]], 1)

    def backbone(self, x, sides=1):
        side = self.cube_size[1]
        x = self.blocks0(x)
        aux0 = self._aux(self.aux0, x, sides)
        x = self._ln(self._pool_keep_last(x, side))
        x = self.blocks1(x)
        aux1 = self._aux(self.aux1, x, sides)
        x = self._ln(self._pool_keep_last(x, side // 2))
        return self.blocks2(x), aux0, aux1

    def _encode(self, patches, designed, sides):
        if designed is None:
            raise ValueError("v5 always consumes designed features (upstream forward_once is not runnable: :1464-1478)")
        S = self.input_scales_num
        B2 = patches[0].shape[0]
        f = self._ln(torch.squeeze(self.designed_feature_embed(designed), dim=1))
        x = torch.cat((self.pos_drop(self.patch_embed(patches)), f.unsqueeze(1)), 1)
        x, aux0, aux1 = self.backbone(x, sides)
        x = self._ln(x)
        n = S * 4
        y = torch.cat([ops.GroupMeanFn.apply(x[:, :n].contiguous(), 4).view(B2, -1), x[:, n:].mean(dim=1)], 1)
        y = ops.LinearFn.apply(y, self.last_block_features.weight, self.last_block_features.bias, None, torch.float32)
        y = torch.cat((y, f), 1)
        w = self.final_features_with_design
        return ops.LinearFn.apply(y, w.weight, w.bias, None, torch.float32), aux0, aux1


class _SingleStage(nn.Module):
    """Shared body of the single-stage variants: S patch embeds -> blocks on the [S, side, side] cube -> norm ->
    per-scale token mean -> (+ normed designed-feature embedding) -> Linear to 100-d."""

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _build_tail(self, embed_dim, norm_layer, drop_ratio, num_classes):
        S = self.input_scales_num
        self.norm = norm_layer(embed_dim)
        self.pos_drop = nn.Dropout(p=drop_ratio)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.final_features = nn.Linear(int(S * embed_dim), 100)
        self.final_features_with_design = nn.Linear(int((S + 1) * embed_dim), 100)
        self.head = nn.Linear(100, num_classes) if num_classes > 0 else nn.Identity()

    def _ln(self, x):
        return ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)

    def designed_feature_embed(self, x):
        return self.feature_embed(x)

    def _scale_means(self, x):
        B = x[0].shape[0]
        t = self._ln(self.blocks(self.pos_drop(self.patch_embed(x))))
        return ops.GroupMeanFn.apply(t, t.shape[1] // self.input_scales_num).view(B, -1)

    def forward_once_design_feature(self, x, designed_features):
        x = self._scale_means(x)
        d = self._ln(torch.squeeze(self.designed_feature_embed(designed_features), dim=1))
        return ops.LinearFn.apply(torch.cat((x, d), 1), self.final_features_with_design.weight,
                                  self.final_features_with_design.bias, None, torch.float32)

    def forward_once(self, x):
        return ops.LinearFn.apply(self._scale_means(x), self.final_features.weight, self.final_features.bias, None, torch.float32)

    def extract_features_with_design_features(self, x_path, x_designed_features):
        return self.forward_once_design_feature(x_path, x_designed_features)

    def extract_features(self, x_path):
        return self.forward_once(x_path)

    def _pair(self, x1, d1, x2, d2):
        """Both sides as one batch of 2B (shared weights, per-sample ops only)."""
        B = x1[0].shape[0]
        both = [torch.cat((x1[i], x2[i]), 0) for i in range(self.input_scales_num)]
        if self.is_designed_feature_embedding:
            f = self.forward_once_design_feature(both, torch.cat((d1, d2), 0))
        else:
            f = self.forward_once(both)
        return ops.split_halves(f)


class ShfitScaleFormer(_SingleStage):
    """First generation (reference :417-607): four fixed scales [28,56,112,224] -> 4 x 49 tokens, `depth` blocks on
    the [4,7,7] cube.  forward dispatches on which arguments are None, not on train/eval (:571-590)."""

    def __init__(self, num_classes=11, is_designed_feature_embedding=True, FeatureEmbed=FeatureEmbed, PatchEmbed=PatchEmbed,
                 cube_size=[7, 7], input_image_scales=[28, 56, 112, 224], embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 drop_path_ratio=0., drop_ratio=0., attn_drop_ratio=0., norm_layer=nn.LayerNorm, act_layer=nn.GELU, cuda=True,
                 numerics=None):
        super().__init__()
        self.numerics = _mode(numerics, self)
        self.num_classes = num_classes
        self.is_designed_feature_embedding = is_designed_feature_embedding
        self.patch_embed_layer, self.feature_embed_layer = PatchEmbed, FeatureEmbed
        self.input_image_scales = input_image_scales
        self.input_scales_num = len(input_image_scales)
        self.cube_size = cube_size
        self.cube_size.insert(0, self.input_scales_num)
        self.num_features = int(self.input_scales_num * embed_dim)
        kw = {"numerics": self.numerics} if PatchEmbed is globals()["PatchEmbed"] else {}
        for i, ps in enumerate((4, 8, 16, 32)):           # hard-coded upstream (:454-457)
            setattr(self, f"patch_embed_scale{i}", PatchEmbed(img_size=input_image_scales[i], patch_size=ps, in_c=3, out_c=768, **kw))
        self.feature_embed = FeatureEmbed(feature_size=19, embed_dim=768) if is_designed_feature_embedding else None
        dpr = [x.item() for x in torch.linspace(0, drop_path_ratio, depth)]      # stochastic depth decay rule (reference :463)
        self.blocks = nn.Sequential(*[
            CrossScaleBlock(dim=embed_dim, num_heads=num_heads, cube_size=self.cube_size, mlp_ratio=mlp_ratio, drop_ratio=drop_ratio,
                            attn_drop_ratio=attn_drop_ratio, drop_path_ratio=dpr[i], norm_layer=norm_layer, act_layer=act_layer,
                            numerics=self.numerics) for i in range(depth)])
        self._build_tail(embed_dim, norm_layer, drop_ratio, num_classes)
        self.apply(self._init_weights)

    def patch_embed(self, x):
        return torch.cat([getattr(self, f"patch_embed_scale{i}")(x[i]) for i in range(4)], 1)

    def forward(self, x1_patches, x1_designed_features, x2_patches=None, x2_designed_features=None):
        if x1_designed_features is not None and x2_patches is None and x2_designed_features is None:
            return self.extract_features_with_design_features(x1_patches, x1_designed_features)
        if x1_designed_features is None and x2_patches is None and x2_designed_features is None:
            # upstream calls extract_features with two arguments here and raises TypeError (:578); the intent is clear
            return self.extract_features(x1_patches)
        return self._pair(x1_patches, x1_designed_features, x2_patches, x2_designed_features)


class ShfitScaleFormer_v2(_SingleStage):
    """Second generation (reference :610-769): per-scale embeds in a ModuleList, always 12 blocks (`depth` is ignored
    upstream, :657), train/eval dispatch like v3."""

    def __init__(self, num_classes=11, is_designed_feature_embedding=True, FeatureEmbed=FeatureEmbed, PatchEmbed=PatchEmbed,
                 cube_size=[7, 7], input_image_scales=[28, 56, 112, 224], embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 drop_path_ratio=0., drop_ratio=0., attn_drop_ratio=0., norm_layer=nn.LayerNorm, act_layer=nn.GELU, cuda=True,
                 numerics=None):
        super().__init__()
        self.numerics = _mode(numerics, self)
        self.num_classes = num_classes
        self.is_designed_feature_embedding = is_designed_feature_embedding
        self.patch_embed_layer, self.feature_embed_layer = PatchEmbed, FeatureEmbed
        self.input_image_scales = input_image_scales
        self.input_scales_num = len(input_image_scales)
        self.cube_size = cube_size
        self.cube_size.insert(0, self.input_scales_num)
        self.num_features = int(self.input_scales_num * embed_dim)
        kw = {"numerics": self.numerics} if PatchEmbed is globals()["PatchEmbed"] else {}
        self.patch_embed_blocks = nn.ModuleList(
            [PatchEmbed(img_size=s, patch_size=int(s / self.cube_size[1]), in_c=3, out_c=768, **kw) for s in input_image_scales])
        self.feature_embed = FeatureEmbed(feature_size=19, embed_dim=768) if is_designed_feature_embedding else None
        self.blocks = nn.Sequential(*[
            CrossScaleBlock(dim=embed_dim, num_heads=num_heads, cube_size=self.cube_size, mlp_ratio=mlp_ratio, drop_ratio=drop_ratio,
                            attn_drop_ratio=attn_drop_ratio, drop_path_ratio=0, norm_layer=norm_layer, act_layer=act_layer,
                            numerics=self.numerics) for _ in range(12)])
        self._build_tail(embed_dim, norm_layer, drop_ratio, num_classes)
        self.apply(self._init_weights)

    def patch_embed(self, x):
        return torch.cat([layer(x[i]) for i, layer in enumerate(self.patch_embed_blocks)], 1)

    def forward(self, x1_patches, x1_designed_features, x2_patches=None, x2_designed_features=None):
        if self.training:
            return self._pair(x1_patches, x1_designed_features, x2_patches, x2_designed_features)
        if self.is_designed_feature_embedding:
            return self.forward_once_design_feature(x1_patches, x1_designed_features)
        return self.forward_once(x1_patches)


class ShfitScaleFormer_v6(_SingleStage):
    """Designed-features-only network (reference :1506-1569): FeatureEmbed -> LayerNorm -> Linear(768, 100)."""

    def __init__(self, num_classes=11, FeatureEmbed=FeatureEmbed, embed_dim=768, mlp_ratio=4.0, drop_path_ratio=0., drop_ratio=0.,
                 norm_layer=nn.LayerNorm, act_layer=nn.GELU, cuda=True, numerics=None):
        super().__init__()
        self.numerics = _mode(numerics, self)
        self.num_classes = num_classes
        self.feature_embed_layer = FeatureEmbed
        self.feature_embed = FeatureEmbed(feature_size=19, embed_dim=768)
        self.norm = norm_layer(embed_dim)
        self.pos_drop = nn.Dropout(p=drop_ratio)
        self.final_features_with_design = nn.Linear(embed_dim, 100)
        self.head = nn.Linear(100, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)

    def designed_feature_embed(self, x):
        return self._ln(torch.squeeze(self.feature_embed(x), dim=1))

    def forward_once_design_feature(self, x, designed_features):
        return ops.LinearFn.apply(self.designed_feature_embed(designed_features), self.final_features_with_design.weight,
                                  self.final_features_with_design.bias, None, torch.float32)

    def forward(self, x1_patches, x1_designed_features, x2_patches=None, x2_designed_features=None):
        if x1_designed_features is not None and x2_patches is None and x2_designed_features is None:
            return self.extract_features_with_design_features(x1_patches, x1_designed_features)
        if x2_designed_features is None:
            raise TypeError("ShfitScaleFormer_v6 needs designed features (upstream raises here too, :1549)")
        B = x1_designed_features.shape[0]
        f = self.forward_once_design_feature(None, torch.cat((x1_designed_features, x2_designed_features), 0))
        return ops.split_halves(f)
