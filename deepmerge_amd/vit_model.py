"""MI355X-native drop-in for the reference module `vit_model.py` (the earlier model generation:
`Train_SMT.py:26, :363`, `ExtractFeatures.py:307`; BASELINE configs[2] = ViT-B/16 pair encoder).

Same class / factory names, constructor keywords, `forward(*args)` arity dispatch and state_dict keys as
the reference.  Blocks run through the fused `ops.BlockFn` (no bias table: plain attention, scale applied
after q @ k^T upstream, which is bit-identical for the power-of-two 64^-0.5); LayerNorm eps = 1e-6.
torch nn.Linear/Conv/LayerNorm objects are parameter containers only.

Supported on the accelerated path: head dims <= 256 (64 on the MFMA attention kernels, others on the generic fp32 family) and
embed dims <= 8192 (<= 1024 on the register-resident LayerNorm kernels, wider rows on the streamed ones of dm_rows_wide.hip).
ViT-H/14 (dim 1280, head dim 80, 14-pixel patches, 257 tokens; vit_model.py:649-662) therefore runs, on those secondary
kernels: a correctness path, not a tuned one (upstream never shipped its weights).
"""
from __future__ import annotations

from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import ops
from .nets.ShfitScaleFormer import FeatureEmbed as _S2FeatureEmbed
from .nets.ShfitScaleFormer import Mlp as _S2Mlp
from .nets.ShfitScaleFormer import DropPath, _CastFn, _mode, drop_path, stochastic_block_forward  # noqa: F401


class PatchEmbed(nn.Module):
    """vit_model.py:43-68 (keyword is `embed_dim` here, `out_c` in nets/ShfitScaleFormer.py)."""

    def __init__(self, img_size=224, patch_size=16, in_c=3, embed_dim=768, norm_layer=None, numerics=None):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        self.numerics = _mode(numerics, self)

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        K = C * self.patch_size[0] * self.patch_size[1]
        # bf16 GEMM operands need 16-byte rows (K % 8 == 0); a 14-pixel patch (K = 588) keeps this one small GEMM in fp32
        cols = ops.patchify(x.float(), self.patch_size[0], ops.act_dtype(self.numerics) if K % 8 == 0 else torch.float32)
        y = ops.LinearFn.apply(cols, self.proj.weight, self.proj.bias, None, torch.float32)
        return self.norm(y.view(B, self.num_patches, -1))


FeatureEmbed = _S2FeatureEmbed      # identical definition upstream (vit_model.py:70-92)
Mlp = _S2Mlp                        # identical definition upstream (vit_model.py:138-157)


class Attention(nn.Module):
    """Plain multi-head attention (vit_model.py:95-135): qkv GEMM, fused softmax(q k^T * scale) v, proj GEMM."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop_ratio=0., proj_drop_ratio=0., numerics=None):
        super().__init__()
        if attn_drop_ratio != 0. or proj_drop_ratio != 0.:
            raise ValueError("dropout > 0 is not part of the accelerated path (reference uses 0)")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim > 256:
            raise NotImplementedError(f"attention kernels cover head dims up to 256 (got {head_dim})")
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop_ratio)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop_ratio)
        self.numerics = _mode(numerics, self)

    def forward(self, x):
        B, N, Cc = x.shape
        H = self.num_heads
        y = _CastFn.apply(x, ops.act_dtype(self.numerics))
        qkv = ops.LinearFn.apply(y.reshape(B * N, Cc), self.qkv.weight, self.qkv.bias, None, y.dtype)
        o = ops.AttentionFn.apply(qkv, None, None, B, N, H, Cc // H, float(self.scale))
        return ops.LinearFn.apply(o.reshape(B * N, Cc), self.proj.weight, self.proj.bias, None, torch.float32).view(B, N, Cc)


class Block(nn.Module):
    """Pre-norm block (vit_model.py:160-185) as ONE fused autograd node."""
    _dm_fused_block = True      # see nets/ShfitScaleFormer.py CrossScaleBlock

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_ratio=0., attn_drop_ratio=0.,
                 drop_path_ratio=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, numerics=None):
        super().__init__()
        self.numerics = _mode(numerics, self)
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop_ratio=attn_drop_ratio,
                              proj_drop_ratio=drop_ratio, numerics=self.numerics)
        # stochastic depth (vit_model.py:12-40, :171): see nets/ShfitScaleFormer.py CrossScaleBlock
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
            zero = torch.zeros(a.qkv.out_features, device=x.device)      # qkv_bias=False: bias-free GEMM epilogue
            qb = zero
        else:
            qb = a.qkv.bias
        return ops.BlockFn.apply(x.float(), self.norm1.weight, self.norm1.bias, None, None, a.qkv.weight, qb, a.proj.weight,
                                 a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight,
                                 m.fc2.bias, a.num_heads, self.norm1.eps, float(a.scale), ops.act_dtype(self.numerics))


def _check_dim(embed_dim):
    if embed_dim > 8192 or embed_dim % 4:
        raise NotImplementedError(f"embed_dim {embed_dim}: the row kernels take multiples of 4 up to 8192")


class _Head(nn.Module):
    """Small fp32 linear (+ optional tanh) heads: `head`, `my_head`, `pre_logits.fc`."""

    @staticmethod
    def linear(x, lin: nn.Linear):
        return ops.LinearFn.apply(x.float().contiguous(), lin.weight, lin.bias, None, torch.float32)


class VisionTransformer(nn.Module):
    """vit_model.py:188-317.  forward(*args): 1 / 2 / 3 tensors -> once / twice / thrice, else ValueError."""
    _dm_first_write_blocks = True

    def __init__(self, img_size=224, patch_size=16, in_c=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, qk_scale=None, representation_size=None, distilled=False, drop_ratio=0.,
                 attn_drop_ratio=0., drop_path_ratio=0., embed_layer=PatchEmbed, norm_layer=None, act_layer=None, numerics=None):
        super().__init__()
        _check_dim(embed_dim)
        self.numerics = _mode(numerics, self)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 2 if distilled else 1
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        kw = {"numerics": self.numerics} if embed_layer is PatchEmbed else {}
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_c=in_c, embed_dim=embed_dim, **kw)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if distilled else None      # DeiT token (:225)
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_ratio)
        dpr = [x.item() for x in torch.linspace(0, drop_path_ratio, depth)]      # stochastic depth decay rule (vit_model.py:229 / :387)
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop_ratio=drop_ratio, attn_drop_ratio=attn_drop_ratio, drop_path_ratio=dpr[i], norm_layer=norm_layer,
                  act_layer=act_layer, numerics=self.numerics) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        if representation_size and not distilled:
            self.has_logits = True
            self.num_features = representation_size
            self.pre_logits = nn.Sequential(OrderedDict([("fc", nn.Linear(embed_dim, representation_size)), ("act", nn.Tanh())]))
        else:
            self.has_logits = False
            self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        if distilled:                                   # :250-253
            self.head_dist = nn.Linear(self.embed_dim, self.num_classes) if num_classes > 0 else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        if self.dist_token is not None:
            nn.init.trunc_normal_(self.dist_token, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(_init_vit_weights)

    def _encode(self, x):
        """tokens -> blocks -> LayerNorm (every row)"""
        x = self.blocks(self.pos_drop(x))
        return ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)

    def forward_features(self, x):
        x = self.patch_embed(x)
        cls = self.cls_token.expand(x.shape[0], -1, -1)
        if self.dist_token is None:
            x = torch.cat((cls, x), dim=1) + self.pos_embed
        else:
            x = torch.cat((cls, self.dist_token.expand(x.shape[0], -1, -1), x), dim=1) + self.pos_embed
        x = self._encode(x)
        if self.dist_token is not None:                  # :277-280: (class-token row, distillation-token row)
            return x[:, 0], x[:, 1]
        x = x[:, 0]
        if self.has_logits:
            x = torch.tanh(_Head.linear(x, self.pre_logits.fc))
        return x

    def forward_once(self, x):
        x = self.forward_features(x)
        if self.head_dist is not None:                   # :285-291: both heads in training, their average otherwise
            a = _Head.linear(x[0], self.head) if isinstance(self.head, nn.Linear) else x[0]
            b = _Head.linear(x[1], self.head_dist) if isinstance(self.head_dist, nn.Linear) else x[1]
            return (a, b) if self.training else (a + b) / 2
        return _Head.linear(x, self.head) if isinstance(self.head, nn.Linear) else x

    def _many(self, xs):
        """Siamese passes share every weight and no op mixes samples -> one batch."""
        sizes = [t.shape[0] for t in xs]
        y = self.forward_once(torch.cat(list(xs), 0))
        if isinstance(y, tuple):                         # distilled, training: one (x, x_dist) pair per input
            parts = [torch.split(v, sizes, 0) for v in y]
            return tuple(tuple(p[i] for p in parts) for i in range(len(sizes)))
        return tuple(torch.split(y, sizes, 0))

    def forward_twice(self, x1, x2):
        return self._many((x1, x2))

    def forward_thrice(self, x1, x2, x3):
        return self._many((x1, x2, x3))

    def forward(self, *args):
        n = len(args)
        if n == 1:
            return self.forward_once(args[0])
        elif n == 2:
            return self.forward_twice(args[0], args[1])
        elif n == 3:
            return self.forward_thrice(args[0], args[1], args[2])
        raise ValueError('Invalid input arguments! You got {} arguments.'.format(n))


class ScaleEmbedTransformer(nn.Module):
    """vit_model.py:321-549: four per-scale patch embeds (28/4, 56/8, 112/16, 224/32 -> 49 tokens each) with
    learned positional embeddings, a cls token and a designed-feature token, 12 blocks, `my_head` 768 -> 100.
    forward(*args): 1 -> forward_once, 2 -> (patches, designed) [NOT a pair], 4 -> pair, else ValueError."""
    _dm_first_write_blocks = True

    def __init__(self, img_size=224, patch_size=16, in_c=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, scales=[1, 1, 1, 1], qkv_bias=True, qk_scale=None, representation_size=None, distilled=False,
                 drop_ratio=0., attn_drop_ratio=0., drop_path_ratio=0., is_multiscale_embed=True, embed_layer=PatchEmbed,
                 is_feature_embed=True, feature_embed=FeatureEmbed, is_label_embed=False, norm_layer=None, act_layer=None,
                 numerics=None):
        super().__init__()
        _check_dim(embed_dim)
        self.numerics = _mode(numerics, self)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.scales = scales
        self.num_tokens = 2 if distilled else 1
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.is_multiscale_embed = is_multiscale_embed
        kw = {"numerics": self.numerics} if embed_layer is PatchEmbed else {}
        mk = lambda s, p: embed_layer(img_size=s, patch_size=p, in_c=in_c, embed_dim=embed_dim, **kw)
        self.patch_embed = mk(img_size, patch_size) if not is_multiscale_embed else None
        self.patch_embed0 = mk(28, 4) if is_multiscale_embed else None
        self.patch_embed1 = mk(56, 8) if is_multiscale_embed else None
        self.patch_embed2 = mk(112, 16) if is_multiscale_embed else None
        self.patch_embed3 = mk(224, 32) if is_multiscale_embed else None
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.is_label_embed = is_label_embed
        if is_label_embed:                       # vit_model.py:369-371
            self.label_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if distilled else None
        self.is_feature_embed = is_feature_embed
        self.feature_embed = feature_embed(feature_size=19, embed_dim=768) if is_feature_embed else None
        self.pos_embed0 = nn.Parameter(torch.zeros(1, 49, embed_dim))
        self.pos_embed1 = nn.Parameter(torch.zeros(1, 49, embed_dim))
        self.pos_embed2 = nn.Parameter(torch.zeros(1, 49, embed_dim))
        self.pos_embed3 = nn.Parameter(torch.zeros(1, 49, embed_dim))
        self.pos_embed_non_multiscale = nn.Parameter(torch.zeros(1, 196, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_ratio)
        dpr = [x.item() for x in torch.linspace(0, drop_path_ratio, depth)]      # stochastic depth decay rule (vit_model.py:229 / :387)
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop_ratio=drop_ratio, attn_drop_ratio=attn_drop_ratio, drop_path_ratio=dpr[i], norm_layer=norm_layer,
                  act_layer=act_layer, numerics=self.numerics) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        if representation_size and not distilled:
            self.has_logits = True
            self.num_features = representation_size
            self.pre_logits = nn.Sequential(OrderedDict([("fc", nn.Linear(embed_dim, representation_size)), ("act", nn.Tanh())]))
        else:
            self.has_logits = False
            self.pre_logits = nn.Identity()
        self.class_logits = nn.Linear(100, 11) if is_label_embed else nn.Identity()          # :408-412
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        if distilled:                                    # :417-418
            self.head_dist = nn.Linear(self.embed_dim, self.num_classes) if num_classes > 0 else nn.Identity()
        self.my_head = nn.Linear(768, 100)
        # :422-432 -- parameter containers; the arithmetic is in _class_head below
        self.my_class_head = nn.Sequential(nn.Linear(embed_dim, 100), nn.GELU(), nn.Dropout(0.3), nn.Linear(100, 100)) \
            if is_label_embed else nn.Identity()
        for pe in (self.pos_embed_non_multiscale, self.pos_embed0, self.pos_embed1, self.pos_embed2, self.pos_embed3, self.cls_token):
            nn.init.trunc_normal_(pe, std=0.02)
        if self.dist_token is not None:
            nn.init.trunc_normal_(self.dist_token, std=0.02)
        if is_label_embed:
            nn.init.trunc_normal_(self.label_token, std=0.02)
        self.apply(_init_vit_weights)

    def _class_head(self, x):
        """my_class_head on the label token's row: Linear -> GELU -> Dropout(0.3) -> Linear (vit_model.py:424-429); the two
        products run on the library, GELU / dropout on [rows, 100] values stay torch element-wise calls."""
        h = self.my_class_head
        y = torch.nn.functional.gelu(_Head.linear(x, h[0]))
        y = torch.nn.functional.dropout(y, h[2].p, self.training)
        return _Head.linear(y, h[3])

    def forward_features(self, x, designed_feature):
        if self.is_multiscale_embed:
            embeds = (self.patch_embed0, self.patch_embed1, self.patch_embed2, self.patch_embed3)
            poss = (self.pos_embed0, self.pos_embed1, self.pos_embed2, self.pos_embed3)
            x = torch.cat([(embeds[i](x[i]) + poss[i]) * self.scales[i] for i in range(4)], 1)
        else:
            x = self.patch_embed(x) + self.pos_embed_non_multiscale
        cls_token = self.cls_token.expand(x.shape[0], -1, -1)
        if self.dist_token is not None:
            # :485-486, :509-510: cls, the designed-feature tensor AS GIVEN (upstream does not embed it on this branch, so it must
            # already be [B, 1, embed_dim]; the usual [B, 1, 19] fails in torch.cat there and here), distillation token, patches;
            # the result is rows 0 and 1 of the normalised sequence -- the class row and the DESIGNED-FEATURE row, as upstream
            x = torch.cat((cls_token, designed_feature, self.dist_token.expand(x.shape[0], -1, -1), x), dim=1)
            x = self.blocks(self.pos_drop(x))
            x = ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)
            return x[:, 0], x[:, 1]
        if self.is_feature_embed:
            f = self.feature_embed(designed_feature)
            x = torch.cat((cls_token, f, x), dim=1)
            if self.is_label_embed:
                # vit_model.py:480-483 concatenates onto the ALREADY prefixed sequence: cls, label, designed, cls, designed,
                # patches (201 tokens) -- kept as the reference does it
                x = torch.cat((cls_token, self.label_token.expand(x.shape[0], -1, -1), f, x), dim=1)
        else:
            x = torch.cat((cls_token, x), dim=1)
        x = self.blocks(self.pos_drop(x))
        x = ops.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, torch.float32)
        y = _Head.linear(x[:, 0], self.my_head)
        if self.has_logits:
            y = torch.tanh(_Head.linear(y, self.pre_logits.fc))
        if self.is_label_embed:                  # :503-506 -> (embedding, class logits [*, 11], class features [*, 100])
            x_class = self._class_head(x[:, 1])
            return y, _Head.linear(x_class, self.class_logits), x_class
        return y

    def forward_once(self, x):
        x = self.forward_features(x, None)
        return _Head.linear(x, self.head) if isinstance(self.head, nn.Linear) else x

    def forward_twice(self, x1, x2):
        return self.forward_features(x1, x2)

    def forward_forice(self, x1, x2, x3, x4):
        """Pair form: both sides as one batch (shared weights, per-sample ops only)."""
        B = x2.shape[0]
        if self.is_multiscale_embed:
            xs = [torch.cat((x1[i], x3[i]), 0) for i in range(4)]
        else:
            xs = torch.cat((x1, x3), 0)
        y = self.forward_features(xs, torch.cat((x2, x4), 0))
        if isinstance(y, tuple):                         # label-token / distilled variants return tuples per side
            return tuple(v[:B] for v in y), tuple(v[B:] for v in y)
        return ops.split_halves(y) if (y.dim() == 2 and y.is_contiguous()) else (y[:B], y[B:])

    def forward(self, *args):
        n = len(args)
        if n == 1:
            return self.forward_once(args[0])
        elif n == 2:
            return self.forward_twice(args[0], args[1])
        elif n == 4:
            return self.forward_forice(args[0], args[1], args[2], args[3])
        raise ValueError('Invalid input arguments! You got {} arguments.'.format(n))


def _init_vit_weights(m):
    """vit_model.py:551-566."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.01)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode="fan_out")
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


def vit_base_patch_scales_224_in21k(num_classes: int = 21843, has_logits: bool = True, is_feature_embed=True,
                                    is_multiscale_embed=True, is_label_embed=False, numerics=None):
    return ScaleEmbedTransformer(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12,
                                 representation_size=768 if has_logits else None, num_classes=num_classes,
                                 is_feature_embed=is_feature_embed, is_multiscale_embed=is_multiscale_embed,
                                 is_label_embed=is_label_embed, numerics=numerics)


def _vit(patch, dim, depth, heads, num_classes, has_logits, numerics):
    return VisionTransformer(img_size=224, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads,
                             representation_size=dim if has_logits else None, num_classes=num_classes, numerics=numerics)


def vit_base_patch16_224_in21k(num_classes: int = 21843, has_logits: bool = True, numerics=None):
    return _vit(16, 768, 12, 12, num_classes, has_logits, numerics)


def vit_base_patch32_224_in21k(num_classes: int = 21843, has_logits: bool = True, numerics=None):
    return _vit(32, 768, 12, 12, num_classes, has_logits, numerics)


def vit_large_patch16_224_in21k(num_classes: int = 21843, has_logits: bool = True, numerics=None):
    return _vit(16, 1024, 24, 16, num_classes, has_logits, numerics)


def vit_large_patch32_224_in21k(num_classes: int = 21843, has_logits: bool = True, numerics=None):
    return _vit(32, 1024, 24, 16, num_classes, has_logits, numerics)


def vit_huge_patch14_224_in21k(num_classes: int = 21843, has_logits: bool = True, numerics=None):
    """vit_model.py:649-662.  Runs on the secondary kernels (generic attention for head dim 80 / 257 tokens, streamed LayerNorm
    for 1280 columns, element-wise patch extraction for the 14-pixel patches)."""
    return _vit(14, 1280, 32, 16, num_classes, has_logits, numerics)
