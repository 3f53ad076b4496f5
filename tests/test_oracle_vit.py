"""CPU: oracle/vit.py (restatement of vit_model.py's pair encoders) against golden vectors from the reference."""
import numpy as np
import pytest
import torch

import recipe
from oracle import losses as OL
from oracle import vit as OV
from util import det_params, load_fx, tin

RTOL = 5e-5


def vit_inputs(tag):
    x1 = tin(tag + ".x1", (2, 3, 224, 224), "unit"); x2 = tin(tag + ".x2", (2, 3, 224, 224), "unit")
    x2[1] = x1[1] * 0.8 + 0.2 * x2[1]
    return x1, x2, torch.tensor([1, 0], dtype=torch.int64)


def scale_inputs(tag):
    sizes = (28, 56, 112, 224)
    xa = [tin(f"{tag}.xa{i}", (2, 3, s, s), "unit") for i, s in enumerate(sizes)]
    xb = [tin(f"{tag}.xb{i}", (2, 3, s, s), "unit") for i, s in enumerate(sizes)]
    fa = tin(tag + ".fa", (2, 1, 19), "designed"); fb = tin(tag + ".fb", (2, 1, 19), "designed")
    for i in range(4):
        xb[i][1] = xa[i][1] * 0.8 + 0.2 * xb[i][1]
    fb[1] = fa[1] * 1.2
    return xa, fa, xb, fb, torch.tensor([1, 0], dtype=torch.int64)


VITH = dict(patch=14, dim=1280, heads=16, hidden=5120)       # vit_model.py:649-662 geometry (head dim 80, 257 tokens)


@pytest.mark.parametrize("tag,depth", [("vitb16_d2", 2), ("vitb16_d12", 12), ("vith14_d2", 2)])
def test_vision_transformer_pair(tag, depth):
    fx = load_fx("model_vit.npz")
    cfg = OV.VitConfig(depth=depth, num_classes=100, **(VITH if tag.startswith("vith") else {}))
    spec = OV.vit_param_spec(cfg)
    assert list(spec.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, s)) for s, _ in spec.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    assert sum(int(np.prod(s)) for s, _ in spec.values()) == int(fx[tag + "/n_params"])
    p = det_params(spec.items())
    x1, x2, flag = vit_inputs(tag)
    ya, yb = OV.vit_forward_pair(p, x1, x2, cfg)
    loss = OL.contrastive_loss(ya, yb, flag, 1.0)
    loss.backward()
    recipe.check_summary(tag + "/out_a", ya.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/out_b", yb.detach().numpy(), fx, RTOL)
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= 5e-5 * abs(float(fx[tag + "/loss"]))
    for k in spec:
        recipe.check_summary(tag + "/grad/" + k, p[k].grad.numpy(), fx, 1e-4, k=512, atol=1e-9)
    assert bool(fx[tag + "/single_equals_pair"]) and bool(fx[tag + "/four_args_raise"])
    if depth == 12:
        assert int(fx[tag + "/n_params"]) == 85875556


@pytest.mark.parametrize("tag,depth", [("vitscale_d2", 2), ("vitscale_d12", 12)])
def test_scale_embed_transformer_pair(tag, depth):
    fx = load_fx("model_vit.npz")
    spec = OV.scale_param_spec(depth=depth, num_classes=512)
    assert list(spec.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, s)) for s, _ in spec.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    p = det_params(spec.items())
    xa, fa, xb, fb, flag = scale_inputs(tag)
    ya, yb = OV.scale_forward_pair(p, xa, fa, xb, fb, depth=depth)
    loss = OL.contrastive_loss(ya, yb, flag, 1.0)
    loss.backward()
    recipe.check_summary(tag + "/out_a", ya.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/out_b", yb.detach().numpy(), fx, RTOL)
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= 5e-5 * abs(float(fx[tag + "/loss"]))
    none = sorted(str(s) for s in fx[tag + "/grad_none"])
    assert none == ["head.bias", "head.weight", "pos_embed_non_multiscale"]
    assert sorted(k for k in spec if p[k].grad is None) == none
    for k in spec:
        if p[k].grad is not None:
            recipe.check_summary(tag + "/grad/" + k, p[k].grad.numpy(), fx, 1e-4, k=512, atol=1e-9)
    assert bool(fx[tag + "/two_args_equals_left"])
    if depth == 12:
        assert int(fx[tag + "/n_params"]) == 90161508


def label_inputs(tag="vitscale_label_d2"):
    xa, fa, xb, fb, flag = scale_inputs(tag)
    return xa, fa, xb, fb, flag, torch.tensor([3, 7]), torch.tensor([0, 10])


def label_loss(ra, rb, flag, la, lb, contrastive):
    """The fixture's loss (tests/golden/make_golden.py): every value of the 3-tuple result carries gradient."""
    ce = torch.nn.functional.cross_entropy
    return contrastive(ra[0], rb[0], flag) + ce(ra[1], la) + ce(rb[1], lb) + 0.01 * (ra[2].pow(2).sum() + rb[2].pow(2).sum())


def test_scale_embed_transformer_label_token():
    """is_label_embed=True (vit_model.py:369-371, :408-432, :480-483, :503-506): 201-token sequence, 3-tuple result."""
    tag = "vitscale_label_d2"
    fx = load_fx("model_vit.npz")
    spec = OV.scale_param_spec(depth=2, num_classes=512, label=True)
    assert list(spec.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, s)) for s, _ in spec.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    p = det_params(spec.items())
    xa, fa, xb, fb, flag, la, lb = label_inputs(tag)
    ra = OV.scale_forward_features(p, xa, fa, depth=2, label=True)
    rb = OV.scale_forward_features(p, xb, fb, depth=2, label=True)
    loss = label_loss(ra, rb, flag, la, lb, lambda a, b, f: OL.contrastive_loss(a, b, f, 1.0))
    loss.backward()
    for side, r in (("a", ra), ("b", rb)):
        recipe.check_summary(f"{tag}/out_{side}", r[0].detach().numpy(), fx, RTOL)
        recipe.check_summary(f"{tag}/logits_{side}", r[1].detach().numpy(), fx, RTOL)
        recipe.check_summary(f"{tag}/class_{side}", r[2].detach().numpy(), fx, RTOL)
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= 5e-5 * abs(float(fx[tag + "/loss"]))
    none = sorted(str(s) for s in fx[tag + "/grad_none"])
    assert sorted(k for k in spec if p[k].grad is None) == none
    for k in spec:
        if p[k].grad is not None:
            recipe.check_summary(tag + "/grad/" + k, p[k].grad.numpy(), fx, 1e-4, k=512, atol=1e-9)
    assert bool(fx[tag + "/two_args_equals_left"])


def test_vision_transformer_distilled():
    """distilled=True (vit_model.py:217, :225, :250-253, :270, :277-291): (x, x_dist) per input in training mode, the average in eval."""
    tag = "vitb16_dist_d2"
    fx = load_fx("model_vit.npz")
    cfg = OV.VitConfig(depth=2, num_classes=100)
    spec = OV.vit_param_spec(cfg, distilled=True)
    assert list(spec.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, s)) for s, _ in spec.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    p = det_params(spec.items())
    x1, x2, flag = vit_inputs(tag)
    ya, da = OV.vit_forward_once_distilled(p, x1, cfg)
    yb, db = OV.vit_forward_once_distilled(p, x2, cfg)
    loss = OL.contrastive_loss(ya, yb, flag, 1.0) + OL.contrastive_loss(da, db, flag, 1.0)
    loss.backward()
    for key, v in (("out_a", ya), ("out_b", yb), ("dist_a", da), ("dist_b", db)):
        recipe.check_summary(f"{tag}/{key}", v.detach().numpy(), fx, RTOL)
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= 5e-5 * abs(float(fx[tag + "/loss"]))
    for k in spec:
        recipe.check_summary(tag + "/grad/" + k, p[k].grad.numpy(), fx, 1e-4, k=512, atol=1e-9)
    with torch.no_grad():
        ev = OV.vit_forward_once_distilled(p, x1, cfg, training=False)
    recipe.check_summary(tag + "/eval_a", ev.numpy(), fx, RTOL)
