"""CPU: the oracle restatement (oracle/s2former.py, losses.py, adam.py) against the golden
vectors produced by the unmodified reference modules (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import recipe
from oracle import adam as OA
from oracle import losses as OL
from oracle import s2former as O
from util import MODEL_CASES, det_params, load_fx, model_inputs, model_params, tin

RTOL = 2e-5   # fp32 CPU vs fp32 CPU (different op fusion/threading): SURVEY measured 4e-7 / 2e-6


def test_relpos_index_closed_form_matches_reference():
    fx = load_fx("relpos_index.npz")
    cubes = [k.split("/")[1] for k in fx.files if k.startswith("index/")]
    assert len(cubes) >= 7
    for key in cubes:
        cube = [int(c) for c in key.split("x")]
        want = fx["index/" + key]
        got = O.relpos_index(cube)
        assert got.dtype == np.int64
        assert np.array_equal(got, want), key
        assert O.relpos_table_rows(cube) == int(fx["table_rows/" + key])


def test_relpos_index_known_answers():
    # SURVEY 8a M4 known answers
    i222 = O.relpos_index([2, 2, 2])
    assert i222[0].tolist() == [13, 12, 10, 9, 4, 3, 1, 0]
    assert i222[7].tolist() == [26, 25, 23, 22, 17, 16, 14, 13]
    i388 = O.relpos_index([3, 8, 8])
    assert i388[0, :5].tolist() == [562, 561, 560, 559, 558]
    assert i388[191, 0] == 1124 and i388[0, 191] == 0 and i388.max() == 1124


@pytest.mark.parametrize("tag,img,patch,in_c", [("pe32", 32, 4, 3), ("pe64", 64, 8, 3), ("pe128", 128, 16, 3), ("pe256c4", 256, 32, 4)])
def test_patch_embed(tag, img, patch, in_c):
    fx = load_fx("ops_s2former.npz")
    p = det_params([("proj.weight", ((768, in_c, patch, patch), "float32")), ("proj.bias", ((768,), "float32"))], tag + ".")
    x = tin(tag + ".x", (2, in_c, img, img), "unit").requires_grad_(True)
    y = O.patch_embed(p, "", x, patch)
    (y * tin(tag + ".go", y.shape)).sum().backward()
    recipe.check_summary(tag + "/y", y.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/dx", x.grad.numpy(), fx, RTOL)
    recipe.check_summary(tag + "/dw", p["proj.weight"].grad.numpy(), fx, RTOL)
    recipe.check_summary(tag + "/db", p["proj.bias"].grad.numpy(), fx, RTOL)


def test_patch_embed_rejects_wrong_size():
    p = det_params([("proj.weight", ((768, 3, 4, 4), "float32")), ("proj.bias", ((768,), "float32"))], "pe32.")
    with pytest.raises(AssertionError):
        O.patch_embed(p, "", torch.zeros(1, 3, 30, 30), 4)


@pytest.mark.parametrize("tag,cin,hid,xs", [("mlp", 768, 3072, (2, 12, 768)), ("mlp128", 128, 128, (64, 128))])
def test_mlp(tag, cin, hid, xs):
    fx = load_fx("ops_s2former.npz")
    spec = [("fc1.weight", ((hid, cin), "float32")), ("fc1.bias", ((hid,), "float32")),
            ("fc2.weight", ((cin, hid), "float32")), ("fc2.bias", ((cin,), "float32"))]
    p = det_params(spec, tag + ".")
    x = tin(tag + ".x", xs).requires_grad_(True)
    y = O.mlp(p, "", x)
    (y * tin(tag + ".go", y.shape)).sum().backward()
    recipe.check_summary(tag + "/y", y.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/dx", x.grad.numpy(), fx, RTOL)
    for k, _ in spec:
        recipe.check_summary(tag + "/d_" + k, p[k].grad.numpy(), fx, RTOL)


def test_feature_embed():
    fx = load_fx("ops_s2former.npz")
    spec = [("proj0.weight", ((768, 19, 1), "float32")), ("proj0.bias", ((768,), "float32")),
            ("proj1.weight", ((768, 768, 1), "float32")), ("proj1.bias", ((768,), "float32")),
            ("proj2.weight", ((768, 768, 1), "float32")), ("proj2.bias", ((768,), "float32"))]
    p = det_params(spec, "fe.")
    x = tin("fe.x", (3, 1, 19), "designed").requires_grad_(True)
    y = O.feature_embed(p, "", x)
    (y * tin("fe.go", y.shape)).sum().backward()
    recipe.check_summary("fe/y", y.detach().numpy(), fx, RTOL)
    recipe.check_summary("fe/dx", x.grad.numpy(), fx, RTOL)
    for k, _ in spec:
        recipe.check_summary("fe/d_" + k, p[k].grad.numpy(), fx, RTOL)


def attn_spec(cube):
    n = cube[0] * cube[1] * cube[2]
    return [("relative_position_bias_table", ((O.relpos_table_rows(cube), 12), "float32")),
            ("relative_position_index", ((n, n), "int64")),
            ("qkv.weight", ((2304, 768), "float32")), ("qkv.bias", ((2304,), "float32")),
            ("proj.weight", ((768, 768), "float32")), ("proj.bias", ((768,), "float32"))]


@pytest.mark.parametrize("cube", [[3, 2, 2], [3, 4, 4], [3, 8, 8], [4, 8, 8], [4, 4, 4], [4, 2, 2]])
def test_cross_scale_attention(cube):
    fx = load_fx("ops_s2former.npz")
    tag = "attn" + "x".join(map(str, cube))
    n = cube[0] * cube[1] * cube[2]
    p = det_params(attn_spec(cube), tag + ".", index_for=lambda k: O.relpos_index(cube))
    x = tin(tag + ".x", (2, n, 768)).requires_grad_(True)
    y = O.cross_scale_attention(p, "", x, 12)
    (y * tin(tag + ".go", y.shape)).sum().backward()
    recipe.check_summary(tag + "/y", y.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/dx", x.grad.numpy(), fx, RTOL)
    for k, (_, dt) in attn_spec(cube):
        if dt == "float32":
            recipe.check_summary(tag + "/d_" + k, p[k].grad.numpy(), fx, RTOL)


@pytest.mark.parametrize("cube", [[3, 4, 4], [4, 8, 8]])
def test_cross_scale_block(cube):
    fx = load_fx("ops_s2former.npz")
    tag = "block" + "x".join(map(str, cube))
    n = cube[0] * cube[1] * cube[2]
    spec = [("norm1.weight", ((768,), "float32")), ("norm1.bias", ((768,), "float32"))]
    spec += [("attn." + k, v) for k, v in attn_spec(cube)]
    spec += [("norm2.weight", ((768,), "float32")), ("norm2.bias", ((768,), "float32")),
             ("mlp.fc1.weight", ((3072, 768), "float32")), ("mlp.fc1.bias", ((3072,), "float32")),
             ("mlp.fc2.weight", ((768, 3072), "float32")), ("mlp.fc2.bias", ((768,), "float32"))]
    p = det_params(spec, tag + ".", index_for=lambda k: O.relpos_index(cube))
    x = tin(tag + ".x", (2, n, 768)).requires_grad_(True)
    y = O.cross_scale_block(p, "", x, 12, 1e-5)
    (y * tin(tag + ".go", y.shape)).sum().backward()
    recipe.check_summary(tag + "/y", y.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/dx", x.grad.numpy(), fx, RTOL)
    for k, (_, dt) in spec:
        if dt == "float32":
            recipe.check_summary(tag + "/d_" + k, p[k].grad.numpy(), fx, RTOL)


def test_contrastive_loss_both_branches_and_flag_dtypes():
    fx = load_fx("ops_s2former.npz")
    assert int(fx["loss/n_below_margin"]) >= 3 and int(fx["loss/n_above_margin"]) >= 3
    a0 = torch.from_numpy(fx["loss/a"]); b0 = torch.from_numpy(fx["loss/b"])
    flag = torch.from_numpy(fx["loss/flag"])
    for tag, fl in (("loss_i64", flag), ("loss_f32", flag.float())):
        a = a0.clone().requires_grad_(True); b = b0.clone().requires_grad_(True)
        val = OL.contrastive_loss(a, b, fl, 1.0)
        val.backward()
        assert abs(val.item() - float(fx[tag + "/value"])) <= 1e-6 * abs(float(fx[tag + "/value"]))
        recipe.check_summary(tag + "/da", a.grad.numpy(), fx, 1e-6)
        recipe.check_summary(tag + "/db", b.grad.numpy(), fx, 1e-6)
    ll = tin("loss.ll", (16, 11)); rl = tin("loss.rl", (16, 11))
    lt = torch.arange(16) % 11; rt = (torch.arange(16) * 3) % 11
    assert abs(OL.multi_loss(a0, b0, flag, ll, lt, rl, rt, 1.0).item() - float(fx["multiloss/value"])) < 1e-5
    assert abs(OL.class_loss(ll, lt, rl, rt).item() - float(fx["classloss/value"])) < 1e-5


@pytest.mark.parametrize("tag", ["v3_3s3c_111", "v3_4s4c_321", "v3_3s3c_642", "v3_4s4c_642"])
def test_whole_model_forward_backward_adam(tag):
    fx = load_fx("model_v3.npz")
    cfg = MODEL_CASES[tag]
    spec = O.param_spec(cfg)
    # --- state_dict manifest: keys, shapes, dtypes, name, parameter count
    assert list(spec.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, s)) for s, _ in spec.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    assert [dt for _, dt in spec.values()] == [str(s) for s in fx[tag + "/manifest_dtypes"]]
    assert O.model_name(cfg) == str(fx[tag + "/name"])
    n_params = sum(int(np.prod(s)) for s, dt in spec.values() if dt == "float32")
    assert n_params == int(fx[tag + "/n_params"])
    # --- one training step
    p = model_params(cfg)
    left, ld, right, rd, flag = model_inputs(tag, cfg.scales, cfg.in_c, 4)
    fparams = {k: v for k, v in p.items() if v.dtype.is_floating_point}
    m = {k: torch.zeros_like(v) for k, v in fparams.items()}
    v2 = {k: torch.zeros_like(v) for k, v in fparams.items()}
    watch = [k[len(tag + "/adam1/"):-len("/shape")] for k in fx.files
             if k.startswith(tag + "/adam1/") and k.endswith("/shape")]
    assert len(watch) >= 6
    for step in range(1, 4):
        for t in fparams.values():
            t.grad = None
        fa, fb = O.forward_pair(p, left, ld, right, rd, cfg)
        loss = OL.contrastive_loss(fa, fb, flag, 1.0)
        loss.backward()
        want_loss = float(fx[tag + f"/loss_step{step}"])
        assert abs(loss.item() - want_loss) <= 2e-5 * abs(want_loss), (step, loss.item(), want_loss)
        if step == 1:
            recipe.check_summary(tag + "/out_a", fa.detach().numpy(), fx, RTOL)
            recipe.check_summary(tag + "/out_b", fb.detach().numpy(), fx, RTOL)
            none = sorted(str(s) for s in fx[tag + "/grad_none"])
            assert sorted(k for k, t in fparams.items() if t.grad is None) == none
            assert none == ["final_features.bias", "final_features.weight", "head.bias", "head.weight"]
            for k, t in fparams.items():
                if t.grad is not None:
                    recipe.check_summary(tag + "/grad/" + k, t.grad.numpy(), fx, 5e-5, k=1024)
            with torch.no_grad():
                ev = O.forward_once(p, left, ld, cfg)
            assert bool(fx[tag + "/eval_equals_train"]) and torch.equal(ev, fa.detach())
        with torch.no_grad():
            for k, t in fparams.items():
                if t.grad is not None:   # torch.optim.Adam skips params whose grad is None
                    OA.adam_step(t, t.grad, m[k], v2[k], step, lr=1e-4)
        if step in (1, 3):
            for k in watch:
                # Adam normalises by sqrt(v): entries whose gradient is at fp32-noise level move by
                # O(lr) in a noise-determined direction, so the post-step pin is looser than the grads'.
                recipe.check_summary(tag + f"/adam{step}/" + k, p[k].detach().numpy(), fx, 1e-4, k=1024)


def test_flops_formula_matches_baseline_table():
    # BASELINE.md section 3
    assert abs(O.flops_forward_per_sample(MODEL_CASES["v3_3s3c_642"]) / 1e9 - 20.18) < 0.02
    assert abs(O.flops_forward_per_sample(MODEL_CASES["v3_4s4c_321"]) / 1e9 - 14.08) < 0.02
    big = O.S2Config(scales=(32, 64, 128, 256), in_c=4, depth=(6, 4, 2))
    assert abs(O.flops_forward_per_sample(big) / 1e9 - 27.62) < 0.02


@pytest.mark.parametrize("tag,variant,depth", [("v1_d2", "v1", 2), ("v2", "v2", 12)])
def test_single_stage_variants(tag, variant, depth):
    fx = load_fx("model_variants.npz")
    spec = O.v12_param_spec(variant, depth)
    assert list(spec.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, s)) for s, _ in spec.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    p = det_params(spec.items(), index_for=lambda k: O.relpos_index([4, 7, 7]))
    left, ld, right, rd, flag = model_inputs(tag, O.V12_SCALES, 3, 4)
    fa, fb = O.v12_forward_once(p, variant, left, ld, depth), O.v12_forward_once(p, variant, right, rd, depth)
    loss = OL.contrastive_loss(fa, fb, flag, 1.0)
    loss.backward()
    recipe.check_summary(tag + "/out_a", fa.detach().numpy(), fx, RTOL)
    recipe.check_summary(tag + "/out_b", fb.detach().numpy(), fx, RTOL)
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= 2e-5 * abs(float(fx[tag + "/loss"]))
    for k, (_, dt) in spec.items():
        if dt == "float32" and p[k].grad is not None:
            # norm.bias and the output bias cancel exactly between the two Siamese sides here (their expected
            # gradient is rounding noise ~1e-7), hence the absolute floor
            recipe.check_summary(tag + "/grad/" + k, p[k].grad.numpy(), fx, 1e-4, k=512, atol=1e-6)
    assert sorted(k for k, (_, dt) in spec.items() if dt == "float32" and p[k].grad is None) == sorted(str(s) for s in fx[tag + "/grad_none"])


def test_v6_designed_features_only():
    fx = load_fx("model_variants.npz")
    spec = O.v6_param_spec()
    assert list(spec.keys()) == [str(k) for k in fx["v6/manifest_keys"]]
    p = det_params(spec.items())
    da, db = tin("v6.da", (4, 1, 19), "designed"), tin("v6.db", (4, 1, 19), "designed")
    db[1] = da[1] * 1.05
    fa, fb = O.v6_forward_once(p, da), O.v6_forward_once(p, db)
    loss = OL.contrastive_loss(fa, fb, torch.tensor([1, 0, 1, 0]), 1.0)
    loss.backward()
    recipe.check_summary("v6/out_a", fa.detach().numpy(), fx, RTOL)
    assert abs(loss.item() - float(fx["v6/loss"])) <= 2e-5 * abs(float(fx["v6/loss"]))
    for k in spec:
        if p[k].grad is not None:
            recipe.check_summary("v6/grad/" + k, p[k].grad.numpy(), fx, 1e-4, k=512, atol=1e-8)
