"""CPU: oracle/s2former_aux.py (v4 aux heads, v5 designed-feature token) against golden vectors produced by the
unmodified reference modules with Dropout2d set to p = 0 (tests/golden/make_golden.py: gen_aux)."""
import numpy as np
import pytest
import torch

import recipe
from oracle import losses as OL
from oracle import s2former_aux as OA
from util import det_params, load_fx, model_inputs

RTOL = 2e-5
CFG = OA.v4_config(depth=(1, 1, 1))


def aux_params(tag):
    v5 = tag.startswith("v5")
    spec = OA.v5_param_spec(CFG) if v5 else OA.v4_param_spec(CFG)

    def index_for(k):
        if k.endswith("num_batches_tracked"):
            return np.zeros((), dtype=np.int64)
        cube = CFG.cube(int(k[len("blocks")]))
        return OA.relpos_index_v5(cube) if v5 else OA.relpos_index(cube)
    p = det_params(spec.items(), "", True, index_for)
    for k in p:
        if "running_" in k:
            p[k].requires_grad_(False)
    return spec, p


@pytest.mark.parametrize("tag", ["v4_111", "v5_111"])
def test_manifest_and_index(tag):
    fx = load_fx("model_aux.npz")
    spec, p = aux_params(tag)
    assert list(spec.keys()) == list(fx[tag + "/manifest_keys"])
    assert [",".join(map(str, s)) for s, _ in spec.values()] == list(fx[tag + "/manifest_shapes"])
    assert [d for _, d in spec.values()] == list(fx[tag + "/manifest_dtypes"])
    assert sum(int(np.prod(s)) for k, (s, d) in spec.items() if d == "float32" and "running_" not in k) == int(fx[tag + "/n_params"])
    for k in fx.files:
        if k.startswith(tag + "/index/"):
            assert np.array_equal(p[k[len(tag + "/index/"):]].numpy(), fx[k]), k
    if tag == "v4_111":
        assert OA.v4_model_name(CFG) == str(fx[tag + "/name"])


@pytest.mark.parametrize("tag", ["v4_111", "v5_111"])
def test_forward_backward_running_stats(tag):
    fx = load_fx("model_aux.npz")
    _, p = aux_params(tag)
    fwd = OA.v5_forward_once if tag.startswith("v5") else OA.v4_forward_once
    left, ld, right, rd, flag = model_inputs(tag, CFG.scales, 3, 4)
    stats = {}
    xa, a0a, a1a = fwd(p, left, ld, CFG, True, stats)
    for k, v in stats.items():                   # the reference runs side 1 then side 2 through the same BatchNorm modules
        p[k] = v
    xb, a0b, a1b = fwd(p, right, rd, CFG, True, stats)
    loss = (OL.contrastive_loss(xa, xb, flag, 1.0) + 0.1 * OL.contrastive_loss(a0a, a0b, flag, 1.0) +
            0.2 * OL.contrastive_loss(a1a, a1b, flag, 1.0))
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= 2e-5 * abs(float(fx[tag + "/loss"]))
    for n, v in (("out_a", xa), ("out_b", xb), ("aux0_a", a0a), ("aux0_b", a0b), ("aux1_a", a1a), ("aux1_b", a1b)):
        recipe.check_summary(f"{tag}/{n}", v.detach().numpy(), fx, RTOL)
    loss.backward()
    none = set(fx[tag + "/grad_none"])
    for k, v in p.items():
        if not v.dtype.is_floating_point or "running_" in k:
            continue
        if k in none:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
        else:
            # aux heads: a ReLU input within rounding of 0 may flip its mask between two fp32 summation orders
            rt = 2e-3 if k.startswith("aux") else 1e-4
            recipe.check_summary(tag + "/grad/" + k, v.grad.numpy(), fx, rt, k=512, atol=1e-6)
    for k, v in stats.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(fx[tag + "/after/" + k]) == 6      # 3 scales x 2 sides through one BatchNorm2d
        else:
            recipe.check_summary(tag + "/after/" + k, v.numpy(), fx, RTOL, k=768)
    for k, v in stats.items():
        p[k] = v
    with torch.no_grad():
        ev = fwd(p, left, ld, CFG, False)[0]
    recipe.check_summary(tag + "/eval_out", ev.numpy(), fx, RTOL)
