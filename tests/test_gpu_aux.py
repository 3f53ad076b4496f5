"""GPU: ShfitScaleFormer_v4 / _v5 (aux heads, designed-feature token) against golden vectors from
the reference modules.  Dropout2d(p=0.3) of the aux heads is set to p = 0 on both sides (its RNG stream cannot be
reproduced by an independent implementation); BatchNorm2d batch statistics and running updates are pinned."""
import numpy as np
import pytest
import torch

import recipe
from test_gpu_modules import DEV, GATE, S2F, load_recipe_weights
from util import load_fx, model_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,cls", [("v4_111", "ShfitScaleFormer_v4"), ("v5_111", "ShfitScaleFormer_v5")])
def test_aux_variants_parity_fp32(tag, cls):
    from deepmerge_amd.Losses import Loss
    fx = load_fx("model_aux.npz")
    net = getattr(S2F(), cls)(cube_size=[8, 8], input_image_scales=[32, 64, 128], depth=[1, 1, 1], numerics="fp32")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in fx[tag + "/manifest_keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in fx[tag + "/manifest_shapes"]]
    for k, v in sd.items():
        if k.endswith("relative_position_index"):
            assert np.array_equal(v.numpy(), fx[tag + "/index/" + k]), k
    if tag == "v4_111":
        assert net.name == str(fx[tag + "/name"])
    else:
        assert not hasattr(net, "name")
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = load_recipe_weights(net).to(DEV).train()
    left, ld, right, rd, flag = model_inputs(tag, (32, 64, 128), 3, 4)
    left = [t.to(DEV) for t in left]; right = [t.to(DEV) for t in right]
    (xa, a0a, a1a), (xb, a0b, a1b) = net(left, ld.to(DEV), right, rd.to(DEV))
    crit = Loss(1.0, 0.1, 0)
    f = flag.to(DEV)
    loss = crit(xa, xb, f) + 0.1 * crit(a0a, a0b, f) + 0.2 * crit(a1a, a1b, f)
    loss.backward()
    for n, v in (("out_a", xa), ("out_b", xb), ("aux0_a", a0a), ("aux0_b", a0b), ("aux1_a", a1a), ("aux1_b", a1b)):
        recipe.check_summary(f"{tag}/{n}", v.detach().cpu().numpy(), fx, GATE)
    assert abs(loss.item() - float(fx[tag + "/loss"])) <= GATE * abs(float(fx[tag + "/loss"]))
    assert sorted(n for n, p in net.named_parameters() if p.grad is None) == sorted(str(s) for s in fx[tag + "/grad_none"])
    for n, p in net.named_parameters():
        if p.grad is not None:
            # aux heads: a ReLU input within rounding of 0 may flip its mask between two summation orders
            recipe.check_summary(tag + "/grad/" + n, p.grad.cpu().numpy(), fx, 5e-3 if n.startswith("aux") else GATE, k=512, atol=1e-6)
    for k, v in net.state_dict().items():
        if "running_" in k:
            recipe.check_summary(tag + "/after/" + k, v.cpu().numpy(), fx, GATE, k=768)
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(fx[tag + "/after/" + k])
    net.eval()
    with torch.no_grad():
        ev = net(left, ld.to(DEV))
    recipe.check_summary(tag + "/eval_out", ev.cpu().numpy(), fx, GATE)


def test_v5_bf16_runs_and_is_close():
    fx = load_fx("model_aux.npz")
    net = S2F().ShfitScaleFormer_v5(cube_size=[8, 8], input_image_scales=[32, 64, 128], depth=[1, 1, 1], numerics="bf16")
    net = load_recipe_weights(net).to(DEV).eval()
    left, ld, _, _, _ = model_inputs("v5_111", (32, 64, 128), 3, 4)
    with torch.no_grad():
        ev = net([t.to(DEV) for t in left], ld.to(DEV))
    l2, _ = recipe.summary_error("v5_111/eval_out", ev.float().cpu().numpy(), fx)
    assert l2 < 0.05, l2
