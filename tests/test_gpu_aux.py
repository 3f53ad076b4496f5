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


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("samples,rows,C", [(6, 49, 768), (3, 9, 256), (1, 1, 64)])
def test_batchnorm_relu_dropout_kernel_vs_torch(training, samples, rows, C):
    """dm_batchnorm_fwd / _bwd through ops.BatchNormReluFn against torch's own BatchNorm2d -> ReLU -> channel mask in float64:
    output, running statistics, dx, dgamma, dbeta; Dropout2d semantics (whole channels of a sample are zeroed or scaled)."""
    from deepmerge_amd import ops
    g = torch.Generator().manual_seed(samples * 100 + rows)
    M = samples * rows
    x = torch.randn(M, C, generator=g) * 1.7 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    mask = (torch.rand(samples, C, generator=g) > 0.3).float() / 0.7 if training else None
    go = torch.randn(M, C, generator=g)
    if M == 1 and training:        # torch: "Expected more than 1 value per channel when training"; same contract here
        with pytest.raises(ValueError):
            ops.BatchNormReluFn.apply(x.to(DEV), gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), None, rows, 1e-5, 0.1, True, True)
        return
    # torch reference on [samples, C, rows, 1] in float64
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(training)
    xr = x.double().view(samples, rows, C).permute(0, 2, 1).unsqueeze(-1).contiguous().requires_grad_(True)
    yr = torch.relu(bn(xr))
    if mask is not None:
        yr = yr * mask.double()[:, :, None, None]
    (yr * go.double().view(samples, rows, C).permute(0, 2, 1).unsqueeze(-1)).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y = ops.BatchNormReluFn.apply(xd, gd, bd, rmd, rvd, None if mask is None else mask.to(DEV), rows, 1e-5, 0.1, training, True)
    (y * go.to(DEV)).sum().backward()
    back = lambda t: t.detach().squeeze(-1).permute(0, 2, 1).reshape(M, C)
    np.testing.assert_allclose(y.detach().cpu().double().numpy(), back(yr).numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(xd.grad.cpu().double().numpy(), back(xr.grad).numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(gd.grad.cpu().double().numpy(), bn.weight.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(bd.grad.cpu().double().numpy(), bn.bias.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rmd.cpu().double().numpy(), bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rvd.cpu().double().numpy(), bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    if mask is not None:
        dropped = (mask == 0)[:, None, :].expand(samples, rows, C).reshape(M, C)
        assert float(y.detach().cpu()[dropped].abs().max()) == 0.0
    # the same without the ReLU (plain BatchNorm2d): the backward must not gate on y > 0
    bn.zero_grad(); xr2 = x.double().view(samples, rows, C).permute(0, 2, 1).unsqueeze(-1).contiguous().requires_grad_(True)
    with torch.no_grad():
        bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    # (a contiguous upstream gradient: torch's CPU batch-norm backward mishandles a permuted grad_output view -- 3.2 off against autograd of the formula)
    (bn(xr2) * go.double().view(samples, rows, C).permute(0, 2, 1).unsqueeze(-1).contiguous()).sum().backward()
    xd2 = x.to(DEV).requires_grad_(True)
    y2 = ops.BatchNormReluFn.apply(xd2, gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), None, rows, 1e-5, 0.1, training, False)
    (y2 * go.to(DEV)).sum().backward()
    np.testing.assert_allclose(xd2.grad.cpu().double().numpy(), back(xr2.grad).numpy(), rtol=2e-4, atol=2e-5)


def test_aux_head_dropout_is_live_in_training():
    """p = 0.3 (the reference default): in training every (sample, channel) of the BatchNorm output is either dropped or scaled
    by 1/0.7, so two forward passes differ; in eval the head is deterministic."""
    net = S2F().ShfitScaleFormer_v4(cube_size=[8, 8], input_image_scales=[32, 64, 128], depth=[1, 1, 1], numerics="fp32")
    net = load_recipe_weights(net).to(DEV).train()
    left, ld, right, rd, _ = model_inputs("v4_111", (32, 64, 128), 3, 4)
    args = ([t.to(DEV) for t in left], ld.to(DEV), [t.to(DEV) for t in right], rd.to(DEV))
    with torch.no_grad():
        (x1, a1, _), _ = net(*args)
        (x2, a2, _), _ = net(*args)
    assert torch.equal(x1, x2) and not torch.equal(a1, a2)
    assert int(net.aux0.aux[1].num_batches_tracked) == 2 * 2 * 3     # 2 passes x 2 sides x 3 scales (one BatchNorm call each)
