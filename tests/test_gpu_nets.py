"""GPU: deepmerge_amd.Nets (MLP, FC of the reference's MNIST sandbox) against golden vectors from the
reference module; fp32, odd layer widths (250, 10) exercise the generic GEMM / column-sum paths."""
import numpy as np
import pytest
import torch

import recipe
from util import load_fx, tin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(m, prefix):
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(recipe.det_weight(prefix + k, v.shape)) for k, v in sd.items()})
    return m.to(DEV)


def test_mlp_and_fc_match_reference():
    from deepmerge_amd import Nets
    fx = load_fx("model_nets.npz")
    m = Nets.MLP()
    assert list(m.state_dict().keys()) == [str(k) for k in fx["mlp/keys"]]
    m = _load(m, "nets.mlp.")
    x = tin("nets.x", (37, 784), "unit").to(DEV).requires_grad_(True)
    a, b = m(x)
    (a.sum() * 1.5 + (b * b).sum()).backward()
    recipe.check_summary("mlp/fc3_map", a.detach().cpu().numpy(), fx, 1e-4)
    recipe.check_summary("mlp/fc2_map", b.detach().cpu().numpy(), fx, 1e-4)
    recipe.check_summary("mlp/dx", x.grad.cpu().numpy(), fx, 1e-4)
    for n, p in m.named_parameters():
        recipe.check_summary("mlp/grad/" + n, p.grad.cpu().numpy(), fx, 1e-4)
    f = _load(Nets.FC(), "nets.fc.")
    y = tin("nets.y", (37, 250), "normal").to(DEV).requires_grad_(True)
    o = f(y)
    (o * o).sum().backward()
    recipe.check_summary("fc/out", o.detach().cpu().numpy(), fx, 1e-4)
    recipe.check_summary("fc/dy", y.grad.cpu().numpy(), fx, 1e-4)
    for n, p in f.named_parameters():
        recipe.check_summary("fc/grad/" + n, p.grad.cpu().numpy(), fx, 1e-4)


def test_rnn_matches_reference():
    """Nets.RNN (4-layer bidirectional GRU + attention pooling, Nets.py:48-111) in eval mode against the reference's own forward /
    backward: GEMM input projections, dm_gru_cell steps, generic attention with one head of 160; then a training-mode pass
    (Dropout(0.5) live on the query) stays finite and differs."""
    from deepmerge_amd import Nets
    fx = load_fx("model_nets.npz")
    m = Nets.RNN()
    assert list(m.state_dict().keys()) == [str(k) for k in fx["rnn/keys"]]
    m = _load(m, "nets.rnn.").eval()
    x = tin("nets.rnn.x", (9, 28, 28), "unit").to(DEV).requires_grad_(True)
    o = m(x)
    (o * o).sum().backward()
    recipe.check_summary("rnn/out", o.detach().cpu().numpy(), fx, 2e-4)
    recipe.check_summary("rnn/dx", x.grad.cpu().numpy(), fx, 5e-4, atol=1e-7)
    for n, p in m.named_parameters():
        assert p.grad is not None, n
        recipe.check_summary("rnn/grad/" + n, p.grad.cpu().numpy(), fx, 5e-4, atol=1e-7)
    wn, we = recipe.worst_gradient("rnn/grad/", ((n, p.grad.cpu().numpy()) for n, p in m.named_parameters()), fx)
    print(f"Nets.RNN: worst gradient rel-L2 error {we:.2e} ({wn})")
    m.train()
    torch.manual_seed(3)
    o2 = m(x.detach())
    assert torch.isfinite(o2).all() and not torch.equal(o2, o.detach())



def test_networks_api_surface():
    """Networks.SpatiallyMmemorizedNetwork: constructor signature, arity dispatch, reduction conv as a GEMM, L2-normalised head."""
    from deepmerge_amd import Networks
    torch.manual_seed(0)
    base = torch.nn.Sequential(torch.nn.Conv2d(3, 64, 3, padding=1), torch.nn.ReLU())
    net = Networks.SpatiallyMmemorizedNetwork(base, "gap", 64, 64, 32).to(DEV)
    x1, x2, x3 = (torch.randn(5, 3, 16, 16, device=DEV) for _ in range(3))
    y = net(x1)
    f = base(x1).mean(dim=(2, 3))
    w, b = net.reduce_conv.weight.view(32, 64), net.reduce_conv.bias
    z = f @ w.T + b
    want = z / (z.norm(dim=1, keepdim=True) + 1e-6)
    assert y.shape == (5, 32) and torch.allclose(y, want, rtol=1e-4, atol=1e-6)
    a, b2 = net(x1, x2)
    assert torch.allclose(a, y, rtol=0, atol=0) and b2.shape == (5, 32)
    assert len(net(x1, x2, x3)) == 3
    y.sum().backward()
    assert net.reduce_conv.weight.grad is not None and base[0].weight.grad is not None
    with pytest.raises(ValueError):
        net(x1, x2, x3, x1)
    with pytest.raises(NotImplementedError):
        net(*([x1] * 6))
    with pytest.raises(NotImplementedError):
        Networks.SpatiallyMmemorizedNetwork("VGG16", "gap", 512, 512, 256)
