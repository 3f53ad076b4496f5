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
    with pytest.raises(NotImplementedError):
        Nets.RNN()
