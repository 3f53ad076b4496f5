"""Host-side helpers of deepmerge_amd.ops that need no GPU: the tagging of the Siamese encoder's stacked output and the unit incoming gradient."""
import torch

from deepmerge_amd import ops


def test_split_halves_tags_and_stacked_halves_recognises_only_the_tagged_pair():
    f = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    a, b = ops.split_halves(f)
    assert torch.equal(a, f[:3]) and torch.equal(b, f[3:])
    assert ops.stacked_halves(a, b) is f
    assert ops.stacked_halves(b, a) is None                      # wrong order
    assert ops.stacked_halves(f[:3], f[3:]) is None              # untagged slices of the same matrix
    g = f.clone()
    c, d = ops.split_halves(g)
    assert ops.stacked_halves(a, d) is None                      # halves of two different matrices
    t = torch.arange(24, dtype=torch.float32).reshape(6, 4)[:, ::2]      # not contiguous: the kernel call would need a copy
    p, q = ops.split_halves(t)
    assert ops.stacked_halves(p, q) is None


def test_unit_grad_is_one_tensor_per_device_and_recognised_by_identity():
    u = ops.unit_grad("cpu")
    assert u.dim() == 0 and float(u) == 1.0 and ops.unit_grad("cpu") is u
    assert ops.is_unit_grad(u)
    assert not ops.is_unit_grad(torch.ones(()))                   # equal value, another tensor
    assert not ops.is_unit_grad(torch.ones(1))
