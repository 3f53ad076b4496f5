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


def test_patch_cols_cat_joins_adjacent_halves_without_a_copy():
    """The two sides of a pair batch as halves of ONE rows buffer (what deepmerge_amd.feed.PairFeed hands to PairTrainer.step) are
    joined as a view; unrelated buffers are copied; mismatched scales are refused."""
    import pytest
    g2, K, B = 64, 48, 3
    both = torch.arange(2 * B * g2 * K, dtype=torch.float32).reshape(2 * B * g2, K)
    whole = ops.PatchCols(both, 2 * B, 32, 4, 3)
    left, right = whole[:B], whole[B:]
    assert left.shape == (B, 3, 32, 32) and right.cols.data_ptr() == both.data_ptr() + B * g2 * K * 4
    joined = ops.PatchCols.cat(left, right)
    assert joined.batch == 2 * B and joined.cols.data_ptr() == both.data_ptr() and torch.equal(joined.cols, both)
    assert ops.cat_batch(left, right).cols.data_ptr() == both.data_ptr()
    other = ops.PatchCols(both[:B * g2].clone(), B, 32, 4, 3)
    copied = ops.PatchCols.cat(other, right)
    assert copied.cols.data_ptr() not in (both.data_ptr(), other.cols.data_ptr()) and torch.equal(copied.cols[:B * g2], other.cols)
    swapped = ops.PatchCols.cat(right, left)                      # adjacent the other way round: a copy, in the order asked for
    assert torch.equal(swapped.cols[:B * g2], right.cols) and torch.equal(swapped.cols[B * g2:], left.cols)
    with pytest.raises(ValueError):
        ops.PatchCols.cat(left, ops.PatchCols(torch.zeros(B * 16, K), B, 16, 4, 3))
    x, y = torch.zeros(2, 3, 8, 8), torch.ones(2, 3, 8, 8)
    assert torch.equal(ops.cat_batch(x, y), torch.cat((x, y), 0))


def test_pair_table_stack_and_feed_argument_checks():
    from deepmerge_amd.feed import PairFeed, PairTable
    import pytest
    B = 2
    mk = lambda off: PairTable(torch.full((B,), off, dtype=torch.int32), torch.zeros((B, 2), dtype=torch.int32), torch.full((B,), 20, dtype=torch.int32),
                               torch.full((B,), 30, dtype=torch.int32), torch.zeros((B, 15)), torch.tensor([1, 0]))
    t = PairTable.stack(mk(0), mk(1))
    assert t.tile_id.tolist() == [0, 0, 1, 1] and t.xy.shape == (2 * B, 2) and t.region.shape == (2 * B, 15) and t.flag.tolist() == [1, 0]
    with pytest.raises(ValueError):                               # tiles must live on the GPU: the feed has no CPU path
        PairFeed(torch.zeros((1, 3, 64, 64), dtype=torch.uint8), [32], B, 64)


def test_workload_runners_and_flops_import_without_a_gpu():
    """bench.py's `extras` runners live in the package (deepmerge_amd/workload.py, not tools/); importing them touches no device."""
    from deepmerge_amd import workload as W
    assert callable(W.config3) and callable(W.config4) and callable(W.config5) and callable(W.synth_batch)
    f = W.pair_step_flops([32, 64, 128, 256], 4, [3, 2, 1])
    assert abs(f / 1e9 - 84.47) < 0.01                           # the figure bench.py's line carries (gflop_per_pair_step)
    left, ld, right, rd, flag = W.synth_batch(4, [32, 64], 3, "cpu", 1)
    assert [tuple(t.shape) for t in left] == [(4, 3, 32, 32), (4, 3, 64, 64)] and ld.shape == (4, 1, 19) and flag.tolist() == [1, 0, 1, 0]
