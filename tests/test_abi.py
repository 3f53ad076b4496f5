"""CPU: the C-ABI library loads and exports every symbol include/deepmerge_hip.h declares; the ctypes
binding mirrors the header; the product never imports the oracle or falls back to CPU."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    import __graft_entry__ as g
    from deepmerge_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib


def test_library_exports_every_declared_symbol(built_lib):
    declared = built_lib.declared_symbols()
    assert len(declared) >= 25
    handle = ctypes.CDLL(built_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(handle, s)]
    assert not missing, missing
    assert sorted(built_lib.SIGNATURES) == declared, "deepmerge_amd/_lib.py SIGNATURES out of sync with the header"
    nm = subprocess.run(["nm", "-D", "--defined-only", built_lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (dm_[a-z0-9_]+)", nm))
    assert set(declared) <= exported


def test_library_identity_calls(built_lib):
    lib = built_lib.lib()
    assert lib.dm_abi_version() == 6
    assert lib.dm_arch() == b"gfx950"
    assert lib.dm_gemm_workspace_bytes(built_lib.DM_NT, 1024, 768, 768) == 0
    assert lib.dm_gemm_workspace_bytes(built_lib.DM_TN, 768, 768, 16384) > 0
    assert lib.dm_attention_bwd_batch_chunks(64, 256, 12, 0) == 8 and lib.dm_attention_bwd_batch_chunks(2, 12, 12, 0) == 2 and lib.dm_attention_bwd_batch_chunks(64, 256, 12, 1) == 10
    assert lib.dm_layernorm_bwd_partial_floats(768) >= 2 * 768
    assert ctypes.sizeof(built_lib.DmGemmArgs) == 232   # == sizeof(DmGemmArgs) in C (gcc, LP64)
    assert ctypes.sizeof(built_lib.DmProfRow) == 96


def test_argument_validation_needs_no_gpu(built_lib):
    """Shape/dtype errors are reported through the status code + dm_last_error before any launch."""
    lib = built_lib.lib()
    a = built_lib.DmGemmArgs()
    a.M, a.N, a.K = 0, 8, 8
    assert lib.dm_gemm(ctypes.byref(a), None) == -1
    assert b"M,N,K" in lib.dm_last_error()
    assert lib.dm_attention_fwd(None, None, None, None, 1, 5000, 12, 64, 0.125, 0, None) == -1
    assert b"N <= 4096" in lib.dm_last_error()
    assert lib.dm_attention_fwd(None, None, None, None, 1, 300, 12, 80, 0.125, 0, None) == -1     # generic family: null pointers
    assert b"bad arguments" in lib.dm_last_error()
    assert lib.dm_layernorm_fwd(None, None, None, None, 0, None, None, 4, 8196, 1e-5, None) == -1
    assert b"<= 8192" in lib.dm_last_error()
    assert lib.dm_edge_similarity(1, 1, 1, None, 4, 1000, 1.0, None) == -6


def test_ops_fail_loudly_without_gpu_tensors(built_lib):
    import torch
    from deepmerge_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-5, torch.float32)
    from deepmerge_amd.nets.ShfitScaleFormer import Mlp
    with pytest.raises(RuntimeError):
        Mlp(8, 16)(torch.zeros(2, 8))


def test_missing_library_raises(monkeypatch, built_lib):
    monkeypatch.setattr(built_lib, "_lib", None)
    monkeypatch.setattr(built_lib, "LIB_PATH", "/nonexistent/libdeepmerge_hip.so")
    with pytest.raises(built_lib.DeepMergeLibraryError, match="no CPU fallback"):
        built_lib.lib()


def test_product_never_imports_the_oracle_or_reference():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deepmerge_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, re.M) or "/root/reference" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_drop_in_surface_names():
    """Class / kwarg surface of the mirrored modules."""
    import inspect
    from deepmerge_amd import Losses
    from deepmerge_amd.nets import ShfitScaleFormer as S
    for name in ("PatchEmbed", "Mlp", "FeatureEmbed", "CrossScaleAttention", "CrossScaleBlock", "ShfitScaleFormer_v3"):
        assert hasattr(S, name)
    sig = inspect.signature(S.ShfitScaleFormer_v3.__init__).parameters
    for kw in ("num_classes", "is_designed_feature_embedding", "FeatureEmbed", "PatchEmbed", "cube_size", "input_image_scales",
               "embed_dim", "depth", "num_heads", "mlp_ratio", "drop_path_ratio", "drop_ratio", "attn_drop_ratio", "norm_layer",
               "act_layer", "cuda"):
        assert kw in sig, kw
    fsig = list(inspect.signature(S.ShfitScaleFormer_v3.forward).parameters)
    assert fsig == ["self", "x1_patches", "x1_designed_features", "x2_patches", "x2_designed_features"]
    for m in ("extract_features_with_design_features", "extract_features", "forward_once", "forward_once_design_feature",
              "patch_embed", "backbone", "designed_feature_embed"):
        assert callable(getattr(S.ShfitScaleFormer_v3, m))
    assert list(inspect.signature(Losses.Loss.__init__).parameters) == ["self", "margin", "lamda", "belta"]
    assert list(inspect.signature(Losses.Loss.forward).parameters) == ["self", "positive", "negative", "flag", "size_average"]
    cube = [8, 8]
    net = S.ShfitScaleFormer_v3(depth=[1, 1, 1], cube_size=cube, input_image_scales=[32, 64, 128])
    assert cube == [3, 8, 8] and net.name == "S2Former_v3-3CH-3DP-SEF-111" and net.depth == [1, 1, 1]
    assert net.input_image_scales == [32, 64, 128]


def test_numerics_mode_switches_need_no_gpu(built_lib):
    """ops.check_numerics / set_fp32_products / fp32_products: validation and scoping of the numerics switches (pure host logic)."""
    from deepmerge_amd import ops
    assert ops.get_fp32_products() == "mfma_f32"
    with pytest.raises(ValueError):
        ops.check_numerics("fp16")
    with pytest.raises(ValueError):
        ops.set_fp32_products("tf32")
    with ops.fp32_products("bf16x3"):
        assert ops.get_fp32_products() == "bf16x3"
        with ops.fp32_products("mfma_f32"):
            assert ops.get_fp32_products() == "mfma_f32"
        assert ops.get_fp32_products() == "bf16x3"
    assert ops.get_fp32_products() == "mfma_f32"
    # validation has no side effect: the product kind belongs to the module
    assert ops.check_numerics("bf16x3") == "bf16x3" and ops.get_fp32_products() == "mfma_f32"
    assert ops.act_dtype("bf16x3") is __import__("torch").float32 and ops.act_dtype("bf16") is __import__("torch").bfloat16


def test_numerics_scope_belongs_to_the_module(built_lib):
    """A "bf16x3" module runs ITS forward (and the backward of the autograd nodes it created) under split-bf16 products and
    leaves the ambient kind alone -- two modules of different modes in one process do not share a setting."""
    import torch
    from deepmerge_amd import ops
    from deepmerge_amd.nets import ShfitScaleFormer as S

    seen = []

    class Probe(torch.nn.Module):
        def __init__(self, numerics):
            super().__init__()
            self.numerics = S._mode(numerics, self)

        def forward(self, x, fail=False):
            seen.append(ops.get_fp32_products())
            if fail:
                raise RuntimeError("boom")
            return x

    a, b = Probe("bf16x3"), Probe("fp32")
    assert ops.get_fp32_products() == "mfma_f32"            # building a bf16x3 module flips nothing
    a(1); b(2); a(3)
    assert seen == ["bf16x3", "mfma_f32", "bf16x3"] and ops.get_fp32_products() == "mfma_f32"
    with pytest.raises(RuntimeError):
        a(1, fail=True)
    assert ops.get_fp32_products() == "mfma_f32"            # the scope closes when forward raises
    with ops.fp32_products("bf16x3"):                       # the context manager stays as sugar for fp32 modules
        b(4)
    assert seen[-1] == "bf16x3"

    # entry points other than __call__ (PairTrainer -> forward_pair_batched) take the module's scope explicitly
    with ops.module_products(a):
        assert ops.get_fp32_products() == "bf16x3"
    with ops.module_products(b):
        assert ops.get_fp32_products() == "mfma_f32"
    with ops.fp32_products("bf16x3"), ops.module_products(b):      # an fp32 module leaves the ambient kind alone
        assert ops.get_fp32_products() == "bf16x3"
    assert ops.get_fp32_products() == "mfma_f32"

    # backward replays the kind its forward recorded, whatever is ambient when autograd runs it
    class Ctx:
        pass

    got = []

    @ops._replay_products
    def bwd(ctx, g):
        got.append(ops.get_fp32_products())
        return g

    c = Ctx()
    with ops.fp32_products("bf16x3"):
        ops._save_products(c)
    bwd(c, 0)
    bwd(Ctx(), 0)
    assert got == ["bf16x3", "mfma_f32"] and ops.get_fp32_products() == "mfma_f32"


def test_attention_routing_decisions_need_no_gpu(built_lib):
    """Which shapes the table-reading (bf16) and split-bf16 (bf16x3) attention entry points take is host logic; the refusals are
    reported before any launch (reference shapes: nets/ShfitScaleFormer.py:84-156 cubes (S, 8, 8), vit_model.py N = 197)."""
    lib = built_lib.lib()
    BF16, F32 = built_lib.DM_BF16, built_lib.DM_F32
    assert lib.dm_attention_relpos_inkernel(64, 256, 12, 64, 4, 8, 8, BF16) == 1
    assert lib.dm_attention_relpos_inkernel(64, 192, 12, 64, 3, 8, 8, BF16) == 1
    assert lib.dm_attention_relpos_inkernel(64, 256, 12, 64, 4, 8, 8, F32) == 0          # fp32 tensors: the split entry points
    assert lib.dm_attention_relpos_inkernel(64, 256, 12, 64, 4, 4, 16, BF16) == 0        # not an (S, 8, 8) cube
    assert lib.dm_attention_relpos_inkernel(64, 192, 12, 64, 4, 8, 8, BF16) == 0         # N != 64 S
    assert lib.dm_attention_relpos_inkernel(64, 128, 12, 64, 2, 8, 8, BF16) == 0         # two scales: N = 128 stays on the 16-row kernels
    assert lib.dm_attention_relpos_inkernel(2, 256, 12, 64, 4, 8, 8, BF16) == 0          # too little work for persistent workgroups
    assert lib.dm_attention_relpos_inkernel(64, 256, 12, 80, 4, 8, 8, BF16) == 0         # head dim
    assert lib.dm_attention_fwd_relpos(None, None, 4, 4, 16, None, None, 64, 256, 12, 64, 0.125, BF16, None) == -6
    assert b"shape not taken" in lib.dm_last_error()
    assert lib.dm_attention_fwd_relpos(None, None, 4, 8, 8, None, None, 64, 256, 12, 64, 0.125, BF16, None) == -1
    assert b"null pointer" in lib.dm_last_error()
    assert lib.dm_attention_split_ok(64, 256, 12, 64, 1, 4, 8, 8) == 1 and lib.dm_attention_split_ok(256, 197, 12, 64, 0, 0, 0, 0) == 1
    assert lib.dm_attention_split_ok(2, 256, 12, 64, 1, 4, 8, 8) == 1                    # (no persistent chunks: any batch)
    assert lib.dm_attention_split_ok(64, 193, 12, 64, 1, 3, 8, 8) == 0                   # v5's extra token: N != 64 S
    assert lib.dm_attention_split_ok(64, 64, 12, 64, 0, 0, 0, 0) == 0 and lib.dm_attention_split_ok(64, 257, 12, 64, 0, 0, 0, 0) == 0
    assert lib.dm_attention_split_bwd_chunks(64, 256, 12) == 10 and lib.dm_attention_split_bwd_chunks(3, 197, 12) == 3
    assert lib.dm_attention_split_fwd(None, None, None, None, 0, 0, 0, None, None, 64, 64, 12, 64, 0.125, None) == -6
    assert lib.dm_split_colsum_partial_floats(16384, 3072) > 0 and lib.dm_split_colsum_partial_floats(16384, 20) == 0


@pytest.mark.parametrize("env,B,expect", [
    ({}, 64, 1),
    ({"DM_ATTN_PIPE": "0"}, 64, 0),            # the 32-row backward kernels sit behind the pipelined family's gate
    ({"DM_ATTN_Q32": "2"}, 2, 0),              # forward forced for a small batch; the backward's B * H rule still refuses
    ({"DM_ATTN_Q32": "2", "DM_ATTN_Q32_BWD": "2", "DM_ATTN_PIPE": "2"}, 2, 1),
    ({"DM_ATTN_Q32_BWD": "3"}, 64, 0),         # new dQ only: dK / dV would need the dense rows
    ({"DM_ATTN_Q32_TABKV": "0"}, 64, 0),
])
def test_table_in_kernel_answer_covers_the_backward_pass(built_lib, env, B, expect):
    """`dm_attention_relpos_inkernel` licenses the caller to hold NO dense bias rows, so it must be the conjunction
    of the forward and the backward gates (their switches and B * H rules differ); a backward call without dense rows that the
    table-reading kernels do not take is refused instead of running bias-free kernels (nets/ShfitScaleFormer.py:123-128)."""
    code = ("import sys; sys.path.insert(0, %r); from deepmerge_amd import _lib; l = _lib.lib();"
            "ok = l.dm_attention_relpos_inkernel(%d, 256, 12, 64, 4, 8, 8, _lib.DM_BF16);"
            "print(ok, 0 if ok else l.dm_attention_bwd_relpos(16, 16, 4, 8, 8, None, None, 16, 16, 16, 16, 16, None, %d, 256, 12, 64,"
            " 0.125, _lib.DM_BF16, None))" % (ROOT, B, B))
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == expect
    if not expect:
        assert int(out[1]) == -6          # DM_ERR_UNSUPPORTED before any launch (no GPU here)


def test_relative_position_index_is_vouched_for_only_when_it_is_the_closed_form():
    """`CrossScaleAttention._index32()` tags the int32 index with its cube -- the licence for the kernels to form the bias from the table
    themselves -- only when the buffer equals the closed form of reference :139-156; a tampered buffer and v5's extended index get none."""
    import torch
    from deepmerge_amd.nets import ShfitScaleFormer as S
    a = S.CrossScaleAttention(dim=768, num_heads=12, cube_size=[4, 8, 8], qkv_bias=True)
    assert getattr(a._index32(), "_dm_cube", None) == (4, 8, 8)
    with torch.no_grad():
        a.relative_position_index[3, 5] += 1
    assert getattr(a._index32(), "_dm_cube", None) is None
    b = S.CrossScaleAttention_v5(dim=768, num_heads=12, cube_size=[3, 8, 8], qkv_bias=True)
    assert b._index32().shape == (193, 193) and getattr(b._index32(), "_dm_cube", None) is None
