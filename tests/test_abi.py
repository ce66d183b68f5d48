"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and
exports every symbol include/plnlp_hip.h declares; argument validation paths
that do not launch anything; product ops refuse CPU tensors loudly."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from plnlp_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "plnlp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plnlp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from plnlp_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in plnlp_hip.h but not exported"
    # and the Python binding types exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared


def test_abi_version_and_error_strings(lib):
    header = open(os.path.join(ROOT, "include", "plnlp_hip.h")).read()
    assert lib.plnlp_abi_version() == int(re.search(r"#define PLNLP_ABI_VERSION (\d+)", header).group(1)) == 12
    assert lib.plnlp_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert lib.plnlp_error_string(code).startswith(b"plnlp:")


def test_argument_validation_without_launch(lib):
    # NULL pointers / bad shapes are rejected before any HIP call
    assert lib.plnlp_csr_aggregate_f32(None, None, None, None, None, None, None, None, None, 4, None, 4, 1, 1, 4, 0, 0, None, None, None) == -1
    assert lib.plnlp_gemm_f32(None, 1, 0, 1, None, 4, 4, 4, None, 1, None, 0, None) == -1
    assert lib.plnlp_adam_step_f32(None, None, None, None, -1, 0.1, 0.9, 0.999, 1e-8, 0.0, 0, 1, None, 0.0, 1.0, None) == -2
    assert lib.plnlp_dropout_f32(None, None, 4, 4, 1.5, 0, None) == -2
    assert lib.plnlp_rmat_edges(0, 16, 0, 4, 1, 1, 2, 3, 1, None, None, None) == -2          # scale < 1
    assert lib.plnlp_rmat_edges(4, 16, 0, 4, 1, 3, 2, 3, 1, None, None, None) == -2          # thresholds not ordered
    assert lib.plnlp_rmat_edges(4, 16, 0, 4, 1, 1, 2, 3, 1, None, None, None) == -1          # NULL outputs
    assert lib.plnlp_loss_workspace_floats(65536) == 257
    assert lib.plnlp_sqnorm_partials(0) == 0 and lib.plnlp_sqnorm_partials(1) == 1


def test_wide_weight_gradient_rule_without_a_device(lib):
    """plnlp_gemm_wide_wgrad_slices is host arithmetic on the launch's arguments (no launch, no device access): which weight
    gradients take the whole-block kernel (csrc/gemm_wgw.hip) and into how many K slices -- the number the host then passes
    as split_k.  Pointers here are made-up aligned addresses; nothing is dereferenced."""
    from plnlp_amd import _lib

    def slices(m, n, k, a=0x10000, b=0x20000, lda=None, ldb=None, math=None, b2=None, ldb2=0, nb_split=None, index=0, on=3,
               a_trans=1, b_trans=0):
        ops = (_lib.GemmOperand * 1)()
        ops[0].a, ops[0].lda, ops[0].b, ops[0].ldb, ops[0].k = a, lda or m, b, ldb or (nb_split or n), k
        ops[0].math = _lib.GEMM_MATH_BF16X3 if math is None else math
        ops[0].b_index = index
        return lib.plnlp_gemm_wide_wgrad_slices(ops, 1, a_trans, b_trans, m, n, b2, ldb2, n if nb_split is None else nb_split, on)

    assert slices(200, 200, 2_927_963) == 256 and slices(200, 180, 40_000) == 256 and slices(224, 132, 32_768) == 256
    assert slices(256, 256, 40_000) == 256 and slices(256, 512, 132_224) == 128 and slices(512, 512, 262_144) == 64
    assert slices(512, 1024, 262_144) == 32 and slices(1024, 1024, 262_144) == 0                    # at most 8 blocks
    assert slices(256, 512, 132_224, b2=0x30000, ldb2=256, nb_split=256, index=0x40000, on=2) == 128    # the collab step's pair
    assert slices(256, 512, 132_224, b2=0x30000, ldb2=256, nb_split=128) == 0                        # seam inside a block
    assert slices(200, 200, 40_000, index=0x40000) == 0                                             # gathered rows: 256-blocks only
    assert slices(256, 256, 40_000, index=0x40004) == 0                                             # row list off 16 bytes
    assert slices(200, 200, 32_767) == 0 and slices(128, 200, 40_000) == 0 and slices(228, 200, 40_000) == 0
    assert slices(202, 200, 40_000) == 0 and slices(200, 200, 40_000, a=0x10008) == 0 and slices(200, 200, 40_000, lda=202) == 0
    assert slices(200, 200, 40_000, math=_lib.GEMM_MATH_F32) == 0
    assert slices(200, 200, 40_000, a_trans=0) == 0 and slices(200, 200, 40_000, b_trans=1) == 0


def test_product_ops_refuse_cpu_tensors():
    import plnlp_amd
    from plnlp_amd._lib import PlnlpHipError
    g = plnlp_amd.Graph.from_coo(torch.tensor([0, 1]), torch.tensor([1, 0]), None, 2, 2)
    enc = plnlp_amd.SAGE(4, 4, 4, 1, 0.0)
    with pytest.raises(PlnlpHipError):
        enc(torch.randn(2, 4), g)
    with pytest.raises(PlnlpHipError):
        plnlp_amd.loss.auc_loss(torch.randn(3), torch.randn(3), 1)
    with pytest.raises(PlnlpHipError):
        plnlp_amd.DotPredictor().score_edges(torch.randn(2, 4), torch.tensor([0]), torch.tensor([1]))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "plnlp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_dense_aggregation_slices_in_whole_rounds_without_a_device(lib):
    """plnlp_dense_aggregate_scratch_bytes is host arithmetic: x's three-term image + one partial result per K slice.  The slice
    count behind it fills whole rounds of the 512 workgroups the chip holds (csrc/aggregate_dense.hip::dense_slices): ogbl-ddi's
    34 x 4 tiles over 267 K-steps take THREE slices (408 workgroups, one round) -- four, the first rule, were 544: a second round
    for 32 of them; plnlp_dense_aggregate_tuning forces a count (and the scratch size follows)."""
    n, f = 4267, 512
    image = 4 * ((n + 15) // 16) * 768 * 16                    # column tiles x K-steps x (3 terms x 2 x 128 units) x 16 bytes

    def slices():
        extra = lib.plnlp_dense_aggregate_scratch_bytes(n, n, f) - image
        assert extra % (n * f * 4) == 0
        return extra // (n * f * 4)
    try:
        assert slices() == 3
        for forced in (1, 4, 7):
            lib.plnlp_dense_aggregate_tuning(forced)
            assert slices() == forced
        lib.plnlp_dense_aggregate_tuning(1000)                  # at least 8 K-steps per slice: 267 // 8 = 33
        assert slices() == 33
    finally:
        lib.plnlp_dense_aggregate_tuning(0)
    assert slices() == 3
    assert lib.plnlp_dense_aggregate_scratch_bytes(0, n, f) == 0 and lib.plnlp_dense_aggregate_scratch_bytes(n, n, 0) == 0
    # a small dense graph (tests): one tile row, few K-steps -- never more slices than K-steps / 8
    small = lib.plnlp_dense_aggregate_scratch_bytes(100, 100, 64)
    assert small == 1 * 7 * 768 * 16 + 1 * 100 * 64 * 4
