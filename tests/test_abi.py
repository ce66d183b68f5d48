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
    assert lib.plnlp_abi_version() == int(re.search(r"#define PLNLP_ABI_VERSION (\d+)", header).group(1)) == 11
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
