"""ctypes binding of libplnlp_hip.so (the C ABI in include/plnlp_hip.h).

No fallback: if the library cannot be loaded, every op raises.  PyTorch is used
only to own device memory and streams; tensors cross the boundary as raw
pointers + sizes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLNLP_HIP_LIB: another build of the same ABI (A/B runs of a kernel variant); read once at import
LIB_PATH = os.environ.get("PLNLP_HIP_LIB") or os.path.join(_HERE, "libplnlp_hip.so")

c_f32p = C.c_void_p   # device pointers travel as integers
c_i64 = C.c_int64


class Epilogue(C.Structure):
    """mirror of plnlp_epilogue"""
    _fields_ = [("flags", C.c_uint32), ("dropout_p", C.c_float), ("dropout_seed", C.c_uint64),
                ("bias", C.c_void_p), ("gate", C.c_void_p), ("ld_gate", C.c_int64),
                ("gate_scale", C.c_float), ("gate_index", C.c_void_p), ("addend", C.c_void_p),
                ("ld_addend", C.c_int64), ("addend_index", C.c_void_p), ("dropout_row_index", C.c_void_p),
                ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_step", C.c_int64),
                ("adam_lr", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
                ("dropout_seed_ptr", C.c_void_p), ("adam_scalars", C.c_void_p),
                ("rowdot_w", C.c_void_p), ("rowdot_out", C.c_void_p), ("rowdot_ld", C.c_int64)]


class RowSplit(C.Structure):
    """mirror of plnlp_row_split"""
    _fields_ = [("threshold", C.c_int64), ("n_long", C.c_int64), ("long_rows", C.c_void_p),
                ("chunk_beg", C.c_void_p), ("chunk_cnt", C.c_void_p), ("n_chunks", C.c_int64),
                ("chunk_long", C.c_void_p), ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
                ("seg_beg", C.c_void_p), ("seg_len", C.c_void_p), ("seg_slot", C.c_void_p)]


class GemmOperand(C.Structure):
    """mirror of plnlp_gemm_operand"""
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int64), ("b", C.c_void_p), ("ldb", C.c_int64),
                ("k", C.c_int64), ("b_index", C.c_void_p), ("a_index", C.c_void_p),
                ("math", C.c_int32), ("flags", C.c_int32), ("a_index2", C.c_void_p), ("b_index2", C.c_void_p),
                ("b_terms", C.c_void_p), ("b_terms_bytes", C.c_int64), ("a_colsum", C.c_void_p)]


class AdamTensor(C.Structure):
    """mirror of plnlp_adam_tensor"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("step", C.c_int64), ("sqnorm", C.c_void_p), ("max_norm", C.c_float),
                ("step_scalars", C.c_void_p)]


MULTI_MAX = 16
GEMM_MATH_F32, GEMM_MATH_BF16X3 = 0, 1
GEMM_FLAG_WIDE_WGRAD = 1          # plnlp_gemm_operand.flags
EPI_BIAS, EPI_RELU, EPI_DROPOUT, EPI_ACCUM, EPI_GATE, EPI_ADDEND, EPI_ADAM, EPI_ROWDOT = 1, 2, 4, 8, 16, 32, 64, 128
REDUCE_SUM, REDUCE_MEAN = 0, 1
AGG_LDS_STAGE = 2
AGG_NT_LOADS = 4
AGG_FEW_IN_FLIGHT = 8
AGG_SLABS_128 = 16
AGG_SLABS_256 = 32
AGG_SLABS_XCD = 64
AGG_HUB_XCD = 128
AGG_FUSED_PASSES = 256
LOSS_KINDS = {"auc": 0, "hinge_auc": 1, "weighted_auc": 2, "adaptive_auc": 3,
              "weighted_hinge_auc": 4, "adaptive_hinge_auc": 5, "log_rank": 6}

# name -> (restype, argtypes); this table is also what tests check against the header
SIGNATURES = {
    "plnlp_abi_version": (C.c_int, []),
    "plnlp_error_string": (C.c_char_p, [C.c_int]),
    "plnlp_row_split_build": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_random_walk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_int, C.c_uint64, C.c_void_p,
                                    C.c_void_p]),
    "plnlp_rmat_edges": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_host_randperm_init": (C.c_int, [C.c_uint64, c_i64, C.c_void_p, C.c_void_p]),
    "plnlp_host_randperm_advance": (C.c_int, [c_i64, C.c_void_p, C.c_void_p, c_i64, c_i64]),
    "plnlp_incidence_temp_bytes": (c_i64, [c_i64]),
    "plnlp_incidence_build": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_compact_rows_workspace": (c_i64, [c_i64]),
    "plnlp_edge_endpoints": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_compact_endpoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_compact_rows": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "plnlp_csr_aggregate_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, c_i64,
                                          C.c_void_p, c_i64, c_i64, c_i64, c_i64, C.c_int, C.c_int, C.POINTER(Epilogue),
                                          C.POINTER(RowSplit), C.c_void_p]),
    "plnlp_csr_aggregate_max_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, c_i64,
                                              C.c_void_p, c_i64, c_i64, c_i64, C.POINTER(RowSplit), C.c_void_p,
                                              C.c_void_p]),
    "plnlp_csr_aggregate_max_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64,
                                                  C.c_void_p, c_i64, C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p]),
    "plnlp_gemm_f32": (C.c_int, [C.POINTER(GemmOperand), C.c_int, C.c_int, C.c_int, C.c_void_p, c_i64, c_i64,
                                 c_i64, C.POINTER(Epilogue), C.c_int, C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_gemm_b_terms_bytes": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "plnlp_gemm_stationary_tuning": (None, [C.c_int, C.c_int]),
    "plnlp_gemm_block_tuning": (None, [C.c_int]),
    "plnlp_edge_segment_tuning": (None, [C.c_int]),
    "plnlp_dense_aggregate_tuning": (None, [C.c_int]),
    "plnlp_gemm_rowdot_tiles": (C.c_int, [c_i64, c_i64]),
    "plnlp_rowdot_finish_f32": (C.c_int, [C.c_void_p, c_i64, C.c_int, c_i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_gemm_stationary_applies": (C.c_int, [C.POINTER(GemmOperand), C.c_int, C.c_int, C.c_int, C.c_void_p, c_i64, c_i64,
                                                c_i64, C.c_void_p, c_i64, c_i64]),
    "plnlp_gemm_wide_wgrad_slices": (C.c_int, [C.POINTER(GemmOperand), C.c_int, C.c_int, C.c_int, c_i64, c_i64, C.c_void_p,
                                               c_i64, c_i64, C.c_int]),
    "plnlp_mlp_head_backward_workspace_floats": (c_i64, [c_i64, c_i64]),
    "plnlp_mlp_head_backward_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_float, c_i64, c_i64, C.c_void_p,
                                              c_i64, C.c_void_p, C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_dense_aggregate_scratch_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "plnlp_dense_aggregate_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, c_i64,
                                            c_i64, c_i64, c_i64, C.POINTER(Epilogue), C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_launch_counts": (C.c_int, [C.c_void_p, C.c_int]),
    "plnlp_launch_kind_name": (C.c_char_p, [C.c_int]),
    "plnlp_gemm_split_out_f32": (C.c_int, [C.POINTER(GemmOperand), C.c_int, C.c_int, C.c_int, C.c_void_p, c_i64,
                                           C.c_void_p, c_i64, c_i64, c_i64, c_i64, C.POINTER(Epilogue),
                                           C.c_void_p]),
    "plnlp_gemm_concat_b_f32": (C.c_int, [C.POINTER(GemmOperand), C.c_void_p, c_i64, c_i64, C.c_int, C.c_int,
                                          C.c_void_p, c_i64, c_i64, c_i64, C.POINTER(Epilogue), C.c_int,
                                          C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_gemm_pair_f32": (C.c_int, [C.POINTER(GemmOperand), C.c_void_p, c_i64, c_i64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      c_i64, C.c_void_p, c_i64, c_i64, c_i64, c_i64, C.c_int, C.c_void_p, c_i64,
                                      C.c_void_p]),
    "plnlp_colsum_workspace_floats": (c_i64, [c_i64, c_i64]),
    "plnlp_colsum_f32": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                   c_i64, C.c_void_p]),
    "plnlp_matvec_f32": (C.c_int, [C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "plnlp_outer_f32": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_void_p, c_i64, C.POINTER(Epilogue),
                                  C.c_void_p]),
    "plnlp_edge_dot_fwd_f32": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, c_i64, c_i64,
                                         C.c_void_p, C.c_void_p]),
    "plnlp_edge_hadamard_fwd_f32": (C.c_int, [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, c_i64, c_i64,
                                              C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_edge_scatter_bwd_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, c_i64, c_i64,
                                             C.c_void_p, c_i64, C.c_int, C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_edge_segment_bwd_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, c_i64, C.c_void_p,
                                             C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_int, C.c_void_p, c_i64,
                                             C.POINTER(Epilogue), C.c_void_p]),
    "plnlp_loss_workspace_floats": (c_i64, [c_i64]),
    "plnlp_pairwise_loss_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_pairwise_loss_tail_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_float,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p,
                                               C.c_void_p, C.c_double, C.c_void_p]),
    "plnlp_sqnorm_partials": (c_i64, [c_i64]),
    "plnlp_sqnorm_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_sum_partials_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_int, C.c_void_p]),
    "plnlp_adam_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_float,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, c_i64, C.c_void_p,
                                      C.c_float, C.c_float, C.c_void_p]),
    "plnlp_sqnorm_multi_f32": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(c_i64), C.c_int, C.c_void_p, c_i64,
                                         C.c_void_p]),
    "plnlp_sqnorm_multi_sum_f32": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(c_i64), C.c_int, C.c_void_p, c_i64,
                                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "plnlp_adam_multi_f32": (C.c_int, [C.POINTER(AdamTensor), C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_int, C.c_float, C.c_void_p]),
    "plnlp_adam_step_scalars": (C.c_int, [C.c_float, C.c_float, C.c_float, c_i64, C.c_void_p]),
    "plnlp_clip_scale_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, C.c_float, C.c_void_p]),
    "plnlp_gate_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, c_i64, C.c_void_p]),
    "plnlp_dropout_f32": (C.c_int, [C.c_void_p, C.c_void_p, c_i64, c_i64, C.c_float, C.c_uint64, C.c_void_p]),
    "plnlp_transpose_f32": (C.c_int, [C.c_void_p, c_i64, C.c_void_p, c_i64, c_i64, c_i64, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class PlnlpHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises if it is missing -- there is no
    fallback path; build it with `python -m plnlp_amd.build`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PlnlpHipError(
            f"{LIB_PATH} not found: the HIP extension is required (python -m plnlp_amd.build); "
            "plnlp_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.plnlp_abi_version() != 12:
        raise PlnlpHipError("libplnlp_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().plnlp_error_string(rc).decode()
        raise PlnlpHipError(f"{what}: {msg} (code {rc})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """hipStream_t of the current device's current stream.  (A step asks ~11 times; the Stream object that
    torch.cuda.current_stream() builds per call was 0.15 ms of the host's ~0.9 ms per step, the raw query is ~1 us.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_device(*tensors: Optional[torch.Tensor]) -> None:
    """every operand must live on the CURRENT device: the kernels are launched on the current
    device's current stream (stream_ptr), so a tensor of another GPU would be touched from the wrong
    device's queue -- callers select the device first (torch.cuda.set_device / torch.cuda.device)"""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise PlnlpHipError(
                "plnlp_amd ops run only on an MI355X device tensor (got a CPU tensor); "
                "there is no CPU path in the product -- the CPU oracle lives in oracle/ for tests")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise PlnlpHipError(
                f"operand on cuda:{t.device.index} but the current device is cuda:{cur}: kernels launch on the "
                "current device's stream -- call torch.cuda.set_device(device) (BaseModel does) first")


def make_epilogue(*, bias=None, relu=False, dropout_p=0.0, dropout_seed=0, accumulate=False,
                  gate=None, gate_scale=1.0, gate_index=None, addend=None,
                  addend_index=None, dropout_rows=None, adam=None, dropout_seed_ptr: int = 0,
                  adam_scalars_ptr: int = 0) -> Optional[Epilogue]:
    """adam = (exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps): PLNLP_EPI_ADAM -- the launch's `out` is the
    parameter, the result its gradient (plnlp_csr_aggregate_f32 only).
    dropout_seed_ptr / adam_scalars_ptr: device addresses of the per-step scalars (a captured step: see
    plnlp_epilogue in include/plnlp_hip.h and plnlp_amd/capture.py); 0 = the by-value fields are used"""
    flags = 0
    e = Epilogue()
    if bias is not None:
        flags |= EPI_BIAS
        e.bias = bias.data_ptr()
    if relu:
        flags |= EPI_RELU
    if dropout_p > 0.0:
        flags |= EPI_DROPOUT
        e.dropout_p = float(dropout_p)
        e.dropout_seed = int(dropout_seed) & 0xFFFFFFFFFFFFFFFF
        if dropout_seed_ptr:
            e.dropout_seed_ptr = int(dropout_seed_ptr)
        if dropout_rows is not None:
            assert dropout_rows.dtype == torch.int32
            e.dropout_row_index = dropout_rows.data_ptr()
    if accumulate:
        flags |= EPI_ACCUM
    if gate is not None:
        flags |= EPI_GATE
        e.gate = gate.data_ptr()
        e.ld_gate = gate.stride(0)
        e.gate_scale = float(gate_scale)
        if gate_index is not None:
            assert gate_index.dtype == torch.int32
            e.gate_index = gate_index.data_ptr()
    if addend is not None:
        flags |= EPI_ADDEND
        e.addend = addend.data_ptr()
        e.ld_addend = addend.stride(0)
        if addend_index is not None:
            assert addend_index.dtype == torch.int32
            e.addend_index = addend_index.data_ptr()
    if adam is not None:
        m, v, step, lr, b1, b2, eps = adam
        assert m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and v.dtype == torch.float32
        flags |= EPI_ADAM
        e.adam_m, e.adam_v, e.adam_step = m.data_ptr(), v.data_ptr(), int(step)
        e.adam_lr, e.adam_beta1, e.adam_beta2, e.adam_eps = float(lr), float(b1), float(b2), float(eps)
        if adam_scalars_ptr:
            e.adam_scalars = int(adam_scalars_ptr)
    if flags == 0:
        return None
    e.flags = flags
    e._keepalive = (bias, gate, gate_index, addend, addend_index, dropout_rows, adam)      # the struct holds raw pointers; keep the tensors alive with it
    return e
