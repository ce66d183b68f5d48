"""Host-side operators of the MI355X path: thin Python over the C ABI
(include/plnlp_hip.h) plus the torch.autograd.Function wrappers that make the
HIP kernels differentiable, so `loss.backward()` in a reference-style training
loop (plnlp/model.py:161) runs the hand-written backward kernels.

Every function here requires device tensors; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from .graph import Graph

# ------------------------------------------------------------------ seeds ------
_seed_state = {"base": None, "counter": 0}


def manual_seed(seed: int) -> None:
    """Seed the counter-based dropout stream (identical on every data-parallel rank so replicated
    encoder passes agree).  Without this call the base is taken from torch's default generator on
    first use (`torch.initial_seed()`), so `torch.manual_seed(s)` alone also fixes the masks and an
    unseeded run -- the reference never seeds -- draws fresh masks per launch."""
    _seed_state["base"] = int(seed) & 0xFFFFFFFFFFFFFFFF
    _seed_state["counter"] = 0


def seed_state():
    """(base, counter) of the dropout stream -- what a data-parallel trainer broadcasts from rank 0"""
    if _seed_state["base"] is None:
        _seed_state["base"] = (int(torch.initial_seed()) ^ 0x5DEECE66D) & 0xFFFFFFFFFFFFFFFF
    return _seed_state["base"], _seed_state["counter"]


def set_seed_state(base: int, counter: int) -> None:
    _seed_state["base"], _seed_state["counter"] = int(base) & 0xFFFFFFFFFFFFFFFF, int(counter)


def next_seed() -> int:
    """splitmix64(base + counter * golden): one fresh 64-bit seed per dropout call."""
    base, _ = seed_state()
    _seed_state["counter"] += 1
    z = (base + _seed_state["counter"] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32, unit stride along the last dim (leading dimension = stride(0))."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() == 2 and (t.stride(1) != 1 or t.stride(0) < t.shape[1]):
        t = t.contiguous()
    elif t.dim() == 1 and t.stride(0) != 1:
        t = t.contiguous()
    return t


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


# ------------------------------------------------------------- raw kernels ------
# rows longer than this are cut into chunks of this many edges (one wave each).  0 = by the size of the matrix the
# rows gather from: 128 up to a million source rows, 1024 beyond.  Measured on MI355X (profiles/r02_split_threshold.txt):
# on the cache-resident graphs the hub pass is latency-bound and short of waves (collab: 735 rows hold 31 % of the
# edges = 2 900 chunks of 256 for 256 CUs) -- 128-edge chunks: collab step -2.7 %, ddi step -3.4 %, ddi aggregation
# -23 %; on citation2 / R-MAT-23 (sources far beyond the caches) there are waves enough and fewer, longer chunks
# save partial sums: 1024 vs 256: citation2 step -1.1 %, R-MAT-23 aggregation -5.7 %; 128 costs 1 %, 64 costs 6 %.
SPLIT_THRESHOLD = 0


def split_threshold(n_source_rows: int) -> int:
    if SPLIT_THRESHOLD > 0:
        return SPLIT_THRESHOLD
    return 128 if n_source_rows <= (1 << 20) else 1024
LDS_STAGE_BUDGET = 152 * 1024   # bytes of LDS a staged feature slab may take (n_src * 16 B at the narrowest)
LDS_STAGE_MIN_DEG = 32          # average row length from which staging x in LDS could pay for itself
LDS_STAGE_AUTO = False          # measured on MI355X (ddi-shaped, F=512): LDS-staged 0.44 ms vs streaming 0.32 ms --
                                # both issue one wave instruction per KiB gathered, and the staged form adds the
                                # per-chunk index loads and a shuffle tree; the L2-resident stream wins, so the
                                # staged form is opt-in (lds_stage=True)


def _vector_path(x: torch.Tensor, out: torch.Tensor, feat: int) -> bool:
    return (feat % 4 == 0 and feat <= 1024 and _ld(x) % 4 == 0 and _ld(out) % 4 == 0
            and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0)


# host-side bit of a tuned form (never passed to the library): the long rows' chunks are cut by SOURCE RANGE and
# processed range-major (graph.SourceOrderedSplit) instead of by position.  With the chunk pass in XCD-pinned slabs
# (AGG_HUB_XCD) the waves in flight on an XCD gather the same 128-byte pieces of the same source rows.  Measured on
# MI355X, collab graph, F = 256 (profiles/r03_agg_hub_pass.md): chunk pass 117 -> 80 us, whole aggregation 0.288 -> 0.246 ms
AGG_HUB_RANGES = 1 << 16
# a range is at least part_rows source rows (65 536 measured best on the collab graph, 16 K ... 128 K within 4 %), and a
# graph is cut into at most max_ranges of them: a hub of degree d leaves d / ranges entries per chunk, and chunks of a
# handful of entries are all latency (12.5 M-node R-MAT at 65 536 rows per range: 10 M chunks of 7 entries)
HUB_RANGES = {"part_rows": 65536, "max_len": 256, "max_ranges": 8}


# the long rows' chunk pass inside the main pass's launch (PLNLP_AGG_FUSED_PASSES): the same sums in the same order --
# not a tuned form, a launch shape; AGG_FUSED["enabled"] = False restores the three-launch sequence for A/B runs
AGG_FUSED = {"enabled": True}


def hub_ranges(n_source_rows: int):
    """(part_rows, max_len) of graph.SourceOrderedSplit for a source matrix of this many rows"""
    per = -(-int(n_source_rows) // HUB_RANGES["max_ranges"])
    return max(HUB_RANGES["part_rows"], per), HUB_RANGES["max_len"]
# PLNLP_AGG_FORM=<int>: no measurement, this form on every static graph (counter-collection runs, whose serialised
# kernels would otherwise time differently from the run they are meant to explain)
AGG_AUTOTUNE = {"enabled": os.environ.get("PLNLP_AGG_AUTOTUNE", "1") != "0", "min_feat": 256,
                "candidates": (0, L.AGG_SLABS_128, L.AGG_SLABS_256, L.AGG_HUB_XCD | AGG_HUB_RANGES),
                "force": int(os.environ["PLNLP_AGG_FORM"]) if os.environ.get("PLNLP_AGG_FORM") else None}   # (counter runs pin the form)


def describe_form(tune: int) -> str:
    """a tuned aggregation form in words (bench lines, profiles)"""
    t = int(tune)
    rows = {0: "one wave per row", L.AGG_SLABS_128: "one wave per (row, 128-column slab)",
            L.AGG_SLABS_256: "one wave per (row, 256-column slab)",
            L.AGG_SLABS_XCD: "one wave per (row, F/8-column slab pinned to an XCD)"}.get(
                t & (L.AGG_SLABS_128 | L.AGG_SLABS_256 | L.AGG_SLABS_XCD), "?")
    hubs = "long rows: position chunks"
    if t & L.AGG_HUB_XCD:
        hubs = ("long rows: chunks by source range, eight XCD-pinned column slabs" if t & AGG_HUB_RANGES
                else "long rows: position chunks in eight XCD-pinned column slabs")
    return "%s; %s (measured choice, ops._agg_tune)" % (rows, hubs)


def mapped_form(tune: int) -> int:
    """the form a launch with a source map runs when the graph's tuned form is `tune`"""
    return int(tune) & ~(L.AGG_HUB_XCD | AGG_HUB_RANGES)


def _multi_rank() -> bool:
    d = torch.distributed
    return d.is_available() and d.is_initialized() and d.get_world_size() > 1


# The forms differ in summation order, so a per-box measurement means per-box BITS.  The chosen form per (graph shape,
# width) is therefore kept in a file that ships with the package: a shape found there is not measured again -- every box
# then runs the same kernels on it and two boxes give the same bits.  A shape that is not in the file is measured as
# before and remembered for the process only; scripts/pin_agg_forms.py measures the benchmark shapes and rewrites the file.
# PLNLP_AGG_FORMS_FILE=none: measure per process.
AGG_FORMS = {"path": os.environ.get("PLNLP_AGG_FORMS_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                            "agg_forms.json"),
             "table": None, "hits": 0, "measured": {}}


def agg_form_key(graph, feat: int) -> str:
    """what a pinned form is keyed by: the static graph's shape (rows, source rows, entries) and the width.  A graph and
    its transposed view share one choice (their passes cannot be timed apart), so the key is orientation-free."""
    lo, hi = sorted((int(graph.n_rows), int(graph.n_cols)))
    return "%d:%d:%d:f%d" % (lo, hi, int(graph.col.numel()), int(feat))


def pinned_agg_form(graph, feat: int):
    """the form the shipped table names for this shape, or None"""
    if AGG_FORMS["path"] in (None, "", "none"):
        return None
    if AGG_FORMS["table"] is None:
        import json
        try:
            with open(AGG_FORMS["path"]) as f:
                AGG_FORMS["table"] = {k: int(v) for k, v in json.load(f).get("forms", {}).items()}
        except (OSError, ValueError):
            AGG_FORMS["table"] = {}
    hit = AGG_FORMS["table"].get(agg_form_key(graph, feat))
    if hit is not None:
        AGG_FORMS["hits"] += 1
    return hit


def _time_agg_forms(graph, x, out, reduce, use_values, src_scale, epilogue) -> int:
    """which candidate form is fastest on this graph, by measurement on this rank: every form is launched once untimed
    (its tables get built, the clocks come up), then five timed rounds go over the forms in turn -- a cold or drifting
    clock then biases no form -- and each form keeps its best time; a form other than the default must win by 3 %"""
    if AGG_AUTOTUNE["force"] is not None:
        return int(AGG_AUTOTUNE["force"])
    pinned = pinned_agg_form(graph, x.shape[1])
    if pinned is not None:
        return pinned
    cands = list(AGG_AUTOTUNE["candidates"])
    # nothing else should run beside the measurement (a trainer reaches this point with the next batch's index preparation
    # queued on the side stream).  (On the collab graph the weighted full-graph launch timed here is a near-tie between the
    # plain and the hub-by-source-range form -- fresh processes pick either; the step differs by ~5 us between them.)
    torch.cuda.synchronize(x.device)

    def run(cand):
        was = DENSE_AGG["enabled"]
        DENSE_AGG["enabled"] = False          # (the CSR kernels' forms are what is compared here, also on a graph the dense form takes)
        try:
            csr_aggregate(graph, x, reduce, use_values, src_scale=src_scale, out=out, epilogue=epilogue, tune=cand)
        finally:
            DENSE_AGG["enabled"] = was
    for cand in cands:
        run(cand)
    for cand in cands:
        run(cand)
    best_t = {c: float("inf") for c in cands}
    for _ in range(5):
        for cand in cands:
            s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_ev.record()
            run(cand)
            e_ev.record()
            e_ev.synchronize()
            best_t[cand] = min(best_t[cand], s_ev.elapsed_time(e_ev))
    best = cands[0]
    for cand in cands[1:]:
        if best_t[cand] < 0.97 * best_t[best]:
            best = cand
    AGG_FORMS["measured"][agg_form_key(graph, x.shape[1])] = {"form": int(best), "ms": {str(c): round(best_t[c], 4) for c in cands}}
    return best


def tune_aggregation(graph, feats, group=None, time_fn=None) -> dict:
    """Decide -- by measurement -- which form of the aggregation kernel runs on the STATIC `graph` at each
    feature width in `feats`, and remember it on the graph object (shared with its transposed views).

    This is the ONLY place where ranks agree on a form: every rank of `group` must call it (a trainer does,
    from BaseModel._tune_graph, at a point all its ranks pass together); rank 0's measurement is broadcast
    so that replicated ranks run identical kernels (the forms differ in summation order, and replicas stay
    bit-identical only on the same form).  group=None: this process decides alone, no collective.
    The aggregation op itself never communicates (csr_aggregate -> _agg_tune).
    time_fn(graph, feat) -> form: injectable measurement (CPU tests of the agreement logic)."""
    picked = {}
    if not AGG_AUTOTUNE["enabled"]:
        return picked
    for feat in sorted(set(int(f) for f in feats)):
        if feat < AGG_AUTOTUNE["min_feat"] or feat % 4 != 0 or feat > 1024:
            continue
        cache = graph._agg_tune
        best = cache.get(feat)
        if best is None:
            if time_fn is not None:
                best = int(time_fn(graph, feat))
            else:
                x = torch.empty(graph.n_cols, feat, dtype=torch.float32, device=graph.device).normal_()
                out = torch.empty(graph.n_rows, feat, dtype=torch.float32, device=graph.device)
                best = _time_agg_forms(graph, x, out, "mean" if graph.val is None else "sum", graph.val is not None,
                                       None, None)
                del x, out
        if group is not None and torch.distributed.get_world_size(group) > 1:
            dev = graph.device if graph.device.type == "cuda" else torch.device("cpu")
            pick = torch.tensor([best], dtype=torch.int64, device=dev)
            torch.distributed.broadcast(pick, torch.distributed.get_global_rank(group, 0), group=group)
            best = int(pick.item())
        cache[feat] = best
        picked[feat] = cache[feat]
    return picked


def _agg_tune(graph, x, out, reduce, use_values, src_scale, src_map, epilogue, feat, row_index=None) -> int:
    """which form of the aggregation kernel to run on a STATIC graph at this feature width: one wave per
    row, or one per (row, 128- / 256-column slab).  Which one wins depends on whether the source matrix
    is cache-resident and on the degree skew (measured on MI355X, uniform 2.9 M-node graph, F = 512:
    0.72 -> 0.79 of the HBM peak with 128-column slabs; R-MAT and the cache-resident collab graph lose
    10-25 % with them), so it is MEASURED once per (graph, width, form) and remembered on the graph
    object.  NO collective happens here: in a multi-rank job the choice is made by tune_aggregation()
    (called by the trainer on all ranks of its group); an untuned graph then runs the default form --
    a rank that aggregates on its own (a control model, a roofline measurement on rank 0) can never
    leave its peers waiting in a broadcast.  In a single process the first plain call measures.
    Per-batch structures (incidence lists) and calls whose epilogue accumulates into `out` are never
    used for timing."""
    if (not AGG_AUTOTUNE["enabled"] or not isinstance(graph, Graph) or feat < AGG_AUTOTUNE["min_feat"]
            or not _vector_path(x, out, feat)):
        return 0
    flags = 0 if epilogue is None else int(epilogue.flags)
    if flags & L.EPI_DROPOUT:
        return 0
    cache = graph._agg_tune
    key = feat                        # one choice per width, shared by the weighted / transposed passes
    if key in cache:
        return cache[key]
    if flags & (L.EPI_ACCUM | L.EPI_ADDEND | L.EPI_GATE | L.EPI_ADAM) or src_map is not None or row_index is not None:
        return 0                      # not a call to time on; an earlier or later plain call decides
    if _multi_rank():
        return 0                      # the ranks agree in tune_aggregation(), never inside the op
    cache[key] = _time_agg_forms(graph, x, out, reduce, use_values, src_scale, epilogue)
    return cache[key]


# The aggregation of a DENSE graph as a product on the matrix cores (csrc/aggregate_dense.hip; plnlp_dense_aggregate_f32): the
# adjacency as a bf16 matrix of counts times x split into three bf16 terms.  For graphs like ogbl-ddi (4 267 nodes, 11.7 % of all
# pairs are edges), where every row is a "long row" of the CSR kernels.  Same f32-grade sums in another order; off = CSR everywhere.
DENSE_AGG = {"enabled": os.environ.get("PLNLP_DENSE_AGG", "1") != "0", "min_density": 0.05, "min_rows": 2048, "max_rows": 16384}


def _dense_aggregate(graph, x, reduce, use_values, src_scale, out, epilogue):
    """csr_aggregate's launch on the dense form, or None where it does not apply (the caller then runs the CSR kernels)"""
    if not (DENSE_AGG["enabled"] and isinstance(graph, Graph) and x.is_cuda and x.shape[1] % 4 == 0):
        return None
    if GEMM_MATH["mode"] != "bf16x3":        # it IS a split-bf16 product: with the dense products on the exact-f32 MFMA (the
        return None                          # reference's arithmetic) the aggregation stays the CSR kernels' chain of f32 adds
    if use_values and graph.val is not None and getattr(graph, "_col_scale", None) is None:
        return None                          # arbitrary entry values (GCN's normalised adjacency): the CSR kernels
    counts = graph.dense_counts(DENSE_AGG["min_density"], DENSE_AGG["min_rows"], DENSE_AGG["max_rows"])
    if counts is None:
        return None
    lib = L.load()
    n, feat = graph.n_rows, x.shape[1]
    scale = getattr(graph, "_col_scale", None) if (use_values and graph.val is not None) else None
    if src_scale is not None:
        scale = src_scale if scale is None else scale * src_scale
    row_scale = graph.inv_degree() if reduce == "mean" else None
    if out is None:
        out = torch.empty(n, feat, dtype=torch.float32, device=x.device)
    if _ld(x) % 4 or _ld(out) % 4 or x.data_ptr() % 16 or out.data_ptr() % 16:
        return None
    nbytes = lib.plnlp_dense_aggregate_scratch_bytes(n, graph.n_cols, feat)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    rc = lib.plnlp_dense_aggregate_f32(counts.data_ptr(), counts.shape[1], L.ptr(scale), L.ptr(row_scale), x.data_ptr(), _ld(x),
                                       out.data_ptr(), _ld(out), n, graph.n_cols, feat,
                                       C.byref(epilogue) if epilogue is not None else None, scratch.data_ptr(), nbytes, L.stream_ptr())
    L.check(rc, "plnlp_dense_aggregate_f32")
    return out


def csr_aggregate(graph, x: torch.Tensor, reduce: str = "sum", use_values: bool = True,
                  src_scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  epilogue: Optional[L.Epilogue] = None, split="auto",
                  lds_stage="auto", src_map: Optional[torch.Tensor] = None, tune="auto",
                  row_index: Optional[torch.Tensor] = None, out_map: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = red_{e in row r} w_e x[col[e]]  (plnlp_csr_aggregate_f32).
    `graph` needs rowptr/col/val/n_rows/n_cols (+ row_split() when split == 'auto').
    src_map (int32 [n_cols]): x holds only some source rows; entry e reads x[src_map[col[e]]] and
    is skipped where that is negative.
    row_index (int32 [R]): only R result rows are produced, out[i] = the aggregate of CSR row
    row_index[i]; out_map (int32 [n_rows]) is the inverse map (-1 = not produced) the long-row pass needs."""
    lib = L.load()
    L.require_device(x, graph.col, src_map)
    x = _f32c(x)
    if src_map is None:
        assert x.shape[0] == graph.n_cols, (x.shape, graph)
    else:
        assert src_map.dtype == torch.int32 and src_map.numel() == graph.n_cols
    feat = x.shape[1]
    n_out = graph.n_rows
    if src_map is None and row_index is None and reduce in ("sum", "mean"):
        done = _dense_aggregate(graph, x, reduce, use_values, src_scale, out, epilogue)       # a dense graph: on the matrix cores
        if done is not None:
            return done
    if row_index is not None:
        assert row_index.dtype == torch.int32 and out_map is not None and out_map.dtype == torch.int32
        n_out = row_index.numel()
        lds_stage = False
    if out is None:
        out = torch.empty(n_out, feat, dtype=torch.float32, device=x.device)
    val = graph.val if use_values else None
    val_index = getattr(graph, "val_index", None) if use_values else None
    if lds_stage == "auto":      # small AND dense (ddi-like): feature slabs of x fit in LDS and are reused
        lds_stage = (LDS_STAGE_AUTO and graph.n_cols * 16 <= LDS_STAGE_BUDGET
                     and graph.col.numel() >= LDS_STAGE_MIN_DEG * graph.n_cols and feat % 4 == 0)
    if tune == "auto":
        tune = _agg_tune(graph, x, out, reduce, use_values, src_scale, src_map, epilogue, feat, row_index)
    if src_map is not None:
        # the mapped gather keeps the position chunks at full width: over half of a hub's entries have no mapped row on
        # the step's transposed pass, and eight slab waves per chunk each repeat the index translation for what is left
        # (measured: the collab step's transposed launch 0.449 -> 0.474 ms with the pinned form)
        tune = mapped_form(tune)
    flags = (L.AGG_LDS_STAGE if lds_stage else 0) | (int(tune) & 0xFFFF)
    if AGG_FUSED["enabled"]:
        flags |= L.AGG_FUSED_PASSES
    if lds_stage:
        split = None             # the staged form walks whole rows
    sp = None
    if split == "auto":
        split = None
        if _vector_path(x, out, feat):
            if (int(tune) & AGG_HUB_RANGES) and isinstance(graph, Graph):
                split = graph.row_split(split_threshold(graph.n_cols), hub_ranges(graph.n_cols))
            else:
                split = graph.row_split(split_threshold(graph.n_cols))
    if split is not None and split.active and _vector_path(x, out, feat):
        ws = torch.empty(split.n_chunks * feat, dtype=torch.float32, device=x.device)
        sp = L.RowSplit(split.threshold, split.n_long, split.long_rows.data_ptr(), split.chunk_beg.data_ptr(),
                        split.chunk_cnt.data_ptr(), split.n_chunks, split.chunk_long.data_ptr(), ws.data_ptr(),
                        ws.numel())
        if hasattr(split, "seg_beg"):       # chunks by source range (graph.SourceOrderedSplit)
            sp.seg_beg, sp.seg_len, sp.seg_slot = (split.seg_beg.data_ptr(), split.seg_len.data_ptr(),
                                                   split.seg_slot.data_ptr())
    rc = lib.plnlp_csr_aggregate_f32(
        graph.rowptr.data_ptr(), graph.col.data_ptr() or graph.rowptr.data_ptr(), L.ptr(val), L.ptr(val_index),
        L.ptr(src_scale), L.ptr(src_map), L.ptr(row_index), L.ptr(out_map),
        x.data_ptr(), _ld(x), out.data_ptr(), _ld(out), n_out, x.shape[0], feat,
        L.REDUCE_MEAN if reduce == "mean" else L.REDUCE_SUM, flags,
        C.byref(epilogue) if epilogue is not None else None,
        C.byref(sp) if sp is not None else None, L.stream_ptr())
    L.check(rc, "plnlp_csr_aggregate_f32")
    return out


def csr_aggregate_max(graph, x: torch.Tensor, use_values: bool = True, split="auto"):
    """(out, arg) = max-aggregation of x over the rows of `graph` (plnlp_csr_aggregate_max_f32):
    out[r] = max_e w_e x[col[e]] (0 for an empty row), arg[r, f] = row-relative position of the first
    maximal entry (-1 for an empty row)."""
    lib = L.load()
    L.require_device(x, graph.col)
    x = _f32c(x)
    assert x.shape[0] == graph.n_cols, (x.shape, graph)
    feat = x.shape[1]
    out = torch.empty(graph.n_rows, feat, dtype=torch.float32, device=x.device)
    arg = torch.empty(graph.n_rows, feat, dtype=torch.int32, device=x.device)
    val = graph.val if use_values else None
    sp = ws_arg = None
    if split == "auto":
        split = graph.row_split(split_threshold(graph.n_cols)) if _vector_path(x, out, feat) else None
    if split is not None and split.active and _vector_path(x, out, feat):
        ws = torch.empty(split.n_chunks * feat, dtype=torch.float32, device=x.device)
        ws_arg = torch.empty(split.n_chunks * feat, dtype=torch.int32, device=x.device)
        sp = L.RowSplit(split.threshold, split.n_long, split.long_rows.data_ptr(), split.chunk_beg.data_ptr(),
                        split.chunk_cnt.data_ptr(), split.n_chunks, split.chunk_long.data_ptr(), ws.data_ptr(),
                        ws.numel())
    L.check(lib.plnlp_csr_aggregate_max_f32(
        graph.rowptr.data_ptr(), graph.col.data_ptr() or graph.rowptr.data_ptr(), L.ptr(val), x.data_ptr(), _ld(x),
        out.data_ptr(), _ld(out), arg.data_ptr(), _ld(arg), graph.n_rows, feat,
        C.byref(sp) if sp is not None else None, L.ptr(ws_arg), L.stream_ptr()), "plnlp_csr_aggregate_max_f32")
    return out, arg


def csr_aggregate_max_bwd(graph, gy: torch.Tensor, arg: torch.Tensor, use_values: bool = True) -> torch.Tensor:
    """gradient of csr_aggregate_max w.r.t. x: a gather over the transposed CSR
    (plnlp_csr_aggregate_max_bwd_f32); deterministic, no atomics."""
    lib = L.load()
    L.require_device(gy, arg)
    gy = _f32c(gy)
    gt, pos = graph.t(), graph.t_pos()
    feat = gy.shape[1]
    gx = torch.empty(graph.n_cols, feat, dtype=torch.float32, device=gy.device)
    val_t = gt.val if use_values else None
    L.check(lib.plnlp_csr_aggregate_max_bwd_f32(
        gt.rowptr.data_ptr(), gt.col.data_ptr() or gt.rowptr.data_ptr(), pos.data_ptr() or gt.rowptr.data_ptr(),
        L.ptr(val_t), gy.data_ptr(), _ld(gy), arg.data_ptr(), _ld(arg), gx.data_ptr(), _ld(gx), graph.n_cols, feat,
        L.stream_ptr()), "plnlp_csr_aggregate_max_bwd_f32")
    return gx


def _pick_split_k(m: int, n: int, ktiles: int) -> int:
    """split-K so that tiles x slices fills the 512 workgroup slots (256 CUs x 2) exactly once: a
    second, partly filled round leaves SIMDs with one wave (measured: 768 blocks -> 1.27 waves/SIMD,
    58 % MFMA utilisation; 512 blocks -> one full round)."""
    blocks = ((m + 127) // 128) * ((n + 127) // 128)
    if blocks == 0 or blocks >= 384 or ktiles <= 1:
        return 1
    return max(1, min(ktiles, SPLIT_K_SLOTS["slots"] // blocks))


# workgroup slots one round of a split-K launch should fill (256 CUs x 2; the split-bf16 kernels run 3 per CU, but
# 768 slots measured +5 % on ddi's 512x512 weight gradient and -6 % on collab's 256x256 -- profiles/r02_gemm_x3_splitk.txt)
SPLIT_K_SLOTS = {"slots": 512}

# how every dense product of the path is formed (include/plnlp_hip.h, PLNLP_GEMM_MATH_*): 'f32' = the f32-input
# MFMA (an fmaf chain), 'bf16x3' = operands split into three bf16 terms in the loader, six bf16 MFMAs per block
# -- the same fp32-in / fp32-out contract, per-product error below the f32 rounding, 2.67x the MFMA rate
GEMM_MATH = {"mode": os.environ.get("PLNLP_GEMM_MATH", "bf16x3")}


# the stationary-weights form of the split-bf16 GEMM (csrc/gemm_x3s.hip): the weights are split into their bf16 terms
# once per launch into a lent buffer instead of once per row panel.  Bit-identical to the 128 x 128 kernels; off =
# those kernels everywhere (A/B runs, tests of both)
GEMM_STATIONARY_B = {"enabled": os.environ.get("PLNLP_GEMM_STATIONARY_B", "1") != "0",
                     "min_rows": 16384}     # (informational: the rule itself lives in the library, plnlp_gemm_stationary_applies)


# the weight gradients with whole 224- / 256-wide blocks of the result held by one workgroup per K slice (csrc/gemm_wgw.hip;
# the rule lives in the library, plnlp_gemm_wide_wgrad_slices).  off = the 128 x 128 kernels (A/B runs, tests of both)
GEMM_WIDE_WGRAD = {"enabled": os.environ.get("PLNLP_GEMM_WIDE_WGRAD", "1") != "0"}


# the stationary-weights product with a whole 256-row block of the result per workgroup (csrc/gemm_x3b.hip; the rule lives in the
# library): 'auto' = the 224-column tiles (a layer 193 .. 224 wide) from 32 768 rows on, 'off' = never, 'all' = the 256-column
# tiles too (measured equal to gemm_x3s within -2 .. +5 %), 'nolead' / 'all-nolead' = without the leading half blocks.  Same bits.
GEMM_BLOCK = {"mode": os.environ.get("PLNLP_GEMM_BLOCK", "auto"), "applied": None}
_GEMM_BLOCK_BITS = {"auto": 0, "off": 1, "nolead": 2, "all": 4, "all-nolead": 6}


def _apply_gemm_block() -> None:
    mode = GEMM_BLOCK["mode"]
    if mode != GEMM_BLOCK["applied"]:
        if mode not in _GEMM_BLOCK_BITS:
            raise ValueError("PLNLP_GEMM_BLOCK / ops.GEMM_BLOCK['mode'] = %r: one of %s" % (mode, ", ".join(_GEMM_BLOCK_BITS)))
        L.load().plnlp_gemm_block_tuning(_GEMM_BLOCK_BITS[mode])
        GEMM_BLOCK["applied"] = mode


def _lend_b_terms(ops, n_seg: int, a_trans: bool, b_trans: bool, out: torch.Tensor, m: int, n: int,
                  out2: Optional[torch.Tensor] = None, n_split: Optional[int] = None):
    """lend the launch a scratch buffer for B's pre-split image (plnlp_gemm_operand.b_terms) exactly where the library will
    run the stationary-weights form with it -- plnlp_gemm_stationary_applies is the launch's own rule (shape, alignment,
    gathered operands, the h = 200 exception): a launch that declines the form keeps its split-K and gets no buffer.
    `ops` must be filled in (pointers, leading dimensions, K, index pointers, math).  Returns the tensor to keep alive."""
    if not GEMM_STATIONARY_B["enabled"] or ops[0].math != L.GEMM_MATH_BF16X3:
        return None
    lib = L.load()
    _apply_gemm_block()
    if not lib.plnlp_gemm_stationary_applies(ops, n_seg, int(a_trans), int(b_trans), out.data_ptr(), _ld(out), m, n,
                                             L.ptr(out2), 0 if out2 is None else _ld(out2), n if n_split is None else n_split):
        return None
    need = lib.plnlp_gemm_b_terms_bytes(m, n, int(ops[0].k), int(ops[1].k) if n_seg > 1 else 0)
    if need <= 0:
        return None
    buf = torch.empty(need, dtype=torch.uint8, device=out.device)
    ops[0].b_terms, ops[0].b_terms_bytes = buf.data_ptr(), need
    return buf


def launch_counts() -> dict:
    """{kernel family: launches so far in this process} (plnlp_launch_counts) -- which forms a run went through"""
    lib = L.load()
    n = lib.plnlp_launch_counts(None, 0)
    buf = (C.c_int64 * n)()
    lib.plnlp_launch_counts(C.cast(buf, C.c_void_p), n)
    return {lib.plnlp_launch_kind_name(i).decode(): int(buf[i]) for i in range(n)}


def _gemm_math() -> int:
    mode = GEMM_MATH["mode"]
    if mode == "bf16x3":
        return L.GEMM_MATH_BF16X3
    if mode == "f32":
        return L.GEMM_MATH_F32
    raise ValueError(f"GEMM_MATH mode {mode!r}: 'f32' or 'bf16x3'")


FUSE_HEAD_FORWARD = {"enabled": True}


def gemm(segs: Sequence[Tuple[torch.Tensor, torch.Tensor]], a_trans: bool, b_trans: bool,
         out: Optional[torch.Tensor] = None, epilogue: Optional[L.Epilogue] = None,
         split_k: Optional[int] = None, b_index: Optional[torch.Tensor] = None,
         a_index: Optional[Sequence[Optional[torch.Tensor]]] = None, a_index2: Optional[torch.Tensor] = None,
         b_index2: Optional[torch.Tensor] = None, rowdot: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None,
         a_colsum: Optional[list] = None):
    """C = EPI(sum_s op(A_s) op(B_s))  (plnlp_gemm_f32).  A_s: [M,K] or [K,M] if
    a_trans; B_s: [N,K] if b_trans (nn.Linear weight layout) else [K,N].
    b_index (int32 [K], one segment, a_trans and not b_trans): B's row for reduction index j
    is b_index[j] -- B is gathered in place.
    a_index (per segment, int32 [M] or None; not a_trans, b_trans): A_s' row for result row i is
    a_index[s][i] -- A_s is gathered in the loader (a layer evaluated at some rows only).
    a_index2 / b_index2 (int32, with a_index[0] / b_index, one segment, bf16x3 only): the gathered row is the
    elementwise PRODUCT of rows index[i] and index2[i] -- the Hadamard of an edge's endpoint rows formed in
    the loader (MLPPredictor's first linear and its weight gradient).
    rowdot = (w [n] or [1, n], bias [1] or None): also evaluate the 1-output linear  s[r] = <C[r, :], w> + bias  on the stored
    result in the launch's epilogue (PLNLP_EPI_ROWDOT: the stationary-weights kernel only) -- returns (C, s [m, 1]), or
    (C, None) when the launch does not take that form and the caller has to make its own pass over C.
    a_colsum (a list; a_trans weight gradients): when the launch takes the whole-block weight-gradient kernel, the column sums of
    A over the reduction index -- the bias gradient next to dW = dz^T x -- come out of the same launch and are appended to
    the list; the list stays empty when the launch runs another kernel (the caller then sums dz itself)."""
    lib = L.load()
    ops = (L.GemmOperand * len(segs))()
    ops[0].math = _gemm_math()
    if a_index2 is not None:
        assert a_index is not None and a_index[0] is not None and len(segs) == 1 and a_index2.dtype == torch.int32
        assert a_index2.numel() == a_index[0].numel()
        ops[0].a_index2 = a_index2.data_ptr()
    if b_index2 is not None:
        assert b_index is not None and len(segs) == 1 and b_index2.dtype == torch.int32
        assert b_index2.numel() == b_index.numel()
        ops[0].b_index2 = b_index2.data_ptr()
    m = n = None
    ktiles = 0
    keep = []
    for i, (a, b) in enumerate(segs):
        L.require_device(a, b)
        a, b = _f32c(a), _f32c(b)
        keep += [a, b]
        ma, ka = (a.shape[1], a.shape[0]) if a_trans else a.shape
        nb, kb = b.shape if b_trans else (b.shape[1], b.shape[0])
        if a_index is not None and a_index[i] is not None:
            assert a_index[i].dtype == torch.int32 and not a_trans and b_trans
            ma = a_index[i].numel()
            ops[i].a_index = a_index[i].data_ptr()
        if b_index is not None:
            assert b_index.dtype == torch.int32 and len(segs) == 1 and a_trans and not b_trans
            kb = b_index.numel()
            ops[i].b_index = b_index.data_ptr()
        assert ka == kb, f"segment {i}: K mismatch {a.shape} {b.shape}"
        assert m in (None, ma) and n in (None, nb)
        m, n = ma, nb
        ops[i].a, ops[i].lda, ops[i].b, ops[i].ldb, ops[i].k = a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), ka
        ktiles += (ka + 31) // 32
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=keep[0].device)
    if split_k is None:
        # (the wide weight-gradient form -- the whole <= 224 x 224 result per workgroup -- is taken by the launch exactly
        # when K is cut into the slices the library names for it)
        wide = (lib.plnlp_gemm_wide_wgrad_slices(ops, len(segs), int(a_trans), int(b_trans), m, n, None, 0, n, 3)
                if GEMM_WIDE_WGRAD["enabled"] else 0)
        if wide:
            ops[0].flags |= L.GEMM_FLAG_WIDE_WGRAD
        split_k = wide or _pick_split_k(m, n, ktiles)
    split_k = max(1, min(split_k, ktiles))
    keep.append(_lend_b_terms(ops, len(segs), a_trans, b_trans, out, m, n))
    if keep[-1] is not None:
        split_k = 1              # (the form never cuts K: few row panels get narrower column tiles instead)
    cs = None
    if a_colsum is not None and split_k > 1 and (ops[0].flags & L.GEMM_FLAG_WIDE_WGRAD) and COLSUM_IN_WGRAD["enabled"] and m % 4 == 0:
        cs = torch.empty(m, dtype=torch.float32, device=out.device)
        ops[0].a_colsum = cs.data_ptr()
        a_colsum.append(cs)
    ws = (torch.empty(split_k * (m * n + (m if cs is not None else 0)), dtype=torch.float32, device=out.device)
          if split_k > 1 else None)
    partial = None
    if rowdot is not None:
        tiles = lib.plnlp_gemm_rowdot_tiles(m, n) if (keep[-1] is not None and FUSE_HEAD_FORWARD["enabled"]) else 0
        rw = _f32c(rowdot[0].reshape(-1))
        if tiles > 0 and rw.data_ptr() % 16 == 0 and rw.numel() == n:
            if epilogue is None:
                epilogue = L.Epilogue()
            partial = torch.empty(tiles, m, dtype=torch.float32, device=out.device)
            epilogue.flags |= L.EPI_ROWDOT
            epilogue.rowdot_w, epilogue.rowdot_out, epilogue.rowdot_ld = rw.data_ptr(), partial.data_ptr(), m
            keep.append(rw)
    rc = lib.plnlp_gemm_f32(ops, len(segs), int(a_trans), int(b_trans), out.data_ptr(), _ld(out), m, n,
                            C.byref(epilogue) if epilogue is not None else None, split_k,
                            L.ptr(ws), 0 if ws is None else ws.numel(), L.stream_ptr())
    L.check(rc, "plnlp_gemm_f32")
    if rowdot is not None:
        if partial is None:
            return out, None
        s_out = torch.empty(m, 1, dtype=torch.float32, device=out.device)
        L.check(lib.plnlp_rowdot_finish_f32(partial.data_ptr(), m, partial.shape[0], m, L.ptr(rowdot[1]), s_out.data_ptr(),
                                            L.stream_ptr()), "plnlp_rowdot_finish_f32")
        return out, s_out
    return out


def wgrad_pair(dz: torch.Tensor, x1: torch.Tensor, x2: torch.Tensor, rows: Optional[torch.Tensor] = None,
               x1_compact: bool = False, a_colsum: Optional[list] = None):
    """(dW1, dW2) = dz^T x1, dz^T x2 in one split-K GEMM that reads dz once (gemm_pair); two products when
    the seam would cut a 128-column tile.
    rows (int32 [K]): dz holds only those rows of a row-sparse gradient; x1 / x2 are read at
    rows[j] (gathered inside the GEMM's loader); x1_compact: x1 already holds just those rows.
    a_colsum: see gemm."""
    out = gemm_pair(dz, x1, x2, True, rows=rows, rows_on=2 if (x1_compact and rows is not None) else 3, a_colsum=a_colsum)
    if out is None:
        return (gemm([(dz, x1)], True, False, b_index=None if x1_compact else rows, a_colsum=a_colsum),
                gemm([(dz, x2)], True, False, b_index=rows))
    return out


# the bias gradient (column sums of dz) out of the whole-block weight-gradient kernel, which stages every row of dz anyway
# (csrc/gemm_wgw.hip, plnlp_gemm_operand.a_colsum), instead of a second pass over dz on the side stream (SideColsum).
# off = that pass everywhere (A/B runs; another summation order, both deterministic)
COLSUM_IN_WGRAD = {"enabled": True}


def dgrad_pair(dz: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, out1: Optional[torch.Tensor] = None):
    """(dz @ w1, dz @ w2) for two weights stored [out, in] (nn.Linear layout): one launch, dz read once,
    no concatenated copy of the weights (gemm_pair); the concatenated form when the seam does not fit"""
    out = gemm_pair(dz, w1, w2, False, out1=out1)
    if out is None:
        return gemm_split_out(dz, torch.cat([w1, w2], dim=1), w1.shape[1], out1=out1)
    return out


def gemm_pair(a: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor, a_trans: bool, out1: Optional[torch.Tensor] = None,
              rows: Optional[torch.Tensor] = None, rows_on: int = 3, a_colsum: Optional[list] = None):
    """(c1, c2) = op(a) @ b1, op(a) @ b2 in ONE launch (plnlp_gemm_pair_f32): `a` is read once, B comes
    from the two buffers as they are (no concatenated copy) and the two results are two contiguous
    tensors (no strided halves to copy out).  b1 / b2: [K, n1] / [K, n2] (row-contiguous).
      a_trans=False: a [M, K]   -- the two data gradients of SAGEConv  [gx | gagg] = dz [Wr | Wl]
      a_trans=True : a [K, M]   -- its two weight gradients  [dWl | dWr] = dz^T [agg | x]  (split-K;
                                  rows (int32 [K]): b1 / b2 are read at rows[j], gathered in the loader;
                                  rows_on = 2: only b2 is gathered, b1 is already compact)
    None when the seam would cut a 128-column tile (the caller falls back to two products)."""
    n1, n2 = b1.shape[1], b2.shape[1]
    if n1 % 128 != 0:
        return None
    lib = L.load()
    L.require_device(a, b1, b2, rows)
    a, b1, b2 = _f32c(a), _f32c(b1), _f32c(b2)
    k, m = (a.shape[0], a.shape[1]) if a_trans else (a.shape[1], a.shape[0])
    n = n1 + n2
    ops = (L.GemmOperand * 1)()
    ops[0].math = _gemm_math()
    ops[0].a, ops[0].lda, ops[0].b, ops[0].ldb, ops[0].k = a.data_ptr(), _ld(a), b1.data_ptr(), _ld(b1), k
    if rows is not None:
        assert rows.dtype == torch.int32 and rows.numel() == k and a_trans
        ops[0].b_index = rows.data_ptr()
    else:
        assert b1.shape[0] == k and b2.shape[0] == k, (a.shape, b1.shape, b2.shape)
    if rows is not None and rows_on == 2:
        assert b1.shape[0] == k
    ktiles = (k + 31) // 32
    split_k = max(1, min(_pick_split_k(m, n, ktiles), ktiles)) if a_trans else 1
    if a_trans and GEMM_WIDE_WGRAD["enabled"]:      # (the wide form is taken by the launch exactly when K is cut its way)
        wide = lib.plnlp_gemm_wide_wgrad_slices(ops, 1, 1, 0, m, n, b2.data_ptr(), _ld(b2), n1, int(rows_on))
        if wide:
            ops[0].flags |= L.GEMM_FLAG_WIDE_WGRAD
            split_k = wide
    c1 = out1 if out1 is not None else torch.empty(m, n1, dtype=torch.float32, device=a.device)
    assert c1.shape == (m, n1) and c1.is_contiguous()
    c2 = torch.empty(m, n2, dtype=torch.float32, device=a.device)
    lent = _lend_b_terms(ops, 1, a_trans, False, c1, m, n, out2=c2, n_split=n1)
    cs = None
    if a_colsum is not None and split_k > 1 and (ops[0].flags & L.GEMM_FLAG_WIDE_WGRAD) and COLSUM_IN_WGRAD["enabled"] and m % 4 == 0:
        cs = torch.empty(m, dtype=torch.float32, device=a.device)          # (see gemm: the bias gradient out of the same launch)
        ops[0].a_colsum = cs.data_ptr()
        a_colsum.append(cs)
    ws = (torch.empty(split_k * (m * n + (m if cs is not None else 0)), dtype=torch.float32, device=a.device)
          if split_k > 1 else None)
    L.check(lib.plnlp_gemm_pair_f32(ops, b2.data_ptr(), _ld(b2), n1, int(rows_on), int(a_trans), 0, c1.data_ptr(), _ld(c1),
                                    c2.data_ptr(), _ld(c2), n1, m, n, split_k, L.ptr(ws),
                                    0 if ws is None else ws.numel(), L.stream_ptr()), "plnlp_gemm_pair_f32")
    return c1, c2


def gemm_split_out(a: torch.Tensor, b: torch.Tensor, n_split: int, b_trans: bool = False,
                   out1: Optional[torch.Tensor] = None):
    """(a @ op(b)) with the result columns split into two contiguous tensors
    (plnlp_gemm_split_out_f32): returns (out[:, :n_split], out[:, n_split:])."""
    lib = L.load()
    L.require_device(a, b)
    a, b = _f32c(a), _f32c(b)
    m, k = a.shape
    n = b.shape[0] if b_trans else b.shape[1]
    ops = (L.GemmOperand * 1)()
    ops[0].math = _gemm_math()
    ops[0].a, ops[0].lda, ops[0].b, ops[0].ldb, ops[0].k = a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), k
    c1 = out1 if out1 is not None else torch.empty(m, n_split, dtype=torch.float32, device=a.device)
    assert c1.shape == (m, n_split) and c1.is_contiguous()
    c2 = torch.empty(m, n - n_split, dtype=torch.float32, device=a.device)
    lent = _lend_b_terms(ops, 1, False, b_trans, c1, m, n, out2=c2, n_split=n_split)
    L.check(lib.plnlp_gemm_split_out_f32(ops, 1, 0, int(b_trans), c1.data_ptr(), _ld(c1), c2.data_ptr(), _ld(c2),
                                         n_split, m, n, None, L.stream_ptr()), "plnlp_gemm_split_out_f32")
    return c1, c2


def colsum(x: torch.Tensor, scale: float = 1.0, row_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[f] = scale * sum_r (row_weight[r] *) x[r, f]"""
    lib = L.load()
    L.require_device(x, row_weight)
    x = _f32c(x)
    n, f = x.shape
    if row_weight is not None:
        row_weight = _f32c(row_weight.reshape(-1))
        assert row_weight.numel() == n
    out = torch.empty(f, dtype=torch.float32, device=x.device)
    nws = lib.plnlp_colsum_workspace_floats(n, f)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    L.check(lib.plnlp_colsum_f32(x.data_ptr(), _ld(x), n, f, L.ptr(row_weight), scale, out.data_ptr(),
                                 ws.data_ptr(), nws, L.stream_ptr()), "plnlp_colsum_f32")
    return out


# a conv's bias gradient (the column sums of dz: a 22 us bandwidth-bound pair of launches on collab) beside the weight-gradient
# GEMM and the transposed aggregation instead of between them: nothing on the main stream needs it before the optimiser
COLSUM_SIDE_STREAM = {"enabled": True, "min_rows": 16384}


class SideColsum:
    """colsum(x) started on the device's second side stream; `join()` -- to be called AFTER the launches it should overlap
    have been enqueued and before the backward function returns -- makes the main stream wait for it and hands out the
    result.  The buffers are allocated on the main stream (they outlive the side kernels through the join), x is kept
    alive by the caller until join()."""

    def __init__(self, x: torch.Tensor):
        self.side = None
        if (not COLSUM_SIDE_STREAM["enabled"] or not x.is_cuda or x.shape[0] < COLSUM_SIDE_STREAM["min_rows"]
                or _step_scalars["active"] is not None):
            self.out = colsum(x)
            return
        lib = L.load()
        x = _f32c(x)
        n, f = x.shape
        self.out = torch.empty(f, dtype=torch.float32, device=x.device)
        nws = lib.plnlp_colsum_workspace_floats(n, f)
        self.ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        main = torch.cuda.current_stream(x.device)
        self.side = side_stream(x.device, 1)
        self.side.wait_stream(main)                       # x is the output of work queued on the main stream
        with torch.cuda.stream(self.side):
            L.check(lib.plnlp_colsum_f32(x.data_ptr(), _ld(x), n, f, None, 1.0, self.out.data_ptr(), self.ws.data_ptr(), nws,
                                         L.stream_ptr()), "plnlp_colsum_f32")
            self.done = torch.cuda.Event()
            self.done.record(self.side)
        self.x = x

    def join(self) -> torch.Tensor:
        if self.side is not None:
            torch.cuda.current_stream(self.out.device).wait_event(self.done)
            self.side = self.x = self.ws = None
        return self.out


def matvec(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = <x[r,:], w> + bias  (plnlp_matvec_f32)"""
    lib = L.load()
    L.require_device(x, w, bias)
    x, w = _f32c(x), _f32c(w.reshape(-1))
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    L.check(lib.plnlp_matvec_f32(x.data_ptr(), _ld(x), x.shape[0], x.shape[1], w.data_ptr(), L.ptr(bias),
                                 out.data_ptr(), L.stream_ptr()), "plnlp_matvec_f32")
    return out


def outer(g: torch.Tensor, w: torch.Tensor, epilogue: Optional[L.Epilogue] = None) -> torch.Tensor:
    """dx[r,:] = EPI(g[r] * w[:])  (plnlp_outer_f32)"""
    lib = L.load()
    L.require_device(g, w)
    g, w = _f32c(g.reshape(-1)), _f32c(w.reshape(-1))
    dx = torch.empty(g.numel(), w.numel(), dtype=torch.float32, device=g.device)
    L.check(lib.plnlp_outer_f32(g.data_ptr(), w.data_ptr(), g.numel(), w.numel(), dx.data_ptr(), _ld(dx),
                                C.byref(epilogue) if epilogue is not None else None, L.stream_ptr()),
            "plnlp_outer_f32")
    return dx


FUSE_HEAD_BACKWARD = {"enabled": True}


def mlp_head_backward(a: torch.Tensor, g: torch.Tensor, w: torch.Tensor, gate_scale: float):
    """(dz, dw, dbp, db) of a 1-output linear head behind a relu / dropout hidden activation `a`: the first three from ONE
    pass over `a` (plnlp_mlp_head_backward_f32), db = the column sum of g -- or None where that form does not apply (width,
    alignment): the caller then runs the separate passes.  dz [rows, feat], dw [1, feat], dbp [feat], db [1]."""
    lib = L.load()
    L.require_device(a, g, w)
    n, f = a.shape
    if not FUSE_HEAD_BACKWARD["enabled"] or n == 0 or f % 4 != 0 or f > 1024:
        return None
    a, g, w = _f32c(a), _f32c(g.reshape(-1)), _f32c(w.reshape(-1))
    if _ld(a) % 4 != 0 or a.data_ptr() % 16 != 0 or w.data_ptr() % 16 != 0:
        return None
    dz = torch.empty(n, f, dtype=torch.float32, device=a.device)
    sums = torch.empty(2 * f, dtype=torch.float32, device=a.device)
    nws = lib.plnlp_mlp_head_backward_workspace_floats(n, f)
    ws = torch.empty(nws, dtype=torch.float32, device=a.device)
    L.check(lib.plnlp_mlp_head_backward_f32(a.data_ptr(), _ld(a), g.data_ptr(), w.data_ptr(), float(gate_scale), n, f,
                                            dz.data_ptr(), _ld(dz), sums.data_ptr(), ws.data_ptr(), nws, L.stream_ptr()),
            "plnlp_mlp_head_backward_f32")
    return dz, sums[:f].reshape(1, f), sums[f:2 * f], colsum(g.reshape(-1, 1))


def gate(g: torch.Tensor, y: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu+dropout backward: g * scale where the forward output y > 0, else 0."""
    lib = L.load()
    L.require_device(g, y)
    g, y = g.contiguous(), y.contiguous()
    if out is None:
        out = torch.empty_like(g)
    L.check(lib.plnlp_gate_f32(g.data_ptr(), y.data_ptr(), scale, out.data_ptr(), g.numel(), L.stream_ptr()),
            "plnlp_gate_f32")
    return out


def dropout(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    lib = L.load()
    L.require_device(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(lib.plnlp_dropout_f32(x.data_ptr(), y.data_ptr(), x.shape[0], x.shape[1], p,
                                  seed & 0xFFFFFFFFFFFFFFFF, L.stream_ptr()), "plnlp_dropout_f32")
    return y


def _edge_idx(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return t.contiguous()


def edge_dot_fwd(h: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    L.require_device(h, src, dst)
    h, src, dst = _f32c(h), _edge_idx(src), _edge_idx(dst)
    out = torch.empty(src.numel(), dtype=torch.float32, device=h.device)
    L.check(lib.plnlp_edge_dot_fwd_f32(h.data_ptr(), _ld(h), h.shape[0], src.data_ptr(), dst.data_ptr(),
                                       src.numel(), h.shape[1], out.data_ptr(), L.stream_ptr()),
            "plnlp_edge_dot_fwd_f32")
    return out


def edge_hadamard_fwd(h: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    L.require_device(h, src, dst)
    h, src, dst = _f32c(h), _edge_idx(src), _edge_idx(dst)
    out = torch.empty(src.numel(), h.shape[1], dtype=torch.float32, device=h.device)
    _apply_edge_segment()            # ('noslab' also turns off this launch's XCD-pinned column slabs)
    L.check(lib.plnlp_edge_hadamard_fwd_f32(h.data_ptr(), _ld(h), h.shape[0], src.data_ptr(), dst.data_ptr(),
                                            src.numel(), h.shape[1], out.data_ptr(), _ld(out), L.stream_ptr()),
            "plnlp_edge_hadamard_fwd_f32")
    return out


class Incidence:
    """Node-sorted incidence list of one edge batch: for every node, the batch edges
    touching it and, per item, the OTHER endpoint (plnlp_incidence_build: one keys-only
    radix sort of unique (node, item) keys on the device, no host sync; the order -- hence
    every bit of the reduced gradient -- is reproducible).  Doubles as the CSR the
    aggregation kernel walks: row = node, col = other endpoint, weight = g[edge]."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, n_nodes: int):
        lib = L.load()
        L.require_device(src, dst)
        src, dst = _edge_idx(src), _edge_idx(dst)
        e = src.numel()
        dev = src.device
        self.n_nodes = self.n_rows = self.n_cols = int(n_nodes)
        self.item_edge = torch.empty(2 * e, dtype=torch.int32, device=dev)
        self.item_other = torch.empty(2 * e, dtype=torch.int32, device=dev)
        self.seg_ptr = torch.empty(n_nodes + 1, dtype=torch.int64, device=dev)
        self.rowptr, self.col = self.seg_ptr, self.item_other
        self.val = None            # set to the per-edge gradient g; indexed through val_index
        self.val_index = self.item_edge
        self._split = None
        keys = torch.empty(2, max(2 * e, 1), dtype=torch.int64, device=dev)
        tbytes = lib.plnlp_incidence_temp_bytes(e)
        temp = torch.empty(max(tbytes, 8), dtype=torch.uint8, device=dev)
        L.check(lib.plnlp_incidence_build(src.data_ptr(), dst.data_ptr(), e, n_nodes, keys[0].data_ptr(),
                                          keys[1].data_ptr(), temp.data_ptr(), temp.numel(),
                                          self.item_edge.data_ptr(), self.item_other.data_ptr(),
                                          self.seg_ptr.data_ptr(), L.stream_ptr()), "plnlp_incidence_build")

    def row_split(self, threshold: int):
        """hot nodes of the batch (tables built on the device, upper-bound sizes)"""
        if self._split is None:
            from .graph import RowSplit
            self._split = RowSplit(self.seg_ptr, self.item_edge.numel(), threshold)
        return self._split

    def compact(self, count_host: Optional[torch.Tensor] = None) -> "CompactIncidence":
        """the same lists over the TOUCHED nodes only (plnlp_compact_rows)"""
        return CompactIncidence(self, count_host)


_pinned_counts = {"ring": None, "next": 0}


def _pinned_count_buffer() -> torch.Tensor:
    """a pinned int64 for reading a device-side count back, from a small ring (allocating pinned
    memory per step would cost more than the copy; a ring so that a few lists can be in flight)"""
    if _pinned_counts["ring"] is None:
        _pinned_counts["ring"] = torch.empty(64, dtype=torch.int64, pin_memory=True)
    i = _pinned_counts["next"]
    _pinned_counts["next"] = (i + 1) % 64
    return _pinned_counts["ring"][i:i + 1]


class CompactIncidence:
    """Incidence restricted to the nodes the batch touches, in increasing node order: row i
    belongs to node rows[i]; node_map[n] = i or -1.  Only these rows of the gathered matrix
    receive a gradient.

    The number of touched nodes T sizes the launches of the backward pass, so it must reach the
    host: the one host read-back of a training step.  It is started asynchronously here -- the
    lists depend only on the batch's edges, so the trainer builds them BEFORE the encoder's forward
    pass -- and awaited on first use (`count`, `rows`, `rowptr`), by which time the GPU has long
    passed it: the host waits, the GPU does not idle."""

    def __init__(self, inc: Incidence, count_host: Optional[torch.Tensor] = None):
        """count_host (a pinned int64 [1] the caller owns): the lists are being built inside a hipGraph capture
        (plnlp_amd/capture.py) -- the count is copied there by a node of the graph, no event is recorded here, and
        the caller sets `_count` after each replay"""
        lib = L.load()
        n = inc.n_nodes
        dev = inc.seg_ptr.device
        self._rows_cap = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        self.node_map = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        self._rowptr_cap = torch.empty(n + 1, dtype=torch.int64, device=dev)
        self._count_dev = torch.empty(1, dtype=torch.int64, device=dev)
        ws = torch.empty(lib.plnlp_compact_rows_workspace(n), dtype=torch.int32, device=dev)
        L.check(lib.plnlp_compact_rows(inc.seg_ptr.data_ptr(), n, self._rows_cap.data_ptr(),
                                       self.node_map.data_ptr(), self._rowptr_cap.data_ptr(),
                                       self._count_dev.data_ptr(), ws.data_ptr(), L.stream_ptr()),
                "plnlp_compact_rows")
        self._host = _pinned_count_buffer() if count_host is None else count_host
        self._host.copy_(self._count_dev, non_blocking=True)
        self._ready = None
        if count_host is None:
            self._ready = torch.cuda.Event()
            self._ready.record()
        self._count = None
        self._base = inc                     # keeps the sorted lists (shared below) alive
        self.n_cols = self.n_nodes = n
        self.col = self.item_other = inc.item_other
        self.val_index = self.item_edge = inc.item_edge
        self.val = None
        self._split = None

    @property
    def count(self) -> int:
        if self._count is None:
            if self._ready is None:
                raise RuntimeError("a captured batch's touched-row count is set by the capture driver after each replay")
            self._ready.synchronize()
            self._count = int(self._host.item())
        return self._count

    @property
    def n_rows(self) -> int:
        """the count rounded up to a multiple of 32 (within the capacity): the lists continue with
        empty rows past the count, so consumers may work on this padded extent -- the weight-gradient
        GEMM then reduces over a whole number of K-tiles (no guarded tail launch)"""
        return min(self._rows_cap.numel(), (self.count + 31) // 32 * 32)

    @property
    def rows(self) -> torch.Tensor:
        return self._rows_cap[:self.n_rows]

    @property
    def rowptr(self) -> torch.Tensor:
        return self._rowptr_cap[:self.n_rows + 1]

    seg_ptr = rowptr

    def row_split(self, threshold: int):
        """hot nodes of the batch; built over the CAPACITY so it needs no count (rows past the count
        are empty: rowptr_c there repeats the end offset)"""
        if self._split is None:
            from .graph import RowSplit
            self._split = RowSplit(self._rowptr_cap, self.item_edge.numel(), threshold)
        return self._split

    def prepare_compact_columns(self) -> None:
        """item_other translated to COMPACT row ids (node_map[other]; every endpoint of the batch is a
        touched node, so all are >= 0): the column list of these same lists over a matrix that holds only
        the touched rows (a row-restricted encoder output).  One gather, built with the lists."""
        if getattr(self, "_other_c", None) is None:
            self._other_c = self.node_map.index_select(0, self.item_other.long())

    def compact_view(self) -> "_CompactCols":
        self.prepare_compact_columns()
        return _CompactCols(self)


class _CompactCols:
    """a CompactIncidence seen over the compact matrix: same rows / segments / weights, columns in
    compact row ids"""

    def __init__(self, ci: CompactIncidence):
        self._ci = ci
        self.col = self.item_other = ci._other_c
        self.val_index = self.item_edge = ci.item_edge
        self.val = None
        self.node_map = ci.node_map

    count = property(lambda self: self._ci.count)
    n_rows = property(lambda self: self._ci.n_rows)
    n_cols = property(lambda self: self._ci.n_rows)
    n_nodes = n_cols
    rows = property(lambda self: self._ci.rows)
    rowptr = property(lambda self: self._ci.rowptr)
    seg_ptr = rowptr

    def row_split(self, threshold: int):
        return self._ci.row_split(threshold)


class RowSparseGrad:
    """gradient of an [n_rows, F] matrix that is zero outside `rows`: values[i] is row rows[i] for
    i < count; entries past `count` are padding (row id 0, zero values)"""
    __slots__ = ("rows", "node_map", "values", "n_rows", "count")

    def __init__(self, rows, node_map, values, n_rows, count=None):
        self.rows, self.node_map, self.values, self.n_rows = rows, node_map, values, n_rows
        self.count = rows.numel() if count is None else int(count)

    def to_dense(self) -> torch.Tensor:
        out = torch.zeros(self.n_rows, self.values.shape[1], dtype=self.values.dtype, device=self.values.device)
        out[self.rows[:self.count].long()] = self.values[:self.count]
        return out


class SparseGradChannel:
    """Side channel between the backward of an edge scorer (producer) and the backward of the
    encoder's last conv (consumer) for ONE forward pass.

    A batch touches only some nodes, so the gradient of the encoder output is zero in every other
    row; autograd would carry it as a dense [N, F] matrix and the conv's backward would run its
    GEMMs and its transposed aggregation over rows of zeros.  With a channel the scorer's backward
    leaves a RowSparseGrad here and returns no gradient through autograd; the conv's backward picks
    it up and works on the touched rows only.  The dense result is identical (rows of exact zeros
    contribute nothing).  Valid only while the scorer is the sole consumer of the encoder output --
    a second consumer's (dense) gradient is added on top if autograd delivers one."""

    def __init__(self):
        self.grad: Optional[RowSparseGrad] = None

    def take(self) -> Optional[RowSparseGrad]:
        g, self.grad = self.grad, None
        return g


SPARSE_BACKWARD = {"enabled": True,
                   # use the channel when the batch can touch at most this fraction of the nodes in
                   # expectation (1 - exp(-endpoints / nodes)); a batch that touches everything (ddi)
                   # gains nothing from the indirection.  Measured on collab at 1/2/3/4/6/8x the batch
                   # (expectation 0.67 / 0.89 / 0.96 / 0.99 / ~1 / ~1; degree-biased positives touch fewer):
                   # row-sparse vs dense 2.52/3.15, 2.90/3.14, 3.10/3.18, 3.24/3.24, 3.41/3.38, 3.59/3.54 ms
                   "max_expected_fraction": 0.97}


def sparse_backward_pays(n_endpoints: int, n_nodes: int) -> bool:
    import math
    if not SPARSE_BACKWARD["enabled"] or n_nodes <= 0:
        return False
    return 1.0 - math.exp(-float(n_endpoints) / float(n_nodes)) <= SPARSE_BACKWARD["max_expected_fraction"]


def random_walk(graph, start: torch.Tensor, walk_length: int, seed: int) -> torch.Tensor:
    """torch_cluster.random_walk(row, col, start, walk_length) on the adjacency `graph`
    (plnlp_random_walk): int64 [S, walk_length + 1], column 0 = start."""
    lib = L.load()
    L.require_device(start, graph.col)
    start = _edge_idx(start)
    walks = torch.empty(start.numel(), walk_length + 1, dtype=torch.int64, device=start.device)
    L.check(lib.plnlp_random_walk(graph.rowptr.data_ptr(), graph.col.data_ptr() or graph.rowptr.data_ptr(),
                                  start.data_ptr(), start.numel(), walk_length, seed & 0xFFFFFFFFFFFFFFFF,
                                  walks.data_ptr(), L.stream_ptr()), "plnlp_random_walk")
    return walks


def rmat_thresholds(probs=(0.57, 0.19, 0.19, 0.05)):
    """a, a + b, a + b + c as fractions of 2^32: the integer thresholds of plnlp_rmat_edges' quadrant draw"""
    a, b, c, _ = probs
    return tuple(min(int(round(v * 4294967296.0)), 0xFFFFFFFF) for v in (a, a + b, a + b + c))


def rmat_edges(scale: int, n_nodes: int, edge_lo: int, n_edges: int, seed: int, device,
               probs=(0.57, 0.19, 0.19, 0.05), relabel: bool = True):
    """(rows, cols) int32 [n_edges]: edges [edge_lo, edge_lo + n_edges) of the R-MAT stream (plnlp_rmat_edges;
    BASELINE.json config 5).  A function of (scale, seed, probs, edge id) alone -- any rank replays any part."""
    lib = L.load()
    device = torch.device(device)
    rows = torch.empty(n_edges, dtype=torch.int32, device=device)
    cols = torch.empty(n_edges, dtype=torch.int32, device=device)
    L.require_device(rows, cols)
    t_a, t_ab, t_abc = rmat_thresholds(probs)
    L.check(lib.plnlp_rmat_edges(int(scale), int(n_nodes), int(edge_lo), int(n_edges), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                 t_a, t_ab, t_abc, int(bool(relabel)), rows.data_ptr(), cols.data_ptr(),
                                 L.stream_ptr()), "plnlp_rmat_edges")
    return rows, cols


def random_walk_pairs(graph, start: torch.Tensor, walk_length: int, seed: int):
    """main.py:241-253: pairs (start, j-th hop) for j = 1..L with weight 1/j, self pairs removed."""
    walk = random_walk(graph, start, walk_length, seed)
    pairs = torch.cat([walk[:, [0, j + 1]] for j in range(walk_length)], dim=0)
    weights = torch.cat([torch.full((walk.size(0),), 1.0 / (j + 1), device=walk.device)
                         for j in range(walk_length)], dim=0)
    keep = pairs[:, 0] != pairs[:, 1]
    return pairs[keep], weights[keep]


# the segment backward's long segments shared by the waves of a workgroup (csrc/edge_ops.hip::edge_segment_bwd_group_kernel: eight
# segments per workgroup of eight waves from 8 192 segments on, one segment per workgroup of four below): 'auto' = that rule,
# 'wave' = one wave per segment whatever its length (the round-5 form), 'group4' = groups of four instead of eight, 'noslab' = few
# segments without the XCD-pinned column slabs (A/B runs)
EDGE_SEGMENT = {"form": os.environ.get("PLNLP_EDGE_SEGMENT", "auto"), "applied": None}


_EDGE_SEGMENT_FORMS = {"auto": 0, "wave": 1, "group4": 2, "noslab": 3}


def _apply_edge_segment() -> None:
    form = EDGE_SEGMENT["form"]
    if form != EDGE_SEGMENT["applied"]:
        if form not in _EDGE_SEGMENT_FORMS:
            raise ValueError("PLNLP_EDGE_SEGMENT / ops.EDGE_SEGMENT['form'] = %r: one of %s" % (form, ", ".join(_EDGE_SEGMENT_FORMS)))
        L.load().plnlp_edge_segment_tuning(_EDGE_SEGMENT_FORMS[form])
        EDGE_SEGMENT["applied"] = form


def edge_segment_bwd(h: torch.Tensor, inc: Incidence, g: torch.Tensor,
                     out: Optional[torch.Tensor] = None,
                     epilogue: Optional[L.Epilogue] = None) -> torch.Tensor:
    """gh[n] = sum_{items of n} g[edge] (.) h[other]   (deterministic)."""
    lib = L.load()
    L.require_device(h, g)
    h, g = _f32c(h), _f32c(g)
    is_vec = g.dim() == 2
    if out is None:
        out = torch.empty(inc.n_rows, h.shape[1], dtype=torch.float32, device=h.device)
    if not is_vec:
        # scalar per-edge gradient (DOT): gh = S h with S[n, other] = g[edge] -- exactly the
        # weighted CSR aggregation, so it runs on K1 (incl. hot-node splitting)
        inc.val = g                       # weight of item = g[item_edge[item]] (val_index)
        try:
            return csr_aggregate(inc, h, "sum", True, out=out, epilogue=epilogue)
        finally:
            inc.val = None
    _apply_edge_segment()
    L.check(lib.plnlp_edge_segment_bwd_f32(
        h.data_ptr(), _ld(h), inc.seg_ptr.data_ptr(), None, inc.n_rows, inc.item_edge.data_ptr(),
        inc.item_other.data_ptr(), h.shape[1], g.data_ptr(), _ld(g) if is_vec else 0, int(is_vec),
        out.data_ptr(), _ld(out), C.byref(epilogue) if epilogue is not None else None, L.stream_ptr()),
        "plnlp_edge_segment_bwd_f32")
    return out


def edge_scatter_bwd(h: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, g: torch.Tensor,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """atomic form (summation order not fixed)."""
    lib = L.load()
    L.require_device(h, g, src, dst)
    h, g, src, dst = _f32c(h), _f32c(g), _edge_idx(src), _edge_idx(dst)
    is_vec = g.dim() == 2
    if out is None:
        out = torch.zeros_like(h)
    L.check(lib.plnlp_edge_scatter_bwd_f32(h.data_ptr(), _ld(h), src.data_ptr(), dst.data_ptr(), src.numel(),
                                           h.shape[1], g.data_ptr(), _ld(g) if is_vec else 0, int(is_vec),
                                           out.data_ptr(), _ld(out), L.stream_ptr()),
            "plnlp_edge_scatter_bwd_f32")
    return out


# one persistent zeroed word per use site (plnlp_pairwise_loss_tail_f32, plnlp_sqnorm_multi_sum_f32: the last
# workgroup finishes the reduction and puts the word back to 0) -- allocated once per device, also what a captured
# hipGraph keeps pointing at
_tail_words = {}
TAIL_LOSS, TAIL_SQNORM = 0, 1


def tail_counter(device, slot: int) -> int:
    """device pointer of the `slot`-th persistent zero word of (this device, the CURRENT stream), 64 bytes apart.
    include/plnlp_hip.h asks for one zeroed word per caller AND stream (last_workgroup()): two loss or clip-norm launches
    in flight on different streams of one device -- a second model on a side stream -- must not share a ticket counter.
    A captured graph keeps the word of the stream it was captured on."""
    key = torch.device(device)
    if key.index is None and key.type == "cuda":
        key = torch.device("cuda", torch.cuda.current_device())
    key = (key, L.stream_ptr() if key.type == "cuda" else 0)
    buf = _tail_words.get(key)
    if buf is None:
        buf = _tail_words[key] = torch.zeros(16 * 8, dtype=torch.int32, device=key[0])
    return buf.data_ptr() + 64 * slot


# the running sum an epoch keeps of loss * examples (plnlp/model.py:169), fed by the loss kernel itself while set:
# {"buf": float64 [1] device tensor, "weight": examples of the step being enqueued,
#  "fed": how many loss launches have taken it so far -- a caller compares before / after to learn whether the step it
#  enqueued went through the fused loss at all: a stock-torch loss does not, and is then added the reference's way)
LOSS_ACC = {"buf": None, "weight": 0.0, "fed": 0}


def pairwise_loss(kind: str, pos: torch.Tensor, neg: torch.Tensor, num_neg: int,
                  weight: Optional[torch.Tensor] = None, grad_scale: float = 1.0,
                  grad_out: Optional[torch.Tensor] = None):
    """returns (loss[1], gpos[B], gneg[B*k]); grad_out: a [B + B*k] buffer the two gradients are
    written into back to back (gpos / gneg are then views of it)"""
    lib = L.load()
    L.require_device(pos, neg, weight)
    pos = _f32c(pos.reshape(-1))
    neg = _f32c(neg.reshape(-1))
    b = pos.numel()
    assert neg.numel() == b * num_neg, (pos.shape, neg.shape, num_neg)
    if weight is not None:
        weight = _f32c(weight.reshape(-1))
        assert weight.numel() == b
    loss = torch.empty(1, dtype=torch.float32, device=pos.device)
    if grad_out is None:
        grad_out = torch.empty(b + b * num_neg, dtype=torch.float32, device=pos.device)
    assert grad_out.numel() == b + b * num_neg and grad_out.is_contiguous() and grad_out.dtype == torch.float32
    gpos, gneg = grad_out[:b], grad_out[b:]
    nws = lib.plnlp_loss_workspace_floats(b)
    ws = torch.empty(nws, dtype=torch.float32, device=pos.device)
    acc = LOSS_ACC["buf"]
    if acc is not None:
        assert acc.dtype == torch.float64 and acc.device == pos.device
        LOSS_ACC["fed"] += 1
    L.check(lib.plnlp_pairwise_loss_tail_f32(L.LOSS_KINDS[kind], pos.data_ptr(), neg.data_ptr(), L.ptr(weight), b,
                                             num_neg, grad_scale, loss.data_ptr(), gpos.data_ptr(), gneg.data_ptr(),
                                             ws.data_ptr(), nws, tail_counter(pos.device, TAIL_LOSS), L.ptr(acc),
                                             float(LOSS_ACC["weight"]), L.stream_ptr()),
            "plnlp_pairwise_loss_tail_f32")
    return loss, gpos, gneg


def sqnorm_into(tensors: Sequence[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """out[0] = sum_t ||t||^2 (fixed order) -- the squared total norm of a parameter group."""
    lib = L.load()
    counts = [lib.plnlp_sqnorm_partials(t.numel()) for t in tensors]
    total = sum(counts)
    part = torch.empty(max(total, 1), dtype=torch.float32, device=out.device)
    off = 0
    s = L.stream_ptr()
    for t in tensors:
        L.require_device(t)
        assert t.is_contiguous() and t.dtype == torch.float32
    if 0 < len(tensors) <= L.MULTI_MAX and total > 0:       # the usual case: one launch, the sum included
        ptrs = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        sizes = (C.c_int64 * len(tensors))(*[t.numel() for t in tensors])
        L.check(lib.plnlp_sqnorm_multi_sum_f32(ptrs, sizes, len(tensors), part.data_ptr(), total, out.data_ptr(),
                                               tail_counter(out.device, TAIL_SQNORM), s), "plnlp_sqnorm_multi_sum_f32")
        return out
    for lo in range(0, len(tensors), L.MULTI_MAX):          # one launch per <= 16 tensors
        chunk = tensors[lo:lo + L.MULTI_MAX]
        ptrs = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
        sizes = (C.c_int64 * len(chunk))(*[t.numel() for t in chunk])
        c = sum(counts[lo:lo + L.MULTI_MAX])
        L.check(lib.plnlp_sqnorm_multi_f32(ptrs, sizes, len(chunk), part.data_ptr() + 4 * off, c, s),
                "plnlp_sqnorm_multi_f32")
        off += c
    L.check(lib.plnlp_sum_partials_f32(part.data_ptr(), total, out.data_ptr(), 0, s), "plnlp_sum_partials_f32")
    return out


def adam_multi(entries, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False,
               grad_scale=1.0) -> None:
    """one fused clip + Adam launch per <= 16 tensors (plnlp_adam_multi_f32).
    entries: (param, grad, exp_avg, exp_avg_sq, step, sqnorm | None, max_norm)"""
    lib = L.load()
    s = L.stream_ptr()
    sc = _step_scalars["active"]        # a step being captured: lr and the bias corrections come from device memory
    for lo in range(0, len(entries), L.MULTI_MAX):
        chunk = entries[lo:lo + L.MULTI_MAX]
        arr = (L.AdamTensor * len(chunk))()
        for i, (p, g, m, v, step, sq, max_norm) in enumerate(chunk):
            L.require_device(p, g, m, v, sq)
            assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32
            arr[i].param, arr[i].grad, arr[i].exp_avg, arr[i].exp_avg_sq = (p.data_ptr(), g.data_ptr(), m.data_ptr(),
                                                                           v.data_ptr())
            arr[i].n, arr[i].step = p.numel(), int(step)
            arr[i].sqnorm = L.ptr(sq)
            arr[i].max_norm = float(max_norm)
            if sc is not None:
                arr[i].step_scalars = sc.adam_ptr()
        L.check(lib.plnlp_adam_multi_f32(arr, len(chunk), lr, beta1, beta2, eps, weight_decay, int(decoupled),
                                         grad_scale, s), "plnlp_adam_multi_f32")


def adam_step(param, grad, exp_avg, exp_avg_sq, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              decoupled=False, step=1, sqnorm=None, max_norm=0.0, grad_scale=1.0) -> None:
    lib = L.load()
    L.require_device(param, grad, exp_avg, exp_avg_sq)
    assert param.is_contiguous() and grad.is_contiguous()
    L.check(lib.plnlp_adam_step_f32(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                    param.numel(), lr, beta1, beta2, eps, weight_decay, int(decoupled), step,
                                    L.ptr(sqnorm), max_norm, grad_scale, L.stream_ptr()), "plnlp_adam_step_f32")


def clip_scale_(grad: torch.Tensor, sqnorm: torch.Tensor, max_norm: float) -> None:
    lib = L.load()
    L.require_device(grad, sqnorm)
    assert grad.is_contiguous()
    L.check(lib.plnlp_clip_scale_f32(grad.data_ptr(), grad.numel(), sqnorm.data_ptr(), max_norm, L.stream_ptr()),
            "plnlp_clip_scale_f32")


# ---------------------------------------------------------- autograd wrappers ----
class StepScalars:
    """The scalars of ONE training step that change from step to step while every pointer and size stays the same
    -- the learning rate, Adam's bias corrections, the dropout seeds -- in DEVICE memory (plnlp_epilogue.
    dropout_seed_ptr / adam_scalars, plnlp_adam_tensor.step_scalars).  While an instance is `active`, the layer
    wrappers hand the kernels these addresses instead of by-value numbers, so the step they enqueue can be captured
    in a hipGraph once and replayed (plnlp_amd/capture.py); `upload` writes the values of the next replay -- computed
    on the host exactly as the eager launchers compute them, so a replayed step is bit-identical to an eager one.
    Layout (4-byte words): [0..2] lr, 1 - beta1^t, sqrt(1 - beta2^t); [4 + 2k, 5 + 2k] dropout seed k (lo, hi)."""
    MAX_SEEDS = 14
    WORDS = 4 + 2 * MAX_SEEDS
    RING = 8

    def __init__(self, device):
        self.dev = torch.zeros(self.WORDS, dtype=torch.int32, device=device)
        self.host = torch.zeros(self.RING, self.WORDS, dtype=torch.int32, pin_memory=True)
        self._next = 0
        self.seed_slots = 0

    def adam_ptr(self) -> int:
        return self.dev.data_ptr()

    def new_seed_slot(self) -> int:
        if self.seed_slots >= self.MAX_SEEDS:
            raise RuntimeError("more dropout calls in one step than StepScalars.MAX_SEEDS")
        k = self.seed_slots
        self.seed_slots += 1
        return self.dev.data_ptr() + 4 * (4 + 2 * k)

    def upload(self, lr: float, beta1: float, beta2: float, step: int, seeds) -> None:
        """enqueue (current stream) the values of the next replay; `seeds`: one 64-bit seed per slot, in call order"""
        import numpy as np
        lib = L.load()
        row = self.host[self._next]
        self._next = (self._next + 1) % self.RING
        L.check(lib.plnlp_adam_step_scalars(float(lr), float(beta1), float(beta2), int(step), row.data_ptr()),
                "plnlp_adam_step_scalars")
        words = row.numpy()
        for k, sd in enumerate(seeds):
            words[4 + 2 * k] = np.uint32(sd & 0xFFFFFFFF).astype(np.int32)
            words[5 + 2 * k] = np.uint32((sd >> 32) & 0xFFFFFFFF).astype(np.int32)
        self.dev.copy_(row, non_blocking=True)


_step_scalars = {"active": None}


class _Act:
    """relu + dropout description of one layer call"""
    __slots__ = ("relu", "p", "seed", "gate_in_consumer", "seed_ptr")

    def __init__(self, relu: bool, p: float, training: bool):
        self.relu = bool(relu)
        self.p = float(p) if training else 0.0
        sc = _step_scalars["active"]
        if sc is not None and self.p > 0.0:
            # a step being captured: the seed of every replay is uploaded to device memory (StepScalars); the
            # host's seed stream is NOT advanced by the capture itself
            self.seed, self.seed_ptr = 0, sc.new_seed_slot()
        else:
            self.seed, self.seed_ptr = (next_seed() if self.p > 0.0 else 0), 0
        # True when the op that consumes this layer's output folds the relu/dropout
        # derivative into its own backward (EdgeDotFn with gate_scale): the incoming
        # gradient is then already d/dz and this layer must not gate it again
        self.gate_in_consumer = False

    @property
    def active(self):
        return self.relu or self.p > 0.0

    @property
    def scale(self):
        return 1.0 / (1.0 - self.p) if self.p > 0.0 else 1.0


def _act_backward(gy: torch.Tensor, y: torch.Tensor, act: _Act) -> torch.Tensor:
    """dz from dy for y = dropout(relu(z)): the kept, positive entries are exactly
    y > 0, so dz = dy * 1/(1-p) there and 0 elsewhere.  The reference never applies
    dropout without a preceding relu (layer.py:21-22,25-26,84-85); asserted."""
    if not act.active or act.gate_in_consumer:
        return gy
    assert act.relu, "dropout without relu is not used by the reference path"
    return gate(gy, y, act.scale)


_side_streams = {}


def side_stream(device, which: int = 0) -> "torch.cuda.Stream":
    """second HIP stream per device: bandwidth-bound backward kernels (transposed aggregation) run
    there next to the MFMA-bound weight-gradient GEMMs instead of after them.  which = 1: a further one (the
    embedding's update), so that it does not queue behind the next batch's index preparation"""
    key = (torch.device(device).index or 0, which, SIDE_STREAM_PRIORITY["value"])
    if key not in _side_streams:
        # high priority: HIP hands out hardware queues per priority class, so this stream does not end
        # up multiplexed onto the main stream's hardware queue once a process group has created its own
        # streams (measured: as the 7th normal-priority stream it ran strictly AFTER the main stream's
        # queued kernels, i.e. no overlap at all), and its short kernels are dispatched promptly
        # next to the long ones
        _side_streams[key] = torch.cuda.Stream(device=device, priority=SIDE_STREAM_PRIORITY["value"])
    return _side_streams[key]


SIDE_STREAM_PRIORITY = {"value": -1}      # (low / normal measured +0.5 %: profiles/r05_same_box_ab.txt)


class GradSink:
    """Where the gradient of an encoder's INPUT goes when the caller wants it early.

    Data-parallel training reduces the (large) embedding gradient over RCCL.  Handing it to
    autograd means it only becomes visible after the whole backward pass is queued, so the
    all-reduce cannot overlap anything.  With a sink, the first conv's backward computes the
    input gradient FIRST, writes it straight into `buffer` (which the caller has installed as
    `param.grad`), calls `on_ready()` -- the caller starts the asynchronous all-reduce there --
    and only then queues its weight-gradient GEMMs, which run while the reduction is in flight."""

    def __init__(self, buffer: Optional[torch.Tensor], on_ready=None, adam=None, like: Optional[torch.Tensor] = None):
        """buffer=None with like=<the parameter>: the buffer is allocated on first use (a backward that applies
        the fused update never touches it)"""
        self._buffer, self._like, self.on_ready = buffer, like, on_ready
        # adam = dict(param, exp_avg, exp_avg_sq, step, lr, betas, eps): the input IS a parameter that is neither
        # clipped nor reduced over ranks -- the backward may then apply its Adam step in the epilogue of the kernel
        # that finishes the gradient (PLNLP_EPI_ADAM) instead of writing the gradient out; it sets
        # `adam_applied` when it did (else the gradient is in `buffer` as usual)
        self.adam, self.adam_applied = adam, False

    @property
    def buffer(self) -> torch.Tensor:
        if self._buffer is None:
            self._buffer = torch.empty_like(self._like)
        return self._buffer

    @property
    def buffer_used(self) -> bool:
        """False: no backward asked for the buffer -- the gradient, if any, went through autograd as usual"""
        return self._buffer is not None


class AggregateFn(torch.autograd.Function):
    """torch_sparse.matmul(adj_t, x, reduce) with its autograd (Appendix A.3)."""

    @staticmethod
    def forward(ctx, x, graph: Graph, reduce: str, use_values: bool):
        ctx.graph, ctx.reduce, ctx.use_values = graph, reduce, use_values
        if reduce == "max":
            out, arg = csr_aggregate_max(graph, x, use_values)
            ctx.save_for_backward(arg)
            return out
        return csr_aggregate(graph, x, reduce, use_values)

    @staticmethod
    def backward(ctx, g):
        graph = ctx.graph
        if ctx.reduce == "max":
            (arg,) = ctx.saved_tensors
            return csr_aggregate_max_bwd(graph, g, arg, ctx.use_values), None, None, None
        gt = graph.t()
        if ctx.reduce == "mean" and not (ctx.use_values and graph.val is not None):
            gx = csr_aggregate(graph.t_mean(), g.contiguous(), "sum", True)
        else:
            gx = csr_aggregate(gt, g.contiguous(), "sum", ctx.use_values,
                               src_scale=graph.inv_degree() if ctx.reduce == "mean" else None)
        return gx, None, None, None


SPARSE_FORWARD = {"enabled": os.environ.get("PLNLP_SPARSE_FORWARD", "1") != "0"}
"""OutputRows -- the LAST conv of an encoder evaluated only at the rows the step reads.

The reference computes h = encoder(x, adj_t) for every node (model.py:150-151) and then reads h only at
the endpoints of the batch's edges (model.py:155-156); the other rows are never used and their gradient is
exactly zero.  When the scorer is fused (it gathers through an index map anyway) the last conv therefore
produces just the touched rows, as a compact [T, out] matrix in the order of `CompactIncidence.rows`:

  * SAGE: the mean aggregation walks only those CSR rows (row_index), the GEMM runs on T rows with the
    root operand x gathered in its loader (a_index); GCN: the weighted aggregation walks only those rows;
  * dropout masks are drawn at the ORIGINAL row positions (dropout_row_index), so every produced value is
    bit-identical to the corresponding row of the full matrix;
  * the backward is the row-sparse backward that already existed (same T rows), with the saved
    aggregate now compact.
Loss, every gradient and the update are those of the full-matrix step; on the collab-shaped workload
T ~ 0.55 N, on citation2 ~ 0.09 N (one layer of aggregation out of four nearly disappears)."""


def _compact_grad(out_rows, sg, gy, y, act: "_Act", n_rows: int):
    """the gradient of a row-restricted conv output as a RowSparseGrad over out_rows, already taken
    w.r.t. the pre-activation: from the channel (the fused scorers fold the activation derivative in
    when act.gate_in_consumer) and / or from autograd (a [T, out] tensor); None when there is none"""
    vals = None
    if sg is not None:
        vals = sg.values
    if gy is not None:
        vals = gy.contiguous() if vals is None else vals + gy
    if vals is None:
        return None
    if act.active and not act.gate_in_consumer:
        vals = _act_backward(vals, y, act)
    elif gy is not None and act.active and sg is not None:
        raise RuntimeError("a row-restricted conv output got a folded and an unfolded gradient at once")
    return RowSparseGrad(out_rows.rows, out_rows.node_map, vals, n_rows, out_rows.count)


class SAGEConvFn(torch.autograd.Function):
    """One SAGEConv (+ optional relu/dropout of BaseGNN.forward, layer.py:20-26):
        y = act( mean_agg(x) @ Wl^T + bl + x @ Wr^T )
    forward : K1 aggregate, then ONE concat-K MFMA GEMM with fused bias/relu/dropout.
    backward: gate, 2 wgrad (split-K), colsum, 2 dgrad, K2 aggregate over the
              transposed CSR accumulating into the root-weight gradient."""

    @staticmethod
    def forward(ctx, x, w_l, b_l, w_r, graph: Graph, act: _Act, in_act: Optional[_Act] = None,
                sink: Optional[GradSink] = None, channel: Optional[SparseGradChannel] = None, out_rows=None):
        """in_act: the relu/dropout that PRODUCED x (previous layer).  When given, backward
        returns the gradient w.r.t. that layer's pre-activation (its derivative rides in the
        epilogue of the last kernel that touches gx) and in_act.gate_in_consumer is set.
        sink: see GradSink (the input gradient is delivered there, autograd gets None).
        channel: the gradient of y may arrive row-sparse through it (see SparseGradChannel).
        out_rows (a CompactIncidence: rows, node_map): produce ONLY those rows of y, as a compact
        [len(rows), out] matrix (see OutputRows below); needs `channel`."""
        ctx.sink = sink
        ctx.channel = channel
        ctx.out_rows = out_rows
        if channel is not None:
            ctx.set_materialize_grads(False)
        x = _f32c(x)
        if out_rows is not None:
            assert channel is not None, "a row-restricted forward hands its gradient back through the channel"
            rows = out_rows.rows
            agg = csr_aggregate(graph, x, "mean", use_values=False, row_index=rows, out_map=out_rows.node_map)
            epi = L.make_epilogue(bias=b_l, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr, dropout_rows=rows)
            y = gemm([(agg, w_l), (x, w_r)], False, True, epilogue=epi, a_index=[None, rows])
        else:
            agg = csr_aggregate(graph, x, "mean", use_values=False)
            epi = L.make_epilogue(bias=b_l, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
            y = gemm([(agg, w_l), (x, w_r)], False, True, epilogue=epi)
        ctx.graph, ctx.act = graph, act
        ctx.in_act = in_act if (in_act is not None and in_act.active) else None
        if ctx.in_act is not None:
            ctx.in_act.gate_in_consumer = True
        ctx.save_for_backward(x, agg, w_l, w_r, y if act.active else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, agg, w_l, w_r, y = ctx.saved_tensors
        graph, act = ctx.graph, ctx.act
        sg = ctx.channel.take() if ctx.channel is not None else None
        if ctx.out_rows is not None:
            # the forward produced compact rows: the gradient is compact too (through the channel, or as
            # a [T, out] tensor from a consumer that used the compact matrix directly)
            sg = _compact_grad(ctx.out_rows, sg, gy, y, act, x.shape[0])
            if sg is None:
                return (None,) * 10
            return SAGEConvFn._backward_sparse(ctx, sg)
        if sg is not None and (gy is not None or (act.active and not act.gate_in_consumer)):
            # a dense gradient arrived as well (second consumer), or the activation derivative
            # was not folded in by the producer: fall back to the dense form
            gy = sg.to_dense() if gy is None else gy + sg.to_dense()
            sg = None
        if sg is not None:
            return SAGEConvFn._backward_sparse(ctx, sg)
        if gy is None:
            return (None,) * 10
        dz = _act_backward(gy.contiguous(), y, act)
        need = ctx.needs_input_grad
        gx = gwl = gbl = gwr = None
        sink = ctx.sink
        if need[0]:
            # both data gradients in ONE GEMM: [gx | gagg] = dz @ [Wr | Wl]  (dz read once)
            cin = w_r.shape[1]
            gx, gagg = dgrad_pair(dz, w_r, w_l, out1=sink.buffer if sink is not None else None)
            ia = ctx.in_act
            epi = L.make_epilogue(accumulate=True, gate=x if ia is not None else None,
                                  gate_scale=ia.scale if ia is not None else 1.0)
            csr_aggregate(graph.t_mean(), gagg, "sum", use_values=True, out=gx, epilogue=epi)
            if sink is not None and sink.on_ready is not None:
                sink.on_ready()          # e.g. start the all-reduce; the GEMMs below overlap it
            if sink is not None:
                gx = None
        if need[1] and need[3]:
            gwl, gwr = wgrad_pair(dz, agg, x)             # [dWl | dWr] = dz^T [agg | x], dz read once
        else:
            if need[1]:
                gwl = gemm([(dz, agg)], True, False)
            if need[3]:
                gwr = gemm([(dz, x)], True, False)
        if need[2]:
            gbl = colsum(dz)
        return gx, gwl, gbl, gwr, None, None, None, None, None, None

    @staticmethod
    def _backward_sparse(ctx, sg: RowSparseGrad):
        """dz is zero outside sg.rows (T rows): the data-gradient GEMM runs on T rows, the
        transposed aggregation gathers only mapped source rows and adds the root term through
        the map, the weight-gradient GEMM reduces over T rows with [agg | x] gathered in its
        loader.  Same sums as the dense form minus exact zeros."""
        x, agg, w_l, w_r, y = ctx.saved_tensors
        graph = ctx.graph
        dz = sg.values                                     # [T, out], already d/d(pre-activation)
        need = ctx.needs_input_grad
        gx = gwl = gbl = gwr = None
        sink = ctx.sink
        if dz.shape[0] == 0:
            zero = lambda t: torch.zeros_like(t)
            if need[0]:
                if sink is not None:
                    sink.buffer.zero_()
                    if sink.on_ready is not None:
                        sink.on_ready()
                else:
                    gx = zero(x)
            return (gx, zero(w_l) if need[1] else None, torch.zeros(w_l.shape[0], device=x.device) if need[2] else None,
                    zero(w_r) if need[3] else None, None, None, None, None, None, None)
        compact_fwd = ctx.out_rows is not None          # agg holds only the rows sg.rows
        if need[0]:
            cin = w_r.shape[1]
            gx_c, gagg_c = dgrad_pair(dz, w_r, w_l)
            ia = ctx.in_act
            epi = L.make_epilogue(addend=gx_c, addend_index=sg.node_map, gate=x if ia is not None else None,
                                  gate_scale=ia.scale if ia is not None else 1.0)
            fuse_adam = (sink is not None and sink.adam is not None and ia is None and x.is_contiguous()
                         and sink.adam["param"].data_ptr() == x.data_ptr() and sink.adam["param"].shape == x.shape
                         and _vector_path(gagg_c, x, cin))
            if fuse_adam:
                # the input is the embedding table and nothing but Adam consumes its gradient: update it in the
                # epilogue of the aggregation that finishes that gradient (no 242 MB gradient written and read
                # back).  The weight gradients read the OLD table, so they go first.
                # the bias gradient (column sums of dz) comes out of the weight-gradient launch where that is the whole-block
                # kernel (`cs` non-empty), else from its own pass beside the aggregation below
                cs = [] if need[2] else None
                if need[1] and need[3]:
                    gwl, gwr = wgrad_pair(dz, agg, x, rows=sg.rows, x1_compact=compact_fwd, a_colsum=cs)
                else:
                    if need[1]:
                        gwl = gemm([(dz, agg)], True, False, b_index=None if compact_fwd else sg.rows, a_colsum=cs)
                    if need[3]:
                        gwr = gemm([(dz, x)], True, False, b_index=sg.rows, a_colsum=cs if not need[1] else None)
                bias_sum = SideColsum(dz) if (need[2] and not cs) else None
                ad = sink.adam
                sc = _step_scalars["active"]
                epi = L.make_epilogue(addend=gx_c, addend_index=sg.node_map,
                                      adam=(ad["exp_avg"], ad["exp_avg_sq"], ad["step"], ad["lr"], ad["betas"][0],
                                            ad["betas"][1], ad["eps"]),
                                      adam_scalars_ptr=sc.adam_ptr() if sc is not None else 0)
                csr_aggregate(graph.t_mean(), gagg_c, "sum", use_values=True, src_map=sg.node_map, out=x.data,
                              epilogue=epi)
                sink.adam_applied = True
                if need[2]:
                    gbl = cs[0] if cs else bias_sum.join()
                return None, gwl, gbl, gwr, None, None, None, None, None, None
            out = sink.buffer if sink is not None else torch.empty(x.shape[0], cin, dtype=torch.float32,
                                                                   device=x.device)
            csr_aggregate(graph.t_mean(), gagg_c, "sum", use_values=True, src_map=sg.node_map, out=out,
                          epilogue=epi)
            if sink is not None and sink.on_ready is not None:
                sink.on_ready()
            if sink is None:
                gx = out
        cs = [] if need[2] else None
        if need[1] and need[3]:
            gwl, gwr = wgrad_pair(dz, agg, x, rows=sg.rows, x1_compact=compact_fwd, a_colsum=cs)
        else:
            if need[1]:
                gwl = gemm([(dz, agg)], True, False, b_index=None if compact_fwd else sg.rows, a_colsum=cs)
            if need[3]:
                gwr = gemm([(dz, x)], True, False, b_index=sg.rows, a_colsum=cs if not need[1] else None)
        if need[2]:
            gbl = cs[0] if cs else colsum(dz)
        return gx, gwl, gbl, gwr, None, None, None, None, None, None


def _root_map(graph: Graph, row_lo: int) -> torch.Tensor:
    """int32 [n_cols]: position of source row r inside the block of destination rows this rank owns
    (r - row_lo), or -1 outside it -- where the root-path gradient of a row block joins the gradient
    of all source rows (addend_index of the transposed aggregation's epilogue).  Cached on the graph."""
    cache = getattr(graph, "_root_maps", None)
    if cache is None:
        cache = graph._root_maps = {}
    if row_lo not in cache:
        m = torch.full((graph.n_cols,), -1, dtype=torch.int32, device=graph.device)
        hi = min(graph.n_cols, row_lo + graph.n_rows)
        m[row_lo:hi] = torch.arange(hi - row_lo, dtype=torch.int32, device=graph.device)
        cache[row_lo] = m
    return cache[row_lo]


class SAGEConvBlockFn(torch.autograd.Function):
    """SAGEConv on ONE destination-row block of a row-sharded encoder (plnlp_amd/shard.py):
        y[S, out] = act( mean_agg_block(x_full) @ Wl^T + bl + x_full[row_lo : row_lo + S] @ Wr^T )
    `graph` is the rank's CSR slice (S rows, sources anywhere in x_full).  Same kernels as SAGEConvFn;
    the gradient of x_full is the PARTIAL sum over this block's rows -- all n_cols rows of it: the
    transposed aggregation writes every source row and adds the root-path term where the source row is
    one of the block's own (indexed addend epilogue).  The caller reduce-scatters it to the row owners."""

    @staticmethod
    def forward(ctx, x, w_l, b_l, w_r, graph: Graph, act: _Act, row_lo: int):
        x = _f32c(x)
        assert x.shape[0] == graph.n_cols, (x.shape, graph)
        s = graph.n_rows
        agg = csr_aggregate(graph, x, "mean", use_values=False)
        x_root = x[row_lo:row_lo + s]
        epi = L.make_epilogue(bias=b_l, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
        y = gemm([(agg, w_l), (x_root, w_r)], False, True, epilogue=epi)
        ctx.graph, ctx.act, ctx.row_lo = graph, act, int(row_lo)
        ctx.save_for_backward(x, agg, w_l, w_r, y if act.active else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, agg, w_l, w_r, y = ctx.saved_tensors
        graph, act, row_lo = ctx.graph, ctx.act, ctx.row_lo
        s = graph.n_rows
        x_root = x[row_lo:row_lo + s]
        dz = _act_backward(gy.contiguous(), y, act)
        need = ctx.needs_input_grad
        gx = gwl = gbl = gwr = None
        if need[0]:
            cin = w_r.shape[1]
            gx_root, gagg = dgrad_pair(dz, w_r, w_l)
            epi = L.make_epilogue(addend=gx_root, addend_index=_root_map(graph, row_lo))
            gx = csr_aggregate(graph.t_mean(), gagg, "sum", use_values=True, epilogue=epi)
        if need[1] and need[3]:
            gwl, gwr = wgrad_pair(dz, agg, x_root)
        else:
            if need[1]:
                gwl = gemm([(dz, agg)], True, False)
            if need[3]:
                gwr = gemm([(dz, x_root)], True, False)
        if need[2]:
            gbl = colsum(dz)
        return gx, gwl, gbl, gwr, None, None, None


class GCNConvBlockFn(torch.autograd.Function):
    """GCNConv(normalize=False) on ONE destination-row block of a row-sharded encoder:
        y[S, out] = act( (A_hat_block x_full) W^T + b )
    aggregate-FIRST (GCNConv itself transforms first): the transform of all N source rows would have to
    be repeated on every rank, the aggregate of the block's S rows is not -- same value up to fp32
    reassociation.  The gradient of x_full is the partial sum over this block's rows (all n_cols rows)."""

    @staticmethod
    def forward(ctx, x, w, b, graph: Graph, act: _Act):
        kin = x.shape[1]
        if kin % 4 != 0:
            xp, _ = _padded_operand(x if x.dtype == torch.float32 else x.float())
            wp = torch.zeros(w.shape[0], xp.shape[1], dtype=torch.float32, device=w.device)
            wp[:, :kin].copy_(w)
        else:
            xp, wp = _f32c(x), w
        assert xp.shape[0] == graph.n_cols, (xp.shape, graph)
        agg = csr_aggregate(graph, xp, "sum", use_values=True)
        epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
        y = gemm([(agg, wp)], False, True, epilogue=epi)
        ctx.graph, ctx.act, ctx.kin = graph, act, kin
        ctx.save_for_backward(agg, wp, y if act.active else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        agg, wp, y = ctx.saved_tensors
        graph, act, kin = ctx.graph, ctx.act, ctx.kin
        dz = _act_backward(gy.contiguous(), y, act)
        need = ctx.needs_input_grad
        gx = gw = gb = None
        if need[0]:
            gagg = gemm([(dz, wp)], False, False)
            gx = csr_aggregate(graph.t(), gagg, "sum", use_values=True)
            if gx.shape[1] != kin:
                gx = gx[:, :kin]
        if need[1]:
            gw = gemm([(dz, agg)], True, False)
            if gw.shape[1] != kin:
                gw = gw[:, :kin].contiguous()
        if need[2]:
            gb = colsum(dz)
        return gx, gw, gb, None, None


def padded_base(t: torch.Tensor) -> Optional[torch.Tensor]:
    """t = buf[:, :e] of a contiguous [n, ep] buffer with ep = e rounded up to 4 floats (BaseModel keeps an embedding table of
    an unaligned width that way: its rows then start on 16-byte boundaries and the two pad columns stay zero for ever --
    their gradient is zero, so Adam never moves them) -> that [n, ep] buffer as a tensor over the same storage; None when t
    is anything else (contiguous, another stride, an offset)."""
    if t.dim() != 2 or t.is_contiguous() or t.stride(1) != 1 or t.storage_offset() != 0:
        return None
    n, e = t.shape
    ep = t.stride(0)
    if ep != _pad_emb(e) or t.untyped_storage().nbytes() < n * ep * t.element_size():
        return None
    return torch.as_strided(t, (n, ep), (ep, 1))


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


# row width (in floats) an embedding table of an unaligned width is padded to: a multiple of this.  4 = whole 16-byte groups
# (the minimum the vector kernels need); 16 = whole 64-byte sectors: citation2's 50 columns become 64 instead of 52 -- every
# gathered row is then exactly two 128-byte lines instead of 2.6 on average, and [A emb | A x] is 64 + 128 = 192 columns with
# no pad block; same box, interleaved x 3: 23.47 -> 23.22 ms per step (profiles/r05_emb_pad_ab.txt)
def _emb_pad_floats(value: str) -> int:
    """PLNLP_EMB_PAD: whole 16-byte groups, at least one (the vector kernels and the GEMM loaders read rows 16 bytes at a time)"""
    g = int(value)
    if g < 4 or g % 4 != 0:
        raise ValueError(f"PLNLP_EMB_PAD={value!r}: the row granule of a padded embedding table is a multiple of 4 floats, >= 4")
    return g


EMB_PAD = {"floats": _emb_pad_floats(os.environ.get("PLNLP_EMB_PAD", "16"))}

# A padded table's gradient set on the parameter directly, in the padded layout (GCNInputConvFn.backward).  Only inside the
# trainer's own step (BaseModel._train_step_core opts in): there the optimiser is FusedAdam, which steps the padded buffer, and
# nothing else looks at the gradient.  Anywhere else the function returns the [N, e] gradient and autograd accumulates it --
# torch.autograd.grad, tensor hooks and foreign optimisers see what they expect.
_DIRECT_TABLE_GRAD = {"depth": 0}


class direct_table_grad:
    def __enter__(self):
        _DIRECT_TABLE_GRAD["depth"] += 1
        return self

    def __exit__(self, *exc):
        _DIRECT_TABLE_GRAD["depth"] -= 1
        return False


def _pad_emb(e: int) -> int:
    g = EMB_PAD["floats"]
    return (e + g - 1) // g * g


class ConcatFeatFn(torch.autograd.Function):
    """create_input_feat's torch.cat([emb.weight, data.x], -1) (model.py:104) into a persistent buffer
    whose row stride is padded to a multiple of 4 floats: the static feature block is written once,
    only the embedding block is refreshed per step, and the result is a 16-byte-aligned operand for
    the first GEMM (a 178-wide row would force the scalar-guarded loaders).  The pad columns are 0."""

    @staticmethod
    def forward(ctx, emb_weight, feats, buf):
        e = emb_weight.shape[1]
        buf[:, :e].copy_(emb_weight)
        ctx.e = e
        return buf[:, : e + feats.shape[1]].detach()

    @staticmethod
    def backward(ctx, g):
        return g[:, : ctx.e].contiguous(), None, None


def concat_features(emb_weight: torch.Tensor, feats: torch.Tensor, cache: dict, defer: bool = False) -> torch.Tensor:
    """defer: the consumer is a first GCNConv that normally takes the PARTS (GCNInputConvFn) and never reads the
    concatenated matrix -- the per-step copy of the embedding block (a 0.58 GB read + strided write on citation2,
    0.34-0.45 ms) is skipped and the result is marked stale; GCNConv.forward materialises it in the rare case it does
    need the matrix (materialize_concat)"""
    n, e, f = emb_weight.shape[0], emb_weight.shape[1], feats.shape[1]
    key = (feats.data_ptr(), feats._version, n, e, f)
    if cache.get("key") != key or cache.get("feats") is not feats:
        buf = torch.zeros(n, _pad4(e + f), dtype=torch.float32, device=emb_weight.device)
        buf[:, e:e + f].copy_(feats)
        cache.clear()
        cache.update(key=key, buf=buf, feats=feats)     # holding `feats` keeps its id / pointer from being reused
    if defer:
        out = cache["buf"][:, : e + f].detach()
        out._plnlp_stale = True
    else:
        out = ConcatFeatFn.apply(emb_weight, feats, cache["buf"])
        out._plnlp_padded = cache["buf"]        # the GEMM wrappers may use the padded width
    out._plnlp_parts = (emb_weight, feats, cache)      # a first GCNConv may take the parts instead (GCNInputConvFn)
    return out


def materialize_concat(x: torch.Tensor) -> torch.Tensor:
    """the concatenated input itself for a tensor concat_features(defer=True) returned"""
    if not getattr(x, "_plnlp_stale", False):
        return x
    emb_weight, feats, cache = x._plnlp_parts
    return concat_features(emb_weight, feats, cache)


def _padded_operand(x: torch.Tensor):
    """(matrix with a 4-float-aligned width, true width): x itself when already aligned, the padded
    parent buffer when x is a column slice made by concat_features, else a zero-padded copy"""
    k = x.shape[1]
    if k % 4 == 0 and x.stride(0) % 4 == 0:
        return x, k
    full = getattr(x, "_plnlp_padded", None)
    if full is not None and full.shape[0] == x.shape[0] and full.data_ptr() == x.data_ptr():
        return full, k
    xp = torch.zeros(x.shape[0], _pad4(k), dtype=torch.float32, device=x.device)
    xp[:, :k].copy_(x)
    return xp, k


GCN_INPUT_FUSION = {"enabled": True}


class GCNInputConvFn(torch.autograd.Function):
    """The FIRST GCNConv of an encoder whose input is create_input_feat's [emb.weight | data.x]
    (model.py:98-105) with constant node features:

        y = act( A_hat [emb | x] W^T + b )  evaluated as  act( [A_hat emb | A_hat x] W^T + b ).

    GCNConv transforms first and aggregates the `out`-wide product; aggregating first costs the input
    width instead -- and here only the embedding block changes between steps: `A_hat x` is computed once
    per (graph, features) and kept, so a step aggregates e = 50 columns instead of 200, forward and
    backward (only the embedding needs a gradient: A_hat^T (dz W)[:, :e]), and the data-gradient GEMM
    shrinks to those columns.  Same value as the reference order up to fp32 reassociation.
    Layout of the aggregated operand: [A emb (e, padded to 4) | A x (f, padded to 4) | zeros up to a multiple
    of 16 columns] -- the reduction then has no ragged K-tile (citation2: 52 + 128 = 180 -> 192 columns:
    0.48 -> 0.56 of the MFMA peak on the forward product, profiles/r02_gemm_microbench_v8.jsonl)."""

    @staticmethod
    def forward(ctx, emb_weight, w, b, graph: Graph, act: _Act, feats, cache: dict, sink: Optional["GradSink"] = None):
        """sink: a GradSink carrying the table's Adam state (sink.adam, moments in the table's padded layout) -- the backward may
        then step the table in the epilogue of the aggregation that finishes its gradient (PLNLP_EPI_ADAM), as SAGEConvFn does
        on the raw table; it sets sink.adam_applied when it did, else the gradient goes the usual way"""
        ctx.sink = sink
        n, e, f = emb_weight.shape[0], emb_weight.shape[1], feats.shape[1]
        ep, fp = _pad_emb(e), _pad4(f)
        key = ("gcn_input", feats.data_ptr(), feats._version, n, e, f)
        st = cache.get("gcn_input")
        # the entry holds the graph and the feature tensor themselves (compared by identity): an id() or a
        # data pointer alone can be reused by a different object once the old one is freed
        kp = (ep + fp + 15) // 16 * 16
        if st is None or st["key"] != key or st["graph"] is not graph or st["feats"] is not feats:
            ax = torch.zeros(n, kp, dtype=torch.float32, device=emb_weight.device)
            fpad = torch.zeros(n, fp, dtype=torch.float32, device=emb_weight.device)
            fpad[:, :f].copy_(feats)
            csr_aggregate(graph, fpad, "sum", use_values=True, out=ax[:, ep:ep + fp])  # A_hat x, once
            st = {"key": key, "graph": graph, "feats": feats, "ax": ax,
                  "emb_pad": None,      # (allocated below, only for a table that is NOT kept padded: 0.75 GB on citation2)
                  # W in the aggregated operand's layout; the pad columns stay zero, the two blocks are
                  # refreshed per step (two copies instead of a fresh zero-filled matrix)
                  "wa": torch.zeros(w.shape[0], kp, dtype=torch.float32, device=w.device),
                  # W's columns in that layout: one indexed copy per step each way instead of two copies / a concatenation
                  "wcols": torch.cat([torch.arange(e, device=w.device), ep + torch.arange(f, device=w.device)])}
            cache["gcn_input"] = st
        ax = st["ax"]
        emb_pad = padded_base(emb_weight.detach())       # the table itself when it is kept padded (BaseModel): no copy
        if emb_pad is None:
            if st["emb_pad"] is None:
                st["emb_pad"] = torch.zeros(n, ep, dtype=torch.float32, device=ax.device)
            emb_pad = st["emb_pad"]
            emb_pad[:, :e].copy_(emb_weight.detach())
        csr_aggregate(graph, emb_pad, "sum", use_values=True, out=ax[:, :ep])          # A_hat emb, every step
        wa = st["wa"]
        wa.index_copy_(1, st["wcols"], w.detach())
        epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
        y = gemm([(ax, wa)], False, True, epilogue=epi)
        ctx.graph, ctx.act, ctx.dims, ctx.wcols = graph, act, (e, f, ep, fp), st["wcols"]
        ctx.ax = ax            # persistent buffer: this step's backward runs before the next forward rewrites it
        # a padded table takes its gradient in the padded layout too, set on the parameter directly (autograd would copy a
        # strided gradient into a contiguous one: the 0.18 ms this avoids on citation2)
        ctx.direct_grad_to = emb_weight if (_DIRECT_TABLE_GRAD["depth"] > 0 and padded_base(emb_weight.detach()) is not None
                                            and emb_weight.is_leaf) else None
        ctx.save_for_backward(wa, y if act.active else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        wa, y = ctx.saved_tensors
        graph, act, ax = ctx.graph, ctx.act, ctx.ax
        e, f, ep, fp = ctx.dims
        dz = _act_backward(gy.contiguous(), y, act)
        need = ctx.needs_input_grad
        gemb = gw = gb = None
        cs = [] if need[2] else None        # (the bias gradient out of the weight-gradient launch where that is the whole-block kernel)
        if need[1]:
            gwa = gemm([(dz, ax)], True, False, a_colsum=cs)                          # [out, ep + fp]
            gw = gwa.index_select(1, ctx.wcols)
        bias_sum = SideColsum(dz) if (need[2] and not cs) else None       # else: beside the products and the transposed aggregation below
        if need[0]:
            g_aemb = gemm([(dz, wa[:, :ep])], False, False)         # [N, ep]: embedding columns only (a view: ldb = the padded width)
            sink = ctx.sink
            table = padded_base(ctx.direct_grad_to.detach()) if ctx.direct_grad_to is not None else None
            ad = sink.adam if (sink is not None and table is not None) else None
            if (ad is not None and ad["param"] is ctx.direct_grad_to and ad["exp_avg"].shape == table.shape
                    and ctx.direct_grad_to.grad is None and _vector_path(g_aemb, table, ep)):
                # nothing but Adam consumes the table's gradient (the reference clips encoder and predictor, not the embedding;
                # one process: nothing to reduce): the aggregation that finishes the gradient steps the table -- no [N, ep]
                # gradient written and read back (0.75 GB each way on citation2).  The weight gradient above read `ax`, the
                # aggregated table, not the table: nothing queued after this launch reads the old values.  Pad columns: zero
                # gradient, zero moments, a zero update.
                sc = _step_scalars["active"]
                epi = L.make_epilogue(adam=(ad["exp_avg"], ad["exp_avg_sq"], ad["step"], ad["lr"], ad["betas"][0], ad["betas"][1],
                                            ad["eps"]),
                                      adam_scalars_ptr=sc.adam_ptr() if sc is not None else 0)
                csr_aggregate(graph.t(), g_aemb, "sum", use_values=True, out=table, epilogue=epi)
                sink.adam_applied = True
                if need[2]:
                    gb = cs[0] if cs else bias_sum.join()
                return None, gw, gb, None, None, None, None, None
            full = csr_aggregate(graph.t(), g_aemb, "sum", use_values=True)           # (its pad columns are exact zeros)
            if ctx.direct_grad_to is not None and ctx.direct_grad_to.grad is None:
                ctx.direct_grad_to.grad = full[:, :e]                                   # a view: no copy
            else:
                gemb = full[:, :e].contiguous()
        if need[2]:
            gb = cs[0] if cs else bias_sum.join()
        return gemb, gw, gb, None, None, None, None, None


class GCNConvFn(torch.autograd.Function):
    """One GCNConv(normalize=False) (+ relu/dropout):  y = act( A_hat (x W^T) + b )
    forward : MFMA GEMM, then K1 weighted aggregate with fused bias/relu/dropout."""

    @staticmethod
    def forward(ctx, x, w, b, graph: Graph, act: _Act, in_act: Optional[_Act] = None,
                channel: Optional[SparseGradChannel] = None, out_rows=None):
        """out_rows: produce only those rows of y, compact (OutputRows above); needs `channel`"""
        ctx.channel = channel
        ctx.out_rows = out_rows
        if channel is not None:
            ctx.set_materialize_grads(False)
        kin = x.shape[1]
        if kin % 4 != 0:
            # unaligned input width (citation2: 50 + 128 = 178): run the GEMMs on 4-float-padded
            # operands (zero pad columns) so they stay on the 16-byte loaders
            xp, _ = _padded_operand(x if x.dtype == torch.float32 else x.float())
            wp = torch.zeros(w.shape[0], xp.shape[1], dtype=torch.float32, device=w.device)
            wp[:, :kin].copy_(w)
        else:
            xp, wp = _f32c(x), w
        xw = gemm([(xp, wp)], False, True)
        if out_rows is not None:
            assert channel is not None, "a row-restricted forward hands its gradient back through the channel"
            epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr,
                                  dropout_rows=out_rows.rows)
            y = csr_aggregate(graph, xw, "sum", use_values=True, epilogue=epi, row_index=out_rows.rows,
                              out_map=out_rows.node_map)
        else:
            epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
            y = csr_aggregate(graph, xw, "sum", use_values=True, epilogue=epi)
        ctx.graph, ctx.act, ctx.kin = graph, act, kin
        ctx.in_act = in_act if (in_act is not None and in_act.active) else None
        if ctx.in_act is not None:
            ctx.in_act.gate_in_consumer = True
        ctx.save_for_backward(xp, wp, y if act.active else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        graph, act = ctx.graph, ctx.act
        sg = ctx.channel.take() if ctx.channel is not None else None
        if ctx.out_rows is not None:
            sg = _compact_grad(ctx.out_rows, sg, gy, y, act, x.shape[0])
            gy = None
        elif sg is not None and (gy is not None or (act.active and not act.gate_in_consumer)
                                 or sg.values.shape[0] == 0):
            gy = sg.to_dense() if gy is None else gy + sg.to_dense()
            sg = None
        if sg is None and gy is None:
            return (None,) * 8
        # row-sparse dz (zero outside sg.rows): the bias gradient sums the touched rows and the
        # transposed aggregation gathers mapped source rows only; its result is dense again
        dz = sg.values if sg is not None else _act_backward(gy.contiguous(), y, act)
        need = ctx.needs_input_grad
        gx = gw = gb = None
        bias_sum = SideColsum(dz) if need[2] else None          # beside the transposed aggregation and the two products below
        if need[0] or need[1]:
            gxw = csr_aggregate(graph.t(), dz, "sum", use_values=True,
                                src_map=sg.node_map if sg is not None else None)
            kin = ctx.kin
            if need[1]:
                gw = gemm([(gxw, x)], True, False)
                if gw.shape[1] != kin:
                    gw = gw[:, :kin].contiguous()
            if need[0]:
                ia = ctx.in_act
                gx = gemm([(gxw, w)], False, False,
                          epilogue=L.make_epilogue(gate=x, gate_scale=ia.scale) if ia is not None else None)
                if gx.shape[1] != kin:
                    gx = gx[:, :kin]
        if bias_sum is not None:
            gb = bias_sum.join()
        return gx, gw, gb, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA GEMM (MLPPredictor.lins, layer.py:83-86)."""

    @staticmethod
    def forward(ctx, x, w, b, act: _Act):
        x = _f32c(x)
        epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
        y = gemm([(x, w)], False, True, epilogue=epi)
        ctx.act = act
        ctx.save_for_backward(x, w, y if act.active else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        dz = _act_backward(gy.contiguous(), y, ctx.act)
        need = ctx.needs_input_grad
        gx = gw = gb = None
        if need[0]:
            gx = gemm([(dz, w)], False, False)
        if need[1]:
            gw = gemm([(dz, x)], True, False)
        if need[2]:
            gb = colsum(dz)
        return gx, gw, gb, None


class MLPStackFn(torch.autograd.Function):
    """MLPPredictor.lins as ONE autograd node (layer.py:82-86):
        x_{i+1} = dropout(relu(x_i W_i^T + b_i)), last layer bare.
    Every hidden layer is an MFMA GEMM with bias/relu/dropout in its epilogue; a single-output
    last layer is a row reduction (matvec), not a GEMM.  Backward walks the stack once, and the
    relu/dropout derivative of layer i rides in the epilogue of the kernel that produces the
    gradient of its output (dgrad GEMM or the outer product of the head) -- no separate passes."""

    @staticmethod
    def forward(ctx, x, dropout_p: float, training: bool, *params):
        x = _f32c(x)
        n_layers = len(params) // 2
        xs, acts, head = [x], [], None
        for i in range(n_layers):
            w, b = params[2 * i], params[2 * i + 1]
            last = i == n_layers - 1
            act = _Act(not last, 0.0 if last else dropout_p, training)
            acts.append(act)
            if last and w.shape[0] == 1:
                y = head if head is not None else matvec(xs[-1], w, b).reshape(-1, 1)
            else:
                epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
                if i == n_layers - 2 and params[2 * i + 2].shape[0] == 1:
                    # the hidden layer under a 1-output head: the head rides in this product's epilogue where the launch
                    # takes the stationary-weights form (else head stays None and the last layer is its own pass)
                    y, head = gemm([(xs[-1], w)], False, True, epilogue=epi, rowdot=(params[2 * i + 2], params[2 * i + 3]))
                else:
                    y = gemm([(xs[-1], w)], False, True, epilogue=epi)
            xs.append(y)
        ctx.acts, ctx.n_layers = acts, n_layers
        ctx.save_for_backward(*xs[:-1], *params)
        return xs[-1]

    @staticmethod
    def backward(ctx, g):
        nl = ctx.n_layers
        saved = ctx.saved_tensors
        xs, params = saved[:nl], saved[nl:]
        need = ctx.needs_input_grad
        grads = [None] * (2 * nl)
        d = _f32c(g)
        bias_grad_ready = None      # the bias gradient of layer i, already produced by the pass that made d
        for i in range(nl - 1, -1, -1):
            w, b = params[2 * i], params[2 * i + 1]
            xi = xs[i]
            head = (i == nl - 1) and w.shape[0] == 1
            if head and i > 0 and ctx.acts[i - 1].active:
                # the head behind a relu / dropout layer: its weight and bias gradients, the gradient of the hidden
                # pre-activation and THAT layer's bias gradient from one pass over the hidden activation
                fused = mlp_head_backward(xi, d, w, ctx.acts[i - 1].scale)
                if fused is not None:
                    d, gw, bias_grad_ready, gb = fused
                    if need[3 + 2 * i]:
                        grads[2 * i] = gw
                    if b is not None and need[4 + 2 * i]:
                        grads[2 * i + 1] = gb
                    continue
            if bias_grad_ready is not None:
                if b is not None and need[4 + 2 * i]:
                    grads[2 * i + 1] = bias_grad_ready
                b_done, bias_grad_ready = True, None
            else:
                b_done = False
            if need[3 + 2 * i]:
                grads[2 * i] = (colsum(xi, row_weight=d).reshape(1, -1) if head
                                else gemm([(d, xi)], True, False))
            if b is not None and need[4 + 2 * i] and not b_done:
                grads[2 * i + 1] = colsum(d.reshape(-1, 1)) if head else colsum(d)
            if i == 0 and not need[0]:
                d = None
                break
            # gradient of this layer's input; if that input is a relu/dropout output (i > 0), fold
            # its derivative in: the result is then d/d(pre-activation of layer i-1)
            epi = L.make_epilogue(gate=xi, gate_scale=ctx.acts[i - 1].scale) if i > 0 else None
            d = outer(d, w, epilogue=epi) if head else gemm([(d, w)], False, False, epilogue=epi)
        return (d, None, None, *grads)


# backward of the edge gathers: "segment" (deterministic gather-reduce) or "atomic"
EDGE_BACKWARD = {"mode": "segment"}


def prepare_edge_backward(src: torch.Tensor, dst: torch.Tensor, n_nodes: int, compact: bool,
                          count_host: Optional[torch.Tensor] = None):
    """Everything the backward of the edge gathers needs that depends on the batch's edges alone:
    the node-sorted incidence lists, their hot-node tables and -- compact=True -- the touched-node
    compaction with its count read-back in flight.  A trainer that knows the edges before the encoder
    runs calls this FIRST and passes the result to the scorer (`incidence=`): the sort and the
    read-back then overlap the forward pass instead of stalling the backward pass."""
    inc = Incidence(src, dst, n_nodes)
    if compact:
        inc = inc.compact(count_host)
    inc.row_split(split_threshold(n_nodes))
    return inc


PROLOGUE_OVERLAP = {"enabled": True}
# (Rounds 4-5 built and measured a sort-free builder of the batch's index structures -- counts by atomics, a scan fused with the
# compaction, scatter, per-segment ordering: the same tensors bit for bit in 10 launches fewer -- and removed it: on three same-box
# A/B runs the collab step was 1.6-2.1 % SLOWER with it, at every side-stream priority (profiles/r04_same_box_ab.txt,
# profiles/r05_same_box_ab.txt): its denser side-stream kernels disturb the step's own launches more than the library sort's many
# short ones do.)


JOIN_STATS = {"waited": 0, "skipped": 0}        # EdgeBatch.join: stream waits enqueued / found unnecessary


class StepThrottle:
    """Keeps the host at most `depth` training steps ahead of the GPU, and carries the lifetime of what a step
    borrowed from another stream.

    The host enqueues a collab step in ~1.0 ms, the GPU runs it in ~1.7 ms: unthrottled, the distance grows by
    0.7 ms per step and every step in flight pins memory (measured 1.7 hipMalloc calls per step in a 30-step run).
    Two steps of look-ahead hide every launch latency; beyond that the host only hoards memory.

    `keep` (tick) is released only once the step's event has been waited for.  That replaces
    Tensor.record_stream for the side-stream-built edge structures: the caching allocator implements
    record_stream by recording one event on the consuming stream per block when the block is freed -- ~15 marker
    packets at the head of the next step's forward pass (profiles/r02_step_gap.txt)."""

    waited_s = 0.0        # host time spent waiting here, process-wide (bench.py: host work = enqueue time - this)

    def __init__(self, depth: int = 2):
        import collections
        self.depth, self.events = depth, collections.deque()

    def tick(self, keep=None) -> None:
        ev = torch.cuda.Event(blocking=True)      # the waiting host thread sleeps instead of spinning (CPU quota, §4)
        ev.record()
        self.events.append((ev, keep))
        if len(self.events) > self.depth:
            old, _ = self.events.popleft()
            if not old.query():
                import time
                t0 = time.perf_counter()
                old.synchronize()
                StepThrottle.waited_s += time.perf_counter() - t0


STEP_THROTTLE = {"depth": 2}


class EdgeBatch:
    """The scored edges of one training step -- src / dst (positives then negatives) -- and the index
    structures of their backward pass (prepare_edge_backward).

    All of it depends on the batch's edge lists alone, and it is ~25 short dependent launches (a radix
    sort, a scan, a few fills: ~130 us of mostly idle GPU).  With overlap=True it is produced on the
    device's side stream: the host runs a step ahead of the GPU, so these launches execute in the shadow
    of the PREVIOUS step's bandwidth-bound kernels (Adam, aggregation) instead of at the head of this
    step.  `join()` makes the current stream wait for them; it must precede the first use of
    src / dst / incidence.  inputs_ready=False (the safe default) first makes the side stream wait for
    everything already queued on the current stream -- correct for any caller, but then nothing
    overlaps; a caller whose edge tensors are not produced by still-pending work on the current
    stream passes True."""

    def __init__(self, src_parts, dst_parts, n_nodes: int, build: bool, compact: bool, overlap: bool,
                 inputs_ready: bool = False, compact_endpoints: bool = False, record_streams: bool = True,
                 count_host: Optional[torch.Tensor] = None):
        """compact_endpoints: also src_c / dst_c = the endpoints as rows of a matrix that holds only the
        touched nodes, and the incidence's compact column list (a row-restricted encoder output).
        count_host: see CompactIncidence (the batch is being built inside a hipGraph capture)"""
        self._compact_endpoints = compact_endpoints and compact
        self._count_host = count_host
        dev = src_parts[0].device
        overlap = overlap and PROLOGUE_OVERLAP["enabled"] and dev.type == "cuda"
        self._done = None
        if overlap:
            main = torch.cuda.current_stream(dev)
            side = side_stream(dev)
            if not inputs_ready:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                self._produce(src_parts, dst_parts, n_nodes, build, compact)
                self._done = torch.cuda.Event()
                self._done.record(side)
            # allocated under the side stream, consumed on the main one: either the allocator is told
            # (record_stream), or the caller keeps this object alive until the consuming step has finished on
            # the device (StepThrottle.tick(keep=...)) -- see StepThrottle for why the latter is cheaper
            if record_streams:
                for t in self._tensors():
                    t.record_stream(main)
        else:
            self._produce(src_parts, dst_parts, n_nodes, build, compact)

    @staticmethod
    def _column_pair(s, d) -> bool:
        """s, d = columns 0 and 1 of one row-major [n, 2] int64 device tensor (`edge[:, 0]`, `edge[:, 1]`)"""
        return (s.is_cuda and s.dtype == torch.int64 and d.dtype == torch.int64 and s.dim() == 1 and d.dim() == 1
                and s.numel() == d.numel() and (s.numel() <= 1 or (s.stride(0) == 2 and d.stride(0) == 2))
                and d.data_ptr() == s.data_ptr() + 8)

    def _produce(self, src_parts, dst_parts, n_nodes, build, compact):
        if (len(src_parts) == 2 and len(dst_parts) == 2 and self._column_pair(src_parts[0], dst_parts[0])
                and self._column_pair(src_parts[1], dst_parts[1])):
            # [pos ; neg] -> src, dst in one launch (plnlp_edge_endpoints) instead of two torch.cat
            n_pos, n_neg = src_parts[0].numel(), src_parts[1].numel()
            both = torch.empty(2, n_pos + n_neg, dtype=torch.int64, device=src_parts[0].device)
            self.src, self.dst = both[0], both[1]
            L.check(L.load().plnlp_edge_endpoints(src_parts[0].data_ptr(), n_pos, src_parts[1].data_ptr(), n_neg,
                                                  self.src.data_ptr(), self.dst.data_ptr(), L.stream_ptr()),
                    "plnlp_edge_endpoints")
        else:
            self.src = torch.cat(src_parts) if len(src_parts) > 1 else src_parts[0].contiguous()
            self.dst = torch.cat(dst_parts) if len(dst_parts) > 1 else dst_parts[0].contiguous()
        self.incidence = None
        self.src_c = self.dst_c = None
        if build and self.src.numel() > 0:
            self.incidence = prepare_edge_backward(self.src, self.dst, n_nodes, compact, self._count_host)
            if self._compact_endpoints:
                inc = self.incidence
                if self.src.is_cuda:
                    # the endpoints and the lists' other endpoints as compact row ids: one launch
                    # (plnlp_compact_endpoints) instead of three gathers and three casts
                    e = self.src.numel()
                    both_c = torch.empty(2, e, dtype=torch.int64, device=self.src.device)
                    inc._other_c = torch.empty_like(inc.item_other)
                    self.src_c, self.dst_c = both_c[0], both_c[1]
                    L.check(L.load().plnlp_compact_endpoints(inc.node_map.data_ptr(), self.src.data_ptr(),
                                                             self.dst.data_ptr(), e, inc.item_other.data_ptr(),
                                                             inc.item_other.numel(), self.src_c.data_ptr(),
                                                             self.dst_c.data_ptr(), inc._other_c.data_ptr(),
                                                             L.stream_ptr()), "plnlp_compact_endpoints")
                else:
                    inc.prepare_compact_columns()
                    self.src_c = inc.node_map.index_select(0, self.src).long()
                    self.dst_c = inc.node_map.index_select(0, self.dst).long()

    def _tensors(self):
        out = [self.src, self.dst]
        inc = self.incidence
        if self.src_c is not None:
            out += [self.src_c, self.dst_c, inc._other_c]
        if inc is not None:
            for name in ("item_edge", "item_other", "seg_ptr", "_rows_cap", "node_map", "_rowptr_cap", "_count_dev", "_ws"):
                t = getattr(inc, name, None)
                if isinstance(t, torch.Tensor):
                    out.append(t)
            base = getattr(inc, "_base", None)
            if base is not None:
                out += [base.item_edge, base.item_other, base.seg_ptr]
            for holder in (inc, base):
                sp = getattr(holder, "_split", None) if holder is not None else None
                if sp is not None:
                    out.append(sp._buf)
        return out

    def join(self) -> "EdgeBatch":
        if self._done is not None:
            # prepared a step ahead (BaseModel.prepare_edges) the work has usually finished: then there is nothing
            # for the stream to wait on -- a cross-queue wait packet at the head of the forward pass measured
            # ~60 us of idle GPU per step on MI355X even when its event had long fired
            if not self._done.query():
                torch.cuda.current_stream().wait_event(self._done)
                JOIN_STATS["waited"] += 1
            else:
                JOIN_STATS["skipped"] += 1
            self._done = None
        return self


def _sparse_edge_backward(h, src, dst, g, gate_scale: float, ci=None, compact: bool = False) -> RowSparseGrad:
    """gradient of the gathered matrix h over the touched nodes only (scalar g: DOT, matrix g:
    Hadamard), with the producing layer's relu/dropout derivative folded in when gate_scale > 0.
    compact: h already holds only the touched rows (row i = node ci.rows[i]) and src / dst are compact."""
    if compact:
        assert ci is not None
        cv = ci.compact_view() if isinstance(ci, CompactIncidence) else ci
        epi = L.make_epilogue(gate=h, gate_scale=gate_scale) if gate_scale > 0.0 else None
        vals = edge_segment_bwd(h, cv, g, epilogue=epi)
        return RowSparseGrad(cv.rows, cv.node_map, vals, cv.node_map.numel(), cv.count)
    if ci is None:
        ci = Incidence(src, dst, h.shape[0]).compact()
    epi = L.make_epilogue(gate=h, gate_scale=gate_scale, gate_index=ci.rows) if gate_scale > 0.0 else None
    vals = edge_segment_bwd(h, ci, g, epilogue=epi)
    return RowSparseGrad(ci.rows, ci.node_map, vals, h.shape[0], ci.count)


class EdgeDotFn(torch.autograd.Function):
    """DotPredictor over gathered endpoints: out[e] = <h[src[e]], h[dst[e]]>
    (model.py:155-156 + layer.py:174-176 in one pass)."""

    @staticmethod
    def forward(ctx, h, src, dst, gate_scale=0.0, channel: Optional[SparseGradChannel] = None,
                compute_forward: bool = True, incidence=None, compact: bool = False):
        """gate_scale > 0: h is the output of relu(+dropout, scale = 1/(1-p)) and the returned
        gradient is taken w.r.t. the pre-activation (h > 0 ? g * gate_scale : 0), folding the
        activation backward into the gather-reduce epilogue (no separate pass over [N, F]).
        channel: deliver the gradient of h row-sparse through it (see SparseGradChannel).
        compute_forward=False: the scores are not needed (the caller already has them and only
        wants this node in the graph for its backward); the returned tensor is uninitialised.
        incidence: prepare_edge_backward(src, dst, ...) of the same edges, built ahead of time.
        compact: h holds only the touched rows, src / dst are rows of it (OutputRows above)."""
        h = _f32c(h)
        ctx.save_for_backward(h, src, dst)
        ctx.gate_scale = float(gate_scale)
        ctx.channel = channel
        ctx.incidence = incidence
        ctx.compact = bool(compact)
        if not compute_forward:
            L.require_device(h, src, dst)
            return torch.empty(src.numel(), dtype=torch.float32, device=h.device)
        return edge_dot_fwd(h, src, dst)

    @staticmethod
    def backward(ctx, g):
        h, src, dst = ctx.saved_tensors
        g = g.contiguous()
        gs = ctx.gate_scale
        inc = ctx.incidence
        if ctx.channel is not None and EDGE_BACKWARD["mode"] == "segment":
            ci = inc if isinstance(inc, (CompactIncidence, _CompactCols)) else None
            ctx.channel.grad = _sparse_edge_backward(h, src, dst, g, gs, ci, compact=ctx.compact)
            return None, None, None, None, None, None, None, None
        assert not ctx.compact, "a compact encoder output needs the row-sparse channel"
        if EDGE_BACKWARD["mode"] == "segment":
            epi = L.make_epilogue(gate=h, gate_scale=gs) if gs > 0.0 else None
            if not isinstance(inc, Incidence):
                inc = Incidence(src, dst, h.shape[0])
            gh = edge_segment_bwd(h, inc, g, epilogue=epi)
        else:
            gh = edge_scatter_bwd(h, src, dst, g)
            if gs > 0.0:
                gh = gate(gh, h, gs)
        return gh, None, None, None, None, None, None, None


class EdgeHadamardFn(torch.autograd.Function):
    """x[e,:] = h[src[e],:] * h[dst[e],:]  (model.py:155-156 + layer.py:81)."""

    @staticmethod
    def forward(ctx, h, src, dst, gate_scale=0.0, channel: Optional[SparseGradChannel] = None, incidence=None,
                compact: bool = False):
        """gate_scale / channel / incidence / compact: as in EdgeDotFn"""
        h = _f32c(h)
        ctx.save_for_backward(h, src, dst)
        ctx.gate_scale = float(gate_scale)
        ctx.channel = channel
        ctx.incidence = incidence
        ctx.compact = bool(compact)
        return edge_hadamard_fwd(h, src, dst)

    @staticmethod
    def backward(ctx, g):
        h, src, dst = ctx.saved_tensors
        g = _f32c(g)
        gs = ctx.gate_scale
        inc = ctx.incidence
        if ctx.channel is not None and EDGE_BACKWARD["mode"] == "segment":
            ci = inc if isinstance(inc, (CompactIncidence, _CompactCols)) else None
            ctx.channel.grad = _sparse_edge_backward(h, src, dst, g, gs, ci, compact=ctx.compact)
            return None, None, None, None, None, None, None
        assert not ctx.compact, "a compact encoder output needs the row-sparse channel"
        if EDGE_BACKWARD["mode"] == "segment":
            epi = L.make_epilogue(gate=h, gate_scale=gs) if gs > 0.0 else None
            if not isinstance(inc, Incidence):
                inc = Incidence(src, dst, h.shape[0])
            gh = edge_segment_bwd(h, inc, g, epilogue=epi)
        else:
            gh = edge_scatter_bwd(h, src, dst, g)
            if gs > 0.0:
                gh = gate(gh, h, gs)
        return gh, None, None, None, None, None, None


# MLPPredictor.score_edges as ONE autograd node with the Hadamard formed inside the first linear's loaders
# (needs the split-bf16 GEMM form; off: EdgeHadamardFn + MLPStackFn, the product written out and read back).
# OFF by default -- measured on the ddi recipe (profiles/r02_fused_edge_mlp_ab.txt): the step is 4 % SLOWER with
# it.  The 183 us Hadamard kernel and the 268 MB it writes do disappear, but a K-tile of the GEMM touches only 64
# bytes of each gathered row, so the two loaders issue 32 scattered 64-byte gathers per instruction where the
# stand-alone kernel reads whole 2 KB rows: the first linear's forward goes 0.39 -> 0.90 ms, its weight
# gradient 0.78 -> 1.05 ms.
FUSE_EDGE_MLP = {"enabled": False}


def edge_mlp_fusable(h: torch.Tensor, params) -> bool:
    return (FUSE_EDGE_MLP["enabled"] and GEMM_MATH["mode"] == "bf16x3" and h.is_cuda and h.dtype == torch.float32
            and len(params) >= 4 and params[0].shape[0] > 1)


class EdgeMLPFn(torch.autograd.Function):
    """MLPPredictor on the edges (src, dst) of h (model.py:155-156 + layer.py:81-86):
        y = lins[-1]( ... dropout(relu(lins[0](h[src] * h[dst]))) ... )
    The [E, K] Hadamard is never materialised: the first linear's GEMM gathers the two endpoint rows and
    multiplies them in its A loader (a_index / a_index2), its weight gradient dz^T (h[src] * h[dst]) does the
    same in its B loader (b_index / b_index2).  Only the data gradient d(Hadamard) = dz W exists as a matrix --
    the input of the edge backward (gate_scale / channel / incidence / compact: as in EdgeHadamardFn)."""

    @staticmethod
    def forward(ctx, h, src, dst, gate_scale, channel, incidence, compact, dropout_p: float, training: bool, *params):
        h = _f32c(h)
        s32, d32 = src.to(torch.int32), dst.to(torch.int32)
        n_layers = len(params) // 2
        xs, acts = [], []
        x = None
        for i in range(n_layers):
            w, b = params[2 * i], params[2 * i + 1]
            last = i == n_layers - 1
            act = _Act(not last, 0.0 if last else dropout_p, training)
            acts.append(act)
            epi = L.make_epilogue(bias=b, relu=act.relu, dropout_p=act.p, dropout_seed=act.seed, dropout_seed_ptr=act.seed_ptr)
            if i == 0:
                y = gemm([(h, w)], False, True, epilogue=epi, a_index=[s32], a_index2=d32)
            elif last and w.shape[0] == 1:
                y = matvec(x, w, b).reshape(-1, 1)
            else:
                y = gemm([(x, w)], False, True, epilogue=epi)
            xs.append(y)
            x = y
        ctx.acts, ctx.n_layers = acts, n_layers
        ctx.gate_scale, ctx.channel, ctx.incidence, ctx.compact = float(gate_scale), channel, incidence, bool(compact)
        ctx.save_for_backward(h, src, dst, s32, d32, *xs[:-1], *params)
        return xs[-1]

    @staticmethod
    def backward(ctx, g):
        nl = ctx.n_layers
        saved = ctx.saved_tensors
        h, src, dst, s32, d32 = saved[:5]
        xs, params = saved[5:5 + nl - 1], saved[5 + nl - 1:]       # xs[i] = output of layer i = input of layer i+1
        need = ctx.needs_input_grad
        grads = [None] * (2 * nl)
        d = _f32c(g)
        for i in range(nl - 1, -1, -1):
            w, b = params[2 * i], params[2 * i + 1]
            head = (i == nl - 1) and w.shape[0] == 1
            if need[9 + 2 * i]:
                if i == 0:
                    grads[0] = gemm([(d, h)], True, False, b_index=s32, b_index2=d32)
                else:
                    grads[2 * i] = (colsum(xs[i - 1], row_weight=d).reshape(1, -1) if head
                                    else gemm([(d, xs[i - 1])], True, False))
            if b is not None and need[10 + 2 * i]:
                grads[2 * i + 1] = colsum(d.reshape(-1, 1)) if head else colsum(d)
            epi = L.make_epilogue(gate=xs[i - 1], gate_scale=ctx.acts[i - 1].scale) if i > 0 else None
            d = outer(d, w, epilogue=epi) if head else gemm([(d, w)], False, False, epilogue=epi)
        # d = gradient of the Hadamard; the gathers' backward as in EdgeHadamardFn
        gs, inc = ctx.gate_scale, ctx.incidence
        if ctx.channel is not None and EDGE_BACKWARD["mode"] == "segment":
            ci = inc if isinstance(inc, (CompactIncidence, _CompactCols)) else None
            ctx.channel.grad = _sparse_edge_backward(h, src, dst, d, gs, ci, compact=ctx.compact)
            gh = None
        else:
            assert not ctx.compact, "a compact encoder output needs the row-sparse channel"
            if EDGE_BACKWARD["mode"] == "segment":
                epi = L.make_epilogue(gate=h, gate_scale=gs) if gs > 0.0 else None
                if not isinstance(inc, Incidence):
                    inc = Incidence(src, dst, h.shape[0])
                gh = edge_segment_bwd(h, inc, d, epilogue=epi)
            else:
                gh = edge_scatter_bwd(h, src, dst, d)
                if gs > 0.0:
                    gh = gate(gh, h, gs)
        return (gh, None, None, None, None, None, None, None, None, *grads)


_unit_grads = {}


def unit_grad(device) -> torch.Tensor:
    """a cached scalar 1.0 to seed `loss.backward(unit_grad(dev))`: autograd otherwise fills a fresh
    one per step, and the fused losses recognise THIS tensor and skip the multiplication by it"""
    key = torch.device(device)
    if key not in _unit_grads:
        _unit_grads[key] = torch.ones((), dtype=torch.float32, device=key)
    return _unit_grads[key]


def _is_unit(g: torch.Tensor) -> bool:
    u = _unit_grads.get(g.device)
    return u is not None and g.data_ptr() == u.data_ptr()


class PairwiseLossJointFn(torch.autograd.Function):
    """the same losses on ONE score tensor [pos (n) | neg (n*k)], as the training step produces it:
    the gradient comes back as one tensor too (slicing the scores first makes autograd rebuild it
    with two zero fills, two copies and an add)"""

    @staticmethod
    def forward(ctx, out, n_pos: int, weight, kind: str, num_neg: int):
        flat = _f32c(out.reshape(-1))
        gout = torch.empty_like(flat)
        loss, _, _ = pairwise_loss(kind, flat[:n_pos], flat[n_pos:], num_neg, weight, grad_out=gout)
        ctx.save_for_backward(gout)
        ctx.shape = out.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (gout,) = ctx.saved_tensors
        if not _is_unit(g):
            gout = gout * g
        return gout.reshape(ctx.shape), None, None, None, None


class PairwiseLossFn(torch.autograd.Function):
    """loss.py:5-48 forward and backward in one kernel; backward just scales."""

    @staticmethod
    def forward(ctx, pos, neg, weight, kind: str, num_neg: int):
        loss, gpos, gneg = pairwise_loss(kind, pos, neg, num_neg, weight)
        ctx.save_for_backward(gpos, gneg)
        ctx.shapes = (pos.shape, neg.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        gpos, gneg = ctx.saved_tensors
        ps, ns = ctx.shapes
        return (gpos * g).reshape(ps), (gneg * g).reshape(ns), None, None, None
