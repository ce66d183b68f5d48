"""Encoders and edge predictors with the surface of the reference's
plnlp/layer.py (same class names, constructor arguments, `forward` arguments,
`reset_parameters`, attribute and state_dict names) over the HIP kernels.

What differs from the reference is only *where* work happens:
  * a conv and the relu/dropout that BaseGNN applies after it are one fused call
    (GEMM / aggregation epilogues) -- `BaseGNN.forward` hands the activation to
    the conv instead of running separate element-wise passes;
  * predictors additionally expose `score_edges(h, src, dst)`, which fuses the
    endpoint gathers of plnlp/model.py:155-156 into the scoring kernel; the
    reference-style `forward(x_i, x_j)` is kept for callers that gather first.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops
from .graph import Graph
from .ops import _Act


def _require_graph(adj_t) -> Graph:
    if not isinstance(adj_t, Graph):
        raise TypeError("adj_t must be a plnlp_amd.Graph (the MI355X stand-in for torch_sparse.SparseTensor); "
                        f"got {type(adj_t).__name__}")
    return adj_t


# ------------------------------------------------------------------ convs ------
class SAGEConv(torch.nn.Module):
    """PyG 2.0.1 SAGEConv(in, out) defaults as used at layer.py:36: mean
    aggregation (edge values ignored), root weight, bias on lin_l only.
    Parameters: lin_l.weight, lin_l.bias, lin_r.weight.
    aggr: 'mean' (the reference's, fused path) | 'max' | 'add' (PyG's other SAGEConv reductions, on the
    max / sum aggregation kernels followed by the two linears)."""

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean"):
        super().__init__()
        if aggr not in ("mean", "max", "add", "sum"):
            raise ValueError(f"aggr must be mean, max or add, not {aggr!r}")
        self.aggr = aggr
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = torch.nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = torch.nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def forward(self, x, adj_t, act: _Act = None, in_act: _Act = None, sink=None, channel=None, out_rows=None):
        act = act if act is not None else _Act(False, 0.0, False)
        if self.aggr != "mean":
            agg = ops.AggregateFn.apply(x, _require_graph(adj_t), "sum" if self.aggr == "add" else self.aggr, False)
            return ops.LinearFn.apply(torch.cat([agg, x], dim=1),
                                      torch.cat([self.lin_l.weight, self.lin_r.weight], dim=1), self.lin_l.bias, act)
        return ops.SAGEConvFn.apply(x, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight,
                                    _require_graph(adj_t), act, in_act, sink, channel, out_rows)


    def forward_block(self, x_full, adj_block, row_lo: int, act: _Act = None):
        """this conv on one destination-row block of a row-sharded encoder: x_full holds every source
        row, adj_block the rank's CSR slice; returns the block's output rows (ops.SAGEConvBlockFn)"""
        if self.aggr != "mean":
            raise NotImplementedError("row-sharded SAGEConv: mean aggregation only")
        act = act if act is not None else _Act(False, 0.0, False)
        return ops.SAGEConvBlockFn.apply(x_full, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight,
                                         _require_graph(adj_block), act, int(row_lo))


class GCNConv(torch.nn.Module):
    """PyG 2.0.1 GCNConv(in, out, normalize=False) as used at layer.py:45:
    glorot `lin` without bias, then aggregation with the stored (pre-normalised,
    main.py:177-179) edge values, then `bias` (zeros at reset)."""

    def __init__(self, in_channels: int, out_channels: int, normalize: bool = False):
        super().__init__()
        if normalize:
            raise NotImplementedError("the reference always passes normalize=False (layer.py:45)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.lin.weight)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x, adj_t, act: _Act = None, in_act: _Act = None, channel=None, out_rows=None, input_grad_sink=None):
        """input_grad_sink: an ops.GradSink with the embedding table's Adam state, honoured on the [emb | features] input path
        only (ops.GCNInputConvFn); otherwise it stays untouched and the table's gradient goes through autograd"""
        act = act if act is not None else _Act(False, 0.0, False)
        parts = getattr(x, "_plnlp_parts", None)
        if parts is not None and channel is None and in_act is None and ops.GCN_INPUT_FUSION["enabled"]:
            # x = [emb.weight | constant features]: aggregate first, the feature block once (ops.GCNInputConvFn)
            emb_weight, feats, cache = parts
            return ops.GCNInputConvFn.apply(emb_weight, self.lin.weight, self.bias, _require_graph(adj_t), act, feats,
                                            cache, input_grad_sink)
        x = ops.materialize_concat(x)
        return ops.GCNConvFn.apply(x, self.lin.weight, self.bias, _require_graph(adj_t), act, in_act, channel, out_rows)

    def forward_block(self, x_full, adj_block, row_lo: int, act: _Act = None):
        """this conv on one destination-row block of a row-sharded encoder (ops.GCNConvBlockFn)"""
        act = act if act is not None else _Act(False, 0.0, False)
        x_full = ops.materialize_concat(x_full)
        return ops.GCNConvBlockFn.apply(x_full, self.lin.weight, self.bias, _require_graph(adj_block), act)


# --------------------------------------------------------------- encoders ------
class BaseGNN(torch.nn.Module):
    """layer.py:7-27.  relu+dropout follow every conv but the last; a 1-layer
    encoder gets them after its only conv too (layer.py:24-26)."""

    def __init__(self, dropout, num_layers):
        super().__init__()
        self.convs = torch.nn.ModuleList()
        self.dropout = dropout
        self.num_layers = num_layers

    def reset_parameters(self):
        for conv in self.convs:
            conv.reset_parameters()

    def _forward_sharded(self, x, adj_block, shard, sparse=None):
        """row-sharded pass (plnlp_amd/shard.py): x holds ALL source rows, adj_block is this rank's CSR
        slice; every conv produces the rank's S output rows, and between layers the blocks are
        all-gathered into the next layer's source matrix (its backward reduce-scatters the gradient).
        sparse = (adj_square, block_rows, channel, sink): the LAST conv -- a mean SAGEConv -- is evaluated only at
        the rows of this rank's block that the global batch touches (shard.BlockRows; adj_square: the block's CSR
        kept at global row positions, Graph.row_block_square) and returns the compact [R, out] matrix; its gradient
        comes back row-sparse through `channel`, the gradient of x goes to `sink` when one is given.  Same kernels
        as the single-process row-sparse step (ops.SAGEConvFn with out_rows)."""
        last = len(self.convs) - 1
        for i, conv in enumerate(self.convs):
            if not isinstance(conv, (SAGEConv, GCNConv)):
                raise NotImplementedError("row-sharded encoder: SAGE / GCN layers")
            activated = i < last or self.num_layers == 1
            if i == last and sparse is not None:
                adj_square, block_rows, channel, sink = sparse
                # masks are indexed by GLOBAL row position here: one stream for every rank, the same mask the
                # unsharded step draws
                act = _Act(True, self.dropout, self.training) if activated else _Act(False, 0.0, False)
                return ops.SAGEConvFn.apply(x, conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight,
                                            _require_graph(adj_square), act, None, sink if i == 0 else None, channel,
                                            block_rows)
            act = _Act(True, self.dropout, self.training) if activated else None
            if act is not None and act.p > 0.0:
                # a row is computed by exactly one rank and the mask is indexed by the row's position in
                # the block: give every block its own stream so equal positions do not share a mask
                act.seed = (act.seed ^ (0x9E3779B97F4A7C15 * (shard.rank + 1))) & 0xFFFFFFFFFFFFFFFF
            y = conv.forward_block(x, adj_block, shard.lo, act)
            x = shard.all_gather(y) if i < last else y
        return x

    def forward(self, x, adj_t, fuse_output_gate: bool = False, input_grad_sink=None,
                output_grad_channel=None, shard=None, output_rows=None, shard_sparse=None):
        """shard: a plnlp_amd.shard.ShardContext -> row-sharded pass (see _forward_sharded).
        output_rows: an ops.CompactIncidence -- the last conv produces only those rows, as a compact
        matrix (ops.SPARSE_FORWARD / "OutputRows"); needs output_grad_channel and native convs.
        input_grad_sink: an ops.GradSink for the gradient of `x` (first conv: a SAGEConv on the raw table, or a GCNConv on
        [table | features] -- there only the fused Adam update of the table is taken from it).
        output_grad_channel: an ops.SparseGradChannel through which the (single) consumer of the
        returned h hands back its gradient row-sparse; only honoured by the native convs.
        fuse_output_gate (only meaningful for a 1-layer encoder, whose output IS a
        relu+dropout result): returns (h, gate_scale) and leaves the derivative of that
        final activation to the consumer's backward (EdgeDotFn), see ops._Act."""
        if shard is not None:
            return self._forward_sharded(x, adj_t, shard, shard_sparse)
        last = len(self.convs) - 1
        out_act = None
        prev_act = None        # activation that produced the current x (native convs only)
        for i, conv in enumerate(self.convs):
            activated = i < last or self.num_layers == 1
            if isinstance(conv, (SAGEConv, GCNConv)):
                act = _Act(True, self.dropout, self.training) if activated else None
                if fuse_output_gate and i == last and act is not None:
                    act.gate_in_consumer = True
                    out_act = act
                # the conv's backward folds the derivative of the activation that produced its input
                ch = output_grad_channel if (i == last and torch.is_grad_enabled()) else None
                if isinstance(conv, SAGEConv) and conv.aggr != "mean":
                    # un-fused reductions (max / add): no sink, no row-sparse channel, no folded input gate
                    if ch is not None or (i == 0 and input_grad_sink is not None):
                        raise ValueError("gradient sinks / channels need the fused (mean) SAGEConv")
                    x = conv(x, adj_t, act)
                elif i == 0 and input_grad_sink is not None and isinstance(conv, SAGEConv):
                    x = conv(x, adj_t, act, None, input_grad_sink, ch, out_rows=output_rows if i == last else None)
                else:
                    extra = ({"input_grad_sink": input_grad_sink}
                             if (i == 0 and input_grad_sink is not None and isinstance(conv, GCNConv)) else {})
                    x = conv(x, adj_t, act, prev_act if torch.is_grad_enabled() else None, channel=ch,
                             out_rows=output_rows if (i == last and ch is not None) else None, **extra)
                prev_act = act
            else:  # foreign conv module: un-fused reference order
                x = conv(x, adj_t)
                if activated:
                    x = F.dropout(F.relu(x), p=self.dropout, training=self.training)
                prev_act = None
        if fuse_output_gate:
            return x, (out_act.scale if out_act is not None else 0.0)
        return x


def _stack(conv_cls, in_channels, hidden_channels, out_channels, num_layers):
    widths = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
    return [conv_cls(widths[i], widths[i + 1]) for i in range(num_layers)]


class SAGE(BaseGNN):
    """layer.py:30-36"""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__(dropout, num_layers)
        self.convs.extend(_stack(SAGEConv, in_channels, hidden_channels, out_channels, num_layers))


class GCN(BaseGNN):
    """layer.py:39-45"""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__(dropout, num_layers)
        self.convs.extend(_stack(GCNConv, in_channels, hidden_channels, out_channels, num_layers))


class _OutOfScopeEncoder(BaseGNN):
    _what = ""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        raise NotImplementedError(
            f"{type(self).__name__} ({self._what}) is outside the hot path this package accelerates "
            "(SURVEY.md 2.1 #1: no README recipe uses it); use SAGE or GCN")


class WSAGE(_OutOfScopeEncoder):
    """layer.py:48-54 (GraphConv) -- constructible name only."""
    _what = "PyG GraphConv"


class Transformer(_OutOfScopeEncoder):
    """layer.py:57-63 (TransformerConv) -- constructible name only."""
    _what = "PyG TransformerConv"


# -------------------------------------------------------------- predictors ------
def _mlp(lins, x, dropout, training):
    """Linear -> relu -> dropout ... -> Linear, every Linear on the MFMA GEMM with
    the activation fused into its epilogue (layer.py:82-86)."""
    last = len(lins) - 1
    for i, lin in enumerate(lins):
        act = _Act(True, dropout, training) if i < last else _Act(False, 0.0, False)
        x = ops.LinearFn.apply(x, lin.weight, lin.bias, act)
    return x


def _hidden_stack(lins, x, dropout, training):
    """every Linear followed by relu+dropout (MLPDot/MLPBil towers, layer.py:130-136)."""
    for lin in lins:
        x = ops.LinearFn.apply(x, lin.weight, lin.bias, _Act(True, dropout, training))
    return x


def _linear_list(widths):
    return torch.nn.ModuleList(torch.nn.Linear(widths[i], widths[i + 1]) for i in range(len(widths) - 1))


class _LinsPredictor(torch.nn.Module):
    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()

    def score_edges(self, h, src, dst):
        """gather-first fallback for the predictors that have no fused kernel"""
        return self.forward(h[src], h[dst])


class MLPPredictor(_LinsPredictor):
    """layer.py:66-87: Hadamard product of the endpoint rows, then an MLP -> [E, out]."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__()
        self.lins = _linear_list([in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels])
        self.dropout = dropout

    def _stack(self, x):
        params = []
        for lin in self.lins:
            params += [lin.weight, lin.bias]
        return ops.MLPStackFn.apply(x, float(self.dropout), self.training, *params)

    def forward(self, x_i, x_j):
        return self._stack(x_i * x_j)

    def score_edges(self, h, src, dst, gate_scale: float = 0.0, channel=None, incidence=None, compact: bool = False):
        params = []
        for lin in self.lins:
            params += [lin.weight, lin.bias]
        if ops.edge_mlp_fusable(h, params):       # the Hadamard formed inside the first linear's loaders
            return ops.EdgeMLPFn.apply(h, src, dst, gate_scale, channel, incidence, compact, float(self.dropout),
                                       self.training, *params)
        return self._stack(ops.EdgeHadamardFn.apply(h, src, dst, gate_scale, channel, incidence, compact))


class MLPCatPredictor(_LinsPredictor):
    """layer.py:90-116: symmetrised MLP over [x_i|x_j] and [x_j|x_i]."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout):
        super().__init__()
        self.lins = _linear_list([2 * in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels])
        self.dropout = dropout

    def forward(self, x_i, x_j):
        a = _mlp(self.lins, torch.cat([x_i, x_j], dim=-1), self.dropout, self.training)
        b = _mlp(self.lins, torch.cat([x_j, x_i], dim=-1), self.dropout, self.training)
        return (a + b) / 2


class MLPDotPredictor(_LinsPredictor):
    """layer.py:119-139: shared tower on both endpoints, then dot."""

    def __init__(self, in_channels, hidden_channels, num_layers, dropout):
        super().__init__()
        self.lins = _linear_list([in_channels] + [hidden_channels] * num_layers)
        self.dropout = dropout

    def forward(self, x_i, x_j):
        x_i = _hidden_stack(self.lins, x_i, self.dropout, self.training)
        x_j = _hidden_stack(self.lins, x_j, self.dropout, self.training)
        return torch.sum(x_i * x_j, dim=-1)


class MLPBilPredictor(_LinsPredictor):
    """layer.py:142-164: shared tower, then bilinear form."""

    def __init__(self, in_channels, hidden_channels, num_layers, dropout):
        super().__init__()
        self.lins = _linear_list([in_channels] + [hidden_channels] * num_layers)
        self.bilin = torch.nn.Linear(hidden_channels, hidden_channels, bias=False)
        self.dropout = dropout

    def reset_parameters(self):
        super().reset_parameters()
        self.bilin.reset_parameters()

    def forward(self, x_i, x_j):
        x_i = _hidden_stack(self.lins, x_i, self.dropout, self.training)
        x_j = _hidden_stack(self.lins, x_j, self.dropout, self.training)
        bx = ops.LinearFn.apply(x_i, self.bilin.weight, None, _Act(False, 0.0, False))
        return torch.sum(bx * x_j, dim=-1)


class DotPredictor(torch.nn.Module):
    """layer.py:167-176: <x_i, x_j> -> [E]; no parameters."""

    def __init__(self):
        super().__init__()

    def reset_parameters(self):
        return

    def forward(self, x_i, x_j):
        return torch.sum(x_i * x_j, dim=-1)

    def score_edges(self, h, src, dst, gate_scale: float = 0.0, channel=None, compute_forward: bool = True,
                    incidence=None, compact: bool = False):
        return ops.EdgeDotFn.apply(h, src, dst, gate_scale, channel, compute_forward, incidence, compact)


class BilinearPredictor(torch.nn.Module):
    """layer.py:179-189"""

    def __init__(self, hidden_channels):
        super().__init__()
        self.bilin = torch.nn.Linear(hidden_channels, hidden_channels, bias=False)

    def reset_parameters(self):
        self.bilin.reset_parameters()

    def forward(self, x_i, x_j):
        bx = ops.LinearFn.apply(x_i, self.bilin.weight, None, _Act(False, 0.0, False))
        return torch.sum(bx * x_j, dim=-1)

    def score_edges(self, h, src, dst):
        return self.forward(h[src], h[dst])
