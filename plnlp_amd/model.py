"""Trainer and factories -- the caller side of the hot path, with the surface of
the reference's plnlp/model.py (BaseModel, create_input_layer,
create_gnn_layer, create_predictor_layer, adjust_lr) so a main.py-style driver
swaps this package in.

What this BaseModel does differently from the reference's, at equal results:
  * positives and negatives of a step are scored by ONE fused gather+score call
    (one backward scatter), not two predictor calls;
  * the batch permutation and the epoch's edge tensors live on the device for
    the whole epoch; the per-step `loss.item()` host sync (model.py:170) is
    replaced by a device accumulator read once per epoch;
  * clipping (per group, model.py:163-165) + Adam run as fused HIP kernels;
  * optional edge-batch data parallelism: one process per GPU, every rank holds a
    replica, takes its slice of each step's batch, and gradients are SUM-reduced
    over RCCL before clipping (the loss is a sum over pairs, so the reduced
    gradient equals the single-GPU gradient of the whole batch).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import loss as loss_mod
from . import ops
from .layer import (GCN, SAGE, WSAGE, BaseGNN, BilinearPredictor, DotPredictor, GCNConv, MLPBilPredictor,
                    MLPCatPredictor, MLPDotPredictor, MLPPredictor, SAGEConv, Transformer)
from .optim import FusedAdam, fused_adam_state, group_sqnorm
from .utils import StreamedPermutation, batch_permutation, evaluate_hits, evaluate_mrr, get_pos_neg_edges

import os

# single process, SAGE on the raw embedding table: the table's Adam step rides in the epilogue of the kernel that
# finishes its gradient (BaseModel._embedding_grad_sink); False keeps gradient and update apart (plnlp_amd/switches.py)
FUSE_EMBEDDING_ADAM = {"enabled": True}
# the epoch's batch permutation shuffled a few batches ahead of the GPU by a host thread (utils.StreamedPermutation);
# False: the whole torch.randperm before the first step, as the reference's DataLoader does
STREAM_PERMUTATION = {"enabled": True}
# the epoch's running sum of loss * examples (model.py:169) fed by the loss kernel's own tail (ops.LOSS_ACC) in a
# one-process run; False: three element-wise launches per step, as the reference's line does it
FUSE_LOSS_ACC = {"enabled": True}
# row-sharded data parallelism: the last SAGE layer evaluated / back-propagated only at the rows of the rank's block that
# the global batch touches (False: every row of the block)
SHARD_SPARSE = {"enabled": True}
# a trainable embedding table of an unaligned width under a first GCN layer kept padded to 16-byte rows (BaseModel.__init__);
# False: a plain contiguous [N, e] table, padded and un-padded by a copy each step
PAD_EMBEDDING_TABLE = {"enabled": True}


class BaseModel(object):
    """plnlp/model.py:9-226.  Constructor arguments as the reference; keyword-only
    extras: `modules=(encoder, predictor[, loss_fn])` to inject pre-built modules,
    `process_group` / `dp_scaling` / `dp_exchange` for data parallelism.

    dp_exchange -- what the ranks exchange per step:
      'grads'  : SUM all-reduce of every parameter gradient (the embedding table's 4*N*h bytes
                 dominate; xGMI is point to point, so at 2-4 GPUs this costs more than the step);
      'scores' : every rank scores its slice of the batch, the ranks all-gather the per-edge score
                 gradients (4*(1+k)*B bytes) and each then runs the SAME backward pass over the
                 global batch.  Needs a parameter-free scorer (DOT); the replicas stay identical
                 because every kernel is deterministic (check_replicas() verifies it);
      'auto'   : 'scores' when the predictor has no parameters, else 'grads';
      'shard'  : the full-graph encoder itself is divided: rank r owns a block of destination rows of the
                 CSR, of the embedding table and of its Adam state, computes only those rows of every
                 layer, and the ranks exchange activations (all-gather between layers, an all-to-all of
                 just the rows a rank's edge slice touches for the scorer) instead of replicating the
                 encoder -- see plnlp_amd/shard.py and train_step_sharded.  SAGE encoders on the
                 embedding table (the ddi / collab recipes)."""

    def __init__(self, lr, dropout, grad_clip_norm, gnn_num_layers, mlp_num_layers, emb_hidden_channels,
                 gnn_hidden_channels, mlp_hidden_channels, num_nodes, num_node_feats, gnn_encoder_name,
                 predictor_name, loss_func, optimizer_name, device, use_node_feats, train_node_emb,
                 pretrain_emb=None, *, modules=None, process_group=None, dp_scaling="weak", dp_exchange="auto"):
        self.loss_func_name = loss_func
        self.num_nodes = num_nodes
        self.num_node_feats = num_node_feats
        self.use_node_feats = use_node_feats
        self.train_node_emb = train_node_emb
        self.clip_norm = grad_clip_norm
        self.device = torch.device(device)
        if self.device.type == "cuda":
            # the kernels launch on the CURRENT device's current stream (plnlp_amd/_lib.py::stream_ptr):
            # a bare "cuda" means the current device, an indexed one becomes current
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            torch.cuda.set_device(self.device)
        self.process_group = process_group
        self.dp_scaling = dp_scaling
        if dp_exchange not in ("auto", "grads", "scores", "shard"):
            raise ValueError(f"dp_exchange must be auto, grads, scores or shard, not {dp_exchange!r}")
        self.dp_exchange = dp_exchange

        self.input_channels, self.emb = create_input_layer(
            num_nodes=num_nodes, num_node_feats=num_node_feats, hidden_channels=emb_hidden_channels,
            use_node_feats=use_node_feats, train_node_emb=train_node_emb, pretrain_emb=pretrain_emb)
        if self.emb is not None:
            self.emb = self.emb.to(self.device)

        self._loss_override = None
        if modules is not None:
            self.encoder, self.predictor = modules[0].to(self.device), modules[1].to(self.device)
            if len(modules) > 2:          # (pos_out, neg_out, num_neg, margin) -> 0-d loss
                self._loss_override = modules[2]
        else:
            self.encoder = create_gnn_layer(input_channels=self.input_channels,
                                            hidden_channels=gnn_hidden_channels, num_layers=gnn_num_layers,
                                            dropout=dropout, encoder_name=gnn_encoder_name).to(self.device)
            self.predictor = create_predictor_layer(hidden_channels=mlp_hidden_channels,
                                                    num_layers=mlp_num_layers, dropout=dropout,
                                                    predictor_name=predictor_name).to(self.device)

        # a trainable table of an unaligned width under a first GCN layer (citation2: 50 columns next to 128 features) is
        # KEPT padded to 16-byte rows: `emb.weight` becomes the [:, :e] view of a zero-padded [N, e + pad] buffer.  The
        # 52-wide aggregation then gathers from the table itself (no per-step padded copy: 0.35 ms on citation2), the
        # gradient arrives in the same layout and Adam steps it there (no strided -> contiguous copy: 0.18 ms); the pad
        # columns hold zeros and a zero gradient for ever.  state_dict / create_input_feat see the [N, e] parameter.
        if (PAD_EMBEDDING_TABLE["enabled"] and self.emb is not None and self.device.type == "cuda" and train_node_emb
                and use_node_feats
                and isinstance(self.encoder, BaseGNN) and len(self.encoder.convs) > 0
                and isinstance(self.encoder.convs[0], GCNConv) and self.emb.weight.shape[1] % 4 != 0
                and self.emb.weight.requires_grad and dp_exchange != "shard"):
            w = self.emb.weight
            buf = torch.zeros(w.shape[0], ops._pad_emb(w.shape[1]), dtype=w.dtype, device=w.device)
            buf[:, :w.shape[1]].copy_(w.detach())
            w.data = buf[:, :w.shape[1]]

        self.para_list = list(self.encoder.parameters()) + list(self.predictor.parameters())
        if self.emb is not None:
            self.para_list += list(self.emb.parameters())

        # what the optimiser updates: the parameter list itself, or -- row-sharded -- the small weights
        # plus only the OWNED rows of the embedding table
        opt_params = self.para_list
        self._shard = None
        if dp_exchange == "shard" and process_group is not None:
            opt_params = self._setup_shard()

        self._fused_step = self.device.type == "cuda" and optimizer_name != 'SGD'
        if self._fused_step:
            self.optimizer = FusedAdam(opt_params, lr=lr, decoupled=(optimizer_name == 'AdamW'),
                                       weight_decay=0.01 if optimizer_name == 'AdamW' else 0.0)
        elif optimizer_name == 'AdamW':
            self.optimizer = torch.optim.AdamW(opt_params, lr=lr)
        elif optimizer_name == 'SGD':
            self.optimizer = torch.optim.SGD(opt_params, lr=lr, momentum=0.9, weight_decay=1e-5,
                                             nesterov=True)
        else:
            self.optimizer = torch.optim.Adam(opt_params, lr=lr)

    def _setup_shard(self):
        """dp_exchange='shard': partition the node rows over the ranks and re-seat the embedding table in a
        buffer padded to world * S rows.  `emb.weight` stays the full [N, F] table (a view of the buffer:
        state_dict, eval and the reference-style surface see what they always saw) but is no longer
        trained directly: the optimiser owns `_emb_shard`, the view of this rank's S rows."""
        from . import shard
        if self.emb is None or not self.train_node_emb:
            raise NotImplementedError("dp_exchange='shard' needs a trainable embedding table in the encoder's input "
                                      "(alone: ddi / collab; next to node features: citation2); use 'grads' otherwise")
        rank, world = self._world()
        part = shard.RowPartition(self.num_nodes, world, rank)
        self._shard = shard.ShardContext(self.process_group, part)
        w = self.emb.weight
        full = torch.zeros(part.padded, w.shape[1], dtype=w.dtype, device=self.device)
        full[:self.num_nodes].copy_(w.detach())
        w.data = full[:self.num_nodes]
        w.requires_grad_(False)
        self._emb_full = full
        self._emb_shard = torch.nn.Parameter(full[part.lo:part.lo + part.rows])      # same storage
        self._blocks = {}
        self._table_work = None
        return list(self.encoder.parameters()) + list(self.predictor.parameters()) + [self._emb_shard]

    def _table_wait(self):
        """the embedding table's all-gather of the previous step must have landed before it is read"""
        if getattr(self, "_table_work", None) is not None:
            self._table_work.wait()
            self._table_work = None

    def _shard_concat_feats(self, emb_full, data):
        """[emb | data.x] over the padded row range of the sharded layout (rows >= num_nodes are zero)"""
        hit = getattr(self, "_feat_pad", None)      # (feature tensor itself, padded copy, concat cache): compared by identity
        if hit is None or hit[0] is not data.x or hit[3] != data.x._version:
            feats = data.x.to(self.device).to(emb_full.dtype)
            pad = torch.zeros(self._shard.part.padded, feats.shape[1], dtype=feats.dtype, device=self.device)
            pad[:feats.shape[0]].copy_(feats)
            self._feat_pad = hit = (data.x, pad, {}, data.x._version)
        feats_pad, cache = hit[1], hit[2]
        if emb_full.is_cuda and isinstance(self.encoder, BaseGNN):
            return ops.concat_features(emb_full, feats_pad, cache)      # persistent 16-byte-aligned buffer
        return torch.cat([emb_full, feats_pad], dim=-1)

    def _adj_block(self, data):
        """this rank's destination-row slice of data.adj_t (built once per graph object)"""
        key = id(data.adj_t)
        hit = self._blocks.get(key)
        if hit is None or hit[0] is not data.adj_t:
            part = self._shard.part
            blk = data.adj_t.row_block(part.lo, part.rows, part.padded)
            self._blocks = {key: (data.adj_t, blk.to(self.device) if blk.device != self.device else blk)}
            hit = self._blocks[key]
        return hit[1]

    def _adj_block_square(self, data):
        """this rank's rows of data.adj_t kept at their global positions in a padded square graph (the row-sparse last
        layer of the sharded encoder: Graph.row_block_square); built once per graph object"""
        hit = getattr(self, "_square_blocks", None)
        if hit is None or hit[0] is not data.adj_t:
            part = self._shard.part
            sq = data.adj_t.row_block_square(part.lo, part.rows, part.padded)
            hit = self._square_blocks = (data.adj_t, sq.to(self.device) if sq.device != self.device else sq)
        return hit[1]

    # ------------------------------------------------------------------ setup ---
    def param_init(self):
        """model.py:92-96"""
        self.encoder.reset_parameters()
        self.predictor.reset_parameters()
        if self.emb is not None:
            torch.nn.init.xavier_uniform_(self.emb.weight)
        if self.process_group is not None:       # replicas must start identical
            for p in self.para_list:
                dense = ops.padded_base(p.data)          # (a table kept padded travels as its whole buffer)
                # (source = the group's first rank by its GLOBAL number: a sub-group need not contain global rank 0 --
                # bench.py's one-rank groups on ranks >= 1 failed here until the shared-GPU run of round 5 executed them)
                torch.distributed.broadcast(p.data if dense is None else dense,
                                            torch.distributed.get_global_rank(self.process_group, 0), group=self.process_group)

    def create_input_feat(self, data):
        """model.py:98-105.  The public surface: always the real [emb.weight | data.x] matrix, with its autograd edge to
        emb.weight -- whatever the caller does with it (slice, cast, a foreign conv) sees current values."""
        return self._input_feat(data, defer=False)

    def _input_feat(self, data, defer=True):
        """create_input_feat for the trainer's own passes, which hand the result STRAIGHT to encoder.convs[0]: with
        defer=True and a first GCNConv (which takes the two parts and never reads the concatenated matrix) the per-step copy
        of the embedding block is skipped and the returned tensor is marked stale (ops.concat_features(defer=True)); never
        returned to a caller outside this class."""
        if not self.use_node_feats:
            return self.emb.weight
        feat = data.x.to(self.device)
        if self.train_node_emb:
            if feat.is_cuda and isinstance(self.encoder, BaseGNN):
                # same values as torch.cat([emb.weight, x], -1), kept in a persistent padded buffer
                from .ops import concat_features
                if not hasattr(self, "_feat_cache"):
                    self._feat_cache = {}
                # (a first GCNConv takes the two parts and never reads the matrix: no per-step copy then)
                from .layer import GCNConv
                defer = (defer and len(self.encoder.convs) > 0 and isinstance(self.encoder.convs[0], GCNConv)
                         and ops.GCN_INPUT_FUSION["enabled"])
                return concat_features(self.emb.weight, feat, self._feat_cache, defer=defer)
            feat = torch.cat([self.emb.weight, feat], dim=-1)
        return feat

    def calculate_loss(self, pos_out, neg_out, num_neg, margin=None):
        """model.py:107-126: names that need a per-edge weight silently become
        plain auc_loss when the split has none; unknown names are auc_loss too."""
        if self._loss_override is not None:
            return self._loss_override(pos_out, neg_out, num_neg, margin)
        fn, weighted = loss_mod.BY_NAME.get(self.loss_func_name, (loss_mod.auc_loss, False))
        if self.loss_func_name == 'CE':
            return fn(pos_out, neg_out)
        if weighted:
            if margin is None:
                return loss_mod.auc_loss(pos_out, neg_out, num_neg)
            return fn(pos_out, neg_out, num_neg, margin)
        return fn(pos_out, neg_out, num_neg)

    def _loss_of_scores(self, out, n_pos, num_neg, margin=None):
        """calculate_loss on the step's single score tensor [pos | neg] without slicing it (the fused
        loss kernel then hands back ONE gradient tensor); falls back to the sliced call"""
        if self._loss_override is None and out.is_cuda:
            loss = loss_mod.joint_loss(self.loss_func_name, out, n_pos, num_neg, margin)
            if loss is not None:
                return loss
        return self.calculate_loss(out[:n_pos], out[n_pos:], num_neg, margin=margin)

    # ------------------------------------------------------------- DP helpers ---
    # losses that average over the batch (loss.py:45-62); every other one is a SUM over pairs
    MEAN_LOSSES = ("CE", "InfoNCE", "LogRank")

    def _slice_loss_scale(self, local: int, global_count: Optional[int]) -> float:
        """factor that turns the loss of this rank's slice into its share of the global batch's loss.
        A sum over pairs splits over slices as it is (1.0); a MEAN over the global batch is
        sum_r (n_r / n) * mean_r, so the slice's mean is weighted by n_r / n -- then the SUM over ranks
        of the slice losses (and of their gradients) equals the one-process loss (gradient) again."""
        if (self.process_group is None or global_count is None or local <= 0
                or self.loss_func_name not in self.MEAN_LOSSES):
            return 1.0
        return float(local) / float(global_count)

    def _world(self):
        if self.process_group is None:
            return 0, 1
        return (torch.distributed.get_rank(self.process_group),
                torch.distributed.get_world_size(self.process_group))

    def _embedding_grad_sink(self, x_in, replicated_update=False):
        """Data parallel + SAGE on the raw embedding table: deliver the (large) embedding gradient
        early so its all-reduce overlaps the encoder's weight-gradient GEMMs (ops.GradSink)."""
        from .layer import SAGEConv, GCNConv
        from .ops import GradSink
        self._early_work = None
        self._adam_sink = None
        parts = getattr(x_in, "_plnlp_parts", None)
        if (self.emb is not None and parts is not None and parts[0] is self.emb.weight and self.emb.weight.requires_grad
                and x_in.is_cuda and isinstance(self.encoder.convs[0], GCNConv)):
            # [table | constant features] into a first GCNConv (ops.GCNInputConvFn): no early all-reduce to start, but the
            # table's Adam step can ride in the aggregation that finishes its gradient -- same conditions as below
            if ((self.process_group is None or replicated_update) and self._fused_step and FUSE_EMBEDDING_ADAM["enabled"]
                    and ops.padded_base(self.emb.weight.data) is not None):
                adam = fused_adam_state(self.optimizer, self.emb.weight, padded=True)
                if adam is not None:
                    self._adam_sink = GradSink(None, None, adam=adam, like=self.emb.weight)
                    return self._adam_sink
            return None
        if (self.emb is None or x_in is not self.emb.weight
                or not self.emb.weight.requires_grad or not x_in.is_cuda
                or not isinstance(self.encoder.convs[0], SAGEConv) or self.encoder.convs[0].aggr != "mean"):
            return None
        adam = None
        if self.process_group is None or replicated_update:
            # (replicated_update: dp_exchange='scores' -- every rank back-propagates the whole global batch, so every
            # rank holds the complete gradient and applies the same deterministic update: nothing to reduce either)
            # One process: nothing to reduce over ranks, and the reference clips the encoder and the predictor but
            # NOT the embedding (model.py:163-165) -- Adam is the only consumer of the table's gradient, so the
            # kernel that finishes that gradient may apply the update itself (ops.GradSink.adam, PLNLP_EPI_ADAM)
            if not (self._fused_step and FUSE_EMBEDDING_ADAM["enabled"]):
                return None
            adam = fused_adam_state(self.optimizer, self.emb.weight)
            if adam is None:
                return None
        if adam is not None:
            # the fused update consumes the gradient in the kernel that finishes it: no [N, F] gradient buffer is
            # allocated unless the backward takes a path without the fused update (GradSink.buffer is lazy)
            self._adam_sink = GradSink(None, None, adam=adam, like=self.emb.weight)
            return self._adam_sink
        if getattr(self, "_emb_grad_buf", None) is None:
            self._emb_grad_buf = torch.empty_like(self.emb.weight)
        self.emb.weight.grad = self._emb_grad_buf

        def start_reduce():
            self._early_work = torch.distributed.all_reduce(self._emb_grad_buf, group=self.process_group,
                                                            async_op=True)
        return GradSink(self._emb_grad_buf, start_reduce)

    def _allreduce_grads(self):
        """SUM over ranks, issued before clipping so clipping sees the global gradient."""
        if self.process_group is None:
            return
        works = []
        early = getattr(self, "_early_work", None)
        for p in self.para_list:
            if early is not None and p is self.emb.weight:
                continue            # already in flight since the middle of the backward pass
            if p.grad is None:      # a rank whose slice was empty contributes zeros
                dense = ops.padded_base(p.data)
                p.grad = torch.zeros_like(p) if dense is None else torch.zeros_like(dense)[:, :p.shape[1]]
            dense = ops.padded_base(p.grad)              # (a padded table's gradient: the whole buffer, pad zeros included)
            works.append(torch.distributed.all_reduce(p.grad if dense is None else dense, group=self.process_group,
                                                      async_op=True))
        if early is not None:
            works.append(early)
            self._early_work = None
        for w in works:
            w.wait()

    def _clip_and_step(self):
        """model.py:163-167: encoder and predictor clipped as separate groups, the
        embedding not at all; then the optimiser."""
        sink = getattr(self, "_adam_sink", None)
        if sink is not None:
            if sink.adam_applied:       # the table was stepped inside the backward pass; its .grad stays None
                self.optimizer.state[self.emb.weight]["step"] += 1
            elif sink.buffer_used:      # the backward took a path without the fused update: the gradient is in the buffer
                self.emb.weight.grad = sink.buffer
            # (else: the sink was never asked -- the gradient went through autograd)
            self._adam_sink = None
        if self._fused_step:
            clip = {}
            if self.clip_norm >= 0:
                for group in (list(self.encoder.parameters()), list(self.predictor.parameters())):
                    sq = group_sqnorm(group)
                    if sq is not None:
                        clip.update({id(p): (sq, float(self.clip_norm)) for p in group})
            self.optimizer.step(clip=clip)
            return
        if self.clip_norm >= 0:
            for module in (self.encoder, self.predictor):
                params = [p for p in module.parameters()]
                if params:
                    torch.nn.utils.clip_grad_norm_(params, self.clip_norm)
        self.optimizer.step()

    def dp_mode(self) -> str:
        """the exchange actually used (see the class docstring)"""
        if self.process_group is None:
            return "none"
        if self._shard is not None:
            return "shard"
        scores_ok = not any(True for _ in self.predictor.parameters())
        if self.dp_exchange == "scores" and not scores_ok:
            raise ValueError("dp_exchange='scores' needs a predictor without parameters (DOT)")
        return "scores" if (scores_ok and self.dp_exchange in ("auto", "scores")) else "grads"

    @torch.no_grad()
    def check_replicas(self) -> bool:
        """True when every rank holds bit-identical parameters (two checksums, MIN == MAX over
        ranks).  'scores' mode never exchanges parameters or their gradients: the replicas agree
        because they compute the same deterministic update; this proves it."""
        if self.process_group is None:
            return True
        self._table_wait()
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        for p in self.para_list:
            d = p.detach().double()
            acc[0] += d.sum()
            acc[1] += (d * d).sum()
        lo, hi = acc.clone(), acc.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN, group=self.process_group)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX, group=self.process_group)
        return bool(torch.equal(lo, hi))

    def _tune_graph(self, graph):
        """once per (static) graph object: let the ranks of this model's group agree on the form of the
        aggregation kernel at the widths this encoder aggregates (ops.tune_aggregation -- rank 0's measurement
        is broadcast).  Every training-step entry point calls this first, i.e. at a point all ranks of the group
        pass together; the aggregation op itself never communicates.  One process: the same call without a group --
        the op's own lazy measurement only fires on a PLAIN launch, and a training step with the row-restricted
        forward and the mapped backward never makes one (it ran the default form until an evaluation pass tuned it)."""
        if not isinstance(self.encoder, BaseGNN) or graph.device.type != "cuda":
            return
        seen = getattr(self, "_tuned_graphs", None)
        if seen is None:
            seen = self._tuned_graphs = {}
        if seen.get(id(graph)) is graph:
            return
        feats = set()
        for conv in self.encoder.convs:
            if isinstance(conv, SAGEConv):
                feats.add(conv.in_channels)
            elif isinstance(conv, GCNConv):
                feats.add(conv.out_channels)
        ops.tune_aggregation(graph, feats, group=self.process_group)
        seen[id(graph)] = graph

    def _edge_flags(self, n_edges: int, on_gpu: bool, rows_only: bool):
        """what the scorer's fusions allow for a batch of n_edges scored edges -- a function of the model and the
        batch size alone, so a batch can be prepared before the step that uses it (prepare_edges):
        (fused scorer, row-sparse gradient channel, build the gather-backward index structures, touched-rows forward)"""
        n_endpoints = 2 * n_edges
        native = isinstance(self.encoder, BaseGNN)
        fused = native and type(self.predictor) in (DotPredictor, MLPPredictor)   # gather fused into the scorer
        use_channel = (n_edges > 0 and fused and all(isinstance(c, (SAGEConv, GCNConv)) for c in self.encoder.convs)
                       and getattr(self.encoder.convs[-1], "aggr", "mean") == "mean"
                       and ops.sparse_backward_pays(n_endpoints, self.num_nodes))
        build = n_edges > 0 and fused and on_gpu and ops.EDGE_BACKWARD["mode"] == "segment"
        sparse_fwd = rows_only and build and use_channel and ops.SPARSE_FORWARD["enabled"]
        return fused, use_channel, build, sparse_fwd

    def prepare_edges(self, pos_edge, neg_edge, edges_ready=True, rows_only=True):
        """Start the edge-only pre-processing of a FUTURE training step (ops.EdgeBatch: endpoint lists, the sort
        behind the deterministic gather backward, the touched-node compaction and its count read-back) on the
        side stream and return the handle to pass as train_step(..., prepared=).  A training loop that calls this
        for batch i+1 before it enqueues step i never waits for that read-back: without the look-ahead the host
        blocks on it at the top of every step while the main stream drains (a ~70 us hole per 1.7 ms collab step).
        pos_edge [b,2], neg_edge [b,k,2] as for train_step (this rank's slice)."""
        neg_flat = neg_edge.reshape(-1, 2)
        n_edges = pos_edge.size(0) + neg_flat.size(0)
        on_gpu = pos_edge.is_cuda
        fused, use_channel, build, sparse_fwd = self._edge_flags(n_edges, on_gpu, rows_only)
        batch = ops.EdgeBatch([pos_edge[:, 0], neg_flat[:, 0]], [pos_edge[:, 1], neg_flat[:, 1]], self.num_nodes,
                              build=build, compact=use_channel, overlap=on_gpu,
                              inputs_ready=edges_ready, compact_endpoints=sparse_fwd,
                              record_streams=not self._throttled())
        batch._flags = (n_edges, fused, use_channel, build, sparse_fwd)
        return batch

    def _encode(self, data, pos_edge, neg_flat, use_sink, edges_ready, rows_only=False, prepared=None,
                keep_alive=False):
        """encoder forward of a training step with the fusions the scorer allows; pos_edge [n,2] and
        neg_flat [n*k,2] are ALL edges whose scores will be back-propagated.  Returns (h, gate_scale,
        channel, fused, batch) with batch an ops.EdgeBatch (src, dst, incidence) -- see train_step.
        rows_only: the caller reads h ONLY through the fused scorer on exactly these edges, so the last
        conv may produce just the touched rows (ops.SPARSE_FORWARD); h is then compact (batch.src_c /
        dst_c address it) -- tell by `batch.src_c is not None`."""
        n_edges = pos_edge.size(0) + neg_flat.size(0)
        native = isinstance(self.encoder, BaseGNN)
        x_in = self._input_feat(data)
        fused, use_channel, build, sparse_fwd = self._edge_flags(n_edges, x_in.is_cuda, rows_only)
        # a 1-layer encoder ends in relu+dropout (layer.py:24-26); with a fused scorer as the only
        # consumer of h, that activation's backward rides in the scorer's gather-reduce epilogue
        fuse_gate = n_edges > 0 and fused and self.encoder.num_layers == 1
        kw = {}
        sink = None
        if native and use_sink:
            sink = self._embedding_grad_sink(x_in, replicated_update=(use_sink == "replicated"))
        if sink is not None:
            kw["input_grad_sink"] = sink
        # the batch touches at most n_endpoints nodes: the gradient of h is zero in every other row,
        # and the scorer hands it to the last conv's backward in row-sparse form
        channel = None
        if use_channel:
            channel = kw["output_grad_channel"] = ops.SparseGradChannel()
        # src / dst and the index structures of the gather backward depend on the edges alone: they were
        # started one step ahead (prepare_edges), or are built NOW -- on the side stream, in the shadow of the
        # previous step's tail -- and joined before the scorer
        if prepared is not None:
            if prepared._flags != (n_edges, fused, use_channel, build, sparse_fwd) or x_in.shape[0] != self.num_nodes:
                raise ValueError("prepare_edges() was called for a different batch shape or model configuration")
            batch = prepared
        else:
            batch = ops.EdgeBatch([pos_edge[:, 0], neg_flat[:, 0]], [pos_edge[:, 1], neg_flat[:, 1]], x_in.shape[0],
                                  build=build, compact=channel is not None, overlap=x_in.is_cuda,
                                  inputs_ready=edges_ready, compact_endpoints=sparse_fwd,
                                  record_streams=not (self._throttled() and keep_alive))
        if sparse_fwd:
            # the last conv needs the touched-row list (and its count, on the host) before it is launched:
            # the lists were started on the side stream at the top of the step, in the shadow of the
            # previous step's tail
            kw["output_rows"] = batch.join().incidence
        if fuse_gate:
            h, gate_scale = self.encoder(x_in, data.adj_t, fuse_output_gate=True, **kw)
        else:
            h, gate_scale = self.encoder(x_in, data.adj_t, **kw), 0.0
        return h, gate_scale, channel, fused, batch.join()

    def _score(self, h, src, dst):
        if hasattr(self.predictor, "score_edges"):
            return self.predictor.score_edges(h, src, dst)
        return self.predictor(h[src], h[dst])

    def train_step(self, data, pos_edge, neg_edge, num_neg, weight_margin=None, edges_ready=False,
                   global_count=None, prepared=None):
        """One iteration of the hot loop, model.py:148-167, on this rank's slice:
        pos_edge [b,2], neg_edge [b,k,2] (device).  Returns the detached local loss -- this rank's
        SHARE of the global batch's loss (global_count = positives in the global batch; only the
        batch-averaged losses need it, see _slice_loss_scale).
        edges_ready=True: the edge tensors are not the output of work still pending on the current
        stream (views of resident tensors, or produced on the side stream) -- their pre-processing
        may then overlap the previous step (ops.EdgeBatch)."""
        loss, batch = self._train_step_core(data, pos_edge, neg_edge, num_neg, weight_margin, edges_ready, global_count,
                                            prepared)
        self._throttle(keep=(batch, pos_edge, neg_edge, weight_margin))
        return loss

    def _train_step_core(self, data, pos_edge, neg_edge, num_neg, weight_margin, edges_ready, global_count, prepared):
        """everything train_step enqueues, without the host-side pacing: (detached loss, the step's EdgeBatch).
        This is also what plnlp_amd/capture.py captures in a hipGraph (nothing in here may touch the host once the
        model is warm: no read-back, no event wait)."""
        with ops.direct_table_grad():          # (a padded table takes its gradient in the padded layout: FusedAdam steps it there)
            return self._train_step_body(data, pos_edge, neg_edge, num_neg, weight_margin, edges_ready, global_count, prepared)

    def _train_step_body(self, data, pos_edge, neg_edge, num_neg, weight_margin, edges_ready, global_count, prepared):
        self._tune_graph(data.adj_t)
        self.optimizer.zero_grad(set_to_none=True)
        local = pos_edge.size(0)
        h, gate_scale, channel, fused, batch = self._encode(data, pos_edge, neg_edge.reshape(-1, 2), True,
                                                             edges_ready, rows_only=True, prepared=prepared,
                                                             keep_alive=True)
        src, dst, incidence = batch.src, batch.dst, batch.incidence
        if local > 0:
            if batch.src_c is not None:          # h holds only the touched rows (ops.SPARSE_FORWARD)
                out = self.predictor.score_edges(h, batch.src_c, batch.dst_c, gate_scale=gate_scale, channel=channel,
                                                 incidence=incidence, compact=True)
            else:
                out = (self.predictor.score_edges(h, src, dst, gate_scale=gate_scale, channel=channel,
                                                  incidence=incidence) if fused
                       else self._score(h, src, dst))
            loss = self._loss_of_scores(out, local, num_neg, weight_margin)
            scale = self._slice_loss_scale(local, global_count)
            if scale != 1.0:
                loss = loss * scale
        else:                                    # empty slice: still take part in the reduction
            loss = h.sum() * 0.0
        if loss.is_cuda and loss.dtype == torch.float32:
            loss.backward(ops.unit_grad(loss.device))
        else:
            loss.backward()
        self._allreduce_grads()
        self._clip_and_step()
        return loss.detach().reshape(()), batch

    def pipeline(self, data, num_neg: int, batch_size: int, weighted: bool, capture=None, tag=None):
        """the hot loop's driver for full batches of this shape on this graph (plnlp_amd/capture.py::StepPipeline:
        one batch of look-ahead, and -- once the model is warm -- the step replayed from two hipGraphs)"""
        from .capture import StepPipeline
        pipes = getattr(self, "_pipes", None)
        if pipes is None:
            pipes = self._pipes = {}
        # (tag: a caller that switches what the step launches -- GEMM form, full forward -- asks for its own
        # pipeline: a captured graph keeps the kernels it was captured with)
        key = (int(num_neg), int(batch_size), bool(weighted), capture, tag)
        hit = pipes.get(key)
        if hit is None or hit.data is not data or hit.graph is not data.adj_t:
            hit = pipes[key] = StepPipeline(self, data, num_neg, batch_size, weighted, capture=capture)
            hit.graph = data.adj_t
        return hit

    def _throttled(self) -> bool:
        return self.device.type == "cuda" and ops.STEP_THROTTLE["depth"] > 0

    def _throttle(self, keep=None):
        """bound how far the host runs ahead of the device, and hold `keep` (what this step borrowed from the
        side stream) until the step has finished there (ops.StepThrottle)"""
        if not self._throttled():
            return
        th = getattr(self, "_step_throttle", None)
        if th is None or th.depth != ops.STEP_THROTTLE["depth"]:
            th = self._step_throttle = ops.StepThrottle(ops.STEP_THROTTLE["depth"])
        th.tick(keep)

    def train_step_global(self, data, pos_edge, neg_edge, num_neg, weight_margin=None, edges_ready=False):
        """One iteration in dp_exchange='scores' mode.  Every rank passes the GLOBAL batch
        (pos_edge [n,2], neg_edge [n,k,2], weights [n]); rank r scores the slice
        [r*per, (r+1)*per), per = ceil(n / world), and differentiates the loss of that slice
        with respect to its scores; the per-edge score gradients are all-gathered and every
        rank back-propagates the whole batch through the encoder.  The update equals the
        one-process step on the global batch (the loss is a sum over pairs) and is the same
        bits on every rank.  Returns the detached loss of the local slice."""
        rank, world = self._world()
        self._tune_graph(data.adj_t)
        self.optimizer.zero_grad(set_to_none=True)
        n, k = pos_edge.size(0), num_neg
        per = (n + world - 1) // world
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        local = hi - lo
        h, gate_scale, channel, fused, batch = self._encode(data, pos_edge, neg_edge.reshape(-1, 2), "replicated",
                                                             edges_ready, keep_alive=True)
        src, dst, incidence = batch.src, batch.dst, batch.incidence
        # 1. local slice: scores (outside the encoder's graph), loss, d loss / d score
        g_pad = torch.zeros(per * (1 + k), dtype=h.dtype, device=h.device)
        loss = torch.zeros((), dtype=h.dtype, device=h.device)
        if local > 0:
            with torch.no_grad():
                out_l = self._score(h, torch.cat([src[lo:hi], src[n + lo * k:n + hi * k]]),
                                    torch.cat([dst[lo:hi], dst[n + lo * k:n + hi * k]]))
            out_l = out_l.detach().requires_grad_(True)
            loss = self._loss_of_scores(out_l, local, k, None if weight_margin is None else weight_margin[lo:hi])
            scale = self._slice_loss_scale(local, n)
            if scale != 1.0:
                loss = loss * scale
            if loss.is_cuda and loss.dtype == torch.float32:
                loss.backward(ops.unit_grad(loss.device))
            else:
                loss.backward()
            gl = out_l.grad.reshape(-1)
            g_pad[:local] = gl[:local]
            g_pad[per:per + local * k] = gl[local:]
        # 2. exchange the score gradients (rank-major order = global edge order)
        gathered = torch.empty(world * per * (1 + k), dtype=h.dtype, device=h.device)
        torch.distributed.all_gather_into_tensor(gathered, g_pad, group=self.process_group)
        gathered = gathered.view(world, per * (1 + k))
        g_all = torch.cat([gathered[:, :per].reshape(-1)[:n], gathered[:, per:].reshape(-1)[:n * k]])
        # 3. the whole batch back through the (replicated) encoder; the scores themselves are not needed
        if fused and isinstance(self.predictor, DotPredictor):
            out = self.predictor.score_edges(h, src, dst, gate_scale=gate_scale, channel=channel,
                                             compute_forward=False, incidence=incidence)
        elif fused:
            out = self.predictor.score_edges(h, src, dst, gate_scale=gate_scale, channel=channel,
                                             incidence=incidence)
        else:
            out = self._score(h, src, dst)
        out.backward(g_all.reshape(out.shape))
        self._clip_and_step()
        self._throttle(keep=(batch, pos_edge, neg_edge, weight_margin))
        return loss.detach().reshape(())

    def shard_plan(self, pos_edge, neg_edge, num_neg):
        """start the request plan of a FUTURE global batch on the side stream (plnlp_amd/shard.py::ShardPlan):
        it depends on the edges alone, so the trainer starts it one batch ahead -- its device work runs in
        the shadow of the current step and its count table is on the host by the time the step needs it.
        The edge tensors must not be the output of work still pending on the current stream."""
        from . import shard
        n = pos_edge.size(0)
        per = (n + self._shard.world - 1) // self._shard.world
        stream = None
        if pos_edge.is_cuda and ops.PROLOGUE_OVERLAP["enabled"]:
            stream = ops.side_stream(self.device)
        return shard.ShardPlan(self._shard.part, pos_edge, neg_edge.reshape(-1, 2), num_neg, per, stream=stream)

    def train_step_sharded(self, data, pos_edge, neg_edge, num_neg, weight_margin=None, plan=None):
        """One iteration in dp_exchange='shard' mode (plnlp_amd/shard.py).  Every rank passes the GLOBAL
        batch (pos_edge [n,2], neg_edge [n,k,2], weights [n]); rank r computes its block of rows of the
        encoder, scores the slice [r*per, (r+1)*per) of the batch on the rows that slice touches, and
        owns the update of its block of the embedding table.  The update equals the one-process step
        on the global batch (model.py:148-167) up to the order of floating-point sums.  Returns the
        detached loss of the local slice (its share of the global loss).
        plan: shard_plan(...) of this same batch, started earlier (else it is built here, in line)."""
        from . import shard
        sc = self._shard
        rank, world = sc.rank, sc.world
        self._tune_graph(self._adj_block(data))
        self.optimizer.zero_grad(set_to_none=True)
        n, k = pos_edge.size(0), num_neg
        per = (n + world - 1) // world
        neg_flat = neg_edge.reshape(-1, 2)
        fused = isinstance(self.encoder, BaseGNN) and type(self.predictor) in (DotPredictor, MLPPredictor)
        want_inc = fused and pos_edge.is_cuda and ops.EDGE_BACKWARD["mode"] == "segment"
        if plan is None:
            plan = shard.ShardPlan(sc.part, pos_edge, neg_flat, k, per)
        plan.finish(build_incidence=want_inc)
        lo, hi, local = plan.lo, plan.hi, plan.local
        self._table_wait()
        # the LAST layer row-sparse: only the rows of this rank's block that the global batch touches are computed
        # (and back-propagated) -- the single-process step's touched-rows forward / row-sparse backward on the block
        last_conv = self.encoder.convs[-1] if isinstance(self.encoder, BaseGNN) else None
        sparse_last = (last_conv is not None and isinstance(last_conv, SAGEConv) and last_conv.aggr == "mean"
                       and pos_edge.is_cuda and ops.SPARSE_FORWARD["enabled"] and SHARD_SPARSE["enabled"])
        rs_work = None
        if sparse_last:
            plan.join(record_streams=not self._throttled())
            channel = ops.SparseGradChannel()
            sink = None
            if self.encoder.num_layers == 1 and not self.use_node_feats:
                # one layer on the table itself: the partial gradient of all N rows is finished by the transposed
                # aggregation BEFORE the weight-gradient GEMMs are queued -- its reduce-scatter to the row owners
                # starts there and overlaps them (ops.GradSink)
                if getattr(self, "_emb_part_grad", None) is None:
                    self._emb_part_grad = torch.empty_like(self._emb_full)
                    self._emb_shard_grad = torch.empty_like(self._emb_shard)
                box = {}

                def start_reduce_scatter():
                    box["work"] = torch.distributed.reduce_scatter_tensor(self._emb_shard_grad, self._emb_part_grad,
                                                                          group=sc.group, async_op=True)
                sink = ops.GradSink(self._emb_part_grad, start_reduce_scatter)
                x_full = self._emb_full.detach().requires_grad_(True)
            else:
                x_full = sc.leaf(self._emb_shard, self._emb_full)
                if self.use_node_feats:
                    x_full = self._shard_concat_feats(x_full, data)
            h_c = self.encoder(x_full, self._adj_block(data), shard=sc,
                               shard_sparse=(self._adj_block_square(data), plan.block_rows, channel, sink))
            hq = shard.ExchangeCompactRows.apply(h_c, plan, sc.group, channel)
        else:
            x_full = sc.leaf(self._emb_shard, self._emb_full)
            if self.use_node_feats:               # model.py:98-105 on the padded row range: [emb | x]
                x_full = self._shard_concat_feats(x_full, data)
            h_block = self.encoder(x_full, self._adj_block(data), shard=sc)
            plan.join(record_streams=not self._throttled())          # (kept alive by the step throttle below instead)
            hq = shard.ExchangeRows.apply(h_block, plan, sc.group)             # [rows my slice touches, h]
        if local > 0:
            if want_inc:
                out = self.predictor.score_edges(hq, plan.src_c, plan.dst_c, incidence=plan.incidence)
            else:
                out = self._score(hq, plan.src_c, plan.dst_c)
            loss = self._loss_of_scores(out, local, k, None if weight_margin is None else weight_margin[lo:hi])
            scale = self._slice_loss_scale(local, n)
            if scale != 1.0:
                loss = loss * scale
        else:                                    # empty slice: still take part in every exchange
            loss = hq.sum() * 0.0 + (h_c if sparse_last else h_block).sum() * 0.0
        if loss.is_cuda and loss.dtype == torch.float32 and local > 0:
            loss.backward(ops.unit_grad(loss.device))
        else:
            loss.backward()
        small = []
        for p in list(self.encoder.parameters()) + list(self.predictor.parameters()):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            small.append(p.grad)
        shard.allreduce_sum(small, sc.group)
        if sparse_last and sink is not None:
            if "work" in box:
                box["work"].wait()
                self._emb_shard.grad = self._emb_shard_grad
            else:                               # no rank's slice touched this block's sources: nothing was queued
                self._emb_part_grad.zero_()
                torch.distributed.reduce_scatter_tensor(self._emb_shard_grad, self._emb_part_grad, group=sc.group)
                self._emb_shard.grad = self._emb_shard_grad
        if self._emb_shard.grad is None:
            self._emb_shard.grad = torch.zeros_like(self._emb_shard)
        self._clip_and_step()
        self._table_work = sc.sync_table(self._emb_full, self._emb_shard)
        self._throttle(keep=(plan, pos_edge, neg_edge, weight_margin))
        return loss.detach().reshape(())

    # ------------------------------------------------------------------ train ---
    def train(self, data, split_edge, batch_size, neg_sampler_name, num_neg):
        """model.py:128-173: one epoch; returns sum(loss_b * B_b) / sum(B_b).

        With a process group, `batch_size` is the per-rank batch when
        dp_scaling == 'weak' (global batch = world * batch_size) and the global
        batch when 'strong'; either way each global batch is cut into `world`
        contiguous slices."""
        self.encoder.train()
        self.predictor.train()
        rank, world = self._world()
        mode = self.dp_mode()
        if world > 1 and self.device.type == "cuda":
            # replicated encoder passes must draw the same dropout masks: rank 0's stream state wins
            st = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in ops.seed_state()], dtype=torch.int64,
                              device=self.device)
            torch.distributed.broadcast(st, torch.distributed.get_global_rank(self.process_group, 0), group=self.process_group)
            base, counter = (int(v) for v in st.tolist())
            ops.set_seed_state(base, counter)

        # the structured ("global") samplers run on the device the edge list lives on: on the host they
        # cost ten times the epoch's GPU time at collab scale (seeded from the CPU generator either way)
        import time
        t_epoch = time.perf_counter()
        edge_index = data.edge_index
        if neg_sampler_name != 'local' and edge_index is not None and self.device.type == "cuda":
            edge_index = edge_index.to(self.device)
        pos_train_edge, neg_train_edge = get_pos_neg_edges(
            'train', split_edge, edge_index=edge_index, num_nodes=self.num_nodes,
            neg_sampler_name=neg_sampler_name, num_neg=num_neg)
        pos_train_edge, neg_train_edge = pos_train_edge.to(self.device), neg_train_edge.to(self.device)
        edge_weight_margin = None
        if 'weight' in split_edge['train']:
            edge_weight_margin = split_edge['train']['weight'].to(self.device)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)       # once per epoch: separates the sampler's time from the steps'
        t_sampled = time.perf_counter()

        global_batch = batch_size * world if (world > 1 and self.dp_scaling == "weak") else batch_size
        # the DataLoader permutation (model.py:147), bit-exact; on the device path it is STREAMED: a host thread
        # shuffles a few batches ahead of the GPU instead of the whole epoch before the first step
        n_train = pos_train_edge.size(0)
        streamed = None
        if self.device.type == "cuda" and 0 < n_train < StreamedPermutation.LIMIT and STREAM_PERMUTATION["enabled"]:
            streamed = StreamedPermutation(n_train, global_batch, self.device)
            sizes = streamed.sizes
            perm_stream = (ops.side_stream(self.device) if ops.PROLOGUE_OVERLAP["enabled"]
                           else torch.cuda.current_stream(self.device))

            def perm_of(bi):
                return streamed.batch(bi, perm_stream)
        else:
            batches = batch_permutation(n_train, global_batch, True)
            order = torch.cat(batches).to(self.device) if batches else None
            sizes = [b.numel() for b in batches]
            offsets = [0]
            for sz in sizes:
                offsets.append(offsets[-1] + sz)

            def perm_of(bi):
                return order[offsets[bi]:offsets[bi + 1]]
        t_permuted = time.perf_counter()          # (host time in FRONT of the first step)

        # Python-float accumulation in the reference (model.py:169); here a resident double that, in one process, the
        # loss kernel itself feeds (ops.LOSS_ACC: no cast / multiply / add launches per step).  Persistent: a captured
        # step keeps pointing at it
        acc_buf = getattr(self, "_loss_acc_buf", None)
        if acc_buf is None or acc_buf.device != self.device:
            acc_buf = self._loss_acc_buf = torch.zeros(1, dtype=torch.float64, device=self.device)
        acc_buf.zero_()
        loss_acc = acc_buf.reshape(())
        total_examples = 0
        # the per-batch gathers of the epoch tensors run on the side stream, like the rest of a batch's
        # pre-processing (ops.EdgeBatch): they depend on nothing the training steps produce
        side = main = None
        if self.device.type == "cuda" and ops.PROLOGUE_OVERLAP["enabled"] and mode != "shard":
            main, side = torch.cuda.current_stream(self.device), ops.side_stream(self.device)
            side.wait_stream(main)              # the epoch tensors above

        def take(perm):
            if side is None:
                return (pos_train_edge[perm], neg_train_edge[perm],
                        edge_weight_margin[perm] if edge_weight_margin is not None else None)
            with torch.cuda.stream(side):
                out = (pos_train_edge[perm], neg_train_edge[perm],
                       edge_weight_margin[perm] if edge_weight_margin is not None else None)
            if not self._throttled():        # (else the step hands them to the step throttle instead)
                for t in out:
                    if t is not None:
                        t.record_stream(main)
            return out

        pending = None

        def shard_batch(perm, first):
            """gather one global batch on the side stream and start its request plan there"""
            if self.device.type == "cuda" and ops.PROLOGUE_OVERLAP["enabled"]:
                cur, sd = torch.cuda.current_stream(self.device), ops.side_stream(self.device)
                if first:
                    sd.wait_stream(cur)                 # the epoch tensors were produced on the current stream
                with torch.cuda.stream(sd):
                    got = take(perm)
                if not self._throttled():                   # (else train_step_sharded hands them to the step throttle)
                    for t in got:
                        if t is not None:
                            t.record_stream(cur)
            else:
                got = take(perm)
            return got + (self.shard_plan(got[0], got[1], num_neg),)

        for bi, n_b in enumerate(sizes):
            perm_all = perm_of(bi)
            if mode == "shard":
                # one batch of look-ahead: the request plan of the NEXT batch is started before this step
                # is enqueued, so its device work and its count read-back hide behind this step
                if pending is None:
                    pending = shard_batch(perm_all, True)
                pos_b, neg_b, weight_margin, plan_b = pending
                pending = None
                if bi + 1 < len(sizes):
                    pending = shard_batch(perm_of(bi + 1), False)
                # (the step touches the batch tensors only after plan.join(), which orders the main stream
                # behind everything the side stream did for this batch, the gather included)
                loss = self.train_step_sharded(data, pos_b, neg_b, num_neg, weight_margin, plan=plan_b)
                loss_acc += loss.double() * n_b
                total_examples += n_b
                continue
            if world > 1 and mode == "scores":
                pos_b, neg_b, weight_margin = take(perm_all)
                loss = self.train_step_global(data, pos_b, neg_b, num_neg, weight_margin, edges_ready=side is not None)
                loss_acc += loss.double() * n_b
                total_examples += n_b
                continue
            def my_slice(perm_of_batch):
                if world > 1:
                    per = (perm_of_batch.numel() + world - 1) // world
                    return perm_of_batch[rank * per:(rank + 1) * per]
                return perm_of_batch

            pipe = None
            if world == 1 and side is not None:
                # one process: the look-ahead and the step itself go through the step pipeline (captured in two
                # hipGraphs once the model is warm)
                pipe = self.pipeline(data, num_neg, batch_size, edge_weight_margin is not None)

            def gather_and_prepare(perm_of_batch):
                got = take(my_slice(perm_of_batch))
                if pipe is not None:
                    return got + (pipe.prepare(got[0], got[1], got[2]),)
                pb = self.prepare_edges(got[0], got[1], edges_ready=side is not None) if side is not None else None
                return got + (pb,)
            # one batch of look-ahead, as in the sharded mode: the next batch's gathers and index structures are
            # started before this step is enqueued, so their count read-back never stalls the host
            if pending is None:
                pending = gather_and_prepare(perm_all)
            pos_b, neg_b, weight_margin, prepared = pending
            pending = None
            if bi + 1 < len(sizes):
                pending = gather_and_prepare(perm_of(bi + 1))
            feed = FUSE_LOSS_ACC["enabled"] and world == 1 and self.device.type == "cuda"
            fed0 = ops.LOSS_ACC["fed"]
            if feed:
                ops.LOSS_ACC.update(buf=acc_buf, weight=float(n_b))
            try:
                if pipe is not None:
                    loss = pipe.step(prepared, global_count=n_b)
                    fed = pipe.last_fed
                else:
                    loss = self.train_step(data, pos_b, neg_b, num_neg, weight_margin, edges_ready=side is not None,
                                           global_count=n_b, prepared=prepared)
                    fed = ops.LOSS_ACC["fed"] != fed0
            finally:
                ops.LOSS_ACC.update(buf=None, weight=0.0)
            if not (feed and fed):           # a loss outside the fused kernel, or several ranks: the reference's way
                loss_acc += loss.double() * n_b
            total_examples += n_b

        if mode == "shard":
            self._table_wait()
        if world > 1 or mode == "shard":
            torch.distributed.all_reduce(loss_acc, group=self.process_group)
            if mode in ("scores", "shard") and not self.check_replicas():
                raise RuntimeError("data-parallel replicas diverged (dp_exchange='scores' relies on every rank "
                                   "computing the same deterministic update)")
        epoch_loss = loss_acc.item() / max(total_examples, 1)          # (the epoch's one read-back: the device is idle after it)
        t_end = time.perf_counter()
        # what SURVEY.md 8(d) defines the metric on: Sum_steps B_step * (1 + k) over the wall time of train(), the
        # negative sampler's time reported separately (model.py:132-136 draws them once per epoch, before the loop)
        if streamed is not None:
            streamed.join()
        self.last_epoch = {"steps": len(sizes), "positives": total_examples,
                           "edges_scored": total_examples * (1 + num_neg),
                           "sampler_s": t_sampled - t_epoch, "permutation_s": t_permuted - t_sampled,
                           "steps_s": t_end - t_permuted, "epoch_s": t_end - t_epoch}
        return epoch_loss

    # ------------------------------------------------------------------- eval ---
    @torch.no_grad()
    def batch_predict(self, h, edges, batch_size, to_cpu=True):
        """model.py:175-182.  The reference copies every batch of scores to the host
        (`.squeeze().cpu()` per batch); here they stay on the device, are concatenated once, and
        move only if the caller asks (test() ranks them on the device)."""
        preds = []
        batch_permutation(edges.size(0), batch_size, False)      # RNG parity: the loader's base-seed draw
        for lo in range(0, edges.size(0), batch_size):
            edge = edges[lo:lo + batch_size]
            preds.append(self._score(h, edge[:, 0], edge[:, 1]).reshape(-1))
        out = torch.cat(preds, dim=0) if preds else torch.empty(0, device=h.device)
        return out.cpu() if to_cpu else out

    @torch.no_grad()
    def test(self, data, split_edge, batch_size, evaluator, eval_metric):
        """model.py:184-226.  The reference recomputes the (deterministic,
        eval-mode) encoder output a second time before the test split
        (model.py:204-206); that pass is redundant and is not repeated."""
        self.encoder.eval()
        self.predictor.eval()
        self._table_wait()

        h = self.encoder(self._input_feat(data), data.adj_t)
        # index -1 = unseen node = mean of all seen representations (model.py:191-194)
        h = torch.cat([h, torch.mean(h, dim=0, keepdim=True)], dim=0)

        preds = {}
        on_device = self.device.type == "cuda"      # Hits@K / MRR are torch ops: rank where the scores are
        for split in ('valid', 'test'):
            pos_edge, neg_edge = get_pos_neg_edges(split, split_edge)
            preds[split] = (self.batch_predict(h, pos_edge.to(self.device), batch_size, to_cpu=not on_device),
                            self.batch_predict(h, neg_edge.to(self.device), batch_size, to_cpu=not on_device))
        # batch_predict consumed base-seed draws as the reference's loaders do (4 loaders)
        fn = evaluate_hits if eval_metric == 'hits' else evaluate_mrr
        return fn(evaluator, preds['valid'][0], preds['valid'][1], preds['test'][0], preds['test'][1])


# -------------------------------------------------------------------- factories --
def create_input_layer(num_nodes, num_node_feats, hidden_channels, use_node_feats=True,
                       train_node_emb=False, pretrain_emb=None):
    """model.py:229-249 -> (input width of the encoder, embedding or None)"""
    have_pretrained = pretrain_emb is not None and pretrain_emb != ''
    emb = None
    input_dim = num_node_feats if use_node_feats else 0
    if train_node_emb and use_node_feats or (not use_node_feats and not have_pretrained):
        emb = torch.nn.Embedding(num_nodes, hidden_channels)
        input_dim += hidden_channels
    elif have_pretrained:
        emb = torch.nn.Embedding.from_pretrained(torch.load(pretrain_emb))
        input_dim += emb.weight.size(1)
    return input_dim, emb


_ENCODERS = {'GCN': GCN, 'WSAGE': WSAGE, 'TRANSFORMER': Transformer}
_PREDICTORS = {
    'DOT': lambda h, n, p: DotPredictor(),
    'BIL': lambda h, n, p: BilinearPredictor(h),
    'MLP': lambda h, n, p: MLPPredictor(h, h, 1, n, p),
    'MLPDOT': lambda h, n, p: MLPDotPredictor(h, 1, n, p),
    'MLPBIL': lambda h, n, p: MLPBilPredictor(h, 1, n, p),
    'MLPCAT': lambda h, n, p: MLPCatPredictor(h, h, 1, n, p),
}


def create_gnn_layer(input_channels, hidden_channels, num_layers, dropout=0, encoder_name='SAGE'):
    """model.py:252-260: name matched case-insensitively, anything unknown is SAGE"""
    cls = _ENCODERS.get(encoder_name.upper(), SAGE)
    return cls(input_channels, hidden_channels, hidden_channels, num_layers, dropout)


def create_predictor_layer(hidden_channels, num_layers, dropout=0, predictor_name='MLP'):
    """model.py:263-276: unknown names give None, as the reference does"""
    make = _PREDICTORS.get(predictor_name.upper())
    return None if make is None else make(hidden_channels, num_layers, dropout)


def adjust_lr(optimizer, decay_ratio, lr):
    """model.py:279-286: linear decay, floor at 1e-4 * lr"""
    new_lr = max(lr * (1 - decay_ratio), lr * 0.0001)
    for group in optimizer.param_groups:
        group['lr'] = new_lr
    return new_lr
