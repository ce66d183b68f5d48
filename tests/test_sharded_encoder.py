"""Row-sharded encoder (plnlp_amd/shard.py, BaseModel dp_exchange='shard') on CPU with gloo, world 2.
The compute modules are the CPU oracle's, wrapped to follow the sharded-forward contract the product's
BaseGNN implements (block aggregation from the full source matrix, all-gather between layers); what is
under test is everything around them: the row partition and CSR blocks, the request plan (which rows
each rank's edge slice touches, who owns them), the all-to-all of rows and of their gradients with
its fixed summation order, the reduce-scatter of the embedding gradient, Adam on the owned rows only,
the table all-gather, loss accounting.  Claim: W ranks == one process on the global batch (float64)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import oracle as O

N, H, K, EPOCHS = 121, 8, 2, 2          # 121 rows over 2 ranks: blocks of 64 with 7 padding rows


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    g = torch.Generator().manual_seed(3)
    a = torch.randint(0, N, (520,), generator=g)
    b = torch.randint(0, N, (520,), generator=g)
    keep = a != b
    lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
    key = torch.unique(lo * N + hi)
    lo, hi = key // N, key % N
    w = torch.rand(lo.numel(), generator=g) * 0.8 + 0.2
    return torch.cat([lo, hi]), torch.cat([hi, lo]), torch.stack([lo, hi], 1), w


class ShardableGNNRef(torch.nn.Module):
    """oracle SAGE / GCN encoder; with shard= it computes one destination-row block the way
    plnlp_amd.BaseGNN._forward_sharded does, on the oracle's own conv arithmetic"""

    def __init__(self, ref: O.GNNRef):
        super().__init__()
        self.ref = ref

    def reset_parameters(self):
        self.ref.reset_parameters()

    def forward(self, x, adj, shard=None):
        val = None if adj.val is None else adj.val.to(x.dtype)
        csr = O.CSR(adj.rowptr, adj.col.to(torch.int64), val, adj.n_cols)
        if shard is None:
            return self.ref(x, csr)
        n = len(self.ref.convs)
        for i, conv in enumerate(self.ref.convs):
            if isinstance(conv, O.GCNConvRef):        # aggregate first, like ops.GCNConvBlockFn
                y = conv.lin(O.spmm(csr, x, "sum", use_values=True)) + conv.bias
            else:
                y = conv.lin_l(O.spmm(csr, x, "mean", use_values=False)) + conv.lin_r(x[shard.lo:shard.lo + shard.rows])
            if i < n - 1 or self.ref.num_layers == 1:
                y = F.relu(y)
            x = shard.all_gather(y) if i < n - 1 else y
        return x


class _Data:
    pass


def _run(rank, world, port, batch, scaling, loss_name, predictor, layers, out_q, kind="SAGE", feats=0):
    import plnlp_amd as P
    pg = None
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        pg = dist.group.WORLD
    torch.set_num_threads(1)
    torch.manual_seed(100)
    row, col, pos, w = _problem()
    enc = ShardableGNNRef(O.GNNRef(kind, H + feats, H, H, layers, 0.0)).double()
    pred = (O.DotPredictorRef() if predictor == "DOT" else O.MLPPredictorRef(H, H, 1, 2, 0.0)).double()
    m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=layers, mlp_num_layers=2,
                    emb_hidden_channels=H, gnn_hidden_channels=H, mlp_hidden_channels=H, num_nodes=N,
                    num_node_feats=feats, gnn_encoder_name=kind, predictor_name=predictor, loss_func=loss_name,
                    optimizer_name="Adam", device="cpu", use_node_feats=feats > 0, train_node_emb=True,
                    modules=(enc, pred, lambda p_, n_, k_, w_: O.LOSSES[O.select_loss(loss_name, w_ is not None)](
                        p_, n_, k_, w_)),
                    process_group=pg, dp_scaling=scaling, dp_exchange="shard" if world > 1 else "auto")
    if world > 1:
        assert m.dp_mode() == "shard"
        from plnlp_amd.shard import RowPartition
        part_ = RowPartition(N, world, rank)
        assert m._emb_shard.shape[0] == part_.rows and m._emb_full.shape[0] == part_.padded
    m.emb.double()
    if world > 1:            # .double() re-seated the table: put the float64 table back on the sharded layout
        full = torch.zeros(m._emb_full.shape, dtype=torch.float64)
        m._emb_full = full
        m.emb.weight.data = full[:N]
        m._emb_shard.data = full[m._shard.lo:m._shard.lo + m._shard.rows]
    m.param_init()
    data = _Data()
    data.adj_t = P.Graph.from_coo(row, col, None, N, N)
    if kind == "GCN":
        data.adj_t = P.gcn_normalization(data.adj_t)
    if feats:
        data.x = torch.randn(N, feats, generator=torch.Generator().manual_seed(12)).double()
    data.edge_index = torch.stack([col, row])
    split = {"train": {"edge": pos, "weight": w.double()}}
    torch.manual_seed(200)
    losses = [m.train(data, split, batch, "local", K) for _ in range(EPOCHS)]
    small = torch.cat([p.detach().reshape(-1) for p in list(m.encoder.parameters()) + list(m.predictor.parameters())])
    table = m.emb.weight.detach().reshape(-1).clone()
    if out_q is not None:
        out_q.put((rank, losses, small.double().numpy(), table.double().numpy()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    else:
        torch.set_num_threads(os.cpu_count() or 1)
    return losses, small.double().numpy(), table.double().numpy()


def _spawn(world, *args, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port) + args + (q,), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("scaling,loss_name,predictor,layers",
                         [("strong", "AUC", "DOT", 1), ("weak", "WeightedHingeAUC", "DOT", 2),
                          ("strong", "AUC", "MLP", 2), ("strong", "LogRank", "DOT", 1)])
def test_sharded_two_ranks_equal_one_process(scaling, loss_name, predictor, layers):
    B = 64
    single_batch = B if scaling == "strong" else 2 * B
    ref_losses, ref_small, ref_table = _run(0, 1, 0, single_batch, scaling, loss_name, predictor, layers, None)
    res = _spawn(2, B, scaling, loss_name, predictor, layers)
    for rank, losses, small, table in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9, err_msg=f"rank {rank}")
        np.testing.assert_allclose(small, ref_small, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(table, ref_table, rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(res[0][2], res[1][2])        # small weights: bit-identical replicas
    np.testing.assert_array_equal(res[0][3], res[1][3])        # and every rank holds the same full table


def test_sharded_gcn_with_node_features_equals_one_process():
    """the citation2 recipe's shape: GCN x2 on [emb | features], MLP predictor -- GCN blocks aggregate first
    (ops.GCNConvBlockFn), the input is [table all-gathered | features] over the padded row range"""
    B = 64
    ref_losses, ref_small, ref_table = _run(0, 1, 0, B, "strong", "AUC", "MLP", 2, None, kind="GCN", feats=5)
    res = _spawn(2, B, "strong", "AUC", "MLP", 2, kind="GCN", feats=5)
    for rank, losses, small, table in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9, err_msg=f"rank {rank}")
        np.testing.assert_allclose(small, ref_small, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(table, ref_table, rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][3], res[1][3])


def test_sharded_uneven_last_batch_and_empty_slice():
    """a last global batch of one edge leaves rank 1 without edges: it still joins every exchange"""
    n_pos = _problem()[2].size(0)
    B = n_pos - 1
    ref_losses, ref_small, ref_table = _run(0, 1, 0, B, "strong", "AUC", "DOT", 2, None)
    res = _spawn(2, B, "strong", "AUC", "DOT", 2)
    for rank, losses, small, table in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9)
        np.testing.assert_allclose(table, ref_table, rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(res[0][3], res[1][3])


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_four_and_eight_ranks_equal_one_process(world):
    """the row-sharded step at world 4 and 8: N is not a multiple of the world (blocks of uneven real height, the last
    ones partly padding), weak scaling, and a last global batch of one edge (empty slices on all ranks but one)"""
    B = 16
    ref_losses, ref_small, ref_table = _run(0, 1, 0, world * B, "weak", "WeightedHingeAUC", "DOT", 2, None)
    res = _spawn(world, B, "weak", "WeightedHingeAUC", "DOT", 2)
    assert len(res) == world
    for rank, losses, small, table in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9, err_msg=f"rank {rank}")
        np.testing.assert_allclose(small, ref_small, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(table, ref_table, rtol=1e-6, atol=1e-9)
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][2], r[2])
        np.testing.assert_array_equal(res[0][3], r[3])
    n_pos = _problem()[2].size(0)
    ref_losses, _, ref_table = _run(0, 1, 0, n_pos - 1, "strong", "AUC", "DOT", 1, None)
    res = _spawn(world, n_pos - 1, "strong", "AUC", "DOT", 1)
    for rank, losses, small, table in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9)
        np.testing.assert_allclose(table, ref_table, rtol=1e-6, atol=1e-9)


def test_shard_plan_lists_exactly_the_touched_rows():
    """ShardPlan on one process for every (asker, owner) pair: row lists == the sorted distinct
    endpoints of each slice, split by owner; compact edge coordinates address the right rows"""
    from plnlp_amd.shard import RowPartition, ShardPlan
    g = torch.Generator().manual_seed(7)
    n_nodes, W, n, k = 103, 4, 37, 3
    pos = torch.randint(0, n_nodes, (n, 2), generator=g)
    neg = torch.randint(0, n_nodes, (n * k, 2), generator=g)
    per = (n + W - 1) // W
    plans = [ShardPlan(RowPartition(n_nodes, W, r), pos, neg, k, per).finish() for r in range(W)]
    S = plans[0].send_rows.new_tensor(RowPartition(n_nodes, W, 0).rows).item()
    for q, p in enumerate(plans):
        lo, hi = min(q * per, n), min((q + 1) * per, n)
        src = torch.cat([pos[lo:hi, 0], neg[lo * k:hi * k, 0]])
        dst = torch.cat([pos[lo:hi, 1], neg[lo * k:hi * k, 1]])
        touched = torch.unique(torch.cat([src, dst]))
        assert p.count == touched.numel() and torch.equal(p.rows, touched)
        assert torch.equal(touched[p.src_c], src) and torch.equal(touched[p.dst_c], dst)
        assert p.out_splits == [int(((touched >= r * S) & (touched < (r + 1) * S)).sum()) for r in range(W)]
        for r in range(W):                      # what owner r sends to asker q
            seg_lo = sum(plans[r].in_splits[:q])
            got = plans[r].send_rows[seg_lo:seg_lo + plans[r].in_splits[q]] + r * S
            assert torch.equal(got, touched[(touched >= r * S) & (touched < (r + 1) * S)])


def test_shard_plan_block_rows_and_gradient_order():
    """the row-sparse last layer's index structures (ShardPlan.block_rows / send_pos / back_csr): the touched rows of
    every rank's block == the union over ALL slices of the endpoints that fall into the block; send_pos addresses
    them in the compact matrix; the gradient rows returning through the all-to-all are added per compact row in
    asker order (a fixed order: the CSR lists, per row, the positions in the returned buffer with askers increasing)"""
    from plnlp_amd.shard import RowPartition, ShardPlan
    g = torch.Generator().manual_seed(11)
    n_nodes, W, n, k = 203, 4, 61, 2
    pos = torch.randint(0, n_nodes, (n, 2), generator=g)
    neg = torch.randint(0, n_nodes, (n * k, 2), generator=g)
    per = (n + W - 1) // W
    everything = torch.unique(torch.cat([pos.reshape(-1), neg.reshape(-1)]))
    for r in range(W):
        part = RowPartition(n_nodes, W, r)
        p = ShardPlan(part, pos, neg, k, per).finish()
        br = p.block_rows
        mine = everything[(everything >= part.lo) & (everything < part.lo + part.rows)]
        assert br.count == mine.numel() and br.n_rows % 32 == 0 or br.n_rows == part.padded
        assert torch.equal(br.rows[:br.count].long(), mine) and bool((br.rows[br.count:] == 0).all())
        want_map = torch.full((part.padded,), -1, dtype=torch.int32)
        want_map[mine] = torch.arange(mine.numel(), dtype=torch.int32)
        assert torch.equal(br.node_map, want_map)
        assert torch.equal(br.rows[p.send_pos].long(), p.send_rows + part.lo)          # positions address the sent rows
        seg, order = p.back_csr
        assert seg.numel() == br.n_rows + 1 and int(seg[-1]) == p.send_pos.numel()
        asker_of = torch.repeat_interleave(torch.arange(W), torch.tensor(p.in_splits))
        for c in range(br.n_rows):
            items = order[seg[c]:seg[c + 1]].long()
            assert bool((p.send_pos[items] == c).all())
            a = asker_of[items]
            assert bool((a[1:] > a[:-1]).all())                # one entry per asker, askers increasing


class _CompactRowsFn(torch.autograd.Function):
    """stand-in for the row-restricted last conv in the CPU test: rows `block_rows` of the rank's block of a shared
    parameter; its gradient arrives row-sparse through the channel, like ops.SAGEConvFn's"""

    @staticmethod
    def forward(ctx, table, block_rows, channel):
        ctx.block_rows, ctx.channel, ctx.shape = block_rows, channel, table.shape
        ctx.set_materialize_grads(False)
        return table.index_select(0, block_rows.rows.long())

    @staticmethod
    def backward(ctx, gy):
        sg = ctx.channel.take()
        br = ctx.block_rows
        out = torch.zeros(ctx.shape, dtype=sg.values.dtype)
        out.index_add_(0, br.rows[:br.count].long(), sg.values[:br.count])
        return out, None, None


def _compact_exchange_rank(rank, world, port, out_q):
    from plnlp_amd import ops
    from plnlp_amd.shard import ExchangeCompactRows, RowPartition, ShardPlan
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    n_nodes, n, k, h = 77, 40, 2, 6
    pos = torch.randint(0, n_nodes, (n, 2), generator=g)
    neg = torch.randint(0, n_nodes, (n * k, 2), generator=g)
    table = torch.randn(n_nodes, h, generator=g, dtype=torch.float64)
    coef = torch.randn(n * (1 + k), generator=g, dtype=torch.float64)
    part = RowPartition(n_nodes, world, rank)
    per = (n + world - 1) // world
    plan = ShardPlan(part, pos, neg, k, per).finish()
    full = torch.zeros(part.padded, h, dtype=torch.float64)
    full[:n_nodes] = table
    full.requires_grad_(True)
    channel = ops.SparseGradChannel()
    h_c = _CompactRowsFn.apply(full, plan.block_rows, channel)
    hq = ExchangeCompactRows.apply(h_c, plan, dist.group.WORLD, channel)
    lo, hi = plan.lo, plan.hi
    c_loc = torch.cat([coef[lo:hi], coef[n + lo * k:n + hi * k]])
    score = (hq[plan.src_c] * hq[plan.dst_c]).sum(-1)
    loss = (c_loc * score * score).sum()
    loss.backward()
    grad = full.grad.clone()
    dist.all_reduce(grad)                     # every rank's partial (non-zero only inside its own block)
    loss_all = loss.detach().clone()
    dist.all_reduce(loss_all)
    out_q.put((rank, float(loss_all), grad[:n_nodes].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_compact_row_exchange_two_ranks_equal_one_process():
    """ExchangeCompactRows over gloo, world 2: loss and d loss / d table == the one-process computation (float64)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_compact_exchange_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    n_nodes, n, k, h = 77, 40, 2, 6
    pos = torch.randint(0, n_nodes, (n, 2), generator=g)
    neg = torch.randint(0, n_nodes, (n * k, 2), generator=g)
    table = torch.randn(n_nodes, h, generator=g, dtype=torch.float64).requires_grad_(True)
    coef = torch.randn(n * (1 + k), generator=g, dtype=torch.float64)
    src, dst = torch.cat([pos[:, 0], neg[:, 0]]), torch.cat([pos[:, 1], neg[:, 1]])
    score = (table[src] * table[dst]).sum(-1)
    loss = (coef * score * score).sum()
    loss.backward()
    for rank, l, gr in res:
        np.testing.assert_allclose(l, float(loss), rtol=1e-12)
        np.testing.assert_allclose(gr, table.grad.numpy(), rtol=1e-10, atol=1e-12)
