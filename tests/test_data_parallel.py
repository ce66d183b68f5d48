"""Edge-batch data parallelism of plnlp_amd.BaseModel, exercised on CPU with the
gloo backend (world_size 2).  The compute modules are the CPU oracle's (injected
through `modules=`), so what is under test is the host logic the GPU path shares:
batch slicing, SUM all-reduce before clipping, identical replicas, loss
accounting.  Claim checked: W ranks on slices of a global batch == one process on
the whole batch (the loss is a sum over pairs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O

N, H, K, EPOCHS = 120, 8, 2, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    g = torch.Generator().manual_seed(3)
    a = torch.randint(0, N, (500,), generator=g)
    b = torch.randint(0, N, (500,), generator=g)
    keep = a != b
    lo, hi = torch.minimum(a, b)[keep], torch.maximum(a, b)[keep]
    key = torch.unique(lo * N + hi)
    lo, hi = key // N, key % N
    adj = O.CSR.from_coo(torch.cat([lo, hi]), torch.cat([hi, lo]), None, N)
    w = torch.rand(lo.numel(), generator=g) * 0.8 + 0.2
    return adj, torch.stack([lo, hi], 1), w


class _Data:
    pass


def _run(rank, world, port, batch, scaling, loss_name, predictor, out_q, exchange="auto"):
    import plnlp_amd as P
    pg = None
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        pg = dist.group.WORLD
    torch.set_num_threads(1)
    torch.manual_seed(100)                       # same init + same index streams on every rank
    adj, pos, w = _problem()
    # float64 compute modules: the equivalence under test is exact in real arithmetic; in fp32 the
    # reassociated gradient sum perturbs near-zero gradients, which Adam's 1/sqrt(v) amplifies
    enc = O.GNNRef("SAGE", H, H, H, 2, 0.0).double()
    pred = (O.DotPredictorRef() if predictor == "DOT" else O.MLPPredictorRef(H, H, 1, 2, 0.0)).double()
    m = P.BaseModel(lr=0.01, dropout=0.0, grad_clip_norm=1.0, gnn_num_layers=2, mlp_num_layers=2,
                    emb_hidden_channels=H, gnn_hidden_channels=H, mlp_hidden_channels=H, num_nodes=N,
                    num_node_feats=0, gnn_encoder_name="SAGE", predictor_name=predictor, loss_func=loss_name,
                    optimizer_name="Adam", device="cpu", use_node_feats=False, train_node_emb=True,
                    modules=(enc, pred, lambda p_, n_, k_, w_: O.LOSSES[O.select_loss(loss_name, w_ is not None)](
                        p_, n_, k_, w_)),
                    process_group=pg, dp_scaling=scaling, dp_exchange=exchange)
    if world > 1:
        want = "grads" if (predictor == "MLP" or exchange == "grads") else "scores"
        assert m.dp_mode() == want, (m.dp_mode(), want)
    m.emb.double()
    m.param_init()
    data = _Data()
    data.adj_t = adj
    data.edge_index = torch.stack([torch.cat([pos[:, 1], pos[:, 0]]), torch.cat([pos[:, 0], pos[:, 1]])])
    split = {"train": {"edge": pos, "weight": w.double()}}
    torch.manual_seed(200)
    losses = [m.train(data, split, batch, "local", K) for _ in range(EPOCHS)]
    flat = torch.cat([p.detach().reshape(-1) for p in m.para_list]).double().numpy()
    if out_q is not None:
        out_q.put((rank, losses, flat))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    else:
        torch.set_num_threads(os.cpu_count() or 1)      # do not leak the setting into other tests
    return losses, flat


def _spawn(world, batch, scaling, loss_name, predictor, exchange="auto"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, batch, scaling, loss_name, predictor, q, exchange))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("scaling,loss_name,predictor,exchange",
                         [("strong", "AUC", "MLP", "auto"), ("weak", "WeightedHingeAUC", "DOT", "grads"),
                          ("weak", "WeightedHingeAUC", "DOT", "scores"), ("strong", "HingeAUC", "DOT", "auto"),
                          # batch-AVERAGED losses (loss.py:45-62): the slice mean is weighted by n_r / n
                          ("strong", "LogRank", "DOT", "grads"), ("weak", "LogRank", "DOT", "scores"),
                          ("strong", "CE", "MLP", "auto")])
def test_two_ranks_equal_one_process(scaling, loss_name, predictor, exchange):
    """both exchange modes: parameter-gradient all-reduce, and all-gather of the per-edge score
    gradients followed by the same global backward on every rank"""
    B = 64
    single_batch = B if scaling == "strong" else 2 * B       # weak: global batch = world * B
    ref_losses, ref_flat = _run(0, 1, 0, single_batch, scaling, loss_name, predictor, None)
    res = _spawn(2, B, scaling, loss_name, predictor, exchange)
    for rank, losses, flat in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9, err_msg=f"rank {rank}")
        np.testing.assert_allclose(flat, ref_flat, rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(res[0][2], res[1][2])        # bit-identical replicas


@pytest.mark.parametrize("exchange", ["grads", "scores"])
def test_uneven_last_batch_and_empty_slice(exchange):
    """a last global batch smaller than the world size leaves a rank with no edges;
    it must still join the exchange"""
    adj, pos, w = _problem()
    n = pos.size(0)
    B = n - 1                      # second global batch has exactly 1 edge -> rank 1 gets none
    ref_losses, ref_flat = _run(0, 1, 0, B, "strong", "AUC", "DOT", None)
    res = _spawn(2, B, "strong", "AUC", "DOT", exchange)
    for rank, losses, flat in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9)
    np.testing.assert_array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("world,exchange", [(4, "grads"), (8, "grads"), (4, "scores")])
def test_four_and_eight_ranks_equal_one_process(world, exchange):
    """the node the path is meant for has 8 GPUs: the same equivalence at world 4 and 8 (N = 120 nodes, so row / edge
    slices are uneven), weak scaling -- W ranks each bringing B positives == one process on W x B -- and a last global
    batch that leaves most ranks without an edge"""
    B = 16
    ref_losses, ref_flat = _run(0, 1, 0, world * B, "weak", "WeightedHingeAUC", "DOT", None)
    res = _spawn(world, B, "weak", "WeightedHingeAUC", "DOT", exchange)
    assert len(res) == world
    for rank, losses, flat in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9, err_msg=f"rank {rank}")
        np.testing.assert_allclose(flat, ref_flat, rtol=1e-6, atol=1e-9)
    for r in res[1:]:
        np.testing.assert_array_equal(res[0][2], r[2])            # bit-identical replicas
    n = _problem()[1].size(0)
    ref_losses, _ = _run(0, 1, 0, n - 1, "strong", "AUC", "DOT", None)        # second global batch: ONE edge
    res = _spawn(world, n - 1, "strong", "AUC", "DOT", exchange)
    for rank, losses, flat in res:
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-9)


def _tuner_rank(rank, world, port, out_q):
    """the aggregation autotuner under a process group: the op never communicates; the ranks agree only in
    ops.tune_aggregation (a rank-0-only aggregation used to broadcast into its idle peers)"""
    import plnlp_amd as P
    from plnlp_amd import ops
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    adj, _, _ = _problem()
    g = P.Graph(adj.rowptr.clone(), adj.col.to(torch.int32), None, adj.n_rows, adj.n_cols)
    x = torch.zeros(g.n_cols, 256)
    out = torch.zeros(g.n_rows, 256)
    alone = None
    if rank == 1:
        # ONE rank consults the tuner from inside the op while its peer does something else: it must return the
        # default form at once, without a collective and without remembering it
        alone = ops._agg_tune(g, x, out, "mean", False, None, None, None, 256)
        assert alone == 0 and 256 not in g._agg_tune
    # all ranks, explicitly: every rank "measures" a different winner, rank 0's is adopted everywhere
    picked = ops.tune_aggregation(g, [256, 64, 512], group=dist.group.WORLD,
                                  time_fn=lambda graph, feat: (16 if feat == 256 else 32) if rank == 0 else 0)
    after = ops._agg_tune(g, x, out, "mean", False, None, None, None, 256)
    # the transposed views share the choice (the backward passes cannot be timed themselves)
    shared = g.t()._agg_tune is g._agg_tune
    out_q.put((rank, alone, picked, after, shared))
    dist.barrier()
    dist.destroy_process_group()


def test_aggregation_tuner_never_communicates_inside_the_op():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tuner_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, alone, picked, after, shared in res:
        assert picked == {256: 16, 512: 32}, (rank, picked)       # width 64 is below the tuned range
        assert after == 16 and shared
    assert res[1][1] == 0
