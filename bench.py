"""bench.py -- pos+neg edges scored / second on the collab-shaped workload
(BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without WORLD_SIZE: starts its own N ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the training hot path (plnlp/model.py:148-167) over one batch of synthetic input:
SAGE encoder forward+backward, fused gather+DOT scoring of B positives and B*k negatives, WeightedHingeAUC
loss, per-group clipping and Adam (N > 1: plus the exchange `config.dp_exchange` names).  Loss, every
gradient and the update are those of the reference's step -- but the work is NOT the reference's launch list:
with `config.sparse_forward` the 1-layer encoder is evaluated only at the rows the batch's edges read
(`config.touched_fraction` of the nodes; the reference computes all N rows, model.py:150-151, and reads only
those, model.py:155-156) and its backward runs on those rows; outputs are bit-identical to the full-matrix
step (tests/test_hip_round2.py) and the full-matrix step is timed beside it (`ms_per_step_full_forward`), as
is the step on the exact-f32 MFMA form (`ms_per_step_f32_mfma`).  Inputs are resident in HBM before the timed
region.  Weak scaling: every rank scores its own B positives (+B*k negatives) per step.

Rank 0 prints ONE JSON line; see the task contract for the fields.  Beyond the contract:
`train_epoch` = edges/s through BaseModel.train for one epoch of the workload (sampler and permutation times
separate), `eval_scoring` = edges/s of BaseModel.test.  `roofline` is measured live with device events on the
launch stream for the aggregation kernel on the graph that does not fit the caches (its `subject` says so);
`roofline_workload_agg`, `roofline_mfma` and `roofline_agg_adam` time the launches the step really makes
(touched rows, gathered operands, the Adam epilogue).  `cpu_baseline` times the CPU oracle (a port of the
reference's PyG path, which cannot run here) on the host cores, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

WORKLOADS = {
    # README.md:35 recipe shape
    "collab": dict(shape="collab", encoder="SAGE", predictor="DOT", loss="WeightedHingeAUC", hidden=256,
                   gnn_layers=1, mlp_layers=2, num_neg=1, dropout=0.3, clip=1.0, batch=65536, weighted=True),
    # README.md:40 recipe shape (config C4): GCN on [emb 50 | features 128], h=200, local negatives
    "citation2": dict(shape="citation2", encoder="GCN", predictor="MLP", loss="AUC", hidden=200, emb=50,
                      feats=128, gnn_layers=2, mlp_layers=2, num_neg=3, dropout=0.0, clip=1.0, batch=65536,
                      weighted=False),
    # README.md:24 recipe shape (config C2)
    "ddi": dict(shape="ddi", encoder="SAGE", predictor="MLP", loss="AUC", hidden=512, gnn_layers=2,
                mlp_layers=2, num_neg=3, dropout=0.3, clip=2.0, batch=65536, weighted=False),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="collab", choices=sorted(WORKLOADS) + ["rmat"],
                    help="rmat = BASELINE config 5, forward-only row-sharded stress (use --scale <= 0.5 on one GPU)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; invalidates the number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the >>256 MiB HBM roofline measurement")
    ap.add_argument("--cpu-steps", type=int, default=6,
                    help="timed steps of the CPU oracle (collab: ~1.8 s each on 16 CPUs -> ~13 s with the warm-up step)")
    ap.add_argument("--epoch-steps", type=int, default=400,
                    help="cap of the train_epoch measurement in batches (collab's random-walk epoch is ~350 steps: whole)")
    ap.add_argument("--no-parity", action="store_true", help="skip the Hits@50 GPU-vs-oracle training parity run")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even with one rank (exercises the DP path on 1 GPU)")
    ap.add_argument("--dp-exchange", default="default", choices=["default", "auto", "grads", "scores", "shard"],
                    help="what the ranks exchange per step (BaseModel docstring): parameter gradients, score "
                         "gradients, or -- shard -- activations of a row-sharded encoder.  default: shard for the "
                         "SAGE-on-embedding workloads (collab, ddi), grads otherwise")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling and control measurements (N > 1)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="FUNCTIONAL run of the N-rank path on ONE GPU: every rank on cuda:0, gloo instead of RCCL (which refuses "
                         "two ranks on one device).  Executes everything an N-GPU run executes -- the launcher, the ranks' agreement "
                         "on the exchange form, the data-parallel steps, the phase timers -- but its numbers are N processes "
                         "time-slicing one GPU: the line says so (`shared_gpu`) and is not a scaling point")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher check without GPUs: the ranks rendezvous over gloo, do one all-reduce and rank 0 "
                         "prints a JSON line (no kernels run; not a measurement)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--as-rank", default="",
                    help="rmat only, one process: 'R/W' = do what rank R of a W-rank job does at --scale 1.0 -- its "
                         "destination-row block of the 50 M x 1 B R-MAT graph gathering from the full replicated "
                         "50 M x 512 source (102 GB): layer 1's aggregation and GEMM on its 6.25 M rows.  No "
                         "collective runs (the other ranks do not exist); reports the aggregation kernel on the "
                         "source size config 5 really has")
    ap.add_argument("--batch-mult", type=int, default=1,
                    help="multiply the per-GPU batch (debug: the per-rank cost of an N-rank 'scores' job is about the "
                         "1-GPU step at N times the batch; invalidates the number)")
    return ap.parse_args()


def launch_ranks(n_ranks, deadline_s=None):
    """`python bench.py --gpus N` outside a launcher: start N rank processes of this same script (one
    per GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line, return the worst exit code.  Runs before
    anything touches the GPU in this process, and starts CHILDREN -- a process that has initialised the
    GPU must never exec another program on this pool.
    The children are POLLED: the first one that exits non-zero (import error, out of memory, a kernel
    fault) takes its siblings down with it -- they would otherwise sit in the rendezvous or in a collective
    until the 10-minute RCCL timeout -- and an overall deadline bounds the whole job.  Every rank's stderr
    is kept in its own file (gpurun_out/ranks/ when that directory can be made, else a temp dir) and the
    failing rank's tail is relayed."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    deadline_s = float(os.environ.get("PLNLP_BENCH_DEADLINE_S", "1500")) if deadline_s is None else deadline_s
    log_dir = os.path.join(ROOT, "gpurun_out", "ranks")
    try:
        os.makedirs(log_dir, exist_ok=True)
    except OSError:
        log_dir = tempfile.mkdtemp(prefix="plnlp_ranks_")
    procs, logs = [], []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        err = open(os.path.join(log_dir, "rank%d.stderr" % r), "w")
        out = open(os.path.join(log_dir, "rank%d.stdout" % r), "w")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out, stderr=err))
    t0 = time.time()
    rc, failed = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0], abs(codes[bad[0]]) or 1
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > deadline_s:
            failed, rc = -1, 124
            break
        time.sleep(0.2)
    if failed is not None:                      # end exactly the children started here (terminate, then kill)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        while any(p.poll() is None for p in procs) and time.time() - t1 < 10:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    for out, err in logs:
        out.close()
        err.close()

    def tail(path, n=30):
        try:
            return "".join(open(path).readlines()[-n:])
        except OSError:
            return ""
    if failed is not None:
        which = "deadline of %.0f s exceeded" % deadline_s if failed < 0 else "rank %d exited with code %d" % (failed, rc)
        print("bench.py --gpus %d: %s; the other ranks were stopped.  Logs: %s" % (n_ranks, which, log_dir), file=sys.stderr)
        for r in range(n_ranks):
            if failed < 0 or r == failed:
                print("---- rank %d stderr (tail) ----\n%s" % (r, tail(os.path.join(log_dir, "rank%d.stderr" % r))),
                      file=sys.stderr)
        return rc
    line = None
    for ln in open(os.path.join(log_dir, "rank0.stdout")).read().splitlines():
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln
        else:
            print(ln, file=sys.stderr)
    sys.stderr.write(tail(os.path.join(log_dir, "rank0.stderr"), 15))
    if line is not None:
        print(line, flush=True)
    return 0 if line is not None else 1


def comm_info(pg):
    """{"backend", "rccl_ranks"}: rccl_ranks counts the ranks of an RCCL ("nccl") process group ONLY -- 0 under gloo (the
    --share-gpu runs and the CPU dry run), so that "did RCCL see N ranks?" cannot be answered yes by a gloo job"""
    if pg is None:                       # one process, no process group: nothing of RCCL is running
        return {"backend": "none", "rccl_ranks": 0}
    backend = str(torch.distributed.get_backend(pg))
    return {"backend": backend, "rccl_ranks": torch.distributed.get_world_size(pg) if backend == "nccl" else 0}


def xgmi_topology():
    """the link matrix `rocm-smi --showtopo` prints (link type and hops between every GPU pair), parsed, or "unavailable".
    Run as a CHILD process with a timeout (a process that has initialised the GPU must not exec another program)."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return "unavailable"
    try:
        out = subprocess.run([exe, "--showtopo"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return "unavailable"
    tables, name, rows = {}, None, []
    for ln in out.splitlines():
        t = ln.strip()
        if t.startswith("=") and "Link Type" in t:
            name = "link_type"
        elif t.startswith("=") and "Hops" in t:
            name = "hops"
        elif t.startswith("=") and "Weight" in t:
            name = "weight"
        elif t.startswith("="):
            name = None
        elif name and t.startswith("GPU") and len(t.split()) > 1 and not t.split()[1].startswith("GPU"):
            tables.setdefault(name, []).append(t.split()[1:])
    return tables if tables else "unavailable"


def rank_devices(pg, rank, local_rank, device):
    """what every rank runs on, gathered into the line: device index, name, the visible-device lists of its environment"""
    props = torch.cuda.get_device_properties(device)
    mine = {"rank": rank, "local_rank": local_rank, "device": str(device), "name": props.name,
            "total_memory_GiB": round(props.total_memory / 2 ** 30, 1), "devices_visible": torch.cuda.device_count(),
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES")}
    if pg is None or torch.distributed.get_world_size(pg) == 1:
        return [mine]
    out = [None] * torch.distributed.get_world_size(pg)
    torch.distributed.all_gather_object(out, mine, group=pg)
    return out


def dry_run_cpu(world, rank, fail_rank=-1):
    """the launcher path without GPUs (gloo): proves that N ranks start, meet and pass the start-up collective
    self-test the real run performs (plnlp_amd.shard.collective_self_test).  fail_rank: that rank exits with an
    error before the rendezvous (launcher test: the siblings must be stopped, not left waiting)."""
    if rank == fail_rank:
        sys.exit("dry run: rank %d fails on purpose" % rank)
    from plnlp_amd import shard
    torch.distributed.init_process_group("gloo")
    passed = shard.collective_self_test(torch.distributed.group.WORLD, torch.device("cpu"))
    ranks = torch.distributed.get_world_size()
    print("rank %d: collective self-test ok: %s" % (rank, ", ".join(sorted(passed))), file=sys.stderr, flush=True)
    torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "backend": "gloo", "rccl_ranks": 0, "ranks": ranks, "n_gpus": world,
                          "all_reduce_ok": True, "collective_self_test": sorted(passed),
                          "note": "launcher check only; no kernel ran, not a measurement"}),
              flush=True)
    return 0


def run_rmat_stress(args, P, world, rank, device, pg):
    """BASELINE.json config 5 -- R-MAT 50 M nodes / 1 B edges, SAGE x2 h = 512, FORWARD ONLY (a training
    replica of it does not exist: X alone is 102 GB).  Row-sharded: rank r builds and keeps
    only its destination-row block of the graph, X is replicated, layer 1 produces the rank's rows, ONE
    all-gather (into X's own storage) makes them the next layer's sources, layer 2 produces the rank's
    output rows.  --scale shrinks nodes and edges alike (one GPU holds scale <= 0.5)."""
    from plnlp_amd import synthetic, shard, _lib
    F = 512
    n = max(1024, int(round(50_000_000 * args.scale)))
    nnz = max(4096, int(round(1_000_000_000 * args.scale)))
    rscale = max(10, (n - 1).bit_length())
    part = shard.RowPartition(n, world, rank)
    S, npad = part.rows, part.padded
    t0 = time.perf_counter()
    blk = synthetic.rmat_row_block(rscale, nnz, n, part.lo, S, npad, device, seed=11)
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    gen = torch.Generator(device=device).manual_seed(12)
    x = torch.empty(npad, F, device=device)
    step_rows = 1 << 20
    for lo in range(0, npad, step_rows):                     # same stream on every rank: replicated X
        x[lo:lo + step_rows].normal_(generator=gen)
    ws = [torch.randn(F, F, device=device, generator=gen) * 0.03 for _ in range(4)]
    bs = [torch.zeros(F, device=device) for _ in range(2)]
    enc = shard.RowShardedSAGEForward(blk, part, [(ws[0], bs[0], ws[1]), (ws[2], bs[1], ws[3])], group=pg)
    agg = enc.agg
    if pg is not None:          # the ranks agree on the kernel form once, collectively (never inside the op)
        P.ops.tune_aggregation(blk, [F], group=pg)
    y2 = None

    def forward():
        nonlocal y2
        y2 = enc.forward(x)

    def sync():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    K, W = args.steps, args.warmup
    for _ in range(W):
        forward()
    sync()
    t0 = time.perf_counter()
    for _ in range(K):
        forward()
    sync()
    dt = time.perf_counter() - t0
    if pg is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    t_agg = time_kernel(lambda: P.ops.csr_aggregate(blk, x, "mean", False, out=agg), iters=5, warm=1)
    by = agg_bytes(blk.nnz, S, F)
    checksum = float(y2.double().sum().item())
    total_nnz = torch.tensor([blk.nnz], dtype=torch.float64, device=device)
    if pg is not None:
        torch.distributed.all_reduce(total_nnz)
    result = {
        "metric": "edges aggregated/sec, R-MAT forward stress (2 SAGE layers)", "value": 2 * float(total_nnz.item()) * K / dt,
        "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "R-MAT (.57,.19,.19,.05) N=%d, nnz=%d, SAGE x2 h=512 forward only, X replicated, "
                               "destination rows sharded over %d rank(s)" % (n, int(total_nnz.item()), world),
                   "scale": args.scale, "rows_per_rank": S, "parallelism": "row-sharded x%d, one all-gather between layers" % world},
        "roofline": {"bound": "hbm", "kernel": "csr_agg_vec_kernel (mean, F=512) on this rank's row block",
                     "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None,
                     "effective_GBps": by / t_agg / 1e9,
                     "traffic": None, "algorithmic_bytes": by, "kernel_ms": t_agg * 1e3,
                     "source_MiB": npad * F * 4 / 2 ** 20,
                     "note": "SKEWED graph: the hub rows of R-MAT are re-read from the caches, so the gather-model byte "
                             "count overstates what crosses the HBM pins and no roofline fraction is claimed from it "
                             "(frac = null); effective_GBps = gather-model bytes / time.  The no-reuse HBM fraction of "
                             "this kernel is the uniform graph of the default workload's `roofline`; measured HBM bytes "
                             "of this launch: scripts/pmc_agg.sh -> profiles/"},
        "graph_build_s": t_graph, "output_checksum": checksum,
    }
    result.update(comm_info(pg))
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


def run_rmat_as_rank(args, P, device):
    """BASELINE.json config 5 at ONE RANK'S TRUE SHARE: rank R of W builds its destination-row block of the full
    R-MAT graph (N = 50 M, 1 B edges) and gathers from the full replicated source X [50 M, 512] (102 GB, generated in
    place) -- the geometry the aggregation kernel meets on an 8-GPU node, which no --scale run reaches (a source 18x
    the one behind the default `roofline`, offsets beyond 2^32 bytes, TLB reach).  Layer 1 on the rank's rows:
    mean aggregation, then the concat-K GEMM with bias + relu.  One process, no collective."""
    from plnlp_amd import synthetic, shard
    R, Wd = (int(v) for v in args.as_rank.split("/"))
    F = 512
    n = max(1024, int(round(50_000_000 * args.scale)))
    nnz = max(4096, int(round(1_000_000_000 * args.scale)))
    rscale = max(10, (n - 1).bit_length())
    part = shard.RowPartition(n, Wd, R)
    S, npad = part.rows, part.padded
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    blk = synthetic.rmat_row_block(rscale, nnz, n, part.lo, S, npad, device, seed=11)
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    gen = torch.Generator(device=device).manual_seed(12)
    x = torch.empty(npad, F, device=device)
    for lo in range(0, npad, 1 << 20):
        x[lo:lo + (1 << 20)].normal_(generator=gen)
    w_l = torch.randn(F, F, device=device, generator=gen) * 0.03
    w_r = torch.randn(F, F, device=device, generator=gen) * 0.03
    bias = torch.zeros(F, device=device)
    agg = torch.empty(S, F, device=device)
    y = torch.empty(S, F, device=device)
    P.ops.tune_aggregation(blk, [F])
    epi = P._lib.make_epilogue(bias=bias, relu=True)
    K, W = args.steps, args.warmup

    def layer():
        P.ops.csr_aggregate(blk, x, "mean", False, out=agg)
        P.ops.gemm([(agg, w_l), (x[part.lo:part.lo + S], w_r)], False, True, out=y, epilogue=epi)
    for _ in range(W):
        layer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        layer()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    t_agg = time_kernel(lambda: P.ops.csr_aggregate(blk, x, "mean", False, out=agg), iters=max(3, K), warm=1)
    t_gemm = time_kernel(lambda: P.ops.gemm([(agg, w_l), (x[part.lo:part.lo + S], w_r)], False, True, out=y, epilogue=epi),
                         iters=max(3, K), warm=1)
    by = agg_bytes(blk.nnz, S, F)
    # what MUST cross the HBM pins for this launch: every index once, every DISTINCT source row once, the result once
    distinct = int(torch.unique(blk.col).numel())
    compulsory = blk.nnz * 4 + (S + 1) * 8 + (distinct + S) * 4 * F
    traffic = None
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):
        traffic = json.load(open(prof)).get("rmat_as_rank_%d_of_%d" % (R, Wd))
    result = {
        "metric": "edges aggregated/sec, R-MAT config 5, ONE rank's share of layer 1", "value": blk.nnz / dt,
        "unit": "edges/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt * 1e3, "higher_is_better": True,
        "scaling": "unmeasured (one process doing rank %d of %d's work; no collective)" % (R, Wd), "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "R-MAT (.57,.19,.19,.05) N=%d, 1e9 x %g edges; rank %d of %d: rows [%d, %d), %d entries, "
                               "source X [%d, %d] = %.1f GB" % (n, args.scale, R, Wd, part.lo, part.lo + S, blk.nnz, npad, F,
                                                               npad * F * 4 / 1e9),
                   "scale": args.scale, "rows_per_rank": S, "as_rank": args.as_rank},
        "roofline": {"bound": "hbm", "kernel": "csr aggregation (mean, F = 512), form: " + P.ops.describe_form(
                         getattr(blk, "_agg_tune", {}).get(F, 0)) if hasattr(P.ops, "describe_form") else "csr aggregation",
                     "subject": "this rank's row block gathering from the full 102 GB source",
                     "kernel_ms": t_agg * 1e3, "peak": 8000.0, "unit": "GB/s",
                     "algorithmic_bytes_gather_model": by, "effective_GBps": by / t_agg / 1e9,
                     "compulsory_bytes": compulsory, "distinct_source_rows": distinct,
                     "achieved": compulsory / t_agg / 1e9, "frac": compulsory / t_agg / 8e12,
                     "traffic": None, "traffic_from_profile": traffic,
                     "fabric_side_rate_over_8TBps": (traffic["bytes"] / t_agg / 8e12) if traffic else None,
                     "note": "frac = COMPULSORY bytes (indices + every distinct source row once + the result) / time / 8 TB/s "
                             "-- a lower bound on the kernel's HBM efficiency: rows re-read after falling out of the caches "
                             "cross the pins again.  fabric_side_rate_over_8TBps = the PMC-measured bytes of the same launch "
                             "(scripts/pmc_agg.sh, profiles/traffic.json) / time / 8 TB/s: those counters sit on the FABRIC side "
                             "and include Infinity-Cache hits (R-MAT's hub rows live there), so this is NOT an HBM fraction -- "
                             "it can exceed the ~0.79 any kernel pulls from the HBM pins.  The kernel's HBM efficiency on this "
                             "graph lies between frac (compulsory) and fabric-bound; the no-reuse HBM fraction is the default "
                             "workload's `roofline` (uniform graph)"},
        "gemm_layer1_ms": t_gemm * 1e3, "gemm_TFLOPs_f32_equivalent": 2.0 * S * F * 2 * F / t_gemm / 1e12,
        "graph_build_s": t_graph, "output_checksum": float(y.double().sum().item()), "backend": "none", "rccl_ranks": 0,
    }
    print(json.dumps(result), flush=True)


def agg_bytes(nnz, n_out, feat, weighted=False):
    """SURVEY.md 8d gather model: nnz*(4F+4) + N_out*4F + (N_out+1)*4 (+ nnz*4 if weighted)"""
    return nnz * (4 * feat + 4) + n_out * 4 * feat + (n_out + 1) * 4 + (nnz * 4 if weighted else 0)


def time_kernel(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def agg_bytes_compulsory(nnz, n_src, n_out, feat, weighted=False):
    """every index once, the source matrix once, the output once"""
    return nnz * 4 + (n_out + 1) * 4 + (n_src + n_out) * 4 * feat + (nnz * 4 if weighted else 0)


def measure_roofline(P, graph, feat, device, weighted=False, shape="collab"):
    """the aggregation kernel on `graph`, timed with device events on the launch stream.
    Source matrix >> 256 MiB (the Infinity Cache): HBM-bound, `achieved` = gather-model bytes / t against
    the 8 TB/s peak.  Source <= 256 MiB: every gathered row is a cache hit after the first touch, the
    kernel is bound by the cache hierarchy, not by HBM -- `bound` says "cache", `achieved` / `frac` are
    taken on the COMPULSORY bytes (what has to cross the HBM pins), and the gather-model rate is
    reported as `effective_GBps` (it may exceed the HBM peak and is not a roofline fraction)."""
    x = torch.randn(graph.n_cols, feat, device=device)
    out = torch.empty(graph.n_rows, feat, device=device)
    if weighted:
        t = time_kernel(lambda: P.ops.csr_aggregate(graph, x, "sum", True, out=out))
    else:
        t = time_kernel(lambda: P.ops.csr_aggregate(graph, x, "mean", False, out=out))
    by = agg_bytes(graph.nnz, graph.n_rows, feat, weighted)
    by_min = agg_bytes_compulsory(graph.nnz, graph.n_cols, graph.n_rows, feat, weighted)
    src_mib = graph.n_cols * feat * 4 / 2 ** 20
    cached = src_mib <= 256
    from_profile = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            v = json.load(open(tpath)).get(f"csr_agg_{shape}_f{feat}")
            if v is not None:
                from_profile = {"bytes": v, "file": "profiles/traffic.json",
                                "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/bench_agg.py (an "
                                       "earlier run on the same shape; NOT measured in this run)"}
        except Exception:
            from_profile = None
    alg = by_min if cached else by
    skewed = not cached and not shape.startswith("uniform")
    if skewed:
        # hub rows are re-read from the caches: the gather model overstates what crosses the HBM pins (it gave
        # fractions above 1 on R-MAT), so no roofline fraction is claimed from it
        return {"bound": "hbm", "kernel": "csr_agg_vec_kernel (%s, F=%d)" % ("weighted sum" if weighted else "mean", feat),
                "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                "traffic_from_profile": from_profile, "algorithmic_bytes": by, "bytes_model": "gather",
                "compulsory_bytes": by_min, "effective_GBps": by / t / 1e9, "compulsory_GBps": by_min / t / 1e9,
                "kernel_ms": t * 1e3, "source_MiB": src_mib,
                "note": "source exceeds the 256 MiB Infinity Cache but the graph is SKEWED: hub rows are re-read from the "
                        "caches, so gather-model bytes / time is an effective rate, not an HBM fraction (frac = null); the "
                        "no-reuse fraction of this kernel is taken on the uniform graph (default workload's `roofline`)"}
    return {"bound": "cache" if cached else "hbm",
            "kernel": "csr_agg_vec_kernel (%s, F=%d)" % ("weighted sum" if weighted else "mean", feat),
            "launches": ("one aggregation = csr_agg_fused_kernel (one wave per short row + the long rows' chunks, in one "
                         "launch) + csr_agg_finalize_kernel (the long rows' partial sums); kernel_ms and achieved cover both")
                        if graph.row_split(P.ops.split_threshold(graph.n_cols)).active else
                        "one aggregation = one launch of csr_agg_vec_kernel (no row of this graph is longer than the split "
                        "threshold)",
            "achieved": alg / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / t / 8.0e12,
            "traffic": None, "traffic_from_profile": from_profile,
            "algorithmic_bytes": alg, "bytes_model": "compulsory" if cached else "gather",
            "gather_model_bytes": by, "compulsory_bytes": by_min, "effective_GBps": by / t / 1e9,
            "kernel_ms": t * 1e3, "source_MiB": src_mib,
            "kernel_form": P.ops.describe_form(getattr(graph, "_agg_tune", {}).get(feat, 0)),
            "note": ("source fits the 256 MiB Infinity Cache: cache-bound, frac is compulsory bytes over the HBM "
                     "peak; effective_GBps is the gather-model rate (not a roofline fraction)") if cached else
                    ("source exceeds the 256 MiB Infinity Cache: HBM-bound, gather-model bytes" +
                     ("" if shape.startswith("uniform") else
                      "; the graph is SKEWED (hub rows are re-read from the caches), so this is an effective rate, not a "
                      "no-reuse HBM fraction -- that one is taken on the uniform graph (default workload's `roofline`)"))}


_X3_KERNELS = {
    "gemm_x3b": "x3b::gemm_x3b_kernel (stationary pre-split weights, a whole 256-row block per workgroup, one persistent workgroup per CU)",
    "gemm_x3s": "x3s::gemm_x3s_kernel (stationary pre-split weights, 128-row panels)",
    "gemm_tile_x3": "x16::gemm_f32_kernel (128 x 128 tiles, both operands split in the loaders)",
}


def _x3_kernel_name(P, launch):
    """which split-bf16 kernel `launch` runs: asked of the library's launch counters (plnlp_launch_counts), not re-derived here"""
    mode0 = P.ops.GEMM_MATH["mode"]
    try:
        P.ops.GEMM_MATH["mode"] = "bf16x3"
        c0 = P.ops.launch_counts()
        launch()
        torch.cuda.synchronize()
        c1 = P.ops.launch_counts()
    finally:
        P.ops.GEMM_MATH["mode"] = mode0
    ran = [k for k in _X3_KERNELS if c1.get(k, 0) > c0.get(k, 0)]
    return " + ".join(_X3_KERNELS[k] for k in ran) if ran else "x16::gemm_f32_kernel"


def _gemm_forms(P, flop, kname, m, n, launch):
    """time `launch` with the dense products formed both ways; -> the roofline_mfma object of the path's own form"""
    mode0 = P.ops.GEMM_MATH["mode"]
    times = {}
    try:
        for mode in ("f32", "bf16x3"):
            P.ops.GEMM_MATH["mode"] = mode
            times[mode] = time_kernel(launch)
    finally:
        P.ops.GEMM_MATH["mode"] = mode0
    f32_form = {"math": "f32", "kernel": "g16::gemm_f32_kernel (%s)" % kname, "achieved": flop / times["f32"] / 1e12,
                "peak": 157.3, "unit": "TFLOP/s", "frac": flop / times["f32"] / 157.3e12, "kernel_ms": times["f32"] * 1e3}
    tx = times["bf16x3"]
    x3_form = {"math": "bf16x3", "kernel": "%s (%s)" % (_x3_kernel_name(P, launch), kname), "achieved": 6 * flop / tx / 1e12,
               "peak": 2500.0, "unit": "TFLOP/s", "frac": 6 * flop / tx / 2.5e15, "kernel_ms": tx * 1e3,
               "f32_equivalent_TFLOPs": flop / tx / 1e12,
               "note": "fp32 in / fp32 out; operands split into three bf16 terms, six bf16 MFMAs per block (executed flops = "
                       "6 x algorithmic) against the dense bf16 peak"}
    mine, other = (x3_form, f32_form) if mode0 == "bf16x3" else (f32_form, x3_form)
    return dict({"bound": "mfma", "traffic": None, "flops": flop}, **mine, other_form=other)


def measure_step_launches_ddi(P, model, data, cfg, device, rows_scored):
    """The ddi step's dominant launches AS THE STEP MAKES THEM (SAGE x2 + MLP, the batch touches every node, so the step is dense):
      aggregation   csr_aggregate(adj, table, mean) over all 4 267 rows, F = 512 -- four such launches per step (two layers,
                    forward and transposed); the 8.7 MB source is cache-resident: `bound: cache`, frac on the COMPULSORY bytes
      scorer GEMM   relu(x W1^T + b1) with dropout and the 1-output head in the epilogue (PLNLP_EPI_ROWDOT), M = B (1 + k) rows
    each timed live with device events on the launch stream."""
    from plnlp_amd import _lib
    ops = P.ops
    adj = data.adj_t
    x = model.emb.weight.detach()
    n, F = x.shape
    out = torch.empty(n, F, device=device)
    t = time_kernel(lambda: ops.csr_aggregate(adj, x, "mean", False, out=out))
    by = agg_bytes(adj.nnz, n, F, False)
    by_min = agg_bytes_compulsory(adj.nnz, n, n, F, False)
    c0 = ops.launch_counts()
    ops.csr_aggregate(adj, x, "mean", False, out=out)
    dense = ops.launch_counts()["agg_dense"] > c0["agg_dense"]
    if dense:
        # the graph is dense enough for the matrix cores (csrc/aggregate_dense.hip): bf16 counts x three-term split = 3 MFMAs per block
        ops.DENSE_AGG["enabled"] = False
        try:
            t_csr = time_kernel(lambda: ops.csr_aggregate(adj, x, "mean", False, out=out))
        finally:
            ops.DENSE_AGG["enabled"] = True
        kp = (n + 15) // 16 * 16
        flop = 2.0 * n * kp * F
        res = {"roofline_workload_agg": {
            "bound": "mfma", "subject": "the step's own aggregation launch (all %d rows, F = %d: the batch touches every node), as a "
                                        "product on the matrix cores -- the graph holds %.1f %% of all node pairs" % (n, F, 100.0 * adj.nnz / n / n),
            "kernel": "aggd::split_x_kernel + aggd::dense_agg_kernel + aggd::dense_reduce_kernel (bf16 counts [%d, %d] x x in three bf16 terms)" % (n, kp),
            "achieved": 3 * flop / t / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": 3 * flop / t / 2.5e15, "traffic": None,
            "flops": flop, "executed_flops": 3 * flop, "kernel_ms": t * 1e3, "compulsory_bytes": by_min,
            "csr_kernels": {"kernel_ms": t_csr * 1e3, "gather_model_bytes": by, "effective_GBps": by / t_csr / 1e9,
                            "form": ops.describe_form(getattr(adj, "_agg_tune", {}).get(F, 0))},
            "note": "a SMALL product for the chip (%.0f GFLOP executed = %.0f us of matrix-pipe time at the peak): the launch is bound by its "
                    "latency chain (K cut into slices, ~2 workgroups per CU), not by the pipe; the CSR kernels gather %.1f GB out of L2 for "
                    "the same result in %.2f x the time (csr_kernels, measured here)" % (3 * flop / 1e9, 3 * flop / 2.5e15 * 1e6, by / 1e9, t_csr / t)}}
    else:
        res = None
    res = res or {"roofline_workload_agg": {
        "bound": "cache", "subject": "the step's own aggregation launch (all %d rows, F = %d: the batch touches every node)" % (n, F),
        "kernel": "csr_agg_vec_kernel / csr_agg_chunk_kernel (mean, F=%d)" % F,
        "kernel_form": ops.describe_form(getattr(adj, "_agg_tune", {}).get(F, 0)),
        "achieved": by_min / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by_min / t / 8.0e12, "traffic": None,
        "algorithmic_bytes": by_min, "bytes_model": "compulsory", "gather_model_bytes": by, "effective_GBps": by / t / 1e9,
        "kernel_ms": t * 1e3, "source_MiB": n * F * 4 / 2 ** 20,
        "note": "the %.1f MiB source is L2 / Infinity-Cache resident: the launch is bound by the caches (effective_GBps = the "
                "gather-model rate out of L2, ~34 TB/s aggregate), frac = compulsory bytes over the HBM peak" % (n * F * 4 / 2 ** 20)}}
    h = cfg["hidden"]
    lin0, lin1 = model.predictor.lins[0], model.predictor.lins[1]
    xs = torch.randn(rows_scored, h, device=device)
    hid = torch.empty(rows_scored, h, device=device)

    def launch():
        epi = _lib.make_epilogue(bias=lin0.bias, relu=True, dropout_p=cfg["dropout"], dropout_seed=1)
        ops.gemm([(xs, lin0.weight)], False, True, out=hid, epilogue=epi, rowdot=(lin1.weight, lin1.bias))
    kname = "M=%d (scored pairs), N=%d, K=%d, bias+relu+dropout + the 1-output head (ROWDOT) in the epilogue" % (rows_scored, h, h)
    res["roofline_mfma"] = dict(_gemm_forms(P, 2.0 * rows_scored * h * h, kname, rows_scored, h, launch),
                                subject="the step's own scorer product (MLPPredictor's hidden layer, layer.py:83-86)")
    return res


def measure_step_launches_citation2(P, model, data, cfg, device):
    """The citation2 step's dominant launches AS THE STEP MAKES THEM (GCN x2 over [table | features], MLP scorer):
      aggregation   the first layer aggregates FIRST (ops.GCNInputConvFn): A_hat applied to the table itself, kept padded to
                    64 columns -- csr_aggregate(adj, table64, weighted sum) over all 2.93 M rows, forward and transposed: the two
                    longest aggregation launches of the step.  The 750 MB source is far beyond the caches: `bound: hbm`; the graph
                    is skewed (hub rows come back out of the caches), so frac is taken on the COMPULSORY bytes, the gather-model
                    rate is `effective_GBps`
      GEMM          the second layer's data gradient  dz1 = (A_hat^T dz2) W2 with the first layer's relu / dropout gate in the
                    epilogue: M = 2.93 M rows, N = K = 200 (one 224-column tile)."""
    from plnlp_amd import _lib
    ops = P.ops
    adj = data.adj_t
    table = ops.padded_base(model.emb.weight.detach())
    if table is None:
        e = model.emb.weight.shape[1]
        table = torch.zeros(model.emb.weight.shape[0], (e + 15) // 16 * 16, device=device)
        table[:, :e].copy_(model.emb.weight.detach())
    n, F = table.shape
    out = torch.empty(n, F, device=device)
    t = time_kernel(lambda: ops.csr_aggregate(adj, table, "sum", True, out=out))
    by = agg_bytes(adj.nnz, n, F, True)
    by_min = agg_bytes_compulsory(adj.nnz, n, n, F, True)
    res = {"roofline_workload_agg": {
        "bound": "hbm", "subject": "the step's own first-layer aggregation launch (A_hat x padded table, all %d rows, F = %d)" % (n, F),
        "kernel": "csr_agg_vec_kernel<weighted, 16 lanes per row> + csr_agg_chunk_kernel (F=%d)" % F,
        "achieved": by_min / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by_min / t / 8.0e12, "traffic": None,
        "algorithmic_bytes": by_min, "bytes_model": "compulsory (index + value lists, every source row once, the result once)",
        "gather_model_bytes": by, "effective_GBps": by / t / 1e9, "kernel_ms": t * 1e3, "source_MiB": n * F * 4 / 2 ** 20,
        "note": "skewed graph: the gather model (%.1f GB) over-counts what crosses the HBM pins -- hub rows are re-read from the "
                "caches -- so the fraction is taken on the compulsory bytes; a 256-byte row is two 128-byte lines per gathered "
                "neighbour, the launch is bound by requests in flight (rows per wave), not by bytes" % (by / 1e9)}}
    h = cfg["hidden"]
    w2 = model.encoder.convs[1].lin.weight
    gxw = torch.randn(n, h, device=device)
    gate = torch.randn(n, h, device=device)
    dz1 = torch.empty(n, h, device=device)

    def launch():
        epi = _lib.make_epilogue(gate=gate, gate_scale=1.0 / (1.0 - cfg["dropout"]) if cfg["dropout"] > 0 else 1.0)
        ops.gemm([(gxw, w2)], False, False, out=dz1, epilogue=epi)
    kname = "M=%d, N=%d, K=%d, gate epilogue (the first layer's relu / dropout derivative)" % (n, h, h)
    res["roofline_mfma"] = dict(_gemm_forms(P, 2.0 * n * h * h, kname, n, h, launch),
                                subject="the step's own largest product: the second GCN layer's data gradient")
    return res


def measure_gemm_roofline(P, n_rows, k_in, hidden, device, sage=True):
    """the other hot kernel: the encoder's forward linear on the f32-input MFMA (SAGE: lin_l(agg) + lin_r(x)
    as one concat-K product with bias/relu/dropout in the epilogue), timed live like the aggregation"""
    from plnlp_amd import _lib
    k_in = (k_in + 3) // 4 * 4          # the encoder's operands are 16-byte-aligned (padded) buffers
    a1 = torch.randn(n_rows, k_in, device=device)
    w1 = torch.randn(hidden, k_in, device=device) * 0.05
    segs = [(a1, w1)]
    if sage:
        segs.append((torch.randn(n_rows, k_in, device=device), torch.randn(hidden, k_in, device=device) * 0.05))
    bias = torch.zeros(hidden, device=device)
    out = torch.empty(n_rows, hidden, device=device)
    epi = _lib.make_epilogue(bias=bias, relu=True, dropout_p=0.3, dropout_seed=1)
    flop = 2.0 * n_rows * hidden * k_in * len(segs)
    kname = "M=%d, N=%d, K=%s, bias+relu+dropout epilogue" % (n_rows, hidden, "+".join([str(k_in)] * len(segs)))
    mode0 = P.ops.GEMM_MATH["mode"]
    times = {}
    try:
        for mode in ("f32", "bf16x3"):             # both forms of the product, the path's own one reported first
            P.ops.GEMM_MATH["mode"] = mode
            times[mode] = time_kernel(lambda: P.ops.gemm(segs, False, True, out=out, epilogue=epi))
    finally:
        P.ops.GEMM_MATH["mode"] = mode0
    f32_form = {"kernel": "g16::gemm_f32_kernel (%s)" % kname, "achieved": flop / times["f32"] / 1e12, "peak": 157.3,
                "unit": "TFLOP/s", "frac": flop / times["f32"] / 157.3e12, "kernel_ms": times["f32"] * 1e3,
                "note": "v_mfma_f32_32x32x2_f32 (an fmaf chain; gfx950 has no TF32), peak = 256 CUs x 256 FLOP/clk x 2.4 GHz"}
    t = times["bf16x3"]
    x3_form = {"kernel": "%s (%s)" % (_x3_kernel_name(P, lambda: P.ops.gemm(segs, False, True, out=out, epilogue=epi)), kname), "achieved": 6 * flop / t / 1e12, "peak": 2500.0,
               "unit": "TFLOP/s", "frac": 6 * flop / t / 2.5e15, "kernel_ms": t * 1e3,
               "f32_equivalent_TFLOPs": flop / t / 1e12, "f32_equivalent_over_f32_mfma_peak": flop / t / 157.3e12,
               "note": "fp32 in / fp32 out; every operand element split in the loader into three bf16 terms, six "
                       "v_mfma_f32_32x32x16_bf16 per 32x32x16 block (executed flops = 6 x algorithmic), f32 accumulate; "
                       "per-product error <= the f32 MFMA's (tests/test_hip_round2.py, profiles/r02_gemm_bf16x3_error.jsonl); "
                       "peak = dense bf16 MFMA at 2.4 GHz, the kernel itself runs power-limited at ~1.75 GHz"}
    mine, other = (x3_form, f32_form) if mode0 == "bf16x3" else (f32_form, x3_form)
    return dict({"bound": "mfma", "math": mode0, "traffic": None, "flops": flop}, **mine,
                **{"other_form": dict({"math": "f32" if mode0 == "bf16x3" else "bf16x3"}, **other)})


def measure_step_launches(P, model, data, pos_b, neg_b, cfg, device):
    """The three dominant launches of the collab step AS THE STEP MAKES THEM (1-layer SAGE on the embedding
    table, row-sparse forward and backward), each timed live with device events on the launch stream, on the
    index structures of a real batch:
      forward aggregation   csr_aggregate(adj, table, mean, row_index = touched rows)              [T, F]
      forward GEMM          [agg | table[rows]] @ [Wl | Wr]^T + bias/relu/dropout, root rows gathered (a_index)
      transposed agg + Adam csr_aggregate(adj^T D^-1, gagg_c, src_map) + indexed addend, Adam step on the table in
                            the epilogue (PLNLP_EPI_ADAM) -- the step's longest launch
    Byte models: `gather` = what the kernel requests (every neighbour row once per entry),
    `compulsory` = what must cross the HBM pins at least once."""
    from plnlp_amd import _lib
    ops = P.ops
    adj = data.adj_t
    n, F = model.emb.weight.shape
    h = cfg["hidden"]
    batch = model.prepare_edges(pos_b, neg_b)
    inc = batch.join().incidence
    T, Tp = inc.count, inc.n_rows
    rows, node_map = inc.rows, inc.node_map
    x = model.emb.weight.detach()
    conv = model.encoder.convs[0]
    deg = adj.degree()
    nnz_T = int(deg[rows[:T].long()].sum().item())
    touched_nnz = node_map[adj.row_index()] >= 0
    distinct_src = int(torch.unique(adj.col[touched_nnz]).numel())
    out = {}
    # ---- forward aggregation over the touched rows
    agg = torch.empty(Tp, F, device=device)
    t = time_kernel(lambda: ops.csr_aggregate(adj, x, "mean", False, row_index=rows, out_map=node_map, out=agg))
    by = nnz_T * (4 * F + 4) + Tp * (4 * F + 4 + 16)
    by_min = nnz_T * 4 + Tp * (4 + 16) + distinct_src * 4 * F + Tp * 4 * F
    src_mib = n * F * 4 / 2 ** 20
    out["roofline_workload_agg"] = {
        "bound": "cache" if src_mib <= 256 else "hbm", "subject": "the step's own forward aggregation launch",
        "kernel": "csr_agg_fused_kernel (mean, F=%d, row_index: %d touched rows of %d; short rows + hub chunks) + "
                  "csr_agg_finalize_kernel" % (F, T, n),
        "kernel_form": P.ops.describe_form(adj._agg_tune.get(F, 0)),
        "achieved": by_min / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by_min / t / 8.0e12, "traffic": None,
        "algorithmic_bytes": by_min, "bytes_model": "compulsory", "gather_model_bytes": by, "effective_GBps": by / t / 1e9,
        "kernel_ms": t * 1e3, "source_MiB": src_mib, "rows": T, "entries": nnz_T, "distinct_source_rows": distinct_src,
        "note": "source matrix (%.0f MiB) fits the 256 MiB Infinity Cache: cache-bound; frac = compulsory bytes over the "
                "HBM peak, effective_GBps = gather-model rate (not a roofline fraction)" % src_mib}
    # ---- forward GEMM on the touched rows, root operand gathered in the loader
    epi = _lib.make_epilogue(bias=conv.lin_l.bias, relu=True, dropout_p=cfg["dropout"], dropout_seed=1,
                             dropout_rows=rows)
    y = torch.empty(Tp, h, device=device)
    flop = 2.0 * Tp * h * 2 * F
    mode0 = ops.GEMM_MATH["mode"]
    times = {}
    try:
        for mode in ("f32", "bf16x3"):
            ops.GEMM_MATH["mode"] = mode
            times[mode] = time_kernel(lambda: ops.gemm([(agg, conv.lin_l.weight), (x, conv.lin_r.weight)], False, True,
                                                       out=y, epilogue=epi, a_index=[None, rows]))
    finally:
        ops.GEMM_MATH["mode"] = mode0
    kname = "M=%d (touched rows), N=%d, K=%d+%d, root rows gathered (a_index), bias+relu+dropout epilogue" % (Tp, h, F, F)
    f32_form = {"math": "f32", "kernel": "g16::gemm_f32_kernel (%s)" % kname, "achieved": flop / times["f32"] / 1e12,
                "peak": 157.3, "unit": "TFLOP/s", "frac": flop / times["f32"] / 157.3e12, "kernel_ms": times["f32"] * 1e3}
    tx = times["bf16x3"]
    x3_form = {"math": "bf16x3", "kernel": "%s (%s)" % (_x3_kernel_name(P, lambda: ops.gemm([(agg, conv.lin_l.weight), (x, conv.lin_r.weight)], False, True,
                                                                                           out=y, epilogue=epi, a_index=[None, rows])), kname), "achieved": 6 * flop / tx / 1e12,
               "peak": 2500.0, "unit": "TFLOP/s", "frac": 6 * flop / tx / 2.5e15, "kernel_ms": tx * 1e3,
               "f32_equivalent_TFLOPs": flop / tx / 1e12,
               "note": "fp32 in / fp32 out; operands split into three bf16 terms, six bf16 MFMAs per block (executed flops = "
                       "6 x algorithmic) against the dense bf16 peak"}
    mine, other = (x3_form, f32_form) if mode0 == "bf16x3" else (f32_form, x3_form)
    out["roofline_mfma"] = dict({"bound": "mfma", "subject": "the step's own forward GEMM launch", "traffic": None,
                                 "flops": flop}, **mine, other_form=other)
    # ---- transposed aggregation with the indexed addend and the table's Adam step in the epilogue
    gt = adj.t_mean()
    gagg_c = torch.randn(Tp, F, device=device) * 1e-3
    gx_c = torch.randn(Tp, F, device=device) * 1e-3
    table, m, v = x.clone(), torch.zeros_like(x), torch.zeros_like(x)
    step_no = [0]

    def agg_adam():
        step_no[0] += 1
        e = _lib.make_epilogue(addend=gx_c, addend_index=node_map, adam=(m, v, step_no[0], 1e-3, 0.9, 0.999, 1e-8))
        ops.csr_aggregate(gt, gagg_c, "sum", True, src_map=node_map, out=table, epilogue=e)
    t = time_kernel(agg_adam)
    mapped = int(touched_nnz.sum().item())       # entries whose source row carries a gradient (graph symmetric)
    by_min = gt.nnz * 8 + (n + 1) * 8 + 2 * n * 4 + 2 * Tp * 4 * F + 6 * n * 4 * F
    by = gt.nnz * 8 + (n + 1) * 8 + n * 4 + gt.nnz * 4 + mapped * 4 * F + T * 4 * F + 6 * n * 4 * F
    out["roofline_agg_adam"] = {
        "bound": "hbm", "subject": "the step's own backward launch: transposed aggregation + Adam on the table (the "
                                   "step's longest kernel)",
        "kernel": "csr_agg_fused_kernel<weighted> (F=%d, src_map, ADDEND|ADAM epilogue; short rows + hub chunks) + "
                  "csr_agg_finalize_kernel" % F,
        "achieved": by_min / t / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by_min / t / 8.0e12, "traffic": None,
        "algorithmic_bytes": by_min, "bytes_model": "compulsory: index + value lists, node map twice, the compact gradient "
                                                    "and addend once, 3 reads + 3 writes of the [N, F] table / moments",
        "gather_model_bytes": by, "effective_GBps": by / t / 1e9, "kernel_ms": t * 1e3,
        "rows_with_gradient": T, "mapped_entries": mapped,
        "note": "param + two moments (3 x %.0f MiB, read and written once each) cannot stay in the 256 MiB Infinity Cache: "
                "that stream is HBM traffic; the %d gathered gradient rows (%.0f MiB) are cache-resident" %
                (src_mib, T, Tp * F * 4 / 2 ** 20)}
    out["touched_fraction"] = T / float(n)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            for key, obj in (("step_collab_fwd_agg", "roofline_workload_agg"), ("step_collab_agg_adam", "roofline_agg_adam")):
                if tj.get(key) is not None and cfg["shape"] == "collab":
                    out[obj]["traffic_from_profile"] = {
                        "bytes": tj[key], "file": "profiles/traffic.json",
                        "ratio_to_algorithmic": tj[key] / out[obj]["algorithmic_bytes"],
                        "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over scripts/bench_step_launches.py (the same "
                               "launch on a real batch; counted on the fabric side of L2, so Infinity-Cache hits are "
                               "included; taken on the round-5 tree -- profiles/r05_agg_pmc_step_launches.json -- in a "
                               "separate counter run, NOT measured in this one)"}
        except Exception:
            pass
    return out


def cpu_baseline(cfg, g, pos, neg, w, steps):
    """the CPU oracle (port of the reference PyG CPU path) on the same workload"""
    import oracle as O
    n, h = g["num_nodes"], cfg["hidden"]
    from plnlp_amd.utils import host_cpu_budget
    cores = host_cpu_budget()            # affinity mask capped by the cgroup quota (a 256-core box may grant 16 CPUs)
    torch.set_num_threads(cores)
    adj = g["adj_t"]
    csr = O.CSR(adj.rowptr.cpu(), adj.col.cpu().to(torch.int64), None if adj.val is None else adj.val.cpu(), n)
    feats = cfg.get("feats", 0)
    e_dim = cfg.get("emb", h)
    enc = O.GNNRef(cfg["encoder"], e_dim + feats, h, h, cfg["gnn_layers"], cfg["dropout"], spmm_impl="sparse_csr")
    pred = O.DotPredictorRef() if cfg["predictor"] == "DOT" else O.MLPPredictorRef(h, h, 1, cfg["mlp_layers"], cfg["dropout"])
    emb = torch.nn.Embedding(n, e_dim)
    x = g["data"].x.cpu() if feats else None
    tr = O.TrainerRef(enc, pred, emb, csr, x=x, loss_name=cfg["loss"], lr=1e-3, clip_norm=cfg["clip"])
    tr.param_init()
    enc.train()
    B = cfg["batch"]
    tr.step(pos[:B], neg[:B], cfg["num_neg"], None if w is None else w[:B])     # warm
    t0 = time.perf_counter()
    for i in range(steps):
        sl = slice((i % 2) * B, (i % 2) * B + B)
        tr.step(pos[sl], neg[sl], cfg["num_neg"], None if w is None else w[sl])
    dt = (time.perf_counter() - t0) / steps
    return {"value": B * (1 + cfg["num_neg"]) / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps (B=%d, k=%d) of the CPU oracle on the same synthetic %s-shaped "
                      "graph, torch %s CPU, %d threads (the host shows %d cores, the container's CPU quota grants %d); "
                      "%.2f s/step" %
                      (steps, B, cfg["num_neg"], cfg["shape"], torch.__version__, cores, os.cpu_count(), cores, dt)}


def hits_parity(P, device, seed=0):
    """Hits@50 parity (BASELINE.json metric) THROUGH THE KERNELS THE TIMED STEPS RUN: the collab recipe at its own width
    (README.md:35 at h = 256: SAGE x1 + DOT, WeightedHingeAUC on walk pairs, lr decay -- tests/trained_parity.py::RECIPES[
    "collab_wide"], a 40 000-node soft geometric graph with hub rows) trained for its 8 epochs on the HIP path from the run's fixed
    initial weights / walks / negatives / permutations, against the float32 and float64 CPU-oracle runs of the SAME seed, which
    are data: tests/golden/g12_trained_curves_wide.npz (written by tests/golden/make_trained_curves.py; 79 CPU-minutes for all
    seeds).  One seed is a smoke-level statement; the statistical one (16 seeds, +-0.3) is tests/test_hip_round5.py."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import trained_parity as T
    recipe = "collab_wide"
    g12 = np.load(os.path.join(ROOT, "tests", "golden", "g12_trained_curves_wide.npz"))
    ref32, ref64 = g12[recipe + "_f32"][seed].astype(float), g12[recipe + "_f64"][seed].astype(float)
    t0 = time.perf_counter()
    c0 = P.ops.launch_counts()
    hits, losses = T.run_hip(P, recipe, seed, P.ops.GEMM_MATH["mode"])
    c1 = P.ops.launch_counts()
    ki = T.metrics_of(recipe).index("Hits@50")
    r = T.RECIPES[recipe]
    return {"metric": "Hits@50", "recipe": "collab_wide: README.md:35 at h = %d on a %d-node geometric graph, %d epochs, batch %d"
                                           % (r["h"], T.PROBLEMS[r["problem"]]["num_nodes"], r["epochs"], r["batch"]),
            "seed": seed, "epochs": int(hits.shape[0]),
            "gpu_valid": float(hits[-1, ki, 0]), "gpu_test": float(hits[-1, ki, 1]),
            "cpu_valid": float(ref32[-1, ki, 0]), "cpu_test": float(ref32[-1, ki, 1]),
            "cpu64_valid": float(ref64[-1, ki, 0]), "cpu64_test": float(ref64[-1, ki, 1]),
            "per_epoch_valid": {"gpu": [round(float(v), 2) for v in hits[:, ki, 0]], "cpu": [round(float(v), 2) for v in ref32[:, ki, 0]],
                                "cpu64": [round(float(v), 2) for v in ref64[:, ki, 0]]},
            "max_abs_diff_points": float(np.abs(hits[:, ki, :] - ref32[:, ki, :]).max()),
            "cpu32_vs_f64_points": float(np.abs(ref32[:, ki, :] - ref64[:, ki, :]).max()),
            "epoch1_loss": {"gpu": float(losses[0]), "cpu": float(g12[recipe + "_f32_loss"][seed, 0]),
                            "cpu64": float(g12[recipe + "_f64_loss"][seed, 0])},
            "kernel_families": {k_: c1[k_] - c0[k_] for k_ in c1 if c1[k_] != c0[k_]},
            "seconds": time.perf_counter() - t0,
            "note": "percent, held-out edges vs 10 000 negatives; `cpu` / `cpu64` = the CPU oracle's float32 / float64 run of the same "
                    "seed (fixture); max_abs_diff_points is over all 8 epochs, valid and test, beside the two oracles' own distance"}


def main():
    args = parse()
    cfg = WORKLOADS.get(args.workload)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))         # nothing has touched the GPU yet in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run_cpu:
        sys.exit(dry_run_cpu(world, rank, args.dry_run_fail_rank))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = None
    if world > 1 or args.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
        import datetime
        if args.share_gpu:
            torch.distributed.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        else:
            torch.distributed.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(minutes=10))
        pg = torch.distributed.group.WORLD
        # every collective the step will use, once, on a tiny tensor with a known answer -- BEFORE anything expensive:
        # a broken fabric / environment fails here, by name, on every rank (its stderr is kept per rank by the launcher)
        from plnlp_amd import shard as _shard
        selftest = sorted(_shard.collective_self_test(pg, device))
        print("rank %d: collective self-test ok: %s" % (rank, ", ".join(selftest)), file=sys.stderr, flush=True)

    import plnlp_amd as P
    from plnlp_amd import synthetic
    P._lib.load()
    from plnlp_amd.utils import limit_host_threads
    limit_host_threads(world)      # stay inside the container's CPU quota (utils.host_cpu_budget)
    if args.workload == "rmat" and args.as_rank:
        return run_rmat_as_rank(args, P, device)
    if args.workload == "rmat":
        return run_rmat_stress(args, P, world, rank, device, pg)

    K, W, B, k = args.steps, args.warmup, cfg["batch"] * args.batch_mult, cfg["num_neg"]
    torch.manual_seed(1234)
    P.manual_seed(1234)
    # ---- synthetic inputs (identical on every rank: same seeds) -------------------
    g = synthetic.make_graph(cfg["shape"], seed=2, device=device, scale=args.scale, weighted=cfg["weighted"])
    n = g["num_nodes"]
    data = g["data"]
    if cfg["encoder"] == "GCN":          # main.py:177-179
        g["adj_t"] = data.adj_t = P.gcn_normalization(g["adj_t"])
    feats = cfg.get("feats", 0)
    if feats:
        data.x = torch.randn(n, feats, device=device, generator=torch.Generator(device=device).manual_seed(5))
    gen = torch.Generator(device=device).manual_seed(777)
    need = max(K + W, 40) * B * world          # (the side measurements -- other arithmetic, captured loop -- take up to 36 steps)
    if cfg["shape"] == "collab":      # random-walk augmented pairs, main.py:241-253
        starts = g["edges"].reshape(-1)
        reps = (need // (starts.numel() * 9)) + 1
        pairs, weights = P.ops.random_walk_pairs(g["adj_t"], starts.repeat(reps), 10, 777)
        sel = torch.randperm(pairs.size(0), generator=gen, device=device)[:need]
        pos_all, w_all = pairs[sel], weights[sel]
        epoch_pairs, epoch_weights = pairs, weights          # one epoch of the recipe = every random-walk pair
    else:
        sel = torch.randint(0, g["edges"].size(0), (need,), generator=gen, device=device)
        pos_all, w_all = g["edges"][sel], None
        epoch_pairs, epoch_weights = g["edges"], None         # one epoch = every training edge (model.py:147)
    # negatives: the recipe's own sampler (README.md:24,35: 'global' for ddi / collab, :40 'local' for
    # citation2), run the way BaseModel.train runs it -- once per "epoch" (here: all the steps of the run),
    # the structured one on the device the edge list lives on.  Timed, reported, outside the timed steps
    # like in the reference (model.py:132-136).
    torch.cuda.synchronize()
    t_s = time.perf_counter()
    sampler = "local" if cfg["shape"] == "citation2" else "global"
    if sampler == "global":
        row, col, _ = g["adj_t"].coo()
        neg_all = P.negative_sample.global_neg_sample(torch.stack([col, row]), n, need, k)
    else:
        neg_all = P.negative_sample.local_neg_sample(pos_all.cpu(), n, k).to(device)
    torch.cuda.synchronize()
    sampler_s = time.perf_counter() - t_s
    assert neg_all.shape == (need, k, 2) and neg_all.is_cuda

    exchange = args.dp_exchange
    prediction = None
    def make_model(group, how):
        m = P.BaseModel(lr=1e-3, dropout=cfg["dropout"], grad_clip_norm=cfg["clip"],
                        gnn_num_layers=cfg["gnn_layers"], mlp_num_layers=cfg["mlp_layers"],
                        emb_hidden_channels=cfg.get("emb", cfg["hidden"]), gnn_hidden_channels=cfg["hidden"],
                        mlp_hidden_channels=cfg["hidden"], num_nodes=n, num_node_feats=feats,
                        gnn_encoder_name=cfg["encoder"], predictor_name=cfg["predictor"], loss_func=cfg["loss"],
                        optimizer_name="Adam", device=device, use_node_feats=feats > 0, train_node_emb=True,
                        process_group=group, dp_exchange=how)
        m.param_init()
        m.encoder.train()
        m.predictor.train()
        return m

    model = None

    def sync():
        if pg is not None:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed_steps(m, mode, ranks, global_batch, my_rank, collective=True, K=K, W=W):
        """W warm-up + K timed steps of `m` at the given GLOBAL batch over `ranks` ranks (every rank holds the
        same resident edge tensors); returns (seconds of the K steps, max over ranks; last loss)"""
        plans = {}
        pipes = {}

        def step(i):
            sl = slice(i * global_batch, (i + 1) * global_batch)
            wts = None if w_all is None else w_all[sl]
            if mode == "shard":       # every rank passes the global batch; it scores its own slice of it
                # the request plan of the next batch is started one step ahead, as BaseModel.train does
                plan = plans.pop(i, None)
                if plan is None:
                    plan = m.shard_plan(pos_all[sl], neg_all[sl], k)
                if i + 1 < W + K:
                    nx = slice((i + 1) * global_batch, (i + 2) * global_batch)
                    plans[i + 1] = m.shard_plan(pos_all[nx], neg_all[nx], k)
                return m.train_step_sharded(data, pos_all[sl], neg_all[sl], k, wts, plan=plan)
            if mode == "scores":
                return m.train_step_global(data, pos_all[sl], neg_all[sl], k, wts, edges_ready=True)
            per = global_batch // ranks

            def mine(j):
                return slice(j * global_batch + my_rank * per, j * global_batch + (my_rank + 1) * per)
            if ranks == 1 and m.process_group is None:
                # one process: the loop BaseModel.train runs -- one batch of look-ahead and the step itself through the
                # step pipeline (plnlp_amd/capture.py: two hipGraphs once the model is warm)
                pipe = m.pipeline(data, k, per, w_all is not None, capture=capture_flag[0],
                                  tag=(P.ops.GEMM_MATH["mode"], P.ops.SPARSE_FORWARD["enabled"], capture_flag[0]))
                pipes[0] = pipe

                def prep_of(j):
                    return pipe.prepare(pos_all[mine(j)], neg_all[mine(j)], None if w_all is None else w_all[mine(j)])
                handle = plans.pop(i, None)
                if handle is None:
                    handle = prep_of(i)
                if i + 1 < W + K:
                    plans[i + 1] = prep_of(i + 1)
                return pipe.step(handle, global_count=global_batch)
            # the edge-only pre-processing of the NEXT batch is started before this step is enqueued, as
            # BaseModel.train does (BaseModel.prepare_edges)
            prep = plans.pop(i, None)
            if prep is None:
                prep = m.prepare_edges(pos_all[mine(i)], neg_all[mine(i)])
            if i + 1 < W + K:
                plans[i + 1] = m.prepare_edges(pos_all[mine(i + 1)], neg_all[mine(i + 1)])
            return m.train_step(data, pos_all[mine(i)], neg_all[mine(i)], k, None if w_all is None else w_all[mine(i)],
                                edges_ready=True, global_count=global_batch, prepared=prep)
        fence = sync if collective else torch.cuda.synchronize      # a solo run beside idle ranks: no barrier
        for i in range(W):
            step(i)
        fence()
        waited0 = P.ops.StepThrottle.waited_s
        t0 = time.perf_counter()
        last = None
        for i in range(W, W + K):
            last = step(i)
        host_enqueue_s.append(time.perf_counter() - t0)      # when the host had queued everything (diagnostic)
        host_busy_s.append(host_enqueue_s[-1] - (P.ops.StepThrottle.waited_s - waited0))   # ... minus its waits for the GPU
        fence()
        dt = time.perf_counter() - t0
        if ranks > 1 and collective:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        if pipes:
            last_pipe.clear()
            last_pipe.append(pipes[0])
        return dt, float(last.item())

    n_ranks = world if pg is not None else 1
    host_enqueue_s = []
    host_busy_s = []
    last_pipe = []
    capture_flag = [None]          # None = the product default (plnlp_amd/capture.py: CAPTURE["enabled"])
    phases = None
    probe_info = None
    if pg is not None:
        # ---- the cost model's inputs, MEASURED by this run on this rank's GPU: the plain one-process step, the table's
        # Adam, and every exchange form's step through a ONE-rank process group (its whole host / plan / separate-
        # optimiser overhead, every collective degenerate).  plnlp_amd/shard.py::cost_model returns these at world = 1.
        from plnlp_amd import shard as _shard
        Km, Wm = 8, 4
        # The phase is BOUNDED: a quarter of the launcher's deadline (PLNLP_BENCH_DEADLINE_S, at most 300 s).  A form whose probe
        # -- priced at 1.5 x what the solo probe took, model construction included -- would not fit the rest of the budget is
        # not measured (the cost model then chooses among the measured ones); the ranks decide that together (MAX over ranks).
        probe_budget_s = min(300.0, 0.25 * float(os.environ.get("PLNLP_BENCH_DEADLINE_S", "1500")))
        t_probe = time.perf_counter()
        solo = make_model(None, "auto")
        dt0, _ = timed_steps(solo, "none", 1, B, 0, collective=False, K=Km, W=Wm)
        solo_probe_s = time.perf_counter() - t_probe
        step_1gpu = dt0 / Km * 1e3
        emb = solo.emb.weight
        if P.ops.padded_base(emb.detach()) is not None:      # (a table kept padded is stepped as its whole buffer)
            emb = P.ops.padded_base(emb.detach())
        st = [torch.zeros_like(emb) for _ in range(3)]
        table_adam = time_kernel(lambda: P.ops.adam_step(emb.detach(), st[0], st[1], st[2], lr=0.0, step=1), iters=5, warm=2) * 1e3
        del solo, st, emb
        mine = pg
        if world > 1:                   # (every rank creates every group: new_group is collective)
            mine = [torch.distributed.new_group([r]) for r in range(world)][rank]
        forms = ["grads"] + (["scores"] if cfg["predictor"] == "DOT" else []) + (["shard"] if cfg["encoder"] == "SAGE" or feats else [])
        step_1rank, skipped = {}, []
        for form in forms:
            over = torch.tensor([time.perf_counter() - t_probe + 1.5 * solo_probe_s], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.all_reduce(over, op=torch.distributed.ReduceOp.MAX, group=pg)
            if float(over.item()) > probe_budget_s and step_1rank:         # (the first form is always measured)
                skipped.append(form)
                continue
            m1 = make_model(mine, form)
            dtf, _ = timed_steps(m1, m1.dp_mode(), 1, B, 0, collective=False, K=Km, W=Wm)
            step_1rank[m1.dp_mode()] = dtf / Km * 1e3
            del m1
        torch.cuda.empty_cache()
        probe_info = {"seconds": round(time.perf_counter() - t_probe, 2), "budget_s": probe_budget_s,
                      "steps_per_probe": Km + Wm, "forms_measured": sorted(step_1rank), "forms_skipped_for_time": skipped,
                      "note": "solo step + one probe per exchange form through a one-rank group; budget = min(300 s, a quarter of "
                              "PLNLP_BENCH_DEADLINE_S)"}
        last_pipe.clear()               # (the solo / one-rank models of this phase are gone: no capture_info about them)
        if world > 1:
            # every rank must reach the SAME choice: each measured on its own GPU with its own noise, and on a near-tie two
            # ranks would build different models and issue mismatched collectives.  The inputs are made rank-identical
            # (MAX over ranks: the slowest GPU paces a data-parallel step), and the choice is checked below
            names = sorted(step_1rank)
            v = torch.tensor([step_1gpu, table_adam] + [step_1rank[f] for f in names], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(v, op=torch.distributed.ReduceOp.MAX, group=pg)
            v = v.tolist()
            step_1gpu, table_adam = v[0], v[1]
            step_1rank = dict(zip(names, v[2:]))
        small = 4 * (cfg["gnn_layers"] * 3 * cfg["hidden"] * max(cfg["hidden"], cfg.get("emb", cfg["hidden"]) + feats)
                     + (cfg["mlp_layers"] * cfg["hidden"] * cfg["hidden"] if cfg["predictor"] == "MLP" else 0))
        prediction = _shard.cost_model(n_nodes=n, emb_width=cfg.get("emb", cfg["hidden"]), hidden=cfg["hidden"],
                                       param_bytes_small=small, batch_per_rank=B, num_neg=k, world=world,
                                       step_ms_1gpu=step_1gpu, table_adam_ms=table_adam,
                                       scorer_has_params=cfg["predictor"] != "DOT", step_ms_1rank=step_1rank)
    if exchange == "default":
        if prediction is not None and prediction["choice"] in prediction["inputs"]["measured_1rank_forms"]:
            exchange = prediction["choice"]
        else:
            exchange = "auto" if pg is None else "grads"
    if pg is not None and world > 1:
        agreed = [exchange]
        torch.distributed.broadcast_object_list(agreed, src=0, group=pg)
        if agreed[0] != exchange:
            raise RuntimeError(f"rank {rank} chose exchange form {exchange!r}, rank 0 chose {agreed[0]!r}")
    model = make_model(pg, exchange)
    dp_mode = model.dp_mode()
    if pg is not None and world > 1:
        modes = [None] * world
        torch.distributed.all_gather_object(modes, dp_mode, group=pg)
        assert len(set(modes)) == 1, f"the ranks run different exchange forms: {modes}"
    if pg is not None and dp_mode != "shard":
        # all ranks agree on the aggregation kernel's form on the full graph NOW, collectively, so that the
        # rank-0-only measurements further down (roofline, control model) find the choice made: the op itself never
        # communicates (ops.tune_aggregation / ops._agg_tune)
        P.ops.tune_aggregation(g["adj_t"], [cfg["hidden"], cfg.get("emb", cfg["hidden"])], group=pg)
    counts0 = P.ops.launch_counts()
    dt, final_loss = timed_steps(model, dp_mode, n_ranks, B * world, rank)
    counts1 = P.ops.launch_counts()
    # which kernel families the measured steps went through (plnlp_launch_counts), per step, warm-up included
    launches_per_step = {kk: round((counts1[kk] - counts0[kk]) / float(K + W), 2) for kk in counts1 if counts1[kk] != counts0[kk]}
    host_ms_per_step = host_enqueue_s[-1] / K * 1e3          # (of THIS call: a process group's first calls are the cost model's)
    host_busy_ms_per_step = host_busy_s[-1] / K * 1e3
    if pg is not None:
        # ---- what the step's time is made of, so that a SCALE run can be decomposed: compute = the same form's step
        # through a one-rank group (same per-rank work, collectives degenerate); collective_alone = this form's payloads
        # moved by the same collectives with nothing else running; exposed = step - compute (what the collectives cost
        # the step after overlap).  At world = 1 exposed ~ 0 by construction.
        table_floats = n * cfg.get("emb", cfg["hidden"])
        buf = torch.empty(table_floats + small // 4, device=device)
        shard_out = torch.empty((table_floats + world - 1) // world * world // world, device=device)

        def coll():
            if dp_mode == "grads":
                torch.distributed.all_reduce(buf, group=pg)
            elif dp_mode == "shard":
                full = buf[: shard_out.numel() * world]
                torch.distributed.reduce_scatter_tensor(shard_out, full, group=pg)
                torch.distributed.all_gather_into_tensor(full, shard_out, group=pg)
            else:
                torch.distributed.all_reduce(buf[: 2 * B * world], group=pg)
        coll_ms = time_kernel(coll, iters=5, warm=2) * 1e3
        del buf, shard_out
        compute_ms = prediction["inputs"]["step_ms_1rank"].get(dp_mode) if prediction else None
        phases = {"step_ms": dt / K * 1e3, "compute_ms_1rank_group": compute_ms, "collective_alone_ms": coll_ms,
                  "exposed_ms": (dt / K * 1e3 - compute_ms) if compute_ms is not None else None,
                  "predicted_ms": prediction["ms_per_step_predicted"].get(dp_mode) if prediction else None,
                  "note": "compute = this exchange form's step through a ONE-rank process group on this GPU (measured in "
                          "this run); collective_alone = the form's table-sized payloads through the same RCCL collectives "
                          "with nothing else running; exposed = step - compute"}
    capture_info = None
    if last_pipe:
        pp = last_pipe[0]
        capture_info = {"default_path_captured": bool(pp.captured and pp.replays > 0), "why_eager": pp.why_eager}
        if world == 1:
            # the OTHER way to drive the same step (plnlp_amd/capture.py): timed the same way, reported beside the default
            other = not capture_info["default_path_captured"]
            capture_flag[0] = other
            try:
                Kc, Wc = max(10, min(K, 20)), 16          # (every bucket's graph captured before the timed steps)
                n_before = len(host_busy_s)
                dtc, _ = timed_steps(model, dp_mode, 1, B, 0, K=Kc, W=Wc)
                po = last_pipe[0]
                capture_info["other_path"] = {
                    "captured": bool(po.captured and po.replays > 0), "why_eager": po.why_eager,
                    "ms_per_step": dtc / Kc * 1e3, "host_busy_ms_per_step": host_busy_s[n_before] / Kc * 1e3,
                    "replayed_steps": po.replays, "graphs": (sum(len(sl.main) for sl in po.slots) if po.captured else 0),
                    "host_launches_per_step": "3 input copies + 2 graph replays + one 120-byte upload" if po.captured else None}
            finally:
                capture_flag[0] = None
        capture_info["note"] = ("plnlp_amd/capture.py: the step replayed from two hipGraphs (prologue one batch ahead on the side "
                                "stream, step on the main stream), bit-identical to the eager step; opt-in (PLNLP_CAPTURE=1): it "
                                "cuts the host's work per step ~4x but the GPU runs the replayed step ~2 % slower than the eagerly "
                                "queued one, and the step is GPU-bound")
    edges_per_step = B * (1 + k) * world

    extra = {}
    if world > 1 and not args.no_strong and K * B * world <= pos_all.size(0):
        # the same job at the reference's FIXED global batch (B positives in total, B / N per rank): what N GPUs
        # buy for the reference's own step; and, on rank 0 alone, ONE GPU fed the N-fold batch (what the
        # weak-scaling number would be without any second GPU)
        dts, _ = timed_steps(model, dp_mode, n_ranks, B, rank)
        extra["strong_scaling"] = {"global_batch": B, "ms_per_step": dts / K * 1e3,
                                   "value": B * (1 + k) * K / dts, "unit": "edges/s",
                                   "note": "same ranks, global batch fixed at the reference's B (B/N per rank)"}
        if rank == 0:
            solo = make_model(None, "auto")
            dtc, _ = timed_steps(solo, "none", 1, B * world, 0, collective=False)
            extra["control_1gpu_at_Nx_batch"] = {"global_batch": B * world, "ms_per_step": dtc / K * 1e3,
                                                 "value": B * world * (1 + k) * K / dtc, "unit": "edges/s",
                                                 "note": "ONE GPU (rank 0, no process group) stepping the N-fold batch"}
            del solo
        torch.distributed.barrier()

    sparse_fwd = bool(P.ops.SPARSE_FORWARD["enabled"] and cfg["gnn_layers"] >= 1)
    if world == 1:
        # ---- the same step under the two switches that change WHAT is launched (not what is computed) ----------
        Kv, Wv = max(5, min(K, 10)), 8          # (warm-up long enough for a fresh pipeline to have captured its graphs)
        old_math = P.ops.GEMM_MATH["mode"]
        other_math = "f32" if old_math == "bf16x3" else "bf16x3"
        P.ops.GEMM_MATH["mode"] = other_math
        try:
            dtv, _ = timed_steps(model, dp_mode, 1, B, 0, K=Kv, W=Wv)
        finally:
            P.ops.GEMM_MATH["mode"] = old_math
        extra["ms_per_step_%s" % ("f32_mfma" if other_math == "f32" else "bf16x3")] = dtv / Kv * 1e3
        # the same metric at the OTHER arithmetic, beside `value`: with the default split-bf16 products `value_f32_mfma` is
        # the rate on the exact-f32 MFMA (bit-for-bit an fma chain -- the reference's sgemm arithmetic)
        extra["value_%s" % ("f32_mfma" if other_math == "f32" else "bf16x3")] = B * (1 + k) * Kv / dtv
        if sparse_fwd:
            P.ops.SPARSE_FORWARD["enabled"] = False
            try:
                dtv, _ = timed_steps(model, dp_mode, 1, B, 0, K=Kv, W=Wv)
            finally:
                P.ops.SPARSE_FORWARD["enabled"] = True
            extra["ms_per_step_full_forward"] = dtv / Kv * 1e3
        # ---- edges/s through BaseModel.train for ONE EPOCH of the workload -----------------------
        # (model.py:128-173: per-epoch negative sampling + the DataLoader permutation + every step + the loss
        # read-back); the sampler's and the permutation's shares reported beside it.  citation2's epoch is 464
        # steps of ~29 ms: capped at --epoch-steps batches of edges.
        cap = args.epoch_steps * B
        ep_edges = epoch_pairs if epoch_pairs.size(0) <= cap else epoch_pairs[:cap]
        if sampler == "local":          # the local sampler draws on the CPU generator from CPU edge lists (negative_sample.py:31-43)
            ep_edges = ep_edges.cpu()
        tr = {"edge": ep_edges}
        if epoch_weights is not None:
            tr["weight"] = epoch_weights[:ep_edges.size(0)]
        split_ep = {"train": tr}
        model.train(data, split_ep, B, sampler, k)               # warm-up epoch (allocator, first-touch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ep_loss = model.train(data, split_ep, B, sampler, k)
        torch.cuda.synchronize()
        t_ep = time.perf_counter() - t0
        st = model.last_epoch
        extra["train_epoch"] = {
            "value": st["edges_scored"] / (t_ep - st["sampler_s"]), "unit": "edges/s", "steps": st["steps"],
            "positives": st["positives"], "epoch_s": t_ep, "sampler_s": st["sampler_s"],
            "permutation_s": st.get("permutation_s"), "ms_per_step": (t_ep - st["sampler_s"]) / st["steps"] * 1e3,
            "value_incl_sampler": st["edges_scored"] / t_ep, "loss": ep_loss,
            "full_epoch": bool(ep_edges.size(0) == epoch_pairs.size(0)),
            "note": "BaseModel.train (model.py:128-173) for one epoch: value = Sum_steps B_step (1 + k) / (wall time - "
                    "negative-sampler time); the wall time includes the DataLoader-equivalent CPU permutation "
                    "(permutation_s, bit-exact to the reference's: torch.randperm on the host) and the epoch's loss read-back"}
        # ---- eval scoring edges/s of test() (model.py:175-226) ----------------------------------------------------
        gen_e = torch.Generator(device=device).manual_seed(99)
        n_pos, n_neg = (60084, 100000) if cfg["shape"] == "collab" else (133489, 100000) if cfg["shape"] == "ddi" else (86596, 86596 * 50)
        def some_edges(cnt):
            return g["edges"][torch.randint(0, g["edges"].size(0), (cnt,), generator=gen_e, device=device)]
        def some_negs(cnt):
            return torch.randint(0, n, (cnt, 2), generator=gen_e, device=device)
        split_ev = {"train": tr, "valid": {"edge": some_edges(n_pos), "edge_neg": some_negs(n_neg)},
                    "test": {"edge": some_edges(n_pos), "edge_neg": some_negs(n_neg)}}
        ev = P.utils.Evaluator("ogbl-collab" if cfg["shape"] == "collab" else "ogbl-ddi")
        model.test(data, split_ev, B, ev, "hits")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps_ev = 3
        for _ in range(reps_ev):
            res_ev = model.test(data, split_ev, B, ev, "hits")
        torch.cuda.synchronize()
        t_ev = (time.perf_counter() - t0) / reps_ev
        model.encoder.train()
        model.predictor.train()
        extra["eval_scoring"] = {
            "value": 2 * (n_pos + n_neg) / t_ev, "unit": "edges/s", "edges": 2 * (n_pos + n_neg), "ms": t_ev * 1e3,
            "note": "BaseModel.test (model.py:184-226): eval-mode encoder over all %d nodes + mean row, valid and test "
                    "positives (%d each) and negatives (%d each) scored in batches of %d, Hits@20/50/100 ranked on the "
                    "device; value = scored edges / wall time of test()" % (n, n_pos, n_neg, B)}

    result = {
        "metric": "pos+neg edges scored/sec", "value": edges_per_step * K / dt, "unit": "edges/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "scaling_measured": bool(world > 1 and not args.share_gpu),
        "shared_gpu": bool(args.share_gpu),
        "scaling_note": ("per-GPU work fixed as N grows (weak).  No multi-GPU hardware was available to the build in any "
                         "round: no N > 1 value of this metric has ever been measured by it -- dp_prediction / dp_phases in an "
                         "N > 1 line are what to hold the driver's SCALE run against"),
        "gemm_math": P.ops.GEMM_MATH["mode"],
        "host_enqueue_ms_per_step": host_ms_per_step,
        "host_busy_ms_per_step": host_busy_ms_per_step,
        "host_note": "host_enqueue = wall time until the host had queued all K steps (it is held two steps ahead of the GPU, so "
                     "this tracks the GPU time); host_busy = the same minus the time the host slept waiting for the GPU: the "
                     "host WORK per step (Python, ctypes, autograd, launches -- or, captured, two graph replays)",
        "step_capture": capture_info,
        "config": {"workload": "ogbl-%s-shaped synthetic graph (N=%d, nnz=%d), %s x%d h=%d, %s predictor, "
                               "%s loss, B=%d/GPU, num_neg=%d, dropout=%.1f, %s"
                               % (cfg["shape"], n, g["adj_t"].nnz, cfg["encoder"], cfg["gnn_layers"], cfg["hidden"],
                                  cfg["predictor"], cfg["loss"], B, k, cfg["dropout"],
                                  "random-walk pairs (walk_length 10)" if cfg["shape"] == "collab"
                                  else "train-edge positives"),
                   "global_batch": B * world,
                   "parallelism": ("dp%d (edge batch sliced over ranks; encoder rows, embedding table and its Adam state "
                                   "sharded over ranks)" if dp_mode == "shard" else
                                   "dp%d (edge-batch, replicated encoder)") % world,
                   "dp_exchange": dp_mode, "scale": args.scale, "batch_mult": args.batch_mult,
                   "sparse_forward": sparse_fwd,
                   "sparse_forward_note": "the last conv is evaluated only at the rows the batch's edges read (and its "
                                          "backward runs on those rows); bit-identical loss / gradients / update to the "
                                          "full-matrix step, which is timed as ms_per_step_full_forward"},
        "final_loss": final_loss, "negative_sampling_s": sampler_s,
        "kernel_families_per_step": launches_per_step,
        "negative_sampler": "%s (plnlp_amd.negative_sample, %d negatives in one call)" % (sampler, need * k),
    }
    result.update(comm_info(pg))
    if pg is not None:
        result["rank_devices"] = rank_devices(pg, rank, local_rank, device)
        if rank == 0:
            result["xgmi_topology"] = xgmi_topology()
        result["probe_phase"] = probe_info
        result["collective_self_test"] = selftest
        result["dp_prediction"] = prediction
        result["dp_phases"] = phases
    result.update(extra)
    if pg is not None:
        result["replicas_in_sync"] = model.check_replicas()
    if rank == 0:
        nb = 2 * B
        pos_cpu, neg_cpu = pos_all[:nb].cpu(), neg_all[:nb].cpu()
        w_cpu = None if w_all is None else w_all[:nb].cpu()
        step_shaped = (world == 1 and cfg["encoder"] == "SAGE" and cfg["gnn_layers"] == 1 and sparse_fwd
                       and not args.no_roofline)
        if step_shaped:
            # the launches the step really makes (touched rows, gathered operands, the Adam epilogue)
            launches = measure_step_launches(P, model, data, pos_all[:B], neg_all[:B], cfg, device)
            result["config"]["touched_fraction"] = launches.pop("touched_fraction")
            result.update(launches)
        elif not args.no_roofline and world == 1 and cfg["shape"] == "ddi":
            result.update(measure_step_launches_ddi(P, model, data, cfg, device, B * (1 + k)))
        elif not args.no_roofline and world == 1 and cfg["shape"] == "citation2" and feats:
            result.update(measure_step_launches_citation2(P, model, data, cfg, device))
        elif not args.no_roofline:
            # the workload's own aggregation launch over all rows; `bound` says whether HBM or the caches limit it
            result["roofline_workload_agg"] = measure_roofline(P, g["adj_t"], cfg["hidden"], device,
                                                               weighted=cfg["encoder"] == "GCN", shape=cfg["shape"])
            if world == 1:
                result["roofline_mfma"] = measure_gemm_roofline(
                    P, n, cfg.get("emb", cfg["hidden"]) + feats, cfg["hidden"], device, sage=cfg["encoder"] == "SAGE")
        if world == 1 and not args.no_roofline and not args.no_stress:
            # `roofline`: the aggregation kernel against the HBM roofline proper (north_star: >= 60 % at h = 512), on a
            # citation2-sized graph without skew or locality at h = 512 (5.7 GB source matrix: no reuse to flatter the
            # byte model).  It is NOT a launch of the benchmarked step -- `subject` says so; the step's own launches are
            # roofline_workload_agg / roofline_mfma / roofline_agg_adam.
            del model, pos_all, neg_all
            torch.cuda.empty_cache()
            big = synthetic.uniform_graph(2_927_963, 30_387_995, device, seed=3)
            r = measure_roofline(P, big, 512, device, shape="uniform_big")
            r["graph"] = "uniform random, N=2927963, nnz=%d (citation2-sized), F=512" % big.nnz
            r["subject"] = ("the aggregation kernel on a graph whose source matrix (5.7 GB) is far beyond the caches: the "
                            "HBM roofline of the kernel -- NOT a launch of the benchmarked %s step (h=%d), whose own "
                            "launches are roofline_workload_agg / roofline_mfma / roofline_agg_adam"
                            % (cfg["shape"], cfg["hidden"]))
            result["roofline"] = r
            del big
            torch.cuda.empty_cache()
        elif "roofline_workload_agg" in result:
            result["roofline"] = dict(result["roofline_workload_agg"], subject="the workload's own aggregation launch")
        if world == 1 and not args.no_parity:
            result["hits50_parity"] = hits_parity(P, device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, g, pos_cpu, neg_cpu, w_cpu, args.cpu_steps)
    if pg is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:        # last thing on stdout: the one JSON line (RCCL's banner sits in the C stdio buffer)
        sys.stdout.flush()
        sys.stderr.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
