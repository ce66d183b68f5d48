"""Which torch.distributed collectives does the gloo backend run on DEVICE tensors when W processes share ONE GPU?
(RCCL refuses two ranks on one device; the world > 1 GPU tests therefore run over gloo -- tests/test_hip_multirank.py --
and need to know which collectives must be staged through host memory by the test-only shim.)
python scripts/probe_gloo_cuda.py [world]"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _rank(rank, world, port, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    res = {}

    def attempt(name, fn):
        try:
            fn()
            torch.cuda.synchronize()
            res[name] = "ok"
        except Exception as exc:      # noqa: BLE001 -- the probe's whole point
            res[name] = "%s: %s" % (type(exc).__name__, str(exc).splitlines()[0][:120])

    x = torch.full((8,), float(rank + 1), device=dev)

    def allreduce():
        t = x.clone()
        dist.all_reduce(t)
        assert float(t[0]) == world * (world + 1) / 2
    attempt("all_reduce", allreduce)

    def allreduce_async():
        t = x.clone()
        w = dist.all_reduce(t, async_op=True)
        w.wait()
        assert float(t[0]) == world * (world + 1) / 2
    attempt("all_reduce_async", allreduce_async)

    def allreduce_max():
        t = x.double().clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t[0]) == world
    attempt("all_reduce_max_f64", allreduce_max)

    def bcast():
        t = x.clone()
        dist.broadcast(t, 0)
        assert float(t[0]) == 1.0
    attempt("broadcast", bcast)

    def bcast_i64():
        t = torch.tensor([rank + 5, 7], dtype=torch.int64, device=dev)
        dist.broadcast(t, 0)
        assert int(t[0]) == 5
    attempt("broadcast_i64", bcast_i64)

    def ag():
        out = torch.empty(8 * world, device=dev)
        dist.all_gather_into_tensor(out, x)
        assert float(out[-1]) == world
    attempt("all_gather_into_tensor", ag)

    def ag_async():
        out = torch.empty(8 * world, device=dev)
        dist.all_gather_into_tensor(out, x, async_op=True).wait()
        assert float(out[-1]) == world
    attempt("all_gather_into_tensor_async", ag_async)

    def rs():
        out = torch.empty(8, device=dev)
        dist.reduce_scatter_tensor(out, torch.ones(8 * world, device=dev))
        assert float(out[0]) == world
    attempt("reduce_scatter_tensor", rs)

    def a2a():
        send = torch.full((2 * world,), float(rank), device=dev)
        recv = torch.empty(2 * world, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=[2] * world, input_split_sizes=[2] * world)
        assert float(recv[-1]) == world - 1
    attempt("all_to_all_single", a2a)

    attempt("barrier", dist.barrier)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for rank, res in out:
        print("rank", rank, res)
    print("exit codes", [p.exitcode for p in procs])


if __name__ == "__main__":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main()
