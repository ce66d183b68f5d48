"""bench.py --gpus N must start its own ranks when no launcher did (the driver's scaling run calls it
both ways).  Exercised without GPUs through --dry-run-cpu: N processes rendezvous over gloo on
127.0.0.1, all-reduce, and rank 0's JSON line is relayed by the parent."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_starts_its_own_ranks():
    r = _run({}, ["--gpus", "2", "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["backend"] == "gloo" and out["rccl_ranks"] == 0 and out["ranks"] == 2      # (RCCL ranks are counted under nccl only)
    assert out["n_gpus"] == 2 and out["all_reduce_ok"] and out["dry_run"]
    # the start-up self-test ran every collective the data-parallel forms use
    assert {"all_reduce_sum", "broadcast", "all_gather_into_tensor", "reduce_scatter_tensor",
            "all_to_all_single_uneven", "barrier"} <= set(out["collective_self_test"])


def test_bench_eight_ranks_dry_run():
    """what the driver's SCALE run launches, minus the GPUs: `bench.py --gpus 8` starts 8 ranks that rendezvous, run the
    collective self-test and an all-reduce over gloo, and rank 0 prints ONE line"""
    r = _run({}, ["--gpus", "8", "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["backend"] == "gloo" and out["rccl_ranks"] == 0 and out["ranks"] == 8
    assert out["n_gpus"] == 8 and out["all_reduce_ok"] and out["dry_run"]


def test_a_failing_rank_stops_its_siblings():
    """rank 1 dies before the rendezvous: the launcher must notice, end rank 0 (which would otherwise wait in the
    rendezvous for minutes) and return the failure -- quickly, with rank 1's message relayed"""
    import time
    t0 = time.time()
    r = _run({}, ["--gpus", "2", "--dry-run-cpu", "--dry-run-fail-rank", "1"])
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert "rank 1" in r.stderr and "fails on purpose" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_launcher_deadline():
    """an overall deadline bounds the job: a rank that never finishes is stopped"""
    r = _run({"PLNLP_BENCH_DEADLINE_S": "0.0"}, ["--gpus", "2", "--dry-run-cpu"])
    assert r.returncode == 124 and "deadline" in r.stderr


def test_bench_under_an_external_launcher_does_not_respawn():
    # WORLD_SIZE set by a launcher that started only this rank with a mismatching --gpus: refuse, do not spawn
    r = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, ["--gpus", "2", "--dry-run-cpu"])
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
