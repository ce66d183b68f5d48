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
    assert out["rccl_ranks"] == 2 and out["n_gpus"] == 2 and out["all_reduce_ok"] and out["dry_run"]


def test_bench_under_an_external_launcher_does_not_respawn():
    # WORLD_SIZE set by a launcher that started only this rank with a mismatching --gpus: refuse, do not spawn
    r = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, ["--gpus", "2", "--dry-run-cpu"])
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
