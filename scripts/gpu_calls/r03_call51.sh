#!/bin/bash
# round 3, call 51: round 2's tree (git archive of its last commit, built here) and this round's, interleaved on ONE box
O=$GRAFT_REPO_ROOT/gpurun_out/r03c51; mkdir -p $O
for i in 1 2 3; do
  (cd _r02_tree && python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/r02_$i.json 2>/dev/null)
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/r03_$i.json 2>/dev/null
done
for w in ddi citation2; do
  (cd _r02_tree && python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/r02_$w.json 2>/dev/null)
  python bench.py --workload $w --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-stress --no-roofline > $O/r03_$w.json 2>/dev/null
done
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03c51")
def last(f):
    return json.loads(open(os.path.join(O, f)).read().strip().splitlines()[-1])
for i in (1, 2, 3):
    a, b = last("r02_%d.json" % i), last("r03_%d.json" % i)
    print("collab run %d: round 2 tree %.4f ms (%.1f M edges/s, train_epoch %s)   round 3 tree %.4f ms (%.1f M, train_epoch %.1f M)" % (
        i, a["ms_per_step"], a["value"] / 1e6, ("%.1f M" % (a["train_epoch"]["value"] / 1e6)) if "train_epoch" in a else "n/a",
        b["ms_per_step"], b["value"] / 1e6, b["train_epoch"]["value"] / 1e6))
for w in ("ddi", "citation2"):
    a, b = last("r02_%s.json" % w), last("r03_%s.json" % w)
    print("%s: round 2 tree %.3f ms   round 3 tree %.3f ms" % (w, a["ms_per_step"], b["ms_per_step"]))
PY
