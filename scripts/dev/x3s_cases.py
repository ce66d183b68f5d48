"""one stationary-weights GEMM per process (a fault kills the process): python scripts/dev/x3s_cases.py <case>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = {
    "nb8_deep": (40000, 256, 256, 8, 0, False), "nb8_shallow": (40000, 256, 256, 8, 1, False),
    "nb4_deep": (40000, 256, 256, 4, 0, False), "nb4_shallow": (40000, 256, 256, 4, 1, False),
    "nb8_deep_ragged": (40000, 256, 200, 8, 0, False), "nb8_deep_k32": (40000, 256, 32, 8, 0, False),
    "nb8_deep_k16": (40000, 256, 16, 8, 0, False), "nb8_deep_k48": (40000, 256, 48, 8, 0, False),
    "nb8_deep_gather": (40000, 256, 256, 8, 0, True),
    "nb8_deep_k64": (40000, 256, 64, 8, 0, False), "nb8_deep_k80": (40000, 256, 80, 8, 0, False),
    "nb8_deep_k128": (40000, 256, 128, 8, 0, False), "nb7_deep_k80": (40000, 200, 80, 7, 0, False),
}
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], (r.stderr.strip().splitlines() or [""])[-1][:200])
    sys.exit(0)
import torch
import plnlp_amd as P
from plnlp_amd import _lib
m, n, k, nb, shallow, gather = CASES[sys.argv[1]]
lib = _lib.load()
lib.plnlp_gemm_stationary_tuning(nb, shallow)
P.ops.GEMM_MATH["mode"] = "bf16x3"
gen = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(m, k, device="cuda", generator=gen)
w = torch.randn(n, k, device="cuda", generator=gen) * 0.1
if gather:
    table = torch.randn(3 * m, k, device="cuda", generator=gen)
    rows = torch.randperm(3 * m, device="cuda", generator=gen)[:m].sort().values.to(torch.int32)
    out = P.ops.gemm([(a, w), (table, w)], False, True, a_index=[None, rows])
    want = a.double() @ w.double().t() + table[rows.long()].double() @ w.double().t()
else:
    out = P.ops.gemm([(a, w)], False, True)
    want = a.double() @ w.double().t()
torch.cuda.synchronize()
err = float((out.double() - want).abs().max())
print("max abs err %.3e" % err, "ok" if err < 1e-3 else "WRONG")
