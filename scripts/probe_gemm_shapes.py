"""product error of both GEMM forms against float64 over small / ragged shapes (the G8 toy's last batch: M = 219,
K = 8; its weight gradient: K = 219) -- max |C - C64| / sum_k |a||b| per shape and operand layout"""
import itertools
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plnlp_amd as P

torch.manual_seed(0)
worst = {}
for (at, bt) in ((False, True), (False, False), (True, False)):
    for m, n, k in itertools.product((8, 73, 219, 292, 1200, 200), (8, 64, 1), (8, 16, 24, 219, 292, 64, 1200)):
        if n == 1 and not (not at and bt):
            continue
        a64 = torch.randn(m, k, dtype=torch.float64)
        b64 = torch.randn(n, k, dtype=torch.float64)
        a, b = a64.float(), b64.float()
        ref = a.double() @ b.double().t()
        mag = a.double().abs() @ b.double().abs().t()
        A = (a.t().contiguous() if at else a).cuda()
        B = (b if bt else b.t().contiguous()).cuda()
        for math in ("f32", "bf16x3"):
            P.ops.GEMM_MATH["mode"] = math
            for sk in (None, 1):
                try:
                    c = P.ops.gemm([(A, B)], at, bt, split_k=sk).cpu().double()
                except Exception as e:
                    print("ERR", at, bt, m, n, k, math, sk, e)
                    continue
                err = float(((c - ref).abs() / mag).max())
                key = (math,)
                if err > 3e-7:
                    print(f"at={at} bt={bt} M={m} N={n} K={k} {math} split_k={sk}: rel err {err:.2e}")
                worst[key] = max(worst.get(key, 0), err)
print("worst", worst)
