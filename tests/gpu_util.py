"""helpers shared by the GPU parity suites"""
import numpy as np
import torch

import oracle as O

RTOL = 1e-5


def dev(t):
    return t.cuda() if t is not None else None


def to_graph(P, csr: O.CSR):
    return P.Graph(csr.rowptr.clone(), csr.col.to(torch.int32), None if csr.val is None else csr.val.float(),
                   csr.n_rows, csr.n_cols).to("cuda")


def rand_csr(n, e, seed, weighted=True, n_cols=None, hub=None):
    g = torch.Generator().manual_seed(seed)
    n_cols = n if n_cols is None else n_cols
    r = torch.randint(0, n, (e,), generator=g)
    c = torch.randint(0, n_cols, (e,), generator=g)
    if hub is not None:       # one very long row and one empty row
        r = torch.cat([r, torch.full((hub,), 3)])
        c = torch.cat([c, torch.randint(0, n_cols, (hub,), generator=g)])
        keep = r != 5
        r, c = r[keep], c[keep]
    v = torch.rand(r.numel(), generator=g) + 0.1 if weighted else None
    return O.CSR.from_coo(r, c, v, n, n_cols)


def close(a, b, rtol=RTOL, atol=None, msg=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    if atol is None:
        atol = rtol * max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)
