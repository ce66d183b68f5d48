"""every aggregation form against form 0 on R-MAT row blocks of growing size (mean of a constant vector + random x)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import plnlp_amd as P
from plnlp_amd import _lib as L, shard, synthetic

F = int(os.environ.get("F", 512))
for n, nnz in ((1 << 18, 1 << 22), (1_000_000, 20_000_000), (3_000_000, 60_000_000), (12_500_000, 250_000_000)):
    scale = (n - 1).bit_length()
    part = shard.RowPartition(n, 1, 0)
    blk = synthetic.rmat_row_block(scale, nnz, n, 0, part.rows, part.padded, "cuda", seed=11)
    deg = blk.degree()
    c = torch.randn(F, device="cuda")
    xc = c.expand(part.padded, F).contiguous()
    ref = P.ops.csr_aggregate(blk, xc, "mean", False, tune=0)
    nz = deg > 0
    print("n", n, "nnz", nnz, "max deg", int(deg.max()), "form 0 err", float((ref - c).abs().amax(1)[nz].max()), flush=True)
    for tune in (16, 32, 64, 128, 128 | P.ops.AGG_HUB_RANGES):
        out = P.ops.csr_aggregate(blk, xc, "mean", False, tune=tune)
        err = (out - c).abs().amax(1)
        bad = torch.nonzero(err * nz > 1e-4).flatten()
        msg = ""
        if bad.numel():
            b = bad[:5].tolist()
            msg = " BAD rows %d first %s deg %s err %s" % (bad.numel(), b, deg[bad[:5]].tolist(), err[bad[:5]].tolist())
            r = b[0]
            cols = torch.nonzero((out[r] - c).abs() > 1e-4).flatten()
            msg += " bad cols of first: n=%d range %d..%d" % (cols.numel(), int(cols.min()), int(cols.max()))
        print("   tune", tune, "max err", float(err[nz].max()), msg, flush=True)
        if tune & P.ops.AGG_HUB_RANGES:
            sp = blk.row_split(P.ops.split_threshold(blk.n_cols), P.ops.hub_ranges(blk.n_cols))
            print("   ranges: long", sp.n_long, "chunks", sp.n_chunks, "seg_len max", int(sp.seg_len.max()), "min", int(sp.seg_len.min()))
    del blk, xc, ref
    torch.cuda.empty_cache()
