"""The fused scorer's deterministic backward (plnlp_edge_segment_bwd_f32) alone, at the two step shapes that use the matrix-gradient
form: citation2-like (262 144 edges over ~300 K touched nodes, F = 200: two items per segment) and ddi-like (262 144 edges over
4 267 nodes, F = 512, endpoints drawn by degree).  Device-event times per form (PLNLP_EDGE_SEGMENT's settings); run under
`rocprofv3 --pmc` for the counters."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plnlp_amd as P

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(5)
shapes = {"citation2": (300_000, 262_144, 200, False), "citation2_hubs": (300_000, 262_144, 200, "local"), "ddi": (4_267, 262_144, 512, True)}
only = os.environ.get("PROBE_SHAPES", "citation2,citation2_hubs,ddi").split(",")
forms = os.environ.get("PROBE_FORMS", "wave,auto").split(",")
iters = int(os.environ.get("PROBE_ITERS", "10"))
for name in only:
    n, e, feat, skew = shapes[name]
    h = torch.randn(n, feat, device=dev, generator=gen)
    if skew == "local":
        # positives' endpoints drawn by a power-law degree, three negatives per positive anchored at its source (the recipe's
        # 'local' sampler): hubs collect hundreds of items
        w = (torch.arange(1, n + 1, device=dev, dtype=torch.float32)) ** -0.8
        w = w[torch.randperm(n, device=dev, generator=gen)]
        pos = torch.multinomial(w, e // 2, replacement=True, generator=gen).reshape(-1, 2)
        anchor = pos[:, 0].repeat_interleave(3)
        neg = torch.stack([anchor, torch.randint(0, n, (anchor.numel(),), device=dev, generator=gen)], 1)
        pairs = torch.cat([pos, neg])[:e]
    elif skew:
        w = torch.rand(n, device=dev, generator=gen) ** 3 + 0.01
        pos = torch.multinomial(w, e // 2, replacement=True, generator=gen).reshape(-1, 2)[: e // 4]
        neg = torch.randint(0, n, (e - pos.shape[0], 2), device=dev, generator=gen)
        pairs = torch.cat([pos, neg])
    else:
        pairs = torch.randint(0, n, (e, 2), device=dev, generator=gen)
    src, dst = pairs[:, 0].contiguous(), pairs[:, 1].contiguous()
    g = torch.randn(e, feat, device=dev, generator=gen)
    inc = P.ops.Incidence(src, dst, n)
    seg = (inc.seg_ptr[1:] - inc.seg_ptr[:-1])
    out = {"shape": name, "segments": n, "items": int(inc.seg_ptr[-1]), "longest": int(seg.max()), "feat": feat}
    for form in forms:
        P.ops.EDGE_SEGMENT["form"] = form
        for _ in range(3):
            P.ops.edge_segment_bwd(h, inc, g)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            P.ops.edge_segment_bwd(h, inc, g)
        b.record()
        torch.cuda.synchronize()
        out[form + "_ms"] = round(a.elapsed_time(b) / iters, 4)
    print(json.dumps(out))
