import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plnlp_amd import synthetic
dev = torch.device("cuda")
g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
n = g["num_nodes"]; B = 65536
gen = torch.Generator(device=dev).manual_seed(777)
starts = g["edges"].reshape(-1)
pairs, w = synthetic.random_walk_pairs(g["adj_t"], starts, 10, gen)
sel = torch.randperm(pairs.size(0), generator=gen, device=dev)[:B * 4]
for i in range(4):
    p = pairs[sel[i*B:(i+1)*B]]
    neg = torch.randint(0, n, (B, 2), generator=gen, device=dev)
    nodes = torch.unique(torch.cat([p.reshape(-1), neg.reshape(-1)]))
    print("batch", i, "distinct nodes touched", nodes.numel(), "of", n, f"= {nodes.numel()/n:.2%}", " (pos-only:", torch.unique(p).numel(), ")")
# train-edge positives instead of RW pairs
e = g["edges"][torch.randperm(g["edges"].size(0), generator=gen, device=dev)[:B]]
print("plain train edges: distinct", torch.unique(torch.cat([e.reshape(-1), neg.reshape(-1)])).numel())
