"""`PygLinkPropPredDataset(name, root)[0]` + `get_edge_split()` (main.py:74-75, 95) read straight from OGB's on-disk
layout -- without the `ogb` / `torch_geometric` wheels, which this image does not have.

What `ogb` 1.3.2 (README.md:18) leaves under `<root>/<name with '-' -> '_'>/` after a download  [3P-recall: the package
is absent from /root/reference; its reader is restated from its published layout, and pinned by the round trip in
tests/test_host_logic.py, not by the reference]:

    raw/edge.csv.gz              E x 2 integers, no header: one row per edge (source, target)
    raw/num-node-list.csv.gz     one integer: the node count           raw/num-edge-list.csv.gz   one integer: E
    raw/node-feat.csv.gz         N x F floats       (ogbl-citation2: 128 columns; absent for ddi / collab)
    raw/edge_weight.csv.gz, raw/edge_year.csv.gz    E x 1 each          (ogbl-collab's "additional edge files")
    raw/node_year.csv.gz         N x 1                                  (ogbl-citation2's "additional node file")
    split/<type>/{train,valid,test}.pt              torch-saved dicts of numpy arrays (or tensors):
        ddi (type `target`), collab (`time`):  'edge' [n,2]; valid / test also 'edge_neg' [m,2]; collab adds 'weight', 'year'
        citation2 (`time`):                    'source_node' [n], 'target_node' [n]; valid / test 'target_node_neg' [n,1000]
    (`.npz` files with the same keys are accepted in place of `.pt`.)

ddi and collab are undirected and stored once per pair: the loader ADDS the inverse edges (`add_inverse_edge=True` in their
`master.csv` rows), duplicating the per-edge attributes; citation2 is directed and kept as stored (main.py:109-110
symmetrises it later).  The result is what main.py holds after line 95, with `T.ToSparseTensor()` (main.py:81) applied:
`data.adj_t` = the transposed adjacency with `edge_weight` as its values (collab), `data.edge_index` rebuilt from its
coordinates (main.py:82-83), `data.x`, `data.num_nodes`, `data.num_features`, and the split dictionary of tensors."""
from __future__ import annotations

import gzip
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .graph import Graph

# name -> (undirected: add the inverse edges, split type directory)
DATASETS = {"ogbl-ddi": (True, "target"), "ogbl-collab": (True, "time"), "ogbl-citation2": (False, "time")}


class Data:
    """what main.py uses of the PyG `data` object"""
    adj_t = None
    edge_index = None
    x = None
    num_nodes = 0
    num_features = 0


def dataset_dir(name: str, root: str) -> str:
    return os.path.join(root, name.replace("-", "_"))


def available(name: str, root: str) -> bool:
    """is there a raw OGB copy of this dataset under root?"""
    d = dataset_dir(name, root)
    return name in DATASETS and os.path.exists(os.path.join(d, "raw", "edge.csv.gz")) and os.path.isdir(os.path.join(d, "split"))


def _read_csv(path: str, dtype) -> np.ndarray:
    """a header-less numeric csv(.gz) as a 2-d array"""
    try:
        import pandas as pd
        return pd.read_csv(path, compression="gzip" if path.endswith(".gz") else None, header=None).values.astype(dtype)
    except ImportError:
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rt") as f:
            return np.loadtxt(f, delimiter=",", dtype=dtype, ndmin=2)


def _read_split_file(stem: str) -> Dict[str, torch.Tensor]:
    if os.path.exists(stem + ".pt"):
        # ogb's split files are pickled dicts of numpy arrays: not loadable as "weights only"; they are local files the
        # user downloaded -- the same trust the reference's own torch.load of them places in them
        d = torch.load(stem + ".pt", weights_only=False)
    elif os.path.exists(stem + ".npz"):
        d = dict(np.load(stem + ".npz"))
    else:
        raise FileNotFoundError(stem + ".pt / .npz")
    out = {}
    for k, v in d.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t
    return out


def read_link_dataset(name: str, root: str) -> Tuple[Data, Dict[str, Dict[str, torch.Tensor]], int]:
    """(data, split_edge, num_nodes) for main.py:74-95; raises FileNotFoundError when the directory is not there."""
    if name not in DATASETS:
        raise ValueError(f"unknown dataset {name!r}: one of {sorted(DATASETS)}")
    d = dataset_dir(name, root)
    raw = os.path.join(d, "raw")
    if not os.path.exists(os.path.join(raw, "edge.csv.gz")):
        raise FileNotFoundError(os.path.join(raw, "edge.csv.gz"))
    undirected, split_type = DATASETS[name]
    edge = torch.from_numpy(_read_csv(os.path.join(raw, "edge.csv.gz"), np.int64)).t().contiguous()      # [2, E]
    nn_path = os.path.join(raw, "num-node-list.csv.gz")
    num_nodes = int(_read_csv(nn_path, np.int64)[0, 0]) if os.path.exists(nn_path) else int(edge.max()) + 1

    def optional(fname, dtype) -> Optional[torch.Tensor]:
        p = os.path.join(raw, fname)
        return torch.from_numpy(_read_csv(p, dtype)) if os.path.exists(p) else None
    edge_weight = optional("edge_weight.csv.gz", np.float32)
    edge_year = optional("edge_year.csv.gz", np.int64)
    x = optional("node-feat.csv.gz", np.float32)
    if undirected:                                    # read_graph_pyg(add_inverse_edge=True): [edges | reversed edges]
        edge = torch.cat([edge, edge.flip(0)], dim=1)
        if edge_weight is not None:
            edge_weight = torch.cat([edge_weight, edge_weight], dim=0)
        if edge_year is not None:
            edge_year = torch.cat([edge_year, edge_year], dim=0)
    data = Data()
    data.num_nodes = num_nodes
    if x is not None:
        data.x = x.to(torch.float)                    # main.py:104-106
        data.num_features = x.shape[1]
    if edge_year is not None:
        data.edge_year = edge_year                    # (main.py:114 only asks whether the attribute exists)
    w = None if edge_weight is None else edge_weight.view(-1).to(torch.float)      # main.py:77-79
    data.adj_t = Graph.from_edge_index(edge, w, num_nodes)                         # main.py:81
    row, col, _ = data.adj_t.coo()
    data.edge_index = torch.stack([col, row], dim=0)                               # main.py:82-83
    split = {s: _read_split_file(os.path.join(d, "split", split_type, s)) for s in ("train", "valid", "test")}
    return data, split, num_nodes


def write_link_dataset(name: str, root: str, edge: torch.Tensor, num_nodes: int, split: Dict[str, Dict[str, torch.Tensor]],
                       edge_weight: Optional[torch.Tensor] = None, edge_year: Optional[torch.Tensor] = None,
                       x: Optional[torch.Tensor] = None, split_ext: str = "pt") -> str:
    """write a dataset in the layout above (tests; converting one's own graph): `edge` [E, 2] stored ONCE per pair for the
    undirected datasets.  Returns the dataset directory."""
    if name not in DATASETS:
        raise ValueError(name)
    d = dataset_dir(name, root)
    raw = os.path.join(d, "raw")
    os.makedirs(raw, exist_ok=True)

    def put(fname, arr, fmt):
        with gzip.open(os.path.join(raw, fname), "wt") as f:
            np.savetxt(f, np.asarray(arr), delimiter=",", fmt=fmt)
    put("edge.csv.gz", edge.numpy(), "%d")
    put("num-node-list.csv.gz", np.array([[num_nodes]]), "%d")
    put("num-edge-list.csv.gz", np.array([[edge.shape[0]]]), "%d")
    if edge_weight is not None:
        put("edge_weight.csv.gz", edge_weight.reshape(-1, 1).numpy(), "%.9g")
    if edge_year is not None:
        put("edge_year.csv.gz", edge_year.reshape(-1, 1).numpy(), "%d")
    if x is not None:
        put("node-feat.csv.gz", x.numpy(), "%.9g")
    sdir = os.path.join(d, "split", DATASETS[name][1])
    os.makedirs(sdir, exist_ok=True)
    for s, part in split.items():
        arrays = {k: v.numpy() for k, v in part.items()}
        if split_ext == "pt":
            torch.save(arrays, os.path.join(sdir, s + ".pt"))
        else:
            np.savez(os.path.join(sdir, s + ".npz"), **arrays)
    return d
