"""train.py -- experiment driver with the command line of the reference's main.py
(same 36 flags, same run / epoch / eval / logging loop: main.py:16-305), running the
MI355X path.  `--data_path` is read like the reference's (main.py:74): when it holds the dataset
in OGB's raw on-disk layout (`<data_path>/ogbl_collab/raw/edge.csv.gz`, `split/time/train.pt`,
...: plnlp_amd/ogb_raw.py reads it without the `ogb` / PyG wheels) that data is used; nothing is
ever downloaded (no network here), so when the directory is absent `--data_name` selects a
synthetic OGB-SHAPED dataset instead (`ogbl-ddi`, `ogbl-collab`, `ogbl-citation2`;
`--data_scale` shrinks it).

    python train.py --data_name=ogbl-collab --predictor=DOT --use_valedges_as_input=True \
        --epochs=3 --runs=1 --eval_steps=1 --dropout=0.3 --gnn_num_layers=1 --grad_clip_norm=1 \
        --use_lr_decay=True --random_walk_augment=True --walk_length=10 --loss_func=WeightedHingeAUC
"""
import argparse
import os
import sys
import time

import torch

import plnlp_amd as P
from plnlp_amd import synthetic
from plnlp_amd.graph import Graph
from plnlp_amd.utils import Evaluator


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def argument(argv=None):
    """main.py:16-55 -- identical names, types and defaults (+ the synthetic-data extras at the end)"""
    p = argparse.ArgumentParser()
    for name, default in (('encoder', 'SAGE'), ('predictor', 'MLP'), ('optimizer', 'Adam'), ('loss_func', 'AUC'),
                          ('neg_sampler', 'global'), ('data_name', 'ogbl-ddi'), ('data_path', 'dataset'),
                          ('eval_metric', 'hits'), ('walk_start_type', 'edge'), ('res_dir', ''),
                          ('pretrain_emb', '')):
        p.add_argument(f'--{name}', type=str, default=default)
    for name, default in (('gnn_num_layers', 2), ('mlp_num_layers', 2), ('emb_hidden_channels', 256),
                          ('gnn_hidden_channels', 256), ('mlp_hidden_channels', 256), ('batch_size', 64 * 1024),
                          ('num_neg', 1), ('walk_length', 5), ('epochs', 500), ('log_steps', 1), ('eval_steps', 5),
                          ('runs', 10), ('year', -1), ('device', 0)):
        p.add_argument(f'--{name}', type=int, default=default)
    for name, default in (('dropout', 0.0), ('grad_clip_norm', 2.0), ('lr', 0.001)):
        p.add_argument(f'--{name}', type=float, default=default)
    for name, default in (('use_lr_decay', False), ('use_node_feats', False), ('use_coalesce', False),
                          ('train_node_emb', True), ('train_on_subgraph', False), ('use_valedges_as_input', False),
                          ('eval_last_best', False), ('random_walk_augment', False)):
        p.add_argument(f'--{name}', type=str2bool, default=default)
    p.add_argument('--data_scale', type=float, default=1.0, help='shrink the synthetic dataset')
    p.add_argument('--dp_exchange', type=str, default='auto',
                   help="under torch.distributed.run: what the ranks exchange per step (BaseModel docstring): "
                        "auto | grads | scores | shard")
    p.add_argument('--seed', type=int, default=None, help='seed torch + the dropout stream (the reference never seeds)')
    return p.parse_args(argv)


class Data:
    """what main.py uses of the PyG `data` object"""
    adj_t = None
    edge_index = None
    x = None
    num_nodes = 0


def load_dataset(args, device):
    """PygLinkPropPredDataset(name=args.data_name, root=args.data_path) + get_edge_split (main.py:74-95): read from OGB's raw
    on-disk layout when `--data_path` holds the dataset (plnlp_amd/ogb_raw.py: no ogb / PyG wheel needed), else the
    synthetic stand-in of the same shape (no network here: nothing is ever downloaded)."""
    from plnlp_amd import ogb_raw
    if ogb_raw.available(args.data_name, args.data_path):
        print(f'reading {args.data_name} from {ogb_raw.dataset_dir(args.data_name, args.data_path)}')
        return ogb_raw.read_link_dataset(args.data_name, args.data_path)
    print(f'{ogb_raw.dataset_dir(args.data_name, args.data_path)} not found: synthetic {args.data_name}-shaped data '
          f'(scale {args.data_scale})')
    return synthetic_dataset(args, device)


def synthetic_dataset(args, device):
    """Synthetic stand-in for PygLinkPropPredDataset + get_edge_split (main.py:74-95)."""
    shape = {'ogbl-ddi': 'ddi', 'ogbl-collab': 'collab', 'ogbl-citation2': 'citation2'}[args.data_name]
    g = synthetic.make_graph(shape, seed=0, device='cpu', scale=args.data_scale, weighted=(shape == 'collab'))
    n = g['num_nodes']
    gen = torch.Generator().manual_seed(20240101)
    edges = g['edges']
    perm = torch.randperm(edges.size(0), generator=gen)
    n_val = max(8, edges.size(0) // 20)
    val_e, test_e, train_e = edges[perm[:n_val]], edges[perm[n_val:2 * n_val]], edges[perm[2 * n_val:]]
    data = Data()
    data.num_nodes = n
    if shape == 'citation2':
        split = {}
        for name, e in (('train', train_e), ('valid', val_e[:2000]), ('test', test_e[:2000])):
            split[name] = {'source_node': e[:, 0], 'target_node': e[:, 1]}
            if name != 'train':
                split[name]['target_node_neg'] = torch.randint(0, n, (e.size(0), 100), generator=gen)
        data.x = torch.randn(n, 128, generator=gen)
        data.num_features = 128
        ei = torch.stack([train_e[:, 0], train_e[:, 1]])
    else:
        n_neg = max(1000, min(100000, n_val * 2))
        split = {'train': {'edge': train_e},
                 'valid': {'edge': val_e, 'edge_neg': torch.randint(0, n, (n_neg, 2), generator=gen)},
                 'test': {'edge': test_e, 'edge_neg': torch.randint(0, n, (n_neg, 2), generator=gen)}}
        if shape == 'collab':
            w = g['weight']
            split['train']['weight'], split['valid']['weight'], split['test']['weight'] = \
                w[perm[2 * n_val:]], w[perm[:n_val]], w[perm[n_val:2 * n_val]]
            split['train']['year'] = torch.randint(2000, 2018, (train_e.size(0),), generator=gen)
        ei = torch.cat([train_e.t(), train_e.flip(1).t()], dim=1)          # stored symmetric like ddi / collab
    weight = None
    if shape == 'collab':
        weight = torch.cat([split['train']['weight']] * 2)
    data.adj_t = Graph.from_edge_index(ei, weight, n)                     # T.ToSparseTensor, main.py:81
    row, col, _ = data.adj_t.coo()
    data.edge_index = torch.stack([col, row], dim=0)                       # main.py:82-83
    return data, split, n


def to_undirected(edge_index, edge_weight, num_nodes):
    """torch_geometric.utils.to_undirected(..., reduce='add'): both directions, duplicates summed"""
    r = torch.cat([edge_index[0], edge_index[1]])
    c = torch.cat([edge_index[1], edge_index[0]])
    w = torch.cat([edge_weight, edge_weight])
    key = r * num_nodes + c
    uniq, inv = torch.unique(key, return_inverse=True)
    out_w = torch.zeros(uniq.numel(), dtype=w.dtype).index_add_(0, inv, w)
    return torch.stack([uniq // num_nodes, uniq % num_nodes]), out_w


def coalesce(edge_index, edge_weight, num_nodes):
    """torch_sparse.coalesce(index, value, m, n) (main.py:140, default op 'add'): entries sorted by
    (row, col), duplicates merged with their values summed"""
    key = edge_index[0] * num_nodes + edge_index[1]
    uniq, inv = torch.unique(key, return_inverse=True)
    w = torch.zeros(uniq.numel(), dtype=edge_weight.dtype).index_add_(0, inv, edge_weight)
    return torch.stack([uniq // num_nodes, uniq % num_nodes]), w


def prepare_graph(args, data, split_edge, num_nodes):
    """main.py:109-150 -- what the driver does to the graph and the training split before the run:
    citation2 is symmetrised; collab optionally keeps the training edges from `--year` on and, with
    `--use_valedges_as_input`, feeds train + validation edges to the encoder and trains on both with
    degree-normalised pair weights.  Edits `data` and `split_edge` in place (host tensors)."""
    if args.data_name == 'ogbl-citation2':
        data.adj_t = data.adj_t.to_symmetric()                             # main.py:109-110
    if args.data_name == 'ogbl-collab':
        if args.year > 0:                                                  # main.py:112-126
            keep = (split_edge['train']['year'] >= args.year).nonzero(as_tuple=False).reshape(-1)
            for k in ('edge', 'weight', 'year'):
                split_edge['train'][k] = split_edge['train'][k][keep]
            ei, ew = to_undirected(split_edge['train']['edge'].t(), split_edge['train']['weight'], num_nodes)
            data.adj_t = Graph.from_coo(ei[0], ei[1], ew.to(torch.float32), num_nodes, num_nodes)
            data.edge_index = ei
        if args.use_valedges_as_input:                                     # main.py:128-150
            # the reference concatenates edges as [valid, train] but weights as [train, valid] (main.py:131-132)
            full_ei = torch.cat([split_edge['valid']['edge'].t(), split_edge['train']['edge'].t()], dim=-1)
            full_w = torch.cat([split_edge['train']['weight'], split_edge['valid']['weight']], dim=-1)
            ei, ew = to_undirected(full_ei, full_w, num_nodes)
            data.adj_t = Graph.from_coo(ei[0], ei[1], ew.to(torch.float32), num_nodes, num_nodes)
            data.edge_index = ei
            if args.use_coalesce:                                          # main.py:139-140
                full_ei, full_w = coalesce(full_ei, full_w, num_nodes)
            split_edge['train']['edge'] = full_ei.t()
            deg = data.adj_t.sum(dim=1).to(torch.float)
            dis = deg.pow(-0.5)
            dis[dis == float('inf')] = 0
            split_edge['train']['weight'] = dis[full_ei[0]] * full_w * dis[full_ei[1]]


def main(argv=None, dataset=None, hooks=None):
    """dataset: (data, split_edge, num_nodes) on the host instead of load_dataset's synthetic stand-in (tests hand a small
    problem in).  hooks: optional callables -- on_run_start(run, model) right after model.param_init() (main.py:236),
    on_epoch(run, epoch, model, loss) after the epoch's train / eval -- the seams the driver parity test uses to give the
    CPU oracle the same initial weights and to read the per-epoch losses."""
    args = argument(argv)
    hooks = hooks or {}
    group = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:      # one process per GPU (torch.distributed.run)
        args.device = int(os.environ.get('LOCAL_RANK', '0'))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    device = torch.device(f'cuda:{args.device}')
    torch.cuda.set_device(device)      # kernels launch on the current device's stream
    from plnlp_amd.utils import limit_host_threads
    limit_host_threads(int(os.environ.get('WORLD_SIZE', '1')))      # stay inside the container's CPU quota
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        torch.distributed.init_process_group('nccl', device_id=device)
        group = torch.distributed.group.WORLD
        if args.seed is None:          # every rank must draw the same negatives / permutations
            args.seed = 0
    if args.seed is not None:
        torch.manual_seed(args.seed)
        P.manual_seed(args.seed)
    data, split_edge, num_nodes = dataset if dataset is not None else load_dataset(args, device)
    num_node_feats = getattr(data, 'num_features', 0) if data.x is not None else 0
    # under torch.distributed.run every rank runs this script: rank 0 alone prints and owns the log file (the
    # others would interleave duplicates, or split across files whose names differ by a second).  Evaluation still
    # runs on every rank -- test() draws from the CPU generator like the reference's loaders do, and the ranks'
    # streams must stay in step for the next epoch's negatives and permutations.
    is_main = int(os.environ.get('RANK', '0')) == 0
    _print = print if is_main else (lambda *a, **k: None)

    class _Log:
        """`with open(log_file, 'a') as f` on rank 0, a sink elsewhere"""
        def __enter__(self):
            self.f = open(log_file, 'a') if is_main else open(os.devnull, 'w')
            return self.f

        def __exit__(self, *exc):
            self.f.close()
            return False

    _print(args)
    if is_main:
        os.makedirs(args.res_dir or '.', exist_ok=True)
    log_file = os.path.join(args.res_dir, 'log_' + args.data_name + '_' + str(int(time.time())) + '.txt')
    with _Log() as f:
        f.write(str(args) + '\n')

    prepare_graph(args, data, split_edge, num_nodes)

    data.adj_t = data.adj_t.to(device)                                     # main.py:175
    if data.x is not None:
        data.x = data.x.to(torch.float).to(device)
    if args.encoder.upper() == 'GCN':
        data.adj_t = P.gcn_normalization(data.adj_t)                       # main.py:177-179
    if args.encoder.upper() == 'WSAGE':
        data.adj_t = P.adj_normalization(data.adj_t)

    model = P.BaseModel(
        lr=args.lr, dropout=args.dropout, grad_clip_norm=args.grad_clip_norm, gnn_num_layers=args.gnn_num_layers,
        mlp_num_layers=args.mlp_num_layers, emb_hidden_channels=args.emb_hidden_channels,
        gnn_hidden_channels=args.gnn_hidden_channels, mlp_hidden_channels=args.mlp_hidden_channels,
        num_nodes=num_nodes, num_node_feats=num_node_feats, gnn_encoder_name=args.encoder,
        predictor_name=args.predictor, loss_func=args.loss_func, optimizer_name=args.optimizer, device=device,
        use_node_feats=args.use_node_feats, train_node_emb=args.train_node_emb, pretrain_emb=args.pretrain_emb,
        process_group=group, dp_scaling='strong', dp_exchange=args.dp_exchange)
    total_params = sum(p.numel() for p in model.para_list)
    msg = f'Total number of model parameters is {total_params}'
    _print(msg)
    with _Log() as f:
        f.write(msg + '\n')

    evaluator = Evaluator(name=args.data_name)
    keys = ['Hits@20', 'Hits@50', 'Hits@100'] if args.eval_metric == 'hits' else ['MRR']
    loggers = {k: P.Logger(args.runs, args) for k in keys}

    if args.random_walk_augment:                                           # main.py:228-233
        rw_graph = data.adj_t
        if args.walk_start_type == 'edge':
            rw_start = split_edge['train']['edge'].reshape(-1).to(device)
        else:
            rw_start = torch.arange(0, num_nodes, dtype=torch.long, device=device)
        rw_seed = args.seed if args.seed is not None else int(time.time())

    for run in range(args.runs):
        model.param_init()
        if 'on_run_start' in hooks:
            hooks['on_run_start'](run, model)
        start_time = time.time()
        cur_lr = args.lr
        for epoch in range(1, 1 + args.epochs):
            if args.random_walk_augment:                                   # main.py:241-253
                rw_seed += 1
                pairs, weights = P.ops.random_walk_pairs(rw_graph, rw_start, args.walk_length, rw_seed)
                split_edge['train']['edge'] = pairs.cpu()
                split_edge['train']['weight'] = weights.cpu()
            loss = model.train(data, split_edge, batch_size=args.batch_size, neg_sampler_name=args.neg_sampler,
                               num_neg=args.num_neg)
            if epoch % args.eval_steps == 0:
                results = model.test(data, split_edge, batch_size=args.batch_size, evaluator=evaluator,
                                     eval_metric=args.eval_metric)
                for key, result in results.items():
                    loggers[key].add_result(run, result)
                if epoch % args.log_steps == 0:
                    spent_time = time.time() - start_time
                    for key, (valid_res, test_res) in results.items():
                        line = (f'Run: {run + 1:02d}, Epoch: {epoch:02d}, Loss: {loss:.4f}, '
                                f'Learning Rate: {cur_lr:.4f}, Valid: {100 * valid_res:.2f}%, '
                                f'Test: {100 * test_res:.2f}%')
                        _print(key)
                        _print(line)
                        with _Log() as f:
                            print(key, file=f)
                            print(line, file=f)
                    _print('---')
                    _print(f'Training Time Per Epoch: {spent_time / args.eval_steps: .4f} s')
                    _print('---')
                    start_time = time.time()
            if 'on_epoch' in hooks:
                hooks['on_epoch'](run, epoch, model, loss)
            if args.use_lr_decay:
                cur_lr = P.adjust_lr(model.optimizer, epoch / args.epochs, args.lr)
        for key in loggers:
            _print(key)
            if is_main:
                loggers[key].print_statistics(run, f=sys.stdout, last_best=args.eval_last_best)
            with _Log() as f:
                print(key, file=f)
                loggers[key].print_statistics(run, f=f, last_best=args.eval_last_best)
    for key in loggers:
        _print(key)
        if is_main:
            loggers[key].print_statistics(f=sys.stdout, last_best=args.eval_last_best)
        with _Log() as f:
            print(key, file=f)
            loggers[key].print_statistics(f=f, last_best=args.eval_last_best)
    return loggers


if __name__ == "__main__":
    main()
