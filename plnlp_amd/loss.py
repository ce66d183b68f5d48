"""Pairwise ranking losses with the names and call signatures of the reference's
plnlp/loss.py -- `f(pos_out, neg_out, num_neg[, weight]) -> 0-d tensor`.

The seven kinds built on d = pos[b] - neg[b,n] run in ONE HIP kernel that
produces the loss and both gradients (plnlp_pairwise_loss_f32); the reference
needs 5-8 element-wise launches plus an autograd graph for the same thing.
`ce_loss` and `info_nce_loss` (unused by any README recipe)
stay on stock device ops.
"""
import torch

from .ops import PairwiseLossFn, PairwiseLossJointFn


def _fused(kind):
    def needs_weight(pos_out, neg_out, num_neg, weight):
        return PairwiseLossFn.apply(pos_out, neg_out, weight, kind, int(num_neg))

    def plain(pos_out, neg_out, num_neg):
        return PairwiseLossFn.apply(pos_out, neg_out, None, kind, int(num_neg))

    fn = needs_weight if kind.startswith(("weighted", "adaptive")) else plain
    fn.__name__ = f"{kind}_loss"
    fn.kind = kind
    return fn


auc_loss = _fused("auc")                                  # loss.py:5-8    sum (1 - d)^2
hinge_auc_loss = _fused("hinge_auc")                      # loss.py:11-14  sum max(1 - d, 0)^2
weighted_auc_loss = _fused("weighted_auc")                # loss.py:17-21  sum w (1 - d)^2
adaptive_auc_loss = _fused("adaptive_auc")                # loss.py:24-28  sum (w - d)^2
weighted_hinge_auc_loss = _fused("weighted_hinge_auc")    # loss.py:31-35  sum w max(w - d, 0)^2
adaptive_hinge_auc_loss = _fused("adaptive_hinge_auc")    # loss.py:38-42  sum max(w - d, 0)^2
log_rank_loss = _fused("log_rank")                        # loss.py:45-48  mean -log(sigmoid(d) + 1e-15)


def ce_loss(pos_out, neg_out):
    """loss.py:51-54"""
    eps = 1e-15
    return (-(torch.sigmoid(pos_out) + eps).log().mean()
            - (1 - torch.sigmoid(neg_out) + eps).log().mean())


def info_nce_loss(pos_out, neg_out, num_neg):
    """loss.py:57-62"""
    p = pos_out.reshape(-1, 1).exp()
    n = neg_out.reshape(-1, num_neg).exp().sum(dim=1, keepdim=True)
    return -((p / (p + n)) + 1e-15).log().mean()


def joint_loss(name, out, n_pos, num_neg, weight=None):
    """loss `name` (CLI name, with the reference's fallbacks of model.py:107-126) on one score tensor
    [pos (n_pos) | neg (n_pos * num_neg)]; None when that loss has no fused kernel"""
    fn, weighted = BY_NAME.get(name, (auc_loss, False))
    if weighted and weight is None:
        fn, weighted = auc_loss, False
    kind = getattr(fn, "kind", None)
    if kind is None:
        return None
    return PairwiseLossJointFn.apply(out, int(n_pos), weight if weighted else None, kind, int(num_neg))


# CLI name -> (function, takes the per-edge weight/margin)   model.py:107-126
BY_NAME = {
    "CE": (ce_loss, False), "InfoNCE": (info_nce_loss, False), "LogRank": (log_rank_loss, False),
    "HingeAUC": (hinge_auc_loss, False), "AdaAUC": (adaptive_auc_loss, True),
    "WeightedAUC": (weighted_auc_loss, True), "AdaHingeAUC": (adaptive_hinge_auc_loss, True),
    "WeightedHingeAUC": (weighted_hinge_auc_loss, True),
}
