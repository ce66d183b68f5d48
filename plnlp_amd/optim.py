"""Optimiser step of the hot loop (plnlp/model.py:163-167) as HIP kernels:
per-group gradient-norm clipping folded into one fused Adam/AdamW update per
tensor.  The clip coefficient is computed on the device from the squared norm
(no host synchronisation, unlike clip_grad_norm_ + optimizer.step())."""
from typing import Dict, Iterable, Optional, Sequence, Tuple

import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW (amsgrad=False) semantics; `param_groups` is kept
    so `adjust_lr` (model.py:279-286) works unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                      decoupled=decoupled))

    @torch.no_grad()
    def step(self, clip: Optional[Dict[int, Tuple[torch.Tensor, float]]] = None, grad_scale: float = 1.0):
        """clip: id(param) -> (squared-norm device scalar of the param's clip group, max_norm)"""
        for group in self.param_groups:
            b1, b2 = group["betas"]
            entries, copy_back = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                # a table kept padded to 16-byte rows (ops.padded_base) is stepped in that layout, gradient and moments
                # too: its pad columns hold zeros and a zero gradient, which Adam leaves where they are
                pd, gd = ops.padded_base(p.data), ops.padded_base(p.grad)
                padded = pd is not None and gd is not None and pd.shape == gd.shape
                if not st:
                    st["step"] = 0
                    like = pd if padded else p
                    st["exp_avg"] = torch.zeros_like(like, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(like, memory_format=torch.contiguous_format)
                st["step"] += 1
                sq, max_norm = (clip or {}).get(id(p), (None, 0.0))
                if padded and st["exp_avg"].shape == pd.shape:
                    entries.append((pd, gd, st["exp_avg"], st["exp_avg_sq"], st["step"], sq, max_norm))
                    continue
                if st["exp_avg"].shape != p.shape:          # (moments in the padded layout, this gradient is not)
                    pd = ops.padded_base(p.data)
                    gd = torch.zeros_like(pd)
                    gd[:, :p.shape[1]].copy_(p.grad)
                    entries.append((pd, gd, st["exp_avg"], st["exp_avg_sq"], st["step"], sq, max_norm))
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.data.is_contiguous():              # a strided parameter with a plain gradient: step a copy
                    pc = p.data.contiguous()
                    entries.append((pc, g, st["exp_avg"], st["exp_avg_sq"], st["step"], sq, max_norm))
                    copy_back.append((p, pc))
                    continue
                entries.append((p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], sq, max_norm))
            # all tensors of the group in one launch (per 16): the handful of small encoder / predictor
            # weights next to the embedding table would otherwise cost a launch each
            ops.adam_multi(entries, lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                           weight_decay=group["weight_decay"], decoupled=group["decoupled"],
                           grad_scale=grad_scale)
            for p, pc in copy_back:
                p.data.copy_(pc)


def fused_adam_state(opt: "FusedAdam", p: torch.nn.Parameter, padded: bool = False) -> Optional[dict]:
    """Adam state and hyper-parameters of `p` for an update applied OUTSIDE opt.step() (ops.GradSink.adam: the
    PLNLP_EPI_ADAM epilogue).  None when the group's settings are not plain Adam.  The caller advances
    state['step'] itself once the update has really been applied.
    padded: `p` is a table kept padded (ops.padded_base) and is stepped in that layout -- fresh moments take the padded shape,
    as FusedAdam.step gives them; None when moments of the unpadded shape already exist."""
    for group in opt.param_groups:
        if any(q is p for q in group["params"]):
            if group["weight_decay"] != 0.0 or group["decoupled"]:
                return None
            st = opt.state[p]
            like = ops.padded_base(p.data) if padded else p
            if like is None:
                return None
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(like, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(like, memory_format=torch.contiguous_format)
            if st["exp_avg"].shape != like.shape:
                return None
            return dict(param=p, exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], step=st["step"] + 1,
                        lr=group["lr"], betas=group["betas"], eps=group["eps"])
    return None


def group_sqnorm(params: Sequence[torch.nn.Parameter]) -> Optional[torch.Tensor]:
    """squared total gradient norm of a clip group as a device scalar"""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return None
    grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
    out = torch.empty(1, dtype=torch.float32, device=grads[0].device)
    return ops.sqnorm_into(grads, out)
