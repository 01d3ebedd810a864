"""
Oracle restatement of learn_nerf/train.py (loss, gradient, optax.adam).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Gradients come from torch
autograd over the torch-CPU restatement, i.e. an independent differentiator of
the same forward mathematics (the reference uses jax.grad, train.py:89-90).
"""

from typing import Callable, Dict, Optional, Tuple

import torch

from . import render as R


def default_loss_weights() -> Dict[str, float]:
    return dict(normal_mse=3e-4, neg_normal=0.1)  # train.py:187-191


def losses(coarse_fn, fine_fn, background, bbox_min, bbox_max, batch, coarse_ts, fine_ts,
           u_coarse, u_fine, loss_weights=None, extra_penalty: Optional[Callable] = None):
    """
    TrainLoop.losses (train.py:114-165). batch is [N,3,3] (origin, direction, rgb in [-1,1]).
    Returns (total_loss, loss_dict, render_out).
    """
    lw = loss_weights if loss_weights is not None else default_loss_weights()
    out = R.render_hierarchy(coarse_fn, fine_fn, background, bbox_min, bbox_max, batch[:, :2],
                             coarse_ts, fine_ts, u_coarse, u_fine)  # train.py:125-139
    targets = batch[:, 2]  # train.py:140
    coarse_loss = ((out["coarse"]["outputs"] - targets) ** 2).mean()  # train.py:141
    fine_loss = ((out["fine"]["outputs"] - targets) ** 2).mean()  # train.py:142
    loss_dict = dict(coarse=coarse_loss, fine=fine_loss)  # train.py:144
    total = coarse_loss + fine_loss  # train.py:145
    for name, v in out["coarse_aux"].items():  # train.py:146-148
        loss_dict[f"coarse_{name}"] = v
        total = total + lw[name] * v
    for name, v in out["fine_aux"].items():  # train.py:149-151
        loss_dict[f"fine_{name}"] = v
        total = total + lw[name] * v
    if extra_penalty is not None:  # train.py:153-163 (density penalty)
        total, loss_dict = extra_penalty(total, loss_dict)
    return total, loss_dict, out


def tree_norm(*tensors) -> torch.Tensor:
    """train.py:92-97: sqrt(sum of squares over all leaves)."""
    return torch.sqrt(sum((t.double() ** 2).sum() for t in tensors))


def adam_update(p, g, m, v, step: int, lr: float, b1: float = 0.9, b2: float = 0.999,
                eps: float = 1e-7):
    """
    optax.adam as used at train.py:59 (scale_by_adam with eps_root=0, then scale(-lr));
    ``step`` counts from 1.  Returns (p, m, v) new tensors.
    """
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    m_hat = m / (1 - b1 ** step)
    v_hat = v / (1 - b2 ** step)
    p = p - lr * m_hat / (torch.sqrt(v_hat) + eps)
    return p, m, v


def nerf_train_step(make_fn, coarse_flat, fine_flat, background, opt_state, step, lr,
                    bbox_min, bbox_max, batch, coarse_ts, fine_ts, u_coarse, u_fine,
                    b1=0.9, b2=0.999, eps=1e-7):
    """
    One TrainLoop.step_fn step (train.py:85-106) for models given by flat parameter vectors.
    make_fn(flat) -> model_fn(x, d).  opt_state = dict(m=[...3 tensors], v=[...]) or None.
    Returns (new_params tuple, new_opt_state, log dict, grads tuple).
    """
    params = [coarse_flat.detach().clone().requires_grad_(True),
              fine_flat.detach().clone().requires_grad_(True),
              background.detach().clone().requires_grad_(True)]
    total, loss_dict, _ = losses(make_fn(params[0]), make_fn(params[1]), params[2], bbox_min,
                                 bbox_max, batch, coarse_ts, fine_ts, u_coarse, u_fine)
    grads = torch.autograd.grad(total, params)
    log = {k: v.detach() for k, v in loss_dict.items()}
    log["grad_norm"] = tree_norm(*grads)  # train.py:99-104
    log["param_norm"] = tree_norm(*[p.detach() for p in params])
    if opt_state is None:
        opt_state = dict(m=[torch.zeros_like(p) for p in params], v=[torch.zeros_like(p) for p in params])
    new_p, new_m, new_v = [], [], []
    for p, g, m, v in zip(params, grads, opt_state["m"], opt_state["v"]):
        p2, m2, v2 = adam_update(p.detach(), g, m, v, step, lr, b1, b2, eps)
        new_p.append(p2)
        new_m.append(m2)
        new_v.append(v2)
    return tuple(new_p), dict(m=new_m, v=new_v), log, tuple(g.detach() for g in grads)
