"""Losses of the reference's Lightning steps, evaluated on the [batch, outputs] predictions of the HIP heads.

These tensors hold at most a few hundred scalars: they stay torch ops on purpose -- the trainer feeds the
HIP logits in, takes d(loss)/d(logits) out of autograd and runs everything with a token dimension in libamdstamp.

* classification: weighted cross-entropy on float one-hot targets (reference src/stamp/modeling/models/__init__.py:254-258)
* regression: L1 (`LitBaseRegressor._compute_loss`, :420-422)
* survival: Cox negative partial log-likelihood with Efron's tie handling (`neg_partial_log_likelihood`, src/stamp/modeling/
  models/cox.py:107-270, used by `LitTileSurvival.training_step`, models/__init__.py:751-776; targets are [time, event]);
  slide / patient-level survival uses the Breslow form `cox_loss` (models/__init__.py:625-659, `LitSlideSurvival.training_step` :812-830)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def weighted_cross_entropy(logits: torch.Tensor, onehot: torch.Tensor, class_weights: torch.Tensor | None = None) -> torch.Tensor:
    return F.cross_entropy(logits, onehot.to(logits), weight=None if class_weights is None else class_weights.to(logits))


def l1_loss(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    return F.l1_loss(preds, targets.to(preds).reshape(preds.shape))


def neg_partial_log_likelihood(log_hz: torch.Tensor, time: torch.Tensor, event: torch.Tensor, ties_method: str = "efron",
                               reduction: str = "mean") -> torch.Tensor:
    """Cox partial likelihood; `log_hz` [B] or [B,1] (differentiable), `time` [B], `event` [B] (bool / 0-1)."""
    if event.sum().item() == 0 or log_hz.dim() == 0:       # cox.py:219-224: "No events OR single sample. Returning zero loss"
        return torch.tensor(0.0, requires_grad=True, device=log_hz.device)
    lh = log_hz.reshape(-1)
    time = time.to(lh.device)
    order = torch.argsort(time)
    t, lh, ev = time[order], lh[order], event.to(lh.device)[order].bool()
    uniq = torch.unique(t)
    if uniq.numel() == t.numel():                          # no ties (cox.py:20-34): denominators by a reversed log-cum-sum-exp
        log_den = torch.logcumsumexp(lh.flip(0), dim=0).flip(0)
        pll = (lh - log_den)[ev]
    elif ties_method == "breslow":                         # cox.py:82-104
        log_den = torch.stack([torch.logsumexp(lh[t >= t[i]], dim=0) for i in range(t.numel())])
        pll = (lh - log_den)[ev]
    elif ties_method == "efron":                           # cox.py:37-79
        terms = []
        for u in uniq:
            tied = (t == u) & ev
            m = int(tied.sum())
            if m == 0:
                continue
            at_risk = torch.exp(lh[t >= u]).sum()
            tied_sum = torch.exp(lh[tied]).sum()
            den = sum(torch.log(at_risk - (s / m) * tied_sum) for s in range(m))
            terms.append(lh[tied].sum() - den)
        pll = torch.stack(terms) if terms else lh.new_zeros(0)
    else:
        raise ValueError(f"unknown ties_method {ties_method!r}")
    loss = -pll
    return loss.sum() if reduction == "sum" else loss.nanmean()


def cox_survival_loss(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """`LitTileSurvival.training_step` (models/__init__.py:759-766): targets[:, 0] = time, targets[:, 1] = event."""
    y = targets.to(preds.device, torch.float32)
    return neg_partial_log_likelihood(preds.squeeze(-1), y[:, 0], y[:, 1])


def cox_breslow_loss(scores: torch.Tensor, times: torch.Tensor, events: torch.Tensor) -> torch.Tensor:
    """The reference's `cox_loss` (models/__init__.py:625-659): Breslow risk sets {j: time_j >= time_i} per event i, max-shifted
    log-sum-exp, MEAN over the events; no event in the batch -> `scores.sum() * 0.0` (zero loss that keeps the graph)."""
    scores, times, ev = scores.flatten(), times.to(scores.device).flatten(), events.to(scores.device).bool().flatten()
    if not bool(ev.any()):
        return scores.sum() * 0.0
    risk = times[ev][:, None] <= times[None, :]
    mx = scores.max()
    lse = torch.log((risk * torch.exp(scores - mx)).sum(dim=1)) + mx
    return -(scores[ev] - lse).mean()


def cox_slide_survival_loss(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """`LitSlideSurvival.training_step` (models/__init__.py:812-830): targets[:, 0] = time, targets[:, 1] = event, Breslow."""
    y = targets.to(preds.device, torch.float32)
    return cox_breslow_loss(preds, y[:, 0], y[:, 1])
