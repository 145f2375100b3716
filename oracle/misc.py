"""Oracles for the small rows of SURVEY.md 8a: H10 bag building, H14 MLP/Linear heads, H15 Cox losses,
H20 vary_precision.  Each is pinned by a fixture under tests/golden/ captured from the imported reference.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ---- H20: reference src/stamp/modeling/transforms.py:5-29 (integer semantics -> numpy, bit exact) -----------
def vary_precision_bits(bits: np.ndarray, shifts: np.ndarray) -> np.ndarray:
    """bits: the float tensor viewed as int16/int32; shifts: per-element number of low mantissa bits to clear.
    mask = ~0 << shift ; out = bits & mask   (transforms.py:26-28)."""
    mask = (np.array(-1, dtype=bits.dtype) << shifts.astype(bits.dtype)).astype(bits.dtype)
    return bits & mask


def vary_precision_shift_range(dtype: str, min_fraction_bits: int) -> int:
    """exclusive upper bound of torch.randint(0, fraction_bits - min_fraction_bits) (transforms.py:13-25)."""
    frac = {"f32": 23, "f16": 10, "bf16": 7}[dtype]
    return frac - min_fraction_bits


# ---- H10: reference src/stamp/modeling/data.py:811-862 ------------------------------------------------------
def fixed_size_bag_indices(n_tiles: int, bag_size: int, deterministic: bool, generator: torch.Generator | None = None):
    if n_tiles <= bag_size:
        return torch.arange(n_tiles)
    if deterministic:
        return torch.linspace(0, n_tiles - 1, steps=bag_size).round().long()      # data.py:825-827
    return torch.randperm(n_tiles, generator=generator)[:bag_size]               # data.py:829


def to_fixed_size_bag(bag, coords, bag_size, deterministic=False, generator=None):
    idx = fixed_size_bag_indices(bag.shape[0], bag_size, deterministic, generator)
    b, c = bag[idx], coords[idx]
    pad = bag_size - b.shape[0]
    if pad > 0:                                                                    # zero-pad to the right
        b = torch.cat([b, b.new_zeros(pad, b.shape[1])])
        c = torch.cat([c, c.new_zeros(pad, c.shape[1])])
    return b, c, min(bag_size, bag.shape[0])


# ---- H14: reference src/stamp/modeling/models/mlp.py:35-44, 53-62 -------------------------------------------
def mlp_forward(x, sd, n_linear: int):
    if x.dim() == 3:
        x = x.mean(dim=1)
    for i in range(n_linear):
        x = F.linear(x, sd[f"mlp.{3 * i}.weight"].float(), sd[f"mlp.{3 * i}.bias"].float())
        if i < n_linear - 1:
            x = F.relu(x)
    return x


def linear_forward(x, sd):
    if x.dim() == 3:
        x = x.mean(dim=1)
    return F.linear(x, sd["fc.weight"].float(), sd["fc.bias"].float())


# ---- H15: reference src/stamp/modeling/models/cox.py:107-270 (Efron / Breslow / no ties) ---------------------
def cox_neg_partial_log_likelihood(log_hz, time, event, ties_method="efron", reduction="mean"):
    order = torch.argsort(time)
    t, lh, ev = time[order], log_hz.reshape(-1)[order].double(), event[order].bool()
    uniq = torch.unique(t)
    if len(uniq) == len(t):                                   # no ties: cox.py:20-34
        log_den = torch.logcumsumexp(lh.flip(0), dim=0).flip(0)
        pll = (lh - log_den)[ev]
    elif ties_method == "breslow":                            # cox.py:82-104
        log_den = torch.stack([torch.logsumexp(lh[t >= t[i]], dim=0) for i in range(len(t))])
        pll = (lh - log_den)[ev]
    else:                                                     # efron, cox.py:37-79
        terms = []
        for u in uniq:
            Hs = (t == u) & ev
            m = int(Hs.sum())
            if m == 0:
                continue
            naive = torch.exp(lh[t >= u]).sum()
            ties = torch.exp(lh[Hs]).sum()
            den = sum(torch.log(naive - (s / m) * ties) for s in range(m))
            terms.append(lh[Hs].sum() - den)
        pll = torch.stack(terms)
    loss = -pll
    return (loss.sum() if reduction == "sum" else loss.nanmean()).float()


# ---- H15: reference src/stamp/modeling/models/__init__.py:625-659 (`cox_loss`, the slide / patient-level survival objective) ----
def cox_breslow_slide_loss(scores, times, events):
    """Breslow negative partial log-likelihood as the reference's slide-level Lit class states it: event i's risk set is every j with
    time_j >= time_i; mean over the events; a batch without events gives `scores.sum() * 0` (a zero that keeps the graph)."""
    scores, times, ev = scores.flatten(), times.flatten(), events.bool().flatten()
    if not ev.any():
        return scores.sum() * 0.0
    risk = times[ev][:, None] <= times[None, :]                                   # [events, N]
    mx = scores.max()
    lse = torch.log((risk * torch.exp(scores - mx)).sum(dim=1)) + mx
    return -(scores[ev] - lse).mean()


def keep_image_head(feats: torch.Tensor, sd: dict) -> torch.Tensor:
    """KEEP's `encode_image` after the trunk (reference src/stamp/preprocessing/extractor/keep.py:38-47): normalize(Linear(GELU(Linear(feats)))), F.normalize's
    x / max(||x||, 1e-12).  Pinned by tests/golden/keep_head.npz (the reference's own class with an identity trunk)."""
    import torch.nn.functional as F
    h = F.gelu(F.linear(feats, sd["visual_head.0.weight"], sd["visual_head.0.bias"]))
    return F.normalize(F.linear(h, sd["visual_head.2.weight"], sd["visual_head.2.bias"]), dim=-1)
