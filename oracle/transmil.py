"""Oracle for TransMIL.

Restates reference src/stamp/modeling/models/trans_mil.py: TransMIL.forward :299-326, Transformer.forward
:258-263, NystromAttention.forward :81-163 (mask=None path), moore_penrose_iter_pinv :23-37, PPEG.forward :274-283
-- eval mode by default (Dropout(0.1) on to_out is the identity); `drop` supplies the train-mode multiplier masks of that one
dropout site (:63-66, dropout=0.1 from :255).  Pinned by tests/golden/transmil_*.npz.  Test infrastructure only.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def pinv_iter(x: torch.Tensor, iters: int = 6) -> torch.Tensor:
    ax = x.abs()
    col, row = ax.sum(dim=-1), ax.sum(dim=-2)
    z = x.transpose(-1, -2) / (col.max() * row.max())       # GLOBAL max over batch and heads (:28)
    eye = torch.eye(x.shape[-1], dtype=x.dtype).unsqueeze(0)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * eye - (xz @ (15 * eye - (xz @ (7 * eye - xz)))))
    return z


def nystrom_attention(x, sd, pre, *, heads=8, landmarks, iters=6, conv_k=33, drop=None):
    b, n, dim = x.shape
    m = landmarks
    rem = n % m
    if rem > 0:
        x = F.pad(x, (0, 0, m - rem, 0), value=0.0)          # FRONT padding (:100)
    q, k, v = F.linear(x, sd[pre + "to_qkv.weight"]).chunk(3, dim=-1)
    hd = q.shape[-1] // heads
    q, k, v = (t.reshape(b, -1, heads, hd).transpose(1, 2) for t in (q, k, v))
    q = q * hd ** -0.5
    l = math.ceil(n / m)
    ql = q.reshape(b, heads, -1, l, hd).sum(dim=3) / l       # segment SUM / l (:114-124)
    kl = k.reshape(b, heads, -1, l, hd).sum(dim=3) / l
    a1 = torch.softmax(q @ kl.transpose(-1, -2), dim=-1)
    a2 = torch.softmax(ql @ kl.transpose(-1, -2), dim=-1)
    a3 = torch.softmax(ql @ k.transpose(-1, -2), dim=-1)
    out = (a1 @ pinv_iter(a2, iters)) @ (a3 @ v)
    out = out + F.conv2d(v, sd[pre + "res_conv.weight"], None, padding=(conv_k // 2, 0), groups=heads)
    out = out.transpose(1, 2).reshape(b, -1, heads * hd)
    out = F.linear(out, sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"])[:, -n:]
    return out if drop is None else out * drop          # Dropout acts elementwise: masking before or after the slice is the same


def ppeg(x, sd, pre, H, W):
    B, _, C = x.shape
    cls, feat = x[:, :1], x[:, 1:]
    f = feat.transpose(1, 2).reshape(B, C, H, W)
    y = (F.conv2d(f, sd[pre + "proj.weight"], sd[pre + "proj.bias"], padding=3, groups=C) + f
         + F.conv2d(f, sd[pre + "proj1.weight"], sd[pre + "proj1.bias"], padding=2, groups=C)
         + F.conv2d(f, sd[pre + "proj2.weight"], sd[pre + "proj2.bias"], padding=1, groups=C))
    return torch.cat([cls, y.flatten(2).transpose(1, 2)], dim=1)


def transmil_forward(bags: torch.Tensor, sd: dict, drop: dict | None = None, dtype=torch.float32) -> torch.Tensor:
    """drop: {"layer1": multiplier [B, n, C], "layer2": ...} for train mode (n = 1 + padded tile count)."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    drop = drop or {}
    h = F.relu(F.linear(bags, sd["_fc1.0.weight"], sd["_fc1.0.bias"]))
    D = h.shape[-1]
    n = h.shape[1]
    side = int(math.ceil(math.sqrt(n)))
    h = torch.cat([h, h[:, : side * side - n]], dim=1)        # wrap-pad with the FIRST tiles (:306-309)
    h = torch.cat([sd["cls_token"].expand(h.shape[0], -1, -1), h], dim=1)
    for name in ("layer1", "pos", "layer2"):
        if name == "pos":
            h = ppeg(h, sd, "pos_layer.", side, side)
            continue
        y = F.layer_norm(h, (D,), sd[f"{name}.norm.weight"], sd[f"{name}.norm.bias"])
        h = h + nystrom_attention(y, sd, f"{name}.attn.", heads=8, landmarks=D // 2, drop=drop.get(name))
    h = F.layer_norm(h, (D,), sd["norm.weight"], sd["norm.bias"])[:, 0]
    return F.linear(h, sd["_fc2.weight"], sd["_fc2.bias"])
