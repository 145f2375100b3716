"""CPU restatement of the reference's barspoon head (`EncDecTransformer`) -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline);
the product path never imports it.

Reference: src/stamp/modeling/models/barspoon.py -- `EncDecTransformer.__init__` :104-162 (projector Linear + ReLU, a pre-norm
nn.TransformerEncoder, one learned class token and one Linear head per target, a pre-norm nn.TransformerDecoder), `.forward` :164-205
(sinusoidal encoding of the tile positions :173-186, encoder :188, class tokens decoded against the tile tokens :190-193, per-target heads
:196-203), `sanitize` :351-352.  The two torch containers are restated from their documented pre-norm form (`norm_first=True`, ReLU,
eps 1e-5; eval mode: dropout = identity):  x += SA(LN1(x)); [decoder: x += MHA(LN2(x), memory)]; x += W2 relu(W1 LN(x)).
Pinned by tests/golden/barspoon.npz, produced by running the reference class itself (tools/make_golden.py::golden_barspoon).
"""
from __future__ import annotations

import re

import torch
import torch.nn.functional as F


def sanitize(x: str) -> str:
    return re.sub(r"[^A-Za-z0-9_]", "_", x)                                               # barspoon.py:351-352


def _mha(q_in, kv_in, sd, p, H):
    """nn.MultiheadAttention (batch_first, no masks, eval): packed in-projection, softmax(q k^T / sqrt(hd)) v, out-projection."""
    D = q_in.shape[-1]
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(q_in, w[:D], b[:D])
    k = F.linear(kv_in, w[D:2 * D], b[D:2 * D])
    v = F.linear(kv_in, w[2 * D:], b[2 * D:])
    B, Lq, _ = q.shape
    Lk, hd = k.shape[1], D // H
    q, k, v = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    a = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Lq, D)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def _ff(x, sd, p):
    return F.linear(torch.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])), sd[p + "linear2.weight"], sd[p + "linear2.bias"])


def positional_encodings(tile_positions: torch.Tensor, d_model: int) -> torch.Tensor:
    """[B, T, 2] -> [B, T, d_model]: sin of (x / f_i, y / f_i) then cos of the same, f_i = 100000^(i / d_model), i < d_model / 4 (:173-186)."""
    x = tile_positions.unsqueeze(-1) / 100_000 ** (torch.arange(d_model // 4).type_as(tile_positions) / d_model)
    return torch.cat([torch.sin(x).flatten(start_dim=-2), torch.cos(x).flatten(start_dim=-2)], dim=-1)


def barspoon_forward(tile_tokens: torch.Tensor, tile_positions: torch.Tensor, sd: dict, target_labels: list[str], *, num_encoder_heads: int = 8,
                     num_decoder_heads: int = 8, positional_encoding: bool = True) -> dict[str, torch.Tensor]:
    """tile_tokens [B, T, F], tile_positions [B, T, 2] -> {target: logits [B, n_out]}."""
    sd = {k: v.to(tile_tokens.dtype) for k, v in sd.items()}
    x = torch.relu(F.linear(tile_tokens, sd["projector.0.weight"], sd["projector.0.bias"]))            # :171
    if positional_encoding:
        x = x + positional_encodings(tile_positions.to(x.dtype), x.shape[-1])                             # :173-186
    l = 0
    while f"transformer_encoder.layers.{l}.norm1.weight" in sd:                                          # :188
        p = f"transformer_encoder.layers.{l}."
        h = _ln(x, sd, p + "norm1.")
        x = x + _mha(h, h, sd, p + "self_attn.", num_encoder_heads)
        x = x + _ff(_ln(x, sd, p + "norm2."), sd, p)
        l += 1
    B = x.shape[0]
    t = torch.stack([sd["class_tokens." + sanitize(tl)] for tl in target_labels]).expand(B, -1, -1)       # :190-192
    l = 0
    while f"transformer_decoder.layers.{l}.norm1.weight" in sd:                                          # :193
        p = f"transformer_decoder.layers.{l}."
        h = _ln(t, sd, p + "norm1.")
        t = t + _mha(h, h, sd, p + "self_attn.", num_decoder_heads)
        t = t + _mha(_ln(t, sd, p + "norm2."), x, sd, p + "multihead_attn.", num_decoder_heads)
        t = t + _ff(_ln(t, sd, p + "norm3."), sd, p)
        l += 1
    return {tl: F.linear(t[:, j], sd[f"heads.{sanitize(tl)}.weight"], sd[f"heads.{sanitize(tl)}.bias"]) for j, tl in enumerate(target_labels)}   # :196-203
