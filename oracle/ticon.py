"""CPU restatement of the reference's TICON tile contextualiser in the form STAMP's extractor uses it -- TEST INFRASTRUCTURE ONLY.

Reference: src/stamp/preprocessing/extractor/ticon.py -- `HOptimusTICON.forward` :691-718 feeds every tile's H-optimus-1 embedding ALONE
(sequence length 1, relative coordinates (0, 0)) through `EncoderDecoder.forward` :543-562 -> `forward_features` :485-541 with
dec_layer=None: `input_proj_<key>` (ProjectionMlp :80-99: Linear, SiLU, Linear, LayerNorm), the encoder's blocks :346-394 (Block :290-343:
x += gamma1 * Attention(LN(x)); x += gamma2 * Mlp(LN(x)); Mlp :54-77 = fc1, chunk, SiLU(x1) * x2, fc2), `enc_norm` :506.
With ONE token the attention (:183-215) is the identity on its value: softmax over a single key is 1 whatever the ALiBi bias
(-slope * 0), so Attention(h) = proj(v_proj(h)) and q_proj / k_proj never touch the output.
Pinned by tests/golden/ticon.npz, produced by running the reference's own `EncoderDecoder` (tools/make_golden.py::golden_ticon).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def ticon_tile_forward(emb: torch.Tensor, sd: dict, key: str, eps: float = 1e-5) -> torch.Tensor:
    """emb [B, in_dim] -> [B, D]: what `EncoderDecoder(x=emb[:, None], relative_coords=zeros, tile_encoder_key=key)[:, 0]` returns."""
    sd = {k: v.to(emb.dtype) for k, v in sd.items()}
    p = f"input_proj_dict.input_proj_{key}."
    x = F.linear(F.silu(F.linear(emb, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])      # :94-98
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps)
    l = 0
    while f"encoder.blocks.{l}.residual1.norm.weight" in sd:
        b = f"encoder.blocks.{l}."
        h = F.layer_norm(x, (D,), sd[b + "residual1.norm.weight"], sd[b + "residual1.norm.bias"], eps)
        v = F.linear(h, sd[b + "residual1.fn.v_proj.weight"], sd[b + "residual1.fn.v_proj.bias"])                                  # one key: attention = value
        x = x + sd[b + "residual1.gamma"] * F.linear(v, sd[b + "residual1.fn.proj.weight"], sd[b + "residual1.fn.proj.bias"])       # :262-263
        h = F.layer_norm(x, (D,), sd[b + "residual2.norm.weight"], sd[b + "residual2.norm.bias"], eps)
        u = F.linear(h, sd[b + "residual2.fn.fc1.weight"], sd[b + "residual2.fn.fc1.bias"])
        x1, x2 = u.chunk(2, dim=-1)                                                                                               # :73-75
        x = x + sd[b + "residual2.gamma"] * F.linear(F.silu(x1) * x2, sd[b + "residual2.fn.fc2.weight"], sd[b + "residual2.fn.fc2.bias"])
        l += 1
    return F.layer_norm(x, (D,), sd["enc_norm.weight"], sd["enc_norm.bias"], eps)                                                  # :506
