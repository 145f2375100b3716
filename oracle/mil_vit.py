"""Oracle for the `vit` MIL head.

Restates reference src/stamp/modeling/models/vision_tranformer.py: VisionTransformer.forward :331-384,
Transformer.forward :281-295, SelfAttention.forward :194-242 (nn.MultiheadAttention branch and MultiHeadALiBi
branch), feed_forward :157-169, _ALiBi.forward :42-74, _RunningMeanScaler :15-31.  Eval mode by default; the TRAIN-mode dropout
sites are restated as explicit multiplier masks (`drop`): `project_features.2` = Dropout(dropout) after the GELU (:314-318),
nn.MultiheadAttention's dropout on the attention probabilities (:191; torch semantics), and the two Dropout(0.5) of feed_forward
(:157-169 -- `Transformer` calls `feed_forward(dim, mlp_dim)` WITHOUT a rate, :268-271, so 0.5 whatever the config says).
Pinned by tests/golden/mil_vit_{plain,alibi}.npz (eval, +-mask) and tests/golden/mil_vit_train_{plain,alibi}.npz (train mode: the
masks the reference's own nn.Dropout modules drew, its logits, loss, parameter gradients and updated scaler buffers), all captured
from the imported reference module by tools/make_golden.py.  Test infrastructure: only tests/ and smoke() import this.
State-dict keys are the reference's (project_features.0, class_token, transformer.layers.{l}.0.{norm,mhsa...},
transformer.layers.{l}.1.{0,1,4}, transformer.norm, mlp_head.0).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _mha(x, sd, pre, heads, attn_mask, drop=None):
    """nn.MultiheadAttention(batch_first=True) self-attention, need_weights=False; attn_mask True = blocked.
    drop: multiplier [B, heads, T, T] applied to the softmax weights (train-mode attention dropout, already scaled by 1/(1-p))."""
    B, T, D = x.shape
    hd = D // heads
    qkv = F.linear(x, sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"])
    q, k, v = (t.reshape(B, T, heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    s = (q @ k.transpose(-1, -2)) / hd ** 0.5
    if attn_mask is not None:
        # The reference passes attn_mask.repeat(heads, 1, 1) (:224): row i of that [heads*B] stack is mask[i % B],
        # while nn.MultiheadAttention reads row b*heads + h for (batch b, head h).  Restated literally:
        # (b, h) is masked with the mask of batch (b*heads + h) % B.
        idx = (torch.arange(B)[:, None] * heads + torch.arange(heads)[None, :]) % B
        s = s.masked_fill(attn_mask[idx], float("-inf"))
    w = torch.softmax(s, dim=-1)
    if drop is not None:
        w = w * drop
    o = w @ v
    o = o.transpose(1, 2).reshape(B, T, D)
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def _alibi_mha(x, coords, sd, pre, heads, attn_mask, alibi_mask):
    """MultiHeadALiBi (:77-154): per-head Linear q/k/v; the distance bias is SUBTRACTED AFTER the softmax (:58-72)."""
    outs = []
    dist = torch.cdist(coords, coords)
    for h in range(heads):
        q = F.linear(x, sd[f"{pre}query_encoders.{h}.weight"], sd[f"{pre}query_encoders.{h}.bias"])
        k = F.linear(x, sd[f"{pre}key_encoders.{h}.weight"], sd[f"{pre}key_encoders.{h}.bias"])
        v = F.linear(x, sd[f"{pre}value_encoders.{h}.weight"], sd[f"{pre}value_encoders.{h}.bias"])
        logits = (q @ k.transpose(-1, -2)) * (k.shape[-1] ** -0.5)
        scaled = dist / sd[f"{pre}attentions.{h}.scale_distance.running_mean"] * sd[f"{pre}attentions.{h}.bias_scale"]
        if alibi_mask is not None:
            scaled = scaled.masked_fill(alibi_mask, 0.0)
        w = torch.softmax(logits, dim=-1) - scaled
        if attn_mask is not None:
            w = w.masked_fill(attn_mask, 0.0)
        outs.append(w @ v)
    return F.linear(torch.cat(outs, dim=-1), sd[pre + "fc.weight"], sd[pre + "fc.bias"])


def mil_vit_forward(bags, coords, mask, sd, *, n_heads: int, use_alibi: bool, drop: dict | None = None, dtype=torch.float32):
    """bags [B,T,F], coords [B,T,2], mask bool [B,T] or None -> logits [B,C].
    drop (train mode): multipliers keyed "proj" [B,T,D], f"attn{l}" [B,H,T+1,T+1], f"ff1_{l}" [B,T+1,FF], f"ff2_{l}" [B,T+1,D]."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    drop = drop or {}
    B = bags.shape[0]
    x = F.gelu(F.linear(bags, sd["project_features.0.weight"], sd["project_features.0.bias"]))
    if "proj" in drop:
        x = x * drop["proj"]
    D = x.shape[-1]
    x = torch.cat([sd["class_token"].reshape(1, 1, D).expand(B, -1, -1), x], dim=1)
    coords = torch.cat([coords.new_zeros(B, 1, 2), coords], dim=1)
    attn_mask = alibi_mask = None
    if mask is not None:
        m = torch.cat([mask.new_zeros(B, 1), mask], dim=1)
        attn_mask = m[:, :, None] & m[:, None, :]          # einsum of bools (:363-365)
        attn_mask[:, 1:, 0] = True                          # tiles may not attend to the class token (:367)
        alibi_mask = torch.zeros_like(attn_mask)
        alibi_mask[:, 0, :] = True
        alibi_mask[:, :, 0] = True
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.layers."))
    for l in range(n_layers):
        p = f"transformer.layers.{l}."
        h = F.layer_norm(x, (D,), sd[p + "0.norm.weight"], sd[p + "0.norm.bias"])
        if use_alibi:
            a = _alibi_mha(h, coords, sd, p + "0.mhsa.", n_heads, attn_mask, alibi_mask)
        else:
            a = _mha(h, sd, p + "0.mhsa.", n_heads, attn_mask, drop.get(f"attn{l}"))
        x = a + x
        h = F.layer_norm(x, (D,), sd[p + "1.0.weight"], sd[p + "1.0.bias"])
        h = F.gelu(F.linear(h, sd[p + "1.1.weight"], sd[p + "1.1.bias"]))
        if f"ff1_{l}" in drop:
            h = h * drop[f"ff1_{l}"]
        h = F.linear(h, sd[p + "1.4.weight"], sd[p + "1.4.bias"])
        if f"ff2_{l}" in drop:
            h = h * drop[f"ff2_{l}"]
        x = h + x
    x = F.layer_norm(x, (D,), sd["transformer.norm.weight"], sd["transformer.norm.bias"])
    return F.linear(x[:, 0], sd["mlp_head.0.weight"], sd["mlp_head.0.bias"])


def running_mean_update(running_mean: torch.Tensor, items_so_far: torch.Tensor, dist: torch.Tensor):
    """_RunningMeanScaler train-mode step (:22-29): rm <- mean(rm + (x - rm)/n); n <- n + 1."""
    rm = (running_mean + (dist - running_mean) / items_so_far).mean().reshape(1)
    return rm, items_so_far + 1
