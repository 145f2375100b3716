"""Oracle for the tile-encoder path: u8 tile -> transform -> timm-style
VisionTransformer -> CLS feature -> fp16.

PARITY STATUS: **unpinned against timm itself** -- the ViT trunk's arithmetic lives in the third-party package
``timm==1.0.25`` (reference uv.lock), which is neither under /root/reference nor installed here; the reference's
own tests only smoke-test this boundary (tests/test_feature_extractors.py:20-80).  The restatement follows timm's
published ``VisionTransformer`` semantics (pre-LN blocks, LayerScale, SwiGLUPacked = fc1 -> chunk(2) ->
SiLU(x1)*x2 -> fc2, register tokens after CLS, ``no_embed_class``) and EVERY branch of it is cross-checked against
independent third-party implementations that ARE installed (HF ``transformers`` ``Dinov2Model`` and
``Dinov2WithRegistersModel``: GELU and SwiGLU FFN, 0/4/8 register tokens, both ``no_embed_class`` settings, head_dim
64 and 80, and the full-size 24-block ViT-L/14) in tests/test_oracle_vit.py.  This is test infrastructure: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
Call sites restated: reference src/stamp/preprocessing/extractor/virchow2.py:29-45 (CLS select),
uni2.py:17-37, reddino.py:40-57, h_optimus_0.py:22-30 (transform), src/stamp/preprocessing/__init__.py:324-325
(``model(tiles).half()``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def tile_transform(tiles_u8_hwc: torch.Tensor, mean, std) -> torch.Tensor:
    """ToTensor + Normalize on already-224x224 tiles: u8 [B,H,W,3] -> f32 [B,3,H,W] = (x/255 - mean)/std.
    (reference h_optimus_0.py:22-30, mstar.py:19-25; Resize/CenterCrop are identities at 224.)"""
    x = tiles_u8_hwc.permute(0, 3, 1, 2).to(torch.float32) / 255.0
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - m) / s


def vit_tokens(x_chw: torch.Tensor, sd: dict, cfg, sdpa: bool = False) -> torch.Tensor:
    """timm VisionTransformer.forward_features on normalised float input -> final-LayerNorm'd tokens [B,T,D].
    sdpa=True evaluates the attention through F.scaled_dot_product_attention (what timm's fused_attn path calls; same
    arithmetic, torch's blocked CPU kernel) -- used by bench.py's cpu_baseline leg; the parity tests use the explicit form."""
    D, p = cfg.dim, cfg.patch
    B = x_chw.shape[0]
    x = F.conv2d(x_chw, sd["patch_embed.proj.weight"].float(), sd["patch_embed.proj.bias"].float(), stride=p)
    x = x.flatten(2).transpose(1, 2)                                   # [B, np, D]
    pos = sd["pos_embed"].float().reshape(1, -1, D)
    to_cat = [sd["cls_token"].float().reshape(1, 1, D).expand(B, -1, -1)]
    if cfg.reg_tokens:
        to_cat.append(sd["reg_token"].float().reshape(1, cfg.reg_tokens, D).expand(B, -1, -1))
    if cfg.no_embed_class:        # timm _pos_embed: add then concat
        x = torch.cat(to_cat + [x + pos], dim=1)
    else:                         # concat then add
        x = torch.cat(to_cat + [x], dim=1) + pos
    H, hd = cfg.heads, D // cfg.heads
    for i in range(cfg.depth):
        g = lambda n: sd[f"blocks.{i}.{n}"].float()  # noqa: E731
        h = F.layer_norm(x, (D,), g("norm1.weight"), g("norm1.bias"), cfg.ln_eps)
        qkv = F.linear(h, g("attn.qkv.weight"), g("attn.qkv.bias"))
        qkv = qkv.reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if sdpa:
            att = F.scaled_dot_product_attention(q, k, v)
        else:
            att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1) @ v
        att = att.transpose(1, 2).reshape(B, -1, D)
        att = F.linear(att, g("attn.proj.weight"), g("attn.proj.bias"))
        if cfg.layerscale:
            att = att * g("ls1.gamma")
        x = x + att
        h = F.layer_norm(x, (D,), g("norm2.weight"), g("norm2.bias"), cfg.ln_eps)
        h = F.linear(h, g("mlp.fc1.weight"), g("mlp.fc1.bias"))
        if cfg.mlp == "swiglu":
            x1, x2 = h.chunk(2, dim=-1)
            h = F.silu(x1) * x2
        else:
            h = F.gelu(h)         # exact erf, nn.GELU default
        h = F.linear(h, g("mlp.fc2.weight"), g("mlp.fc2.bias"))
        if cfg.layerscale:
            h = h * g("ls2.gamma")
        x = x + h
    return F.layer_norm(x, (D,), sd["norm.weight"].float(), sd["norm.bias"].float(), cfg.ln_eps)


def extract_features(tiles_u8_hwc: torch.Tensor, sd: dict, cfg, return_tokens: bool = False, sdpa: bool = False):
    """The reference's per-batch step: model(transform(tiles))[:, 0].half()
    (src/stamp/preprocessing/__init__.py:324-325 + virchow2.py:29-30)."""
    with torch.no_grad():
        toks = vit_tokens(tile_transform(tiles_u8_hwc, cfg.mean, cfg.std), sd, cfg, sdpa=sdpa)
    feats = toks[:, 0].half()
    return (feats, toks) if return_tokens else feats
