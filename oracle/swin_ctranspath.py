"""Oracle for the in-tree tile encoder: CTransPath = ConvStem + Swin-T, u8 tile -> 768-d.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY STATUS: **pinned** -- unlike the timm ViT trunks, this network's arithmetic is in the reference tree
(src/stamp/preprocessing/extractor/ctranspath.py), so tools/make_golden.py instantiates the reference's own
``_swin_tiny_patch4_window7_224(embed_layer=_ConvStem)`` (and a two-stage 112-px variant), loads the seeded weights
of ``stamp_amd.swin.random_swin_state_dict`` into it with ``strict=True`` and stores its outputs in
tests/golden/ctranspath_*.npz; tests/test_oracle_golden.py holds this restatement to those.

The restatement is written token-major with explicit index arithmetic instead of roll / view / permute:
  * ConvStem (ctranspath.py:386-444): conv3x3 s2 p1 (no bias) -> BatchNorm2d (eval: running stats) -> ReLU, twice
    (3 -> C/8 -> C/4), conv1x1 (C/4 -> C, bias), flatten to tokens, LayerNorm (patch_norm=True, :905-911).
  * Swin block (ctranspath.py:654-698): x += proj(WMSA(LN1 x)); x += fc2(gelu(fc1(LN2 x))).  A window of the
    (cyclically shifted) grid holds the tokens (wh*7+i, ww*7+j); `torch.roll(x, -s)` means shifted[h] = x[(h+s) % H]
    (:663-668), so the window's natural token ids are ((wh*7+i+s) % H, (ww*7+j+s) % W), and the outputs go back to
    the same ids (:683-690).
  * Window attention (ctranspath.py:510-547): softmax(q*scale k^T + B[rel_idx] + mask) v with rel_idx(p,q) =
    (pi-qi+6)*13 + (pj-qj+6) (:478-496) and mask = -100 where the two tokens carry different region labels of the
    shifted grid (labels by h in [0,H-7), [H-7,H-3), [H-3,H), same for w; :620-645).
  * PatchMerging (ctranspath.py:717-738): concat of the (0,0),(1,0),(0,1),(1,1) members of each 2x2 cell ->
    LayerNorm(4C) -> Linear(4C, 2C, no bias).
  * head (ctranspath.py:975-988): LayerNorm -> mean over tokens; `model.head = nn.Identity()` (:51).
Transform: v2.Resize(224)/CenterCrop(224) are identities on 224-px tiles; ToDtype(scale=True) + Normalize(ImageNet
mean/std) = (u8/255 - mean)/std (ctranspath.py:56-64, chief_ctranspath.py:45-51).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

WS = 7           # window size (ctranspath.py:1003)
HEAD_DIM = 32    # 96/3 = 192/6 = 384/12 = 768/24


def window_token_ids(H: int, W: int, shift: int) -> torch.Tensor:
    """[nW, 49] natural token ids (h*W + w) of every window of the grid rolled by -shift."""
    wh = torch.arange(H // WS).view(-1, 1, 1, 1)
    ww = torch.arange(W // WS).view(1, -1, 1, 1)
    i = torch.arange(WS).view(1, 1, -1, 1)
    j = torch.arange(WS).view(1, 1, 1, -1)
    h = (wh * WS + i + shift) % H
    w = (ww * WS + j + shift) % W
    return (h * W + w).reshape(-1, WS * WS)


def window_region_labels(H: int, W: int, shift: int) -> torch.Tensor:
    """[nW, 49] region label (0..8) of each window token in the SHIFTED grid (ctranspath.py:620-640)."""
    def lab(n, size):
        return (n >= size - WS).long() + (n >= size - shift).long()
    wh = torch.arange(H // WS).view(-1, 1, 1, 1)
    ww = torch.arange(W // WS).view(1, -1, 1, 1)
    i = torch.arange(WS).view(1, 1, -1, 1)
    j = torch.arange(WS).view(1, 1, 1, -1)
    return (3 * lab(wh * WS + i, H) + lab(ww * WS + j, W)).reshape(-1, WS * WS)


def rel_pos_index() -> torch.Tensor:
    """[49,49] index into the (2*7-1)^2-row bias table (ctranspath.py:478-496)."""
    p = torch.arange(WS * WS)
    pi, pj = p // WS, p % WS
    return (pi[:, None] - pi[None, :] + WS - 1) * (2 * WS - 1) + (pj[:, None] - pj[None, :] + WS - 1)


def conv_stem(x_chw: torch.Tensor, sd: dict, eps_bn: float = 1e-5) -> torch.Tensor:
    dt = x_chw.dtype
    g = lambda n: sd["patch_embed." + n].to(dt)  # noqa: E731
    x = x_chw
    for conv, bn in ((0, 1), (3, 4)):
        x = F.conv2d(x, g(f"proj.{conv}.weight"), None, stride=2, padding=1)
        scale = g(f"proj.{bn}.weight") / torch.sqrt(g(f"proj.{bn}.running_var") + eps_bn)
        x = (x - g(f"proj.{bn}.running_mean").view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1) + g(f"proj.{bn}.bias").view(1, -1, 1, 1)
        x = torch.relu(x)
    x = F.conv2d(x, g("proj.6.weight"), g("proj.6.bias"))
    x = x.flatten(2).transpose(1, 2)
    C = x.shape[-1]
    return F.layer_norm(x, (C,), g("norm.weight"), g("norm.bias"), 1e-5)


def swin_block(x: torch.Tensor, sd: dict, pre: str, H: int, W: int, heads: int, shift: int) -> torch.Tensor:
    dt = x.dtype
    g = lambda n: sd[pre + n].to(dt)  # noqa: E731
    B, L, C = x.shape
    if min(H, W) <= WS:            # window covers the grid: no shift, no mask (ctranspath.py:590-593)
        shift = 0
    ids = window_token_ids(H, W, shift)                      # [nW,49]
    nW = ids.shape[0]
    h = F.layer_norm(x, (C,), g("norm1.weight"), g("norm1.bias"), 1e-5)
    hw = h[:, ids.reshape(-1), :].reshape(B * nW, WS * WS, C)
    qkv = F.linear(hw, g("attn.qkv.weight"), g("attn.qkv.bias")).reshape(B * nW, WS * WS, 3, heads, C // heads)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))     # [B*nW, heads, 49, hd]
    att = (q * (C // heads) ** -0.5) @ k.transpose(-2, -1)
    bias = g("attn.relative_position_bias_table")[rel_pos_index().reshape(-1)].reshape(WS * WS, WS * WS, heads)
    att = att + bias.permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        lab = window_region_labels(H, W, shift)
        mask = torch.where(lab[:, :, None] != lab[:, None, :], -100.0, 0.0).to(dt)      # [nW,49,49]
        att = (att.reshape(B, nW, heads, WS * WS, WS * WS) + mask[None, :, None]).reshape(B * nW, heads, WS * WS, WS * WS)
    o = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).reshape(B * nW, WS * WS, C)
    o = F.linear(o, g("attn.proj.weight"), g("attn.proj.bias")).reshape(B, nW * WS * WS, C)
    a = torch.empty_like(x)
    a[:, ids.reshape(-1), :] = o
    x = x + a
    h = F.layer_norm(x, (C,), g("norm2.weight"), g("norm2.bias"), 1e-5)
    h = F.gelu(F.linear(h, g("mlp.fc1.weight"), g("mlp.fc1.bias")))
    return x + F.linear(h, g("mlp.fc2.weight"), g("mlp.fc2.bias"))


def patch_merge(x: torch.Tensor, sd: dict, pre: str, H: int, W: int) -> torch.Tensor:
    dt = x.dtype
    B, L, C = x.shape
    g4 = x.reshape(B, H // 2, 2, W // 2, 2, C)          # [B, h2, dh, w2, dw, C]
    cat = torch.cat([g4[:, :, 0, :, 0], g4[:, :, 1, :, 0], g4[:, :, 0, :, 1], g4[:, :, 1, :, 1]], dim=-1)
    cat = cat.reshape(B, (H // 2) * (W // 2), 4 * C)
    cat = F.layer_norm(cat, (4 * C,), sd[pre + "norm.weight"].to(dt), sd[pre + "norm.bias"].to(dt), 1e-5)
    return F.linear(cat, sd[pre + "reduction.weight"].to(dt))


def swin_features(tiles_u8_hwc: torch.Tensor, sd: dict, cfg, dtype=torch.float32, taps: dict | None = None) -> torch.Tensor:
    """u8 [B,S,S,3] -> [B, C_last] in `dtype` (fp32 = the reference's arithmetic; fp64 for conditioning studies).
    `taps`, if given, receives the token tensor after the stem and after every stage."""
    x = tiles_u8_hwc.permute(0, 3, 1, 2).to(dtype) / 255.0
    x = (x - torch.tensor(cfg.mean, dtype=dtype).view(1, 3, 1, 1)) / torch.tensor(cfg.std, dtype=dtype).view(1, 3, 1, 1)
    x = conv_stem(x, sd)
    if taps is not None:
        taps["stem"] = x
    H = W = cfg.img // 4
    for s, depth in enumerate(cfg.depths):
        for b in range(depth):
            x = swin_block(x, sd, f"layers.{s}.blocks.{b}.", H, W, cfg.heads[s], 0 if b % 2 == 0 else WS // 2)
        if s + 1 < len(cfg.depths):
            x = patch_merge(x, sd, f"layers.{s}.downsample.", H, W)
            H, W = H // 2, W // 2
        if taps is not None:
            taps[f"stage{s}"] = x
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd["norm.weight"].to(dtype), sd["norm.bias"].to(dtype), 1e-5)
    return x.mean(dim=1)


def swin_encode_f16(tiles_u8_hwc: torch.Tensor, sd: dict, cfg) -> torch.Tensor:
    """`model(tiles).half()` of the reference feature loop (src/stamp/preprocessing/__init__.py:324-325)."""
    return swin_features(tiles_u8_hwc, sd, cfg).half()
