"""Host side of the HIP CTransPath tile encoder: ConvStem + Swin-T behind the reference's
Extractor seam.  Mirrors `ctranspath()` / `chief_ctranspath()` (reference
src/stamp/preprocessing/extractor/ctranspath.py:34-70, chief_ctranspath.py:20-57): the model object is called as
`model(tiles)` and returns `[B, 768]`; weights arrive as the reference `_SwinTransformer.state_dict()` (the
`ctranspath.pth["model"]` dict), same key names.

All arithmetic is in libamdstamp.so (`amds_swin_forward`); this file only packs weights and owns the workspace.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib


@dataclass(frozen=True)
class SwinConfig:
    """`_SwinTransformer` hyper-parameters (ctranspath.py:857-878); window 7 and head_dim 32 are fixed by the kernels."""
    img: int = 224
    embed: int = 96
    depths: tuple[int, ...] = (2, 2, 6, 2)
    heads: tuple[int, ...] = (3, 6, 12, 24)
    mean: tuple[float, float, float] = (0.485, 0.456, 0.406)     # ctranspath.py:62
    std: tuple[float, float, float] = (0.229, 0.224, 0.225)

    @property
    def out_dim(self) -> int:
        return self.embed * 2 ** (len(self.depths) - 1)

    @property
    def grid(self) -> int:
        return self.img // 4

    def matmul_flops_per_tile(self) -> float:
        """2*MAC of every conv / linear / attention contraction (LN, GELU, softmax excluded)."""
        g, C0 = self.grid, self.embed
        fl = 2.0 * (g * 2) ** 2 * 27 * (C0 // 8) + 2.0 * g * g * 9 * (C0 // 8) * (C0 // 4) + 2.0 * g * g * (C0 // 4) * C0
        for s, d in enumerate(self.depths):
            L, Cs = (g >> s) ** 2, C0 << s
            fl += d * (2.0 * L * Cs * (3 * Cs + Cs + 8 * Cs) + 4.0 * L * 49 * Cs)
            if s + 1 < len(self.depths):
                fl += 2.0 * (L // 4) * 4 * Cs * 2 * Cs
        return fl


SWIN_PRESETS = {
    "ctranspath": SwinConfig(),                                             # ctranspath.py:999-1010
    "test_swin_tiny": SwinConfig(img=112, depths=(2, 2), heads=(3, 6)),     # 28x28 -> 14x14 grid, 192-d
}


def swin_param_shapes(cfg: SwinConfig) -> list[tuple[str, tuple[int, ...]]]:
    """(name, shape) of every floating-point entry of the reference state_dict, in its order."""
    C0 = cfg.embed
    out: list[tuple[str, tuple[int, ...]]] = []
    cin = 3
    for conv, bn, cout in ((0, 1, C0 // 8), (3, 4, C0 // 4)):
        out.append((f"patch_embed.proj.{conv}.weight", (cout, cin, 3, 3)))
        for n in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"patch_embed.proj.{bn}.{n}", (cout,)))
        cin = cout
    out += [("patch_embed.proj.6.weight", (C0, cin, 1, 1)), ("patch_embed.proj.6.bias", (C0,)),
            ("patch_embed.norm.weight", (C0,)), ("patch_embed.norm.bias", (C0,))]
    for s, d in enumerate(cfg.depths):
        Cs, nh = C0 << s, cfg.heads[s]
        for b in range(d):
            p = f"layers.{s}.blocks.{b}."
            out += [(p + "norm1.weight", (Cs,)), (p + "norm1.bias", (Cs,)),
                    (p + "attn.relative_position_bias_table", (169, nh)),
                    (p + "attn.qkv.weight", (3 * Cs, Cs)), (p + "attn.qkv.bias", (3 * Cs,)),
                    (p + "attn.proj.weight", (Cs, Cs)), (p + "attn.proj.bias", (Cs,)),
                    (p + "norm2.weight", (Cs,)), (p + "norm2.bias", (Cs,)),
                    (p + "mlp.fc1.weight", (4 * Cs, Cs)), (p + "mlp.fc1.bias", (4 * Cs,)),
                    (p + "mlp.fc2.weight", (Cs, 4 * Cs)), (p + "mlp.fc2.bias", (Cs,))]
        if s + 1 < len(cfg.depths):
            p = f"layers.{s}.downsample."
            out += [(p + "reduction.weight", (2 * Cs, 4 * Cs)), (p + "norm.weight", (4 * Cs,)), (p + "norm.bias", (4 * Cs,))]
    Cl = cfg.out_dim
    out += [("norm.weight", (Cl,)), ("norm.bias", (Cl,))]
    return out


def random_swin_state_dict(cfg: SwinConfig, seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded weights with the reference's names (no checkpoint is reachable offline; the real one is fetched by
    gdown, ctranspath.py:36-41).  A pure function of (cfg, seed) and torch's CPU generator, so the golden fixtures
    only need to store inputs and outputs.  Fan-in scaled weights, random biases / norm affines / BatchNorm running
    statistics and an O(1) relative-position table, so every term of the network matters; residual branches carry
    gain 0.5 to keep the map well conditioned."""
    g = torch.Generator().manual_seed(seed)
    sd: dict[str, torch.Tensor] = {}
    for name, shape in swin_param_shapes(cfg):
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("relative_position_bias_table"):
            t = torch.randn(shape, generator=g) * 0.7
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.2
        elif len(shape) == 1:
            is_scale = leaf == "weight"
            t = (1.0 if is_scale else 0.0) + torch.randn(shape, generator=g) * (0.2 if is_scale else 0.1)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 0.5 if (".proj.weight" in name and "attn" in name) or "fc2.weight" in name else 1.0
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        sd[name] = t
    return sd
