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
    "ctranspath": SwinConfig(),                                             # ctranspath.py:999-1009
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


# ---- lane-ordered attention tables ------------------------------------------------------------------
LOG2E = 1.4426950408889634
_PAD_KEY = -30000.0          # finite "minus infinity" for the 15 pad keys of a 64-key MFMA tile


def _lane_key_query() -> tuple[torch.Tensor, torch.Tensor]:
    """(key, query) window positions addressed by accumulator element [kt][qt][lane][r] of the S^T = K Q^T MFMA tiles
    (32x32 C layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))."""
    kt = torch.arange(2).view(2, 1, 1, 1)
    qt = torch.arange(2).view(1, 2, 1, 1)
    lane = torch.arange(64).view(1, 1, 64, 1)
    r = torch.arange(16).view(1, 1, 1, 16)
    key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) + 0 * qt
    query = 32 * qt + (lane & 31) + 0 * kt + 0 * r
    return key.expand(2, 2, 64, 16), query.expand(2, 2, 64, 16)


def rel_bias_lane_table(table: torch.Tensor) -> torch.Tensor:
    """relative_position_bias_table [169, heads] -> [heads][2][2][64][16] fp32, bias * log2(e) in lane order
    (index rule of ctranspath.py:478-496); pad keys get a large negative value, pad queries 0."""
    key, query = _lane_key_query()
    kv, qv = key < 49, query < 49
    k, q = key.clamp(max=48), query.clamp(max=48)
    idx = (q // 7 - k // 7 + 6) * 13 + (q % 7 - k % 7 + 6)
    t = table.detach().double().cpu()
    out = t[idx.reshape(-1)].reshape(2, 2, 64, 16, -1).permute(4, 0, 1, 2, 3) * LOG2E
    out = torch.where(qv.unsqueeze(0), out, torch.zeros_like(out))
    out = torch.where(kv.unsqueeze(0), out, torch.full_like(out, _PAD_KEY))
    return out.float().contiguous()


def shift_mask_lane_table() -> torch.Tensor:
    """[4][2][2][64][16] bool: True where query and key of a shifted window carry different region labels, i.e. where
    the reference adds -100 (ctranspath.py:620-645).  Window type = 2*(last window row) + (last window column); inside
    such a window the label along an axis is 1 for in-window offsets 0..3 and 2 for 4..6 (the rows rolled around)."""
    key, query = _lane_key_query()
    k, q = key.clamp(max=48), query.clamp(max=48)
    out = torch.zeros(4, 2, 2, 64, 16, dtype=torch.bool)
    for typ in range(4):
        def lab(p):
            lh = (1 + (p // 7 >= 4).long()) if typ & 2 else torch.zeros_like(p)
            lw = (1 + (p % 7 >= 4).long()) if typ & 1 else torch.zeros_like(p)
            return 3 * lh + lw
        out[typ] = lab(q) != lab(k)
    return out


def shift_mask_bits() -> torch.Tensor:
    """The same masks packed for `amds_window_attention`: int64 [4][64], bit (kt*2+qt)*16 + r of a lane's word."""
    m = shift_mask_lane_table().long()                       # [4][kt][qt][lane][r]
    w = (1 << torch.arange(16, dtype=torch.int64)).view(1, 1, 1, 1, 16)
    per_tile = (m * w).sum(-1)                               # [4][2][2][64]
    sh = (torch.arange(2).view(2, 1) * 2 + torch.arange(2).view(1, 2)) * 16
    bits = (per_tile << sh.view(1, 2, 2, 1)).sum(dim=(1, 2))
    # reinterpret as signed 64-bit (bit 63 may be set)
    return bits.contiguous()


def pack_stem_params(sd: dict[str, torch.Tensor], cfg: SwinConfig, eps_bn: float = 1e-5) -> torch.Tensor:
    """ConvStem parameters in the layout `amds_swin_stem` documents; eval-mode BatchNorm folded into the convs
    (w' = w * g/sqrt(var+eps), b' = beta - mean * g/sqrt(var+eps)), evaluated in fp64."""
    d = lambda n: sd["patch_embed." + n].detach().double().cpu()  # noqa: E731
    mean, std = torch.tensor(cfg.mean, dtype=torch.float64), torch.tensor(cfg.std, dtype=torch.float64)
    parts = [1.0 / (255.0 * std), -mean / std, torch.zeros(2, dtype=torch.float64)]
    for conv, bn in ((0, 1), (3, 4)):
        s = d(f"proj.{bn}.weight") / torch.sqrt(d(f"proj.{bn}.running_var") + eps_bn)
        w = d(f"proj.{conv}.weight") * s.view(-1, 1, 1, 1)                 # [co][ci][ky][kx]
        parts += [w.permute(1, 2, 3, 0).reshape(-1), d(f"proj.{bn}.bias") - d(f"proj.{bn}.running_mean") * s]
    w3 = d("proj.6.weight").reshape(cfg.embed, -1)                          # [co][ci]
    parts += [w3.t().reshape(-1), d("proj.6.bias"), d("norm.weight"), d("norm.bias")]
    return torch.cat([p.reshape(-1) for p in parts]).float().contiguous()


class HipSwin(torch.nn.Module):
    """`model` object for the reference Extractor seam: `HipSwin(cfg, state_dict)(tiles) -> fp16 [B, 768]`.

    `state_dict` is `_SwinTransformer.state_dict()` of the reference (ctranspath.pth["model"]); integer buffers
    (relative_position_index, attn_mask, num_batches_tracked) are ignored -- their content is a function of the
    geometry and is rebuilt here in MFMA lane order."""

    def __init__(self, cfg: SwinConfig, state_dict: dict[str, torch.Tensor], device: str | torch.device = "cuda",
                 act_dtype: torch.dtype = torch.float16, chunk: int = 1024):
        super().__init__()
        from . import ops
        self.cfg, self.act_dtype, self.chunk = cfg, act_dtype, int(chunk)
        self.device_ = torch.device(device)
        if self.device_.type != "cuda":
            raise RuntimeError("HipSwin runs on the GPU only (no CPU fallback)")
        _lib.lib()
        missing = [n for n, _ in swin_param_shapes(cfg) if n not in state_dict]
        if missing:
            raise KeyError(f"state_dict lacks {len(missing)} entries, e.g. {missing[:3]}")
        self._keep: list[torch.Tensor] = []
        self._ws: torch.Tensor | None = None
        dev = self.device_

        def f32(t):
            t = t.detach().to(dev, torch.float32).contiguous()
            self._keep.append(t)
            return t

        def act(w, ld=None):
            w = w.detach().to(dev, torch.float32)
            out = ops.cast_pad(w.reshape(w.shape[0], -1), ld or w.shape[1], act_dtype)
            self._keep.append(out)
            return out

        sd = state_dict
        self.stem = f32(pack_stem_params(sd, cfg))
        self.mask_bits = shift_mask_bits().to(dev)
        nblk = sum(cfg.depths)
        blocks = (_lib.SwinBlock * nblk)()
        i = 0
        for s, depth in enumerate(cfg.depths):
            Cs = cfg.embed << s
            kp = Cs
            for d in range(depth):
                g = lambda n: sd[f"layers.{s}.blocks.{d}.{n}"]  # noqa: E731
                b = blocks[i]
                i += 1
                b.ln1_w, b.ln1_b = f32(g("norm1.weight")).data_ptr(), f32(g("norm1.bias")).data_ptr()
                b.ln2_w, b.ln2_b = f32(g("norm2.weight")).data_ptr(), f32(g("norm2.bias")).data_ptr()
                b.qkv_w, b.qkv_b = act(g("attn.qkv.weight"), kp).data_ptr(), f32(g("attn.qkv.bias")).data_ptr()
                b.proj_w, b.proj_b = act(g("attn.proj.weight"), kp).data_ptr(), f32(g("attn.proj.bias")).data_ptr()
                b.fc1_w, b.fc1_b = act(g("mlp.fc1.weight"), kp).data_ptr(), f32(g("mlp.fc1.bias")).data_ptr()
                b.fc2_w, b.fc2_b = act(g("mlp.fc2.weight")).data_ptr(), f32(g("mlp.fc2.bias")).data_ptr()
                b.bias_lane = f32(rel_bias_lane_table(g("attn.relative_position_bias_table"))).data_ptr()
                if Cs == 192:        # stage 2: the fused MLP kernel streams a fragment-ordered image of its weights
                    pk = ops.swin_mlp192_pack(act(g("mlp.fc1.weight")), act(g("mlp.fc2.weight")))
                    self._keep.append(pk)
                    b.mlp_pack = pk.data_ptr()
                else:
                    b.mlp_pack = None
        merges = (_lib.SwinMerge * 3)()
        for s in range(len(cfg.depths) - 1):
            p = f"layers.{s}.downsample."
            merges[s].ln_w, merges[s].ln_b = f32(sd[p + "norm.weight"]).data_ptr(), f32(sd[p + "norm.bias"]).data_ptr()
            merges[s].red_w = act(sd[p + "reduction.weight"]).data_ptr()
        self.norm_w, self.norm_b = f32(sd["norm.weight"]), f32(sd["norm.bias"])
        self._blocks = blocks
        dep = (C.c_int * 4)(*(list(cfg.depths) + [0] * (4 - len(cfg.depths))))
        hds = (C.c_int * 4)(*(list(cfg.heads) + [0] * (4 - len(cfg.heads))))
        self._cfg_c = _lib.SwinCfg(cfg.img, cfg.embed, len(cfg.depths), dep, hds, ops.act_code(act_dtype), 1e-5)
        self._w_c = _lib.SwinWeights(self.stem.data_ptr(), C.cast(blocks, C.POINTER(_lib.SwinBlock)), nblk, merges,
                                     self.norm_w.data_ptr(), self.norm_b.data_ptr(), self.mask_bits.data_ptr())
        torch.cuda.synchronize(dev)

    def _as_u8_hwc(self, tiles: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        if tiles.dtype == torch.uint8:
            if tiles.dim() != 4 or tiles.shape[-1] != 3:
                raise ValueError(f"u8 tiles must be [B,H,W,3], got {tuple(tiles.shape)}")
            return tiles.contiguous()
        if tiles.dim() != 4 or tiles.shape[1] != 3:
            raise ValueError(f"float tiles must be [B,3,H,W], got {tuple(tiles.shape)}")
        # a float batch already went through ToDtype(scale=True) + Normalize (ctranspath.py:56-64): that map is a
        # bijection on u8 values, so undo it exactly and take the fused u8 path
        mean = torch.tensor(c.mean, device=tiles.device, dtype=torch.float32).view(1, 3, 1, 1)
        std = torch.tensor(c.std, device=tiles.device, dtype=torch.float32).view(1, 3, 1, 1)
        u8 = ((tiles.float() * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8)
        return u8.permute(0, 2, 3, 1).contiguous()

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor, return_f32: bool = False):
        if not tiles.is_cuda:
            raise RuntimeError("HipSwin.forward needs tiles on the GPU (no CPU fallback)")
        c = self.cfg
        tiles = self._as_u8_hwc(tiles)
        B = tiles.shape[0]
        if tiles.shape[1] != c.img or tiles.shape[2] != c.img:
            raise ValueError(f"expected {c.img}x{c.img} tiles, got {tuple(tiles.shape)}")
        feats = torch.empty(B, c.out_dim, dtype=torch.float16, device=tiles.device)
        f32o = torch.empty(B, c.out_dim, dtype=torch.float32, device=tiles.device) if return_f32 else None
        if B == 0:
            return (feats, f32o) if return_f32 else feats
        chunk = min(self.chunk, B)
        need = _lib.lib().amds_swin_workspace_bytes(C.byref(self._cfg_c), chunk)
        if need == 0:
            _lib.check(-1, "swin_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device_)
        rc = _lib.lib().amds_swin_forward(C.byref(self._cfg_c), C.byref(self._w_c), tiles.data_ptr(), feats.data_ptr(),
                                          f32o.data_ptr() if f32o is not None else None, B, chunk, self._ws.data_ptr(),
                                          self._ws.numel(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "swin_forward")
        return (feats, f32o) if return_f32 else feats

    def to(self, *args, **kwargs):  # weights are packed for one device at construction (preprocessing/__init__.py:243)
        return self


__all__ = ["SwinConfig", "SWIN_PRESETS", "HipSwin", "random_swin_state_dict", "swin_param_shapes",
           "rel_bias_lane_table", "shift_mask_lane_table", "shift_mask_bits", "pack_stem_params"]
