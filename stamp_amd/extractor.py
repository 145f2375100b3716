"""Extractor seam: the object `stamp preprocess` accepts via ``extract_(extractor=<Extractor>)``
(reference src/stamp/preprocessing/__init__.py:118, 237-238).

`Extractor` mirrors the reference dataclass field for field (src/stamp/preprocessing/extractor/__init__.py:17-28:
keyword-only ``model``, ``transform``, ``identifier``) so an instance built here can be handed to an unmodified
STAMP; when STAMP itself is importable, build `stamp.preprocessing.extractor.Extractor` with the same three values.

`transform` does NOT normalise on the host: it returns the decoded tile as a ``uint8 [H, W, 3]`` tensor; the
``(x/255-mean)/std`` step of the reference's transforms (h_optimus_0.py:22-30) is folded into the HIP patch
embedding.  The reference's loop only does ``model(tiles.to(device)).detach().half().cpu()`` on whatever the
transform produced (:324-325), so the dtype change is invisible to it.
"""
from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass

import numpy as np
import torch

from .swin import SWIN_PRESETS, HipSwin, SwinConfig
from .vit import PRESETS, HipViT, ViTConfig


@dataclass(frozen=True, kw_only=True)
class Extractor:
    model: torch.nn.Module
    transform: Callable[[object], torch.Tensor]
    identifier: str


def u8_tile_transform(img) -> torch.Tensor:
    """PIL.Image / ndarray tile -> uint8 [H, W, 3] tensor (no float conversion, no normalisation)."""
    arr = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img)
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError(f"expected an RGB uint8 tile, got {arr.dtype} {arr.shape}")
    return torch.from_numpy(np.ascontiguousarray(arr))


def hip_vit_extractor(name: str, state_dict: dict[str, torch.Tensor], *, identifier: str | None = None,
                      cfg: ViTConfig | None = None, device="cuda", act_dtype=torch.float16, chunk: int = 1020) -> Extractor:
    """Extractor whose model is the HIP tile encoder.  `name` is a key of `stamp_amd.vit.PRESETS`; `state_dict`
    uses timm VisionTransformer names (what the reference's factories load, e.g. uni2.py:32-34)."""
    cfg = cfg or PRESETS[name]
    model = HipViT(cfg, state_dict, device=device, act_dtype=act_dtype, chunk=chunk)
    return Extractor(model=model, transform=u8_tile_transform, identifier=identifier or f"amdstamp-{name}")


def hip_dinobloom_extractor(state_dict: dict[str, torch.Tensor], *, identifier: str = "dino-bloom", device="cuda", chunk: int = 1020) -> Extractor:
    """The reference's `dino_bloom()` factory (src/stamp/preprocessing/extractor/dinobloom.py:56-84) with the HIP trunk.  `state_dict`: the checkpoint's
    `"teacher"` entries without the `dino_head` / `ibot_head` keys and with the `"backbone."` prefix removed, as the reference prepares them (:39-46)."""
    return Extractor(model=HipViT(PRESETS["dinobloom_s"], state_dict, device=device, chunk=chunk), transform=u8_tile_transform, identifier=identifier)


def _cls_rows_f32(vit, tiles: torch.Tensor) -> torch.Tensor:
    """fp32 class rows [B, dim] of the final-norm'd token tensor.  The token tensor is asked for one encoder chunk at a time: [chunk, T, dim] fp32 is
    ~1 GB at 1020 tiles of a ViT-L, [B, T, dim] for a whole slide's tiles would not fit."""
    step = max(1, int(vit.chunk))
    rows = []
    for i in range(0, max(tiles.shape[0], 1), step):
        _, toks = vit(tiles[i:i + step], return_tokens=True)
        rows.append(toks[:, 0].contiguous())
        del toks
    return rows[0] if len(rows) == 1 else torch.cat(rows)


class HipKeep(torch.nn.Module):
    """`KEEPImageModel` of the reference (src/stamp/preprocessing/extractor/keep.py:25-50): timm ViT-L/16 trunk, then `visual_head` (Linear, GELU,
    Linear) and an L2 normalisation -- trunk = the ViT-L/16 preset, head = ONE library call in exact fp32 (`amds_proj_head_l2norm`) on the trunk's
    fp32 class row (the reference's head sees the trunk's fp32 output, keep.py:44-49; rounding it to half first would add 2^-11 in front of a GELU MLP).  `state_dict`: the checkpoint's `visual.*` / `visual_head.*` entries (:83-88); LayerScale keys named `.ls1.weight`
    are accepted like the reference's `_remap_layerscale_keys` does (:53-59).  Output fp32 [B, projection_dim] (the reference's loop casts to half)."""

    def __init__(self, state_dict: dict[str, torch.Tensor], *, device="cuda", chunk: int = 1020, vit_cfg: ViTConfig | None = None) -> None:
        super().__init__()
        sd = {}
        for k, v in state_dict.items():
            if ".ls1.weight" in k or ".ls2.weight" in k:
                k = k.replace(".weight", ".gamma")
            sd[k] = v
        trunk = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
        head = {k[len("visual_head."):]: v for k, v in sd.items() if k.startswith("visual_head.")}
        missing = [k for k in ("0.weight", "0.bias", "2.weight", "2.bias") if k not in head]
        if missing:
            raise KeyError(f"KEEP state_dict lacks visual_head.{missing}")
        self.vit = HipViT(vit_cfg or PRESETS["vit_large_patch16_224"], trunk, device=device, chunk=chunk)
        dev = self.vit.device_
        self._h = [head[k].detach().to(dev, torch.float32).contiguous() for k in ("0.weight", "0.bias", "2.weight", "2.bias")]
        self.proj_dim, self.in_dim = self._h[0].shape
        if self.in_dim != self.vit.cfg.dim or tuple(self._h[2].shape) != (self.proj_dim, self.proj_dim):
            raise ValueError(f"visual_head shapes {tuple(self._h[0].shape)}, {tuple(self._h[2].shape)} do not fit a {self.vit.cfg.dim}-d trunk")

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor) -> torch.Tensor:
        from . import _lib, ops
        feats = _cls_rows_f32(self.vit, tiles)                         # the class row, fp32 as the reference's head receives it (final norm applied)
        B, dev = feats.shape[0], feats.device
        out = torch.empty(B, self.proj_dim, dtype=torch.float32, device=dev)
        lib = _lib.lib()
        nb = lib.amds_proj_head_l2norm_workspace_bytes(B, self.in_dim, self.proj_dim)
        ws = ops.scratch("keep_head", dev, nb)
        w1, b1, w2, b2 = self._h
        _lib.check(lib.amds_proj_head_l2norm(feats.data_ptr(), ops._DT[feats.dtype], w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), B,
                                             self.in_dim, self.proj_dim, ws.data_ptr(), ws.numel(), ops._stream()), "proj_head_l2norm")
        return out


def hip_keep_extractor(state_dict: dict[str, torch.Tensor], *, identifier: str = "keep", device="cuda", chunk: int = 1020) -> Extractor:
    """The reference's `keep()` factory (keep.py:95-116); its transform (Resize(224, bicubic) + CenterCrop(224), :99-106) is the identity on 224-pixel tiles."""
    return Extractor(model=HipKeep(state_dict, device=device, chunk=chunk), transform=u8_tile_transform, identifier=identifier)


def clip_vision_to_timm_names(sd: dict[str, torch.Tensor]) -> tuple[dict[str, torch.Tensor], torch.Tensor]:
    """HF `CLIPModel` state_dict -> (the vision tower under timm's VisionTransformer names, visual_projection.weight).  Data movement only: q / k / v
    projections concatenated into `attn.qkv`, the bias-free patch convolution given a zero bias, embeddings reshaped."""
    p = "vision_model."
    D = sd[p + "embeddings.class_embedding"].numel()
    out = {"patch_embed.proj.weight": sd[p + "embeddings.patch_embedding.weight"], "patch_embed.proj.bias": torch.zeros(D),
           "cls_token": sd[p + "embeddings.class_embedding"].reshape(1, 1, D), "pos_embed": sd[p + "embeddings.position_embedding.weight"].unsqueeze(0),
           "norm_pre.weight": sd[p + "pre_layrnorm.weight"], "norm_pre.bias": sd[p + "pre_layrnorm.bias"],
           "norm.weight": sd[p + "post_layernorm.weight"], "norm.bias": sd[p + "post_layernorm.bias"]}
    l = 0
    while f"{p}encoder.layers.{l}.layer_norm1.weight" in sd:
        q, b = f"{p}encoder.layers.{l}.", f"blocks.{l}."
        for a, t in (("layer_norm1", "norm1"), ("layer_norm2", "norm2"), ("self_attn.out_proj", "attn.proj"), ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            out[b + t + ".weight"], out[b + t + ".bias"] = sd[q + a + ".weight"], sd[q + a + ".bias"]
        out[b + "attn.qkv.weight"] = torch.cat([sd[q + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v")], dim=0)
        out[b + "attn.qkv.bias"] = torch.cat([sd[q + f"self_attn.{n}_proj.bias"] for n in ("q", "k", "v")], dim=0)
        l += 1
    return out, sd["visual_projection.weight"]


class HipPlip(torch.nn.Module):
    """`PLIP` of the reference (src/stamp/preprocessing/extractor/plip.py:16-22: `CLIPModel.get_image_features`): CLIP's vision tower on the HIP tile
    encoder (quick_gelu MLP, pre-LayerNorm: `ViTConfig(mlp="quick_gelu", pre_norm=True)`), then `visual_projection` in exact fp32 on the fp32 class
    row.  `state_dict`: the HF `CLIPModel`'s (text tower entries are ignored).  Output fp32 [B, projection_dim]."""

    def __init__(self, state_dict: dict[str, torch.Tensor], *, device="cuda", chunk: int = 1020, vit_cfg: ViTConfig | None = None) -> None:
        super().__init__()
        vsd, proj = clip_vision_to_timm_names({k: v for k, v in state_dict.items() if k.startswith("vision_model.") or k.startswith("visual_projection.")})
        self.vit = HipViT(vit_cfg or PRESETS["plip"], vsd, device=device, chunk=chunk)
        self.proj = proj.detach().to(self.vit.device_, torch.float32).contiguous()
        if self.proj.shape[1] != self.vit.cfg.dim:
            raise ValueError(f"visual_projection takes {self.proj.shape[1]}-d features, the tower gives {self.vit.cfg.dim}")

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor) -> torch.Tensor:
        from . import ops
        return ops.linear_f32(_cls_rows_f32(self.vit, tiles), self.proj, None)      # class rows in fp32, post_layernorm applied


def hip_plip_extractor(state_dict: dict[str, torch.Tensor], *, identifier: str = "plip", device="cuda", chunk: int = 1020) -> Extractor:
    """The reference's `plip()` factory (plip.py:25-40); Resize(224) is the identity on 224-pixel tiles, ToTensor + Normalize are folded into the patch
    embedding like every preset's."""
    return Extractor(model=HipPlip(state_dict, device=device, chunk=chunk), transform=u8_tile_transform, identifier=identifier)


class ResizeCropThenModel(torch.nn.Module):
    """`model(resize_center_crop(tiles))`: a tile transform that is not the identity on the tile size, done on the GPU in front of the HIP model
    (Pillow's bicubic resample bit for bit + torchvision's crop offset: `stamp_amd.tiling.resize_center_crop`)."""

    def __init__(self, model: torch.nn.Module, resized: int, crop: int) -> None:
        super().__init__()
        self.model, self.resized, self.crop = model, int(resized), int(crop)

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor) -> torch.Tensor:
        from .tiling import resize_center_crop
        return self.model(resize_center_crop(tiles, self.resized, self.crop))


def hip_gigapath_extractor(state_dict: dict[str, torch.Tensor], *, identifier: str = "gigapath", device="cuda", chunk: int = 512) -> Extractor:
    """The reference's `gigapath()` factory (src/stamp/preprocessing/extractor/gigapath.py:14-34) with the HIP trunk: Resize(256, BICUBIC) +
    CenterCrop(224) on the GPU (bit-identical with what torchvision does to the PIL tile), then the ViT-g/16 preset; ToTensor + Normalize are
    folded into the patch embedding as for every other preset."""
    trunk = HipViT(PRESETS["gigapath"], state_dict, device=device, chunk=chunk)
    return Extractor(model=ResizeCropThenModel(trunk, 256, 224), transform=u8_tile_transform, identifier=identifier)


def hip_ticon_extractor(vit_state_dict: dict[str, torch.Tensor], ticon_state_dict: dict[str, torch.Tensor], *, identifier: str = "ticon", device="cuda",
                        chunk: int = 512) -> Extractor:
    """The reference's `ticon()` factory (src/stamp/preprocessing/extractor/ticon.py:721-741) with the HIP model: H-optimus-1 trunk + TICON on every
    tile alone (`stamp_amd.ticon.HipHOptimusTicon`); the normalisation constants of the reference's transform (:729-732) are the trunk preset's."""
    from .ticon import HipHOptimusTicon
    return Extractor(model=HipHOptimusTicon(vit_state_dict, ticon_state_dict, device=device, chunk=chunk), transform=u8_tile_transform, identifier=identifier)


def hip_ctranspath_extractor(state_dict: dict[str, torch.Tensor], *, identifier: str = "ctranspath", cfg: SwinConfig | None = None,
                             device="cuda", act_dtype=torch.float16, chunk: int = 1024) -> Extractor:
    """The reference's `ctranspath()` / `chief_ctranspath()` factories (src/stamp/preprocessing/extractor/ctranspath.py:34-70,
    chief_ctranspath.py:20-57) with the HIP model: `state_dict` is what they pass to `model.load_state_dict`
    (`torch.load("ctranspath.pth")["model"]`, after the sha256 check they do), `identifier` the ExtractorName value
    ("ctranspath" or "chief-ctranspath") so that the CHIEF slide encoder accepts the feature files."""
    model = HipSwin(cfg or SWIN_PRESETS["ctranspath"], state_dict, device=device, act_dtype=act_dtype, chunk=chunk)
    return Extractor(model=model, transform=u8_tile_transform, identifier=identifier)


def has_enough_texture(tiles_u8: torch.Tensor, cutoff: float = 0.02) -> torch.Tensor:
    """Batched `_has_enough_texture` (reference src/stamp/preprocessing/tiling.py:280-291) for decoded tiles already on
    the GPU: bool [B], True = keep (Canny edge fraction >= canny_cutoff; 40 / 100 are the reference's hard-coded thresholds)."""
    from . import ops
    return ops.tile_edge_fraction(tiles_u8, 40, 100) >= cutoff


def macenko_normalize(tiles_u8: torch.Tensor, *, Io: float = 240.0, alpha: float = 1.0, beta: float = 0.15, return_fit: bool = False):
    """OPTIONAL Macenko stain normalisation of decoded tiles on the GPU: u8 [B, H, W, 3] -> u8 [B, H, W, 3] (+ per-tile fit [B, 8]).
    Not part of the reference's pipeline -- named by BASELINE.json's north_star, off by default, parity unpinned."""
    from . import _lib, ops
    if not tiles_u8.is_cuda:
        raise RuntimeError("macenko_normalize needs tiles on the GPU (no CPU fallback)")
    if tiles_u8.dtype != torch.uint8 or tiles_u8.dim() != 4 or tiles_u8.shape[-1] != 3:
        raise ValueError(f"expected u8 [B, H, W, 3], got {tiles_u8.dtype} {tuple(tiles_u8.shape)}")
    t = tiles_u8.contiguous()
    out = torch.empty_like(t)
    fit = torch.empty(t.shape[0], 8, dtype=torch.float32, device=t.device) if return_fit else None
    _lib.check(_lib.lib().amds_macenko_normalize_u8(t.data_ptr(), out.data_ptr(), None if fit is None else fit.data_ptr(), t.shape[0], t.shape[1], t.shape[2],
                                                    Io, alpha, beta, ops._stream()), "macenko_normalize")
    return (out, fit) if return_fit else out


class TilePipeline:
    """The per-slide hot loop of the reference (src/stamp/preprocessing/__init__.py:315-327: batches from the DataLoader ->
    ``model(tiles.to(device))`` -> ``.detach().half().cpu()``) as a three-stage pipeline on three HIP streams:

        pinned host u8 tiles --H2D (copy-in stream)--> device buffer A/B --encode (compute stream)--> fp16 features
                                                     --D2H (copy-out stream)--> pinned host feature rows

    Two device tile buffers alternate, so the PCIe copy of batch i+1 runs under the encoder of batch i (a 1020-tile batch is
    154 MB = ~3 ms at PCIe Gen5 x16 against ~190 ms of ViT-L/14 compute); features leave over their own stream.  Ordering is by
    events only -- the host never blocks until `finish()`.  The reference's batch of 64 is a host-RAM choice; any batch size is
    accepted (the model chunks internally) and `submit` may be fed straight from a decoder thread.
    """

    def __init__(self, model: torch.nn.Module, *, batch_size: int = 1020, device="cuda", tile_shape=(224, 224, 3), feat_dim: int | None = None) -> None:
        self.model = model
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("TilePipeline runs on the GPU only (no CPU fallback)")
        self.bs = int(batch_size)
        self.dim = feat_dim or getattr(model.cfg, "out_dim", None) or model.cfg.dim
        self.s_in, self.s_out = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self.buf = [torch.empty(self.bs, *tile_shape, dtype=torch.uint8, device=self.dev) for _ in range(2)]
        self.ev_in = [torch.cuda.Event() for _ in range(2)]
        self.ev_free = [torch.cuda.Event() for _ in range(2)]          # the encoder is done reading buffer k
        self.ev_feat = [torch.cuda.Event() for _ in range(2)]
        self.feats = [None, None]
        self.n = 0

    @torch.inference_mode()
    def submit(self, tiles_host: torch.Tensor, out_host: torch.Tensor) -> None:
        """tiles_host: u8 [b <= batch_size, H, W, 3] (pinned for a truly asynchronous copy); out_host: fp16 [b, dim] (pinned)."""
        b = tiles_host.shape[0]
        if b == 0:
            return
        if b > self.bs:
            raise ValueError(f"batch of {b} tiles exceeds the pipeline's batch_size {self.bs}")
        k = self.n & 1
        comp = torch.cuda.current_stream(self.dev)
        if self.n >= 2:
            self.s_in.wait_event(self.ev_free[k])
        with torch.cuda.stream(self.s_in):
            self.buf[k][:b].copy_(tiles_host, non_blocking=True)
            self.ev_in[k].record(self.s_in)
        comp.wait_event(self.ev_in[k])
        if self.n >= 2:
            comp.wait_event(self.ev_feat[k])                             # the previous feature block of this slot has left the device
        f = self.model(self.buf[k][:b])
        self.ev_free[k].record(comp)
        self.feats[k] = f
        self.s_out.wait_event(self.ev_free[k])
        with torch.cuda.stream(self.s_out):
            out_host.copy_(f.detach().half() if f.dtype != torch.float16 else f, non_blocking=True)
            self.ev_feat[k].record(self.s_out)
        self.n += 1

    def finish(self) -> None:
        self.s_out.synchronize()
        torch.cuda.current_stream(self.dev).synchronize()


@torch.inference_mode()
def extract_tiles(extractor: Extractor, tiles_u8: torch.Tensor, batch_size: int = 1020, device="cuda", pin: bool = True) -> torch.Tensor:
    """The reference's per-slide hot loop (preprocessing/__init__.py:322-327) on an in-memory stack of decoded tiles ->
    fp16 features on the host, through `TilePipeline` (double-buffered H2D, encode, D2H overlapped)."""
    model = extractor.model
    dim = getattr(model.cfg, "out_dim", None) or model.cfg.dim
    n = tiles_u8.shape[0]
    out = torch.empty(n, dim, dtype=torch.float16)
    if n == 0:
        return out
    if pin:
        out = out.pin_memory()
        if not tiles_u8.is_pinned():
            tiles_u8 = tiles_u8.contiguous().pin_memory()
    pipe = TilePipeline(model, batch_size=min(batch_size, n), device=device, tile_shape=tuple(tiles_u8.shape[1:]), feat_dim=dim)
    for i in range(0, n, pipe.bs):
        pipe.submit(tiles_u8[i:i + pipe.bs], out[i:i + pipe.bs])
    pipe.finish()
    return out
