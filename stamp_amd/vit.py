"""Tile-encoder (timm-style VisionTransformer) on the HIP path.

`HipViT` is the `model` of a STAMP ``Extractor`` (reference
src/stamp/preprocessing/extractor/__init__.py:17-28): called as ``model(tiles)`` under
``torch.inference_mode()`` and expected to return ``[B, D]`` features that the caller casts with ``.half()``
(reference src/stamp/preprocessing/__init__.py:324-325).  It consumes the *decoded u8 tile* directly -- the
``(x/255-mean)/std`` transform of the reference's extractors (e.g. h_optimus_0.py:22-30) is folded into the
patch-embedding weights -- and returns fp16 CLS features straight from the final-LayerNorm kernel.

Weights come in as a timm ``VisionTransformer`` state_dict (the naming every ViT factory of the reference ends
up with: virchow2.py:34-39, uni2.py:17-34, reddino.py:40-45, uni.py:26-31).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field, replace

import torch
from torch import nn

from . import _lib, ops

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass(frozen=True)
class ViTConfig:
    img: int = 224
    patch: int = 14
    dim: int = 1024
    depth: int = 24
    heads: int = 16
    hidden: int = 4096           # fc2 input width (for SwiGLUPacked: half of fc1's output width)
    mlp: str = "gelu"            # "gelu" (timm Mlp) | "swiglu" (timm SwiGLUPacked + SiLU) | "quick_gelu" (HF CLIP: x * sigmoid(1.702 x))
    pre_norm: bool = False       # LayerNorm on the embedded tokens before the first block (HF CLIP `pre_layrnorm` / timm `norm_pre`)
    reg_tokens: int = 0
    no_embed_class: bool = False
    layerscale: bool = True      # timm init_values is not None
    ln_eps: float = 1e-6
    mean: tuple = IMAGENET_MEAN
    std: tuple = IMAGENET_STD

    @property
    def n_patches(self) -> int:
        return (self.img // self.patch) ** 2

    @property
    def n_prefix(self) -> int:
        return 1 + self.reg_tokens

    @property
    def tokens(self) -> int:
        return self.n_patches + self.n_prefix

    @property
    def hidden_pad(self) -> int:
        return (self.hidden + 63) // 64 * 64

    @property
    def kp(self) -> int:
        return (3 * self.patch * self.patch + 63) // 64 * 64

    def matmul_flops_per_tile(self) -> float:
        """2 FLOP per MAC, matmuls only: patch-embed + depth x (linears + attention)."""
        T, D, np_ = self.tokens, self.dim, self.n_patches
        fc1_out = self.hidden * (2 if self.mlp == "swiglu" else 1)
        lin = 2 * T * (D * 3 * D + D * D + D * fc1_out + self.hidden * D)
        att = 2 * 2 * T * T * D
        return 2 * np_ * 3 * self.patch ** 2 * D + self.depth * (lin + att)

    def matmul_flops_skipped_by_cls_tail(self) -> float:
        """Products of the LAST block that the class-row tail (HipViT(cls_tail=True), include/amdstamp.h `amds_vit_weights.cls_tail`) never
        computes: the q / proj / fc1 / fc2 rows and the attention of the T - 1 tokens nothing reads."""
        T, D = self.tokens, self.dim
        fc1_out = self.hidden * (2 if self.mlp == "swiglu" else 1)
        return 2 * (T - 1) * (D * D + D * D + D * fc1_out + self.hidden * D) + 2 * 2 * T * (T - 1) * D


PRESETS: dict[str, ViTConfig] = {
    # DINOv2-style ViT-L/14 (reference: RedDino-large, reddino.py:40-45); the BASELINE.json headline shape
    "vit_large_patch14_224": ViTConfig(),
    # UNI2-h, fully specified in-tree (reference uni2.py:17-31)
    "uni2_h": ViTConfig(dim=1536, depth=24, heads=24, hidden=4096, mlp="swiglu", reg_tokens=8, no_embed_class=True),
    # Virchow2 = ViT-H/14, SwiGLUPacked, 4 register tokens (reference virchow2.py:34-39; D=1280 pinned by
    # tests/test_encoders.py:31, the other hyper-parameters are the public model card's: SURVEY.md F4)
    "virchow2": ViTConfig(dim=1280, depth=32, heads=16, hidden=3416, mlp="swiglu", reg_tokens=4, no_embed_class=False),
    # Virchow (v1) = the same ViT-H/14 SwiGLU trunk without register tokens; used with `HipViTClsMean` (reference virchow_full.py,
    # virchow.py: the factories pass mlp_layer=SwiGLUPacked, act_layer=SiLU; remaining hyper-parameters from the model card)
    "virchow": ViTConfig(dim=1280, depth=32, heads=16, hidden=3416, mlp="swiglu", reg_tokens=0, no_embed_class=False),
    "test_tiny_hd80": ViTConfig(dim=640, depth=2, heads=8, hidden=696, mlp="swiglu", reg_tokens=4, no_embed_class=False),
    # H-optimus-0 / H-optimus-1 = timm vit_giant_patch14_reg4_dinov2 (reference h_optimus_0.py:15-20, h_optimus_1.py:15-20: the
    # factory only passes init_values / dynamic_img_size; width 1536, depth 40, 24 heads, SwiGLUPacked 8192 -> 4096, 4 register
    # tokens, no_embed_class are the timm architecture's published hyper-parameters, checked against the state_dict shapes at
    # pack time).  Mean / std are in-tree (h_optimus_0.py:26-28).
    "h_optimus_0": ViTConfig(dim=1536, depth=40, heads=24, hidden=4096, mlp="swiglu", reg_tokens=4, no_embed_class=True,
                             mean=(0.707223, 0.578729, 0.703617), std=(0.211883, 0.230117, 0.177517)),
    # Prov-GigaPath's tile encoder = timm vit_giant_patch14_dinov2 with 16-pixel patches (reference gigapath.py:18: the factory passes the hub name
    # only; width 1536, depth 40, 24 heads, SwiGLUPacked 8192 -> 4096, no register tokens are the model card's hyper-parameters, checked against
    # the state_dict shapes at pack time).  Its transform is NOT the identity on 224-pixel tiles: Resize(256, bicubic) + CenterCrop(224)
    # (gigapath.py:21-28) -- `stamp_amd.extractor.hip_gigapath_extractor` puts `tiling.resize_center_crop` in front of the trunk.
    "gigapath": ViTConfig(patch=16, dim=1536, depth=40, heads=24, hidden=4096, mlp="swiglu", reg_tokens=0, no_embed_class=False),
    # DinoBloom-S = DINOv2 ViT-S/14 fine-tuned (reference dinobloom.py:25-53: `torch.hub.load("facebookresearch/dinov2", "dinov2_vits14")`, embed size 384
    # from `_embed_sizes`, pos_embed re-made for 257 tokens; depth 12, 6 heads, GELU MLP 1536, LayerScale are DINOv2-S's published hyper-parameters,
    # checked against the state_dict at pack time; the hub model's state_dict uses the same names as timm's, plus `mask_token`, ignored here).
    "dinobloom_s": ViTConfig(dim=384, depth=12, heads=6, hidden=1536),
    # PLIP = HF CLIP ViT-B/32's vision tower (reference plip.py:26: `CLIPModel.from_pretrained("vinid/plip")`; width 768, 12 layers, 12 heads, MLP 3072
    # with quick_gelu, 32-pixel patches, pre-LayerNorm, no LayerScale, eps 1e-5 are the CLIP ViT-B/32 configuration, checked against the state_dict
    # at pack time; CLIP's normalisation constants are in-tree, plip.py:31-32).  `stamp_amd.extractor.HipPlip` adds the visual projection.
    "plip": ViTConfig(patch=32, dim=768, depth=12, heads=12, hidden=3072, mlp="quick_gelu", pre_norm=True, layerscale=False, ln_eps=1e-5,
                      mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)),
    # ViT-L/16 (reference UNI, uni.py:26-31)
    "vit_large_patch16_224": ViTConfig(patch=16),
    # small shapes for tests
    "test_tiny": ViTConfig(dim=128, depth=2, heads=2, hidden=256),
    "test_tiny_swiglu": ViTConfig(dim=128, depth=2, heads=2, hidden=192, mlp="swiglu", reg_tokens=4, no_embed_class=True),
    # the smallest shape the DEFAULT path of the full-size presets takes (LayerNorm folded, hi | lo residual planes, class-row tail: dim % 256 == 0)
    "test_tiny_fold": ViTConfig(dim=256, depth=4, heads=4, hidden=512),
}


def expected_state_dict_shapes(cfg: ViTConfig) -> dict[str, tuple]:
    """Parameter names and shapes of the timm `VisionTransformer` this config describes (num_classes = 0), i.e. what the reference's
    factories hand to `load_state_dict` (uni2.py:32-34, virchow2.py:34-39, h_optimus_0.py:15-20, reddino.py:40-45)."""
    D, p = cfg.dim, cfg.patch
    fc1 = cfg.hidden * (2 if cfg.mlp == "swiglu" else 1)
    sh: dict[str, tuple] = {"patch_embed.proj.weight": (D, 3, p, p), "patch_embed.proj.bias": (D,), "cls_token": (1, 1, D),
                            "pos_embed": (1, cfg.n_patches + (0 if cfg.no_embed_class else cfg.n_prefix), D), "norm.weight": (D,), "norm.bias": (D,)}
    if cfg.reg_tokens:
        sh["reg_token"] = (1, cfg.reg_tokens, D)
    if cfg.pre_norm:
        sh.update({"norm_pre.weight": (D,), "norm_pre.bias": (D,)})
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        sh.update({b + "norm1.weight": (D,), b + "norm1.bias": (D,), b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
                   b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,), b + "norm2.weight": (D,), b + "norm2.bias": (D,),
                   b + "mlp.fc1.weight": (fc1, D), b + "mlp.fc1.bias": (fc1,), b + "mlp.fc2.weight": (D, cfg.hidden), b + "mlp.fc2.bias": (D,)})
        if cfg.layerscale:
            sh.update({b + "ls1.gamma": (D,), b + "ls2.gamma": (D,)})
    return sh


# entries of real checkpoints that carry no arithmetic on this path (classifier head when num_classes > 0, DINOv2's mask token, pooling norm)
_IGNORED_KEYS = ("head.", "fc_norm.", "mask_token", "head_drop", "attn_pool.")


def validate_state_dict(cfg: ViTConfig, sd: dict[str, torch.Tensor]) -> None:
    """Fail loudly, before any packing, when a checkpoint does not describe `cfg`: every expected key present with the expected shape, no
    unexpected parameter (a stray key usually means a different architecture: qk-norm, a different MLP, an untied pos_embed layout)."""
    want = expected_state_dict_shapes(cfg)
    missing = [k for k in want if k not in sd]
    wrong = [(k, tuple(sd[k].shape), v) for k, v in want.items() if k in sd and tuple(sd[k].shape) != v]
    extra = [k for k in sd if k not in want and not k.startswith(_IGNORED_KEYS)]
    if missing or wrong or extra:
        msg = ["state_dict does not match the ViTConfig:"]
        if missing:
            msg.append(f"  missing ({len(missing)}): {missing[:6]}{' ...' if len(missing) > 6 else ''}")
        if wrong:
            msg.append("  wrong shape: " + "; ".join(f"{k} is {a}, expected {b}" for k, a, b in wrong[:6]) + (" ..." if len(wrong) > 6 else ""))
        if extra:
            msg.append(f"  unexpected ({len(extra)}): {extra[:6]}{' ...' if len(extra) > 6 else ''}")
        raise ValueError("\n".join(msg))


def packed_weight_bytes(cfg: ViTConfig) -> int:
    D = cfg.dim
    fc1 = cfg.hidden_pad * (2 if cfg.mlp == "swiglu" else 1)
    return 2 * (D * cfg.kp + cfg.depth * (3 * D * D + D * D + fc1 * D + D * cfg.hidden_pad))


def host_weights(cfg: ViTConfig, sd: dict[str, torch.Tensor]):
    """(amds_vit_host_weights, keep-alive list): the checkpoint as contiguous host fp32 tensors, addressed by pointer."""
    keep: list[torch.Tensor] = []

    def ptr(key: str):
        t = sd[key].detach().to("cpu", torch.float32).contiguous()
        keep.append(t)
        return t.data_ptr()

    blocks = (_lib.VitHostBlock * cfg.depth)()
    for i in range(cfg.depth):
        b, pre = blocks[i], f"blocks.{i}."
        b.norm1_w, b.norm1_b, b.norm2_w, b.norm2_b = ptr(pre + "norm1.weight"), ptr(pre + "norm1.bias"), ptr(pre + "norm2.weight"), ptr(pre + "norm2.bias")
        b.qkv_w, b.qkv_b, b.proj_w, b.proj_b = ptr(pre + "attn.qkv.weight"), ptr(pre + "attn.qkv.bias"), ptr(pre + "attn.proj.weight"), ptr(pre + "attn.proj.bias")
        b.fc1_w, b.fc1_b, b.fc2_w, b.fc2_b = ptr(pre + "mlp.fc1.weight"), ptr(pre + "mlp.fc1.bias"), ptr(pre + "mlp.fc2.weight"), ptr(pre + "mlp.fc2.bias")
        b.ls1, b.ls2 = (ptr(pre + "ls1.gamma"), ptr(pre + "ls2.gamma")) if cfg.layerscale else (None, None)
    keep.append(blocks)
    hw = _lib.VitHostWeights(ptr("patch_embed.proj.weight"), ptr("patch_embed.proj.bias"), ptr("cls_token"), ptr("reg_token") if cfg.reg_tokens else None,
                             ptr("pos_embed"), C.cast(blocks, C.POINTER(_lib.VitHostBlock)), ptr("norm.weight"), ptr("norm.bias"), cfg.hidden,
                             1 if cfg.no_embed_class else 0, (C.c_double * 3)(*cfg.mean), (C.c_double * 3)(*cfg.std))
    return hw, keep


class FeatureRangeError(RuntimeError):
    """The features of a call hold non-finite values: an intermediate of the network left the range of the 16-bit activation format and no
    safer packing is left to fall back to.  Raised instead of returning NaNs -- STAMP's per-slide try/except
    (src/stamp/preprocessing/__init__.py:328-336) then logs and skips the slide instead of writing them into an .h5."""


class HipViT(nn.Module):
    """timm VisionTransformer forward on libamdstamp; eval/inference only (tile extraction never trains)."""

    def __init__(self, cfg: ViTConfig, state_dict: dict[str, torch.Tensor], *, device="cuda",
                 act_dtype: torch.dtype = torch.float16, chunk: int = 1020, ln_fold: bool | None = None,
                 patch_split: bool | None = None, exact: bool = False, fp8: bool = False, cls_tail: bool | None = None,
                 check: str = "fallback") -> None:
        """check: what happens when a call's features come out non-finite (`amds_check_finite`, one 4-byte read-back per call; the fast path
        keeps the residual stream as two fp16 planes and un-normalised fp16 rows as GEMM operands, so a checkpoint whose residual stream
        leaves +-65504 -- massive-activation channels -- overflows there, and every overflow reaches the class row as NaN):
          "fallback" (default)  re-pack on a safer level and re-run the call, permanently for this object, with one warning:
                                level 1 = LayerNorm un-folded, fp32 residual rows (same 16-bit operands otherwise);
                                level 2 = bf16 activations on top (fp32 range, 8 bits of mantissa: outside the 1e-3 parity bar, said so in the warning);
                                then FeatureRangeError.
          "raise"               FeatureRangeError at once (the slide is skipped by STAMP's per-slide try/except).
          "off"                 no check (benchmark A/B only).
        The pipelined `preprocess.extract_slide` sets `defer_check` while it runs and checks the whole slide's features once, before the file is written.

        exact=True (opt-in): the class-token row -- the only row the reference stores, `model(tiles)[:, 0].half()` -- is ALSO carried on
        an exact-fp32 class stream (fp32 MFMA, the original un-folded fp32 weights: include/amdstamp.h `amds_vit_exact_block`,
        csrc/vit_exact.hip) and written over the main path's class rows after every sub-layer.  Costs the fp32 weights in HBM (4 bytes per
        parameter of q / proj / fc1 / fc2) and ~10 % of the throughput; lowers the stored feature's error ~3x.

        cls_tail (default on; AMDS_VIT_CLS_TAIL=0 / cls_tail=False off): when only the class features are asked for -- what `extract_` stores,
        `model(tiles)[:, 0]`, src/stamp/preprocessing/__init__.py:324-325 -- the last block computes keys / values for all tokens and then the
        class row's own chain only (fp32, the kernels of the exact path; include/amdstamp.h `amds_vit_weights.cls_tail`): the other rows of
        that block are read by nothing.  Costs the last block's fp32 rows in HBM (ViT-L/14: 50 MB); `return_tokens=True` runs the full block."""
        super().__init__()
        if cfg.dim % cfg.heads or cfg.dim // cfg.heads not in (64, 80):
            raise ValueError(f"head_dim must be 64 or 80 (dim={cfg.dim}, heads={cfg.heads})")
        if check not in ("fallback", "raise", "off"):
            raise ValueError(f"check must be 'fallback', 'raise' or 'off', not {check!r}")
        self.cfg = cfg
        self.act_dtype = act_dtype
        self.chunk = int(chunk)
        self.check = check
        self.defer_check = False        # set by a caller that checks the features itself, once, later (preprocess.extract_slide)
        self.safe_level = 0             # 0 = as constructed; 1 / 2: see `check`
        object.__setattr__(self, "_safe", None)     # "HipViT | None"; kept out of nn.Module's child registry: modules() / state_dict() see ONE ViT
        self._sd_ref = state_dict       # kept (by reference) for the safe re-pack
        self._ctor = dict(device=device, chunk=chunk, patch_split=patch_split, exact=exact, cls_tail=cls_tail)
        self.overlap = False            # two chunks in flight on two streams (amds_vit_forward_overlapped)
        self.device_ = torch.device(device)
        if self.device_.type != "cuda":
            raise RuntimeError("HipViT runs on the GPU only (no CPU fallback)")
        _lib.lib()  # fail early and loudly if the extension is missing
        _lib.ctx(self.device_.index if self.device_.index is not None else torch.cuda.current_device())   # side stream of the ragged-tail schedule
        self._ws: torch.Tensor | None = None
        # LayerNorm folded into the qkv / fc1 GEMMs (include/amdstamp.h, amds_gemm_lnfold): default on where the shapes allow it
        # (every preset); AMDS_VIT_LNFOLD=0 or ln_fold=False packs the plain weights and runs the stand-alone LayerNorm kernels (A/B).
        n_fc1 = cfg.hidden_pad * (2 if cfg.mlp == "swiglu" else 1)
        can_fold = cfg.dim % 256 == 0 and n_fc1 % 256 == 0
        if ln_fold is None:
            ln_fold = can_fold and cfg.mlp != "quick_gelu" and os.environ.get("AMDS_VIT_LNFOLD", "1") != "0"
        if ln_fold and not can_fold:
            raise ValueError(f"ln_fold needs dim % 256 == 0 and fc1 rows % 256 == 0 (dim={cfg.dim}, fc1 rows={n_fc1})")
        self.ln_fold = bool(ln_fold)
        # Patch-embedding weight as a 16-bit [hi | lo] pair (include/amdstamp.h, amds_vit_weights.patch_lo_shift): default on -- the weight with
        # the tile transform folded in multiplies RAW 0..255 values, so its rounding is the largest single error source of the whole path
        #  and the fix costs 0.2 % of the flops.  AMDS_VIT_PATCH_SPLIT=0 / patch_split=False: the single-rounded weight (A/B).
        if patch_split is None:
            patch_split = os.environ.get("AMDS_VIT_PATCH_SPLIT", "1") != "0"
        self.patch_lo_shift = (11 if act_dtype == torch.float16 else 8) if patch_split else 0
        self.exact = bool(exact)
        self.fp8 = bool(fp8)
        if cls_tail is None:
            cls_tail = os.environ.get("AMDS_VIT_CLS_TAIL", "1") != "0"
        self.cls_tail = bool(cls_tail) and cfg.mlp != "quick_gelu"
        if cfg.mlp == "quick_gelu":          # CLIP's MLP: fc1 (plain bias epilogue) -> activation pass -> fc2; plain packing only
            if exact or fp8 or ln_fold:
                raise ValueError("a quick_gelu (CLIP) trunk runs on the plain packing only: no ln_fold, exact or fp8")
            self.ln_fold = False
        if self.fp8:
            if act_dtype != torch.float16 or exact or cfg.dim % 256 or cfg.hidden % 256 or cfg.mlp == "quick_gelu":
                raise ValueError("fp8=True needs fp16 activations, exact=False, and dim / hidden multiples of 256 (ViT-L, UNI2-h, H-optimus; not Virchow's 3416)")
            self.ln_fold = False             # the fp8 chain normalises, THEN quantises: plain packing
        self._pack(state_dict)
        if self.fp8:
            self._pack_fp8(state_dict)

    # -- weight packing (one time): in the library (amds_vit_pack, csrc/vit_pack.hip) -------------------------------
    def _pack(self, sd: dict[str, torch.Tensor]) -> None:
        """Hands the checkpoint's tensors to `amds_vit_pack` as host fp32 pointers; every fold / rounding / padding / interleave happens in the
        library (include/amdstamp.h), so a non-Python host packs exactly the same image."""
        c = self.cfg
        validate_state_dict(c, sd)
        hw, keep = host_weights(c, sd)
        flags = (_lib.PACK_LNFOLD if self.ln_fold else 0) | (_lib.PACK_PATCH_SPLIT if self.patch_lo_shift else 0) | (_lib.PACK_EXACT if self.exact else 0) \
            | (_lib.PACK_CLS_TAIL if self.cls_tail else 0)
        self._cfg_c = _lib.VitCfg(c.img, c.patch, c.dim, c.depth, c.heads, c.hidden_pad, c.n_prefix, {"gelu": 0, "swiglu": 1, "quick_gelu": 2}[c.mlp],
                                  1 if c.layerscale else 0, ops.act_code(self.act_dtype), c.ln_eps)
        lib = _lib.lib()
        need = lib.amds_vit_pack_bytes(C.byref(self._cfg_c), C.byref(hw), flags)
        if need == 0:
            _lib.check(-1, "vit_pack_bytes")
        self._image = torch.empty(need, dtype=torch.uint8, device=self.device_)
        self._w_c = _lib.VitWeights()
        self._blocks = (_lib.VitBlock * c.depth)()
        self._exact = (_lib.VitExactBlock * c.depth)() if (self.exact or self.cls_tail) else None
        with torch.cuda.device(self.device_):
            rc = lib.amds_vit_pack(C.byref(self._cfg_c), C.byref(hw), flags, self._image.data_ptr(), need, C.byref(self._w_c), self._blocks,
                                   self._exact, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "vit_pack")
        del keep
        if c.pre_norm:          # not part of the packed image: two fp32 vectors the host hands over (include/amdstamp.h, amds_vit_weights.pre_norm_*)
            self._pre_norm = [sd[k].detach().to(self.device_, torch.float32).contiguous() for k in ("norm_pre.weight", "norm_pre.bias")]
            self._w_c.pre_norm_w, self._w_c.pre_norm_b = self._pre_norm[0].data_ptr(), self._pre_norm[1].data_ptr()

    def _pack_fp8(self, sd: dict[str, torch.Tensor]) -> None:
        """e4m3 weights with per-output-channel scales (the library's own row quantiser applied to the weight rows), LayerScale multiplied into
        the channel vectors of proj / fc2."""
        c, dev = self.cfg, self.device_
        self._fp8_keep: list[torch.Tensor] = []
        self._fp8 = (_lib.VitFp8Block * c.depth)()

        def keep(t):
            self._fp8_keep.append(t)
            return t.data_ptr()
        for i in range(c.depth):
            g = lambda n: sd[f"blocks.{i}.{n}"].detach().to(dev, torch.float32).contiguous()  # noqa: E731
            ls1 = g("ls1.gamma") if c.layerscale else torch.ones(c.dim, device=dev)
            ls2 = g("ls2.gamma") if c.layerscale else torch.ones(c.dim, device=dev)
            f = self._fp8[i]
            w8, sw = ops.quantize_rows_e4m3(g("attn.qkv.weight"))
            f.qkv_w8, f.qkv_cs = keep(w8), keep(sw)
            w8, sw = ops.quantize_rows_e4m3(g("attn.proj.weight"))
            f.proj_w8, f.proj_cs, f.proj_b = keep(w8), keep((sw * ls1).contiguous()), keep((g("attn.proj.bias") * ls1).contiguous())
            w1 = g("mlp.fc1.weight")
            if c.mlp == "swiglu":               # the 32-row gate / value interleave of the packed fc1 (the bias of the plain pack already has it)
                w1 = ops.pack_swiglu_rows(w1)
            w8, sw = ops.quantize_rows_e4m3(w1)
            f.fc1_w8, f.fc1_cs = keep(w8), keep(sw)
            # the bound that lets fc1's epilogue write e4m3 directly (amds_row_bound_scale); margin 1.15 >= (1 + 2^-4)^2: both operands of the product
            # are e4m3-rounded, each element by at most 2^-4 of itself
            f.fc1_wnorm_max, f.fc1_babs_max = 1.15 * float(w1.double().norm(dim=1).max()), float(g("mlp.fc1.bias").abs().max())
            w8, sw = ops.quantize_rows_e4m3(g("mlp.fc2.weight"))
            f.fc2_w8, f.fc2_cs, f.fc2_b = keep(w8), keep((sw * ls2).contiguous()), keep((g("mlp.fc2.bias") * ls2).contiguous())
        self._w_c.fp8_host = C.cast(self._fp8, C.POINTER(_lib.VitFp8Block))
        torch.cuda.synchronize(dev)

    # -- forward ----------------------------------------------------------------------------------
    def _workspace(self, chunk: int) -> torch.Tensor:
        need = _lib.lib().amds_vit_workspace_bytes(C.byref(self._cfg_c), chunk)
        if need == 0:
            _lib.check(-1, "vit_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device_)
            self._ws_chunk = None
        if getattr(self, "_ws_chunk", None) != chunk:      # the counters live at a chunk-dependent offset: a new geometry starts a new count
            self._ws_chunk = chunk
            self._diag_view(chunk).zero_()
        return self._ws

    def _diag_view(self, chunk: int) -> torch.Tensor:
        off = _lib.lib().amds_vit_workspace_diag_offset(C.byref(self._cfg_c), chunk)
        return self._ws[off:off + 8].view(torch.int32)

    def range_diagnostics(self, reset: bool = False) -> dict:
        """Counts, accumulated over the forwards since the workspace was made (or the last reset), of residual rows entering a FOLDED LayerNorm
        (a) whose sum of squares reached fp16's max^2 -- only such rows can hold an element their 16-bit copy cannot represent (massive-
        activation channels of real ViT-H / ViT-g checkpoints; the features of an affected tile come out non-finite) -- and (b) with
        |mean| > 8 sigma, where the folded form loses precision.  Both 0 on every preset with the synthetic weights; non-zero on a real
        checkpoint means: construct with ln_fold=False (stand-alone LayerNorm kernels, no 16-bit copy of un-normalised rows).  Synchronises."""
        if self._ws is None or not self.ln_fold or getattr(self, "_ws_chunk", None) is None:
            return {"rows_beyond_fp16_range_possible": 0, "rows_mean_over_8_sigma": 0}
        d = self._diag_view(self._ws_chunk)
        out = d.cpu().tolist()
        if reset:
            d.zero_()
        return {"rows_beyond_fp16_range_possible": int(out[0]), "rows_mean_over_8_sigma": int(out[1])}

    def export_range_counters(self, dst_pinned: torch.Tensor) -> bool:
        """The two range counters (see `range_diagnostics`) -> two int32 of PINNED host memory, in stream order, by a kernel's stores (no copy command,
        no synchronisation), and reset: how a caller that keeps many encoder calls in flight (preprocess.extract_slides) learns, later, what each call
        counted.  False when this object has no counters (LayerNorm not folded, overlapped schedule, safe mode)."""
        m = self._safe if self._safe is not None else self
        if m is not self or self._ws is None or not self.ln_fold or getattr(self, "_ws_chunk", None) is None:
            return False
        assert dst_pinned.dtype == torch.int32 and dst_pinned.numel() >= 2 and dst_pinned.is_pinned()
        _lib.check(_lib.lib().amds_export_words(self._diag_view(self._ws_chunk).data_ptr(), dst_pinned.data_ptr(), 2, 1, torch.cuda.current_stream().cuda_stream),
                   "export_words")
        return True

    def _as_u8_hwc(self, tiles: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        if tiles.dtype == torch.uint8:
            if tiles.dim() != 4 or tiles.shape[-1] != 3:
                raise ValueError(f"u8 tiles must be [B,H,W,3], got {tuple(tiles.shape)}")
            return tiles.contiguous()
        # float [B,3,H,W] that already went through ToTensor+Normalize: the map is a bijection on u8
        # values, so undo it exactly and take the fused u8 path.
        if tiles.dim() != 4 or tiles.shape[1] != 3:
            raise ValueError(f"float tiles must be [B,3,H,W], got {tuple(tiles.shape)}")
        mean = torch.tensor(c.mean, device=tiles.device, dtype=torch.float32).view(1, 3, 1, 1)
        std = torch.tensor(c.std, device=tiles.device, dtype=torch.float32).view(1, 3, 1, 1)
        u8 = ((tiles.float() * std + mean) * 255.0).round().clamp(0, 255).to(torch.uint8)
        return u8.permute(0, 2, 3, 1).contiguous()

    # -- the guard in front of the feature file ----------------------------------------------------
    def features_finite(self, feats: torch.Tensor) -> bool:
        """`amds_check_finite` over a feature tensor on the GPU (synchronises the current stream)."""
        if feats.numel() == 0:
            return True
        if getattr(self, "_cnt", None) is None:
            self._cnt = torch.zeros(1, dtype=torch.int32, device=self.device_)
        host = (C.c_int * 1)()
        rc = _lib.lib().amds_check_finite(feats.data_ptr(), feats.numel(), {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}[feats.dtype],
                                          self._cnt.data_ptr(), host, torch.cuda.current_stream().cuda_stream)
        if rc == _lib.ERR_RANGE:
            return False
        _lib.check(rc, "check_finite")
        return True

    def enable_safe_mode(self, why: str = "non-finite features") -> bool:
        """Moves this object one level up the ladder of `check` (re-packs from the kept state_dict); False when there is no level left."""
        import warnings
        cur = self._safe if self._safe is not None else self
        if self.fp8 or cur.act_dtype == torch.bfloat16 and not cur.ln_fold:
            return False
        if cur.ln_fold:
            level, kw, what = 1, dict(act_dtype=cur.act_dtype, ln_fold=False), "LayerNorm un-folded, fp32 residual rows"
        else:
            level, kw, what = 2, dict(act_dtype=torch.bfloat16, ln_fold=False), "bf16 activations (fp32 range, 8-bit mantissa: outside the 1e-3 parity bar)"
        if self.cfg.mlp == "quick_gelu" and level == 1:
            level, kw, what = 2, dict(act_dtype=torch.bfloat16, ln_fold=False), "bf16 activations (fp32 range, 8-bit mantissa: outside the 1e-3 parity bar)"
        warnings.warn(f"HipViT: {why} on the {'default' if self.safe_level == 0 else 'level-%d' % self.safe_level} path; re-packing on safe level {level}: "
                      f"{what}.  Range counters of the folded LayerNorms: {self.range_diagnostics()}", RuntimeWarning, stacklevel=3)
        # this object never runs its own image again: its packed weights (incl. an fp8 copy), workspace and the previous level's go before the re-pack
        object.__setattr__(self, "_safe", None)
        self._image = None
        self._ws = None
        self._ws_chunk = None
        torch.cuda.empty_cache()
        object.__setattr__(self, "_safe", HipViT(self.cfg, self._sd_ref, check="off", **kw, **self._ctor))
        self.safe_level = level
        return True

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor, return_tokens: bool = False):
        """tiles: u8 [B,H,W,3] (preferred) or normalised float [B,3,H,W] on the GPU -> fp16 [B,D]."""
        while True:
            m = self._safe if self._safe is not None else self
            m.chunk, m.overlap = self.chunk, self.overlap
            out = m._forward(tiles, return_tokens)
            if self.check == "off" or self.defer_check:
                return out
            why = self.call_verdict(out[0] if return_tokens else out)
            if why is None:
                return out
            if self.check == "raise" or not self.enable_safe_mode(why):
                raise FeatureRangeError(f"HipViT: {why} (safe level {self.safe_level}; range counters {self.range_diagnostics()})")

    def call_verdict(self, feats: torch.Tensor) -> str | None:
        """None when the features of the call(s) since the last verdict can be stored; otherwise the reason they cannot: non-finite values (an
        intermediate overflowed the 16-bit activation format), or -- on the LayerNorm-folded path only -- rows whose |mean| exceeds 8 sigma entered a
        folded LayerNorm (`amds_ln_rowstat_diag`): the folded form subtracts mean * colsum AFTER the product, which costs |mean| / sigma in
        relative precision there (finite, but outside the parity bar).  Synchronises; resets the counters it read."""
        if not self.features_finite(feats):
            return "non-finite features: an intermediate of this checkpoint leaves the 16-bit activation range"
        if self._safe is None and self.ln_fold and self._ws is not None and getattr(self, "_ws_chunk", None) is not None:
            d = self.range_diagnostics(reset=False)
            if d["rows_mean_over_8_sigma"]:
                self._diag_view(self._ws_chunk).zero_()
                return f"{d['rows_mean_over_8_sigma']} rows with |mean| > 8 sigma entered a folded LayerNorm (precision loss ~ |mean| / sigma)"
        return None

    @torch.no_grad()
    def _forward(self, tiles: torch.Tensor, return_tokens: bool = False):
        if not tiles.is_cuda:
            raise RuntimeError("HipViT.forward needs tiles on the GPU (no CPU fallback)")
        c = self.cfg
        tiles = self._as_u8_hwc(tiles)
        B = tiles.shape[0]
        if tiles.shape[1] != c.img or tiles.shape[2] != c.img:
            raise ValueError(f"expected {c.img}x{c.img} tiles, got {tuple(tiles.shape)}")
        feats = torch.empty(B, c.dim, dtype=torch.float16, device=tiles.device)
        toks = torch.empty(B, c.tokens, c.dim, dtype=torch.float32, device=tiles.device) if return_tokens else None
        if B == 0:
            return (feats, toks) if return_tokens else feats
        chunk = min(self.chunk, B)
        if self.overlap and not return_tokens and B > chunk:
            need = 2 * _lib.lib().amds_vit_workspace_bytes(C.byref(self._cfg_c), chunk)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device_)
            self._ws_chunk = None           # (the two-stream schedule keeps two plans in the buffer; the range counters are not read back from it)
            rc = _lib.lib().amds_vit_forward_overlapped(_lib.ctx(tiles.device.index or 0), C.byref(self._cfg_c), C.byref(self._w_c), tiles.data_ptr(), feats.data_ptr(),
                                                        B, chunk, self._ws.data_ptr(), self._ws.numel(),
                                                        torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "vit_forward_overlapped")
            return feats
        ws = self._workspace(chunk)
        rc = _lib.lib().amds_vit_forward_tokens(
            C.byref(self._cfg_c), C.byref(self._w_c), tiles.data_ptr(), feats.data_ptr(),
            toks.data_ptr() if toks is not None else None, B, chunk, ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "vit_forward")
        return (feats, toks) if return_tokens else feats

    # nn.Module plumbing the Extractor seam exercises: `.to(device).eval()` (preprocessing/__init__.py:243)
    def to(self, *args, **kwargs):  # weights are packed for one device at construction
        return self

    def with_chunk(self, chunk: int) -> "HipViT":
        self.chunk = int(chunk)
        return self


class HipViTClsMean(nn.Module):
    """`VirchowConcatenated` of the reference (src/stamp/preprocessing/extractor/virchow_full.py:25-35): the model output is the
    final-LayerNorm'd token tensor; the tile embedding is cat(class token, mean of output[:, 1:]) -> [B, 2*dim] (2560 for Virchow).
    The trunk and the token mean run in libamdstamp (`amds_vit_forward_tokens`, `amds_mean_pool`); slicing / concatenation are
    data movement."""

    def __init__(self, vit: "HipViT") -> None:
        super().__init__()
        self.vit = vit
        self.cfg = vit.cfg

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor) -> torch.Tensor:
        _, toks = self.vit(tiles, return_tokens=True)                 # fp32 [B, T, D], final norm applied
        if toks.shape[0] == 0:
            return toks.new_zeros(0, 2 * self.cfg.dim, dtype=torch.float16)
        mean = ops.mean_pool(toks[:, 1:].contiguous())                 # virchow_full.py:33-34: everything after the class token
        return torch.cat([toks[:, 0], mean], dim=-1).half()

    def to(self, *args, **kwargs):
        return self


def hf_dinov2_to_timm_names(sd: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """HF `transformers` `Dinov2Model` / `Dinov2WithRegistersModel` state_dict -> timm VisionTransformer names (what `HipViT` and the reference's
    factories use).  Data movement only: query / key / value concatenated into `attn.qkv`, `layer_scale*.lambda1` -> `ls*.gamma`, SwiGLU
    `weights_in` / `weights_out` -> `mlp.fc1` / `mlp.fc2`; with register tokens the position table gets zero rows for them (HF adds positions to
    the class token and the patches only), which is timm's `no_embed_class=False` layout."""
    p = "embeddings."
    out = {"patch_embed.proj.weight": sd[p + "patch_embeddings.projection.weight"], "patch_embed.proj.bias": sd[p + "patch_embeddings.projection.bias"],
           "cls_token": sd[p + "cls_token"], "norm.weight": sd["layernorm.weight"], "norm.bias": sd["layernorm.bias"]}
    pos = sd[p + "position_embeddings"]
    if p + "register_tokens" in sd:
        reg = sd[p + "register_tokens"]
        out["reg_token"] = reg
        pos = torch.cat([pos[:, :1], torch.zeros(1, reg.shape[1], pos.shape[2], dtype=pos.dtype), pos[:, 1:]], dim=1)
    out["pos_embed"] = pos
    l = 0
    while f"encoder.layer.{l}.norm1.weight" in sd:
        q, b = f"encoder.layer.{l}.", f"blocks.{l}."
        for n in ("norm1", "norm2"):
            out[b + n + ".weight"], out[b + n + ".bias"] = sd[q + n + ".weight"], sd[q + n + ".bias"]
        a = q + "attention.attention."
        out[b + "attn.qkv.weight"] = torch.cat([sd[a + f"{n}.weight"] for n in ("query", "key", "value")], dim=0)
        out[b + "attn.qkv.bias"] = torch.cat([sd[a + f"{n}.bias"] for n in ("query", "key", "value")], dim=0)
        out[b + "attn.proj.weight"], out[b + "attn.proj.bias"] = sd[q + "attention.output.dense.weight"], sd[q + "attention.output.dense.bias"]
        out[b + "ls1.gamma"], out[b + "ls2.gamma"] = sd[q + "layer_scale1.lambda1"], sd[q + "layer_scale2.lambda1"]
        f1, f2 = ("mlp.weights_in", "mlp.weights_out") if (q + "mlp.weights_in.weight") in sd else ("mlp.fc1", "mlp.fc2")
        out[b + "mlp.fc1.weight"], out[b + "mlp.fc1.bias"] = sd[q + f1 + ".weight"], sd[q + f1 + ".bias"]
        out[b + "mlp.fc2.weight"], out[b + "mlp.fc2.bias"] = sd[q + f2 + ".weight"], sd[q + f2 + ".bias"]
        l += 1
    return out


def random_vit_state_dict(cfg: ViTConfig, seed: int = 0, init: str = "moderate") -> dict[str, torch.Tensor]:
    """Random timm-named weights (no checkpoints are reachable offline).

    init="timm":   timm's own init (trunc-normal 0.02, zero bias, LayerScale 1e-5) -- later blocks barely matter;
    init="moderate": O(1) activations everywhere (fan-in scaled weights, random biases / LN affine, LayerScale
                   ~0.3) so every block contributes, and WELL-CONDITIONED: the fp32 and fp64 evaluations of the
                   oracle differ by ~5e-7 (about 9x the fp32 epsilon) on ViT-L/14.  The parity bar is stated on it.
    init="stress": same but qkv gain 2 and LayerScale ~0.5 -> sharp attention, a CHAOTIC map: fp32 vs fp64 of the
                   oracle already differ by 1.1e-5 (180x epsilon), i.e. any rounding is amplified ~180x.
    init="massive": "moderate" with the two statistics real DINOv2-family checkpoints are known for and random init lacks
                   (Sun et al., "Massive Activations in LLMs" / Darcet et al., "Vision Transformers Need Registers"): (a) a few residual
                   CHANNELS carrying values 1e2 ... 1e4 x the median -- three channels of the class token (1e2, 1e3, 1e4 from the embedding on,
                   so the stored row itself is a massive-activation row through every block) and two channels of EVERY token switched on by the
                   fc2 bias of the block at 2/3 depth (3e2, 3e3), so the late blocks' LayerNorms, operands and residual updates all see them;
                   (b) LayerScale gammas log-uniform over [1e-5, 1] instead of ~0.3.  Everything stays inside fp16's range.
    init="overflow": "massive" with one channel of every token driven past fp16's maximum (1e5) at 2/3 depth: the fast path cannot
                   represent this residual stream -- its features must come out through the guard (`HipViT(check=...)`), never as NaN.
    """
    g = torch.Generator().manual_seed(seed)
    D, p = cfg.dim, cfg.patch
    fc1_out = cfg.hidden * (2 if cfg.mlp == "swiglu" else 1)
    sd: dict[str, torch.Tensor] = {}

    def rn(*shape, s=1.0):
        return torch.randn(*shape, generator=g) * s

    assert init in ("timm", "moderate", "stress", "massive", "overflow")
    massive = init in ("massive", "overflow")
    stress = init in ("stress", "moderate") or massive
    qgain, lsm = (2.0, 0.5) if init == "stress" else (1.0, 0.3)
    ws = (lambda fan_in: 1.0 / fan_in ** 0.5) if stress else (lambda fan_in: 0.02)
    bs = 0.1 if stress else 0.0
    sd["patch_embed.proj.weight"] = rn(D, 3, p, p, s=ws(3 * p * p))
    sd["patch_embed.proj.bias"] = rn(D, s=bs)
    sd["cls_token"] = rn(1, 1, D, s=0.5 if stress else 1e-6)
    if cfg.reg_tokens:
        sd["reg_token"] = rn(1, cfg.reg_tokens, D, s=0.5 if stress else 1e-6)
    if cfg.pre_norm:
        sd["norm_pre.weight"], sd["norm_pre.bias"] = 1.0 + rn(D, s=0.1), rn(D, s=0.1)
    n_pos = cfg.n_patches + (0 if cfg.no_embed_class else cfg.n_prefix)
    sd["pos_embed"] = rn(1, n_pos, D, s=0.5 if stress else 0.02)
    for i in range(cfg.depth):
        pre = f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[pre + n + ".weight"] = 1.0 + rn(D, s=0.2 if stress else 0.0)
            sd[pre + n + ".bias"] = rn(D, s=bs)
        sd[pre + "attn.qkv.weight"] = rn(3 * D, D, s=ws(D) * (qgain if stress else 1.0))
        sd[pre + "attn.qkv.bias"] = rn(3 * D, s=bs)
        sd[pre + "attn.proj.weight"] = rn(D, D, s=ws(D))
        sd[pre + "attn.proj.bias"] = rn(D, s=bs)
        sd[pre + "mlp.fc1.weight"] = rn(fc1_out, D, s=ws(D))
        sd[pre + "mlp.fc1.bias"] = rn(fc1_out, s=bs)
        sd[pre + "mlp.fc2.weight"] = rn(D, cfg.hidden, s=ws(cfg.hidden))
        sd[pre + "mlp.fc2.bias"] = rn(D, s=bs)
        if cfg.layerscale:
            sd[pre + "ls1.gamma"] = (lsm + rn(D, s=0.1 * lsm / 0.5)) if stress else torch.full((D,), 1e-5)
            sd[pre + "ls2.gamma"] = (lsm + rn(D, s=0.1 * lsm / 0.5)) if stress else torch.full((D,), 1e-5)
    sd["norm.weight"] = 1.0 + rn(D, s=0.2 if stress else 0.0)
    sd["norm.bias"] = rn(D, s=bs)
    if massive:
        ch = torch.randperm(D, generator=g)[:6].tolist()
        for c_, v in zip(ch[:3], (1.0e2, -1.0e3, 1.0e4)):
            sd["cls_token"][0, 0, c_] = v
        k0 = (2 * cfg.depth) // 3
        if cfg.layerscale:
            for i in range(cfg.depth):
                for n in ("ls1.gamma", "ls2.gamma"):
                    sd[f"blocks.{i}.{n}"] = torch.exp(torch.rand(D, generator=g) * math.log(1.0e5) - math.log(1.0e5))     # log-uniform [1e-5, 1]
        big = [(ch[3], 3.0e2), (ch[4], -3.0e3)] + ([(ch[5], 1.0e5)] if init == "overflow" else [])
        for c_, v in big:      # the residual stream gains ls2[c] * fc2.bias[c] = v in channel c of every token
            ls = float(sd[f"blocks.{k0}.ls2.gamma"][c_]) if cfg.layerscale else 1.0
            sd[f"blocks.{k0}.mlp.fc2.bias"][c_] = v / ls
    return sd


__all__ = ["ViTConfig", "PRESETS", "HipViT", "FeatureRangeError", "HipViTClsMean", "random_vit_state_dict", "packed_weight_bytes", "expected_state_dict_shapes",
           "validate_state_dict", "replace", "field"]
