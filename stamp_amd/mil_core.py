"""Functional core of the MIL `vit` head on the HIP path: weight packing, inference forward, training forward and backward.

Everything with a token dimension runs in libamdstamp (GEMMs, LayerNorm, attention, GELU / dropout, reductions); this file is
the launch sequence and the bookkeeping around it.  The two front ends are `stamp_amd.mil.VisionTransformer` (an nn.Module with
the reference's constructor, parameter tree and ``forward(bags, coords=, mask=)``; differentiable through one
torch.autograd.Function) and `stamp_amd.mil_train.HipMilVitTrainer` (flat fp32 master buffer + fused AdamW).

Reference: src/stamp/modeling/models/vision_tranformer.py -- VisionTransformer.forward :331-384, Transformer.forward :281-295,
SelfAttention :172-242, feed_forward :157-169, MultiHeadALiBi / _ALiBi / _RunningMeanScaler :15-154.

Shapes.  The kernels want GEMM dimensions in multiples of 256 (the split-K weight-gradient GEMMs) and attention heads of 64 channels.  Models that do not have them
(the reference's own tests use dim_input 456, dim_model 4 x 33, dim_feedforward 135; tests/test_model.py:9-32) are run on a
ZERO-PADDED copy of the weights: input / model / feed-forward widths padded to multiples of 256, every head padded from
head_dim to 64 channels with `sqrt(64 / head_dim)` folded into the query rows (the kernels scale scores by 1/8), all-zero heads
appended up to a multiple of 4 heads.  Padding channels stay exactly zero through the whole network (zero weights and biases;
LayerNorm is evaluated over the true `dim_model` columns only), so the padded model computes the same function; gradients are
sliced back to the reference shapes.  Restrictions that remain: head_dim <= 64 and dim_model % 4 == 0 (LayerNorm kernels read
float4).  When nothing needs padding (the defaults: 512 / 8 heads / 512) the packed tensors are plain views / casts.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import _lib, ops
from . import train_ops as T

# GEMM configuration of the training step: by shape, and a ragged last row tile (M = bags x 1025 tokens is never a multiple of 256) may run as
# its own small launch when that saves a wave of workgroups (include/amdstamp.h, amds_gemm_ex cfg -2).  The inference forward keeps -1: its
# rows come from one kernel whatever the batch.
_CFG_TRAIN = -2

BF = torch.bfloat16
_ENC = ("query_encoders", "key_encoders", "value_encoders")


def _up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


@dataclass(frozen=True)
class VitDims:
    F: int
    D: int
    H: int
    FF: int
    C: int
    L: int
    alibi: bool
    p_drop: float = 0.0          # `dropout` of the constructor: project_features' Dropout and nn.MultiheadAttention's dropout
    p_ff: float = 0.5            # feed_forward's two Dropouts: the reference never forwards `dropout` to them (:268-271 -> :160)

    def __post_init__(self):
        if self.D % self.H:
            raise ValueError(f"dim_model={self.D} has to be divisible by n_heads={self.H}")
        if self.hd > 64 or self.D % 4:
            raise NotImplementedError(f"HIP MIL vit needs head_dim <= 64 and dim_model % 4 == 0 (dim_model={self.D}, n_heads={self.H})")

    @property
    def hd(self) -> int:
        return self.D // self.H

    @property
    def Fp(self) -> int:
        return _up(self.F, 256)

    @property
    def Dp(self) -> int:
        return _up(self.D, 256)

    @property
    def FFp(self) -> int:
        return _up(self.FF, 256)

    @property
    def Ha(self) -> int:
        return _up(self.H, 4)

    @property
    def Da(self) -> int:
        return 64 * self.Ha

    @property
    def qscale(self) -> float:
        return 8.0 / math.sqrt(self.hd)


# ---- reference parameter names ------------------------------------------------------------------------------------------------
def layer_prefix(l: int) -> str:
    return f"transformer.layers.{l}."


def param_names(d: VitDims) -> list[str]:
    """state_dict order of the reference module (parameters and the ALiBi scaler buffers)."""
    names = ["class_token", "project_features.0.weight", "project_features.0.bias"]
    for l in range(d.L):
        p = layer_prefix(l)
        names += [p + "0.norm.weight", p + "0.norm.bias"]
        if d.alibi:
            for e in _ENC:
                for h in range(d.H):
                    names += [p + f"0.mhsa.{e}.{h}.weight", p + f"0.mhsa.{e}.{h}.bias"]
            for h in range(d.H):
                names += [p + f"0.mhsa.attentions.{h}.bias_scale", p + f"0.mhsa.attentions.{h}.scale_distance.running_mean",
                          p + f"0.mhsa.attentions.{h}.scale_distance.items_so_far"]
            names += [p + "0.mhsa.fc.weight", p + "0.mhsa.fc.bias"]
        else:
            names += [p + "0.mhsa.in_proj_weight", p + "0.mhsa.in_proj_bias", p + "0.mhsa.out_proj.weight", p + "0.mhsa.out_proj.bias"]
        names += [p + "1.0.weight", p + "1.0.bias", p + "1.1.weight", p + "1.1.bias", p + "1.4.weight", p + "1.4.bias"]
    names += ["transformer.norm.weight", "transformer.norm.bias", "mlp_head.0.weight", "mlp_head.0.bias"]
    return names


def flat_order(d: VitDims, names: list[str]) -> list[str]:
    """Order of the tensors inside a trainer's flat master / gradient / moment buffers.  The reference's state_dict interleaves the ALiBi head's
    per-head Linears (weight, bias, weight, bias ...); laid out per layer as [3 x H weights][3 x H biases][H bias scales][H running means][H counts]
    the stacked in-projection the kernels consume IS a contiguous view of the buffer (`_stack`), in both directions: no per-head copies when the
    operands are refreshed, and the library writes the gradients straight into the trainer's buffer (`_direct_grad_structs`).  Every name keeps
    its own view; nothing outside the trainer sees the order."""
    if not d.alibi:
        return list(names)
    out, done = [], set()
    for n in names:
        if n in done:
            continue
        first = layer_prefix(0)[:-2]            # "transformer.layers."
        if n.startswith(first) and n.endswith(f"0.mhsa.{_ENC[0]}.0.weight"):
            p = n[: -len(f"0.mhsa.{_ENC[0]}.0.weight")]
            grp = [p + f"0.mhsa.{e}.{h}.{kind}" for kind in ("weight", "bias") for e in _ENC for h in range(d.H)]
            grp += [p + f"0.mhsa.attentions.{h}.{leaf}" for leaf in ("bias_scale", "scale_distance.running_mean", "scale_distance.items_so_far") for h in range(d.H)]
            out += grp
            done.update(grp)
        else:
            out.append(n)
            done.add(n)
    assert sorted(out) == sorted(names)
    return out


def is_buffer(name: str) -> bool:
    return name.endswith("scale_distance.running_mean") or name.endswith("scale_distance.items_so_far")


def _stack(ts: list) -> torch.Tensor:
    """torch.stack(ts) -- as a strided VIEW (no launch) when the tensors are equally shaped slices of one storage at a constant pitch: the
    trainer's parameters are views of one flat buffer, and the ALiBi head's per-head Linears (3 x H weights + 3 x H biases per layer, then 2 H
    scalars) cost ~150 tiny device copies per refresh as a stack of separate tensors -- host time the step then waits for."""
    t0 = ts[0]
    if len(ts) > 1 and all(t.shape == t0.shape and t.stride() == t0.stride() and t.dtype == t0.dtype and t.device == t0.device
                           and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for t in ts):
        offs = [t.storage_offset() for t in ts]
        pitch = offs[1] - offs[0]
        if pitch > 0 and all(b - a == pitch for a, b in zip(offs, offs[1:])):
            return t0.as_strided((len(ts),) + tuple(t0.shape), (pitch,) + tuple(t0.stride()), offs[0])
    return torch.stack(ts)


# ---- zero padding of the weights (data movement only) --------------------------------------------------------------------------
def _pad2(w: torch.Tensor, R: int, Cc: int) -> torch.Tensor:
    if w.shape == (R, Cc):
        return w.contiguous()
    return F.pad(w, (0, Cc - w.shape[1], 0, R - w.shape[0])).contiguous()


def _pad1(v: torch.Tensor, n: int) -> torch.Tensor:
    return v.contiguous() if v.numel() == n else F.pad(v, (0, n - v.numel())).contiguous()


class PackedVit:
    """fp32 padded master copies (`m`), 16-bit MFMA operand copies W [N][K] (`w`) and, for training, W^T [K][N] (`wt`)."""

    def __init__(self, dims: VitDims, get, act: torch.dtype, train: bool) -> None:
        self.dims, self.act, self.train = dims, act, train
        self.refresh(get)

    # in-projection [3][H][hd][D] <-> padded [3*Da][Dp]; q rows carry sqrt(64/hd)
    def _pad_in(self, w3: torch.Tensor) -> torch.Tensor:
        d = self.dims
        if d.hd == 64 and d.Ha == d.H and (w3.dim() == 3 or d.D == d.Dp):
            return w3.reshape(3 * d.Da, -1).contiguous() if w3.dim() == 4 else w3.reshape(3 * d.Da).contiguous()
        if w3.dim() == 4:       # weight [3, H, hd, D]
            out = F.pad(w3, (0, d.Dp - d.D, 0, 64 - d.hd, 0, d.Ha - d.H))
            out[0] *= d.qscale
            return out.reshape(3 * d.Da, d.Dp).contiguous()
        out = F.pad(w3, (0, 64 - d.hd, 0, d.Ha - d.H))      # bias [3, H, hd]
        out[0] *= d.qscale
        return out.reshape(3 * d.Da).contiguous()

    def unpad_in_w(self, g: torch.Tensor) -> torch.Tensor:
        d = self.dims
        g4 = g.view(3, d.Ha, 64, d.Dp)[:, : d.H, : d.hd, : d.D]
        if d.hd != 64:
            g4 = g4.clone()
            g4[0] *= d.qscale
        return g4

    def unpad_in_b(self, g: torch.Tensor) -> torch.Tensor:
        d = self.dims
        g3 = g.view(3, d.Ha, 64)[:, : d.H, : d.hd]
        if d.hd != 64:
            g3 = g3.clone()
            g3[0] *= d.qscale
        return g3

    def _pad_out(self, w: torch.Tensor) -> torch.Tensor:      # [D, H*hd] -> [Dp, Da]
        d = self.dims
        if w.shape == (d.Dp, d.Da):
            return w.contiguous()
        return F.pad(w.reshape(d.D, d.H, d.hd), (0, 64 - d.hd, 0, d.Ha - d.H, 0, d.Dp - d.D)).reshape(d.Dp, d.Da).contiguous()

    def unpad_out_w(self, g: torch.Tensor) -> torch.Tensor:
        d = self.dims
        return g.view(d.Dp, d.Ha, 64)[: d.D, : d.H, : d.hd].reshape(d.D, d.D)

    def refresh(self, get) -> None:
        d = self.dims
        m: dict = {"cls": _pad1(get("class_token"), d.Dp), "proj_w": _pad2(get("project_features.0.weight"), d.Dp, d.Fp),
                   "proj_b": _pad1(get("project_features.0.bias"), d.Dp), "layers": []}
        for l in range(d.L):
            p = layer_prefix(l)
            Lm: dict = {"ln1": (get(p + "0.norm.weight").contiguous(), get(p + "0.norm.bias").contiguous()),
                        "ln2": (get(p + "1.0.weight").contiguous(), get(p + "1.0.bias").contiguous())}
            if d.alibi:     # per-head Linear(D, hd) encoders = a row-blocked in-projection [q heads | k heads | v heads]
                w3 = _stack([_stack([get(p + f"0.mhsa.{e}.{h}.weight") for h in range(d.H)]) for e in _ENC])
                b3 = _stack([_stack([get(p + f"0.mhsa.{e}.{h}.bias") for h in range(d.H)]) for e in _ENC])
                Lm["out_w"], Lm["out_b"] = self._pad_out(get(p + "0.mhsa.fc.weight")), _pad1(get(p + "0.mhsa.fc.bias"), d.Dp)
                bs = _stack([get(p + f"0.mhsa.attentions.{h}.bias_scale").reshape(1) for h in range(d.H)]).reshape(d.H)
                rm = _stack([get(p + f"0.mhsa.attentions.{h}.scale_distance.running_mean").reshape(1) for h in range(d.H)]).reshape(d.H)
                Lm["bias_scale"] = _pad1(bs, d.Ha)
                Lm["inv_rm"] = F.pad(1.0 / rm, (0, d.Ha - d.H), value=1.0).contiguous()
            else:
                w3 = get(p + "0.mhsa.in_proj_weight").view(3, d.H, d.hd, d.D)
                b3 = get(p + "0.mhsa.in_proj_bias").view(3, d.H, d.hd)
                Lm["out_w"], Lm["out_b"] = self._pad_out(get(p + "0.mhsa.out_proj.weight")), _pad1(get(p + "0.mhsa.out_proj.bias"), d.Dp)
            Lm["in_w"], Lm["in_b"] = self._pad_in(w3), self._pad_in(b3)
            Lm["fc1_w"], Lm["fc1_b"] = _pad2(get(p + "1.1.weight"), d.FFp, d.Dp), _pad1(get(p + "1.1.bias"), d.FFp)
            Lm["fc2_w"], Lm["fc2_b"] = _pad2(get(p + "1.4.weight"), d.Dp, d.FFp), _pad1(get(p + "1.4.bias"), d.Dp)
            m["layers"].append(Lm)
        m["norm"] = (get("transformer.norm.weight").contiguous(), get("transformer.norm.bias").contiguous())
        m["head_w"], m["head_b"] = get("mlp_head.0.weight").contiguous(), get("mlp_head.0.bias").contiguous()
        self.m = m
        # the 16-bit operand copies (and their transposes) are refreshed IN PLACE after every optimiser step: same buffers, no zero-fill launches, and the
        # C structs built over them stay valid (they are rebuilt only when a master tensor moved: padded / stacked ones are fresh torch tensors)
        old_w, old_wt = getattr(self, "w", None), getattr(self, "wt", None)
        prev = lambda dct, *keys: (dct[keys[0]] if len(keys) == 1 else dct["layers"][keys[0]][keys[1]]) if dct is not None and len(dct.get("layers", ())) == d.L else None  # noqa: E731
        # ONE launch for all of it (amds_cast_transpose_multi: cast + transposed copy per matrix, 9 matrices for the default head; the per-matrix
        # amds_cast_pad / amds_transpose16 pairs were ~20 launches of every training step); a matrix whose padded shape is not a multiple of 64 keeps them
        batch: list = []

        def cast(w, dt, out, out_t, want_t):
            dt = dt or self.act
            w = w.contiguous().float()
            R, Cc = w.shape
            dst = out if out is not None and out.shape == (R, Cc) and out.dtype == dt and out.is_contiguous() else torch.empty(R, Cc, dtype=dt, device=w.device)
            dst_t = None
            if want_t:
                dst_t = out_t if out_t is not None and out_t.shape == (Cc, R) and out_t.dtype == dt and out_t.is_contiguous() else torch.empty(Cc, R, dtype=dt, device=w.device)
            if R % 64 == 0 and Cc % 64 == 0:
                batch.append((w, dst, dst_t))
            else:
                ops.cast_pad(w, Cc, dt, out=dst)
                if want_t:
                    T.transpose16(dst, out=dst_t)
            return dst, dst_t

        new_w: dict = {"layers": []}
        new_wt: dict = {"layers": []}
        new_w["proj_w"], pt = cast(m["proj_w"], None, prev(old_w, "proj_w"), prev(old_wt, "proj_w") if old_wt and "proj_w" in old_wt else None, self.train)
        if self.train:
            new_wt["proj_w"] = pt
        for l, Lm in enumerate(m["layers"]):
            # ALiBi inference: the attention output is bf16 (range, see amds_attention_alibi), so its output projection runs on bf16 operands; the training
            # step keeps every 16-bit tensor in ITS operand type (fp16 at float32_matmul_precision "high": amds::attention_alibi_fwd_train_dt)
            Lw, Lt = {}, {}
            for k in ("in_w", "out_w", "fc1_w", "fc2_w"):
                Lw[k], Lt[k] = cast(Lm[k], BF if (d.alibi and k == "out_w" and not self.train) else None, prev(old_w, l, k), prev(old_wt, l, k) if self.train else None,
                                    self.train)
            new_w["layers"].append(Lw)
            if self.train:
                new_wt["layers"].append(Lt)
        if batch:
            T.cast_transpose_multi(batch)
        self.w, self.wt = new_w, new_wt
        self._c = None

    def c_structs(self):
        """(amds_mil_vit_cfg, amds_mil_vit_weights) over this pack's device tensors (include/amdstamp.h, "MIL `vit` head"); the ctypes
        objects and the per-layer ALiBi scales they point to live as long as the pack is not refreshed."""
        if self._c is None:
            d, m, w = self.dims, self.m, self.w
            cfg = _lib.MilVitCfg(d.F, d.D, d.H, d.FF, d.C, d.L, int(d.alibi), ops.act_code(self.act))
            layers = (_lib.MilVitLayer * max(d.L, 1))()
            keep = []
            for l, (Lm, Lw) in enumerate(zip(m["layers"], w["layers"])):
                hs = None
                if d.alibi:
                    hs = (Lm["bias_scale"] * Lm["inv_rm"]).contiguous()
                    keep.append(hs)
                layers[l] = _lib.MilVitLayer(Lm["ln1"][0].data_ptr(), Lm["ln1"][1].data_ptr(), Lw["in_w"].data_ptr(), Lm["in_b"].data_ptr(),
                                             Lw["out_w"].data_ptr(), Lm["out_b"].data_ptr(), hs.data_ptr() if hs is not None else None,
                                             Lm["ln2"][0].data_ptr(), Lm["ln2"][1].data_ptr(), Lw["fc1_w"].data_ptr(), Lm["fc1_b"].data_ptr(),
                                             Lw["fc2_w"].data_ptr(), Lm["fc2_b"].data_ptr())
                if self.train:
                    Lt = self.wt["layers"][l]
                    layers[l].in_wt, layers[l].out_wt = Lt["in_w"].data_ptr(), Lt["out_w"].data_ptr()
                    layers[l].fc1_wt, layers[l].fc2_wt = Lt["fc1_w"].data_ptr(), Lt["fc2_w"].data_ptr()
                if d.alibi:
                    layers[l].bias_scale, layers[l].inv_running_mean = Lm["bias_scale"].data_ptr(), Lm["inv_rm"].data_ptr()
            wc = _lib.MilVitWeights(m["cls"].data_ptr(), w["proj_w"].data_ptr(), m["proj_b"].data_ptr(), layers, m["norm"][0].data_ptr(),
                                    m["norm"][1].data_ptr(), m["head_w"].data_ptr(), m["head_b"].data_ptr(),
                                    self.wt["proj_w"].data_ptr() if self.train else None)
            self._c = (cfg, wc, layers, keep)
        return self._c[0], self._c[1]


def _coords_with_cls(coords: torch.Tensor, Bb: int, dev) -> torch.Tensor:
    return torch.cat([coords.new_zeros(Bb, 1, 2), coords], dim=1).to(dev, torch.float32).contiguous()      # class token at (0, 0), :349-351


def _ln(x, rows, cols, ld_in, gamma, beta, out_dtype, ld_out, buf=None):
    """LayerNorm over the first `cols` columns of rows pitched ld_in; output pitched ld_out, padding columns zero."""
    if buf is None:
        buf = (torch.zeros if ld_out != cols else torch.empty)(rows, ld_out, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().amds_layernorm(x.data_ptr(), ld_in, gamma.data_ptr(), beta.data_ptr(), buf.data_ptr(), ld_out, rows, cols, 1e-5,
                                         ops._DT[out_dtype], ops._stream()), "layernorm")
    return buf


# ---- inference forward (fp16 operands; deploy / validation / the reference's `mask` path) -----------------------------------------


def forward_infer(pk: PackedVit, bags: torch.Tensor, coords: torch.Tensor | None, mask: torch.Tensor | None) -> torch.Tensor:
    """bags [batch, tile, feature] -> logits [batch, dim_output]: ONE library call (amds_mil_vit_forward, csrc/mil_vit.hip) -- the host
    checks shapes, hands over pointers and keeps one workspace per device."""
    d = pk.dims
    if bags.dim() != 3 or bags.shape[-1] != d.F:
        raise ValueError(f"bags must be [batch, tile, {d.F}], got {tuple(bags.shape)}")
    ops._dev(bags)          # no CPU fallback
    Bb, Tn, _ = bags.shape
    dev = bags.device
    if d.alibi and coords is None:
        raise ValueError("use_alibi=True needs coords")
    if bags.dtype not in ops._DT:
        bags = bags.float()
    bags = bags.contiguous()
    c = m8 = None
    if d.alibi:
        if coords.shape != (Bb, Tn, 2):
            raise ValueError(f"coords must be [batch, tile, 2] = {(Bb, Tn, 2)}, got {tuple(coords.shape)}")
        c = coords.to(dev, torch.float32).contiguous()
    if mask is not None:
        if mask.shape != (Bb, Tn):
            raise ValueError(f"mask must be [batch, tile] = {(Bb, Tn)}, got {tuple(mask.shape)}")
        m8 = mask.to(dev, torch.uint8).contiguous()
    cfg, wc = pk.c_structs()
    lib = _lib.lib()
    need = lib.amds_mil_vit_workspace_bytes(C.byref(cfg), Bb, Tn)
    if need == 0:
        _lib.check(-1, "mil_vit_workspace_bytes")
    ws = ops.scratch("mil_vit_infer", dev, need)
    logits = torch.empty(Bb, d.C, dtype=torch.float32, device=dev)
    _lib.check(lib.amds_mil_vit_forward(C.byref(cfg), C.byref(wc), bags.data_ptr(), ops._DT[bags.dtype], c.data_ptr() if c is not None else None,
                                        m8.data_ptr() if m8 is not None else None, logits.data_ptr(), Bb, Tn, ws.data_ptr(), ws.numel(),
                                        ops._stream()), "mil_vit_forward")
    return logits


# ---- training forward / backward (bf16 operands, saved statistics) ----------------------------------------------------------------
def _gelu_drop_fwd(z, out_dtype, p, seed, sid):
    if p <= 0.0:
        return T.gelu_fwd(z, out_dtype)
    u = torch.empty(z.shape, dtype=out_dtype or z.dtype, device=z.device)
    _lib.check(_lib.lib().amds_gelu_dropout_fwd(z.data_ptr(), u.data_ptr(), z.numel(), ops._DT[z.dtype], ops._DT[u.dtype], p, seed, sid, ops._stream()),
               "gelu_dropout_fwd")
    return u


def _gelu_drop_bwd(z, du, p, seed, sid):
    if p <= 0.0:
        return T.gelu_bwd(z, du)
    dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    _lib.check(_lib.lib().amds_gelu_dropout_bwd(z.data_ptr(), du.data_ptr(), dz.data_ptr(), z.numel(), ops._DT[z.dtype], ops._DT[du.dtype],
                                                ops._DT[dz.dtype], p, seed, sid, ops._stream()), "gelu_dropout_bwd")
    return dz


def update_running_means(get, d: VitDims, cc: torch.Tensor) -> None:
    """Train-mode `_RunningMeanScaler` of every head and layer, BEFORE use (:24-29): rm <- rm + (mean(dist) - rm) / n ; n <- n + 1.
    Every scaler sees the same distance matrix, so one mean serves all of them (buffers updated in place)."""
    md = T.cdist_mean(cc)
    rms = [get(layer_prefix(l) + f"0.mhsa.attentions.{h}.scale_distance.running_mean") for l in range(d.L) for h in range(d.H)]
    ns = [get(layer_prefix(l) + f"0.mhsa.attentions.{h}.scale_distance.items_so_far") for l in range(d.L) for h in range(d.H)]
    with torch.no_grad():
        delta = torch._foreach_neg(rms)
        torch._foreach_add_(delta, md)
        torch._foreach_div_(delta, ns)
        torch._foreach_add_(rms, delta)
        torch._foreach_add_(ns, 1.0)


def _drop_struct(d: VitDims, training: bool, seed: int, device_index: int = 0) -> "_lib.MilVitDropout":
    # cls_tail: the context's switch is read ONCE here; the struct travels with the saved arena, so forward and backward agree by construction
    tail = int(_lib.lib().amds_get_mil_cls_tail(_lib.ctx(device_index)))
    return _lib.MilVitDropout(d.p_drop if training else 0.0, d.p_drop if (training and not d.alibi) else 0.0, d.p_ff if training else 0.0,
                              int(seed) & (2 ** 64 - 1), tail)


def forward_train(pk: PackedVit, bags: torch.Tensor, coords: torch.Tensor | None, *, training: bool, seed: int = 0):
    """-> (logits fp32 [Bb, C], saved).  `training` switches the dropout sites on (the running means are the caller's job).
    ONE library call (amds_mil_vit_train_forward, csrc/mil_vit_train.hip); `saved` holds the activation arena the backward reads."""
    d = pk.dims
    if not pk.train or pk.act not in (BF, torch.float16):
        raise RuntimeError("forward_train needs a training pack (PackedVit(..., torch.bfloat16 or torch.float16, train=True))")
    if bags.dim() != 3 or bags.shape[-1] != d.F:
        raise ValueError(f"bags must be [batch, tile, {d.F}], got {tuple(bags.shape)}")
    ops._dev(bags)
    Bb, Tn, Fd = bags.shape
    dev = bags.device
    if d.alibi and coords is None:
        raise ValueError("use_alibi=True needs coords")
    if bags.dtype not in ops._DT:
        bags = bags.float()
    bags = bags.contiguous()
    c = None
    if d.alibi:
        if coords.shape != (Bb, Tn, 2):
            raise ValueError(f"coords must be [batch, tile, 2] = {(Bb, Tn, 2)}, got {tuple(coords.shape)}")
        c = coords.to(dev, torch.float32).contiguous()
    cfg, wc = pk.c_structs()
    lib = _lib.lib()
    need = lib.amds_mil_vit_train_saved_bytes(C.byref(cfg), Bb, Tn)
    if need == 0:
        _lib.check(-1, "mil_vit_train_saved_bytes")
    arena = torch.empty(need, dtype=torch.uint8, device=dev)
    logits = torch.empty(Bb, d.C, dtype=torch.float32, device=dev)
    drop = _drop_struct(d, training, seed, dev.index or 0)
    _lib.check(lib.amds_mil_vit_train_forward(C.byref(cfg), C.byref(wc), bags.data_ptr(), ops._DT[bags.dtype], c.data_ptr() if c is not None else None,
                                              C.byref(drop), logits.data_ptr(), Bb, Tn, arena.data_ptr(), arena.numel(), ops._stream()),
               "mil_vit_train_forward")
    return logits, dict(arena=arena, shape=(Bb, Tn, Fd), drop=drop)


def _grad_buffers(d: VitDims, dev):
    """One flat fp32 buffer holding every gradient in the padded layout + the amds_mil_vit_grads over it + named views."""
    sizes = [("class_token", (d.Dp,)), ("proj_w", (d.Dp, d.Fp)), ("proj_b", (d.Dp,)), ("norm_w", (d.D,)), ("norm_b", (d.D,)), ("head_w", (d.C, d.D)),
             ("head_b", (d.C,))]
    per = [("ln1_w", (d.D,)), ("ln1_b", (d.D,)), ("in_w", (3 * d.Da, d.Dp)), ("in_b", (3 * d.Da,)), ("out_w", (d.Dp, d.Da)), ("out_b", (d.Dp,)),
           ("bias_scale", (d.Ha,)), ("ln2_w", (d.D,)), ("ln2_b", (d.D,)), ("fc1_w", (d.FFp, d.Dp)), ("fc1_b", (d.FFp,)), ("fc2_w", (d.Dp, d.FFp)),
           ("fc2_b", (d.Dp,))]
    al = lambda n: (n + 63) // 64 * 64  # noqa: E731   (256-byte aligned sub-buffers)
    total = sum(al(math.prod(sh)) for _, sh in sizes) + d.L * sum(al(math.prod(sh)) for _, sh in per)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0

    def view(sh):
        nonlocal off
        n = math.prod(sh)
        v = flat[off:off + n].view(sh)
        off += al(n)
        return v

    top = {k: view(sh) for k, sh in sizes}
    layers = [{k: view(sh) for k, sh in per} for _ in range(d.L)]
    lg = (_lib.MilVitLayerGrads * max(d.L, 1))()
    for l, Lg in enumerate(layers):
        lg[l] = _lib.MilVitLayerGrads(*[Lg[k].data_ptr() for k, _ in per])
    gc = _lib.MilVitGrads(top["class_token"].data_ptr(), top["proj_w"].data_ptr(), top["proj_b"].data_ptr(), lg, top["norm_w"].data_ptr(),
                          top["norm_b"].data_ptr(), top["head_w"].data_ptr(), top["head_b"].data_ptr())
    return flat, gc, lg, top, layers


def _direct_grad_structs(d: VitDims, grad_views):
    """When nothing is padded (widths multiples of 256, 64-channel heads in multiples of 4) the padded gradient layout IS the reference's (ALiBi: that of a
    buffer in `flat_order`):
    amds_mil_vit_grads can point straight at the caller's own gradient tensors (`grad_views(name)`, e.g. views of a trainer's flat buffer) and
    the per-step copies disappear.  -> (MilVitGrads, keep-alive, {name: tensor}) or None when the shapes (or a pointer's alignment) do not allow it."""
    if d.D != d.Dp or d.F != d.Fp or d.FF != d.FFp or d.hd != 64 or d.Ha != d.H:
        return None
    names = [n for n in param_names(d) if not is_buffer(n)]
    G = {n: grad_views(n) for n in names}
    # (the ALiBi head's per-head tensors are addressed through their stacks below: a lone bias scale is one float)
    if any((not t.is_contiguous()) or t.dtype != torch.float32 or (t.data_ptr() % 16 and ".mhsa.attentions." not in n) for n, t in G.items()):
        return None
    lg = (_lib.MilVitLayerGrads * max(d.L, 1))()
    for l in range(d.L):
        p = layer_prefix(l)
        if d.alibi:     # the stacked in-projection [3][H][64][D] | [3][H][64] and the H bias scales must BE contiguous runs of the caller's buffer (flat_order)
            in_w = _stack([_stack([G[p + f"0.mhsa.{e}.{h}.weight"] for h in range(d.H)]) for e in _ENC])
            in_b = _stack([_stack([G[p + f"0.mhsa.{e}.{h}.bias"] for h in range(d.H)]) for e in _ENC])
            bsc = _stack([G[p + f"0.mhsa.attentions.{h}.bias_scale"].reshape(1) for h in range(d.H)])
            own = G[p + "0.norm.weight"].untyped_storage().data_ptr()
            if any((not t.is_contiguous()) or t.untyped_storage().data_ptr() != own or t.data_ptr() % 16 for t in (in_w, in_b, bsc)):
                return None
            attn = (in_w.data_ptr(), in_b.data_ptr(), G[p + "0.mhsa.fc.weight"].data_ptr(), G[p + "0.mhsa.fc.bias"].data_ptr(), bsc.data_ptr())
        else:
            attn = (G[p + "0.mhsa.in_proj_weight"].data_ptr(), G[p + "0.mhsa.in_proj_bias"].data_ptr(), G[p + "0.mhsa.out_proj.weight"].data_ptr(),
                    G[p + "0.mhsa.out_proj.bias"].data_ptr(), None)
        lg[l] = _lib.MilVitLayerGrads(G[p + "0.norm.weight"].data_ptr(), G[p + "0.norm.bias"].data_ptr(), *attn,
                                      G[p + "1.0.weight"].data_ptr(), G[p + "1.0.bias"].data_ptr(), G[p + "1.1.weight"].data_ptr(), G[p + "1.1.bias"].data_ptr(),
                                      G[p + "1.4.weight"].data_ptr(), G[p + "1.4.bias"].data_ptr())
    gc = _lib.MilVitGrads(G["class_token"].data_ptr(), G["project_features.0.weight"].data_ptr(), G["project_features.0.bias"].data_ptr(), lg,
                          G["transformer.norm.weight"].data_ptr(), G["transformer.norm.bias"].data_ptr(), G["mlp_head.0.weight"].data_ptr(), G["mlp_head.0.bias"].data_ptr())
    return gc, lg, G


def backward(pk: PackedVit, saved: dict, dlogits: torch.Tensor, *, need_params: bool = True, need_bags: bool = False, split_k: int = 32, grad_views=None):
    """-> (grads: reference-named, reference-shaped fp32 tensors (empty dict if not need_params), dbags fp32 [Bb,T,F] or None).
    ONE library call (amds_mil_vit_train_backward); the host slices the reference shapes out of the padded gradient buffers -- or, with
    `grad_views` (name -> the caller's own gradient tensor) and an unpadded geometry, the library writes into those tensors directly."""
    d = pk.dims
    dev = dlogits.device
    Bb, Tn, Fd = saved["shape"]
    cfg, wc = pk.c_structs()
    lib = _lib.lib()
    dlogits = dlogits.contiguous().float()
    if dlogits.shape != (Bb, d.C):
        raise ValueError(f"dlogits must be {(Bb, d.C)}, got {tuple(dlogits.shape)}")
    need = lib.amds_mil_vit_train_workspace_bytes(C.byref(cfg), Bb, Tn, split_k)
    if need == 0:
        _lib.check(-1, "mil_vit_train_workspace_bytes")
    ws = ops.scratch("mil_vit_train", dev, need)
    gc = top = layers = direct = None
    if need_params and grad_views is not None:
        direct = _direct_grad_structs(d, grad_views)
    if direct is not None:
        gc = direct[0]
    elif need_params:
        _flat, gc, _lg, top, layers = _grad_buffers(d, dev)
    dbp = torch.empty(Bb * Tn, d.Fp, dtype=torch.float32, device=dev) if need_bags else None
    arena = saved["arena"]
    _lib.check(lib.amds_mil_vit_train_backward(C.byref(cfg), C.byref(wc), dlogits.data_ptr(), C.byref(saved["drop"]), Bb, Tn, arena.data_ptr(), arena.numel(),
                                               C.byref(gc) if gc is not None else None, dbp.data_ptr() if dbp is not None else None, split_k,
                                               ws.data_ptr(), ws.numel(), ops._stream()), "mil_vit_train_backward")
    dbags = dbp[:, :Fd].reshape(Bb, Tn, Fd) if need_bags else None
    G: dict[str, torch.Tensor] = {}
    if not need_params:
        return G, dbags
    if direct is not None:
        return direct[2], dbags
    D = d.D
    G["mlp_head.0.weight"], G["mlp_head.0.bias"] = top["head_w"], top["head_b"]
    G["transformer.norm.weight"], G["transformer.norm.bias"] = top["norm_w"], top["norm_b"]
    G["class_token"] = top["class_token"][:D]
    G["project_features.0.weight"], G["project_features.0.bias"] = top["proj_w"][:D, :Fd], top["proj_b"][:D]
    for l, Lg in enumerate(layers):
        p = layer_prefix(l)
        G[p + "1.4.weight"], G[p + "1.4.bias"] = Lg["fc2_w"][:D, : d.FF], Lg["fc2_b"][:D]
        G[p + "1.1.weight"], G[p + "1.1.bias"] = Lg["fc1_w"][: d.FF, :D], Lg["fc1_b"][: d.FF]
        G[p + "1.0.weight"], G[p + "1.0.bias"] = Lg["ln2_w"], Lg["ln2_b"]
        G[p + "0.norm.weight"], G[p + "0.norm.bias"] = Lg["ln1_w"], Lg["ln1_b"]
        out_name = "0.mhsa.fc." if d.alibi else "0.mhsa.out_proj."
        G[p + out_name + "weight"], G[p + out_name + "bias"] = pk.unpad_out_w(Lg["out_w"]), Lg["out_b"][:D]
        gw, gb = pk.unpad_in_w(Lg["in_w"]), pk.unpad_in_b(Lg["in_b"])
        if d.alibi:
            for h in range(d.H):
                G[p + f"0.mhsa.attentions.{h}.bias_scale"] = Lg["bias_scale"][h:h + 1]
            for i, e in enumerate(_ENC):
                for h in range(d.H):
                    G[p + f"0.mhsa.{e}.{h}.weight"], G[p + f"0.mhsa.{e}.{h}.bias"] = gw[i, h], gb[i, h]
        else:
            G[p + "0.mhsa.in_proj_weight"], G[p + "0.mhsa.in_proj_bias"] = gw.reshape(3 * D, D), gb.reshape(3 * D)
    return G, dbags


def flops_per_bag(T_: int = 1024, Fd: int = 1024, D: int = 512, FF: int = 512, L: int = 2) -> float:
    """matmul FLOPs of one forward (2 per MAC): projection + L x (qkv, attention, out, fc1, fc2); training ~ 3x."""
    S = T_ + 1
    return 2 * T_ * Fd * D + L * (2 * S * D * 3 * D + 4 * S * S * D + 2 * S * D * D + 2 * S * D * FF + 2 * S * FF * D)
