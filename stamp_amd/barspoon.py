"""The reference's barspoon head (`EncDecTransformer`, src/stamp/modeling/models/barspoon.py:27-205; `ModelName.BARSPOON` of the train registry,
src/stamp/modeling/registry.py:18-25) on the HIP path -- deploy / validation forward.

Same constructor, same module tree and therefore the same state_dict keys as the reference class (the torch containers `nn.TransformerEncoder`
/ `nn.TransformerDecoder` are instantiated as parameter holders and never called): a checkpoint's `model.*` entries load with
`load_state_dict(strict=True)`.  `forward(tile_tokens, tile_positions)` -> `{target_label: logits [batch, n_out]}` is ONE library call
(`amds_barspoon_forward`, csrc/barspoon.hip): the tile side on the 16-bit MFMA path (weights zero-padded like the MIL `vit` head's), the class
tokens in exact fp32.  Eval + no-grad only: training this head (the reference's `LitMilClassificationMixin.step`, :263-321) is NOT built --
a forward that needs gradients raises.
"""
from __future__ import annotations

import ctypes as C
import re

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib, ops
from .mil_core import PackedVit, VitDims, layer_prefix



def sanitize(x: str) -> str:
    return re.sub(r"[^A-Za-z0-9_]", "_", x)                    # barspoon.py:351-352


class EncDecTransformer(nn.Module):
    def __init__(self, d_features: int, target_n_outs: dict[str, int], *, d_model: int = 512, num_encoder_heads: int = 8, num_decoder_heads: int = 8,
                 num_encoder_layers: int = 2, num_decoder_layers: int = 2, dim_feedforward: int = 2048, positional_encoding: bool = True) -> None:
        super().__init__()
        # the reference's module tree (:118-162), as parameter containers
        self.projector = nn.Sequential(nn.Linear(d_features, d_model), nn.ReLU())
        enc = nn.TransformerEncoderLayer(d_model=d_model, nhead=num_encoder_heads, dim_feedforward=dim_feedforward, batch_first=True, norm_first=True)
        self.transformer_encoder = nn.TransformerEncoder(enc, num_layers=num_encoder_layers, enable_nested_tensor=False)
        self.target_labels = target_n_outs.keys()
        self.class_tokens = nn.ParameterDict({sanitize(t): torch.rand(d_model) for t in target_n_outs})
        dec = nn.TransformerDecoderLayer(d_model=d_model, nhead=num_decoder_heads, dim_feedforward=dim_feedforward, batch_first=True, norm_first=True)
        self.transformer_decoder = nn.TransformerDecoder(dec, num_layers=num_decoder_layers)
        self.heads = nn.ModuleDict({sanitize(t): nn.Linear(in_features=d_model, out_features=n) for t, n in target_n_outs.items()})
        self.positional_encoding = positional_encoding
        self.d_features, self.d_model, self.dim_feedforward = d_features, d_model, dim_feedforward
        self.num_encoder_heads, self.num_decoder_heads = num_encoder_heads, num_decoder_heads
        self.num_encoder_layers, self.num_decoder_layers = num_encoder_layers, num_decoder_layers
        self.target_n_outs = dict(target_n_outs)
        if d_model % num_encoder_heads or d_model % num_decoder_heads or d_model // num_encoder_heads > 64 or d_model // num_decoder_heads > 64 or d_model % 4:
            raise NotImplementedError("HIP barspoon needs head_dim <= 64 in both stacks and d_model % 4 == 0")
        self._pack_key = None

    # ---- device weights, rebuilt when a parameter changed ---------------------------------------------------------------------------
    def _pack(self, dev):
        tensors = dict(self.named_parameters())
        key = (str(dev), tuple((t.data_ptr(), t._version) for t in tensors.values()))
        if self._pack_key == key:
            return self._packed
        g = lambda n: tensors[n].detach().to(dev, torch.float32).contiguous()  # noqa: E731
        D, Hd, FF = self.d_model, self.num_decoder_heads, self.dim_feedforward
        dims = VitDims(F=self.d_features, D=D, H=self.num_encoder_heads, FF=FF, C=1, L=self.num_encoder_layers, alibi=False)
        enc_map = {"0.norm.": "norm1.", "0.mhsa.in_proj_": "self_attn.in_proj_", "0.mhsa.out_proj.": "self_attn.out_proj.", "1.0.": "norm2.", "1.1.": "linear1.",
                   "1.4.": "linear2."}

        def vit_get(name: str) -> torch.Tensor:                # the encoder stack under the MIL `vit` head's parameter names (zero-padded by PackedVit)
            if name.startswith("project_features.0."):
                return g("projector.0." + name.rsplit(".", 1)[1])
            for l in range(self.num_encoder_layers):
                p = layer_prefix(l)
                if name.startswith(p):
                    rest = name[len(p):]
                    for a, b in enc_map.items():
                        if rest.startswith(a):
                            return g(f"transformer_encoder.layers.{l}.{b}{rest[len(a):]}")
            f32 = dict(dtype=torch.float32, device=dev)
            return {"class_token": torch.zeros(D, **f32), "transformer.norm.weight": torch.ones(D, **f32), "transformer.norm.bias": torch.zeros(D, **f32),
                    "mlp_head.0.weight": torch.zeros(1, D, **f32), "mlp_head.0.bias": torch.zeros(1, **f32)}[name]

        pk = PackedVit(dims, vit_get, torch.float16, train=False)
        pk.c_structs()
        enc_layers = pk._c[2]
        keep = [pk]

        def T(t):
            keep.append(t)
            return t.data_ptr()

        hd, Dp, Db = D // Hd, dims.Dp, 64 * Hd
        dec = (_lib.BarspoonDecLayer * max(self.num_decoder_layers, 1))()
        for l in range(self.num_decoder_layers):
            p = f"transformer_decoder.layers.{l}."
            w, b = g(p + "multihead_attn.in_proj_weight"), g(p + "multihead_attn.in_proj_bias")
            kvw = F.pad(w[D:].view(2, Hd, hd, D), (0, Dp - D, 0, 64 - hd)).reshape(2 * Db, Dp).contiguous()
            kvb = F.pad(b[D:].view(2, Hd, hd), (0, 64 - hd)).reshape(2 * Db).contiguous()
            dec[l] = _lib.BarspoonDecLayer(T(g(p + "norm1.weight")), T(g(p + "norm1.bias")), T(g(p + "self_attn.in_proj_weight")), T(g(p + "self_attn.in_proj_bias")),
                                           T(g(p + "self_attn.out_proj.weight")), T(g(p + "self_attn.out_proj.bias")), T(g(p + "norm2.weight")), T(g(p + "norm2.bias")),
                                           T(w[:D].contiguous()), T(b[:D].contiguous()), T(ops.cast_pad(kvw, Dp, torch.float16)), T(kvb),
                                           T(g(p + "multihead_attn.out_proj.weight")), T(g(p + "multihead_attn.out_proj.bias")), T(g(p + "norm3.weight")),
                                           T(g(p + "norm3.bias")), T(g(p + "linear1.weight")), T(g(p + "linear1.bias")), T(g(p + "linear2.weight")), T(g(p + "linear2.bias")))
        labels = [sanitize(t) for t in self.target_labels]
        nt = len(labels)
        hw, hb, no = (C.c_void_p * nt)(), (C.c_void_p * nt)(), (C.c_int * nt)()
        for j, t in enumerate(labels):
            hw[j], hb[j], no[j] = T(g(f"heads.{t}.weight")), T(g(f"heads.{t}.bias")), self.heads[t].out_features
        ct = torch.stack([g(f"class_tokens.{t}") for t in labels]).contiguous()
        pe = (100_000 ** (torch.arange(D // 4, dtype=torch.float32) / D)).to(dev).contiguous()              # :176-178, in torch's own fp32 arithmetic
        wc = _lib.BarspoonWeights(pk.w["proj_w"].data_ptr(), pk.m["proj_b"].data_ptr(), enc_layers, T(ct), dec, hw, hb, no, T(pe))
        cfg = _lib.BarspoonCfg(self.d_features, D, self.num_encoder_heads, Hd, FF, self.num_encoder_layers, self.num_decoder_layers, nt,
                               int(bool(self.positional_encoding)), _lib.F16)
        self._packed, self._pack_key = (cfg, wc, keep, dec, hw, hb, no, sum(no)), key
        return self._packed

    def forward(self, tile_tokens: torch.Tensor, tile_positions: torch.Tensor) -> dict[str, torch.Tensor]:
        if not tile_tokens.is_cuda:
            raise RuntimeError("HIP barspoon needs bags on the GPU (no CPU fallback)")
        if self.training or (torch.is_grad_enabled() and (tile_tokens.requires_grad or any(p.requires_grad for p in self.parameters()))):
            raise NotImplementedError("HIP barspoon: only the deploy / validation forward is built -- call it under .eval() and torch.no_grad() "
                                      "(training this head is not implemented)")
        if tile_tokens.dim() != 3 or tile_tokens.shape[-1] != self.d_features:
            raise ValueError(f"tile_tokens must be [batch, tile, {self.d_features}], got {tuple(tile_tokens.shape)}")
        Bb, T, _ = tile_tokens.shape
        dev = tile_tokens.device
        if self.positional_encoding and (tile_positions is None or tile_positions.shape != (Bb, T, 2)):
            raise ValueError(f"tile_positions must be [batch, tile, 2] = {(Bb, T, 2)}")
        x = tile_tokens if tile_tokens.dtype in ops._DT else tile_tokens.float()
        x = x.contiguous()
        pos = tile_positions.to(dev, torch.float32).contiguous() if tile_positions is not None else None
        cfg, wc, _keep, _dec, _hw, _hb, no, total = self._pack(dev)
        lib = _lib.lib()
        need = lib.amds_barspoon_workspace_bytes(C.byref(cfg), Bb, T)
        if need == 0:
            _lib.check(-1, "barspoon_workspace_bytes")
        ws = ops.scratch("barspoon", dev, need)
        logits = torch.empty(Bb, total, dtype=torch.float32, device=dev)
        _lib.check(lib.amds_barspoon_forward(C.byref(cfg), C.byref(wc), x.data_ptr(), ops._DT[x.dtype], pos.data_ptr() if pos is not None else None,
                                             logits.data_ptr(), Bb, T, ws.data_ptr(), ws.numel(), ops._stream()), "barspoon_forward")
        out, col = {}, 0
        for j, t in enumerate(self.target_labels):
            out[t] = logits[:, col:col + no[j]]
            col += no[j]
        return out
