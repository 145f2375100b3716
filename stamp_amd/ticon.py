"""The reference's H-optimus-1 + TICON tile extractor (src/stamp/preprocessing/extractor/ticon.py: `HOptimusTICON` :626-718, factory `ticon()`
:721-741, `ExtractorName.TICON`) on the HIP path.

Stage 1 is the ViT-g/14 trunk the package already runs (`stamp_amd.vit.PRESETS["h_optimus_0"]`: H-optimus-0 and -1 share the architecture and the
normalisation constants, h_optimus_1.py:15-32).  Stage 2 -- TICON's `EncoderDecoder` called with ONE token per tile and zero coordinates, as
the reference's `forward` does (:697-718) -- is `HipTiconTile`: one library call (`amds_ticon_tile_forward`, csrc/ticon.hip) in exact fp32 on the
batch of tile embeddings.  With a single key the attention returns its value, so only `v_proj` and `proj` of every attention module carry
arithmetic; `q_proj` / `k_proj`, the decoder and the output projections of the checkpoint are not on this path and are ignored.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib, ops
from .vit import PRESETS, HipViT



class HipTiconTile(nn.Module):
    """TICON on single tiles: `forward(emb [B, in_dim] f16 / f32 on the GPU) -> [B, dim]` (f32, or f16 with `out_dtype=torch.float16`).
    `state_dict`: the `EncoderDecoder`'s own (the reference strips the checkpoint's "backbone." prefix, ticon.py:614-619); `key`: which input
    projection to use ("hoptimus1" in the reference's extractor)."""

    def __init__(self, state_dict: dict[str, torch.Tensor], *, key: str = "hoptimus1", device="cuda", out_dtype: torch.dtype = torch.float32) -> None:
        super().__init__()
        self.device_ = torch.device(device)
        if self.device_.type != "cuda":
            raise RuntimeError("HipTiconTile runs on the GPU only (no CPU fallback)")
        self.key, self.out_dtype = key, out_dtype
        p = f"input_proj_dict.input_proj_{key}."
        need = [p + n for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "norm.weight", "norm.bias")] + ["enc_norm.weight", "enc_norm.bias"]
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"TICON state_dict lacks {missing}")
        g = lambda n: state_dict[n].detach().to(self.device_, torch.float32).contiguous()  # noqa: E731
        self._keep: list[torch.Tensor] = []

        def T(t: torch.Tensor) -> int:
            t = t.contiguous()
            self._keep.append(t)
            return t.data_ptr()

        depth = 0
        while f"encoder.blocks.{depth}.residual1.norm.weight" in state_dict:
            depth += 1
        self.in_dim, self.dim = state_dict[p + "fc1.weight"].shape[1], state_dict[p + "fc1.weight"].shape[0]
        self.hidden = state_dict["encoder.blocks.0.residual2.fn.fc1.weight"].shape[0] if depth else 2
        self.depth = depth
        self._blocks = (_lib.TiconBlock * max(depth, 1))()
        for l in range(depth):
            b = f"encoder.blocks.{l}."
            g1 = g(b + "residual1.gamma") if (b + "residual1.gamma") in state_dict else torch.ones(self.dim, device=self.device_)
            g2 = g(b + "residual2.gamma") if (b + "residual2.gamma") in state_dict else torch.ones(self.dim, device=self.device_)
            self._blocks[l] = _lib.TiconBlock(T(g(b + "residual1.norm.weight")), T(g(b + "residual1.norm.bias")), T(g(b + "residual1.fn.v_proj.weight")),
                                              T(g(b + "residual1.fn.v_proj.bias")), T(g(b + "residual1.fn.proj.weight") * g1[:, None]), T(g(b + "residual1.fn.proj.bias") * g1),
                                              T(g(b + "residual2.norm.weight")), T(g(b + "residual2.norm.bias")), T(g(b + "residual2.fn.fc1.weight")),
                                              T(g(b + "residual2.fn.fc1.bias")), T(g(b + "residual2.fn.fc2.weight") * g2[:, None]), T(g(b + "residual2.fn.fc2.bias") * g2))
        self._w = _lib.TiconWeights(self.in_dim, self.dim, self.hidden, depth, T(g(p + "fc1.weight")), T(g(p + "fc1.bias")), T(g(p + "fc2.weight")), T(g(p + "fc2.bias")),
                                    T(g(p + "norm.weight")), T(g(p + "norm.bias")), self._blocks, T(g("enc_norm.weight")), T(g("enc_norm.bias")))

    @torch.no_grad()
    def forward(self, emb: torch.Tensor) -> torch.Tensor:
        if not emb.is_cuda:
            raise RuntimeError("HipTiconTile needs its input on the GPU (no CPU fallback)")
        if emb.dim() != 2 or emb.shape[1] != self.in_dim:
            raise ValueError(f"expected tile embeddings [batch, {self.in_dim}], got {tuple(emb.shape)}")
        if emb.dtype not in (torch.float32, torch.float16):
            emb = emb.float()
        emb = emb.contiguous()
        B, dev = emb.shape[0], emb.device
        lib = _lib.lib()
        need = lib.amds_ticon_tile_workspace_bytes(C.byref(self._w), B)
        if need == 0 and B > 0:
            _lib.check(-1, "ticon_tile_workspace_bytes")
        ws = ops.scratch("ticon", dev, need)
        out = torch.empty(B, self.dim, dtype=self.out_dtype, device=dev)
        _lib.check(lib.amds_ticon_tile_forward(C.byref(self._w), emb.data_ptr(), ops._DT[emb.dtype], out.data_ptr(), ops._DT[self.out_dtype], B, ws.data_ptr(), ws.numel(),
                                               ops._stream()), "ticon_tile_forward")
        return out


class HipHOptimusTicon(nn.Module):
    """`HOptimusTICON` of the reference (ticon.py:626-718): u8 tiles [B, 224, 224, 3] (or what `HipViT` accepts) -> fp16 features [B, 1536]:
    the H-optimus trunk, then TICON on every tile alone.  `vit_state_dict`: timm names of "bioptimus/H-optimus-1" (:634-641);
    `ticon_state_dict`: the TICON backbone (:680-688)."""

    def __init__(self, vit_state_dict: dict[str, torch.Tensor], ticon_state_dict: dict[str, torch.Tensor], *, device="cuda", chunk: int = 512, vit_cfg=None) -> None:
        super().__init__()
        self.vit = HipViT(vit_cfg or PRESETS["h_optimus_0"], vit_state_dict, device=device, chunk=chunk)
        self.ticon = HipTiconTile(ticon_state_dict, key="hoptimus1", device=device, out_dtype=torch.float16)
        if self.ticon.in_dim != self.vit.cfg.dim:
            raise ValueError(f"TICON's hoptimus1 projection takes {self.ticon.in_dim}-d embeddings, the trunk gives {self.vit.cfg.dim}")

    @torch.no_grad()
    def forward(self, tiles: torch.Tensor) -> torch.Tensor:
        return self.ticon(self.vit(tiles))
