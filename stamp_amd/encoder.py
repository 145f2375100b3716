"""Encoder seam: gated-attention (CHIEF-style) slide / patient encoder on the HIP path.

Mirrors the reference's `Encoder` contract (src/stamp/encoding/encoder/__init__.py:29-229): constructor fields
``model, identifier, precision, required_extractors`` and the two methods the base class calls,
``_generate_slide_embedding(feats, device, coords=..., **kw) -> np.ndarray[D]`` (:164-171) and
``_generate_patient_embedding(feats_list, device, **kw)``.  The arithmetic is reference
src/stamp/encoding/encoder/chief.py:74-89, 255-275 (only ``WSI_feature = softmax(A) @ h_ori`` is consumed by STAMP,
chief.py:123-127).  Weights: the reference checkpoint's ``attention_net.*`` tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

_KEYS = {"fc_w": "attention_net.0.weight", "fc_b": "attention_net.0.bias",
         "a_w": "attention_net.3.attention_a.0.weight", "a_b": "attention_net.3.attention_a.0.bias",
         "b_w": "attention_net.3.attention_b.0.weight", "b_b": "attention_net.3.attention_b.0.bias",
         "c_w": "attention_net.3.attention_c.weight", "c_b": "attention_net.3.attention_c.bias"}


class HipGatedAttentionEncoder:
    def __init__(self, state_dict: dict[str, torch.Tensor], *, identifier: str = "chief",
                 required_extractors: tuple[str, ...] = ("chief-ctranspath",), device="cuda") -> None:
        self.identifier = identifier
        self.precision = torch.float32              # chief.py:117
        self.required_extractors = list(required_extractors)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipGatedAttentionEncoder runs on the GPU only (no CPU fallback)")
        missing = [v for v in _KEYS.values() if v not in state_dict]
        if missing:
            raise KeyError(f"state_dict lacks {missing}")
        self.weights = {k: state_dict[v].detach().to(self.device, torch.float32).contiguous() for k, v in _KEYS.items()}
        self.model = self                            # the base class only does .to(device).eval() on it

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def attention_raw(self, feats: torch.Tensor) -> torch.Tensor:
        return ops.gated_attn_pool(feats.to(self.device, torch.float32).contiguous(), self.weights, return_attn=True)[1]

    @torch.no_grad()
    def _generate_slide_embedding(self, feats: torch.Tensor, device=None, **kwargs) -> np.ndarray:
        if feats.dim() != 2 or feats.shape[0] == 0:
            raise ValueError(f"expected a non-empty [N, F] feature matrix, got {tuple(feats.shape)}")
        out = ops.gated_attn_pool(feats.to(self.device, torch.float32).contiguous(), self.weights)
        return out.detach().cpu().numpy()

    @torch.no_grad()
    def _generate_patient_embedding(self, feats_list: list, device=None, **kwargs) -> np.ndarray:
        return self._generate_slide_embedding(torch.cat([f.to(self.device) for f in feats_list], dim=0))
