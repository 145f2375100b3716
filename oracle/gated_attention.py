"""Oracle for gated-attention pooling.

Restates reference src/stamp/encoding/encoder/chief.py:74-89 (CHIEFModel.forward) and :255-275
(Attn_Net_Gated.forward) in eval mode (the Dropout(0.25) layers are identities).  Pinned by
tests/golden/chief_gated_attention_*.npz, produced by executing the reference's own class definitions
(tools/make_golden.py:golden_chief).  State-dict keys are the reference's:
  attention_net.0.{weight,bias}                     Linear(F, L)   (+ ReLU)
  attention_net.3.attention_a.0.{weight,bias}       Linear(L, D)   (+ Tanh)
  attention_net.3.attention_b.0.{weight,bias}       Linear(L, D)   (+ Sigmoid)
  attention_net.3.attention_c.{weight,bias}         Linear(D, 1)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

KEYS = {"fc_w": "attention_net.0.weight", "fc_b": "attention_net.0.bias",
        "a_w": "attention_net.3.attention_a.0.weight", "a_b": "attention_net.3.attention_a.0.bias",
        "b_w": "attention_net.3.attention_b.0.weight", "b_b": "attention_net.3.attention_b.0.bias",
        "c_w": "attention_net.3.attention_c.weight", "c_b": "attention_net.3.attention_c.bias"}


def gated_attention_pool(x: torch.Tensor, sd: dict) -> dict:
    """x: [N, F] fp32 -> {"WSI_feature": [1, F], "attention_raw": [1, N]}."""
    w = {k: sd[v].float() for k, v in KEYS.items()}
    h = F.relu(F.linear(x, w["fc_w"], w["fc_b"]))                    # chief.py:45 (fc) ; dropout = id
    a = torch.tanh(F.linear(h, w["a_w"], w["a_b"]))                  # chief.py:271
    b = torch.sigmoid(F.linear(h, w["b_w"], w["b_b"]))               # chief.py:272
    A = F.linear(a * b, w["c_w"], w["c_b"])                          # chief.py:273-274  [N,1]
    A_raw = A.transpose(1, 0)                                        # chief.py:77-78    [1,N]
    P = torch.softmax(A_raw, dim=1)                                  # chief.py:79
    return {"WSI_feature": P @ x, "attention_raw": A_raw}            # chief.py:82 (pool the ORIGINAL features)
