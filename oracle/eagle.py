"""CPU restatement of the reference's EAGLE slide encoder -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); the product
path never imports it.

Reference: src/stamp/encoding/encoder/eagle.py -- `Eagle._generate_slide_embedding` :96-120 (CHIEF's gated-attention scores on the
CTransPath features -> the 25 highest-scoring tiles -> mean of THEIR Virchow2 features), `_generate_patient_embedding` :122-134 (all
slides of the patient concatenated), `_align_vir2_to_ctp_by_coords` :265-300 (stable matching of the two files' tiles by rounded
coordinates).  Pinned by tests/golden/eagle.npz, produced by executing those reference functions (tools/make_golden.py::golden_eagle).
"""
from __future__ import annotations

from collections import defaultdict, deque

import numpy as np
import torch

from .gated_attention import gated_attention_pool

TOP_K = 25                                                   # eagle.py:107


def eagle_slide_embedding(feats: torch.Tensor, agg_feats: torch.Tensor, chief_sd: dict) -> tuple[np.ndarray, np.ndarray]:
    """feats [N, 768] fp32 (CTransPath), agg_feats [N, D] (Virchow2) -> (embedding fp32 [D], indices of the selected tiles, descending score)."""
    araw = gated_attention_pool(feats.float(), chief_sd)["attention_raw"].squeeze(0)          # :101-103
    k = min(TOP_K, araw.shape[0])                                                              # :107
    _, top = torch.topk(araw, k)                                                               # :108
    emb = torch.stack([agg_feats[i] for i in top.numpy()]).float().mean(dim=0)                 # :112-116
    return emb.numpy(), top.numpy()


def eagle_patient_embedding(feats_list, agg_list, chief_sd: dict) -> np.ndarray:
    return eagle_slide_embedding(torch.cat(list(feats_list), 0), torch.cat(list(agg_list), 0), chief_sd)[0]     # :131-133


def align_by_coords(ref_coords_um: np.ndarray, other_coords_um: np.ndarray, decimals: int = 5) -> np.ndarray:
    """Permutation `perm` with other[perm[i]] at ref[i]'s (rounded) coordinate; duplicates are matched first come, first served; a missing or
    an extra coordinate is an error (:272-293)."""
    ref = np.round(np.asarray(ref_coords_um, dtype=np.float64), decimals)
    oth = np.round(np.asarray(other_coords_um, dtype=np.float64), decimals)
    buckets: dict = defaultdict(deque)
    for j, key in enumerate(map(tuple, oth)):
        buckets[key].append(j)
    perm = np.empty(ref.shape[0], dtype=np.int64)
    for i, key in enumerate(map(tuple, ref)):
        if not buckets[key]:
            raise ValueError(f"Missing coord in other set: {key}")
        perm[i] = buckets[key].popleft()
    unused = sum(len(q) for q in buckets.values())
    if unused != 0:
        raise ValueError(f"virchow2 features contain {unused} extra coords not in ref.")
    return perm
