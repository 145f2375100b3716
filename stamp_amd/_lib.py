"""ctypes binding of libamdstamp.so (the C ABI declared in include/amdstamp.h).

The library is the product; there is no Python/torch fallback.  `lib()` raises if the shared object is
missing or does not export a declared symbol, and every wrapper raises ``RuntimeError`` with
``amds_last_error()`` when a call returns non-zero -- STAMP's per-slide ``try/except`` then logs and skips
the slide exactly as it does for the reference model (reference src/stamp/preprocessing/__init__.py:328-336).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libamdstamp.so"

F16, BF16, F32 = 0, 1, 2
ERR_RANGE = -5          # amds_check_finite: non-finite values in a result
(EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_RESIDUAL, EPI_BIAS_F32, EPI_SWIGLU, EPI_PATCH,
 EPI_BIAS_GELU_F32, EPI_BIAS_RELU_F32) = range(9)


class VitCfg(C.Structure):
    _fields_ = [("img", C.c_int), ("patch", C.c_int), ("dim", C.c_int), ("depth", C.c_int),
                ("heads", C.c_int), ("hidden", C.c_int), ("n_prefix", C.c_int), ("mlp_kind", C.c_int),
                ("layerscale", C.c_int), ("dtype", C.c_int), ("ln_eps", C.c_float)]


class VitBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "qkv_colsum", "fc1_colsum")]


class VitExactBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_w", "q_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VitFp8Block(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("qkv_w8", "qkv_cs", "proj_w8", "proj_cs", "proj_b", "fc1_w8", "fc1_cs")] + [("fc1_wnorm_max", C.c_float), ("fc1_babs_max", C.c_float)]
                + [(n, C.c_void_p) for n in ("fc2_w8", "fc2_cs", "fc2_b")])


class VitWeights(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("prefix", C.c_void_p),
                ("pos_patch", C.c_void_p), ("blocks_host", C.POINTER(VitBlock)),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("patch_lo_shift", C.c_int),
                ("exact_host", C.POINTER(VitExactBlock)), ("exact_hidden", C.c_int), ("fp8_host", C.POINTER(VitFp8Block)),
                ("pre_norm_w", C.c_void_p), ("pre_norm_b", C.c_void_p), ("cls_tail", C.POINTER(VitExactBlock))]


class VitHostBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "norm2_w", "norm2_b",
                                          "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2")]


class VitHostWeights(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("cls_token", C.c_void_p), ("reg_token", C.c_void_p), ("pos_embed", C.c_void_p),
                ("blocks", C.POINTER(VitHostBlock)), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("hidden", C.c_int), ("no_embed_class", C.c_int),
                ("mean", C.c_double * 3), ("std", C.c_double * 3)]


PACK_LNFOLD, PACK_PATCH_SPLIT, PACK_EXACT, PACK_CLS_TAIL = 1, 2, 4, 8


class CastEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld_src", C.c_long), ("dst", C.c_void_p), ("ld_dst", C.c_long), ("dst_t", C.c_void_p), ("ld_dst_t", C.c_long),
                ("rows", C.c_int), ("cols", C.c_int), ("dtype", C.c_int)]


class ColsumEntry(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("ld", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("kind", C.c_int)]


class MilVitCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_feats", "dim", "heads", "ff", "classes", "layers", "alibi", "dtype")]


class MilVitLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "head_scale", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                                          "fc2_w", "fc2_b", "in_wt", "out_wt", "fc1_wt", "fc2_wt", "bias_scale", "inv_running_mean")]


class MilVitWeights(C.Structure):
    _fields_ = [("class_token", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("layers_host", C.POINTER(MilVitLayer)),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("proj_wt", C.c_void_p)]


class MilVitDropout(C.Structure):
    _fields_ = [("p_proj", C.c_float), ("p_att", C.c_float), ("p_ff", C.c_float), ("seed", C.c_uint64), ("cls_tail", C.c_int)]


class MilVitLayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "bias_scale", "ln2_w", "ln2_b", "fc1_w", "fc1_b",
                                          "fc2_w", "fc2_b")]


class MilVitGrads(C.Structure):
    _fields_ = [("class_token", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("layers_host", C.POINTER(MilVitLayerGrads)),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p)]


class TransMilCfg(C.Structure):
    _fields_ = [("n_feats", C.c_int), ("dim", C.c_int), ("classes", C.c_int), ("train_cls_tail", C.c_int)]


class TransMilLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_w", "norm_b", "qkv_w", "out_w", "out_b", "conv_w")]


class TransMilWeights(C.Structure):
    _fields_ = [("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("cls_token", C.c_void_p), ("layer", TransMilLayer * 2)] + \
               [(n, C.c_void_p) for n in ("ppeg_w7", "ppeg_b7", "ppeg_w5", "ppeg_b5", "ppeg_w3", "ppeg_b3", "norm_w", "norm_b", "fc2_w", "fc2_b")]


class TransMilLayerGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_w", "norm_b", "qkv_w", "out_w", "out_b", "conv_w")]


class TransMilGrads(C.Structure):
    _fields_ = [("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("cls_token", C.c_void_p), ("layer", TransMilLayerGrads * 2), ("ppeg_corr", C.c_void_p),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p)]


class NystromGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "out_w", "out_b", "conv_w")]


class BarspoonCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_feats", "dim", "enc_heads", "dec_heads", "ff", "enc_layers", "dec_layers", "n_targets", "positional_encoding", "dtype")]


class BarspoonDecLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ln2_w", "ln2_b", "ca_q_w", "ca_q_b", "ca_kv_w", "ca_kv_b",
                                          "ca_out_w", "ca_out_b", "ln3_w", "ln3_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class BarspoonWeights(C.Structure):
    _fields_ = [("proj_w", C.c_void_p), ("proj_b", C.c_void_p), ("enc_layers_host", C.POINTER(MilVitLayer)), ("class_tokens", C.c_void_p),
                ("dec_layers_host", C.POINTER(BarspoonDecLayer)), ("head_w_host", C.POINTER(C.c_void_p)), ("head_b_host", C.POINTER(C.c_void_p)),
                ("n_out_host", C.POINTER(C.c_int)), ("pe_div", C.c_void_p)]


class TiconBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "v_w", "v_b", "proj_w", "proj_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class TiconWeights(C.Structure):
    _fields_ = [("in_dim", C.c_int), ("dim", C.c_int), ("hidden", C.c_int), ("depth", C.c_int)] + \
               [(n, C.c_void_p) for n in ("in_fc1_w", "in_fc1_b", "in_fc2_w", "in_fc2_b", "in_norm_w", "in_norm_b")] + \
               [("blocks_host", C.POINTER(TiconBlock)), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p)]


class SwinCfg(C.Structure):
    _fields_ = [("img", C.c_int), ("embed", C.c_int), ("n_stages", C.c_int), ("depths", C.c_int * 4),
                ("heads", C.c_int * 4), ("dtype", C.c_int), ("ln_eps", C.c_float)]


class SwinBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "qkv_w", "qkv_b", "bias_lane", "proj_w", "proj_b",
        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "mlp_pack")]


class SwinMerge(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln_w", "ln_b", "red_w")]


class SwinWeights(C.Structure):
    _fields_ = [("stem", C.c_void_p), ("blocks_host", C.POINTER(SwinBlock)), ("n_blocks", C.c_int),
                ("merges", SwinMerge * 3), ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("mask_bits", C.c_void_p)]


class GapWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fc_w", "fc_b", "a_w", "a_b", "b_w", "b_b", "c_w", "c_b", "packed")]


_vp, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t
_u64, _u32 = C.c_uint64, C.c_uint32

# name -> (restype, argtypes); must list every symbol include/amdstamp.h declares
PROTOTYPES = {
    "amds_version": (_i, []),
    "amds_last_error": (C.c_char_p, []),
    "amds_device_info": (_i, [_i, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_sz)]),
    "amds_create": (_vp, [_i]),
    "amds_destroy": (None, [_vp]),
    "amds_ctx_device": (_i, [_vp]),
    "amds_profile_enable": (_i, [_vp, _i]),
    "amds_profile_reset": (_i, [_vp]),
    "amds_profile_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "amds_cast_pad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "amds_layernorm": (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _f, _i, _vp]),
    "amds_gemm": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "amds_gemm_ex": (_i, [_i, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "amds_gemm_fp8": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp]),
    "amds_layernorm_quant_e4m3": (_i, [_vp, _l, _vp, _vp, _f, _vp, _l, _vp, _vp, _i, _i, _vp]),
    "amds_row_bound_scale": (_i, [_vp, _f, _f, _vp, _i, _vp]),
    "amds_gemm_fp8_out8": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp]),
    "amds_quantize_rows_e4m3": (_i, [_vp, _l, _vp, _l, _vp, _i, _i, _i, _vp]),
    "amds_gemm_lnfold": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amds_ln_rowstat": (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    "amds_ln_rowstat_diag": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _i, _vp]),
    "amds_vit_workspace_diag_offset": (_sz, [_vp, _i]),
    "amds_check_finite": (_i, [_vp, _l, _i, _vp, _vp, _vp]),
    "amds_export_words": (_i, [_vp, _vp, _l, _i, _vp]),
    "amds_ln_stats_cast": (_i, [_vp, _l, _i, _i, _f, _vp, _l, _vp, _i, _vp]),
    "amds_gemm_lnfold_planes": (_i, [_vp, _l, _vp, _l, _i, _i, _i, _vp, _vp, _l, _vp, _vp, _vp, _vp]),
    "amds_ln_stats_split": (_i, [_vp, _l, _i, _i, _f, _vp, _vp, _l, _vp, _vp]),
    "amds_planes_to_f32": (_i, [_vp, _vp, _l, _l, _vp, _l, _l, _i, _vp]),
    "amds_gemm_rowstream": (_i, [_vp, _l, _vp, _vp, _f, _vp, _l, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp]),
    "amds_swin_mlp96": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp]),
    "amds_swin_mlp192_pack": (_i, [_vp, _vp, _vp, _i, _vp]),
    "amds_swin_mlp192": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp]),
    "amds_pack_swiglu_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "amds_attention_vit": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_attention_row": (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _i, _i, _vp]),
    "amds_attention_row_fwd_train": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_attention_row_bwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_set_mil_cls_tail": (_i, [_vp, _i]),
    "amds_get_mil_cls_tail": (_i, [_vp]),
    "amds_attention": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_attention_vit_hd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amds_attention_alibi": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_vit_workspace_bytes": (_sz, [C.POINTER(VitCfg), _i]),
    "amds_vit_forward": (_i, [C.POINTER(VitCfg), C.POINTER(VitWeights), _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_vit_forward_overlapped": (_i, [_vp, C.POINTER(VitCfg), C.POINTER(VitWeights), _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_vit_forward_tokens": (_i, [C.POINTER(VitCfg), C.POINTER(VitWeights), _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_swin_workspace_bytes": (_sz, [C.POINTER(SwinCfg), _i]),
    "amds_swin_forward": (_i, [C.POINTER(SwinCfg), C.POINTER(SwinWeights), _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_swin_stem": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "amds_window_attention": (_i, [_vp, _l, _vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "amds_swin_attn96": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "amds_patch_merge_ln": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "amds_layernorm_meanpool": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "amds_tile_edge_fraction_u8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_tile_im2col_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amds_tile_im2col_u8_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "amds_vit_pack_bytes": (_sz, [_vp, _vp, _i]),
    "amds_vit_pack": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "amds_vit_pack_host": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "amds_compact_shift_u8": (_i, [_vp, _vp, _l, _i, _i, _vp, _vp]),
    "amds_compact_rows_u8": (_i, [_vp, _l, _vp, _f, _vp, _i, _vp, _vp, _i, _vp]),
    "amds_attention_cls_f32": (_i, [_vp, _l, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp]),
    "amds_vit_cls_gather": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "amds_vit_cls_scatter": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "amds_mlp_act_f32": (_i, [_vp, _l, _i, _i, _i, _vp]),
    "amds_tile_normalize_u8": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_f), C.POINTER(_f), _vp]),
    "amds_macenko_normalize_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp]),
    "amds_supertiles_to_tiles_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amds_supertiles_to_tiles_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "amds_gather_rows": (_i, [_vp, _l, _vp, _i, _vp, _l, _i, _i, _i, _i, _vp]),
    "amds_vary_precision": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "amds_mean_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_linear_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_mil_vit_workspace_bytes": (_sz, [_vp, _i, _i]),
    "amds_mil_vit_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_barspoon_workspace_bytes": (_sz, [_vp, _i, _i]),
    "amds_barspoon_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_ticon_tile_workspace_bytes": (_sz, [_vp, _i]),
    "amds_ticon_tile_forward": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_tile_resize_crop_workspace_bytes": (_sz, [_i, _i, _i]),
    "amds_tile_resize_crop_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "amds_proj_head_l2norm_workspace_bytes": (_sz, [_i, _i, _i]),
    "amds_proj_head_l2norm": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "amds_quick_gelu_inplace": (_i, [_vp, _l, _l, _i, _i, _vp]),
    "amds_transmil_workspace_bytes": (_sz, [_vp, _i, _i]),
    "amds_transmil_train_saved_bytes": (_sz, [_vp, _i, _i]),
    "amds_transmil_train_workspace_bytes": (_sz, [_vp, _i, _i]),
    "amds_transmil_train_forward": (_i, [_vp, _vp, _vp, _i, _f, C.c_uint64, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_transmil_train_backward": (_i, [_vp, _vp, _vp, _f, C.c_uint64, _i, _i, _vp, _sz, _vp, _vp, _vp, _sz, _vp]),
    "amds_nystrom_attn_saved_bytes": (_sz, [_i, _i, _i]),
    "amds_nystrom_attn_workspace_bytes": (_sz, [_i, _i, _i]),
    "amds_nystrom_attn_fwd": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, C.c_uint64, C.c_uint32, _vp, _sz, _vp]),
    "amds_nystrom_attn_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, C.c_uint64, C.c_uint32, _vp, _sz, _vp, _sz, _vp]),
    "amds_transmil_forward": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_mil_vit_train_saved_bytes": (_sz, [_vp, _i, _i]),
    "amds_mil_vit_train_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "amds_mil_vit_train_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_mil_vit_train_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp, _vp, _i, _vp, _sz, _vp]),
    "amds_bgemm_f32": (_i, [_vp, _i, _l, _l, _vp, _i, _l, _l, _i, _vp, _i, _l, _l, _i, _i, _i, _i, _i, _f, _f, _vp, _i, _vp]),
    "amds_bgemm_f32_dual": (_i, [_vp, _i, _l, _l, _vp, _i, _l, _l, _i, _vp, _vp, _i, _l, _l, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp]),
    "amds_softmax_rows": (_i, [_vp, _l, _i, _vp]),
    "amds_landmark_mean": (_i, [_vp, _l, _l, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "amds_pinv_init": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "amds_dwconv_seq_row": (_i, [_vp, _l, _l, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "amds_dwconv_seq": (_i, [_vp, _l, _l, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "amds_ppeg": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_softmax_rows_bwd": (_i, [_vp, _vp, _l, _i, _vp]),
    "amds_landmark_mean_bwd": (_i, [_vp, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "amds_dwconv_seq_wgrad_workspace_bytes": (_sz, [_i, _i, _i]),
    "amds_dwconv_seq_wgrad": (_i, [_vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amds_ppeg_wgrad_workspace_bytes": (_sz, [_i, _i]),
    "amds_ppeg_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amds_relu_bwd": (_i, [_vp, _vp, _vp, _l, _vp]),
    "amds_mean_pool_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "amds_topk_rows_mean": (_i, [_vp, _i, _i, _vp, _l, _i, _i, _vp, _vp, _vp]),
    "amds_pinv_init_bwd_workspace_bytes": (_sz, [_i]),
    "amds_pinv_init_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "amds_set_matmul_precision": (_i, [_vp, _i]),
    "amds_get_matmul_precision": (_i, [_vp]),
    "amds_gemm_batched": (_i, [_vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _i, _i, _vp, _l, _l, _vp, _f, _vp]),
    "amds_wgrad_tn": (_i, [_vp, _l, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp]),
    "amds_transpose16": (_i, [_vp, _l, _vp, _l, _i, _i, _vp]),
    "amds_cast_transpose_multi": (_i, [C.POINTER(CastEntry), _i, _vp]),
    "amds_colsum_workspace_bytes": (_sz, [_i, _i]),
    "amds_colsum": (_i, [_vp, _l, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amds_colsum_partials": (_i, [_vp, _l, _vp, _i, _i, _i, C.POINTER(_i), _vp]),
    "amds_colsum_multi": (_i, [C.POINTER(ColsumEntry), _i, _vp]),
    "amds_sum_partials_multi": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_l), _i, _i, _vp]),
    "amds_layernorm_train": (_i, [_vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _f, _i, _vp]),
    "amds_layernorm_train_copy": (_i, [_vp, _l, _vp, _vp, _vp, _l, _vp, _vp, _i, _i, _f, _i, _vp, _l, _i, _vp]),
    "amds_layernorm_bwd_workspace_bytes": (_sz, [_i, _i]),
    "amds_layernorm_bwd": (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "amds_layernorm_bwd_partials": (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _l, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_layernorm_bwd_cast": (_i, [_vp, _l, _vp, _l, _vp, _vp, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp, _l, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_gelu_fwd": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "amds_gelu_bwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _vp]),
    "amds_attention_fwd_lse": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_attention_alibi_fwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_attention_alibi_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "amds_cdist_rowsum": (_i, [_vp, _vp, _i, _i, _vp]),
    "amds_attention_masked": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amds_attention_alibi_masked": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "amds_attention_fwd_train": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _u32, _vp]),
    "amds_attention_bwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _u32, _vp]),
    "amds_dropout_keep_scale": (_f, [_f]),
    "amds_gelu_dropout_fwd": (_i, [_vp, _vp, _l, _i, _i, _f, _u64, _u32, _vp]),
    "amds_gelu_dropout_bwd": (_i, [_vp, _vp, _vp, _l, _i, _i, _i, _f, _u64, _u32, _vp]),
    "amds_dropout_add": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _i, _f, _u64, _u32, _vp]),
    "amds_attention_row_alibi_fwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amds_attention_row_alibi_bwd_train": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "amds_gelu_dropout_fwd_rows": (_i, [_vp, _l, _vp, _l, _l, _i, _l, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_gelu_dropout_bwd_rows": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _i, _l, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_dropout_add_rows": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _i, _l, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_dropout_cast_bwd_rows": (_i, [_vp, _l, _vp, _l, _l, _i, _l, _f, C.c_uint64, C.c_uint32, _vp]),
    "amds_dropout_cast_bwd": (_i, [_vp, _l, _vp, _l, _l, _i, _i, _f, _u64, _u32, _vp]),
    "amds_dropout_mask": (_i, [_vp, _l, _f, _u64, _u32, _vp]),
    "amds_attention_dropout_mask": (_i, [_vp, _i, _i, _i, _f, _u64, _u32, _vp]),
    "amds_convert_f16_bf16": (_i, [_vp, _vp, _l, _vp]),
    "amds_adamw": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _vp]),
    "amds_gated_attn_pool_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amds_gated_attn_pool": (_i, [_vp, C.POINTER(GapWeights), _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amds_gated_attn_pool_batched_supported": (_i, [_i, _i, _i]),
    "amds_gated_attn_packed_floats": (_sz, [_i, _i, _i]),
    "amds_gated_attn_pack": (_i, [C.POINTER(GapWeights), _vp, _i, _i, _i, _vp]),
    "amds_gated_attn_pool_batched_workspace_bytes": (_sz, [_l, _i, _i, _i, _i]),
    "amds_gated_attn_pool_batched": (_i, [_vp, _vp, _i, _l, C.POINTER(GapWeights), _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "amds_gated_attn_pool_unfused_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amds_gated_attn_pool_unfused": (_i, [_vp, C.POINTER(GapWeights), _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libamdstamp.so once; fail loudly if it (or a declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("AMDSTAMP_LIB", LIB_PATH))
    if not path.is_file():
        raise RuntimeError(
            f"libamdstamp.so not found at {path}: build it with `make` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    handle = C.CDLL(str(path))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise RuntimeError(f"libamdstamp.so does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = handle
    return handle


_ctx: dict[int, int] = {}


def ctx(device: int = 0) -> int:
    """The library context of a device (amds_create): owns the live profiler and the overlapped schedule's side stream."""
    if device not in _ctx:
        h = lib().amds_create(int(device))
        if not h:
            raise RuntimeError("libamdstamp amds_create failed: " + lib().amds_last_error().decode("utf-8", "replace"))
        _ctx[device] = h
    return _ctx[device]


def destroy_contexts() -> None:
    for h in _ctx.values():
        lib().amds_destroy(h)
    _ctx.clear()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().amds_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libamdstamp {what} failed (status {rc}): {msg}")
