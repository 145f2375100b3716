"""Thin torch-tensor wrappers over the C ABI (device pointers + the current HIP stream).

torch is plumbing here: it owns device memory and streams; all arithmetic happens in libamdstamp.so.
Every wrapper requires CUDA(HIP) tensors and raises otherwise -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F16, F32  # noqa: F401

_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("stamp_amd ops need tensors on the GPU (no CPU fallback)")


_SCRATCH: dict = {}


def scratch(tag: str, dev: torch.device, nbytes: int) -> torch.Tensor:
    """A grow-only uint8 workspace for the whole-model library calls, one per (purpose, device, stream, host thread): two forwards that may be in
    flight at once (different streams, or two host threads interleaving launches on one stream) never share scratch memory."""
    import threading
    key = (tag, str(dev), _stream(), threading.get_ident())
    ws = _SCRATCH.get(key)
    if ws is None or ws.numel() < nbytes:
        _SCRATCH[key] = ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
    return ws


def _p(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def act_code(dtype: torch.dtype) -> int:
    if dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"activation dtype must be float16 or bfloat16, got {dtype}")
    return _DT[dtype]


def sync_float32_matmul_precision() -> str:
    """Forward torch's process-wide `torch.get_float32_matmul_precision()` to the library's context of the current device (`amds_set_matmul_precision(ctx, level)`): "highest" (torch's default) = exact
    fp32 MFMA products; "high" / "medium" = every fp32 operand as the sum of two bf16 numbers, three bf16 MFMAs per product (one of torch's two documented
    forms of "high"; >= 16 mantissa bits against TF32's 10).  The reference sets "high" before training (`src/stamp/modeling/train.py:519`) and "medium"
    before deployment (`deploy.py:398`), so a drop-in run gets the bf16 x 3 products exactly where the reference asked torch for cheaper ones.  Called by every
    host entry point in front of `amds_bgemm_f32` (TransMIL / Nystrom, the MLP heads' training GEMMs)."""
    level = torch.get_float32_matmul_precision()
    dev = torch.cuda.current_device() if torch.cuda.is_available() else 0
    _lib.check(_lib.lib().amds_set_matmul_precision(_lib.ctx(dev), 0 if level == "highest" else 1), "set_matmul_precision")
    return level


class float32_matmul_precision:
    """`with ops.float32_matmul_precision("high"): ...` -- sets torch's flag (which the library follows) and restores the previous level on exit."""

    def __init__(self, precision: str):
        self.precision = precision

    def __enter__(self):
        self.prev = torch.get_float32_matmul_precision()
        torch.set_float32_matmul_precision(self.precision)
        sync_float32_matmul_precision()
        return self

    def __exit__(self, *exc):
        torch.set_float32_matmul_precision(self.prev)
        sync_float32_matmul_precision()
        return False


def cast_pad(src: torch.Tensor, ld_dst: int, dtype: torch.dtype, out: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 [rows, cols] -> dtype [rows, ld_dst] with zero padded columns (`out`: a buffer of that shape to refresh in place)."""
    _dev(src)
    src = src.contiguous().float()
    rows, cols = src.shape
    dst = out if out is not None and out.shape == (rows, ld_dst) and out.dtype == dtype and out.is_contiguous() else torch.empty(rows, ld_dst, dtype=dtype, device=src.device)
    _lib.check(_lib.lib().amds_cast_pad(_p(src), cols, _p(dst), ld_dst, rows, cols, _DT[dtype], _stream()), "cast_pad")
    return dst


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
              out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    _dev(x, gamma, beta)
    assert x.dtype == torch.float32 and x.is_contiguous()
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().amds_layernorm(_p(x), cols, _p(gamma), _p(beta), _p(y), cols, rows, cols, eps,
                                         _DT[out_dtype], _stream()), "layernorm")
    return y


def layernorm_rows(x: torch.Tensor, rows: int, cols: int, x_row_stride: int, gamma: torch.Tensor, beta: torch.Tensor,
                   eps: float, out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """LayerNorm of `rows` rows of `cols` elements that start every `x_row_stride` elements of x (e.g. CLS tokens)."""
    _dev(x, gamma, beta)
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = torch.empty(rows, cols, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().amds_layernorm(_p(x), x_row_stride, _p(gamma), _p(beta), _p(y), cols, rows, cols, eps,
                                         _DT[out_dtype], _stream()), "layernorm")
    return y


def gemm(a: torch.Tensor, w: torch.Tensor, epi: int, *, bias=None, scale=None, out=None, pos=None, np_=0, T=0, P=0,
         acc_scale: float = 1.0, cfg: int = -1, n_out: int | None = None) -> torch.Tensor:
    """out = epilogue(a[M,K] @ w[N,K]^T).  See include/amdstamp.h for the epilogues."""
    _dev(a, w, bias, scale, out, pos)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == w.dtype
    dt = act_code(a.dtype)
    f32_out = epi in (_lib.EPI_RESIDUAL, _lib.EPI_BIAS_F32, _lib.EPI_PATCH, _lib.EPI_BIAS_GELU_F32, _lib.EPI_BIAS_RELU_F32)
    if out is None:
        cols = n_out if n_out is not None else (N // 2 if epi == _lib.EPI_SWIGLU else N)
        out = torch.empty(M, cols, dtype=torch.float32 if f32_out else a.dtype, device=a.device)
    ldo = out.stride(-2) if out.dim() >= 2 else out.shape[-1]
    _lib.check(_lib.lib().amds_gemm_ex(cfg, _p(a), a.stride(0), _p(w), w.stride(0), M, N, K, dt, epi, _p(out), ldo,
                                       _p(bias), _p(scale), _p(pos), np_, T, P, acc_scale, _stream()), "gemm")
    return out


def gemm_lnfold(a: torch.Tensor, w: torch.Tensor, epi: int, *, out=None, bias=None, scale=None, xh=None, rowpart=None, rowstat=None,
                colsum=None) -> torch.Tensor:
    """LayerNorm folded into the GEMMs around it (include/amdstamp.h, amds_gemm_lnfold): producer form (RESIDUAL: `xh`, `rowpart`
    filled besides out += ...) or consumer form (BIAS / BIAS_GELU / SWIGLU with `rowstat`, `colsum`)."""
    _dev(a, w, out, bias, scale, xh, rowpart, rowstat, colsum)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == w.dtype
    if out is None:
        assert epi != _lib.EPI_RESIDUAL
        out = torch.empty(M, N // 2 if epi == _lib.EPI_SWIGLU else N, dtype=a.dtype, device=a.device)
    if xh is not None:
        assert xh.dtype == a.dtype and xh.stride(0) == out.stride(0) and rowpart.shape == (M, N // 128, 2) and rowpart.is_contiguous()
    _lib.check(_lib.lib().amds_gemm_lnfold(_p(a), a.stride(0), _p(w), w.stride(0), M, N, K, act_code(a.dtype), epi, _p(out), out.stride(0),
                                           _p(bias), _p(scale), _p(xh), _p(rowpart), _p(rowstat), _p(colsum), _stream()), "gemm_lnfold")
    return out


def ln_rowstat(rowpart: torch.Tensor, D: int, eps: float) -> torch.Tensor:
    """[M, NP, 2] partial (sum, sum of squares) -> [M, 2] (rstd, -mean * rstd)."""
    _dev(rowpart)
    assert rowpart.dtype == torch.float32 and rowpart.is_contiguous() and rowpart.dim() == 3 and rowpart.shape[2] == 2
    out = torch.empty(rowpart.shape[0], 2, dtype=torch.float32, device=rowpart.device)
    _lib.check(_lib.lib().amds_ln_rowstat(_p(rowpart), rowpart.shape[0], rowpart.shape[1], D, eps, _p(out), _stream()), "ln_rowstat")
    return out


def ln_stats_cast(x: torch.Tensor, eps: float, dtype: torch.dtype) -> tuple[torch.Tensor, torch.Tensor]:
    """fp32 rows -> (16-bit copy, [M, 2] (rstd, -mean * rstd)): the first LayerNorm of a stack in the folded form."""
    _dev(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, D = x.shape
    xh = torch.empty(M, D, dtype=dtype, device=x.device)
    rs = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_ln_stats_cast(_p(x), x.stride(0), M, D, eps, _p(xh), D, _p(rs), act_code(dtype), _stream()), "ln_stats_cast")
    return xh, rs


def ln_stats_split(x: torch.Tensor, eps: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """fp32 rows -> (hi = fp16(x), lo = fp16(x - hi), [M, 2] (rstd, -mean * rstd)): the residual stream's two-plane form (amds_ln_stats_split)."""
    _dev(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, D = x.shape
    hi, lo = torch.empty(M, D, dtype=torch.float16, device=x.device), torch.empty(M, D, dtype=torch.float16, device=x.device)
    rs = torch.empty(M, 2, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_ln_stats_split(_p(x), x.stride(0), M, D, eps, _p(hi), _p(lo), D, _p(rs), _stream()), "ln_stats_split")
    return hi, lo, rs


def planes_to_f32(hi: torch.Tensor, lo: torch.Tensor, row_stride: int = 1) -> torch.Tensor:
    """out[i] = hi[i * row_stride] + lo[i * row_stride] in fp32 (amds_planes_to_f32)."""
    _dev(hi, lo)
    assert hi.dtype == lo.dtype == torch.float16 and hi.shape == lo.shape and hi.is_contiguous() and lo.is_contiguous()
    rows = (hi.shape[0] + row_stride - 1) // row_stride
    out = torch.empty(rows, hi.shape[1], dtype=torch.float32, device=hi.device)
    _lib.check(_lib.lib().amds_planes_to_f32(_p(hi), _p(lo), hi.stride(0), row_stride, _p(out), out.stride(0), rows, hi.shape[1], _stream()), "planes_to_f32")
    return out


def gemm_lnfold_planes(a: torch.Tensor, w: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, *, bias=None, scale=None) -> torch.Tensor:
    """(hi, lo) += scale * (a w^T + bias) on the two-plane residual stream, in place; returns the partial row sums [M, N / 128, 2]
    (include/amdstamp.h, amds_gemm_lnfold_planes)."""
    _dev(a, w, hi, lo, bias, scale)
    M, K = a.shape
    N = w.shape[0]
    assert a.dtype == w.dtype == hi.dtype == lo.dtype == torch.float16 and hi.shape == lo.shape == (M, N) and hi.stride(0) == lo.stride(0)
    rowpart = torch.empty(M, N // 128, 2, dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().amds_gemm_lnfold_planes(_p(a), a.stride(0), _p(w), w.stride(0), M, N, K, _p(hi), _p(lo), hi.stride(0), _p(bias), _p(scale),
                                                  _p(rowpart), _stream()), "gemm_lnfold_planes")
    return rowpart


def attention_vit(qkv: torch.Tensor, B: int, T: int, H: int, head_dim: int = 64) -> torch.Tensor:
    _dev(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * head_dim)
    out = torch.empty(B * T, H * head_dim, dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_vit_hd(_p(qkv), _p(out), B, T, H, head_dim, act_code(qkv.dtype), _stream()), "attention_vit")
    return out


def attention(qkv: torch.Tensor, B: int, T: int, H: int) -> torch.Tensor:
    """softmax(q k^T / 8) v for any T (streaming K/V kernel); qkv [B*T, 3*H*64] -> [B*T, H*64]."""
    _dev(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * 64)
    out = torch.empty(B * T, H * 64, dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.lib().amds_attention(_p(qkv), _p(out), B, T, H, act_code(qkv.dtype), _stream()), "attention")
    return out


def set_mil_cls_tail(on: bool, device: int | None = None) -> bool:
    """The class-row tail of the MIL `vit` head / TransMIL deploy forward, a setting of the library's context of `device` (default: the current one;
    `amds_set_mil_cls_tail(ctx, on)`; default on, AMDS_MIL_CLS_TAIL=0 turns the default off): returns the previous setting."""
    lib = _lib.lib()
    cx = _lib.ctx(torch.cuda.current_device() if device is None else device)
    prev = bool(lib.amds_get_mil_cls_tail(cx))
    _lib.check(lib.amds_set_mil_cls_tail(cx, 1 if on else 0), "set_mil_cls_tail")
    return prev


def attention_row(q: torch.Tensor, qkv: torch.Tensor, B: int, T: int, H: int) -> torch.Tensor:
    """One query row per (bag, head) -- q [B, H*64] (16-bit) -- against all T keys / values of the packed qkv [B*T, 3*H*64]: the class token's attention in the
    last block of the MIL `vit` head (`amds_attention_row`).  -> [B, H*64]."""
    _dev(q, qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * 64) and q.shape == (B, H * 64) and q.dtype == qkv.dtype and q.stride(1) == 1
    out = torch.empty(B, H * 64, dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_row(_p(q), q.stride(0), _p(qkv), _p(out), H * 64, B, T, H, act_code(qkv.dtype), _stream()), "attention_row")
    return out


def attention_alibi(qkv: torch.Tensor, coords: torch.Tensor, head_scale: torch.Tensor, B: int, T: int, H: int) -> torch.Tensor:
    """softmax(q k^T/8) v - head_scale[h] * cdist(coords, coords) v; coords fp32 [B,T,2], head_scale fp32 [H]."""
    _dev(qkv, coords, head_scale)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * 64)
    assert coords.dtype == torch.float32 and coords.is_contiguous() and coords.shape == (B, T, 2)
    assert head_scale.dtype == torch.float32 and head_scale.numel() == H
    out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device=qkv.device)     # bf16: see include/amdstamp.h
    _lib.check(_lib.lib().amds_attention_alibi(_p(qkv), _p(coords), _p(head_scale), _p(out), B, T, H, act_code(qkv.dtype),
                                               _stream()), "attention_alibi")
    return out


def quantize_rows_e4m3(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """f16 / fp32 [rows, cols] -> (e4m3 bytes [rows, cols] as uint8, fp32 scale [rows]): q = e4m3(x / scale), scale = rowmax|x| / 448."""
    _dev(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.float16, torch.float32)
    rows, cols = x.shape
    q = torch.empty(rows, cols, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_quantize_rows_e4m3(_p(x), x.stride(0), _p(q), cols, _p(sc), rows, cols, _DT[x.dtype], _stream()), "quantize_rows_e4m3")
    return q, sc


def layernorm_quant_e4m3(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, want_norm: bool = False):
    """nn.LayerNorm of fp32 rows fused with the per-row e4m3 quantisation -> (q uint8 [rows, cols], scale [rows][, L2 norm of the normalised rows])."""
    _dev(x, gamma, beta)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    q = torch.empty(rows, cols, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    nrm = torch.empty(rows, dtype=torch.float32, device=x.device) if want_norm else None
    _lib.check(_lib.lib().amds_layernorm_quant_e4m3(_p(x), x.stride(0), _p(gamma), _p(beta), eps, _p(q), cols, _p(sc), _p(nrm), rows, cols, _stream()), "layernorm_quant_e4m3")
    return (q, sc, nrm) if want_norm else (q, sc)


def row_bound_scale(rownorm: torch.Tensor, c0: float, c1: float) -> torch.Tensor:
    _dev(rownorm)
    us = torch.empty_like(rownorm)
    _lib.check(_lib.lib().amds_row_bound_scale(_p(rownorm), float(c0), float(c1), _p(us), rownorm.numel(), _stream()), "row_bound_scale")
    return us


def gemm_fp8_out8(a8: torch.Tensor, w8: torch.Tensor, epi: int, out_rowscale: torch.Tensor, *, rowscale=None, colscale=None, bias=None) -> torch.Tensor:
    """amds_gemm_fp8 with an e4m3 OUTPUT: out8[m][n] = e4m3(act(...) / out_rowscale[m]) -> uint8 [M, N]."""
    _dev(a8, w8, out_rowscale, rowscale, colscale, bias)
    M, K = a8.shape
    N = w8.shape[0]
    out8 = torch.empty(M, N, dtype=torch.uint8, device=a8.device)
    _lib.check(_lib.lib().amds_gemm_fp8_out8(_p(a8), a8.stride(0), _p(w8), w8.stride(0), M, N, K, epi, _p(out8), N, _p(out_rowscale), _p(bias), _p(colscale), _p(rowscale),
                                             _stream()), "gemm_fp8_out8")
    return out8


def gemm_fp8(a8: torch.Tensor, w8: torch.Tensor, epi: int, *, rowscale=None, colscale=None, bias=None, out=None) -> torch.Tensor:
    """OPT-IN fp8 GEMM: act((a8 @ w8^T) * rowscale[:, None] * colscale[None, :] + bias); a8 [M, K], w8 [N, K] e4m3 bytes (uint8).
    epi EPI_BIAS / EPI_BIAS_GELU -> f16 [M, N]; EPI_SWIGLU (w8 = the 32-row interleaved packed fc1) -> f16 [M, N / 2]; EPI_RESIDUAL -> `out` fp32 [M, N] += ..."""
    _dev(a8, w8, rowscale, colscale, bias, out)
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.stride(1) == 1 and w8.stride(1) == 1
    M, K = a8.shape
    N = w8.shape[0]
    if epi == _lib.EPI_RESIDUAL:
        assert out is not None and out.dtype == torch.float32 and out.shape == (M, N)
    elif out is None:
        out = torch.empty(M, N // 2 if epi == _lib.EPI_SWIGLU else N, dtype=torch.float16, device=a8.device)
    _lib.check(_lib.lib().amds_gemm_fp8(_p(a8), a8.stride(0), _p(w8), w8.stride(0), M, N, K, epi, _p(out), out.stride(0), _p(bias), _p(colscale), _p(rowscale),
                                        _stream()), "gemm_fp8")
    return out


def attention_cls_f32(q: torch.Tensor, qkv: torch.Tensor, B: int, T: int, H: int, head_dim: int = 64) -> torch.Tensor:
    """ONE fp32 query row per tile (q [B, H*head_dim]) against the stored keys / values of all T tokens (packed act-dtype qkv) -> fp32 [B, H*hd]."""
    _dev(q, qkv)
    assert q.dtype == torch.float32 and q.is_contiguous() and qkv.is_contiguous() and q.shape == (B, H * head_dim)
    out = torch.empty(B, H * head_dim, dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().amds_attention_cls_f32(_p(q), H * head_dim, _p(qkv), _p(out), H * head_dim, B, T, H, head_dim, act_code(qkv.dtype), _stream()),
               "attention_cls_f32")
    return out


def vit_cls_gather(x: torch.Tensor, B: int, T: int) -> torch.Tensor:
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    D = x.shape[-1]
    xc = torch.empty(B, D, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_vit_cls_gather(_p(x), _p(xc), B, T, D, _stream()), "vit_cls_gather")
    return xc


def vit_cls_scatter(xc: torch.Tensor, x: torch.Tensor, T: int, eps: float, xh: torch.Tensor | None = None, rowstat: torch.Tensor | None = None) -> None:
    """x[b*T] = xc[b] (in place); with xh / rowstat also the 16-bit copy of those rows and their (rstd, -mean*rstd)."""
    _dev(xc, x, xh, rowstat)
    B, D = xc.shape
    code = act_code(xh.dtype) if xh is not None else F16
    _lib.check(_lib.lib().amds_vit_cls_scatter(_p(xc), _p(x), _p(xh), _p(rowstat), B, T, D, eps, code, _stream()), "vit_cls_scatter")


def mlp_act_f32(u: torch.Tensor, hidden: int, kind: int) -> None:
    """in place on fp32 [rows, ld]: kind 0 GELU(erf) over `hidden` columns; kind 1 SwiGLUPacked: u[:, j] = silu(u[:, j]) * u[:, hidden + j]."""
    _dev(u)
    assert u.dtype == torch.float32 and u.is_contiguous() and u.dim() == 2
    _lib.check(_lib.lib().amds_mlp_act_f32(_p(u), u.shape[1], u.shape[0], hidden, kind, _stream()), "mlp_act_f32")


def pack_swiglu_rows(w: torch.Tensor) -> torch.Tensor:
    """[2H, cols] fp32 (gate rows then value rows) -> 32-row block-interleaved layout."""
    _dev(w)
    w2 = w.contiguous().float().reshape(w.shape[0], -1)
    H, cols = w2.shape[0] // 2, w2.shape[1]
    dst = torch.empty_like(w2)
    _lib.check(_lib.lib().amds_pack_swiglu_rows(_p(w2), _p(dst), H, cols, _stream()), "pack_swiglu_rows")
    return dst.reshape(w.shape)


def tile_im2col_u8(tiles: torch.Tensor, patch: int, kp: int, dtype: torch.dtype, lo_shift: int = 0) -> torch.Tensor:
    """lo_shift > 0: rows of 2*kp, the second half = the first times 2^-lo_shift (the split patch-embedding form)."""
    _dev(tiles)
    assert tiles.dtype == torch.uint8 and tiles.is_contiguous() and tiles.shape[-1] == 3
    B, img = tiles.shape[0], tiles.shape[1]
    g = img // patch
    out = torch.empty(B * g * g, kp * (2 if lo_shift else 1), dtype=dtype, device=tiles.device)
    _lib.check(_lib.lib().amds_tile_im2col_u8_ex(_p(tiles), _p(out), B, img, patch, kp, act_code(dtype), int(lo_shift), _stream()), "im2col")
    return out


def tile_normalize_u8(tiles: torch.Tensor, mean, std) -> torch.Tensor:
    """u8 [B,H,W,3] -> fp32 [B,3,H,W] = (x/255 - mean)/std  (ToTensor + Normalize)."""
    _dev(tiles)
    assert tiles.dtype == torch.uint8 and tiles.is_contiguous() and tiles.shape[-1] == 3
    B, H, W, _ = tiles.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=tiles.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.lib().amds_tile_normalize_u8(_p(tiles), _p(out), B, H, W, m, s, _stream()), "tile_normalize")
    return out


_GAP_KEYS = ("fc_w", "fc_b", "a_w", "a_b", "b_w", "b_b", "c_w", "c_b")
_gap_ws: dict = {}


def _gap_workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """One grow-only workspace per device and stream for the pooling entries (they are called once per slide: a fresh allocation per call was the round-5 form)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _gap_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _gap_ws[key] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
    return ws


_gap_packed: dict = {}


def _gap_weights(weights: dict[str, torch.Tensor], pack: bool = True):
    """The C struct for a weights dict.  The fused kernels' tile-ordered copy of fc_w / a_w / b_w (amds_gated_attn_pack) is made once per set of tensors and
    kept while they are unchanged (data pointers and torch's in-place version counters)."""
    for k in _GAP_KEYS:
        assert weights[k].dtype == torch.float32 and weights[k].is_contiguous(), k
    gw = _lib.GapWeights(**{k: _p(weights[k]) for k in _GAP_KEYS})
    F, L, D = weights["fc_w"].shape[1], weights["fc_w"].shape[0], weights["a_w"].shape[0]
    n = _lib.lib().amds_gated_attn_packed_floats(F, L, D) if pack else 0
    if n:
        big = [weights[k] for k in ("fc_w", "a_w", "b_w")]
        key = tuple(t.data_ptr() for t in big)
        ver = tuple(t._version for t in big)
        hit = _gap_packed.get(key)
        if hit is None or hit[0] != ver:
            pk = torch.empty(n, dtype=torch.float32, device=big[0].device)
            _lib.check(_lib.lib().amds_gated_attn_pack(C.byref(gw), _p(pk), F, L, D, _stream()), "gated_attn_pack")
            if len(_gap_packed) > 16:
                _gap_packed.clear()
            hit = _gap_packed[key] = (ver, pk, big)          # `big` keeps the tensors alive: a freed pointer cannot be mistaken for these weights
        gw.packed = _p(hit[1])
    return gw


GAP_MODES = {"auto": 0, "slab": 1, "split": 2}      # include/amdstamp.h AMDS_GAP_*


def gated_attn_pool(x: torch.Tensor, weights: dict[str, torch.Tensor], return_attn: bool = False, fused: bool = True, mode: str = "auto"):
    """x fp32 [N,F]; weights: fc_w/fc_b/a_w/a_b/b_w/b_b/c_w/c_b fp32 device tensors -> out [F] (+ A_raw [N]).  One launch (csrc/gap_fused.hip) for CHIEF's
    shapes; fused=False asks for the six-launch form (csrc/gap.hip) -- the A/B of tools/gap_only.py; mode "slab" / "split" forces one decomposition of the
    fused launch (through the batched entry with one bag)."""
    _dev(x, *weights.values())
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    if mode != "auto":
        assert fused
        r = gated_attn_pool_batched(x, [x.shape[0]], weights, return_attn=return_attn, mode=mode, offsets=False)
        return (r[0][0], r[1]) if return_attn else r[0]
    N, F = x.shape
    L, D = weights["fc_w"].shape[0], weights["a_w"].shape[0]
    lib = _lib.lib()
    sz, fn = ((lib.amds_gated_attn_pool_workspace_bytes, lib.amds_gated_attn_pool) if fused else
              (lib.amds_gated_attn_pool_unfused_workspace_bytes, lib.amds_gated_attn_pool_unfused))
    nbytes = sz(N, F, L, D)
    ws = _gap_workspace(x.device, nbytes)
    gw = _gap_weights(weights, pack=fused)
    out = torch.empty(F, dtype=torch.float32, device=x.device)
    araw = torch.empty(N, dtype=torch.float32, device=x.device) if return_attn else None
    _lib.check(fn(_p(x), C.byref(gw), _p(out), _p(araw), N, F, L, D, _p(ws), nbytes, _stream()), "gated_attn_pool")
    return (out, araw) if return_attn else out


def gated_attn_pool_batched_supported(F: int, L: int, D: int) -> bool:
    return bool(_lib.lib().amds_gated_attn_pool_batched_supported(F, L, D))


def gated_attn_pool_batched(x: torch.Tensor, lengths, weights: dict[str, torch.Tensor], return_attn: bool = False, offsets=None, mode: str = "auto"):
    """MANY bags in one launch.  x fp32 [sum(lengths), F] = the bags' rows concatenated (a [B, N, F] batch is the special case lengths = [N] * B);
    lengths: host ints, all > 0 (the reference raises on an empty feature matrix); offsets: the matching int64 device tensor [B + 1] if the caller
    already holds it (False: a single bag, no table).  -> out [B, F] (+ A_raw [sum(lengths)]).  mode: "auto" picks the decomposition by the total row count
    (include/amdstamp.h); with "slab" / "split" forced, each bag's result is bit-for-bit the one it gets alone."""
    _dev(x, *weights.values())
    lengths = [int(n) for n in lengths]
    if not lengths or min(lengths) <= 0:
        raise ValueError(f"every bag needs at least one row, got lengths {lengths[:8]}{'...' if len(lengths) > 8 else ''}")
    if x.dim() == 3:
        assert len(lengths) == x.shape[0] and all(n == x.shape[1] for n in lengths)
        x = x.reshape(-1, x.shape[-1])
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    total, F = x.shape
    assert total == sum(lengths), f"{total} rows for lengths summing to {sum(lengths)}"
    B = len(lengths)
    L, D = weights["fc_w"].shape[0], weights["a_w"].shape[0]
    lib = _lib.lib()
    if offsets is False:
        assert B == 1
        offsets = None
    else:
        if offsets is None:
            offsets = torch.tensor([0] + lengths, dtype=torch.int64).cumsum(0).to(x.device)
        assert offsets.dtype == torch.int64 and offsets.numel() == B + 1 and offsets.is_contiguous()
        _dev(offsets)
    nbytes = lib.amds_gated_attn_pool_batched_workspace_bytes(total, B, F, L, D)
    ws = _gap_workspace(x.device, nbytes)
    gw = _gap_weights(weights)
    out = torch.empty(B, F, dtype=torch.float32, device=x.device)
    araw = torch.empty(total, dtype=torch.float32, device=x.device) if return_attn else None
    _lib.check(lib.amds_gated_attn_pool_batched(_p(x), _p(offsets), B, total, C.byref(gw), _p(out), _p(araw), F, L, D, GAP_MODES[mode], _p(ws), nbytes,
                                                _stream()), "gated_attn_pool_batched")
    return (out, araw) if return_attn else out


def topk_rows_mean(score: torch.Tensor, rows: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]:
    """(indices int32 [k] of the k largest scores, descending; mean fp32 [cols] of rows[indices]) -- EAGLE's selection (eagle.py:106-118)."""
    _dev(score, rows)
    assert score.dtype == torch.float32 and score.is_contiguous() and score.dim() == 1
    assert rows.dim() == 2 and rows.stride(1) == 1 and rows.shape[0] == score.numel() and rows.dtype in (torch.float32, torch.float16)
    idx = torch.empty(k, dtype=torch.int32, device=score.device)
    mean = torch.empty(rows.shape[1], dtype=torch.float32, device=score.device)
    _lib.check(_lib.lib().amds_topk_rows_mean(_p(score), score.numel(), k, _p(rows), rows.stride(0), rows.shape[1], _DT[rows.dtype], _p(idx), _p(mean),
                                              _stream()), "topk_rows_mean")
    return idx, mean


def gather_rows(src: torch.Tensor, idx: torch.Tensor, n_out: int, out_dtype: torch.dtype) -> torch.Tensor:
    """dst[i] = src[idx[i]] (cast to out_dtype) for i < len(idx), zero rows up to n_out."""
    _dev(src, idx)
    assert src.dim() == 2 and src.is_contiguous() and idx.dtype == torch.int64
    dst = torch.empty(n_out, src.shape[1], dtype=out_dtype, device=src.device)
    _lib.check(_lib.lib().amds_gather_rows(_p(src), src.shape[1], _p(idx), idx.numel(), _p(dst), src.shape[1], n_out,
                                           src.shape[1], _DT[src.dtype], _DT[out_dtype], _stream()), "gather_rows")
    return dst


def vary_precision(data: torch.Tensor, shifts: torch.Tensor) -> torch.Tensor:
    """data: f32 / f16 / bf16 tensor; shifts: uint8, same shape = number of low mantissa bits to clear per element."""
    _dev(data, shifts)
    assert data.is_contiguous() and shifts.is_contiguous() and shifts.dtype == torch.uint8 and shifts.shape == data.shape
    out = torch.empty_like(data)
    _lib.check(_lib.lib().amds_vary_precision(_p(data), _p(shifts), _p(out), data.numel(), data.element_size(), _stream()),
               "vary_precision")
    return out


def mean_pool(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    assert x.dim() == 3 and x.is_contiguous()
    B, T, F = x.shape
    out = torch.empty(B, F, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_mean_pool(_p(x), _p(out), B, T, F, _DT[x.dtype], _stream()), "mean_pool")
    return out


def linear_f32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None, relu: bool = False) -> torch.Tensor:
    _dev(x, w, b)
    assert x.dtype == torch.float32 and w.dtype == torch.float32 and x.is_contiguous() and w.is_contiguous()
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_linear_f32(_p(x), _p(w), _p(b), _p(out), M, N, K, 1 if relu else 0, _stream()), "linear_f32")
    return out


# ---- CTransPath / Swin building blocks (include/amdstamp.h, "CTransPath tile encoder") --------------------------
def swin_stem(tiles: torch.Tensor, params: torch.Tensor, embed: int = 96, eps: float = 1e-5) -> torch.Tensor:
    """u8 [B,S,S,3] -> fp32 tokens [B,(S/4)^2,embed] (ConvStem + patch LayerNorm)."""
    _dev(tiles, params)
    assert tiles.dtype == torch.uint8 and tiles.is_contiguous() and tiles.shape[-1] == 3
    B, S = tiles.shape[0], tiles.shape[1]
    x = torch.empty(B, (S // 4) ** 2, embed, dtype=torch.float32, device=tiles.device)
    _lib.check(_lib.lib().amds_swin_stem(_p(tiles), _p(x), _p(params), B, S, embed, eps, _stream()), "swin_stem")
    return x


def window_attention(qkv: torch.Tensor, bias_lane: torch.Tensor, mask_bits: torch.Tensor, B: int, grid: int, heads: int,
                     shift: int) -> torch.Tensor:
    """qkv [B*grid^2, 3*heads*32] (raster token order) -> [B*grid^2, heads*32]."""
    _dev(qkv, bias_lane, mask_bits)
    assert mask_bits.dtype == torch.int64 and mask_bits.shape == (4, 64)
    dim = heads * 32
    assert qkv.is_contiguous() and qkv.shape == (B * grid * grid, 3 * dim)
    out = torch.empty(B * grid * grid, dim, dtype=qkv.dtype, device=qkv.device)
    _lib.check(_lib.lib().amds_window_attention(_p(qkv), 3 * dim, _p(out), dim, _p(bias_lane), _p(mask_bits), B, grid, dim,
                                                heads, shift, act_code(qkv.dtype), _stream()), "window_attention")
    return out


def patch_merge_ln(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, grid: int, eps: float = 1e-5,
                   out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """x fp32 [B, grid^2, C] -> LayerNorm(2x2 concat) [B, (grid/2)^2, 4C]."""
    _dev(x, gamma, beta)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, L, Cd = x.shape
    assert L == grid * grid
    y = torch.empty(B, L // 4, 4 * Cd, dtype=out_dtype, device=x.device)
    _lib.check(_lib.lib().amds_patch_merge_ln(_p(x), _p(y), _p(gamma), _p(beta), B, grid, Cd, eps, act_code(out_dtype),
                                              _stream()), "patch_merge_ln")
    return y


def layernorm_meanpool(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> tuple[torch.Tensor, torch.Tensor]:
    """x fp32 [B,L,C] -> (fp16 [B,C], fp32 [B,C]) = mean over tokens of LayerNorm(x)."""
    _dev(x, gamma, beta)
    assert x.dtype == torch.float32 and x.is_contiguous()
    B, L, Cd = x.shape
    o16 = torch.empty(B, Cd, dtype=torch.float16, device=x.device)
    o32 = torch.empty(B, Cd, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_layernorm_meanpool(_p(x), _p(o16), _p(o32), _p(gamma), _p(beta), B, L, Cd, eps, _stream()),
               "layernorm_meanpool")
    return o16, o32


def gemm_rowstream(a: torch.Tensor, w: torch.Tensor, epi: int, *, bias=None, ln_gamma=None, ln_beta=None, eps: float = 1e-5,
                   out=None) -> torch.Tensor:
    """Weights-stationary narrow GEMM; `a` is fp32 (with ln_gamma/ln_beta: LayerNorm fused) or the dtype of `w`."""
    _dev(a, w, bias, ln_gamma, ln_beta, out)
    M, K = a.shape
    N = w.shape[0]
    f32_out = epi in (_lib.EPI_RESIDUAL, _lib.EPI_BIAS_F32)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32 if f32_out else w.dtype, device=a.device)
    _lib.check(_lib.lib().amds_gemm_rowstream(_p(a), a.stride(0), _p(ln_gamma), _p(ln_beta), eps, _p(w), w.stride(0), M, N, K,
                                              act_code(w.dtype), epi, _p(out), out.stride(0), _p(bias), _stream()), "gemm_rowstream")
    return out


def swin_mlp96(x: torch.Tensor, fc1_w: torch.Tensor, fc1_b: torch.Tensor, fc2_w: torch.Tensor, fc2_b: torch.Tensor,
               ln_gamma: torch.Tensor, ln_beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """In place: x += fc2(gelu(fc1(LayerNorm(x)))) for x fp32 [M, 96]."""
    _dev(x, fc1_w, fc1_b, fc2_w, fc2_b, ln_gamma, ln_beta)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == 96
    assert fc1_w.shape == (384, 96) and fc2_w.shape == (96, 384) and fc1_w.is_contiguous() and fc2_w.is_contiguous()
    M = x.numel() // 96
    _lib.check(_lib.lib().amds_swin_mlp96(_p(x), M, _p(fc1_w), _p(fc1_b), _p(fc2_w), _p(fc2_b), _p(ln_gamma), _p(ln_beta), eps,
                                          act_code(fc1_w.dtype), _stream()), "swin_mlp96")
    return x


def tile_edge_fraction(tiles: torch.Tensor, low: int = 40, high: int = 100, return_maps: bool = False):
    """u8 [B,S,S,3] -> fp32 [B] = mean(Canny(grey(tile), low, high)) / 255 (reference tiling.py:280-291)."""
    _dev(tiles)
    assert tiles.dtype == torch.uint8 and tiles.is_contiguous() and tiles.dim() == 4 and tiles.shape[-1] == 3 and tiles.shape[1] == tiles.shape[2]
    B, S = tiles.shape[0], tiles.shape[1]
    frac = torch.empty(B, dtype=torch.float32, device=tiles.device)
    edges = torch.empty(B, S, S, dtype=torch.uint8, device=tiles.device) if return_maps else None
    gray = torch.empty(B, S, S, dtype=torch.uint8, device=tiles.device) if return_maps else None
    _lib.check(_lib.lib().amds_tile_edge_fraction_u8(_p(tiles), _p(frac), _p(edges), _p(gray), B, S, low, high, _stream()), "tile_edge_fraction")
    return (frac, edges, gray) if return_maps else frac


def swin_attn96(x: torch.Tensor, qkv_w, qkv_b, proj_w, proj_b, ln_gamma, ln_beta, bias_lane, mask_bits, grid: int, shift: int,
                eps: float = 1e-5) -> torch.Tensor:
    """In place: x += proj(window_attention(qkv(LayerNorm(x)))) for x fp32 [B, grid^2, 96]."""
    _dev(x, qkv_w, qkv_b, proj_w, proj_b, ln_gamma, ln_beta, bias_lane, mask_bits)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == 96 and x.shape[1] == grid * grid
    assert qkv_w.shape == (288, 96) and proj_w.shape == (96, 96) and qkv_w.is_contiguous() and proj_w.is_contiguous()
    _lib.check(_lib.lib().amds_swin_attn96(_p(x), _p(qkv_w), _p(qkv_b), _p(proj_w), _p(proj_b), _p(ln_gamma), _p(ln_beta), _p(bias_lane),
                                           _p(mask_bits), x.shape[0], grid, shift, eps, act_code(qkv_w.dtype), _stream()), "swin_attn96")
    return x


def swin_mlp192_pack(fc1_w: torch.Tensor, fc2_w: torch.Tensor) -> torch.Tensor:
    """[768,192] and [192,768] (act dtype) -> the fragment-ordered weight image amds_swin_mlp192 streams through LDS."""
    _dev(fc1_w, fc2_w)
    assert fc1_w.shape == (768, 192) and fc2_w.shape == (192, 768) and fc1_w.is_contiguous() and fc2_w.is_contiguous() and fc1_w.dtype == fc2_w.dtype
    out = torch.empty(2 * 768 * 192, dtype=fc1_w.dtype, device=fc1_w.device)
    _lib.check(_lib.lib().amds_swin_mlp192_pack(_p(fc1_w), _p(fc2_w), _p(out), act_code(fc1_w.dtype), _stream()), "swin_mlp192_pack")
    return out


def swin_mlp192(x: torch.Tensor, packed_w: torch.Tensor, fc1_b, fc2_b, ln_gamma, ln_beta, eps: float = 1e-5) -> torch.Tensor:
    """In place: x += fc2(gelu(fc1(LayerNorm(x)))) for x fp32 [M, 192]."""
    _dev(x, packed_w, fc1_b, fc2_b, ln_gamma, ln_beta)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == 192 and packed_w.numel() == 2 * 768 * 192
    _lib.check(_lib.lib().amds_swin_mlp192(_p(x), x.numel() // 192, _p(packed_w), _p(fc1_b), _p(fc2_b), _p(ln_gamma), _p(ln_beta), eps,
                                           act_code(packed_w.dtype), _stream()), "swin_mlp192")
    return x
