"""torch-tensor wrappers over the training entry points of the C ABI (see include/amdstamp.h, "MIL training step")."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _DT, _dev, _p, _stream, act_code


def transpose16(src: torch.Tensor, ld_dst: int | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """[R, C] 16-bit -> [C, ld_dst] (columns R..ld_dst of a fresh output are zero)."""
    _dev(src)
    assert src.dim() == 2 and src.element_size() == 2 and src.stride(1) == 1
    R, Cc = src.shape
    ld = ld_dst or R
    if out is None:
        out = torch.zeros(Cc, ld, dtype=src.dtype, device=src.device)
    _lib.check(_lib.lib().amds_transpose16(_p(src), src.stride(0), _p(out), out.stride(0), R, Cc, _stream()), "transpose16")
    return out


def cast_transpose_multi(entries: list) -> None:
    """entries: (src fp32 [R, C], dst 16-bit [R, >= C], dst_t 16-bit [C, >= R] or None): every 16-bit operand copy (and transposed copy) of a model's
    weight matrices refreshed from the fp32 masters in ONE launch per 32 matrices (`amds_cast_transpose_multi`); R, C multiples of 64."""
    import ctypes as C
    for i in range(0, len(entries), 32):
        chunk = entries[i:i + 32]
        arr = (_lib.CastEntry * len(chunk))()
        for j, (src, dst, dst_t) in enumerate(chunk):
            _dev(src, dst, dst_t)
            assert src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1 and dst.shape[0] == src.shape[0] and dst.stride(1) == 1
            assert dst_t is None or (dst_t.dtype == dst.dtype and dst_t.shape[0] == src.shape[1] and dst_t.stride(1) == 1)
            arr[j] = _lib.CastEntry(_p(src), src.stride(0), _p(dst), dst.stride(0), _p(dst_t), dst_t.stride(0) if dst_t is not None else 0, src.shape[0], src.shape[1],
                                    act_code(dst.dtype))
        _lib.check(_lib.lib().amds_cast_transpose_multi(arr, len(chunk), _stream()), "cast_transpose_multi")


def colsum_multi(xs: list[torch.Tensor]) -> list[torch.Tensor]:
    """Column sums of several matrices (fp32 with at most 2048 rows summed directly; anything else through its chunk partials) finished by ONE launch per 32
    matrices (`amds_colsum_multi`): the bits of `colsum` on each, which is how the MIL `vit` backward finishes its LayerNorm / bias / class-token gradients."""
    lib = _lib.lib()
    outs, keep = [], []
    entries = []
    for x in xs:
        _dev(x)
        assert x.dim() == 2 and x.stride(1) == 1
        M, N = x.shape
        out = torch.empty(N, dtype=torch.float32, device=x.device)
        outs.append(out)
        if M <= 2048 and x.dtype == torch.float32:
            entries.append(_lib.ColsumEntry(_p(x), _p(out), x.stride(0), M, N, 0))
        else:
            part = torch.empty(max(lib.amds_colsum_workspace_bytes(M, N), 4 * N), dtype=torch.uint8, device=x.device)
            keep.append(part)
            nchunk = _lib.C.c_int(0)
            _lib.check(lib.amds_colsum_partials(_p(x), x.stride(0), _p(part), M, N, _DT[x.dtype], _lib.C.byref(nchunk), _stream()), "colsum_partials")
            entries.append(_lib.ColsumEntry(_p(part), _p(out), N, nchunk.value, N, 1))
    for i in range(0, len(entries), 32):
        chunk = entries[i:i + 32]
        arr = (_lib.ColsumEntry * len(chunk))(*chunk)
        _lib.check(lib.amds_colsum_multi(arr, len(chunk), _stream()), "colsum_multi")
    return outs


def colsum(x: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    _dev(x)
    if x.dim() == 2 and x.shape[1] == 1 and x.is_contiguous():
        x = x.flatten().unsqueeze(1)        # (a single column: torch leaves the stride of a size-1 dimension arbitrary)
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    lib = _lib.lib()
    nb = lib.amds_colsum_workspace_bytes(M, N)
    ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=x.device)
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x.device)
    _lib.check(lib.amds_colsum(_p(x), x.stride(0), _p(out), M, N, _DT[x.dtype], 1 if accumulate else 0, _p(ws), nb, _stream()), "colsum")
    return out


def layernorm_train(x: torch.Tensor, gamma, beta, eps: float, out_dtype=torch.bfloat16, rows=None, row_stride=None, out=None, ld_out=None):
    """LayerNorm over the first gamma.numel() columns of rows pitched `row_stride`; `out` / `ld_out`: a wider, zero-padded destination."""
    _dev(x, gamma, beta, out)
    assert x.dtype == torch.float32 and x.is_contiguous()
    cols = gamma.numel()
    rows = rows if rows is not None else x.numel() // cols
    xs = row_stride if row_stride is not None else cols
    ld = ld_out if ld_out is not None else cols
    y = out if out is not None else torch.empty(rows, ld, dtype=out_dtype, device=x.device)
    assert y.dtype == out_dtype and y.shape == (rows, ld) and (out is not None or ld == cols)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().amds_layernorm_train(_p(x), xs, _p(gamma), _p(beta), _p(y), ld, _p(mean), _p(rstd), rows, cols, eps,
                                               _DT[out_dtype], _stream()), "layernorm_train")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, add_skip: bool, dgamma, dbeta, accumulate_params: bool = False,
                  rows=None, dy_stride=None, x_stride=None, dx_stride=None):
    _dev(dy, x, mean, rstd, gamma, dx, dgamma, dbeta)
    cols = gamma.numel()
    rows = rows if rows is not None else dy.numel() // cols
    lib = _lib.lib()
    nb = lib.amds_layernorm_bwd_workspace_bytes(rows, cols)
    ws = torch.empty(nb, dtype=torch.uint8, device=dy.device)
    _lib.check(lib.amds_layernorm_bwd(_p(dy), dy_stride or cols, _p(x), x_stride or cols, _p(mean), _p(rstd), _p(gamma), _p(dx),
                                      dx_stride or cols, 1 if add_skip else 0, _p(dgamma), _p(dbeta), 1 if accumulate_params else 0,
                                      rows, cols, _p(ws), nb, _stream()), "layernorm_bwd")
    return dx


def gelu_fwd(z: torch.Tensor, out_dtype=None) -> torch.Tensor:
    _dev(z)
    u = torch.empty(z.shape, dtype=out_dtype or z.dtype, device=z.device)
    _lib.check(_lib.lib().amds_gelu_fwd(_p(z), _p(u), z.numel(), _DT[z.dtype], _DT[u.dtype], _stream()), "gelu_fwd")
    return u


def gelu_bwd(z: torch.Tensor, du: torch.Tensor, out_dtype=None) -> torch.Tensor:
    _dev(z, du)
    assert z.is_contiguous() and du.is_contiguous() and z.shape == du.shape
    dz = torch.empty(z.shape, dtype=out_dtype or z.dtype, device=z.device)
    _lib.check(_lib.lib().amds_gelu_bwd(_p(z), _p(du), _p(dz), z.numel(), _DT[z.dtype], _DT[du.dtype], _DT[dz.dtype], _stream()), "gelu_bwd")
    return dz


def attention_fwd_lse(qkv: torch.Tensor, B: int, T: int, H: int):
    _dev(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * 64)
    out = torch.empty(B * T, H * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_fwd_lse(_p(qkv), _p(out), _p(lse), B, T, H, act_code(qkv.dtype), _stream()), "attention_fwd_lse")
    return out, lse


def attention_bwd(qkv, out, dout, lse, B: int, T: int, H: int) -> torch.Tensor:
    _dev(qkv, out, dout, lse)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dout.dtype == qkv.dtype == out.dtype
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(ws), _p(dqkv), B, T, H, act_code(qkv.dtype), _stream()),
               "attention_bwd")
    return dqkv


def attention_fwd_train(qkv: torch.Tensor, B: int, T: int, H: int, p: float = 0.0, seed: int = 0, stream_id: int = 0):
    """nn.MultiheadAttention forward in train mode: dropout p on the attention probabilities (counter-based mask), + lse."""
    _dev(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * 64)
    out = torch.empty(B * T, H * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_fwd_train(_p(qkv), _p(out), _p(lse), B, T, H, act_code(qkv.dtype), float(p), int(seed), int(stream_id),
                                                   _stream()), "attention_fwd_train")
    return out, lse


def attention_bwd_train(qkv, out, dout, lse, B: int, T: int, H: int, p: float = 0.0, seed: int = 0, stream_id: int = 0) -> torch.Tensor:
    _dev(qkv, out, dout, lse)
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dout.dtype == qkv.dtype == out.dtype
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_bwd_train(_p(qkv), _p(out), _p(dout), _p(lse), _p(ws), _p(dqkv), B, T, H, act_code(qkv.dtype), float(p),
                                                   int(seed), int(stream_id), _stream()), "attention_bwd_train")
    return dqkv


def dropout_mask(n: int, p: float, seed: int, stream_id: int, device) -> torch.Tensor:
    """The keep mask (u8) an elementwise dropout site draws for (seed, stream_id) -- for tests."""
    m = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().amds_dropout_mask(_p(m), n, float(p), int(seed), int(stream_id), _stream()), "dropout_mask")
    return m


def attention_dropout_mask(B: int, H: int, T: int, p: float, seed: int, stream_id: int, device) -> torch.Tensor:
    m = torch.empty(B, H, T, T, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().amds_attention_dropout_mask(_p(m), B, H, T, float(p), int(seed), int(stream_id), _stream()), "attention_dropout_mask")
    return m


def gemm_batched(a, lda, bsA, w, ldw, bsW, M, N, K, nbatch, dtype, out, ldo, bsOut, f32_out: bool, bias=None, acc_scale=1.0):
    _lib.check(_lib.lib().amds_gemm_batched(_p(a), lda, bsA, _p(w), ldw, bsW, M, N, K, nbatch, act_code(dtype),
                                            _lib.EPI_BIAS_F32 if f32_out else _lib.EPI_BIAS, _p(out), ldo, bsOut, _p(bias), acc_scale,
                                            _stream()), "gemm_batched")
    return out


def adamw(p, g, m, v, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01) -> None:
    _dev(p, g, m, v)
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (p, g, m, v)) and p.numel() == g.numel() == m.numel() == v.numel()
    _lib.check(_lib.lib().amds_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, _stream()), "adamw")


def attention_alibi_fwd_train(qkv: torch.Tensor, coords: torch.Tensor, inv_running_mean: torch.Tensor, bias_scale: torch.Tensor,
                              B: int, T: int, H: int):
    """ALiBi attention forward that saves what its backward needs: returns (out, U, Osm: bf16; lse fp32 [B,H,T])."""
    _dev(qkv, coords, inv_running_mean, bias_scale)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * H * 64) and coords.shape == (B, T, 2) and coords.dtype == torch.float32
    out = torch.empty(B * T, H * 64, dtype=torch.bfloat16, device=qkv.device)
    u, osm = torch.empty_like(out), torch.empty_like(out)
    lse = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_alibi_fwd_train(_p(qkv), _p(coords.contiguous()), _p(inv_running_mean), _p(bias_scale), _p(out), _p(u),
                                                         _p(osm), _p(lse), B, T, H, act_code(qkv.dtype), _stream()), "attention_alibi_fwd_train")
    return out, u, osm, lse


def attention_alibi_bwd(qkv, osm, u, dout, lse, coords, bias_scale, dist_scale, B: int, T: int, H: int):
    """-> (dqkv bf16, d bias_scale [H] fp32)."""
    _dev(qkv, osm, u, dout, lse, coords, bias_scale, dist_scale)
    assert qkv.dtype == osm.dtype == u.dtype == dout.dtype == torch.bfloat16 and all(t.is_contiguous() for t in (qkv, osm, u, dout))
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    part = torch.empty(B, H, T, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().amds_attention_alibi_bwd(_p(qkv), _p(osm), _p(u), _p(dout), _p(lse), _p(coords.contiguous()), _p(bias_scale), _p(dist_scale),
                                                   _p(ws), _p(part), _p(dqkv), B, T, H, _stream()), "attention_alibi_bwd")
    dbs = colsum(part.permute(1, 0, 2).reshape(H, B * T).t().contiguous())        # [B*T, H] -> column sums = per-head gradient
    return dqkv, dbs


def cdist_mean(coords: torch.Tensor) -> torch.Tensor:
    """mean of torch.cdist(coords, coords) over [B, T, T] as a 0-d device tensor (no host sync)."""
    _dev(coords)
    B, T, _ = coords.shape
    rows = torch.empty(B * T, 1, dtype=torch.float32, device=coords.device)
    _lib.check(_lib.lib().amds_cdist_rowsum(_p(coords.contiguous()), _p(rows), B, T, _stream()), "cdist_rowsum")
    return colsum(rows).reshape(()) / float(B * T * T)


def wgrad_tn(dy: torch.Tensor, x: torch.Tensor, split_k: int = 32) -> torch.Tensor:
    """dW [N, K] = dy^T x from TOKEN-major 16-bit operands dy [tokens, N], x [tokens, K] (rows may be views with a pitch): `amds_wgrad_tn` partials
    (no transposed copies; include/amdstamp.h) summed by the deterministic column sum."""
    _dev(dy, x)
    assert dy.dim() == 2 and x.dim() == 2 and dy.shape[0] == x.shape[0] and dy.dtype == x.dtype and dy.stride(1) == 1 and x.stride(1) == 1
    tokens, N = dy.shape
    K = x.shape[1]
    part = torch.empty(split_k, N, K, dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().amds_wgrad_tn(_p(dy), dy.stride(0), _p(x), x.stride(0), tokens, N, K, split_k, act_code(dy.dtype), _p(part), _stream()), "wgrad_tn")
    return colsum(part.view(split_k, N * K)).view(N, K)
