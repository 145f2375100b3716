"""TransMIL training on the HIP path: forward with saved intermediates and the hand-derived backward.

Reference: src/stamp/modeling/models/trans_mil.py -- TransMIL.forward :299-326, TransLayer :245-263, NystromAttention.forward
:81-163 (mask=None), moore_penrose_iter_pinv :23-37, PPEG :266-283 -- differentiated by autograd inside the reference's
LitTileClassifier._step (models/__init__.py:239-279).  Train mode adds ONE dropout site: `to_out = Sequential(Linear, Dropout(0.1))`
(:63-66 with dropout=0.1 from :255).  Everything is fp32 like the reference (the pseudo-inverse iteration is numerically touchy):
matrix products on the exact-fp32 MFMA (`amds_bgemm_f32`), the rest in the kernels of transmil.hip / train.hip / dropout.hip.
Transposes (`.transpose(-1, -2).contiguous()`), slicing and concatenation are data movement done through torch.

Backward of one Nystrom attention (per head; S1 = scale q kl^T, S2 = ql kl^T, S3 = ql k^T, a_i = softmax(S_i), z = pinv(a2),
out = (a1 z)(a3 v) + conv33(v)):
    d(a1 z) = do (a3 v)^T        d(a3 v) = (a1 z)^T do        da1 = d(a1 z) z^T        dz = a1^T d(a1 z)
    da3 = d(a3 v) v^T            dv = a3^T d(a3 v) + conv33^T(do)
    pinv, iteration k (A = a2 z_k, T1 = 7I - A, T2 = 15I - A T1, T3 = 13I - A T2, z_{k+1} = z_k T3 / 4), given g = dz_{k+1}:
        dz_k = g T3^T / 4 ; dT3 = z_k^T g / 4 ; dA = -dT3 T2^T ; dT2 = -A^T dT3 ; dA -= dT2 T1^T ; dT1 = -A^T dT2 ; dA -= dT1 (= + A^T dT2)
        da2 += dA z_k^T ; dz_k += a2^T dA          and z_0 = a2^T / (max row-sum * max col-sum) through amds_pinv_init_bwd
    dS_i = a_i o (da_i - rowsum(a_i o da_i))
    dq = scale dS1 kl + (scale / l) broadcast(dS2 kl + dS3 k) ; dk = dS3^T ql + (1 / l) broadcast(scale dS1^T q + dS2^T ql)
"""
from __future__ import annotations

import math

import torch

from . import _lib, ops
from . import train_ops as T

HEADS, ITERS, CONV_K, P_OUT = 8, 6, 33, 0.1


def _bg(A, lda, sAo, sAi, B, ldb, sBo, sBi, transb, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha=1.0, diag=0.0, bias=None, accumulate=False,
        transa=False):
    """transa: A is stored [K][M] with pitch lda and the product is A^T B (no explicit transpose: the kernel transposes while staging)."""
    p = lambda t: t if isinstance(t, int) else t.data_ptr()  # noqa: E731
    _lib.check(_lib.lib().amds_bgemm_f32(p(A), lda, sAo, sAi, p(B), ldb, sBo, sBi, (1 if transb else 0) | (2 if transa else 0), p(Cm), ldc, sCo, sCi, outer, inner,
                                         M, N, K, alpha, diag, None if bias is None else p(bias), 1 if accumulate else 0, ops._stream()), "bgemm_f32")


def _mm(A: torch.Tensor, B: torch.Tensor, transb: bool, out: torch.Tensor | None = None, alpha: float = 1.0, diag: float = 0.0, accumulate: bool = False,
        transa: bool = False):
    """Batched product of contiguous [Z, M, K] (or [Z, K, M] if transa) with [Z, K, N] (or [Z, N, K] if transb) -> [Z, M, N]."""
    if transa:
        Z, K, M = A.shape
    else:
        Z, M, K = A.shape
    N = B.shape[1] if transb else B.shape[2]
    if out is None:
        out = torch.empty(Z, M, N, dtype=torch.float32, device=A.device)
    zc = 32768
    for z0 in range(0, Z, zc):                        # gridDim.z limit of one launch
        z1 = min(Z, z0 + zc)
        _bg(A[z0:z1], A.shape[2], M * K, 0, B[z0:z1], B.shape[2], B.shape[1] * B.shape[2], 0, transb, out[z0:z1], N, M * N, 0, z1 - z0, 1, M, N, K,
            alpha=alpha, diag=diag, accumulate=accumulate, transa=transa)
    return out


def _tr(x: torch.Tensor) -> torch.Tensor:
    return x.transpose(-1, -2).contiguous()


def _wgrad(dy3: torch.Tensor, x3: torch.Tensor) -> torch.Tensor:
    """Weight gradient of a Linear over all rows of all bags, dW[M, N] = sum_b dy_b^T x_b for dy3 [b, n, M], x3 [b, n, N] (contiguous):
    one product per bag into fp32 partials, then a fixed-order sum over the bags.  As ONE product with K = b * n (65 k - 82 k) the output's
    few tiles left 220 of the 256 CUs idle for 3 - 11 ms per weight."""
    b, n, M = dy3.shape
    N = x3.shape[2]
    part = torch.empty(b, M, N, dtype=torch.float32, device=dy3.device)
    _bg(dy3, M, n * M, 0, x3, N, n * N, 0, False, part, N, M * N, 0, b, 1, M, N, n, transa=True)
    return T.colsum(part.view(b, M * N)).view(M, N)


def _f(t):
    return t.detach().float().contiguous()


class _Nys:
    """Parameters of one TransLayer as fp32 device tensors."""

    def __init__(self, get, name: str):
        self.norm_w, self.norm_b = get(f"{name}.norm.weight"), get(f"{name}.norm.bias")
        self.wqkv = get(f"{name}.attn.to_qkv.weight")
        self.wo, self.bo = get(f"{name}.attn.to_out.0.weight"), get(f"{name}.attn.to_out.0.bias")
        self.wconv = get(f"{name}.attn.res_conv.weight").reshape(HEADS, -1).contiguous()


STEPWISE = 0            # tests: 0 = the whole-step library calls, 1 = host loop around amds_nystrom_attn_fwd / _bwd, 2 = everything kernel by kernel from the host


def _layer_struct(P: _Nys) -> "_lib.TransMilLayer":
    return _lib.TransMilLayer(P.norm_w.data_ptr(), P.norm_b.data_ptr(), P.wqkv.data_ptr(), P.wo.data_ptr(), P.bo.data_ptr(), P.wconv.data_ptr())


def nystrom_forward(y: torch.Tensor, P: _Nys, x_res: torch.Tensor, p_drop: float, seed: int, sid: int):
    """x_res += Dropout(to_out(NystromAttention(y)))[:, -n:]; returns what the backward needs.  ONE library call (amds_nystrom_attn_fwd,
    csrc/nystrom_train.hip): the intermediates live in one arena the backward reads."""
    if STEPWISE >= 2:
        return nystrom_forward_stepwise(y, P, x_res, p_drop, seed, sid)
    import ctypes as C
    b, n, Cd = y.shape
    assert y.is_contiguous() and x_res.is_contiguous() and y.dtype == torch.float32 and x_res.dtype == torch.float32
    lib = _lib.lib()
    need = lib.amds_nystrom_attn_saved_bytes(Cd, b, n)
    if need == 0:
        _lib.check(-1, "nystrom_attn_saved_bytes")
    arena = torch.empty(need, dtype=torch.uint8, device=y.device)
    L = _layer_struct(P)
    _lib.check(lib.amds_nystrom_attn_fwd(C.byref(L), Cd, y.data_ptr(), x_res.data_ptr(), b, n, float(p_drop), int(seed) & (2 ** 64 - 1), int(sid),
                                         arena.data_ptr(), arena.numel(), ops._stream()), "nystrom_attn_fwd")
    return dict(arena=arena, n=n, p_drop=p_drop, seed=seed, sid=sid)


def nystrom_backward(S: dict, P: _Nys, dx: torch.Tensor, need_params: bool = True):
    """dx: gradient of the residual stream after the block [b, n, C] -> (dy [b, n, C] gradient w.r.t. the LayerNorm output, grads).
    ONE library call (amds_nystrom_attn_bwd)."""
    if "arena" not in S:
        return nystrom_backward_stepwise(S, P, dx, need_params)
    import ctypes as C
    b, n, Cd = dx.shape
    dev = dx.device
    dx = dx.contiguous()
    lib = _lib.lib()
    need = lib.amds_nystrom_attn_workspace_bytes(Cd, b, n)
    if need == 0:
        _lib.check(-1, "nystrom_attn_workspace_bytes")
    ws = ops.scratch("nystrom", dev, need)
    dy = torch.empty(b, n, Cd, dtype=torch.float32, device=dev)
    G, gc = {}, None
    if need_params:
        f32 = dict(dtype=torch.float32, device=dev)
        G = {"attn.to_qkv.weight": torch.empty(3 * Cd, Cd, **f32), "attn.to_out.0.weight": torch.empty(Cd, Cd, **f32), "attn.to_out.0.bias": torch.empty(Cd, **f32),
             "attn.res_conv.weight": torch.empty(HEADS, 1, CONV_K, 1, **f32)}
        gc = _lib.NystromGrads(G["attn.to_qkv.weight"].data_ptr(), G["attn.to_out.0.weight"].data_ptr(), G["attn.to_out.0.bias"].data_ptr(),
                               G["attn.res_conv.weight"].data_ptr())
    L = _layer_struct(P)
    arena = S["arena"]
    _lib.check(lib.amds_nystrom_attn_bwd(C.byref(L), Cd, dx.data_ptr(), dy.data_ptr(), C.byref(gc) if gc is not None else None, b, n, float(S["p_drop"]),
                                         int(S["seed"]) & (2 ** 64 - 1), int(S["sid"]), arena.data_ptr(), arena.numel(), ws.data_ptr(), ws.numel(), ops._stream()),
               "nystrom_attn_bwd")
    return dy, G


def nystrom_forward_stepwise(y: torch.Tensor, P: _Nys, x_res: torch.Tensor, p_drop: float, seed: int, sid: int):
    """The same forward, one library call per kernel from the host (what `nystrom_forward` did before amds_nystrom_attn_fwd existed): kept as
    the cross-check of the C entry points in tests/ -- results are bit-identical."""
    b, n, Cd = y.shape
    H, m = HEADS, Cd // 2
    d = Cd // H
    rem = n % m
    pad = (m - rem) if rem > 0 else 0
    yp = torch.nn.functional.pad(y, (0, 0, pad, 0), value=0.0) if pad else y          # FRONT padding (:100)
    np_ = n + pad
    dev = y.device
    f32 = dict(dtype=torch.float32, device=dev)
    lib, st = _lib.lib(), ops._stream()
    qkv = torch.empty(b, np_, 3 * Cd, **f32)
    _bg(yp, Cd, 0, 0, P.wqkv, Cd, 0, 0, True, qkv, 3 * Cd, 0, 0, 1, 1, b * np_, 3 * Cd, Cd)
    e4 = 4
    qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + Cd * e4, qkv.data_ptr() + 2 * Cd * e4
    sb, sh, ld = np_ * 3 * Cd, d, 3 * Cd
    scale = d ** -0.5
    l = math.ceil(n / m)
    ql, kl = torch.empty(b, H, m, d, **f32), torch.empty(b, H, m, d, **f32)
    _lib.check(lib.amds_landmark_mean(qp, sb, sh, ld, ql.data_ptr(), b, H, m, l, d, scale / l, st), "landmark_mean")
    _lib.check(lib.amds_landmark_mean(kp, sb, sh, ld, kl.data_ptr(), b, H, m, l, d, 1.0 / l, st), "landmark_mean")
    a1, a2, a3 = torch.empty(b, H, np_, m, **f32), torch.empty(b, H, m, m, **f32), torch.empty(b, H, m, np_, **f32)
    _bg(qp, ld, sb, sh, kl, d, H * m * d, m * d, True, a1, m, H * np_ * m, np_ * m, b, H, np_, m, d, alpha=scale)
    _bg(ql, d, H * m * d, m * d, kl, d, H * m * d, m * d, True, a2, m, H * m * m, m * m, b, H, m, m, d)
    _bg(ql, d, H * m * d, m * d, kp, ld, sb, sh, True, a3, np_, H * m * np_, m * np_, b, H, m, np_, d)
    for t, cols in ((a1, m), (a2, m), (a3, np_)):
        _lib.check(lib.amds_softmax_rows(t.data_ptr(), t.numel() // cols, cols, st), "softmax_rows")
    Z = b * H
    x2 = a2.view(Z, m, m)
    z = torch.empty(Z, m, m, **f32)
    scratch = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(lib.amds_pinv_init(x2.data_ptr(), z.data_ptr(), Z, m, scratch.data_ptr(), st), "pinv_init")
    its = []
    for _ in range(ITERS):                                  # (:29-35)
        A = _mm(x2, z, False)
        T1 = _mm(x2, z, False, alpha=-1.0, diag=7.0)
        T2 = _mm(A, T1, False, alpha=-1.0, diag=15.0)
        T3 = _mm(A, T2, False, alpha=-1.0, diag=13.0)
        znew = _mm(z, T3, False, alpha=0.25)
        its.append((z, A, T1, T2, T3))
        z = znew
    av = torch.empty(b, H, m, d, **f32)                                                        # attn3 @ v
    _bg(a3, np_, H * m * np_, m * np_, vp, ld, sb, sh, False, av, d, H * m * d, m * d, b, H, m, d, np_)
    a1z = _mm(a1.view(Z, np_, m), z, False)                                                     # attn1 @ pinv
    merged = torch.empty(b, np_, Cd, **f32)                                                    # heads merged: [b, n, (h d)]
    _bg(a1z, m, H * np_ * m, np_ * m, av, d, H * m * d, m * d, False, merged, Cd, np_ * Cd, d, b, H, np_, d, m)
    _lib.check(lib.amds_dwconv_seq(vp, sb, sh, ld, P.wconv.data_ptr(), merged.data_ptr(), np_ * Cd, d, Cd, b, H, np_, d, P.wconv.shape[1], st), "dwconv_seq")
    tail = merged[:, pad:, :]                                                                   # the last n rows of every bag (:155)
    if p_drop > 0.0:
        out = torch.empty(b, n, Cd, **f32)
        _bg(tail, Cd, np_ * Cd, 0, P.wo, Cd, 0, 0, True, out, Cd, n * Cd, 0, b, 1, n, Cd, Cd, bias=P.bo)
        _lib.check(lib.amds_dropout_add(out.data_ptr(), Cd, x_res.data_ptr(), Cd, x_res.data_ptr(), Cd, b * n, Cd, p_drop, seed, sid, st), "dropout_add")
    else:
        _bg(tail, Cd, np_ * Cd, 0, P.wo, Cd, 0, 0, True, x_res, Cd, n * Cd, 0, b, 1, n, Cd, Cd, bias=P.bo, accumulate=True)
    return dict(yp=yp, qkv=qkv, ql=ql, kl=kl, a1=a1, a2=a2, a3=a3, its=its, z=z, av=av, a1z=a1z, merged=merged, pad=pad, n=n, l=l, p_drop=p_drop,
                seed=seed, sid=sid)


def nystrom_backward_stepwise(S: dict, P: _Nys, dx: torch.Tensor, need_params: bool = True):
    """Backward of `nystrom_forward_stepwise` (its saved dict), kernel by kernel from the host; same results as `nystrom_backward`."""
    b, n, Cd = dx.shape
    H, m = HEADS, Cd // 2
    d = Cd // H
    pad, l = S["pad"], S["l"]
    np_ = n + pad
    dev = dx.device
    f32 = dict(dtype=torch.float32, device=dev)
    lib, st = _lib.lib(), ops._stream()
    Z = b * H
    scale = d ** -0.5
    qkv, merged = S["qkv"], S["merged"]
    e4 = 4
    sb, sh, ld = np_ * 3 * Cd, d, 3 * Cd
    qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + Cd * e4, qkv.data_ptr() + 2 * Cd * e4
    G = {}
    # to_out (+ Dropout(0.1))
    if S["p_drop"] > 0.0:
        dout = torch.empty(b * n, Cd, **f32)
        _lib.check(lib.amds_dropout_cast_bwd(dx.data_ptr(), Cd, dout.data_ptr(), Cd, b * n, Cd, _lib.F32, S["p_drop"], S["seed"], S["sid"], st), "dropout_cast_bwd")
    else:
        dout = dx.reshape(b * n, Cd)
    tail = merged[:, pad:, :]
    if need_params:
        tail_c = tail.reshape(b * n, Cd) if pad == 0 else tail.contiguous().view(b * n, Cd)
        gwo = _wgrad(dout.view(b, n, Cd), tail_c.view(b, n, Cd))                                        # dWo = dout^T merged_tail
        G["attn.to_out.0.weight"], G["attn.to_out.0.bias"] = gwo, T.colsum(dout)
    dmerged = torch.zeros(b, np_, Cd, **f32)
    _bg(dout, Cd, n * Cd, 0, P.wo, Cd, 0, 0, False, dmerged.view(-1)[pad * Cd:], Cd, np_ * Cd, 0, b, 1, n, Cd, Cd)   # dmerged_tail = dout Wo
    dqkv = torch.zeros(b, np_, 3 * Cd, **f32)
    dqp, dkp, dvp = dqkv.data_ptr(), dqkv.data_ptr() + Cd * e4, dqkv.data_ptr() + 2 * Cd * e4
    # 33-tap residual conv on v: data gradient = the same conv with reversed taps; weight gradient = a reduction
    wflip = P.wconv.flip(1).contiguous()
    _lib.check(lib.amds_dwconv_seq(dmerged.data_ptr(), np_ * Cd, d, Cd, wflip.data_ptr(), dvp, sb, sh, ld, b, H, np_, d, wflip.shape[1], st), "dwconv_seq(bwd)")
    if need_params:
        gconv = torch.empty(H, CONV_K, **f32)
        nb = lib.amds_dwconv_seq_wgrad_workspace_bytes(b, H, CONV_K)
        cws = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.amds_dwconv_seq_wgrad(dmerged.data_ptr(), np_ * Cd, d, Cd, vp, sb, sh, ld, gconv.data_ptr(), b, H, np_, d, CONV_K, cws.data_ptr(), nb, st),
                   "dwconv_seq_wgrad")
        G["attn.res_conv.weight"] = gconv.view(H, 1, CONV_K, 1)
    # out_h = a1z av  (do = head slice of dmerged, [np, d] at row pitch Cd)
    da1z = torch.empty(b, H, np_, m, **f32)
    _bg(dmerged, Cd, np_ * Cd, d, S["av"], d, H * m * d, m * d, True, da1z, m, H * np_ * m, np_ * m, b, H, np_, m, d)             # do av^T
    dav = torch.empty(b, H, m, d, **f32)
    _bg(S["a1z"], m, H * np_ * m, np_ * m, dmerged, Cd, np_ * Cd, d, False, dav, d, H * m * d, m * d, b, H, m, d, np_, transa=True)   # a1z^T do
    z = S["z"]
    da1 = _mm(da1z.view(Z, np_, m), z, True)                                                                                     # d(a1z) z^T
    dz = _mm(S["a1"].view(Z, np_, m), da1z.view(Z, np_, m), False, transa=True)                                                  # a1^T d(a1z)
    da3 = torch.empty(b, H, m, np_, **f32)
    _bg(dav, d, H * m * d, m * d, vp, ld, sb, sh, True, da3, np_, H * m * np_, m * np_, b, H, m, np_, d)                          # d(av) v^T
    _bg(S["a3"], np_, H * m * np_, m * np_, dav, d, H * m * d, m * d, False, dvp, ld, sb, sh, b, H, np_, d, m, accumulate=True, transa=True)   # dv += a3^T d(av)
    # pseudo-inverse iterations, last to first
    x2 = S["a2"].view(Z, m, m)
    da2 = torch.zeros(Z, m, m, **f32)
    for (zk, A, T1, T2, T3) in reversed(S["its"]):
        dzk = _mm(dz, T3, True, alpha=0.25)                       # g T3^T / 4
        dT3 = _mm(zk, dz, False, alpha=0.25, transa=True)         # z_k^T g / 4
        dA = _mm(dT3, T2, True, alpha=-1.0)                       # -dT3 T2^T
        dT2 = _mm(A, dT3, False, alpha=-1.0, transa=True)         # -A^T dT3
        _mm(dT2, T1, True, out=dA, alpha=-1.0, accumulate=True)   # dA -= dT2 T1^T
        _mm(A, dT2, False, out=dA, accumulate=True, transa=True)  # dA -= dT1, dT1 = -A^T dT2
        _mm(dA, zk, True, out=da2, accumulate=True)               # da2 += dA z_k^T
        _mm(x2, dA, False, out=dzk, accumulate=True, transa=True)  # dz_k += a2^T dA
        dz = dzk
    nb = lib.amds_pinv_init_bwd_workspace_bytes(Z)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(lib.amds_pinv_init_bwd(x2.data_ptr(), dz.data_ptr(), da2.data_ptr(), Z, m, ws.data_ptr(), nb, st), "pinv_init_bwd")
    # the three softmaxes (in place: da_i becomes dS_i)
    for p_, g_, cols in ((S["a1"], da1, m), (S["a2"], da2, m), (S["a3"], da3, np_)):
        _lib.check(lib.amds_softmax_rows_bwd(p_.data_ptr(), g_.data_ptr(), g_.numel() // cols, cols, st), "softmax_rows_bwd")
    dS1, dS2, dS3 = da1.view(b, H, np_, m), da2.view(b, H, m, m), da3
    ql, kl = S["ql"], S["kl"]
    # dq = scale dS1 kl  (+ landmark part) ; dkl = scale dS1^T q + dS2^T ql ; dql = dS2 kl + dS3 k ; dk = dS3^T ql (+ landmark part)
    _bg(dS1, m, H * np_ * m, np_ * m, kl, d, H * m * d, m * d, False, dqp, ld, sb, sh, b, H, np_, d, m, alpha=scale)
    dkl = torch.empty(b, H, m, d, **f32)
    _bg(dS1, m, H * np_ * m, np_ * m, qp, ld, sb, sh, False, dkl, d, H * m * d, m * d, b, H, m, d, np_, alpha=scale, transa=True)      # dS1^T q
    _bg(dS2, m, H * m * m, m * m, ql, d, H * m * d, m * d, False, dkl, d, H * m * d, m * d, b, H, m, d, m, accumulate=True, transa=True)   # + dS2^T q_l
    dql = torch.empty(b, H, m, d, **f32)
    _bg(dS2, m, H * m * m, m * m, kl, d, H * m * d, m * d, False, dql, d, H * m * d, m * d, b, H, m, d, m)
    _bg(dS3, np_, H * m * np_, m * np_, kp, ld, sb, sh, False, dql, d, H * m * d, m * d, b, H, m, d, np_, accumulate=True)
    _bg(dS3, np_, H * m * np_, m * np_, ql, d, H * m * d, m * d, False, dkp, ld, sb, sh, b, H, np_, d, m, transa=True)            # dS3^T q_l
    _lib.check(lib.amds_landmark_mean_bwd(dql.data_ptr(), dqp, sb, sh, ld, b, H, m, l, d, scale / l, 1, st), "landmark_mean_bwd")
    _lib.check(lib.amds_landmark_mean_bwd(dkl.data_ptr(), dkp, sb, sh, ld, b, H, m, l, d, 1.0 / l, 1, st), "landmark_mean_bwd")
    # to_qkv (no bias)
    if need_params:
        G["attn.to_qkv.weight"] = _wgrad(dqkv, S["yp"].reshape(b, np_, Cd))
    dyp = torch.empty(b, np_, Cd, **f32)
    _bg(dqkv, 3 * Cd, 0, 0, P.wqkv, Cd, 0, 0, False, dyp, Cd, 0, 0, 1, 1, b * np_, Cd, 3 * Cd)
    return dyp[:, pad:, :].contiguous(), G


def _c_weights(get, Cd: int):
    """amds_transmil_weights over the parameters `get` returns (fp32 device tensors), + the tensors it points to."""
    keep = {}

    def ptr(name, shape=None):
        t = get(name)
        if shape is not None:
            t = t.reshape(shape).contiguous()
        keep[name] = t
        return t.data_ptr()

    w = _lib.TransMilWeights()
    w.fc1_w, w.fc1_b, w.cls_token = ptr("_fc1.0.weight"), ptr("_fc1.0.bias"), ptr("cls_token", (Cd,))
    for i, nm in enumerate(("layer1", "layer2")):
        w.layer[i] = _lib.TransMilLayer(ptr(f"{nm}.norm.weight"), ptr(f"{nm}.norm.bias"), ptr(f"{nm}.attn.to_qkv.weight"), ptr(f"{nm}.attn.to_out.0.weight"),
                                        ptr(f"{nm}.attn.to_out.0.bias"), ptr(f"{nm}.attn.res_conv.weight", (HEADS, -1)))
    w.ppeg_w7, w.ppeg_b7 = ptr("pos_layer.proj.weight", (Cd, -1)), ptr("pos_layer.proj.bias")
    w.ppeg_w5, w.ppeg_b5 = ptr("pos_layer.proj1.weight", (Cd, -1)), ptr("pos_layer.proj1.bias")
    w.ppeg_w3, w.ppeg_b3 = ptr("pos_layer.proj2.weight", (Cd, -1)), ptr("pos_layer.proj2.bias")
    w.norm_w, w.norm_b, w.fc2_w, w.fc2_b = ptr("norm.weight"), ptr("norm.bias"), ptr("_fc2.weight"), ptr("_fc2.bias")
    return w, keep


def forward_train(get, bags: torch.Tensor, dims: tuple[int, int, int], *, training: bool, seed: int = 0):
    """-> (logits [Bb, C], saved).  dims = (dim_input, dim_hidden, dim_output).  ONE library call (amds_transmil_train_forward,
    csrc/transmil_train.hip); `saved` holds the activation arena the backward reads."""
    if STEPWISE >= 1:
        return forward_train_stepwise(get, bags, dims, training=training, seed=seed)
    import ctypes as C
    Fd, Cd, Cc = dims
    Bb, Tn, _ = bags.shape
    dev = bags.device
    if bags.dtype not in ops._DT:
        bags = bags.float()
    bags = bags.contiguous()
    cfg = _lib.TransMilCfg(Fd, Cd, Cc)
    w, keep = _c_weights(get, Cd)
    lib = _lib.lib()
    need = lib.amds_transmil_train_saved_bytes(C.byref(cfg), Bb, Tn)
    if need == 0:
        _lib.check(-1, "transmil_train_saved_bytes")
    arena = torch.empty(need, dtype=torch.uint8, device=dev)
    logits = torch.empty(Bb, Cc, dtype=torch.float32, device=dev)
    p_out = P_OUT if training else 0.0
    _lib.check(lib.amds_transmil_train_forward(C.byref(cfg), C.byref(w), bags.data_ptr(), ops._DT[bags.dtype], p_out, int(seed) & (2 ** 64 - 1), logits.data_ptr(), Bb, Tn,
                                               arena.data_ptr(), arena.numel(), ops._stream()), "transmil_train_forward")
    return logits, dict(arena=arena, cfg=cfg, w=w, keep=keep, shape=(Bb, Tn, Fd), dims=dims, p_out=p_out, seed=seed)


def backward(saved: dict, dlogits: torch.Tensor, *, need_params: bool = True, need_bags: bool = False):
    """-> (grads keyed by the reference's state_dict names, dbags or None).  ONE library call (amds_transmil_train_backward)."""
    if "arena" not in saved:
        return backward_stepwise(saved, dlogits, need_params=need_params, need_bags=need_bags)
    import ctypes as C
    Fd, Cd, Cc = saved["dims"]
    Bb, Tn, _ = saved["shape"]
    dev = dlogits.device
    f32 = dict(dtype=torch.float32, device=dev)
    dlogits = dlogits.contiguous().float()
    lib = _lib.lib()
    need = lib.amds_transmil_train_workspace_bytes(C.byref(saved["cfg"]), Bb, Tn)
    if need == 0:
        _lib.check(-1, "transmil_train_workspace_bytes")
    ws = ops.scratch("transmil_train", dev, need)
    G: dict[str, torch.Tensor] = {}
    gc = corr = None
    if need_params:
        G = {"_fc1.0.weight": torch.empty(Cd, Fd, **f32), "_fc1.0.bias": torch.empty(Cd, **f32), "cls_token": torch.empty(1, 1, Cd, **f32),
             "norm.weight": torch.empty(Cd, **f32), "norm.bias": torch.empty(Cd, **f32), "_fc2.weight": torch.empty(Cc, Cd, **f32), "_fc2.bias": torch.empty(Cc, **f32)}
        corr = torch.empty(50, Cd, **f32)
        gc = _lib.TransMilGrads()
        gc.fc1_w, gc.fc1_b, gc.cls_token, gc.ppeg_corr = G["_fc1.0.weight"].data_ptr(), G["_fc1.0.bias"].data_ptr(), G["cls_token"].data_ptr(), corr.data_ptr()
        gc.norm_w, gc.norm_b, gc.fc2_w, gc.fc2_b = G["norm.weight"].data_ptr(), G["norm.bias"].data_ptr(), G["_fc2.weight"].data_ptr(), G["_fc2.bias"].data_ptr()
        for i, nm in enumerate(("layer1", "layer2")):
            L = {f"{nm}.norm.weight": torch.empty(Cd, **f32), f"{nm}.norm.bias": torch.empty(Cd, **f32), f"{nm}.attn.to_qkv.weight": torch.empty(3 * Cd, Cd, **f32),
                 f"{nm}.attn.to_out.0.weight": torch.empty(Cd, Cd, **f32), f"{nm}.attn.to_out.0.bias": torch.empty(Cd, **f32),
                 f"{nm}.attn.res_conv.weight": torch.empty(HEADS, 1, CONV_K, 1, **f32)}
            G.update(L)
            gc.layer[i] = _lib.TransMilLayerGrads(*[t.data_ptr() for t in L.values()])
    dbags = torch.empty(Bb * Tn, Fd, **f32) if need_bags else None
    arena = saved["arena"]
    _lib.check(lib.amds_transmil_train_backward(C.byref(saved["cfg"]), C.byref(saved["w"]), dlogits.data_ptr(), saved["p_out"], int(saved["seed"]) & (2 ** 64 - 1), Bb, Tn,
                                                arena.data_ptr(), arena.numel(), C.byref(gc) if gc is not None else None, dbags.data_ptr() if dbags is not None else None,
                                                ws.data_ptr(), ws.numel(), ops._stream()), "transmil_train_backward")
    if need_params:      # PPEG: the three kernels' gradients are windows of one tap-correlation table (include/amdstamp.h, amds_ppeg_wgrad)
        c7 = corr[:49].t().reshape(Cd, 7, 7)
        G["pos_layer.proj.weight"] = c7.reshape(Cd, 1, 7, 7).contiguous()
        G["pos_layer.proj1.weight"] = c7[:, 1:6, 1:6].reshape(Cd, 1, 5, 5).contiguous()
        G["pos_layer.proj2.weight"] = c7[:, 2:5, 2:5].reshape(Cd, 1, 3, 3).contiguous()
        for c in ("proj", "proj1", "proj2"):
            G[f"pos_layer.{c}.bias"] = corr[49].clone()
    return G, (dbags.view(Bb, Tn, Fd) if need_bags else None)


def forward_train_stepwise(get, bags: torch.Tensor, dims: tuple[int, int, int], *, training: bool, seed: int = 0):
    """The training forward as a host-side loop around the per-layer attention calls (what `forward_train` did before amds_transmil_train_forward
    existed): the cross-check of the C entry points in tests/ -- bit-identical results."""
    Fd, Cd, C = dims
    Bb, Tn, _ = bags.shape
    dev = bags.device
    lib, st = _lib.lib(), ops._stream()
    p_out = P_OUT if training else 0.0
    a = bags.reshape(Bb * Tn, Fd).float().contiguous()
    w1, b1 = get("_fc1.0.weight"), get("_fc1.0.bias")
    h = ops.linear_f32(a, w1, b1, relu=True).view(Bb, Tn, Cd)
    side = int(math.ceil(math.sqrt(Tn)))
    add = side * side - Tn
    x = torch.cat([get("cls_token").reshape(1, 1, Cd).expand(Bb, -1, -1), h, h[:, :add]], dim=1).contiguous()        # [Bb, n, C] (:306-313)
    n = x.shape[1]
    saved = dict(a=a, h=h, shape=(Bb, Tn, Fd), side=side, add=add, n=n, dims=dims)
    layers = {}
    for name in ("layer1", "pos", "layer2"):
        if name == "pos":
            pw = [get(f"pos_layer.{c}.weight").reshape(Cd, -1).contiguous() for c in ("proj", "proj1", "proj2")]
            pb = [get(f"pos_layer.{c}.bias").contiguous() for c in ("proj", "proj1", "proj2")]
            y = torch.empty_like(x)
            _lib.check(lib.amds_ppeg(x.data_ptr(), y.data_ptr(), pw[0].data_ptr(), pb[0].data_ptr(), pw[1].data_ptr(), pb[1].data_ptr(), pw[2].data_ptr(),
                                     pb[2].data_ptr(), Bb, side, side, Cd, st), "ppeg")
            saved["ppeg"] = dict(x=x, w=pw)
            x = y
            continue
        P = _Nys(get, name)
        x_in = x
        y, mu, rs = T.layernorm_train(x_in, P.norm_w, P.norm_b, 1e-5, torch.float32, rows=Bb * n, row_stride=Cd)
        x = x_in.clone()
        S = nystrom_forward(y.view(Bb, n, Cd), P, x, p_out, seed, 1 if name == "layer1" else 2)
        S.update(x_in=x_in, mu=mu, rs=rs, P=P)
        layers[name] = S
    saved["layers"] = layers
    nw, nb_ = get("norm.weight"), get("norm.bias")
    clsn, muf, rsf = T.layernorm_train(x, nw, nb_, 1e-5, torch.float32, rows=Bb, row_stride=n * Cd)
    w2, b2 = get("_fc2.weight"), get("_fc2.bias")
    logits = ops.linear_f32(clsn, w2, b2)
    saved.update(x=x, clsn=clsn, muf=muf, rsf=rsf, nw=nw, w2=w2, w1=w1)
    return logits, saved


def backward_stepwise(saved: dict, dlogits: torch.Tensor, *, need_params: bool = True, need_bags: bool = False):
    """Backward of `forward_train_stepwise` (its saved dict): host-side loop, same results as `backward`."""
    Fd, Cd, C = saved["dims"]
    Bb, Tn, _ = saved["shape"]
    n, side, add = saved["n"], saved["side"], saved["add"]
    dev = dlogits.device
    f32 = dict(dtype=torch.float32, device=dev)
    lib, st = _lib.lib(), ops._stream()
    G: dict[str, torch.Tensor] = {}
    dlogits = dlogits.contiguous().float()
    clsn, x = saved["clsn"], saved["x"]
    if need_params:
        gW = torch.empty(C, Cd, **f32)
        _bg(dlogits.t().contiguous(), Bb, 0, 0, clsn, Cd, 0, 0, False, gW, Cd, 0, 0, 1, 1, C, Cd, Bb)
        G["_fc2.weight"], G["_fc2.bias"] = gW, T.colsum(dlogits)
    dcls = torch.empty(Bb, Cd, **f32)
    _bg(dlogits, C, 0, 0, saved["w2"], Cd, 0, 0, False, dcls, Cd, 0, 0, 1, 1, Bb, Cd, C)
    dx = torch.zeros(Bb * n, Cd, **f32)
    gw, gb = torch.empty(Cd, **f32), torch.empty(Cd, **f32)
    T.layernorm_bwd(dcls, x, saved["muf"], saved["rsf"], saved["nw"], dx, False, gw, gb, rows=Bb, dy_stride=Cd, x_stride=n * Cd, dx_stride=n * Cd)
    G["norm.weight"], G["norm.bias"] = gw, gb
    for name in ("layer2", "pos", "layer1"):
        if name == "pos":
            pp = saved["ppeg"]
            if need_params:
                nb = lib.amds_ppeg_wgrad_workspace_bytes(Bb, Cd)
                ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                corr = torch.empty(50, Cd, **f32)
                _lib.check(lib.amds_ppeg_wgrad(pp["x"].data_ptr(), dx.data_ptr(), corr.data_ptr(), Bb, side, side, Cd, ws.data_ptr(), nb, st), "ppeg_wgrad")
                c7 = corr[:49].t().reshape(Cd, 7, 7)
                G["pos_layer.proj.weight"] = c7.reshape(Cd, 1, 7, 7).contiguous()
                G["pos_layer.proj1.weight"] = c7[:, 1:6, 1:6].reshape(Cd, 1, 5, 5).contiguous()
                G["pos_layer.proj2.weight"] = c7[:, 2:5, 2:5].reshape(Cd, 1, 3, 3).contiguous()
                for c in ("proj", "proj1", "proj2"):
                    G[f"pos_layer.{c}.bias"] = corr[49].clone()
            wf = [w.flip(1).contiguous() for w in pp["w"]]
            zb = torch.zeros(Cd, **f32)
            dxin = torch.empty(Bb * n, Cd, **f32)
            _lib.check(lib.amds_ppeg(dx.data_ptr(), dxin.data_ptr(), wf[0].data_ptr(), zb.data_ptr(), wf[1].data_ptr(), zb.data_ptr(), wf[2].data_ptr(),
                                     zb.data_ptr(), Bb, side, side, Cd, st), "ppeg(bwd)")
            dx = dxin
            continue
        S = saved["layers"][name]
        dy, Gl = nystrom_backward(S, S["P"], dx.view(Bb, n, Cd), need_params)
        for k, v in Gl.items():
            G[f"{name}.{k}"] = v
        gw, gb = torch.empty(Cd, **f32), torch.empty(Cd, **f32)
        T.layernorm_bwd(dy.view(Bb * n, Cd), S["x_in"], S["mu"], S["rs"], S["P"].norm_w, dx, True, gw, gb, rows=Bb * n, dy_stride=Cd, x_stride=Cd, dx_stride=Cd)
        G[f"{name}.norm.weight"], G[f"{name}.norm.bias"] = gw, gb
    dx3 = dx.view(Bb, n, Cd)
    if need_params:
        G["cls_token"] = T.colsum(dx3[:, 0, :]).view(1, 1, Cd)
    dh = dx3[:, 1:1 + Tn, :].contiguous()
    if add:         # the wrap-padding repeats the first tiles (:306-309): their gradients add up (amds_dropout_add with p = 0 is a plain add)
        tail = dx3[:, 1 + Tn:, :]
        _lib.check(lib.amds_dropout_add(tail.data_ptr(), n * Cd, dh.data_ptr(), Tn * Cd, dh.data_ptr(), Tn * Cd, Bb, add * Cd, 0.0, 0, 0, st), "add")
    dzh = torch.empty(Bb * Tn, Cd, **f32)
    _lib.check(lib.amds_relu_bwd(saved["h"].data_ptr(), dh.data_ptr(), dzh.data_ptr(), dzh.numel(), st), "relu_bwd")
    if need_params:
        G["_fc1.0.weight"], G["_fc1.0.bias"] = _wgrad(dzh.view(Bb, Tn, Cd), saved["a"].view(Bb, Tn, Fd)), T.colsum(dzh)
    dbags = None
    if need_bags:
        dbags = torch.empty(Bb * Tn, Fd, **f32)
        _bg(dzh, Cd, 0, 0, saved["w1"], Fd, 0, 0, False, dbags, Fd, 0, 0, 1, 1, Bb * Tn, Fd, Cd)
        dbags = dbags.view(Bb, Tn, Fd)
    return (G if need_params else {}), dbags
