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
    ops.sync_float32_matmul_precision()
    _lib.check(_lib.lib().amds_bgemm_f32(p(A), lda, sAo, sAi, p(B), ldb, sBo, sBi, (1 if transb else 0) | (2 if transa else 0), p(Cm), ldc, sCo, sCi, outer, inner,
                                         M, N, K, alpha, diag, None if bias is None else p(bias), 1 if accumulate else 0, ops._stream()), "bgemm_f32")


def _mm_dual(A: torch.Tensor, B: torch.Tensor, alpha: float, diag: float, alpha2: float, diag2: float, transb: bool = False, transa: bool = False):
    """One batched product, two outputs: (alpha A op(B) + diag I, alpha2 A op(B) + diag2 I) -- `amds_bgemm_f32_dual` (the pinv iteration's `xz` and `7 I - xz`,
    reference trans_mil.py:31-33); the bits of two `_mm` calls."""
    if transa:
        Z, K, M = A.shape
    else:
        Z, M, K = A.shape
    N = B.shape[1] if transb else B.shape[2]
    c1 = torch.empty(Z, M, N, dtype=torch.float32, device=A.device)
    c2 = torch.empty_like(c1)
    ops.sync_float32_matmul_precision()
    _lib.check(_lib.lib().amds_bgemm_f32_dual(A.data_ptr(), A.shape[2], M * K, 0, B.data_ptr(), B.shape[2], B.shape[1] * B.shape[2], 0, (1 if transb else 0) | (2 if transa else 0),
                                              c1.data_ptr(), c2.data_ptr(), N, M * N, 0, Z, 1, M, N, K, alpha, diag, alpha2, diag2, ops._stream()), "bgemm_f32_dual")
    return c1, c2


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


def _layer_struct(P: _Nys) -> "_lib.TransMilLayer":
    return _lib.TransMilLayer(P.norm_w.data_ptr(), P.norm_b.data_ptr(), P.wqkv.data_ptr(), P.wo.data_ptr(), P.bo.data_ptr(), P.wconv.data_ptr())


def nystrom_forward(y: torch.Tensor, P: _Nys, x_res: torch.Tensor, p_drop: float, seed: int, sid: int):
    """x_res += Dropout(to_out(NystromAttention(y)))[:, -n:]; returns what the backward needs.  ONE library call (amds_nystrom_attn_fwd,
    csrc/nystrom_train.hip): the intermediates live in one arena the backward reads."""
    import ctypes as C
    b, n, Cd = y.shape
    assert y.is_contiguous() and x_res.is_contiguous() and y.dtype == torch.float32 and x_res.dtype == torch.float32
    lib = _lib.lib()
    need = lib.amds_nystrom_attn_saved_bytes(Cd, b, n)
    if need == 0:
        _lib.check(-1, "nystrom_attn_saved_bytes")
    arena = torch.empty(need, dtype=torch.uint8, device=y.device)
    L = _layer_struct(P)
    ops.sync_float32_matmul_precision()
    _lib.check(lib.amds_nystrom_attn_fwd(C.byref(L), Cd, y.data_ptr(), x_res.data_ptr(), b, n, float(p_drop), int(seed) & (2 ** 64 - 1), int(sid),
                                         arena.data_ptr(), arena.numel(), ops._stream()), "nystrom_attn_fwd")
    return dict(arena=arena, n=n, p_drop=p_drop, seed=seed, sid=sid)


def nystrom_backward(S: dict, P: _Nys, dx: torch.Tensor, need_params: bool = True):
    """dx: gradient of the residual stream after the block [b, n, C] -> (dy [b, n, C] gradient w.r.t. the LayerNorm output, grads).
    ONE library call (amds_nystrom_attn_bwd)."""
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
    ops.sync_float32_matmul_precision()
    _lib.check(lib.amds_nystrom_attn_bwd(C.byref(L), Cd, dx.data_ptr(), dy.data_ptr(), C.byref(gc) if gc is not None else None, b, n, float(S["p_drop"]),
                                         int(S["seed"]) & (2 ** 64 - 1), int(S["sid"]), arena.data_ptr(), arena.numel(), ws.data_ptr(), ws.numel(), ops._stream()),
               "nystrom_attn_bwd")
    return dy, G


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
    import ctypes as C
    Fd, Cd, Cc = dims
    Bb, Tn, _ = bags.shape
    dev = bags.device
    if bags.dtype not in ops._DT:
        bags = bags.float()
    bags = bags.contiguous()
    lib = _lib.lib()
    # resolved HERE so that the backward of this step reads the arena the way this forward wrote it, whatever the context says by then
    cfg = _lib.TransMilCfg(Fd, Cd, Cc, 1 if lib.amds_get_mil_cls_tail(_lib.ctx(dev.index if dev.index is not None else torch.cuda.current_device())) else 0)
    w, keep = _c_weights(get, Cd)
    need = lib.amds_transmil_train_saved_bytes(C.byref(cfg), Bb, Tn)
    if need == 0:
        _lib.check(-1, "transmil_train_saved_bytes")
    arena = torch.empty(need, dtype=torch.uint8, device=dev)
    logits = torch.empty(Bb, Cc, dtype=torch.float32, device=dev)
    p_out = P_OUT if training else 0.0
    ops.sync_float32_matmul_precision()
    _lib.check(lib.amds_transmil_train_forward(C.byref(cfg), C.byref(w), bags.data_ptr(), ops._DT[bags.dtype], p_out, int(seed) & (2 ** 64 - 1), logits.data_ptr(), Bb, Tn,
                                               arena.data_ptr(), arena.numel(), ops._stream()), "transmil_train_forward")
    return logits, dict(arena=arena, cfg=cfg, w=w, keep=keep, shape=(Bb, Tn, Fd), dims=dims, p_out=p_out, seed=seed)


def backward(saved: dict, dlogits: torch.Tensor, *, need_params: bool = True, need_bags: bool = False):
    """-> (grads keyed by the reference's state_dict names, dbags or None).  ONE library call (amds_transmil_train_backward)."""
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
    ops.sync_float32_matmul_precision()
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


