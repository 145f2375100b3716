"""TransMIL's training step issued kernel by kernel from the host -- what `stamp_amd.transmil_core` did before the Nystrom block (amds_nystrom_attn_fwd /
_bwd) and then the whole step (amds_transmil_train_forward / _backward) became single C calls.  Test infrastructure (moved out of the product
package in round 4): tests/test_gpu_transmil_train.py asserts the C entry points return the same bits as these chains."""
import ctypes as C  # noqa: F401
import math  # noqa: F401

import torch  # noqa: F401

from stamp_amd import _lib, ops  # noqa: F401
from stamp_amd import transmil_core as _tc

CONV_K = _tc.CONV_K
HEADS = _tc.HEADS
ITERS = _tc.ITERS
P_OUT = _tc.P_OUT
T = _tc.T
_Nys = _tc._Nys
_bg = _tc._bg
_mm = _tc._mm
_wgrad = _tc._wgrad
NYSTROM_KERNEL_BY_KERNEL = False      # False: the host loop below calls amds_nystrom_attn_fwd / _bwd per layer; True: every kernel from the host


def nystrom_forward(y, P, x_res, p_drop, seed, sid):
    return (nystrom_forward_stepwise if NYSTROM_KERNEL_BY_KERNEL else _tc.nystrom_forward)(y, P, x_res, p_drop, seed, sid)


def nystrom_backward(S, P, dx, need_params=True):
    return (_tc.nystrom_backward if "arena" in S else nystrom_backward_stepwise)(S, P, dx, need_params)


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



# ---- the eval / deploy forward of the TransMIL head, kernel by kernel (what `stamp_amd.mil.TransMIL.forward` did before amds_transmil_forward existed);
# `self` is the stamp_amd.mil.TransMIL module ------------------------------------------------------------------------------------------------
from stamp_amd.mil import _bgemm  # noqa: E402


def _nystrom_eval(self, y, layer, x_res) -> None:
    """x_res += NystromAttention(y)  (reference :81-163 with mask=None, eval; :258-263 residual)."""
    import math

    b, n, Cd = y.shape
    H, m, iters = 8, Cd // 2, 6
    d = Cd // H
    rem = n % m
    pad = (m - rem) if rem > 0 else 0
    if pad:
        y = torch.nn.functional.pad(y, (0, 0, pad, 0), value=0.0)                      # FRONT padding (:100)
    np_ = n + pad
    dev = y.device
    f32 = dict(dtype=torch.float32, device=dev)
    attn = layer.attn
    qkv = torch.empty(b, np_, 3 * Cd, **f32)
    wq = self._f(attn.to_qkv.weight)
    _bgemm(y.data_ptr(), Cd, 0, 0, wq.data_ptr(), Cd, 0, 0, True, qkv.data_ptr(), 3 * Cd, 0, 0, 1, 1, b * np_, 3 * Cd, Cd)
    e4 = 4
    qp, kp, vp = qkv.data_ptr(), qkv.data_ptr() + Cd * e4, qkv.data_ptr() + 2 * Cd * e4
    sb, sh, ld = np_ * 3 * Cd, d, 3 * Cd
    scale = d ** -0.5
    l = math.ceil(n / m)
    ql, kl = torch.empty(b, H, m, d, **f32), torch.empty(b, H, m, d, **f32)
    lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    _lib.check(lib.amds_landmark_mean(qp, sb, sh, ld, ql.data_ptr(), b, H, m, l, d, scale / l, st), "landmark_mean")
    _lib.check(lib.amds_landmark_mean(kp, sb, sh, ld, kl.data_ptr(), b, H, m, l, d, 1.0 / l, st), "landmark_mean")
    a1 = torch.empty(b, H, np_, m, **f32)
    a2 = torch.empty(b, H, m, m, **f32)
    a3 = torch.empty(b, H, m, np_, **f32)
    _bgemm(qp, ld, sb, sh, kl.data_ptr(), d, H * m * d, m * d, True, a1.data_ptr(), m, H * np_ * m, np_ * m, b, H, np_, m, d, alpha=scale)
    _bgemm(ql.data_ptr(), d, H * m * d, m * d, kl.data_ptr(), d, H * m * d, m * d, True, a2.data_ptr(), m, H * m * m, m * m, b, H, m, m, d)
    _bgemm(ql.data_ptr(), d, H * m * d, m * d, kp, ld, sb, sh, True, a3.data_ptr(), np_, H * m * np_, m * np_, b, H, m, np_, d)
    for t, cols in ((a1, m), (a2, m), (a3, np_)):
        _lib.check(lib.amds_softmax_rows(t.data_ptr(), t.numel() // cols, cols, st), "softmax_rows")
    # Moore-Penrose iteration (:23-37): z <- 0.25 z (13 I - xz (15 I - xz (7 I - xz))),  xz = x z
    z, z2 = torch.empty_like(a2), torch.empty_like(a2)
    xz, t1, t2 = torch.empty_like(a2), torch.empty_like(a2), torch.empty_like(a2)
    scratch = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.check(lib.amds_pinv_init(a2.data_ptr(), z.data_ptr(), b * H, m, scratch.data_ptr(), st), "pinv_init")
    mm = m * m

    def sq(A, B, Cm, alpha, diag):
        _bgemm(A.data_ptr(), m, mm, 0, B.data_ptr(), m, mm, 0, False, Cm.data_ptr(), m, mm, 0, b * H, 1, m, m, m, alpha=alpha, diag=diag)
    for _ in range(iters):
        sq(a2, z, xz, 1.0, 0.0)
        sq(a2, z, t1, -1.0, 7.0)          # 7I - xz
        sq(xz, t1, t2, -1.0, 15.0)        # 15I - xz(7I - xz)
        sq(xz, t2, t1, -1.0, 13.0)        # 13I - xz(...)
        sq(z, t1, z2, 0.25, 0.0)
        z, z2 = z2, z
    av = torch.empty(b, H, m, d, **f32)                                                     # attn3 @ v
    _bgemm(a3.data_ptr(), np_, H * m * np_, m * np_, vp, ld, sb, sh, False, av.data_ptr(), d, H * m * d, m * d, b, H, m, d, np_)
    a1z = torch.empty(b, H, np_, m, **f32)                                                  # attn1 @ pinv
    _bgemm(a1.data_ptr(), m, H * np_ * m, np_ * m, z.data_ptr(), m, H * mm, mm, False, a1z.data_ptr(), m, H * np_ * m, np_ * m, b, H, np_, m, m)
    merged = torch.empty(b, np_, Cd, **f32)                                                 # heads merged: [b, n, (h d)]
    _bgemm(a1z.data_ptr(), m, H * np_ * m, np_ * m, av.data_ptr(), d, H * m * d, m * d, False, merged.data_ptr(), Cd, np_ * Cd, d, b, H, np_, d, m)
    wc = self._f(attn.res_conv.weight).reshape(H, -1)                                       # [H, 33]
    _lib.check(lib.amds_dwconv_seq(vp, sb, sh, ld, wc.data_ptr(), merged.data_ptr(), np_ * Cd, d, Cd, b, H, np_, d, wc.shape[1], st), "dwconv_seq")
    wo, bo = self._f(attn.to_out[0].weight), self._f(attn.to_out[0].bias)
    # to_out on the LAST n rows of every bag (:155), accumulated into the residual stream
    _bgemm(merged.data_ptr() + pad * Cd * e4, Cd, np_ * Cd, 0, wo.data_ptr(), Cd, 0, 0, True, x_res.data_ptr(), Cd, n * Cd, 0, b, 1, n, Cd, Cd,
           bias=bo.data_ptr(), accumulate=True)


def transmil_forward_stepwise(self, h):
    """The same forward, one library call per kernel from the host (what `forward` did before amds_transmil_forward existed): kept as the
    cross-check of the C entry point in tests/ -- bit-identical logits."""
    import math
    Bb, T, F = h.shape
    Cd = self.dim_hidden
    x = ops.linear_f32(h.reshape(Bb * T, F).float().contiguous(), self._f(self._fc1[0].weight), self._f(self._fc1[0].bias), relu=True)
    x = x.view(Bb, T, Cd)
    side = int(math.ceil(math.sqrt(T)))
    x = torch.cat([x, x[:, : side * side - T]], dim=1)                                       # wrap-pad with the FIRST tiles (:306-309)
    x = torch.cat([self._f(self.cls_token).expand(Bb, -1, -1), x], dim=1).contiguous()        # [Bb, n, C]
    n = x.shape[1]
    lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    for name in ("layer1", "pos", "layer2"):
        if name == "pos":
            y = torch.empty_like(x)
            pl = self.pos_layer
            w7, w5, w3 = (self._f(c.weight).reshape(Cd, -1) for c in (pl.proj, pl.proj1, pl.proj2))
            _lib.check(lib.amds_ppeg(x.data_ptr(), y.data_ptr(), w7.data_ptr(), self._f(pl.proj.bias).data_ptr(), w5.data_ptr(),
                                     self._f(pl.proj1.bias).data_ptr(), w3.data_ptr(), self._f(pl.proj2.bias).data_ptr(), Bb, side, side, Cd, st), "ppeg")
            x = y
            continue
        layer = getattr(self, name)
        y = ops.layernorm(x.view(Bb * n, Cd), self._f(layer.norm.weight), self._f(layer.norm.bias), 1e-5, torch.float32).view(Bb, n, Cd)
        _nystrom_eval(self, y, layer, x)                                                           # x += attn(norm(x))
    cls = ops.layernorm_rows(x.view(-1), Bb, Cd, n * Cd, self._f(self.norm.weight), self._f(self.norm.bias), 1e-5, torch.float32)
    return ops.linear_f32(cls, self._f(self._fc2.weight), self._f(self._fc2.bias))
