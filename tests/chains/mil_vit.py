"""The MIL `vit` head's launch chains issued kernel by kernel from the host -- what `stamp_amd.mil_core.forward_infer / forward_train / backward` did
before the whole forward / training step became ONE C call each (amds_mil_vit_forward, amds_mil_vit_train_forward / _backward).  Test
infrastructure (moved out of the product package in round 4): the tests assert that the C entry points return the same bits as these chains."""
import ctypes as C  # noqa: F401

import torch  # noqa: F401

from stamp_amd import _lib, ops  # noqa: F401
from stamp_amd import mil_core as _mc

BF = _mc.BF
PackedVit = _mc.PackedVit
T = _mc.T
_CFG_TRAIN = _mc._CFG_TRAIN
_ENC = _mc._ENC
_coords_with_cls = _mc._coords_with_cls
_gelu_drop_bwd = _mc._gelu_drop_bwd
_gelu_drop_fwd = _mc._gelu_drop_fwd
_ln = _mc._ln
_up = _mc._up
layer_prefix = _mc.layer_prefix


def forward_infer_stepwise(pk: PackedVit, bags: torch.Tensor, coords: torch.Tensor | None, mask: torch.Tensor | None) -> torch.Tensor:
    """The same forward, one library call per kernel from the host (what `forward_infer` did before amds_mil_vit_forward existed): kept
    as the cross-check of the C entry point in tests/ -- results are bit-identical."""
    d = pk.dims
    Bb, Tn, Fd = bags.shape
    dev = bags.device
    act = pk.act
    a = bags.reshape(Bb * Tn, Fd)
    a = a.contiguous() if (a.dtype == act and d.Fp == Fd) else ops.cast_pad(a.float(), d.Fp, act)
    proj = ops.gemm(a, pk.w["proj_w"], _lib.EPI_BIAS_GELU_F32, bias=pk.m["proj_b"])                  # [Bb*T, Dp] fp32, eval: Dropout = identity
    S = Tn + 1
    M = Bb * S
    x = torch.empty(Bb, S, d.Dp, dtype=torch.float32, device=dev)
    x[:, 0] = pk.m["cls"]                                                                            # :347-348
    x[:, 1:] = proj.view(Bb, Tn, d.Dp)
    x = x.view(M, d.Dp)
    c = pad = None
    if d.alibi:
        if coords is None:
            raise ValueError("use_alibi=True needs coords")
        c = _coords_with_cls(coords, Bb, dev)
    if mask is not None:
        if mask.shape != (Bb, Tn):
            raise ValueError(f"mask must be [batch, tile] = {(Bb, Tn)}, got {tuple(mask.shape)}")
        pad = torch.cat([mask.new_zeros(Bb, 1), mask], dim=1).to(dev, torch.uint8).contiguous()      # class token never padded (:356-358)
    hbuf = torch.zeros(M, d.Dp, dtype=act, device=dev) if d.Dp != d.D else None
    lib, st = _lib.lib(), ops._stream()
    cls_tail = bool(_lib.lib().amds_get_mil_cls_tail(_lib.ctx(torch.cuda.current_device()))) and not d.alibi          # (a padding mask does not touch the class query's row: csrc/mil_vit.hip)
    n_layers = len(pk.m["layers"])
    for li, (Lm, Lw) in enumerate(zip(pk.m["layers"], pk.w["layers"])):
        h = _ln(x, M, d.D, d.Dp, *Lm["ln1"], act, d.Dp, hbuf)
        if cls_tail and li == n_layers - 1:      # class-row tail of the last block (csrc/mil_vit.hip): k | v of all tokens, the rest on the class rows alone
            qkv = torch.empty(M, 3 * d.Da, dtype=act, device=dev)
            ops.gemm(h, Lw["in_w"][d.Da:], _lib.EPI_BIAS, bias=Lm["in_b"][d.Da:], out=qkv[:, d.Da:])
            hc, xc = h.view(Bb, S, d.Dp)[:, 0], x.view(Bb, S, d.Dp)[:, 0]
            qc = ops.gemm(hc, Lw["in_w"][:d.Da], _lib.EPI_BIAS, bias=Lm["in_b"][:d.Da])
            oc = ops.attention_row(qc, qkv, Bb, S, d.Ha)
            ops.gemm(oc, Lw["out_w"], _lib.EPI_RESIDUAL, bias=Lm["out_b"], out=xc)
            h2 = _ln(x, Bb, d.D, S * d.Dp, *Lm["ln2"], act, d.Dp, hbuf[:Bb] if hbuf is not None else None)
            u = ops.gemm(h2, Lw["fc1_w"], _lib.EPI_BIAS_GELU, bias=Lm["fc1_b"])
            ops.gemm(u, Lw["fc2_w"], _lib.EPI_RESIDUAL, bias=Lm["fc2_b"], out=xc)
            continue
        qkv = ops.gemm(h, Lw["in_w"], _lib.EPI_BIAS, bias=Lm["in_b"])
        if d.alibi:
            scale = (Lm["bias_scale"] * Lm["inv_rm"]).contiguous()
            if pad is None:
                att = ops.attention_alibi(qkv, c, scale, Bb, S, d.Ha)
            else:
                att = torch.empty(M, d.Da, dtype=BF, device=dev)
                _lib.check(lib.amds_attention_alibi_masked(qkv.data_ptr(), c.data_ptr(), scale.data_ptr(), pad.data_ptr(), att.data_ptr(), Bb, S,
                                                           d.Ha, ops.act_code(act), st), "attention_alibi_masked")
        elif pad is None:
            att = ops.attention(qkv, Bb, S, d.Ha)
        else:       # d.H: the reference's head-repeated mask indexing (b*H + h) % B counts REAL heads; padded heads output zeros anyway
            att = torch.empty(M, d.Da, dtype=act, device=dev)
            _lib.check(lib.amds_attention_masked(qkv.data_ptr(), pad.data_ptr(), att.data_ptr(), Bb, S, d.Ha, d.H, ops.act_code(act), st), "attention_masked")
        ops.gemm(att, Lw["out_w"], _lib.EPI_RESIDUAL, bias=Lm["out_b"], out=x)                       # x = attn(x) + x   (:291-292)
        h = _ln(x, M, d.D, d.Dp, *Lm["ln2"], act, d.Dp, hbuf)
        u = ops.gemm(h, Lw["fc1_w"], _lib.EPI_BIAS_GELU, bias=Lm["fc1_b"])
        ops.gemm(u, Lw["fc2_w"], _lib.EPI_RESIDUAL, bias=Lm["fc2_b"], out=x)                         # x = ff(x) + x     (:293)
    cls = _ln(x, Bb, d.D, S * d.Dp, *pk.m["norm"], torch.float32, d.D)                               # final LN, class token only
    return ops.linear_f32(cls, pk.m["head_w"], pk.m["head_b"])


def forward_train_stepwise(pk: PackedVit, bags: torch.Tensor, coords: torch.Tensor | None, *, training: bool, seed: int = 0):
    """The training forward, one library call per kernel from the host (what `forward_train` did before amds_mil_vit_train_forward
    existed): kept as the cross-check of the C entry points in tests/ -- logits and gradients are bit-identical."""
    d = pk.dims
    dev = bags.device
    Bb, Tn, Fd = bags.shape
    S = Tn + 1
    Mt, M = Bb * Tn, Bb * S
    Dp, Da, Ha = d.Dp, d.Da, d.Ha
    p_proj = d.p_drop if training else 0.0
    p_att = d.p_drop if (training and not d.alibi) else 0.0
    p_ff = d.p_ff if training else 0.0
    lib, st = _lib.lib(), ops._stream()
    src = bags.reshape(Mt, Fd).contiguous()
    if src.dtype == torch.float16 and d.Fp == Fd:
        a = torch.empty(Mt, Fd, dtype=BF, device=dev)
        _lib.check(lib.amds_convert_f16_bf16(src.data_ptr(), a.data_ptr(), src.numel(), st), "convert")
    elif src.dtype == BF and d.Fp == Fd:
        a = src
    else:
        a = ops.cast_pad(src.float(), d.Fp, BF)
    zp = ops.gemm(a, pk.w["proj_w"], _lib.EPI_BIAS, bias=pk.m["proj_b"], cfg=_CFG_TRAIN)                             # bf16 [Mt, Dp]
    xp = _gelu_drop_fwd(zp, torch.float32, p_proj, seed, 1000)
    x = torch.empty(Bb, S, Dp, dtype=torch.float32, device=dev)
    x[:, 0] = pk.m["cls"]
    x[:, 1:] = xp.view(Bb, Tn, Dp)
    x = x.view(M, Dp)
    cc = None
    if d.alibi:
        if coords is None:
            raise ValueError("use_alibi=True needs coords")
        cc = _coords_with_cls(coords, Bb, dev)
    layers = []
    zbuf = (lambda: torch.zeros(M, Dp, dtype=BF, device=dev)) if Dp != d.D else (lambda: None)
    for l, (Lm, Lw) in enumerate(zip(pk.m["layers"], pk.w["layers"])):
        h1, mu1, rs1 = T.layernorm_train(x, *Lm["ln1"], 1e-5, BF, rows=M, row_stride=Dp, out=zbuf(), ld_out=Dp)
        qkv = ops.gemm(h1, Lw["in_w"], _lib.EPI_BIAS, bias=Lm["in_b"], cfg=_CFG_TRAIN)
        x_mid = x.clone()
        if d.alibi:
            att, u_al, osm, lse = T.attention_alibi_fwd_train(qkv, cc, Lm["inv_rm"], Lm["bias_scale"], Bb, S, Ha)
            lse = (lse, u_al, osm)
        else:
            att, lse = T.attention_fwd_train(qkv, Bb, S, Ha, p_att, seed, 10 * l + 1)
        ops.gemm(att, Lw["out_w"], _lib.EPI_RESIDUAL, bias=Lm["out_b"], out=x_mid, cfg=_CFG_TRAIN)
        h2, mu2, rs2 = T.layernorm_train(x_mid, *Lm["ln2"], 1e-5, BF, rows=M, row_stride=Dp, out=zbuf(), ld_out=Dp)
        z = ops.gemm(h2, Lw["fc1_w"], _lib.EPI_BIAS, bias=Lm["fc1_b"], cfg=_CFG_TRAIN)
        u = _gelu_drop_fwd(z, None, p_ff, seed, 10 * l + 2)
        if p_ff > 0.0:          # x_out = x_mid + Dropout(fc2(u))   (:167-168)
            y = ops.gemm(u, Lw["fc2_w"], _lib.EPI_BIAS_F32, bias=Lm["fc2_b"], cfg=_CFG_TRAIN)
            x_out = torch.empty_like(x_mid)
            _lib.check(lib.amds_dropout_add(y.data_ptr(), Dp, x_mid.data_ptr(), Dp, x_out.data_ptr(), Dp, M, Dp, p_ff, seed, 10 * l + 3, st), "dropout_add")
        else:
            x_out = x_mid.clone()
            ops.gemm(u, Lw["fc2_w"], _lib.EPI_RESIDUAL, bias=Lm["fc2_b"], out=x_out, cfg=_CFG_TRAIN)
        layers.append((x, h1, mu1, rs1, qkv, att, lse, x_mid, h2, mu2, rs2, z, u))
        x = x_out
    clsn, muf, rsf = T.layernorm_train(x, *pk.m["norm"], 1e-5, torch.float32, rows=Bb, row_stride=S * Dp)
    logits = ops.linear_f32(clsn, pk.m["head_w"], pk.m["head_b"])
    saved = dict(a=a, zp=zp, x=x, clsn=clsn, muf=muf, rsf=rsf, layers=layers, cc=cc, shape=(Bb, Tn, Fd), p=(p_proj, p_att, p_ff), seed=seed)
    return logits, saved


def _bgemm(A, lda, B, ldb, transb, Cm, ldc, M, N, K):
    _lib.check(_lib.lib().amds_bgemm_f32(A.data_ptr(), lda, 0, 0, B.data_ptr(), ldb, 0, 0, 1 if transb else 0, Cm.data_ptr(), ldc, 0, 0, 1, 1,
                                         M, N, K, 1.0, 0.0, None, 0, ops._stream()), "bgemm_f32")


def backward_stepwise(pk: PackedVit, saved: dict, dlogits: torch.Tensor, *, need_params: bool = True, need_bags: bool = False, split_k: int = 32):
    """Backward of `forward_train_stepwise` (its `saved`), kernel by kernel from the host; same results as `backward`."""
    d = pk.dims
    dev = dlogits.device
    Bb, Tn, Fd = saved["shape"]
    S = Tn + 1
    Mt, M = Bb * Tn, Bb * S
    D, Dp, Da, Ha, FFp, Fp, C = d.D, d.Dp, d.Da, d.Ha, d.FFp, d.Fp, d.C
    p_proj, p_att, p_ff = saved["p"]
    seed = saved["seed"]
    lib, st = _lib.lib(), ops._stream()
    G: dict[str, torch.Tensor] = {}
    unit = 64 * split_k
    Mp, Mtp = _up(M, unit), _up(Mt, unit)
    tbuf: dict = {}

    def tr(t: torch.Tensor, key: str) -> torch.Tensor:      # [M, cols] bf16 -> [cols, Mp], zero-padded scratch reused per width
        k = (key, t.shape[1])
        if k not in tbuf:
            tbuf[k] = torch.zeros(t.shape[1], Mp, dtype=BF, device=dev)
        return T.transpose16(t, out=tbuf[k])

    def wgrad(dyT: torch.Tensor, xT: torch.Tensor, Nn: int, Kk: int, Mpad: int) -> torch.Tensor:
        """dW[N][K] = dy^T x, contraction over the (padded) token dimension split into split_k fp32 partials."""
        chunk = Mpad // split_k
        part = torch.empty(split_k, Nn * Kk, dtype=torch.float32, device=dev)
        T.gemm_batched(dyT, Mpad, chunk, xT, Mpad, chunk, Nn, Kk, chunk, split_k, BF, part, Kk, Nn * Kk, True)
        return T.colsum(part).view(Nn, Kk)

    dlogits = dlogits.contiguous().float()
    clsn, x = saved["clsn"], saved["x"]
    if need_params:
        gW = torch.empty(C, D, dtype=torch.float32, device=dev)
        _bgemm(dlogits.t().contiguous(), Bb, clsn, D, False, gW, D, C, D, Bb)                           # dW_head = dlogits^T clsn
        G["mlp_head.0.weight"], G["mlp_head.0.bias"] = gW, T.colsum(dlogits)
    dcls = torch.empty(Bb, D, dtype=torch.float32, device=dev)
    _bgemm(dlogits, C, pk.m["head_w"], D, False, dcls, D, Bb, D, C)                                      # dclsn = dlogits W_head
    dx = torch.zeros(M, Dp, dtype=torch.float32, device=dev)
    gn_w, gn_b = torch.empty(D, device=dev), torch.empty(D, device=dev)
    T.layernorm_bwd(dcls, x, saved["muf"], saved["rsf"], pk.m["norm"][0], dx, False, gn_w, gn_b, rows=Bb, dy_stride=D, x_stride=S * Dp, dx_stride=S * Dp)
    G["transformer.norm.weight"], G["transformer.norm.bias"] = gn_w, gn_b
    for l in reversed(range(d.L)):
        p = layer_prefix(l)
        Lm, Lw, Lt = pk.m["layers"][l], pk.w["layers"][l], pk.wt["layers"][l]
        x_in, h1, mu1, rs1, qkv, att, lse, x_mid, h2, mu2, rs2, z, u = saved["layers"][l]
        # ---- feed-forward branch: x_out = x_mid + drop(fc2(drop(gelu(fc1(LN(x_mid))))))
        if p_ff > 0.0:
            dyb = torch.empty(M, Dp, dtype=BF, device=dev)
            _lib.check(lib.amds_dropout_cast_bwd(dx.data_ptr(), Dp, dyb.data_ptr(), Dp, M, Dp, _lib.BF16, p_ff, seed, 10 * l + 3, st), "dropout_cast_bwd")
        else:
            dyb = ops.cast_pad(dx, Dp, BF)
        du = ops.gemm(dyb, Lt["fc2_w"], _lib.EPI_BIAS, cfg=_CFG_TRAIN)                                                   # [M, FFp] = dy W2
        if need_params:
            G[p + "1.4.weight"] = wgrad(tr(dyb, "g"), tr(u, "a"), Dp, FFp, Mp)[:D, : d.FF]
            G[p + "1.4.bias"] = (T.colsum(dyb) if p_ff > 0.0 else T.colsum(dx))[:D]
        dz = _gelu_drop_bwd(z, du, p_ff, seed, 10 * l + 2)
        dh2 = ops.gemm(dz, Lt["fc1_w"], _lib.EPI_BIAS_F32, cfg=_CFG_TRAIN)                                               # [M, Dp] fp32
        if need_params:
            G[p + "1.1.weight"] = wgrad(tr(dz, "g"), tr(h2, "a"), FFp, Dp, Mp)[: d.FF, :D]
            G[p + "1.1.bias"] = T.colsum(dz)[: d.FF]
        g_w, g_b = torch.empty(D, device=dev), torch.empty(D, device=dev)
        T.layernorm_bwd(dh2, x_mid, mu2, rs2, Lm["ln2"][0], dx, True, g_w, g_b, rows=M, dy_stride=Dp, x_stride=Dp, dx_stride=Dp)
        G[p + "1.0.weight"], G[p + "1.0.bias"] = g_w, g_b
        # ---- attention branch: x_mid = x_in + out_proj(attention(in_proj(LN(x_in))))
        dxb = ops.cast_pad(dx, Dp, BF)                                                                   # d(x_mid)
        datt = ops.gemm(dxb, Lt["out_w"], _lib.EPI_BIAS, cfg=_CFG_TRAIN)                                                 # [M, Da]
        out_name = "0.mhsa.fc." if d.alibi else "0.mhsa.out_proj."
        if need_params:
            G[p + out_name + "weight"] = pk.unpad_out_w(wgrad(tr(dxb, "g"), tr(att, "a"), Dp, Da, Mp))
            G[p + out_name + "bias"] = T.colsum(dx)[:D]
        if d.alibi:
            lse_, u_al, osm = lse
            dqkv, dbs = T.attention_alibi_bwd(qkv, osm, u_al, datt, lse_, saved["cc"], Lm["bias_scale"], (Lm["bias_scale"] * Lm["inv_rm"]).contiguous(),
                                              Bb, S, Ha)
            if need_params:
                for h in range(d.H):
                    G[p + f"0.mhsa.attentions.{h}.bias_scale"] = dbs[h:h + 1].clone()
        else:
            dqkv = T.attention_bwd_train(qkv, att, datt, lse, Bb, S, Ha, p_att, seed, 10 * l + 1)
        if need_params:
            gw = pk.unpad_in_w(wgrad(tr(dqkv, "g"), tr(h1, "a"), 3 * Da, Dp, Mp))                          # [3, H, hd, D]
            gb = pk.unpad_in_b(T.colsum(dqkv))                                                           # [3, H, hd]
            if d.alibi:
                for i, e in enumerate(_ENC):
                    for h in range(d.H):
                        G[p + f"0.mhsa.{e}.{h}.weight"], G[p + f"0.mhsa.{e}.{h}.bias"] = gw[i, h], gb[i, h]
            else:
                G[p + "0.mhsa.in_proj_weight"], G[p + "0.mhsa.in_proj_bias"] = gw.reshape(3 * D, D), gb.reshape(3 * D)
        dh1 = ops.gemm(dqkv, Lt["in_w"], _lib.EPI_BIAS_F32, cfg=_CFG_TRAIN)
        g_w, g_b = torch.empty(D, device=dev), torch.empty(D, device=dev)
        T.layernorm_bwd(dh1, x_in, mu1, rs1, Lm["ln1"][0], dx, True, g_w, g_b, rows=M, dy_stride=Dp, x_stride=Dp, dx_stride=Dp)
        G[p + "0.norm.weight"], G[p + "0.norm.bias"] = g_w, g_b
    dx3 = dx.view(Bb, S, Dp)
    if need_params:
        G["class_token"] = T.colsum(dx3[:, 0, :])[:D]                                                    # rows at stride S*Dp
    dxp = dx3[:, 1:, :].reshape(Mt, Dp)                                                                  # contiguous copy (data movement)
    dzp = _gelu_drop_bwd(saved["zp"], dxp, p_proj, seed, 1000)                                           # bf16
    if need_params:
        dzpT = T.transpose16(dzp, ld_dst=Mtp)
        aT = T.transpose16(saved["a"], ld_dst=Mtp)
        G["project_features.0.weight"] = wgrad(dzpT, aT, Dp, Fp, Mtp)[:D, :Fd]
        G["project_features.0.bias"] = T.colsum(dzp)[:D]
    dbags = None
    if need_bags:
        dbags = ops.gemm(dzp, pk.wt["proj_w"], _lib.EPI_BIAS_F32, cfg=_CFG_TRAIN)[:, :Fd].reshape(Bb, Tn, Fd)
    if not need_params:
        G = {}
    return G, dbags
