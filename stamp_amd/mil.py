"""MIL heads on the HIP path -- inference forward (deploy / validation), reference state_dict keys.

`VisionTransformer` here mirrors the reference's MIL model class of the same name
(src/stamp/modeling/models/vision_tranformer.py:298-384): same keyword-only constructor, same parameter names
(`class_token`, `project_features.0.*`, `transformer.layers.{l}.0.{norm,mhsa.in_proj_*,mhsa.out_proj.*}`,
`transformer.layers.{l}.1.{0,1,4}.*`, `transformer.norm.*`, `mlp_head.0.*`), so checkpoints interchange through
`load_state_dict`; ``forward(bags, *, coords, mask)`` has the reference signature (pinned by tests/test_model.py:28-32).

Round-1 coverage: the ``mask=None`` forward the Lightning wrappers actually use for validation / predict
(src/stamp/modeling/models/__init__.py:286-313), without ALiBi, in ``torch.no_grad`` / eval.  Training (backward),
and ``mask != None`` raise NotImplementedError -- loudly, there is no torch fallback.  ``use_alibi=True`` runs the
reference's MultiHeadALiBi in eval mode (distance bias subtracted after the softmax, frozen running mean).

Arithmetic plan: bags are fp16 on disk (preprocessing/__init__.py:325), so the projection GEMM consumes them
exactly; MFMA operands fp16, fp32 accumulate, fp32 residual stream and LayerNorm, exact-erf GELU on the fp32
projection output, streaming attention (never materialises T x T, so whole-slide bags fit).
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib, ops


def _pad_rows(w: torch.Tensor, mult: int) -> torch.Tensor:
    r = (-w.shape[0]) % mult
    return w if r == 0 else torch.cat([w, w.new_zeros(r, *w.shape[1:])])


class VisionTransformer(nn.Module):
    def __init__(self, *, dim_output: int, dim_input: int, dim_model: int, n_layers: int, n_heads: int,
                 dim_feedforward: int, dropout: float, use_alibi: bool) -> None:
        super().__init__()
        self.use_alibi = bool(use_alibi)
        if n_heads * 64 != dim_model or dim_model % 128 or dim_feedforward % 128:
            raise NotImplementedError(f"HIP MIL vit needs head_dim 64 and dims that are multiples of 128 "
                                      f"(dim_model={dim_model}, n_heads={n_heads}, dim_feedforward={dim_feedforward})")
        self.dim_output, self.dim_input, self.dim_model = dim_output, dim_input, dim_model
        self.n_layers, self.n_heads, self.dim_feedforward = n_layers, n_heads, dim_feedforward
        D = dim_model
        # parameters with the reference's names and shapes (registered flat; state_dict keys match the reference)
        self.class_token = nn.Parameter(torch.randn(D))
        P = nn.ParameterDict()
        def add(name, *shape, ones=False, zeros=False):
            t = torch.ones(*shape) if ones else torch.zeros(*shape) if zeros else torch.randn(*shape) / (shape[-1] ** 0.5)
            P[name.replace(".", "/")] = nn.Parameter(t)
        add("project_features.0.weight", D, dim_input); add("project_features.0.bias", D, zeros=True)
        for l in range(n_layers):
            p = f"transformer.layers.{l}."
            add(p + "0.norm.weight", D, ones=True); add(p + "0.norm.bias", D, zeros=True)
            if use_alibi:       # MultiHeadALiBi: one Linear(D, 64) per head for q, k, v (+ running-mean scaler, bias_scale), fc
                for h in range(n_heads):
                    for enc in ("query_encoders", "key_encoders", "value_encoders"):
                        add(p + f"0.mhsa.{enc}.{h}.weight", 64, D); add(p + f"0.mhsa.{enc}.{h}.bias", 64, zeros=True)
                    add(p + f"0.mhsa.attentions.{h}.scale_distance.running_mean", 1, ones=True)
                    add(p + f"0.mhsa.attentions.{h}.scale_distance.items_so_far", 1, ones=True)
                    P[(p + f"0.mhsa.attentions.{h}.bias_scale").replace(".", "/")] = nn.Parameter(torch.rand(1))
                add(p + "0.mhsa.fc.weight", D, D); add(p + "0.mhsa.fc.bias", D, zeros=True)
            else:
                add(p + "0.mhsa.in_proj_weight", 3 * D, D); add(p + "0.mhsa.in_proj_bias", 3 * D, zeros=True)
                add(p + "0.mhsa.out_proj.weight", D, D); add(p + "0.mhsa.out_proj.bias", D, zeros=True)
            add(p + "1.0.weight", D, ones=True); add(p + "1.0.bias", D, zeros=True)
            add(p + "1.1.weight", dim_feedforward, D); add(p + "1.1.bias", dim_feedforward, zeros=True)
            add(p + "1.4.weight", D, dim_feedforward); add(p + "1.4.bias", D, zeros=True)
        add("transformer.norm.weight", D, ones=True); add("transformer.norm.bias", D, zeros=True)
        add("mlp_head.0.weight", dim_output, D); add("mlp_head.0.bias", dim_output, zeros=True)
        self._p = P
        self._packed = None

    # ---- reference-compatible state_dict ---------------------------------------------------------------------
    def state_dict(self, *a, **k):   # type: ignore[override]
        sd = {"class_token": self.class_token.detach()}
        sd.update({n.replace("/", "."): p.detach() for n, p in self._p.items()})
        return sd

    def load_state_dict(self, sd, strict: bool = True):   # type: ignore[override]
        own = self.state_dict()
        missing, unexpected = [k for k in own if k not in sd], [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state_dict mismatch: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for k, v in sd.items():
                if k == "class_token":
                    self.class_token.copy_(v)
                elif k in own:
                    self._p[k.replace(".", "/")].copy_(v)
        self._packed = None
        return self

    def _pack(self, dev):
        g = lambda n: self._p[n.replace(".", "/")].detach().to(dev, torch.float32).contiguous()  # noqa: E731
        h16 = lambda w: ops.cast_pad(_pad_rows(w, 128), (w.shape[1] + 63) // 64 * 64, torch.float16)  # noqa: E731
        pk = {"cls": self.class_token.detach().to(dev, torch.float32),
              "proj_w": h16(g("project_features.0.weight")), "proj_b": g("project_features.0.bias"), "layers": []}
        for l in range(self.n_layers):
            p = f"transformer.layers.{l}."
            if self.use_alibi:      # per-head encoders are just a row-blocked in_proj: [q heads | k heads | v heads]
                H = self.n_heads
                cat = lambda enc, what: torch.cat([g(p + f"0.mhsa.{enc}.{h}.{what}") for h in range(H)])  # noqa: E731
                w_in = torch.cat([cat(e, "weight") for e in ("query_encoders", "key_encoders", "value_encoders")])
                b_in = torch.cat([cat(e, "bias") for e in ("query_encoders", "key_encoders", "value_encoders")])
                w_out, b_out = g(p + "0.mhsa.fc.weight"), g(p + "0.mhsa.fc.bias")
                scale = torch.cat([g(p + f"0.mhsa.attentions.{h}.bias_scale") / g(p + f"0.mhsa.attentions.{h}.scale_distance.running_mean")
                                   for h in range(H)]).contiguous()
            else:
                w_in, b_in = g(p + "0.mhsa.in_proj_weight"), g(p + "0.mhsa.in_proj_bias")
                w_out, b_out = g(p + "0.mhsa.out_proj.weight"), g(p + "0.mhsa.out_proj.bias")
                scale = None
            # ALiBi: the attention output is bf16 (range), so its output projection runs on bf16 operands
            w_out_p = ops.cast_pad(_pad_rows(w_out, 128), w_out.shape[1], torch.bfloat16) if self.use_alibi else h16(w_out)
            pk["layers"].append(dict(
                ln1=(g(p + "0.norm.weight"), g(p + "0.norm.bias")),
                qkv=(h16(w_in), b_in), out=(w_out_p, b_out), alibi_scale=scale,
                ln2=(g(p + "1.0.weight"), g(p + "1.0.bias")),
                fc1=(h16(g(p + "1.1.weight")), g(p + "1.1.bias")),
                fc2=(h16(g(p + "1.4.weight")), g(p + "1.4.bias"))))
        pk["norm"] = (g("transformer.norm.weight"), g("transformer.norm.bias"))
        hb = g("mlp_head.0.bias")
        pk["head"] = (h16(g("mlp_head.0.weight")), torch.cat([hb, hb.new_zeros((-hb.shape[0]) % 128)]))
        return pk

    def forward(self, bags: torch.Tensor, *, coords: torch.Tensor | None = None, mask: torch.Tensor | None = None):
        if mask is not None:
            raise NotImplementedError("mask != None is not on the HIP path (the reference's Lit wrappers always pass None)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise NotImplementedError("training (backward) through the HIP MIL head is not implemented; call under "
                                      "torch.no_grad() / .eval() for deploy-time forward")
        if not bags.is_cuda:
            raise RuntimeError("HIP MIL head needs bags on the GPU (no CPU fallback)")
        Bb, T, F = bags.shape
        if F != self.dim_input:
            raise ValueError(f"bags have {F} features, model expects {self.dim_input}")
        dev, D, H = bags.device, self.dim_model, self.n_heads
        if self._packed is None or self._packed["cls"].device != dev:
            self._packed = self._pack(dev)
        pk = self._packed
        Kp = (F + 63) // 64 * 64
        a = bags.reshape(Bb * T, F)
        a = a.contiguous() if (a.dtype == torch.float16 and Kp == F) else ops.cast_pad(a.float(), Kp, torch.float16)
        proj = ops.gemm(a, pk["proj_w"], _lib.EPI_BIAS_GELU_F32, bias=pk["proj_b"])          # [Bb*T, D] fp32
        S = T + 1
        x = torch.empty(Bb, S, D, dtype=torch.float32, device=dev)
        x[:, 0] = pk["cls"]                                                                  # vision_tranformer.py:347-348
        x[:, 1:] = proj.view(Bb, T, D)
        x = x.view(Bb * S, D)
        if self.use_alibi:
            if coords is None:
                raise ValueError("use_alibi=True needs coords")
            c = torch.cat([coords.new_zeros(Bb, 1, 2), coords], dim=1).to(dev, torch.float32).contiguous()   # :349-351
        for L in pk["layers"]:
            h = ops.layernorm(x, *L["ln1"], 1e-5, torch.float16)
            qkv = ops.gemm(h, L["qkv"][0], _lib.EPI_BIAS, bias=L["qkv"][1])
            att = ops.attention_alibi(qkv, c, L["alibi_scale"], Bb, S, H) if self.use_alibi else ops.attention(qkv, Bb, S, H)
            ops.gemm(att, L["out"][0], _lib.EPI_RESIDUAL, bias=L["out"][1], out=x)           # x = attn(x) + x   (:291-292)
            h = ops.layernorm(x, *L["ln2"], 1e-5, torch.float16)
            u = ops.gemm(h, L["fc1"][0], _lib.EPI_BIAS_GELU, bias=L["fc1"][1])
            ops.gemm(u, L["fc2"][0], _lib.EPI_RESIDUAL, bias=L["fc2"][1], out=x)             # x = ff(x) + x     (:293)
        cls = ops.layernorm_rows(x, Bb, D, S * D, *pk["norm"], 1e-5, torch.float16)          # final LN, class token only
        logits = ops.gemm(cls, pk["head"][0], _lib.EPI_BIAS_F32, bias=pk["head"][1])
        return logits[:, : self.dim_output].contiguous()


# ---- bag building (reference src/stamp/modeling/data.py:811-862) ------------------------------------------------
def fixed_size_bag_indices(n_tiles: int, bag_size: int, deterministic: bool = False,
                           generator: torch.Generator | None = None) -> torch.Tensor:
    """The reference's sampling rule; drawn on the host exactly like the reference (torch.randperm on its CPU RNG)."""
    if n_tiles <= bag_size:
        return torch.arange(n_tiles)
    if deterministic:
        return torch.linspace(0, n_tiles - 1, steps=bag_size).round().long()
    return torch.randperm(n_tiles, generator=generator)[:bag_size]


def to_fixed_size_bag(feats: torch.Tensor, coords: torch.Tensor, bag_size: int, deterministic: bool = False,
                      generator: torch.Generator | None = None):
    """feats [N,F] f16/f32 + coords [N,2] f32 on the GPU -> (bag f32 [bag_size,F], coords [bag_size,2], n) via the HIP gather."""
    idx = fixed_size_bag_indices(feats.shape[0], bag_size, deterministic, generator).to(feats.device)
    bag = ops.gather_rows(feats.contiguous(), idx, bag_size, torch.float32)       # ".float()" of data.py:617 fused in
    c = ops.gather_rows(coords.contiguous().float(), idx, bag_size, torch.float32)
    return bag, c, min(bag_size, feats.shape[0])


def vary_precision(data: torch.Tensor, *, min_fraction_bits: int, generator: torch.Generator | None = None) -> torch.Tensor:
    """reference src/stamp/modeling/transforms.py:5-29; the shift draw stays torch.randint on the host RNG."""
    if min_fraction_bits < 1:
        raise ValueError("min_fraction bits has to be at least 1")
    frac = {torch.float32: 23, torch.float16: 10, torch.bfloat16: 7}.get(data.dtype)
    if frac is None:
        raise NotImplementedError(f"precision variation not implemented for {data.dtype}")
    shifts = torch.randint(0, frac - min_fraction_bits, data.shape, generator=generator).to(torch.uint8).to(data.device)
    return ops.vary_precision(data.contiguous(), shifts)


class MLP(nn.Module):
    """reference src/stamp/modeling/models/mlp.py:6-44 (state_dict keys `mlp.{0,3,...}`); eval forward on the HIP path."""

    def __init__(self, dim_input: int, dim_hidden: int, dim_output: int, num_layers: int, dropout: float):
        super().__init__()
        layers, d = [], dim_input
        for _ in range(num_layers - 1):
            layers += [nn.Linear(d, dim_hidden), nn.ReLU(), nn.Dropout(dropout)]
            d = dim_hidden
        layers.append(nn.Linear(d, dim_output))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("training through the HIP MLP head is not implemented")
        if x.ndim == 3:
            x = ops.mean_pool(x.contiguous())
        elif x.ndim != 2:
            raise ValueError(f"Expected 2D or 3D input, got {x.shape}")
        x = x.float().contiguous()
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        for i, m in enumerate(lin):
            x = ops.linear_f32(x, m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous(), relu=i < len(lin) - 1)
        return x


class Linear(nn.Module):
    """reference src/stamp/modeling/models/mlp.py:47-62."""

    def __init__(self, dim_input: int, dim_output: int):
        super().__init__()
        self.fc = nn.Linear(dim_input, dim_output)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if x.ndim == 3:
            x = ops.mean_pool(x.contiguous())
        elif x.ndim != 2:
            raise ValueError(f"Expected 2D or 3D input, got {x.shape}")
        return ops.linear_f32(x.float().contiguous(), self.fc.weight.detach().float().contiguous(),
                              self.fc.bias.detach().float().contiguous())


# ---- TransMIL (reference src/stamp/modeling/models/trans_mil.py:286-326) ------------------------------------------------
class _NysAttnParams(nn.Module):
    """Parameter container with the reference NystromAttention's names (to_qkv, to_out.0, res_conv)."""

    def __init__(self, dim: int, heads: int = 8, conv_k: int = 33):
        super().__init__()
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Dropout(0.1))
        self.res_conv = nn.Conv2d(heads, heads, (conv_k, 1), padding=(conv_k // 2, 0), groups=heads, bias=False)


class _TransLayerParams(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.attn = _NysAttnParams(dim)


class _PPEGParams(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 7, 1, 3, groups=dim)
        self.proj1 = nn.Conv2d(dim, dim, 5, 1, 2, groups=dim)
        self.proj2 = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)


def _bgemm(A, lda, sAo, sAi, B, ldb, sBo, sBi, transb, Cm, ldc, sCo, sCi, outer, inner, M, N, K, alpha=1.0, diag=0.0,
           bias=None, accumulate=False):
    _lib.check(_lib.lib().amds_bgemm_f32(A, lda, sAo, sAi, B, ldb, sBo, sBi, 1 if transb else 0, Cm, ldc, sCo, sCi, outer, inner,
                                         M, N, K, alpha, diag, bias, 1 if accumulate else 0,
                                         torch.cuda.current_stream().cuda_stream), "bgemm_f32")


class TransMIL(nn.Module):
    """Same constructor and state_dict keys as the reference TransMIL; eval-mode forward on the HIP path (fp32)."""

    def __init__(self, dim_output: int, dim_input: int, dim_hidden: int):
        super().__init__()
        if dim_hidden % 8 or dim_hidden % 4:
            raise ValueError("dim_hidden must be a multiple of 8")
        self.pos_layer = _PPEGParams(dim_hidden)
        self._fc1 = nn.Sequential(nn.Linear(dim_input, dim_hidden), nn.ReLU())
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim_hidden))
        self.n_classes = dim_output
        self.layer1 = _TransLayerParams(dim_hidden)
        self.layer2 = _TransLayerParams(dim_hidden)
        self.norm = nn.LayerNorm(dim_hidden)
        self._fc2 = nn.Linear(dim_hidden, dim_output)
        self.dim_hidden = dim_hidden

    @staticmethod
    def _f(t):
        return t.detach().float().contiguous()

    def _nystrom(self, y: torch.Tensor, layer: _TransLayerParams, x_res: torch.Tensor) -> None:
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

    def forward(self, h: torch.Tensor, **kwargs) -> torch.Tensor:
        import math

        if self.training and torch.is_grad_enabled():
            raise NotImplementedError("training through the HIP TransMIL head is not implemented")
        if not h.is_cuda:
            raise RuntimeError("HIP TransMIL needs bags on the GPU (no CPU fallback)")
        Bb, T, F = h.shape
        if T < 1:
            raise ValueError("empty bag")
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
            self._nystrom(y, layer, x)                                                           # x += attn(norm(x))
        cls = ops.layernorm_rows(x.view(-1), Bb, Cd, n * Cd, self._f(self.norm.weight), self._f(self.norm.bias), 1e-5, torch.float32)
        return ops.linear_f32(cls, self._f(self._fc2.weight), self._f(self._fc2.bias))
