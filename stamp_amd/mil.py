"""MIL heads on the HIP path: model classes with the reference's constructors, parameter trees (state_dict keys) and
``forward`` signatures, whose arithmetic runs in libamdstamp.

`VisionTransformer` mirrors the reference's MIL model class of the same name
(src/stamp/modeling/models/vision_tranformer.py:298-384): same keyword-only constructor, the same module tree -- hence the same
state_dict keys WITH prefix / nesting semantics (`class_token`, `project_features.0.*`,
`transformer.layers.{l}.0.{norm,mhsa.in_proj_*,mhsa.out_proj.*}` or the MultiHeadALiBi tree with its `running_mean` /
`items_so_far` BUFFERS, `transformer.layers.{l}.1.{0,1,4}.*`, `transformer.norm.*`, `mlp_head.0.*`) -- so Lightning checkpoints
of the reference's `Lit*` wrappers load into it and vice versa; ``forward(bags, *, coords, mask)`` is pinned by the reference's
tests/test_model.py:28-32.  The submodules are parameter containers only; ``forward`` never calls them.

The forward is DIFFERENTIABLE: one torch.autograd.Function over (bags, parameters) whose forward / backward are the HIP
training kernels (stamp_amd/mil_core.py), so ``loss.backward()`` fills ``.grad`` of the nn.Parameters and any torch optimiser /
Lightning drives it (reference models/__init__.py:133-141, 239-279), and ``torch.func.jacrev`` w.r.t. the bag works
(heatmaps/__init__.py:36-56).  Modes:
  * ``.eval()`` under ``torch.no_grad()`` / ``inference_mode()``: inference kernels, fp16 MFMA operands, any bag length, ``mask``
    supported (restated literally from :355-381).
  * gradient needed, or ``.train()``: training kernels, bf16 operands, fp32 accumulate / residual stream / statistics.  In
    ``.train()`` mode the reference's dropout sites are live (project_features and nn.MultiheadAttention: `dropout`; both
    feed-forward Dropouts: the hard-coded 0.5 of :160) with counter-based masks seeded from torch's CPU generator, and every
    ALiBi `_RunningMeanScaler` buffer is updated before use (:24-29).  ``mask`` is not available on this path.
There is no CPU / torch fallback: CPU tensors raise.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib, mil_core, ops
from .mil_core import PackedVit, VitDims


# ---- parameter containers with the reference's names (never called) ----------------------------------------------------------------
class _RunningMeanScalerParams(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.register_buffer("running_mean", torch.ones(1))
        self.register_buffer("items_so_far", torch.ones(1))


class _ALiBiParams(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.scale_distance = _RunningMeanScalerParams()
        self.bias_scale = nn.Parameter(torch.rand(1))


class _MultiHeadALiBiParams(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int) -> None:
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError(f"{embed_dim=} has to be divisible by {num_heads=}")
        hd = embed_dim // num_heads
        self.query_encoders = nn.ModuleList([nn.Linear(embed_dim, hd) for _ in range(num_heads)])
        self.key_encoders = nn.ModuleList([nn.Linear(embed_dim, hd) for _ in range(num_heads)])
        self.value_encoders = nn.ModuleList([nn.Linear(embed_dim, hd) for _ in range(num_heads)])
        self.attentions = nn.ModuleList([_ALiBiParams() for _ in range(num_heads)])
        self.fc = nn.Linear(embed_dim, embed_dim)


class _SelfAttentionParams(nn.Module):
    def __init__(self, dim: int, num_heads: int, dropout: float, use_alibi: bool) -> None:
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.mhsa = _MultiHeadALiBiParams(dim, num_heads) if use_alibi else nn.MultiheadAttention(dim, num_heads, dropout, batch_first=True)


class _TransformerParams(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, mlp_dim: int, dropout: float, use_alibi: bool) -> None:
        super().__init__()
        ff = lambda: nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, mlp_dim), nn.GELU(), nn.Dropout(0.5), nn.Linear(mlp_dim, dim), nn.Dropout(0.5))  # noqa: E731
        self.layers = nn.ModuleList([nn.ModuleList([_SelfAttentionParams(dim, heads, dropout, use_alibi), ff()]) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)


class _Saved:
    """Carrier of the forward's saved activations between the two autograd Functions (a non-tensor input)."""
    saved = None
    pk = None
    shapes = None


def _unwrap_batched(t: torch.Tensor):
    """(plain tensor, batch dim or None, vmap level) of a functorch BatchedTensor; plain tensors pass through."""
    F_ = torch._C._functorch
    if F_.is_batchedtensor(t):
        return F_.get_unwrapped(t), F_.maybe_get_bdim(t), F_.maybe_get_level(t)
    return t, None, None


class _MilVitBackward(torch.autograd.Function):
    """The HIP backward as its own Function so that torch.func.jacrev (= vmap over the backward, heatmaps/__init__.py:45-52) has a
    vmap rule to call: one HIP backward per basis vector of the logits."""

    @staticmethod
    def forward(dlogits, holder, need_params, need_bags, names):
        sc = getattr(holder, "loss_scale", 1.0)        # fp16 operands ("high"): the 16-bit gradient tensors carry a static power-of-two scale (mil_train.py)
        G, dbags = mil_core.backward(holder.pk, holder.saved, dlogits * sc if sc != 1.0 else dlogits, need_params=need_params, need_bags=need_bags)
        if sc != 1.0:
            dbags = dbags * (1.0 / sc) if need_bags else dbags
            for k in (list(G) if need_params else ()):         # in place: the mapping mil_core.backward returned is kept as it is
                G[k] = G[k] * (1.0 / sc)
        outs = [dbags if need_bags else dlogits.new_zeros(())]
        outs += [G[n].contiguous() for n in names] if need_params else []
        return tuple(outs)

    @staticmethod
    def setup_context(ctx, inputs, output):
        pass

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("double backward through the HIP MIL head is not implemented")

    @staticmethod
    def vmap(info, in_dims, dlogits, holder, need_params, need_bags, names):
        bd = in_dims[0]
        if bd is None:
            outs = _MilVitBackward.apply(dlogits, holder, need_params, need_bags, names)
            return outs, tuple(None for _ in outs)
        rows = [_MilVitBackward.apply(dlogits.select(bd, i), holder, need_params, need_bags, names) for i in range(info.batch_size)]
        outs = tuple(torch.stack([r[j] for r in rows]) for j in range(len(rows[0])))
        return outs, tuple(0 for _ in outs)


class _MilVitFunction(torch.autograd.Function):
    @staticmethod
    def forward(bags, coords, holder, model, training, seed, *params):
        names = model._param_names
        P = dict(zip(names, params))
        dev = bags.device

        def get(name):
            t = P[name] if name in P else model.get_buffer(name)
            return t.detach().to(dev, torch.float32)

        # operand type by torch's own flag, as HipMilVitTrainer: "medium" -> bf16; "high" (the reference's training setting, train.py:519) / "highest" -> fp16
        act = torch.bfloat16 if torch.get_float32_matmul_precision() == "medium" else torch.float16
        pk = PackedVit(model.dims, get, act, train=True)
        logits, saved = mil_core.forward_train(pk, bags.detach(), None if coords is None else coords.detach(), training=training, seed=seed)
        holder.pk, holder.saved, holder.loss_scale = pk, saved, (1.0 if act == torch.bfloat16 else 1024.0)
        return logits

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.holder, ctx.names = inputs[2], inputs[3]._param_names
        ctx.bags_dtype = inputs[0].dtype

    @staticmethod
    def backward(ctx, dlogits):
        need_bags = ctx.needs_input_grad[0]
        need_params = any(ctx.needs_input_grad[6:])
        outs = _MilVitBackward.apply(dlogits, ctx.holder, need_params, need_bags, tuple(ctx.names))
        dbags = outs[0].to(ctx.bags_dtype) if need_bags else None
        gp = list(outs[1:]) if need_params else [None] * len(ctx.names)
        return (dbags, None, None, None, None, None, *gp)


class VisionTransformer(nn.Module):
    def __init__(self, *, dim_output: int, dim_input: int, dim_model: int, n_layers: int, n_heads: int,
                 dim_feedforward: int, dropout: float, use_alibi: bool) -> None:
        super().__init__()
        self.dims = VitDims(F=dim_input, D=dim_model, H=n_heads, FF=dim_feedforward, C=dim_output, L=n_layers, alibi=bool(use_alibi),
                            p_drop=float(dropout))
        self.use_alibi = bool(use_alibi)
        self.dim_output, self.dim_input, self.dim_model = dim_output, dim_input, dim_model
        self.n_layers, self.n_heads, self.dim_feedforward, self.dropout = n_layers, n_heads, dim_feedforward, float(dropout)
        # the reference's module tree (vision_tranformer.py:312-329): parameter containers, initialised like the reference's
        self.class_token = nn.Parameter(torch.randn(dim_model))
        self.project_features = nn.Sequential(nn.Linear(dim_input, dim_model, bias=True), nn.GELU(), nn.Dropout(dropout))
        self.transformer = _TransformerParams(dim_model, n_layers, n_heads, dim_feedforward, dropout, use_alibi)
        self.mlp_head = nn.Sequential(nn.Linear(dim_model, dim_output))
        self._param_names = [n for n, _ in self.named_parameters()]
        self._packed: PackedVit | None = None
        self._packed_key = None

    # ---- inference weights, re-packed only when a parameter / buffer changed ---------------------------------------------------------
    def _infer_pack(self, dev) -> PackedVit:
        tensors = dict(self.named_parameters())
        tensors.update(dict(self.named_buffers()))
        key = (str(dev), tuple((t.data_ptr(), t._version) for t in tensors.values()))
        if self._packed is None or self._packed_key != key:
            self._packed = PackedVit(self.dims, lambda n: tensors[n].detach().to(dev, torch.float32), torch.float16, train=False)
            self._packed_key = key
        return self._packed

    def forward(self, bags: torch.Tensor, *, coords: torch.Tensor | None = None, mask: torch.Tensor | None = None):
        if not bags.is_cuda:
            raise RuntimeError("HIP MIL head needs bags on the GPU (no CPU fallback)")
        if bags.dim() != 3 or bags.shape[-1] != self.dim_input:
            raise ValueError(f"bags must be [batch, tile, {self.dim_input}], got {tuple(bags.shape)}")
        need_grad = torch.is_grad_enabled() and (bags.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not (self.training or need_grad):
            return mil_core.forward_infer(self._infer_pack(bags.device), bags, coords, mask)
        if mask is not None:
            raise NotImplementedError("mask != None is available on the inference path only (eval + no_grad); the reference's "
                                      "Lightning wrappers never pass one (models/__init__.py:286-313)")
        seed = 0
        if self.training:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())            # torch's CPU generator: reproducible under torch.manual_seed
            if self.use_alibi:
                if coords is None:
                    raise ValueError("use_alibi=True needs coords")
                bufs = dict(self.named_buffers())
                Bb = bags.shape[0]
                cc = mil_core._coords_with_cls(coords.detach(), Bb, bags.device)
                mil_core.update_running_means(lambda n: bufs[n], self.dims, cc)
        holder = _Saved()
        params = [p for _, p in self.named_parameters()]
        if need_grad:
            return _MilVitFunction.apply(bags, coords, holder, self, self.training, seed, *params)
        with torch.no_grad():
            return _MilVitFunction.forward(bags, coords, holder, self, self.training, seed, *params)


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


# ---- MLP / Linear heads (reference src/stamp/modeling/models/mlp.py) -- forward AND backward on the HIP path ----------------------------
# These are the slide / patient-level models (`LitSlide*` / `LitPatient*`, models/__init__.py:778-937): they consume the table of slide
# embeddings the all-gather collates (stamp_amd.distributed).  Everything is fp32 like the reference: products on the exact-fp32 MFMA
# (amds_linear_f32 forward; amds_bgemm_f32 for dx = dz W and dW = dz^T x), bias gradients by the deterministic column sum, ReLU and
# Dropout backward in the library (amds_relu_bwd; the counter-based mask regenerated from its seed, never stored).
class _LinearF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu: bool):
        x, w, b = x.detach().float().contiguous(), w.detach().float().contiguous(), b.detach().float().contiguous()
        y = ops.linear_f32(x, w, b, relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        from . import train_ops as T
        from .transmil_core import _mm
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous().float()
        if ctx.relu:
            dz = torch.empty_like(dy)
            _lib.check(_lib.lib().amds_relu_bwd(y.data_ptr(), dy.data_ptr(), dz.data_ptr(), dz.numel(), ops._stream()), "relu_bwd")
        else:
            dz = dy
        dx = _mm(dz[None], w[None], False)[0] if ctx.needs_input_grad[0] else None              # [M, N] [N, K]
        dw = _mm(dz[None], x[None], False, transa=True)[0] if ctx.needs_input_grad[1] else None    # (dz^T) x, no explicit transpose
        db = T.colsum(dz) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


class _DropoutF32Fn(torch.autograd.Function):
    """nn.Dropout(p) in train mode on an fp32 [rows, cols] tensor: keep mask = the library's counter-based function of (seed, site, element);
    the backward applies the same mask to the gradient."""

    @staticmethod
    def _apply(x, p, seed, site):
        from . import _lib
        y = torch.empty_like(x)
        rows, cols = x.shape
        _lib.check(_lib.lib().amds_dropout_cast_bwd(x.data_ptr(), cols, y.data_ptr(), cols, rows, cols, _lib.F32, float(p), int(seed), int(site),
                                                    ops._stream()), "dropout_f32")
        return y

    @staticmethod
    def forward(ctx, x, p: float, seed: int, site: int):
        ctx.args = (p, seed, site)
        return _DropoutF32Fn._apply(x.contiguous().float(), p, seed, site)

    @staticmethod
    def backward(ctx, dy):
        return _DropoutF32Fn._apply(dy.contiguous().float(), *ctx.args), None, None, None


class _MeanPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = tuple(x.shape)
        return ops.mean_pool(x.detach().contiguous())

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        B, Tn, Fd = ctx.shape
        dx = torch.empty(B, Tn, Fd, dtype=torch.float32, device=dy.device)
        dy = dy.contiguous().float()
        _lib.check(_lib.lib().amds_mean_pool_bwd(dy.data_ptr(), dx.data_ptr(), B, Tn, Fd, ops._stream()), "mean_pool_bwd")
        return dx


def _pool_if_bag(x: torch.Tensor) -> torch.Tensor:
    if x.ndim == 3:
        return _MeanPoolFn.apply(x)
    if x.ndim != 2:
        raise ValueError(f"Expected 2D or 3D input, got {x.shape}")
    return x


class MLP(nn.Module):
    """reference src/stamp/modeling/models/mlp.py:6-44 (state_dict keys `mlp.{0,3,...}`): mean over tiles if given bags, then
    (Linear -> ReLU -> Dropout) x (num_layers - 1) -> Linear.  Differentiable: `loss.backward()` fills `.grad` of every parameter (and of the
    input if it requires grad) from the library's backward kernels, so any torch optimiser / Lightning trains it.  Train mode draws one
    dropout seed per forward from torch's CPU generator (`dropout_seed` pins it)."""

    def __init__(self, dim_input: int, dim_hidden: int, dim_output: int, num_layers: int, dropout: float):
        super().__init__()
        layers, d = [], dim_input
        for _ in range(num_layers - 1):
            layers += [nn.Linear(d, dim_hidden), nn.ReLU(), nn.Dropout(dropout)]
            d = dim_hidden
        layers.append(nn.Linear(d, dim_output))
        self.mlp = nn.Sequential(*layers)
        self.dropout_p = float(dropout)
        self.dropout_seed: int | None = None

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("stamp_amd.mil.MLP runs on the GPU only (no CPU fallback)")
        x = _pool_if_bag(x).float()
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        drop = self.training and self.dropout_p > 0.0
        seed = (self.dropout_seed if self.dropout_seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())) if drop else 0
        for i, m in enumerate(lin):
            hidden = i < len(lin) - 1
            x = _LinearF32Fn.apply(x, m.weight, m.bias, hidden)
            if hidden and drop:
                x = _DropoutF32Fn.apply(x, self.dropout_p, seed, i)
        return x

    def dropout_masks(self, rows: int, seed: int) -> list[torch.Tensor]:
        """The keep masks (bool [rows, dim_hidden]) the train-mode forward draws for `seed`, one per hidden layer (for parity tests)."""
        from . import train_ops as T
        lin = [m for m in self.mlp if isinstance(m, nn.Linear)]
        dev = lin[0].weight.device
        return [T.dropout_mask(rows * m.out_features, self.dropout_p, seed, i, dev).view(rows, m.out_features).bool() for i, m in enumerate(lin[:-1])]


class Linear(nn.Module):
    """reference src/stamp/modeling/models/mlp.py:47-62; differentiable like `MLP`."""

    def __init__(self, dim_input: int, dim_output: int):
        super().__init__()
        self.fc = nn.Linear(dim_input, dim_output)

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("stamp_amd.mil.Linear runs on the GPU only (no CPU fallback)")
        return _LinearF32Fn.apply(_pool_if_bag(x).float(), self.fc.weight, self.fc.bias, False)


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
    ops.sync_float32_matmul_precision()
    _lib.check(_lib.lib().amds_bgemm_f32(A, lda, sAo, sAi, B, ldb, sBo, sBi, 1 if transb else 0, Cm, ldc, sCo, sCi, outer, inner,
                                         M, N, K, alpha, diag, bias, 1 if accumulate else 0,
                                         torch.cuda.current_stream().cuda_stream), "bgemm_f32")


class TransMIL(nn.Module):
    """Same constructor and state_dict keys as the reference TransMIL (trans_mil.py:286-326); fp32 on the HIP path.  Inference forward
    below; training (forward with saved intermediates + hand-derived backward) in stamp_amd/transmil_core.py, reached through one
    torch.autograd.Function so that torch optimisers / Lightning drive it like the reference's module."""

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


    def _get(self, dev):
        tensors = dict(self.named_parameters())
        return lambda n: tensors[n].detach().to(dev, torch.float32).contiguous()

    def forward(self, h: torch.Tensor, **kwargs) -> torch.Tensor:
        """``forward(bags, coords=..., mask=...)`` like the reference (:299-303: coords and mask are accepted and ignored).
        eval + no_grad: the inference path below.  Gradient needed or ``.train()``: the training kernels (transmil_core.py) behind a
        torch.autograd.Function, with the reference's one dropout site (Dropout(0.1) on `to_out`) live in train mode."""
        import math

        if not h.is_cuda:
            raise RuntimeError("HIP TransMIL needs bags on the GPU (no CPU fallback)")
        Bb, T, F = h.shape
        if T < 1:
            raise ValueError("empty bag")
        need_grad = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.parameters()))
        if self.training or need_grad:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0
            holder = _Saved()
            params = [p for _, p in self.named_parameters()]
            if need_grad:
                return _TransMilFunction.apply(h, holder, self, self.training, seed, *params)
            with torch.no_grad():
                return _TransMilFunction.forward(h, holder, self, self.training, seed, *params)
        return self._forward_c(h)

    # ---- deploy / validation forward: ONE library call (amds_transmil_forward, csrc/transmil_fwd.hip) -----------------------------------
    def _c_weights(self, dev):
        """amds_transmil_weights over fp32 device copies of the parameters, rebuilt when a parameter changed (version counters)."""
        tensors = dict(self.named_parameters())
        key = (str(dev), tuple((t.data_ptr(), t._version) for t in tensors.values()))
        if getattr(self, "_cw_key", None) != key:
            g = lambda n: tensors[n].detach().to(dev, torch.float32).contiguous()  # noqa: E731
            keep = {}

            def ptr(name, shape=None):
                t = g(name)
                if shape is not None:
                    t = t.reshape(shape).contiguous()
                keep[name] = t
                return t.data_ptr()

            Cd = self.dim_hidden
            w = _lib.TransMilWeights()
            w.fc1_w, w.fc1_b, w.cls_token = ptr("_fc1.0.weight"), ptr("_fc1.0.bias"), ptr("cls_token", (Cd,))
            for i, nm in enumerate(("layer1", "layer2")):
                w.layer[i] = _lib.TransMilLayer(ptr(f"{nm}.norm.weight"), ptr(f"{nm}.norm.bias"), ptr(f"{nm}.attn.to_qkv.weight"), ptr(f"{nm}.attn.to_out.0.weight"),
                                                ptr(f"{nm}.attn.to_out.0.bias"), ptr(f"{nm}.attn.res_conv.weight", (8, -1)))
            w.ppeg_w7, w.ppeg_b7 = ptr("pos_layer.proj.weight", (Cd, -1)), ptr("pos_layer.proj.bias")
            w.ppeg_w5, w.ppeg_b5 = ptr("pos_layer.proj1.weight", (Cd, -1)), ptr("pos_layer.proj1.bias")
            w.ppeg_w3, w.ppeg_b3 = ptr("pos_layer.proj2.weight", (Cd, -1)), ptr("pos_layer.proj2.bias")
            w.norm_w, w.norm_b, w.fc2_w, w.fc2_b = ptr("norm.weight"), ptr("norm.bias"), ptr("_fc2.weight"), ptr("_fc2.bias")
            self._cw, self._cw_keep, self._cw_key = w, keep, key
        return self._cw

    def _forward_c(self, h: torch.Tensor) -> torch.Tensor:
        import ctypes as C
        Bb, T, F = h.shape
        dev = h.device
        if h.dtype not in ops._DT:
            h = h.float()
        h = h.contiguous()
        cfg = _lib.TransMilCfg(F, self.dim_hidden, self.n_classes)
        if self._fc1[0].in_features != F:
            raise ValueError(f"bags must be [batch, tile, {self._fc1[0].in_features}], got {tuple(h.shape)}")
        w = self._c_weights(dev)
        lib = _lib.lib()
        need = lib.amds_transmil_workspace_bytes(C.byref(cfg), Bb, T)
        if need == 0:
            _lib.check(-1, "transmil_workspace_bytes")
        ws = ops.scratch("transmil", dev, need)
        logits = torch.empty(Bb, self.n_classes, dtype=torch.float32, device=dev)
        ops.sync_float32_matmul_precision()
        _lib.check(lib.amds_transmil_forward(C.byref(cfg), C.byref(w), h.data_ptr(), ops._DT[h.dtype], logits.data_ptr(), Bb, T, ws.data_ptr(), ws.numel(),
                                             torch.cuda.current_stream().cuda_stream), "transmil_forward")
        return logits


class _TransMilBackward(torch.autograd.Function):
    @staticmethod
    def forward(dlogits, holder, need_params, need_bags, names):
        from . import transmil_core
        G, dbags = transmil_core.backward(holder.saved, dlogits, need_params=need_params, need_bags=need_bags)
        outs = [dbags if need_bags else dlogits.new_zeros(())]
        outs += [G[n].reshape(holder.shapes[n]).contiguous() for n in names] if need_params else []
        return tuple(outs)

    @staticmethod
    def setup_context(ctx, inputs, output):
        pass

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("double backward through the HIP TransMIL head is not implemented")

    @staticmethod
    def vmap(info, in_dims, dlogits, holder, need_params, need_bags, names):
        bd = in_dims[0]
        if bd is None:
            outs = _TransMilBackward.apply(dlogits, holder, need_params, need_bags, names)
            return outs, tuple(None for _ in outs)
        rows = [_TransMilBackward.apply(dlogits.select(bd, i), holder, need_params, need_bags, names) for i in range(info.batch_size)]
        outs = tuple(torch.stack([r[j] for r in rows]) for j in range(len(rows[0])))
        return outs, tuple(0 for _ in outs)


class _TransMilFunction(torch.autograd.Function):
    @staticmethod
    def forward(bags, holder, model, training, seed, *params):
        from . import transmil_core
        names = [n for n, _ in model.named_parameters()]
        P = dict(zip(names, params))
        dev = bags.device
        get = lambda n: P[n].detach().to(dev, torch.float32).contiguous()  # noqa: E731
        logits, saved = transmil_core.forward_train(get, bags.detach(), (bags.shape[-1], model.dim_hidden, model.n_classes), training=training, seed=seed)
        holder.saved, holder.shapes = saved, {n: tuple(p.shape) for n, p in P.items()}
        return logits

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.holder, ctx.names = inputs[1], tuple(n for n, _ in inputs[2].named_parameters())
        ctx.bags_dtype = inputs[0].dtype

    @staticmethod
    def backward(ctx, dlogits):
        need_bags = ctx.needs_input_grad[0]
        need_params = any(ctx.needs_input_grad[5:])
        outs = _TransMilBackward.apply(dlogits, ctx.holder, need_params, need_bags, ctx.names)
        dbags = outs[0].to(ctx.bags_dtype) if need_bags else None
        gp = list(outs[1:]) if need_params else [None] * len(ctx.names)
        return (dbags, None, None, None, None, *gp)
