"""Training step of the MIL `vit` head on the HIP path: forward + backward + AdamW, no autograd through the network.

Mirrors what `stamp train` does per batch for tile-level classification -- reference
src/stamp/modeling/models/__init__.py:239-279 (`LitTileClassifier._step`: ``logits = self.model(bags, coords=coords,
mask=None)``, ``F.cross_entropy(logits, float one-hot targets, weight=class_weights)``) and :133-141 (AdamW with
torch defaults wrapped in OneCycleLR(max_lr, div_factor, total_steps)).  Scope: `use_alibi` False or True, `mask=None`,
dropout 0 (the reference's code default, src/stamp/modeling/config.py:92-100).  With ALiBi (MultiHeadALiBi,
vision_tranformer.py:77-154) the per-head q/k/v encoders are trained as one row-blocked in-projection, every head's
`_RunningMeanScaler` is updated from the batch's mean pairwise tile distance before it is used (train mode, :24-29), and
`bias_scale` receives its gradient from the distance term of the attention backward.

Mixed precision (stated, not hidden): bf16 MFMA operands for activations, weights and gradients (fp32 exponent range,
so no loss scaling), fp32 accumulation, fp32 residual stream and its gradient, fp32 LayerNorm / softmax statistics,
fp32 master weights, gradients and Adam moments.  The loss on the [batch, classes] logits is the one tiny piece left to
torch; everything with a token dimension runs in libamdstamp.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import _lib, ops
from . import train_ops as T
from .mil import VisionTransformer, _bgemm

BF = torch.bfloat16


class HipMilVitTrainer:
    def __init__(self, model: VisionTransformer, *, device="cuda", max_lr: float = 1e-4, div_factor: float = 25.0,
                 total_steps: int = 1000, weight_decay: float = 0.01, split_k: int = 32) -> None:
        self.alibi = bool(model.use_alibi)
        self.model = model
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("HipMilVitTrainer runs on the GPU only")
        self.D, self.H, self.FF, self.F, self.C, self.L = (model.dim_model, model.n_heads, model.dim_feedforward, model.dim_input,
                                                           model.dim_output, model.n_layers)
        if self.F % 256 or self.D % 256 or self.FF % 256:
            raise NotImplementedError("training needs dim_input, dim_model and dim_feedforward to be multiples of 256")
        self.split_k = split_k
        # flat fp32 master parameters (reference state_dict order), gradients and Adam moments
        sd = model.state_dict()
        self.names = list(sd.keys())
        self.shapes = {k: tuple(v.shape) for k, v in sd.items()}
        sizes = [int(v.numel()) for v in sd.values()]
        self.offs = {k: (sum(sizes[:i]), sizes[i]) for i, k in enumerate(self.names)}
        n = sum(sizes)
        self.P = torch.cat([v.detach().float().reshape(-1) for v in sd.values()]).to(self.dev).contiguous()
        self.G = torch.zeros(n, device=self.dev)
        self.m = torch.zeros(n, device=self.dev)
        self.v = torch.zeros(n, device=self.dev)
        self.step_count = 0
        self.wd = weight_decay
        # ALiBi: running_mean / items_so_far are buffers of the reference module (no gradient, no optimiser update)
        rm_names = [k for k in self.names if k.endswith("scale_distance.running_mean")]
        self._rm_idx = torch.tensor([self.offs[k][0] for k in rm_names], dtype=torch.long, device=self.dev)
        self._n_idx = torch.tensor([self.offs[k[: -len("running_mean")] + "items_so_far"][0] for k in rm_names], dtype=torch.long, device=self.dev)
        self._stat_idx = torch.cat([self._rm_idx, self._n_idx])
        # the reference's schedule, evaluated by torch itself on a dummy optimizer (host-side plumbing)
        dummy = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
        sched = torch.optim.lr_scheduler.OneCycleLR(dummy, total_steps=total_steps, max_lr=max_lr, div_factor=div_factor)
        self._lrs = []
        for _ in range(total_steps):
            self._lrs.append(dummy.param_groups[0]["lr"])
            dummy.step()
            sched.step() if len(self._lrs) < total_steps else None
        self._refresh_weights()

    # ---- parameter views ------------------------------------------------------------------------------------------------
    def p(self, name: str) -> torch.Tensor:
        o, n = self.offs[name]
        return self.P[o:o + n].view(self.shapes[name])

    def g(self, name: str) -> torch.Tensor:
        o, n = self.offs[name]
        return self.G[o:o + n].view(self.shapes[name])

    def _refresh_weights(self) -> None:
        """bf16 copies W [N][K] (forward, dgrad of the previous layer uses W^T) and W^T [K][N] of every Linear."""
        self.wb, self.wbt = {}, {}
        names = ["project_features.0.weight"]
        for l in range(self.L):
            p = f"transformer.layers.{l}."
            names += [p + self._in_w, p + self._out_w, p + "1.1.weight", p + "1.4.weight"]
        for nme in names:
            w = self._stacked_in_proj(nme)[0] if nme.endswith("0.mhsa.in_proj_weight") and self.alibi else self.p(nme)
            wb = ops.cast_pad(w, w.shape[1], BF)
            self.wb[nme] = wb
            self.wbt[nme] = T.transpose16(wb)          # [K][N]

    _ENCODERS = ("query_encoders", "key_encoders", "value_encoders")

    def _stacked_in_proj(self, name: str):
        """MultiHeadALiBi keeps one Linear(D, 64) per head for q, k and v (vision_tranformer.py:100-117): row-blocked they are
        the [3D, D] in-projection [q heads | k heads | v heads] the attention kernels consume (data movement only)."""
        p = name[: -len("in_proj_weight")]
        w = torch.cat([self.p(p + f"{e}.{h}.weight") for e in self._ENCODERS for h in range(self.H)])
        b = torch.cat([self.p(p + f"{e}.{h}.bias") for e in self._ENCODERS for h in range(self.H)])
        return w.contiguous(), b.contiguous()

    def _scatter_in_proj_grad(self, p: str, gw: torch.Tensor, gb: torch.Tensor) -> None:
        i = 0
        for e in self._ENCODERS:
            for h in range(self.H):
                self.g(p + f"{e}.{h}.weight").copy_(gw[i * 64:(i + 1) * 64])
                self.g(p + f"{e}.{h}.bias").copy_(gb[i * 64:(i + 1) * 64])
                i += 1

    @property
    def _in_w(self) -> str:
        return "0.mhsa.in_proj_weight"

    @property
    def _out_w(self) -> str:
        return "0.mhsa.fc.weight" if self.alibi else "0.mhsa.out_proj.weight"

    def sync_to_model(self) -> None:
        self.model.load_state_dict({k: self.p(k).detach().clone() for k in self.names})

    # ---- helpers ----------------------------------------------------------------------------------------------------------
    def _wgrad(self, dyT: torch.Tensor, xT: torch.Tensor, name: str, Mp: int, out: torch.Tensor | None = None) -> None:
        """G[name][N][K] = dy^T x, contraction over the (padded) token dimension split into split_k fp32 partials."""
        Nn, Kk = (out.shape if out is not None else self.shapes[name])
        S = self.split_k
        chunk = Mp // S
        part = torch.empty(S, Nn * Kk, dtype=torch.float32, device=self.dev)
        T.gemm_batched(dyT, Mp, chunk, xT, Mp, chunk, Nn, Kk, chunk, S, BF, part, Kk, Nn * Kk, True)
        T.colsum(part, out=(out if out is not None else self.g(name)).view(-1))

    def _pad_M(self, M: int) -> int:
        unit = 64 * self.split_k
        return (M + unit - 1) // unit * unit

    # ---- one optimisation step ------------------------------------------------------------------------------------------------
    def step(self, bags: torch.Tensor, targets: torch.Tensor, class_weights: torch.Tensor | None = None, *, update: bool = True,
             data_parallel: bool = False, coords: torch.Tensor | None = None, loss_fn=None):
        """bags [Bb,T,F] fp16/bf16/fp32 on the GPU, targets float one-hot [Bb,C]. Returns (loss, logits).

        loss_fn(logits, targets) -> scalar selects the task (stamp_amd.losses): default = the classifier's weighted
        cross-entropy; `losses.l1_loss` = LitTileRegressor (dim_output 1); `losses.cox_survival_loss` = LitTileSurvival
        (dim_output 1, targets [time, event]).

        data_parallel=True: every rank of the initialised process group holds a replica and its own bags; the flat
        fp32 gradient buffer (14.7 MB for the default head) is averaged with ONE RCCL all-reduce before AdamW
        (the reference itself is single-device, src/stamp/modeling/train.py:541-547)."""
        dev, D, H, FF, Fd, C = self.dev, self.D, self.H, self.FF, self.F, self.C
        Bb, Tn, _ = bags.shape
        S = Tn + 1
        Mt, M = Bb * Tn, Bb * S
        lib = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        # ---- forward -----------------------------------------------------------------------------------------------------------
        a = torch.empty(Mt, Fd, dtype=BF, device=dev)
        src = bags.reshape(Mt, Fd).contiguous()
        if src.dtype == torch.float16:
            _lib.check(lib.amds_convert_f16_bf16(src.data_ptr(), a.data_ptr(), src.numel(), st), "convert")
        elif src.dtype == BF:
            a = src
        else:
            a = ops.cast_pad(src.float(), Fd, BF)
        pn = "project_features.0."
        zp = ops.gemm(a, self.wb[pn + "weight"], _lib.EPI_BIAS, bias=self.p(pn + "bias"))                   # bf16 [Mt, D]
        xp = T.gelu_fwd(zp, torch.float32)
        x = torch.empty(Bb, S, D, dtype=torch.float32, device=dev)
        x[:, 0] = self.p("class_token")
        x[:, 1:] = xp.view(Bb, Tn, D)
        x = x.view(M, D)
        saved = []
        cc = None
        if self.alibi:
            if coords is None:
                raise ValueError("use_alibi=True needs coords")
            cc = torch.cat([coords.new_zeros(Bb, 1, 2), coords], dim=1).to(dev, torch.float32).contiguous()     # class token at (0, 0), :349-351
            # train-mode `_RunningMeanScaler` of every head and layer, before use (:24-29): rm <- rm + (mean(dist) - rm) / n ; n <- n + 1
            md = T.cdist_mean(cc)
            rm_i, n_i = self._rm_idx, self._n_idx
            self.P[rm_i] = self.P[rm_i] + (md - self.P[rm_i]) / self.P[n_i]
            self.P[n_i] = self.P[n_i] + 1.0
        for l in range(self.L):
            p = f"transformer.layers.{l}."
            h1, mu1, rs1 = T.layernorm_train(x, self.p(p + "0.norm.weight"), self.p(p + "0.norm.bias"), 1e-5, BF)
            x_mid = x.clone()
            if self.alibi:
                b_in = self._stacked_in_proj(p + self._in_w)[1]
                qkv = ops.gemm(h1, self.wb[p + self._in_w], _lib.EPI_BIAS, bias=b_in)
                bs = torch.cat([self.p(p + f"0.mhsa.attentions.{h}.bias_scale") for h in range(H)]).contiguous()
                inv_rm = (1.0 / torch.cat([self.p(p + f"0.mhsa.attentions.{h}.scale_distance.running_mean") for h in range(H)])).contiguous()
                att, u_al, osm, lse = T.attention_alibi_fwd_train(qkv, cc, inv_rm, bs, Bb, S, H)
                lse = (lse, u_al, osm, bs, inv_rm)
                ops.gemm(att, self.wb[p + self._out_w], _lib.EPI_RESIDUAL, bias=self.p(p + "0.mhsa.fc.bias"), out=x_mid)
            else:
                qkv = ops.gemm(h1, self.wb[p + "0.mhsa.in_proj_weight"], _lib.EPI_BIAS, bias=self.p(p + "0.mhsa.in_proj_bias"))
                att, lse = T.attention_fwd_lse(qkv, Bb, S, H)
                ops.gemm(att, self.wb[p + "0.mhsa.out_proj.weight"], _lib.EPI_RESIDUAL, bias=self.p(p + "0.mhsa.out_proj.bias"), out=x_mid)
            h2, mu2, rs2 = T.layernorm_train(x_mid, self.p(p + "1.0.weight"), self.p(p + "1.0.bias"), 1e-5, BF)
            z = ops.gemm(h2, self.wb[p + "1.1.weight"], _lib.EPI_BIAS, bias=self.p(p + "1.1.bias"))
            u = T.gelu_fwd(z)
            x_out = x_mid.clone()
            ops.gemm(u, self.wb[p + "1.4.weight"], _lib.EPI_RESIDUAL, bias=self.p(p + "1.4.bias"), out=x_out)
            saved.append((x, h1, mu1, rs1, qkv, att, lse, x_mid, h2, mu2, rs2, z, u))
            x = x_out
        clsn, muf, rsf = T.layernorm_train(x, self.p("transformer.norm.weight"), self.p("transformer.norm.bias"), 1e-5, torch.float32,
                                           rows=Bb, row_stride=S * D)
        logits = ops.linear_f32(clsn, self.p("mlp_head.0.weight").contiguous(), self.p("mlp_head.0.bias").contiguous())
        # ---- loss on [Bb, C]: the reference's weighted CE with float one-hot targets (models/__init__.py:254-258) ---------------------
        lg = logits.detach().clone().requires_grad_(True)
        if loss_fn is None:
            loss = F.cross_entropy(lg, targets.to(dev, torch.float32), weight=None if class_weights is None else class_weights.to(dev, torch.float32))
        else:
            loss = loss_fn(lg, targets.to(dev))
        loss.backward()
        dlogits = lg.grad.contiguous()
        if not update and not torch.is_grad_enabled():
            return loss.detach(), logits
        # ---- backward ----------------------------------------------------------------------------------------------------------
        Mp = self._pad_M(M)
        gW, gb = self.g("mlp_head.0.weight"), self.g("mlp_head.0.bias")
        dlT = dlogits.t().contiguous()                                                                     # [C, Bb] (data movement)
        _bgemm(dlT.data_ptr(), Bb, 0, 0, clsn.data_ptr(), D, 0, 0, False, gW.data_ptr(), D, 0, 0, 1, 1, C, D, Bb)   # dW_head = dlogits^T clsn
        T.colsum(dlogits, out=gb)
        dcls = torch.empty(Bb, D, dtype=torch.float32, device=dev)
        wh = self.p("mlp_head.0.weight").contiguous()
        _bgemm(dlogits.data_ptr(), C, 0, 0, wh.data_ptr(), D, 0, 0, False, dcls.data_ptr(), D, 0, 0, 1, 1, Bb, D, C)  # dclsn = dlogits W_head
        dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        T.layernorm_bwd(dcls, x, muf, rsf, self.p("transformer.norm.weight"), dx, False, self.g("transformer.norm.weight"),
                        self.g("transformer.norm.bias"), rows=Bb, dy_stride=D, x_stride=S * D, dx_stride=S * D)
        tbuf = {}

        def tr(t: torch.Tensor, key: str) -> torch.Tensor:      # [M, cols] bf16 -> [cols, Mp], zero-padded scratch reused per width
            cols = t.shape[1]
            k = (key, cols)
            if k not in tbuf:
                tbuf[k] = torch.zeros(cols, Mp, dtype=BF, device=dev)
            return T.transpose16(t, out=tbuf[k])

        for l in reversed(range(self.L)):
            p = f"transformer.layers.{l}."
            x_in, h1, mu1, rs1, qkv, att, lse, x_mid, h2, mu2, rs2, z, u = saved[l]
            dxb = ops.cast_pad(dx, D, BF)                                                                   # d(x_out) as a bf16 operand
            du = ops.gemm(dxb, self.wbt[p + "1.4.weight"], _lib.EPI_BIAS)                                   # [M, FF] = dx W2
            self._wgrad(tr(dxb, "g"), tr(u, "a"), p + "1.4.weight", Mp)
            T.colsum(dx, out=self.g(p + "1.4.bias"))
            dz = T.gelu_bwd(z, du)
            dh2 = ops.gemm(dz, self.wbt[p + "1.1.weight"], _lib.EPI_BIAS_F32)                               # [M, D] fp32
            self._wgrad(tr(dz, "g"), tr(h2, "a"), p + "1.1.weight", Mp)
            T.colsum(dz, out=self.g(p + "1.1.bias"))
            T.layernorm_bwd(dh2, x_mid, mu2, rs2, self.p(p + "1.0.weight"), dx, True, self.g(p + "1.0.weight"), self.g(p + "1.0.bias"))
            dxb = ops.cast_pad(dx, D, BF)                                                                   # d(x_mid)
            datt = ops.gemm(dxb, self.wbt[p + self._out_w], _lib.EPI_BIAS)
            self._wgrad(tr(dxb, "g"), tr(att, "a"), p + self._out_w, Mp)
            T.colsum(dx, out=self.g(p + ("0.mhsa.fc.bias" if self.alibi else "0.mhsa.out_proj.bias")))
            if self.alibi:
                lse_, u_al, osm, bs, inv_rm = lse
                dqkv, dbs = T.attention_alibi_bwd(qkv, osm, u_al, datt, lse_, cc, bs, (bs * inv_rm).contiguous(), Bb, S, H)
                for h in range(H):
                    self.g(p + f"0.mhsa.attentions.{h}.bias_scale").copy_(dbs[h:h + 1])
                gw = torch.empty(3 * D, D, dtype=torch.float32, device=dev)
                self._wgrad(tr(dqkv, "g"), tr(h1, "a"), "", Mp, out=gw)
                self._scatter_in_proj_grad(p + "0.mhsa.", gw, T.colsum(dqkv))
            else:
                dqkv = T.attention_bwd(qkv, att, datt, lse, Bb, S, H)
                self._wgrad(tr(dqkv, "g"), tr(h1, "a"), p + "0.mhsa.in_proj_weight", Mp)
                T.colsum(dqkv, out=self.g(p + "0.mhsa.in_proj_bias"))
            dh1 = ops.gemm(dqkv, self.wbt[p + "0.mhsa.in_proj_weight"], _lib.EPI_BIAS_F32)
            T.layernorm_bwd(dh1, x_in, mu1, rs1, self.p(p + "0.norm.weight"), dx, True, self.g(p + "0.norm.weight"), self.g(p + "0.norm.bias"))
        dx3 = dx.view(Bb, S, D)
        T.colsum(dx3[:, 0, :], out=self.g("class_token"))                                                   # rows at stride S*D
        dxp = dx3[:, 1:, :].reshape(Mt, D)                                                                  # contiguous copy (data movement)
        dzp = T.gelu_bwd(zp, dxp)                                                                           # bf16
        Mtp = self._pad_M(Mt)
        dzpT = T.transpose16(dzp, ld_dst=Mtp)
        aT = T.transpose16(a, ld_dst=Mtp)
        Nn, Kk = self.shapes[pn + "weight"]
        chunk = Mtp // self.split_k
        part = torch.empty(self.split_k, Nn * Kk, dtype=torch.float32, device=dev)
        T.gemm_batched(dzpT, Mtp, chunk, aT, Mtp, chunk, Nn, Kk, chunk, self.split_k, BF, part, Kk, Nn * Kk, True)
        T.colsum(part, out=self.g(pn + "weight").view(-1))
        T.colsum(dzp, out=self.g(pn + "bias"))
        # ---- AdamW + OneCycleLR ---------------------------------------------------------------------------------------------------
        if data_parallel and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            torch.distributed.all_reduce(self.G, op=torch.distributed.ReduceOp.AVG)
        if update:
            self.step_count += 1
            lr = self._lrs[min(self.step_count - 1, len(self._lrs) - 1)]
            stats = self.P[self._stat_idx].clone() if self.alibi else None      # buffers: not touched by the optimiser (weight decay)
            T.adamw(self.P, self.G, self.m, self.v, lr, self.step_count, weight_decay=self.wd)
            if stats is not None:
                self.P[self._stat_idx] = stats
            self._refresh_weights()
        return loss.detach(), logits


def reference_flops_per_bag(T: int = 1024, F: int = 1024, D: int = 512, FF: int = 512, L: int = 2) -> float:
    """matmul FLOPs of one forward (2 per MAC): projection + L x (qkv, attention, out, fc1, fc2); training ~ 3x."""
    S = T + 1
    return 2 * T * F * D + L * (2 * S * D * 3 * D + 4 * S * S * D + 2 * S * D * D + 2 * S * D * FF + 2 * S * FF * D)
