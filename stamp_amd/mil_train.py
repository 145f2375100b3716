"""Training of the MIL `vit` head on the HIP path: one optimisation step (forward + backward + AdamW, no autograd through the
network) and the epoch loop around it.

Mirrors what `stamp train` does for tile-level models:
* per batch -- reference src/stamp/modeling/models/__init__.py:239-279 (`LitTileClassifier._step`: ``logits = self.model(bags,
  coords=coords, mask=None)``, ``F.cross_entropy(logits, float one-hot targets, weight=class_weights)``), :444-483 (L1), :751-776
  (Cox); train mode, i.e. WITH the reference's dropout sites (`dropout` on project_features and inside nn.MultiheadAttention, the
  hard-coded 0.5 on both feed-forward Dropouts, vision_tranformer.py:157-169 / 268-271) and with every ALiBi `_RunningMeanScaler`
  updated before use (:24-29);
* optimiser -- :133-141: ``AdamW(lr=1e-3)`` with torch defaults wrapped in ``OneCycleLR(total_steps, max_lr, div_factor)``.  torch's
  OneCycleLR also cycles AdamW's beta1 (cycle_momentum=True: 0.95 -> 0.85 at the LR peak -> 0.95); both the LR and beta1 are read
  off torch's own scheduler evaluated on a dummy optimiser and fed to the fused kernel.  HOW OFTEN the schedule advances is not
  STAMP's decision: `configure_optimizers` returns a bare ``[optimizer], [scheduler]`` pair, and Lightning wraps an un-annotated
  scheduler in its default config ``interval="epoch", frequency=1`` -- so under `stamp train` the schedule, although sized in STEPS
  (``total_steps = len(train_dl) * max_epochs``, train.py:197-198), moves ONE position per EPOCH and the learning rate stays on the
  first ``max_epochs`` positions of the warm-up ramp.  `sched_interval="epoch"` (the default here) reproduces that;
  `sched_interval="step"` is the per-step OneCycle the schedule's size suggests was intended;
* epochs -- src/stamp/modeling/train.py:504-564: validation after every epoch (the caller's validation loader: full bags,
  batch size 1, :467-477), early stopping with `patience` on the validation loss (mode min), the best epoch's weights restored
  at the end (`shutil.copy(best_model_path)` + reload).

Mixed precision (stated, not hidden): 16-bit MFMA operands for activations, weights and gradients, fp32 accumulation, fp32 residual
stream and its gradient, fp32 LayerNorm / softmax statistics, fp32 master weights, gradients and Adam moments.  WHICH 16-bit type
follows torch's own flag, like the TransMIL head's products do: the reference calls `torch.set_float32_matmul_precision("high")` before
training (src/stamp/modeling/train.py:519) -- TF32-class products, 10 explicit mantissa bits per operand -- so under "high" (and
"highest") the operands are **fp16** (10 explicit bits, like TF32) and the loss is scaled by a static 2^10 on its way into the backward
(fp16's exponent range; the fp32 gradients are un-scaled before AdamW); under "medium" (bf16 products in torch's own wording) they are
**bf16** (8 bits, fp32's exponent range, no loss scaling) -- the only mode before round 6.  `precision=` overrides the flag.
The loss on the [batch, classes] logits is the one tiny piece left to torch.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import mil_core
from . import train_ops as T
from .distributed import average_buffers, average_gradients
from .mil import VisionTransformer
from .mil_core import PackedVit

BF = torch.bfloat16


def onecycle_schedule(total_steps: int, max_lr: float, div_factor: float) -> tuple[list[float], list[float]]:
    """(lr, beta1) per optimiser step exactly as torch's OneCycleLR drives AdamW (reference models/__init__.py:133-141)."""
    dummy = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(dummy, total_steps=total_steps, max_lr=max_lr, div_factor=div_factor)
    lrs, b1s = [], []
    for i in range(total_steps):
        lrs.append(dummy.param_groups[0]["lr"])
        b1s.append(dummy.param_groups[0]["betas"][0])
        dummy.step()
        if i + 1 < total_steps:
            sched.step()
    return lrs, b1s


class OneCycleClock:
    """Position of a trainer on torch's OneCycle (lr, beta1) table.  interval "epoch": the position moves in `epoch_end()` only
    (what Lightning does with the reference's bare scheduler); "step": it moves after every optimiser step.  Past the table's end
    the last entry is held (torch's scheduler would raise; Lightning never gets there: max_epochs <= total_steps)."""

    def __init__(self, total_steps: int, max_lr: float, div_factor: float, interval: str = "epoch") -> None:
        if interval not in ("epoch", "step"):
            raise ValueError(f"sched_interval must be 'epoch' or 'step', got {interval!r}")
        self.interval = interval
        self.lrs, self.b1s = onecycle_schedule(total_steps, max_lr, div_factor)
        self.pos = 0

    def current(self) -> tuple[float, float]:
        i = min(self.pos, len(self.lrs) - 1)
        return self.lrs[i], self.b1s[i]

    def after_step(self) -> None:
        if self.interval == "step":
            self.pos += 1

    def epoch_end(self) -> None:
        if self.interval == "epoch":
            self.pos += 1


class HipMilVitTrainer:
    def __init__(self, model: VisionTransformer, *, device="cuda", max_lr: float = 1e-4, div_factor: float = 25.0,
                 total_steps: int = 1000, weight_decay: float = 0.01, split_k: int = 32, dropout: bool | None = None,
                 sched_interval: str = "epoch", precision: str | None = None, loss_scale: float = 1024.0) -> None:
        """dropout: None = as the reference's train mode (live when the model's rates are > 0; the feed-forward rate is always 0.5);
        False = all dropout sites off (deterministic steps, e.g. for parity tests against autograd).
        sched_interval: "epoch" (default; what Lightning does with the reference's bare scheduler, see the module docstring: call
        `epoch_end()` after every epoch -- `fit` does) or "step" (OneCycle advanced after every optimiser step).
        precision: None = torch.get_float32_matmul_precision() at construction ("high" under the reference's train_model_, train.py:519); "high" /
        "highest" -> fp16 operands + static loss scale `loss_scale`; "medium" -> bf16 operands, no scaling (module docstring)."""
        self.precision = precision or torch.get_float32_matmul_precision()
        if self.precision not in ("medium", "high", "highest"):
            raise ValueError(f"precision must be 'medium', 'high' or 'highest', got {self.precision!r}")
        self.act = BF if self.precision == "medium" else torch.float16
        # fp16 gradients: a static power-of-two scale on dlogits (exact) lifts the 16-bit gradient tensors out of fp16's subnormal range; un-scaled in fp32
        self.loss_scale = 1.0 if self.act == BF else float(loss_scale)
        self.model = model
        self.dims = model.dims
        self.alibi = bool(model.use_alibi)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("HipMilVitTrainer runs on the GPU only")
        self.use_dropout = True if dropout is None else bool(dropout)
        self.split_k = split_k
        # flat fp32 master parameters + buffers (reference state_dict order), gradients and Adam moments
        sd = model.state_dict()
        self.names = list(sd.keys())                                   # the reference's state_dict order (checkpoints, sync_to_model)
        self.order = mil_core.flat_order(self.dims, self.names)        # order inside the flat buffers (ALiBi: per-head tensors stacked, mil_core.flat_order)
        self.shapes = {k: tuple(v.shape) for k, v in sd.items()}
        self.offs, n = {}, 0
        for k in self.order:
            self.offs[k] = (n, int(sd[k].numel()))
            n += self.offs[k][1]
        self._pv, self._gv = {}, {}                                    # name -> view (P and G are allocated once)
        self.P = torch.cat([sd[k].detach().float().reshape(-1) for k in self.order]).to(self.dev).contiguous()
        self.G = torch.zeros(n, device=self.dev)
        self.m = torch.zeros(n, device=self.dev)
        self.v = torch.zeros(n, device=self.dev)
        self.step_count = 0
        self.wd = weight_decay
        # ALiBi: running_mean / items_so_far are BUFFERS of the reference module (no gradient, no optimiser update)
        stat = [k for k in self.names if mil_core.is_buffer(k)]
        self._stat_idx = torch.tensor([self.offs[k][0] for k in stat], dtype=torch.long, device=self.dev)
        self.clock = OneCycleClock(total_steps, max_lr, div_factor, sched_interval)
        self._lrs, self._b1s = self.clock.lrs, self.clock.b1s
        self.pk = PackedVit(self.dims, self.p, self.act, train=True)

    # ---- parameter views ------------------------------------------------------------------------------------------------
    def p(self, name: str) -> torch.Tensor:
        v = self._pv.get(name)
        if v is None:
            o, n = self.offs[name]
            v = self._pv[name] = self.P[o:o + n].view(self.shapes[name])
        return v

    def g(self, name: str) -> torch.Tensor:
        v = self._gv.get(name)
        if v is None:
            o, n = self.offs[name]
            v = self._gv[name] = self.G[o:o + n].view(self.shapes[name])
        return v

    def sync_to_model(self) -> None:
        self.model.load_state_dict({k: self.p(k).detach().clone() for k in self.names})

    def load_from_model(self) -> None:
        sd = self.model.state_dict()
        self.P.copy_(torch.cat([sd[k].detach().float().reshape(-1) for k in self.order]).to(self.dev))
        self._refresh()

    # ---- one optimisation step ------------------------------------------------------------------------------------------------
    def step(self, bags: torch.Tensor, targets: torch.Tensor, class_weights: torch.Tensor | None = None, *, update: bool = True,
             data_parallel: bool = False, coords: torch.Tensor | None = None, loss_fn=None, seed: int | None = None):
        """bags [Bb,T,F] fp16/bf16/fp32 on the GPU, targets float one-hot [Bb,C]. Returns (loss, logits).

        loss_fn(logits, targets) -> scalar selects the task (stamp_amd.losses): default = the classifier's weighted
        cross-entropy; `losses.l1_loss` = LitTileRegressor (dim_output 1); `losses.cox_survival_loss` = LitTileSurvival
        (dim_output 1, targets [time, event]).  seed: dropout seed of this step (default: drawn from torch's CPU generator).

        data_parallel=True: every rank of the initialised process group holds a replica and its own bags; the flat
        fp32 gradient buffer (14.7 MB for the default head) is averaged with ONE RCCL all-reduce before AdamW
        (the reference itself is single-device, src/stamp/modeling/train.py:541-547).

        update=False computes loss, logits and gradients and leaves EVERY piece of state as it was: no optimiser step, and the ALiBi
        running-mean buffers (which a train-mode forward updates before use, vision_tranformer.py:24-29) are put back afterwards."""
        if update or not self.alibi:
            return self._step(bags, targets, class_weights, update, data_parallel, coords, loss_fn, seed)
        stats_before = self.P[self._stat_idx].clone()
        try:
            return self._step(bags, targets, class_weights, update, data_parallel, coords, loss_fn, seed)
        finally:
            self.P[self._stat_idx] = stats_before
            self._refresh()

    def _step(self, bags, targets, class_weights, update, data_parallel, coords, loss_fn, seed):
        dev = self.dev
        dist_on = data_parallel and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.use_dropout else 0
        if self.alibi:
            if coords is None:
                raise ValueError("use_alibi=True needs coords")
            cc = mil_core._coords_with_cls(coords, bags.shape[0], dev)
            mil_core.update_running_means(self.p, self.dims, cc)
            if dist_on:      # replicas see different bags: keep the scaler buffers identical (mean of the ranks' updates)
                self.P[self._stat_idx] = average_buffers(self.P[self._stat_idx])
            self._refresh()
        logits, saved = mil_core.forward_train(self.pk, bags, coords, training=self.use_dropout, seed=seed)
        # ---- loss on [Bb, C]: the reference's weighted CE with float one-hot targets (models/__init__.py:254-258) ---------------------
        with torch.enable_grad():
            lg = logits.detach().clone().requires_grad_(True)
            if loss_fn is None:
                loss = F.cross_entropy(lg, targets.to(dev, torch.float32), weight=None if class_weights is None else class_weights.to(dev, torch.float32))
            else:
                loss = loss_fn(lg, targets.to(dev))
            if not update and not loss.requires_grad:
                return loss.detach(), logits
            dlogits = torch.autograd.grad(loss, lg, allow_unused=True)[0] if loss.requires_grad else None
        if dlogits is None:      # e.g. a Cox batch without events (cox.py:219-224 returns a constant 0): nothing to learn from, no step
            return loss.detach(), logits
        if self.loss_scale != 1.0:
            dlogits = dlogits * self.loss_scale
        G, _ = mil_core.backward(self.pk, saved, dlogits, need_params=True, need_bags=False, split_k=self.split_k, grad_views=self.g)
        grouped = self._copy_grouped_grads(G) if self.alibi else ()
        for k, gk in G.items():          # (unpadded geometries: the library wrote into the flat buffer's views themselves -- nothing to copy)
            if k not in grouped and gk.data_ptr() != self.g(k).data_ptr():
                self.g(k).copy_(gk)
        if self.loss_scale != 1.0:
            self.G.mul_(1.0 / self.loss_scale)
        # ---- AdamW + OneCycleLR ---------------------------------------------------------------------------------------------------
        if dist_on:
            average_gradients(self.G)
        if update:
            self.step_count += 1
            lr, b1 = self.clock.current()
            stats = self.P[self._stat_idx].clone() if self.alibi else None      # buffers: not touched by the optimiser (weight decay)
            T.adamw(self.P, self.G, self.m, self.v, lr, self.step_count, betas=(b1, 0.999), weight_decay=self.wd)
            self.clock.after_step()
            if stats is not None:
                self.P[self._stat_idx] = stats
            self._refresh()
        return loss.detach(), logits

    def _copy_grouped_grads(self, G: dict) -> set:
        """ALiBi: the per-head Linears are 3 x H weights + 3 x H biases + H bias scales per layer -- separate tensors in the reference's state_dict, equally
        pitched slices both of the library's packed gradient and of this trainer's flat buffer.  One strided copy per group instead of one launch per
        tensor (~110 per layer and step, which the GPU then waited for).  -> the names served."""
        d, done = self.dims, set()
        first = mil_core.layer_prefix(0) + f"0.mhsa.{mil_core._ENC[0]}.0.weight"
        if G[first].data_ptr() == self.g(first).data_ptr():
            return done                  # (the library wrote into the flat buffer itself: mil_core._direct_grad_structs)
        for l in range(d.L):
            p = mil_core.layer_prefix(l)
            groups = [[[p + f"0.mhsa.{e}.{h}.{kind}" for h in range(d.H)] for e in mil_core._ENC] for kind in ("weight", "bias")]
            groups.append([[p + f"0.mhsa.attentions.{h}.bias_scale" for h in range(d.H)]])
            for grp in groups:
                dst = mil_core._stack([mil_core._stack([self.g(n) for n in row]) for row in grp])
                if dst.untyped_storage().data_ptr() != self.G.untyped_storage().data_ptr():
                    continue             # (not a view of the flat buffer: the per-tensor loop serves these)
                dst.copy_(mil_core._stack([mil_core._stack([G[n] for n in row]) for row in grp]))
                done.update(n for row in grp for n in row)
        return done

    def epoch_end(self) -> None:
        """Lightning steps an epoch-interval scheduler once after every training epoch."""
        self.clock.epoch_end()

    def _refresh(self) -> None:
        """The packed operands follow the master parameters; any cached inference pack is stale from here on."""
        self.pk.refresh(self.p)
        self._eval_pk_step = -1

    # ---- evaluation (validation / deploy): inference kernels, fp16 operands, any bag length ---------------------------------------------
    @torch.no_grad()
    def predict(self, bags: torch.Tensor, coords: torch.Tensor | None = None) -> torch.Tensor:
        # the fp16 inference pack is rebuilt whenever the master parameters or buffers changed since it was made (`_refresh`)
        if getattr(self, "_eval_pk_step", -1) != self.step_count or getattr(self, "_eval_pk", None) is None:
            self._eval_pk = PackedVit(self.dims, self.p, torch.float16, train=False)
            self._eval_pk_step = self.step_count
        return mil_core.forward_infer(self._eval_pk, bags, coords, None)


def fit(trainer: HipMilVitTrainer, train_batches, valid_batches, *, max_epochs: int, patience: int = 16, class_weights=None,
        loss_fn=None, log=None) -> dict:
    """The epoch loop of the reference's `train_model_` (src/stamp/modeling/train.py:504-564) around `HipMilVitTrainer.step`.

    train_batches / valid_batches: callables returning an iterable of (bags, coords, bag_sizes, targets) per epoch -- the
    reference's loaders (train: fixed-size bags, batch 64, shuffled; validation: full bags, batch 1, `bag_size=None`, train.py:455-477).
    After every epoch the validation loss (same objective, eval mode: no dropout, frozen scalers; Lightning's mean over batches
    weighted by batch size) is computed; training stops when it has not improved for `patience` epochs (EarlyStopping, mode min);
    the best epoch's weights are restored and copied into `trainer.model` (the reference copies the best checkpoint and reloads
    it).  `num_sanity_val_steps=0` like the reference.  Returns the history."""
    dev = trainer.dev
    best = {"loss": float("inf"), "epoch": -1, "P": None}
    hist = {"train_loss": [], "validation_loss": [], "best_epoch": -1, "stopped_epoch": None}
    wait = 0
    for epoch in range(max_epochs):
        tot, cnt = 0.0, 0
        for bags, coords, _sizes, targets in train_batches():
            loss, _ = trainer.step(bags.to(dev), targets, class_weights, coords=None if coords is None else coords.to(dev), loss_fn=loss_fn)
            tot += float(loss) * bags.shape[0]
            cnt += bags.shape[0]
        trainer.epoch_end()
        hist["train_loss"].append(tot / max(cnt, 1))
        vtot, vcnt = 0.0, 0
        for bags, coords, _sizes, targets in valid_batches():
            lg = trainer.predict(bags.to(dev), None if coords is None else coords.to(dev))
            if loss_fn is None:
                vl = F.cross_entropy(lg, targets.to(dev, torch.float32), weight=None if class_weights is None else class_weights.to(dev, torch.float32))
            else:
                vl = loss_fn(lg, targets.to(dev))
            vtot += float(vl) * bags.shape[0]
            vcnt += bags.shape[0]
        vloss = vtot / max(vcnt, 1)
        hist["validation_loss"].append(vloss)
        if log:
            log(f"epoch {epoch}: train {hist['train_loss'][-1]:.5f} validation {vloss:.5f}")
        if vloss < best["loss"]:
            best.update(loss=vloss, epoch=epoch, P=trainer.P.clone())
            wait = 0
        else:
            wait += 1
            if wait >= patience:
                hist["stopped_epoch"] = epoch
                break
    if best["P"] is not None:
        trainer.P.copy_(best["P"])
        trainer._refresh()
    trainer.sync_to_model()
    hist["best_epoch"] = best["epoch"]
    return hist


def reference_flops_per_bag(T: int = 1024, F: int = 1024, D: int = 512, FF: int = 512, L: int = 2) -> float:
    return mil_core.flops_per_bag(T, F, D, FF, L)
