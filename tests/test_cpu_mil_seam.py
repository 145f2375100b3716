"""Host logic of the MIL seam on CPU (no kernels run): the module tree / state_dict interchange with the reference's
checkpoints, the optimiser schedule, the autograd / torch.func plumbing around the HIP forward + backward (exercised with
stand-in arithmetic patched in for the two kernel entry points), the zero-padding of odd shapes, and the epoch loop."""
import dataclasses
from pathlib import Path

import numpy as np
import pytest
import torch
from torch import nn

from stamp_amd import mil_core
from stamp_amd.mil import VisionTransformer

G = Path(__file__).parent / "golden"
KW = dict(dim_output=2, dim_input=256, dim_model=128, n_layers=2, n_heads=2, dim_feedforward=128, dropout=0.1)


@pytest.mark.parametrize("tag,alibi", [("plain", False), ("alibi", True)])
def test_state_dict_keys_are_the_references(tag, alibi):
    """Keys, shapes and buffer-vs-parameter status equal those of the reference module (fixture = its state_dict)."""
    z = np.load(G / f"mil_vit_{tag}.npz")
    ref = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    C, F, D, L, H, FF = (int(v) for v in z["hparams"])
    m = VisionTransformer(dim_output=C, dim_input=F, dim_model=D, n_layers=L, n_heads=H, dim_feedforward=FF, dropout=0.0, use_alibi=alibi)
    sd = m.state_dict()
    assert list(sd) == list(ref)                                   # same keys in the same order
    assert all(tuple(sd[k].shape) == ref[k].shape for k in ref)
    bufs = {n for n, _ in m.named_buffers()}
    assert bufs == {k for k in ref if k.endswith(("running_mean", "items_so_far"))}
    assert not any("running_mean" in n or "items_so_far" in n for n, _ in m.named_parameters())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)


def test_nested_module_round_trips_a_reference_keyed_checkpoint():
    """The reference's Lit* wrappers hold the backbone as `self.model` (models/__init__.py:119-131): a parent's state_dict must
    carry `model.<key>` entries and load them back (the old override returned zero keys when nested)."""
    class Lit(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = VisionTransformer(use_alibi=True, **KW)
            self.register_buffer("class_weights", torch.ones(2))

    a, b = Lit(), Lit()
    sd = a.state_dict()
    assert "model.class_token" in sd and "model.transformer.layers.1.0.mhsa.attentions.1.scale_distance.running_mean" in sd
    assert len(sd) == len(a.model.state_dict()) + 1
    b.load_state_dict(sd, strict=True)
    assert all(torch.equal(p, q) for p, q in zip(a.model.state_dict().values(), b.model.state_dict().values()))
    with pytest.raises(RuntimeError):
        b.load_state_dict({k: v for k, v in sd.items() if k != "model.class_token"}, strict=True)
    # parameters() feeds an external optimiser; the scaler buffers are not in it
    n_par = sum(p.numel() for p in a.model.parameters())
    assert n_par == sum(v.numel() for k, v in a.model.state_dict().items() if not mil_core.is_buffer(k))


def test_onecycle_schedule_is_torchs_including_beta1():
    from stamp_amd.mil_train import onecycle_schedule

    lrs, b1s = onecycle_schedule(40, 1e-4, 25.0)
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, total_steps=40, max_lr=1e-4, div_factor=25.0)
    for i in range(40):
        assert opt.param_groups[0]["lr"] == lrs[i] and opt.param_groups[0]["betas"][0] == b1s[i]
        opt.step()
        if i < 39:
            sch.step()
    assert abs(lrs[0] - 4e-6) < 1e-12 and abs(b1s[0] - 0.95) < 1e-12 and abs(min(b1s) - 0.85) < 1e-9 and max(lrs) == pytest.approx(1e-4)


def test_onecycle_clock_epoch_interval_is_lightnings_default_and_step_interval_is_per_step():
    """The reference returns a bare `[optimizer], [scheduler]` (models/__init__.py:133-141); Lightning's default for that is
    interval="epoch": the schedule sized in steps moves once per epoch.  Both intervals against torch's own scheduler."""
    from stamp_amd.mil_train import OneCycleClock

    steps_per_epoch, epochs = 5, 4
    total = steps_per_epoch * epochs
    for interval in ("epoch", "step"):
        clock = OneCycleClock(total, 1e-4, 25.0, interval)
        opt = torch.optim.AdamW([nn.Parameter(torch.zeros(1))], lr=1e-3)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, total_steps=total, max_lr=1e-4, div_factor=25.0)
        n_sched = 0
        for e in range(epochs):
            for i in range(steps_per_epoch):
                assert clock.current() == (opt.param_groups[0]["lr"], opt.param_groups[0]["betas"][0]), (interval, e, i)
                opt.step()
                clock.after_step()
                if interval == "step" and n_sched < total - 1:
                    sch.step()
                    n_sched += 1
            clock.epoch_end()
            if interval == "epoch":
                sch.step()
        if interval == "epoch":      # after 4 epochs the LR is still on the warm-up ramp, 4 positions in
            assert clock.pos == epochs and clock.pos < clock.lrs.index(max(clock.lrs))
    with pytest.raises(ValueError):
        OneCycleClock(10, 1e-4, 25.0, "batch")
    held = OneCycleClock(3, 1e-4, 25.0, "step")
    for _ in range(7):
        held.after_step()
    assert held.current() == (held.lrs[-1], held.b1s[-1])


def test_padding_is_a_no_op_on_aligned_shapes_and_invertible_on_odd_ones():
    for kw, alibi in ((dict(F=1024, D=512, H=8, FF=512, C=2, L=1), False), (dict(F=456, D=132, H=4, FF=135, C=3, L=1), False),
                      (dict(F=40, D=60, H=3, FF=64, C=2, L=1), True)):
        d = mil_core.VitDims(alibi=alibi, **kw)
        assert d.Fp % 256 == 0 and d.Dp % 256 == 0 and d.FFp % 256 == 0 and d.Da % 256 == 0 and d.Ha % 4 == 0
        pk = mil_core.PackedVit.__new__(mil_core.PackedVit)
        pk.dims = d
        g = torch.Generator().manual_seed(0)
        w = torch.randn(3, d.H, d.hd, d.D, generator=g)
        wp = pk._pad_in(w)
        assert wp.shape == (3 * d.Da, d.Dp)
        back = pk.unpad_in_w(wp)
        ref = w.clone()
        ref[0] *= d.qscale ** 2                                     # q rows: scaled on the way in, gradient scaled on the way out
        assert torch.allclose(back, ref)
        wo = torch.randn(d.D, d.D, generator=g)
        assert torch.equal(pk.unpad_out_w(pk._pad_out(wo)), wo)
        if kw["D"] == 512:
            assert pk._pad_out(wo).data_ptr() == wo.data_ptr()        # aligned: a view, nothing copied
    with pytest.raises(NotImplementedError):
        mil_core.VitDims(F=8, D=170, H=5, FF=8, C=2, L=1, alibi=False)   # dim_model % 4 != 0
    with pytest.raises(NotImplementedError):
        mil_core.VitDims(F=8, D=256, H=2, FF=8, C=2, L=1, alibi=False)   # head_dim 128


def _patch_kernels(monkeypatch):
    """Stand-ins with the signatures of the two kernel entry points: a linear model logits = mean_t(bags) @ W^T + b on the
    module's mlp_head (dim_input == dim_model here).  Only the plumbing is under test."""
    class PK:
        def __init__(self, dims, get, act, train):
            self.dims, self.W, self.b = dims, get("mlp_head.0.weight"), get("mlp_head.0.bias")

    def fwd(pk, bags, coords, *, training, seed=0):
        pooled = bags.float().mean(1)
        return pooled @ pk.W.t() + pk.b, dict(pooled=pooled, shape=tuple(bags.shape))

    def bwd(pk, saved, dlogits, *, need_params=True, need_bags=False, split_k=32):
        Bb, Tn, Fd = saved["shape"]
        Gd_ = {"mlp_head.0.weight": dlogits.t() @ saved["pooled"], "mlp_head.0.bias": dlogits.sum(0)}
        dbags = (dlogits @ pk.W)[:, None, :].expand(Bb, Tn, Fd) / Tn if need_bags else None
        return (_Zero(Gd_) if need_params else {}), dbags

    class _Zero(dict):
        def __missing__(self, k):
            return torch.zeros(self.shapes[k])

    monkeypatch.setattr(mil_core, "PackedVit", PK)
    import stamp_amd.mil as mil
    monkeypatch.setattr(mil, "PackedVit", PK)
    monkeypatch.setattr(mil_core, "forward_train", fwd)
    monkeypatch.setattr(mil_core, "backward", bwd)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    return _Zero


def test_flat_order_makes_the_alibi_heads_per_head_tensors_contiguous_views():
    """The trainer's flat buffers follow mil_core.flat_order: every name keeps its own view, and the ALiBi head's 3 x H per-head Linears (weight, bias, weight,
    bias ... in the reference's state_dict) stack into CONTIGUOUS views of that buffer -- `_stack` copies nothing; on tensors that do not share a storage (an
    nn.Module's own parameters) it is torch.stack."""
    m = VisionTransformer(dim_output=2, dim_input=64, dim_model=128, n_layers=2, n_heads=2, dim_feedforward=64, dropout=0.0, use_alibi=True)
    sd = m.state_dict()
    names = list(sd.keys())
    order = mil_core.flat_order(m.dims, names)
    assert sorted(order) == sorted(names) and order != names
    assert mil_core.flat_order(mil_core.VitDims(F=64, D=128, H=2, FF=64, C=2, L=2, alibi=False), ["a", "b"]) == ["a", "b"]
    flat = torch.cat([sd[k].detach().float().reshape(-1) for k in order])
    offs, n = {}, 0
    for k in order:
        offs[k] = n
        n += sd[k].numel()
    get = lambda k: flat[offs[k]: offs[k] + sd[k].numel()].view(sd[k].shape)  # noqa: E731
    d = m.dims
    for l in range(d.L):
        p = mil_core.layer_prefix(l)
        for kind, shape in (("weight", (3, d.H, d.hd, d.D)), ("bias", (3, d.H, d.hd))):
            rows = [[p + f"0.mhsa.{e}.{h}.{kind}" for h in range(d.H)] for e in mil_core._ENC]
            w3 = mil_core._stack([mil_core._stack([get(n_) for n_ in row]) for row in rows])
            assert w3.shape == shape and w3.is_contiguous() and w3.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
            assert torch.equal(w3, torch.stack([torch.stack([sd[n_].float() for n_ in row]) for row in rows]))
        bs = mil_core._stack([get(p + f"0.mhsa.attentions.{h}.bias_scale").reshape(1) for h in range(d.H)]).reshape(d.H)
        assert bs.is_contiguous() and bs.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
    # the reference's own order: equally pitched but not contiguous -> still a view; separate tensors -> a copy with the same values
    flat2 = torch.cat([sd[k].detach().float().reshape(-1) for k in names])
    o2 = {k: sum(sd[j].numel() for j in names[:i]) for i, k in enumerate(names)}
    g2 = lambda k: flat2[o2[k]: o2[k] + sd[k].numel()].view(sd[k].shape)  # noqa: E731
    p = mil_core.layer_prefix(0)
    v = mil_core._stack([g2(p + f"0.mhsa.{mil_core._ENC[0]}.{h}.weight") for h in range(d.H)])
    assert not v.is_contiguous() and v.untyped_storage().data_ptr() == flat2.untyped_storage().data_ptr()
    sep = [sd[p + f"0.mhsa.{mil_core._ENC[0]}.{h}.weight"] for h in range(d.H)]
    assert torch.equal(mil_core._stack(sep), torch.stack(sep)) and torch.equal(v, torch.stack(sep))


def test_autograd_and_jacrev_plumbing(monkeypatch):
    """loss.backward() fills .grad of the module's nn.Parameters through the custom Function, gradients w.r.t. the bag flow, and
    torch.func.jacrev (the reference's heatmaps, heatmaps/__init__.py:36-56) works: vmap over the backward has a rule."""
    Zero = _patch_kernels(monkeypatch)
    torch.manual_seed(0)
    m = VisionTransformer(dim_output=3, dim_input=16, dim_model=16, n_layers=1, n_heads=1, dim_feedforward=16, dropout=0.0, use_alibi=False)
    Zero.shapes = {n: tuple(p.shape) for n, p in m.named_parameters()}
    bags = torch.randn(2, 5, 16, requires_grad=True)
    out = m.train()(bags, coords=None, mask=None)
    ref = bags.mean(1) @ m.mlp_head[0].weight.t() + m.mlp_head[0].bias
    assert torch.allclose(out, ref, atol=1e-6)
    w = torch.randn(2, 3)
    (out * w).sum().backward()
    gW, gb, gx = torch.autograd.grad((ref * w).sum(), [m.mlp_head[0].weight, m.mlp_head[0].bias, bags])
    assert torch.allclose(m.mlp_head[0].weight.grad, gW, atol=1e-6) and torch.allclose(m.mlp_head[0].bias.grad, gb, atol=1e-6)
    assert torch.allclose(bags.grad, gx, atol=1e-6)
    assert m.class_token.grad is not None and float(m.class_token.grad.abs().sum()) == 0.0
    # an external torch optimiser steps the parameters
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    before = m.mlp_head[0].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.mlp_head[0].weight)
    # jacrev as the reference's _gradcam_per_category calls it (eval mode, gradient w.r.t. the features)
    from torch.func import jacrev
    m.eval()
    feats = torch.randn(7, 16)
    jac = jacrev(lambda b: m.forward(b.unsqueeze(0), coords=torch.zeros(1, 7, 2), mask=None).squeeze(0))(feats)
    assert jac.shape == (3, 7, 16)
    assert torch.allclose(jac, (m.mlp_head[0].weight.detach() / 7)[:, None, :].expand(3, 7, 16), atol=1e-6)
    with pytest.raises(NotImplementedError):
        m.train()(bags, coords=None, mask=torch.zeros(2, 5, dtype=torch.bool))


def test_cpu_tensors_raise():
    m = VisionTransformer(use_alibi=False, **KW)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 256), coords=None, mask=None)
    with pytest.raises(ValueError):
        VisionTransformer(dim_output=2, dim_input=8, dim_model=30, n_layers=1, n_heads=4, dim_feedforward=8, dropout=0.0, use_alibi=False)


def test_fit_loop_early_stopping_and_best_weights():
    """Epoch loop semantics of the reference's train_model_ (train.py:504-564): validation every epoch, EarlyStopping(patience),
    best weights restored.  A stub trainer supplies a scripted validation-loss curve."""
    from stamp_amd import mil_train

    curve = [1.0, 0.8, 0.9, 0.85, 0.7, 0.75, 0.76, 0.77, 0.9]       # best at epoch 4; with patience 3 stops at epoch 7

    class Stub:
        dev = torch.device("cpu")

        def __init__(self):
            self.P = torch.zeros(1)
            self.epoch = -1
            self.step_count = 0
            self.synced = None
            self.epochs_ended = 0

        def epoch_end(self):
            self.epochs_ended += 1

        def _refresh(self):
            pass

        p = None

        def step(self, bags, targets, cw, coords=None, loss_fn=None):
            self.step_count += 1
            self.P += 1.0
            return torch.tensor(0.5), None

        def predict(self, bags, coords=None):
            return torch.tensor([[curve[self.epoch]]])

        def sync_to_model(self):
            self.synced = self.P.clone()

    st = Stub()

    def train_batches():
        st.epoch += 1
        return [(torch.zeros(2, 1, 1), None, None, torch.zeros(2, 1))] * 3

    hist = mil_train.fit(st, train_batches, lambda: [(torch.zeros(1, 1, 1), None, None, torch.zeros(1, 1))], max_epochs=9, patience=3,
                         loss_fn=lambda lg, t: lg.sum())
    assert hist["best_epoch"] == 4 and hist["stopped_epoch"] == 7 and len(hist["validation_loss"]) == 8
    assert hist["validation_loss"][:5] == pytest.approx(curve[:5])
    assert float(st.synced) == 15.0                                  # weights after epoch 4 (5 epochs x 3 steps), not the last ones
    assert st.epochs_ended == 8                                      # the scheduler clock is told about every finished epoch
