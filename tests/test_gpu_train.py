"""Training-step kernels (through the C ABI) against torch autograd / torch.optim on the CPU in fp64/fp32."""
import pytest
import torch

from stamp_amd import _lib, ops, train_ops as T

pytestmark = pytest.mark.gpu


def test_transpose_colsum(gpu):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(777, 300, generator=g).to(gpu, torch.bfloat16)
    t = T.transpose16(x, ld_dst=832)
    assert torch.equal(t[:, :777], x.t()) and (t[:, 777:] == 0).all()
    s = T.colsum(x)
    assert (s.cpu() - x.float().sum(0).cpu()).abs().max() < 1e-2
    s2 = T.colsum(x.float(), out=s.clone(), accumulate=True)
    assert torch.allclose(s2, 2 * s, rtol=1e-3, atol=2e-2)
    assert torch.equal(T.colsum(x), s)                # deterministic


def test_colsum_multi_is_colsum_bit_for_bit(gpu):
    """amds_colsum_multi finishes many column sums with one launch: direct sums of small fp32 matrices (also pitched rows, odd widths) and the second
    stage over amds_colsum_partials' chunk partials -- the bits of amds_colsum on every matrix."""
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(1025, 512, generator=g).to(gpu),                       # LayerNorm parameter-gradient partials of 65 600 rows
          torch.randn(64, 2, generator=g).to(gpu),                           # head bias: two columns, unaligned rows
          torch.randn(64, 1025 * 512, generator=g).to(gpu)[:, :512],         # class-token rows: a pitched view
          torch.randn(2048, 300, generator=g).to(gpu),
          torch.randn(65600, 512, generator=g).to(gpu, torch.bfloat16),      # bias gradients over 16-bit tensors: 65 chunks
          torch.randn(65600, 1536, generator=g).to(gpu, torch.bfloat16),
          torch.randn(4100, 130, generator=g).to(gpu),                       # fp32, 5 chunks, scalar column tail
          torch.randn(1, 64, generator=g).to(gpu)]
    xs = xs * 5                                                              # 40 entries: two launches
    got = T.colsum_multi(xs)
    for x, s in zip(xs, got):
        assert torch.equal(s, T.colsum(x)), tuple(x.shape)
        assert (s.cpu().double() - x.double().sum(0).cpu()).abs().max() < 2e-2 * max(1.0, (x.shape[0] / 1000) ** 0.5)


@pytest.mark.parametrize("R,C,ld", [(128, 64, None), (1024, 512, None), (640, 2048, 704), (64 * 1025, 512, None)])
def test_transpose_vector_path(gpu, R, C, ld):
    """R and C multiples of 64 take the 16-byte-per-lane kernel; `ld` = a padded destination pitch."""
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g).to(gpu, torch.bfloat16)
    t = T.transpose16(x) if ld is None else T.transpose16(x, ld_dst=ld)
    assert torch.equal(t[:, :R], x.t())
    xs = torch.randn(R, C + 64, generator=g).to(gpu, torch.bfloat16)[:, 64:]      # a view: source pitch != columns
    assert torch.equal(T.transpose16(xs), xs.t())


@pytest.mark.parametrize("rows,cols", [(1025, 512), (70, 128), (4100, 1024)])
def test_layernorm_train_and_bwd(gpu, rows, cols):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.5)
    gamma, beta = 1 + 0.2 * torch.randn(cols, generator=g), 0.1 * torch.randn(cols, generator=g)
    dy = torch.randn(rows, cols, generator=g)
    skip = torch.randn(rows, cols, generator=g)
    xd = x.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (cols,), gd, bd, 1e-5)
    y.backward(dy.double())
    yh, mean, rstd = T.layernorm_train(x.to(gpu), gamma.to(gpu), beta.to(gpu), 1e-5, torch.float32)
    assert (yh.cpu().double() - y.detach()).abs().max() < 1e-5
    dx = skip.clone().to(gpu)
    dgam, dbet = torch.zeros(cols, device=gpu), torch.zeros(cols, device=gpu)
    T.layernorm_bwd(dy.to(gpu), x.to(gpu), mean, rstd, gamma.to(gpu), dx, True, dgam, dbet)
    assert (dx.cpu().double() - (skip.double() + xd.grad)).abs().max() < 2e-5
    assert (dgam.cpu().double() - gd.grad).abs().max() < 1e-3 * max(1.0, gd.grad.abs().max().item())
    assert (dbet.cpu().double() - bd.grad).abs().max() < 1e-3 * max(1.0, bd.grad.abs().max().item())
    dx2 = torch.empty(rows, cols, device=gpu)
    T.layernorm_bwd(dy.to(gpu), x.to(gpu), mean, rstd, gamma.to(gpu), dx2, False, dgam, dbet, accumulate_params=True)
    assert (dx2.cpu().double() - xd.grad).abs().max() < 2e-5
    assert (dgam.cpu().double() - 2 * gd.grad).abs().max() < 2e-3 * max(1.0, gd.grad.abs().max().item())


def test_gelu_fwd_bwd(gpu):
    z = torch.linspace(-8, 8, 100001)
    du = torch.randn(100001, generator=torch.Generator().manual_seed(1))
    zd = z.double().requires_grad_(True)
    u = torch.nn.functional.gelu(zd)
    u.backward(du.double())
    assert (T.gelu_fwd(z.to(gpu)).cpu().double() - u.detach()).abs().max() < 1e-6
    assert (T.gelu_bwd(z.to(gpu), du.to(gpu)).cpu().double() - zd.grad).abs().max() < 1e-5
    zb = z.to(gpu, torch.bfloat16)
    dzb = T.gelu_bwd(zb, du.to(gpu, torch.bfloat16))
    ref = torch.autograd.grad(torch.nn.functional.gelu(zb.cpu().double().requires_grad_(True)).sum(), [], allow_unused=True) if False else None
    assert torch.isfinite(dzb.float()).all() and dzb.dtype == torch.bfloat16


def test_adamw_matches_torch(gpu):
    torch.manual_seed(0)
    p0 = torch.randn(10007)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone().to(gpu), torch.zeros(10007, device=gpu), torch.zeros(10007, device=gpu)
    for step in range(1, 6):
        gcpu = torch.randn(10007)
        ref.grad = gcpu.clone()
        opt.step()
        T.adamw(p, gcpu.to(gpu), m, v, 3e-4, step)
        assert (p.cpu() - ref.detach()).abs().max() < 1e-6


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Tn,H", [(2, 1025, 8), (1, 200, 2), (1, 64, 1), (2, 129, 3), (1, 65, 1), (1, 1, 1), (2, 7, 2), (1, 128, 2), (1, 4097, 1)])
def test_attention_backward_vs_autograd(gpu, dt, B, Tn, H):
    g = torch.Generator().manual_seed(B * 100 + Tn + H)
    D = H * 64
    qkv = torch.randn(B * Tn, 3 * D, generator=g).to(dt)
    dout = torch.randn(B * Tn, D, generator=g).to(dt)
    x = qkv.double().requires_grad_(True)
    q, k, v = x.reshape(B, Tn, 3, H, 64).permute(2, 0, 3, 1, 4)
    o = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(B * Tn, D)
    o.backward(dout.double())
    out, lse = T.attention_fwd_lse(qkv.to(gpu), B, Tn, H)
    eps = 2 ** -7 if dt == torch.bfloat16 else 2 ** -10
    assert (out.cpu().double() - o.detach()).abs().max() < 4 * eps * max(1.0, o.abs().max().item())
    lref = torch.logsumexp((q @ k.transpose(-1, -2) / 8.0).detach(), -1) / torch.log(torch.tensor(2.0, dtype=torch.float64))
    assert (lse.cpu().double() - lref).abs().max() < 1e-2
    dqkv = T.attention_bwd(qkv.to(gpu), out, dout.to(gpu), lse, B, Tn, H)
    gref = x.grad
    err = (dqkv.cpu().double() - gref).abs().max().item()
    assert err < 8 * eps * max(1.0, gref.abs().max().item()), (err, gref.abs().max().item())
    assert torch.equal(dqkv, T.attention_bwd(qkv.to(gpu), out, dout.to(gpu), lse, B, Tn, H))


def test_gemm_batched_split_k(gpu):
    """weight-gradient shape: dW[N=256][K'=512] = dy^T[256][M] . x^T[512][M]^T, contraction over M split in 8 chunks"""
    g = torch.Generator().manual_seed(2)
    M, Nn, Kk, S = 8 * 1024, 256, 512, 8
    dyT = torch.randn(Nn, M, generator=g).to(gpu, torch.bfloat16)
    xT = torch.randn(Kk, M, generator=g).to(gpu, torch.bfloat16)
    part = torch.empty(S, Nn, Kk, dtype=torch.float32, device=gpu)
    # out[n][k] = sum_m dyT[n][m] xT[k][m]: A = dyT (rows n), W = xT (rows k)  ->  "N" of the GEMM is Kk
    T.gemm_batched(dyT, M, M // S, xT, M, M // S, Nn, Kk, M // S, S, torch.bfloat16, part, Kk, Nn * Kk, True)
    dW = T.colsum(part.view(S, Nn * Kk)).view(Nn, Kk)
    ref = dyT.double() @ xT.double().t()
    assert (dW.cpu().double() - ref.cpu()).abs().max() < 1e-3 * ref.abs().max().item()


def _oracle_loss_and_grads(sd, bags, targets, weights, n_heads):
    from oracle.mil_vit import mil_vit_forward

    p = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
    logits = mil_vit_forward_d(bags.double(), p, n_heads)
    loss = torch.nn.functional.cross_entropy(logits, targets.double(), weight=None if weights is None else weights.double())
    loss.backward()
    return loss.item(), logits.detach(), {k: v.grad for k, v in p.items()}


def mil_vit_forward_d(bags, sd, n_heads):
    """fp64 autograd twin of oracle.mil_vit.mil_vit_forward (which casts its weights to fp32)."""
    import torch.nn.functional as F
    B = bags.shape[0]
    x = F.gelu(F.linear(bags, sd["project_features.0.weight"], sd["project_features.0.bias"]))
    D = x.shape[-1]
    x = torch.cat([sd["class_token"].reshape(1, 1, D).expand(B, -1, -1), x], dim=1)
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.layers."))
    hd = D // n_heads
    for l in range(L):
        p = f"transformer.layers.{l}."
        h = F.layer_norm(x, (D,), sd[p + "0.norm.weight"], sd[p + "0.norm.bias"])
        qkv = F.linear(h, sd[p + "0.mhsa.in_proj_weight"], sd[p + "0.mhsa.in_proj_bias"])
        q, k, v = (t.reshape(B, -1, n_heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        o = (torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, -1) @ v).transpose(1, 2).reshape(B, -1, D)
        x = F.linear(o, sd[p + "0.mhsa.out_proj.weight"], sd[p + "0.mhsa.out_proj.bias"]) + x
        h = F.layer_norm(x, (D,), sd[p + "1.0.weight"], sd[p + "1.0.bias"])
        h = F.linear(F.gelu(F.linear(h, sd[p + "1.1.weight"], sd[p + "1.1.bias"])), sd[p + "1.4.weight"], sd[p + "1.4.bias"])
        x = h + x
    x = F.layer_norm(x, (D,), sd["transformer.norm.weight"], sd["transformer.norm.bias"])
    return F.linear(x[:, 0], sd["mlp_head.0.weight"], sd["mlp_head.0.bias"])


def test_mil_vit_training_step_matches_autograd(gpu):
    """forward + backward of the HIP trainer vs fp64 autograd of the same network (the oracle's architecture, which is
    pinned to the reference): loss, logits and EVERY parameter gradient; then AdamW steps vs torch.optim.AdamW."""
    from oracle.mil_vit import mil_vit_forward
    from stamp_amd.mil import VisionTransformer
    from stamp_amd.mil_train import HipMilVitTrainer

    torch.manual_seed(5)
    Bb, Tn, Fd, C = 3, 200, 256, 2
    model = VisionTransformer(dim_output=C, dim_input=Fd, dim_model=256, n_layers=2, n_heads=4, dim_feedforward=256, dropout=0.0, use_alibi=False)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and "class_token" not in n:
                p.add_(0.05 * torch.randn_like(p))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    bags = torch.randn(Bb, Tn, Fd).half()
    targets = torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.0, 1.0]])
    weights = torch.tensor([0.7, 0.3])
    # the twin reproduces the pinned oracle
    with torch.no_grad():
        assert torch.allclose(mil_vit_forward_d(bags.double(), {k: v.double() for k, v in sd.items()}, 4).float(),
                              mil_vit_forward(bags.float(), torch.zeros(Bb, Tn, 2), None, sd, n_heads=4, use_alibi=False), atol=1e-4)
    ref_loss, ref_logits, ref_g = _oracle_loss_and_grads(sd, bags.float(), targets, weights, 4)
    tr = HipMilVitTrainer(model, device=gpu, max_lr=1e-3, div_factor=25.0, total_steps=50, sched_interval="step", split_k=4, dropout=False)
    loss, logits = tr.step(bags.to(gpu), targets, weights, update=False)
    assert abs(loss.item() - ref_loss) < 2e-2 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    assert (logits.cpu().double() - ref_logits).abs().max() < 3e-2 * max(1.0, ref_logits.abs().max().item())
    worst = 0.0
    for k in tr.names:
        g, r = tr.g(k).cpu().double(), ref_g[k]
        rel = ((g - r).norm() / (r.norm() + 1e-12)).item()
        worst = max(worst, rel)
        assert rel < 5e-2, (k, rel, r.norm().item())
    print("worst relative-L2 gradient error (bf16 operands):", worst)
    # a few optimisation steps must reduce the loss and keep parameters finite; step count / lr schedule as torch's
    losses = [tr.step(bags.to(gpu), targets, weights)[0].item() for _ in range(8)]
    assert losses[-1] < losses[0] and torch.isfinite(tr.P).all()
    ref_sched = torch.optim.lr_scheduler.OneCycleLR(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1e-3), total_steps=50, max_lr=1e-3, div_factor=25.0)
    assert abs(tr._lrs[0] - 1e-3 / 25.0) < 1e-12 and len(tr._lrs) == 50
    tr.sync_to_model()
    assert torch.allclose(model.state_dict()["mlp_head.0.bias"].cpu(), tr.p("mlp_head.0.bias").cpu())


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize("B,Tn,H", [(2, 200, 2), (1, 1025, 8), (1, 65, 1), (3, 5, 2)])
def test_alibi_attention_fwd_bwd_vs_autograd(gpu, B, Tn, H):
    """Post-softmax distance bias (reference _ALiBi.forward): out = softmax(qk^T/8) v - bs_h * (cdist/rm_h) v.
    HIP forward + backward (dq, dk, dv, d bias_scale) against fp64 autograd on the same bf16-rounded q, k, v."""
    g = torch.Generator().manual_seed(B * Tn)
    D = H * 64
    qkv = (torch.randn(B * Tn, 3 * D, generator=g) * 0.8).bfloat16()
    coords = torch.rand(B, Tn, 2, generator=g) * 4000.0
    coords[:, 0] = 0.0                                           # the class token sits at (0, 0) (vision_tranformer.py:349-351)
    bs = torch.rand(H, generator=g) * 0.5 + 0.1
    rm = torch.rand(H, generator=g) * 500.0 + 1800.0
    dout = (torch.randn(B * Tn, D, generator=g) * 0.5).bfloat16()

    q3 = qkv.double().reshape(B, Tn, 3, H, 64).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)      # [3,B,H,T,64]
    bsd = bs.double().clone().requires_grad_(True)
    dist = torch.cdist(coords.double(), coords.double())                                                  # [B,T,T]
    w = torch.softmax(q3[0] @ q3[1].transpose(-2, -1) / 8.0, -1) - (dist[:, None] / rm.double().view(1, H, 1, 1)) * bsd.view(1, H, 1, 1)
    ref = (w @ q3[2]).permute(0, 2, 1, 3).reshape(B * Tn, D)
    ref.backward(dout.double())
    ref_dqkv = q3.grad.permute(1, 3, 0, 2, 4).reshape(B * Tn, 3 * D)

    inv_rm = (1.0 / rm).to(gpu)
    out, u, osm, lse = T.attention_alibi_fwd_train(qkv.to(gpu), coords.to(gpu), inv_rm, bs.to(gpu), B, Tn, H)
    assert _rel(out.float().cpu(), ref.detach()) < 8e-3                  # the distance operand is rounded to bf16
    dqkv, dbs = T.attention_alibi_bwd(qkv.to(gpu), osm, u, dout.to(gpu), lse, coords.to(gpu), bs.to(gpu), (bs / rm).to(gpu), B, Tn, H)
    d = dqkv.float().cpu().reshape(B * Tn, 3, D)
    r = ref_dqkv.reshape(B * Tn, 3, D)
    for i, name in enumerate("qkv"):
        assert _rel(d[:, i], r[:, i]) < 1.5e-2, (name, _rel(d[:, i], r[:, i]))
    # (one scalar per head: a sum of B T^2 mixed-sign terms whose distance factors are rounded to bf16 -- on a few thousand terms the rounding does not average out)
    assert _rel(dbs.cpu(), bsd.grad) < (1e-2 if B * Tn * Tn > 20000 else 3e-2), (dbs.cpu(), bsd.grad)
    m = T.cdist_mean(coords.to(gpu))
    assert abs(m.item() - dist.mean().item()) < 1e-3 * dist.mean().item()


def test_mil_vit_alibi_training_step_matches_autograd(gpu):
    """use_alibi=True: forward + backward of the HIP trainer vs fp32 autograd through the pinned oracle
    (oracle.mil_vit, golden-tested against the reference's MultiHeadALiBi): loss, logits, every parameter gradient incl.
    bias_scale and the 3 x heads per-head encoders, and the train-mode running-mean update of every head."""
    from oracle.mil_vit import mil_vit_forward, running_mean_update
    from stamp_amd.mil import VisionTransformer
    from stamp_amd.mil_train import HipMilVitTrainer

    torch.manual_seed(7)
    Bb, Tn, Fd, C, H = 3, 150, 256, 2, 4
    model = VisionTransformer(dim_output=C, dim_input=Fd, dim_model=256, n_layers=2, n_heads=H, dim_feedforward=256, dropout=0.0, use_alibi=True)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1 and "class_token" not in n and "scale_distance" not in n and "bias_scale" not in n:
                p.add_(0.05 * torch.randn_like(p))
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    bags = torch.randn(Bb, Tn, Fd).half()
    coords = torch.rand(Bb, Tn, 2) * 3000.0
    targets = torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.0, 1.0]])
    weights = torch.tensor([0.7, 0.3])
    # reference semantics of a train-mode forward: every scaler folds this batch's distances in BEFORE it is used
    cc = torch.cat([coords.new_zeros(Bb, 1, 2), coords], dim=1)
    dist = torch.cdist(cc, cc)
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in sd0:
        if k.endswith("scale_distance.running_mean"):
            rm, n = running_mean_update(sd0[k], sd0[k[: -len("running_mean")] + "items_so_far"], dist)
            sd[k], sd[k[: -len("running_mean")] + "items_so_far"] = rm, n
    params = {k: v.clone().requires_grad_(not k.endswith(("running_mean", "items_so_far"))) for k, v in sd.items()}
    logits_ref = mil_vit_forward(bags.float(), coords, None, params, n_heads=H, use_alibi=True)
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, targets, weight=weights)
    loss_ref.backward()

    tr = HipMilVitTrainer(model, device=gpu, max_lr=1e-3, div_factor=25.0, total_steps=50, sched_interval="step", split_k=4, dropout=False)
    loss, logits = tr.step(bags.to(gpu), targets, weights, update=False, coords=coords.to(gpu))
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * max(1.0, abs(loss_ref.item())), (loss.item(), loss_ref.item())
    assert (logits.cpu() - logits_ref.detach()).abs().max() < 3e-2 * max(1.0, logits_ref.abs().max().item())
    worst, report = 0.0, []
    for k in tr.names:
        if k.endswith(("running_mean", "items_so_far")):
            assert torch.equal(tr.p(k).cpu(), sd0[k].float()), k       # update=False: the scaler buffers are put back (the forward itself used the updated ones)
            continue
        g, r = tr.g(k).cpu().double(), params[k].grad.double()
        if "key_encoders" in k and k.endswith(".bias"):
            # softmax is invariant to a per-query constant, so the TRUE gradient of a key bias is 0 (autograd: 1e-10);
            # hold the bf16 path to "small against the sibling value-bias gradient" instead of a relative error
            sib = params[k.replace("key_encoders", "value_encoders")].grad.double().norm()
            assert r.norm() < 1e-6 and g.norm() < 5e-2 * sib, (k, g.norm().item(), sib.item())
            continue
        # q / k encoders of the LAST layer only get gradient through the class token's single query row, and with ALiBi the
        # softmax part is a small fraction of the attention output: their gradients are ~20x smaller than the value encoder's.
        # Error is measured against max(|r|, 5 % of the sibling value-encoder gradient) -- what matters to the optimiser.
        floor = 0.0
        if "query_encoders" in k or "key_encoders" in k:
            floor = 0.05 * params[k.replace("query_encoders", "value_encoders").replace("key_encoders", "value_encoders")].grad.double().norm().item()
        rel = ((g - r).norm() / max(r.norm().item(), floor, 1e-12)).item()
        worst = max(worst, rel)
        report.append((rel, k, r.norm().item()))
    report.sort(reverse=True)
    print("largest gradient errors with ALiBi (bf16 operands):", [(round(a, 4), b, float(f"{c:.2e}")) for a, b, c in report[:8]])
    for rel, k, rn in report:
        # bias_scale is a scalar: its gradient is a signed sum over (in the last layer) only the class-token rows
        assert rel < (5e-2 if k.endswith("bias_scale") else 2e-2), (k, rel, rn)      # measured: 1.8e-2 (one head's scalar) / 8.2e-3
    losses = []
    for i in range(8):
        losses.append(tr.step(bags.to(gpu), targets, weights, coords=coords.to(gpu))[0].item())
        if i == 0:      # the first real step: buffers updated exactly as the reference's train-mode forward does, and kept
            for k in tr.names:
                if k.endswith(("running_mean", "items_so_far")):
                    assert torch.allclose(tr.p(k).cpu(), sd[k], rtol=1e-5), (k, tr.p(k).cpu(), sd[k])
    assert losses[-1] < losses[0] and torch.isfinite(tr.P).all()
    n_key = next(k for k in tr.names if k.endswith("items_so_far"))
    assert tr.p(n_key).item() == 1.0 + 8                                   # eight train-mode forwards that count, weight decay never touched it
    tr.sync_to_model()
    model.eval()
    with torch.no_grad():
        y = model(bags.to(gpu), coords=coords.to(gpu), mask=None)          # deploy-time forward of the trained head still runs
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("task", ["regression", "survival"])
def test_trainer_regression_and_survival_losses(gpu, task):
    """dim_output = 1 heads trained with the reference's other two objectives (LitTileRegressor: L1, LitTileSurvival: Cox/Efron):
    the step's loss equals the objective evaluated on the HIP predictions, d(loss)/d(head bias) matches autograd of the same
    objective through the oracle network, and a few steps reduce it."""
    from oracle.mil_vit import mil_vit_forward
    from stamp_amd import losses
    from stamp_amd.mil import VisionTransformer
    from stamp_amd.mil_train import HipMilVitTrainer

    torch.manual_seed(11)
    Bb, Tn, Fd = 6, 120, 256
    model = VisionTransformer(dim_output=1, dim_input=Fd, dim_model=256, n_layers=1, n_heads=4, dim_feedforward=256, dropout=0.0, use_alibi=False)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    bags = torch.randn(Bb, Tn, Fd).half()
    if task == "regression":
        targets, fn = torch.randn(Bb, 1), losses.l1_loss
    else:
        targets = torch.stack([torch.tensor([5.0, 3.0, 3.0, 9.0, 1.0, 7.0]), torch.tensor([1.0, 1.0, 0.0, 1.0, 1.0, 0.0])], 1)   # a tie at t = 3
        fn = losses.cox_survival_loss
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_pred = mil_vit_forward(bags.float(), torch.zeros(Bb, Tn, 2), None, params, n_heads=4, use_alibi=False)
    ref_loss = fn(ref_pred, targets)
    ref_loss.backward()
    tr = HipMilVitTrainer(model, device=gpu, max_lr=2e-3, div_factor=25.0, total_steps=50, sched_interval="step", split_k=4, dropout=False)
    loss, pred = tr.step(bags.to(gpu), targets, update=False, loss_fn=fn)
    assert pred.shape == (Bb, 1)
    assert abs(loss.item() - fn(pred.cpu(), targets).item()) < 1e-5
    assert abs(loss.item() - ref_loss.item()) < 3e-2 * max(1.0, abs(ref_loss.item()))
    for k in ("mlp_head.0.bias", "mlp_head.0.weight", "transformer.norm.weight", "project_features.0.bias"):
        g, r = tr.g(k).cpu().double(), params[k].grad.double()
        if task == "survival" and k == "mlp_head.0.bias":      # the Cox partial likelihood is invariant to a common shift: true gradient 0
            assert r.norm() < 1e-6 and g.norm() < 1e-5
            continue
        assert ((g - r).norm() / (r.norm() + 1e-12)).item() < 6e-2, k
    ls = [tr.step(bags.to(gpu), targets, loss_fn=fn)[0].item() for _ in range(10)]
    assert ls[-1] < ls[0]


@pytest.mark.parametrize("alibi,dims,p_drop,bdt", [(False, (512, 512, 8, 512), 0.0, torch.float16), (False, (512, 512, 8, 512), 0.25, torch.float16),
                                                   (True, (512, 512, 8, 512), 0.0, torch.float16), (False, (456, 132, 4, 135), 0.1, torch.float32),
                                                   (True, (456, 132, 4, 135), 0.0, torch.float32), (False, (768, 256, 4, 512), 0.0, torch.bfloat16)])
def test_mil_vit_train_one_call_equals_the_kernel_by_kernel_chain(gpu, alibi, dims, p_drop, bdt):
    """amds_mil_vit_train_forward / _backward (one C call each) against the same kernels launched one by one from the host: logits, every
    parameter gradient and the gradient w.r.t. the bags are bit-identical -- default and odd (padded) shapes, dropout live (same counter-based
    masks from the same seed), ALiBi, every bag dtype; a second backward from the same saved activations (Jacobian rows) reproduces the first."""
    from chains import mil_vit as chain
    from stamp_amd import mil_core
    from stamp_amd.mil import VisionTransformer
    F, D, H, FF = dims
    torch.manual_seed(F + D + int(alibi))
    model = VisionTransformer(dim_output=3, dim_input=F, dim_model=D, n_layers=2, n_heads=H, dim_feedforward=FF, dropout=p_drop, use_alibi=alibi)
    sd = {k: v.detach().to(gpu, torch.float32) for k, v in model.state_dict().items()}
    pk = mil_core.PackedVit(model.dims, lambda n: sd[n], torch.bfloat16, train=True)
    Bb, Tn = 3, 157
    bags = torch.randn(Bb, Tn, F).to(bdt).to(gpu)
    coords = (torch.rand(Bb, Tn, 2) * 2000).to(gpu)
    dlogits = torch.randn(Bb, 3, device=gpu)
    training = p_drop > 0
    from stamp_amd import ops
    tail_was = ops.set_mil_cls_tail(False)          # the chain runs every row of the last block; the class-row tail has its own test below
    try:
        lg1, sv1 = mil_core.forward_train(pk, bags, coords, training=training, seed=1234)
        lg0, sv0 = chain.forward_train_stepwise(pk, bags, coords, training=training, seed=1234)
        assert torch.isfinite(lg1).all() and torch.equal(lg1, lg0)
        G1, db1 = mil_core.backward(pk, sv1, dlogits, need_params=True, need_bags=True)
        G0, db0 = chain.backward_stepwise(pk, sv0, dlogits, need_params=True, need_bags=True)
    finally:
        ops.set_mil_cls_tail(tail_was)
    assert set(G1) == set(G0) and set(G1) == {k for k in sd if not mil_core.is_buffer(k)}
    for k in G0:
        assert G1[k].shape == G0[k].shape == sd[k].shape, k
        assert torch.equal(G1[k], G0[k]), (k, (G1[k] - G0[k]).abs().max().item())
    assert db1.shape == (Bb, Tn, F) and torch.equal(db1, db0)
    ops.set_mil_cls_tail(False)
    try:
        G2, db2 = mil_core.backward(pk, sv1, 2.0 * dlogits, need_params=False, need_bags=True)      # input gradient only, saved activations untouched
        assert G2 == {} and torch.allclose(db2, 2.0 * db1, rtol=2e-2, atol=2e-2 * db1.abs().max().item())
        G3, _ = mil_core.backward(pk, sv1, dlogits, need_params=True, need_bags=False)
        assert all(torch.equal(G3[k], G1[k]) for k in G1)
        if training:        # another seed, other masks
            lg5, _ = mil_core.forward_train(pk, bags, coords, training=True, seed=99)
            assert not torch.equal(lg5, lg1)
    finally:
        ops.set_mil_cls_tail(tail_was)


@pytest.mark.parametrize("dims,p_drop,bdt,Tn", [((512, 512, 8, 512), 0.0, torch.float16, 157), ((512, 512, 8, 512), 0.25, torch.float16, 1024),
                                                ((456, 132, 4, 135), 0.1, torch.float32, 211), ((768, 256, 4, 512), 0.0, torch.bfloat16, 64)])
def test_mil_vit_train_class_row_tail_equals_the_full_last_block(gpu, dims, p_drop, bdt, Tn):
    """The head reads `x[:, 0]` behind the last block (reference vision_tranformer.py), so in that block only the class query's attention is computed
    (amds_attention_row_fwd_train / _bwd_train; the other rows of its output are zeros that carry no gradient).  Against the full block on the same inputs, seed and
    dropout counters: logits and EVERY gradient (parameters and bags) agree to the rounding of one attention row -- the two paths draw the same masks, so this holds
    with dropout live.  Stated tolerance: 8e-3 relative L2 per tensor (measured <= 4e-3; bf16 operands: the full path rounds the probabilities to bf16 in front of
    P V, the row kernel keeps them fp32 -- each path is itself ~7e-3 from the fp32 reference, tests above), logits 1e-2 absolute."""
    from stamp_amd import mil_core, ops
    from stamp_amd.mil import VisionTransformer
    F, D, H, FF = dims
    torch.manual_seed(F + D + Tn)
    model = VisionTransformer(dim_output=3, dim_input=F, dim_model=D, n_layers=2, n_heads=H, dim_feedforward=FF, dropout=p_drop, use_alibi=False)
    sd = {k: v.detach().to(gpu, torch.float32) for k, v in model.state_dict().items()}
    pk = mil_core.PackedVit(model.dims, lambda n: sd[n], torch.bfloat16, train=True)
    Bb = 3
    bags = torch.randn(Bb, Tn, F).to(bdt).to(gpu)
    dlogits = torch.randn(Bb, 3, device=gpu)
    res = {}
    was = ops.set_mil_cls_tail(True)
    try:
        for tail in (True, False):
            ops.set_mil_cls_tail(tail)
            lg, sv = mil_core.forward_train(pk, bags, None, training=p_drop > 0, seed=77)
            G, db = mil_core.backward(pk, sv, dlogits, need_params=True, need_bags=True)
            res[tail] = (lg.clone(), {k: v.clone() for k, v in G.items()}, db.clone())
    finally:
        ops.set_mil_cls_tail(was)
    (lt, Gt, dbt), (lf, Gf, dbf) = res[True], res[False]
    assert torch.isfinite(lt).all() and (lt - lf).abs().max().item() < 1e-2 * max(1.0, lf.abs().max().item())
    worst = sorted(((_rel2(Gt[k], Gf[k]), k) for k in Gf), reverse=True)
    print("class-row tail vs full last block, largest gradient differences:", [(round(a, 6), b) for a, b in worst[:4]], "bags", round(_rel2(dbt, dbf), 6))
    for r, k in worst:
        assert r < 8e-3, (k, r)
    assert _rel2(dbt, dbf) < 8e-3


@pytest.mark.parametrize("act", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("dims,p_drop,Tn,L", [((512, 512, 8, 512), 0.25, 1024, 2), ((456, 132, 4, 135), 0.1, 211, 2), ((256, 256, 4, 256), 0.0, 3, 1)])
def test_mil_vit_train_alibi_class_row_tail_equals_the_full_last_block(gpu, dims, p_drop, Tn, L, act):
    """The class-row tail of the ALiBi head (round 6): with mask = None -- every Lightning step, models/__init__.py:252 -- `alibi_mask` is None, so the class query's row is
    out_0 = sum_k (p_k - bias_scale |c_k| / running_mean) v_k with the class token at (0, 0) (vision_tranformer.py:60-72, 349-351): amds_attention_row_alibi_fwd_train /
    _bwd_train.  Against the full last block on the same inputs, seed and dropout counters: logits, EVERY gradient -- bias_scale of the last layer included, a sum of one
    -dO . U term per bag -- and the bag gradient.  Tolerance as the plain head's tail test (the blocked kernel rounds probabilities and distances to 16 bits in front of
    its MFMAs, the row kernel keeps them fp32): 8e-3 relative L2 (1.6e-2 for the 8 bias_scale scalars), + an absolute floor for numerically-zero tensors."""
    from stamp_amd import mil_core, ops
    from stamp_amd.mil import VisionTransformer
    F, D, H, FF = dims
    torch.manual_seed(F + D + Tn)
    model = VisionTransformer(dim_output=3, dim_input=F, dim_model=D, n_layers=L, n_heads=H, dim_feedforward=FF, dropout=p_drop, use_alibi=True)
    sd = {k: v.detach().to(gpu, torch.float32) for k, v in model.state_dict().items()}
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = sd[k] * 0 + 9000.0                   # the scale of the distances below (a trained scaler's state)
    pk = mil_core.PackedVit(model.dims, lambda n: sd[n], act, train=True)
    Bb = 3
    bags = torch.randn(Bb, Tn, F).half().to(gpu)
    coords = ((torch.rand(Bb, Tn, 2) * 2e4 / 256).round() * 256).to(gpu)
    dlogits = torch.randn(Bb, 3, device=gpu)
    res = {}
    was = ops.set_mil_cls_tail(True)
    try:
        for tail in (True, False):
            ops.set_mil_cls_tail(tail)
            lg, sv = mil_core.forward_train(pk, bags, coords, training=p_drop > 0, seed=77)
            G, db = mil_core.backward(pk, sv, dlogits, need_params=True, need_bags=True)
            res[tail] = (lg.clone(), {k: v.clone() for k, v in G.items()}, db.clone())
    finally:
        ops.set_mil_cls_tail(was)
    (lt, Gt, dbt), (lf, Gf, dbf) = res[True], res[False]
    assert torch.isfinite(lt).all() and (lt - lf).abs().max().item() < 1e-2 * max(1.0, lf.abs().max().item())
    # Judged as tests/test_gpu_mil_seam.py judges the bench-size step: key-encoder biases have a zero true gradient (a constant added to every score of a row) and
    # hold rounding noise on both paths; q / k encoder gradients are small differences of large terms and are measured against 5 % of the sibling value-encoder
    # gradient; the 8 bias_scale scalars of a layer are one vector.
    worst, bsg = [], {}
    for k in Gf:
        if "key_encoders" in k and k.endswith(".bias"):
            continue
        a, b = Gt[k].double().reshape(-1), Gf[k].double().reshape(-1)
        assert torch.isfinite(a).all(), k
        if k.endswith("bias_scale"):
            ga, gb = bsg.setdefault(k.split(".mhsa.")[0], ([], []))
            ga.append(a), gb.append(b)
            continue
        floor = 1e-5
        if "query_encoders" in k or "key_encoders" in k:
            floor = max(floor, 0.05 * Gf[k.replace("query_encoders", "value_encoders").replace("key_encoders", "value_encoders")].double().norm().item())
        worst.append(((a - b).norm().item() / max(b.norm().item(), floor), k))
    for layer, (ga, gb) in bsg.items():
        worst.append((_rel2(torch.cat(ga), torch.cat(gb)), layer + ".mhsa.attentions.*.bias_scale"))
    worst.sort(reverse=True)
    print("ALiBi class-row tail vs full last block, largest gradient differences:", [(round(a, 6), b) for a, b in worst[:4]], "bags", round(_rel2(dbt, dbf), 6))
    for r, k in worst:
        assert r < (1.6e-2 if "bias_scale" in k else 8e-3), (k, r)
    assert (dbt.double() - dbf.double()).norm().item() < 8e-3 * dbf.double().norm().item() + 1e-5


@pytest.mark.parametrize("L,Bb,Tn,p_drop", [(1, 1, 1, 0.0), (1, 2, 300, 0.25), (3, 2, 65, 0.1), (2, 5, 2, 0.0)])
def test_mil_vit_class_row_tail_edge_shapes(gpu, L, Bb, Tn, p_drop):
    """The class-row tail on the shapes its row pitches could get wrong: a single block (the tail IS the model), one bag, one tile per bag (S = 2), three blocks; training
    step and deploy forward, tail against full block (logits 1e-2; gradients 2e-2 relative L2 -- a dozen token rows average the bf16 rounding of the two paths less than
    65 600 do; a wrong pitch or mask index gives O(1) -- + an absolute floor for tensors that are numerically zero); and the input
    gradient alone (`need_params=False`: the heat-map path)."""
    from stamp_amd import mil_core, ops
    from stamp_amd.mil import VisionTransformer
    torch.manual_seed(L * 100 + Tn)
    model = VisionTransformer(dim_output=2, dim_input=256, dim_model=256, n_layers=L, n_heads=4, dim_feedforward=256, dropout=p_drop, use_alibi=False).eval()
    sd = {k: v.detach().to(gpu, torch.float32) for k, v in model.state_dict().items()}
    pk = mil_core.PackedVit(model.dims, lambda n: sd[n], torch.bfloat16, train=True)
    bags = torch.randn(Bb, Tn, 256).half().to(gpu)
    dlogits = torch.randn(Bb, 2, device=gpu)
    res = {}
    was = ops.set_mil_cls_tail(True)
    try:
        for tail in (True, False):
            ops.set_mil_cls_tail(tail)
            lg, sv = mil_core.forward_train(pk, bags, None, training=p_drop > 0, seed=5)
            G, db = mil_core.backward(pk, sv, dlogits, need_params=True, need_bags=True)
            _, db_only = mil_core.backward(pk, sv, dlogits, need_params=False, need_bags=True)
            with torch.no_grad():
                dep = model.to(gpu)(bags, coords=None, mask=None)
            res[tail] = (lg.clone(), {k: v.clone() for k, v in G.items()}, db.clone(), db_only.clone(), dep.clone())
    finally:
        ops.set_mil_cls_tail(was)
    (lt, Gt, dbt, dbot, dept), (lf, Gf, dbf, dbof, depf) = res[True], res[False]
    assert torch.isfinite(lt).all() and torch.isfinite(dept).all()
    assert (lt - lf).abs().max().item() < 1e-2 * max(1.0, lf.abs().max().item()) and (dept - depf).abs().max().item() < 1e-2 * max(1.0, depf.abs().max().item())
    for k in Gf:
        assert torch.isfinite(Gt[k]).all(), k
        err = (Gt[k].double() - Gf[k].double()).norm().item()
        assert err < 2e-2 * Gf[k].double().norm().item() + 1e-5, (k, err, Gf[k].norm().item())
    assert (dbt.double() - dbf.double()).norm().item() < 2e-2 * dbf.double().norm().item() + 1e-5
    assert (dbot.double() - dbof.double()).norm().item() < 2e-2 * dbof.double().norm().item() + 1e-5


def _rel2(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def test_mil_vit_train_c_abi_guards(gpu):
    import ctypes as C
    lib = _lib.lib()
    cfg32 = _lib.MilVitCfg(512, 512, 8, 512, 2, 2, 0, _lib.F32)           # the step's 16-bit tensors are bf16 ("medium") or fp16 ("high"), nothing else
    assert lib.amds_mil_vit_train_saved_bytes(C.byref(cfg32), 2, 64) == 0 and b"bf16 or fp16" in lib.amds_last_error()
    cfg16 = _lib.MilVitCfg(512, 512, 8, 512, 2, 2, 0, _lib.F16)
    assert lib.amds_mil_vit_train_saved_bytes(C.byref(cfg16), 2, 64) > 0
    cfg = _lib.MilVitCfg(512, 512, 8, 512, 2, 2, 0, _lib.BF16)
    n_saved, n_ws = lib.amds_mil_vit_train_saved_bytes(C.byref(cfg), 2, 64), lib.amds_mil_vit_train_workspace_bytes(C.byref(cfg), 2, 64, 32)
    assert n_saved > 0 and n_ws > 0
    assert lib.amds_mil_vit_train_workspace_bytes(C.byref(cfg), 2, 64, 0) == 0
    from stamp_amd import mil_core
    from stamp_amd.mil import VisionTransformer
    model = VisionTransformer(dim_output=2, dim_input=512, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.0, use_alibi=False)
    sd = {k: v.detach().to(gpu, torch.float32) for k, v in model.state_dict().items()}
    pk = mil_core.PackedVit(model.dims, lambda n: sd[n], torch.bfloat16, train=True)
    cfg, wc = pk.c_structs()
    bags = torch.randn(2, 64, 512, device=gpu).half()
    out = torch.zeros(2, 2, device=gpu)
    small = torch.empty(4096, dtype=torch.uint8, device=gpu)
    drop = _lib.MilVitDropout(0.0, 0.0, 0.0, 0, -1)
    rc = lib.amds_mil_vit_train_forward(C.byref(cfg), C.byref(wc), bags.data_ptr(), _lib.F16, None, C.byref(drop), out.data_ptr(), 2, 64, small.data_ptr(), small.numel(), None)
    assert rc == -2 and b"arena" in lib.amds_last_error()
    with pytest.raises(RuntimeError, match="training pack"):
        mil_core.forward_train(model._infer_pack(torch.device(gpu)), bags, None, training=False)


def test_fit_driven_from_a_directory_of_feature_files(gpu, tmp_path):
    """`stamp train`'s data path end to end (reference modeling/train.py:455-477, 504-621): feature .h5 files on disk -> stamp_amd.bags
    (patients -> fixed-size bags, shuffled batches for training; full bags, batch 1 for validation; inverse-frequency class weights) ->
    `fit` around the HIP training step.  Two separable classes: the validation loss must fall, the best epoch's weights end up in the module.
    And: the fixed-size bag gathered on the GPU (`BagDataset(device=...)`, amds_gather_rows) equals the host item bit for bit."""
    import numpy as np

    from stamp_amd import bags as B
    from stamp_amd import h5io
    from stamp_amd.mil import VisionTransformer
    from stamp_amd.mil_train import HipMilVitTrainer, fit
    rng = np.random.default_rng(0)
    Fd, pdata = 64, []
    for i in range(24):
        label = "pos" if i % 3 else "neg"
        files = []
        for s in range(1 + i % 2):                                                  # one or two slides per patient
            n = int(rng.integers(20, 90))
            feats = (rng.standard_normal((n, Fd)) + (1.5 if label == "pos" else -1.5)).astype(np.float16)
            coords = np.stack([rng.integers(0, 50, n), rng.integers(0, 50, n)], 1).astype(np.float32) * 256.0
            p = tmp_path / f"p{i}_s{s}.h5"
            h5io.write_tile_features(p, feats, coords, extractor="test", tile_size_um=256.0, tile_size_px=224, code_hash="0", stamp_version="2.4.0")
            files.append(p)
        pdata.append(B.PatientData(ground_truth=label, feature_files=files))
    train, valid = pdata[:18], pdata[18:]
    torch.manual_seed(0)
    dl_t, cats = B.tile_bag_dataloader(patient_data=train, bag_size=32, task="classification", batch_size=6, shuffle=True, num_workers=0, transform=None)
    dl_v, _ = B.tile_bag_dataloader(patient_data=valid, bag_size=None, task="classification", categories=cats, batch_size=1, shuffle=False, num_workers=0, transform=None)
    assert cats == ["neg", "pos"]
    w = B.class_weights(dl_t.dataset.ground_truths, cats)
    assert torch.allclose(w, torch.tensor([2 / 3, 1 / 3]), atol=1e-6)                # 6 neg, 12 pos -> weights (N/6, N/12) normalised
    model = VisionTransformer(dim_output=2, dim_input=Fd, dim_model=64, n_layers=1, n_heads=2, dim_feedforward=64, dropout=0.0, use_alibi=False)
    tr = HipMilVitTrainer(model, device=gpu, max_lr=3e-3, total_steps=4 * len(dl_t), sched_interval="step", dropout=False)
    hist = fit(tr, lambda: dl_t, lambda: dl_v, max_epochs=4, patience=8, class_weights=w)
    assert all(np.isfinite(hist["validation_loss"])) and hist["validation_loss"][-1] < 0.5 * hist["validation_loss"][0], hist
    # GPU-side gather of the fixed-size bag == the host item
    ds_h = B.BagDataset(bags=[p.feature_files for p in train], bag_size=16, ground_truths=dl_t.dataset.ground_truths, transform=None, deterministic=True)
    ds_d = B.BagDataset(bags=[p.feature_files for p in train], bag_size=16, ground_truths=dl_t.dataset.ground_truths, transform=None, deterministic=True, device=gpu)
    for i in (0, 5, 17):
        (bh, ch, nh, _), (bd, cd, nd, _) = ds_h[i], ds_d[i]
        assert bd.is_cuda and torch.equal(bd.cpu(), bh) and torch.equal(cd.cpu(), ch) and nh == nd


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tokens,N,K,split_k,pitch", [(2048, 256, 256, 1, 0), (65600, 512, 512, 32, 0), (1000, 256, 512, 4, 64), (77, 512, 256, 2, 0), (4100, 768, 256, 8, 8),
                                                      (3000, 256, 1536, 3, 0)])
def test_wgrad_tn_token_major_operands(gpu, dt, tokens, N, K, split_k, pitch):
    """amds_wgrad_tn (kernel id 15: LDS-DMA of 64-token row tiles + ds_read_b64_tr_b16 fragments) against fp64 dy^T x on the same 16-bit operands, and
    BIT-identical to the form it replaces (two transposes with zeroed pad columns + the split-K batched GEMM + the same column sum): token counts that
    are no multiple of 64 or of the split, pitched row views, one split, every tile position."""
    g = torch.Generator().manual_seed(tokens + N + K)
    dyb = (torch.randn(tokens, N + pitch, generator=g) * 0.5).to(gpu, dt)
    xb = (torch.randn(tokens, K + pitch, generator=g) * 0.5).to(gpu, dt)
    dy, x = dyb[:, :N], xb[:, pitch:pitch + K] if pitch % 8 == 0 else xb[:, :K]
    got = T.wgrad_tn(dy, x, split_k)
    ref = dy.double().t() @ x.double()
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6 * max(1.0, (tokens / 1000) ** 0.5), err                  # fp32 accumulation of exact products
    # the replaced form: transposes (pad columns zero) -> split-K batched GEMM -> column sum
    unit = 64 * split_k
    Mp = (tokens + unit - 1) // unit * unit
    dyT, xT = T.transpose16(dy.contiguous(), Mp), T.transpose16(x.contiguous(), Mp)
    chunk = Mp // split_k
    part = torch.empty(split_k, N, K, dtype=torch.float32, device=gpu)
    _lib.check(_lib.lib().amds_gemm_batched(dyT.data_ptr(), Mp, chunk, xT.data_ptr(), Mp, chunk, N, K, chunk, split_k, ops.act_code(dt), _lib.EPI_BIAS_F32,
                                            part.data_ptr(), K, N * K, None, 1.0, None), "gemm_batched")
    old = T.colsum(part.view(split_k, N * K)).view(N, K)
    assert torch.equal(got, old)
    assert torch.equal(got, T.wgrad_tn(dy, x, split_k))


@pytest.mark.parametrize("rows,D,Dp,p", [(1000, 512, 512, 0.5), (333, 256, 256, 0.0), (70, 192, 256, 0.25)])
def test_layernorm_kernels_that_carry_a_second_output(gpu, rows, D, Dp, p):
    """amds_layernorm_train_copy = amds_layernorm_train + a device-to-device copy of the (pitched) rows; amds_layernorm_bwd_cast = amds_layernorm_bwd +
    amds_dropout_cast_bwd (p > 0) / amds_cast_pad (p = 0) of the dx it produced -- bit for bit, including the skip add and pad columns of the copy."""
    import ctypes as C
    lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(rows + D)
    x = torch.zeros(rows, Dp, device=gpu)
    x[:, :D] = torch.randn(rows, D, generator=g).to(gpu) * 2 + 0.3
    x[:, D:] = 7.0                                                       # pad columns travel with the copy
    gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).to(gpu), (0.1 * torch.randn(D, generator=g)).to(gpu)
    y0, m0, r0 = T.layernorm_train(x, gamma, beta, 1e-5, torch.bfloat16, rows=rows, row_stride=Dp, out=torch.zeros(rows, Dp, dtype=torch.bfloat16, device=gpu), ld_out=Dp)
    y1 = torch.zeros(rows, Dp, dtype=torch.bfloat16, device=gpu)
    m1, r1, xc = torch.empty(rows, device=gpu), torch.empty(rows, device=gpu), torch.full((rows, Dp), -1.0, device=gpu)
    _lib.check(lib.amds_layernorm_train_copy(x.data_ptr(), Dp, gamma.data_ptr(), beta.data_ptr(), y1.data_ptr(), Dp, m1.data_ptr(), r1.data_ptr(), rows, D, 1e-5,
                                             _lib.BF16, xc.data_ptr(), Dp, Dp, st), "ln_train_copy")
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1) and torch.equal(xc, x)
    # backward: dx (with skip) + its bf16 (dropout-masked) copy
    dy = torch.randn(rows, D, generator=g).to(gpu)
    skip = torch.randn(rows, D, generator=g).to(gpu)
    dg0, db0, dg1, db1 = (torch.empty(D, device=gpu) for _ in range(4))
    dx0 = T.layernorm_bwd(dy, x, m0, r0, gamma, skip.clone(), True, dg0, db0, rows=rows, x_stride=Dp)
    ref16 = torch.empty(rows, D, dtype=torch.bfloat16, device=gpu)
    if p > 0:
        _lib.check(lib.amds_dropout_cast_bwd(dx0.data_ptr(), D, ref16.data_ptr(), D, rows, D, _lib.BF16, p, 77, 13, st), "dropout_cast_bwd")
    else:
        ref16 = ops.cast_pad(dx0, D, torch.bfloat16)
    dx1, got16 = skip.clone(), torch.empty(rows, D, dtype=torch.bfloat16, device=gpu)
    nb = lib.amds_layernorm_bwd_workspace_bytes(rows, D)
    ws = torch.empty(nb, dtype=torch.uint8, device=gpu)
    _lib.check(lib.amds_layernorm_bwd_cast(dy.data_ptr(), D, x.data_ptr(), Dp, m0.data_ptr(), r0.data_ptr(), gamma.data_ptr(), dx1.data_ptr(), D, 1, dg1.data_ptr(),
                                           db1.data_ptr(), 0, rows, D, ws.data_ptr(), nb, got16.data_ptr(), D, p, 77, 13, st), "ln_bwd_cast")
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1) and torch.equal(ref16.view(torch.int16), got16.view(torch.int16))
