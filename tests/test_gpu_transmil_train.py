"""TransMIL training on the HIP path against autograd through the oracle (oracle/transmil.py, pinned to the reference's goldens):
loss, logits, every parameter gradient, d/d(bags); the module under torch AdamW; the one dropout site with the kernels' own mask."""
import pytest
import torch

from oracle.transmil import transmil_forward
from stamp_amd import _lib
from stamp_amd import train_ops as T
from stamp_amd.mil import TransMIL

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _setup(Bb, Tn, Fd, Cd, C, seed):
    torch.manual_seed(seed)
    model = TransMIL(dim_output=C, dim_input=Fd, dim_hidden=Cd)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    bags = torch.randn(Bb, Tn, Fd).half().float()
    targets = torch.nn.functional.one_hot(torch.arange(Bb) % C, C).float()
    return model, bags, targets


def _oracle(model, bags, targets, drop=None):
    params = {k: v.detach().clone().double().requires_grad_(True) for k, v in model.state_dict().items()}
    x = bags.double().requires_grad_(True)
    logits = transmil_forward(x, params, drop=drop, dtype=torch.float64)
    loss = torch.nn.functional.cross_entropy(logits, targets.double())
    loss.backward()
    return loss.item(), logits.detach(), {k: v.grad for k, v in params.items()}, x.grad


@pytest.mark.parametrize("Bb,Tn,Fd,Cd", [(2, 50, 96, 64), (3, 300, 128, 128), (2, 1024, 1024, 512)])
def test_transmil_backward_matches_autograd(gpu, Bb, Tn, Fd, Cd):
    """Shapes: 50 tiles (8 x 8 grid with 14 wrap-padded tiles, n = 65 < one landmark block), 300 tiles (18 x 18, wrap 24, front padding
    59), and the bench geometry 1024 tiles x 1024-d, hidden 512 (32 x 32 grid, n = 1025 -> padded 1280, 5 tokens per landmark).
    Stated tolerance (fp32 arithmetic here, fp64 autograd there, six cubic pinv iterations in between): every gradient <= 1e-4 relative L2
    (measured on the MI355X: <= 3e-6 at all three sizes)."""
    model, bags, targets = _setup(Bb, Tn, Fd, Cd, 2, seed=Tn)
    ref_loss, ref_logits, ref_g, ref_dx = _oracle(model, bags, targets)
    model = model.to(gpu).eval()                       # eval: no dropout; gradients still flow (the heat-map use)
    x = bags.to(gpu).requires_grad_(True)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, targets.to(gpu))
    loss.backward()
    assert abs(loss.item() - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    assert (logits.detach().cpu().double() - ref_logits).abs().max() < 2e-3 * max(1.0, ref_logits.abs().max().item())
    worst = []
    for n, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n
        worst.append((_rel(p.grad.cpu(), ref_g[n]), n))
    worst.append((_rel(x.grad.cpu(), ref_dx), "bags"))
    worst.sort(reverse=True)
    print(f"TransMIL {Bb}x{Tn}x{Fd} hidden {Cd}: largest gradient errors", [(round(a, 6), b) for a, b in worst[:5]])
    for rel, n in worst:
        assert rel < 1e-4, (n, rel)
    with torch.no_grad():                              # the inference path gives the same logits
        assert (model(bags.to(gpu)) - logits.detach()).abs().max() < 1e-4


def test_transmil_train_mode_dropout_and_optimizer(gpu):
    """.train(): Dropout(0.1) on both `to_out` outputs, masks regenerated in the backward; against the oracle fed with the kernels' masks.
    Then torch AdamW on the module's parameters reduces the loss."""
    Bb, Tn, Fd, Cd, C = 3, 120, 128, 128, 2
    model, bags, targets = _setup(Bb, Tn, Fd, Cd, C, seed=7)
    model = model.to(gpu).train()
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # the draw the module will make
    torch.manual_seed(123)
    logits = model(bags.to(gpu))
    loss = torch.nn.functional.cross_entropy(logits, targets.to(gpu))
    loss.backward()
    import math
    n = 1 + math.ceil(math.sqrt(Tn)) ** 2
    ks = _lib.lib().amds_dropout_keep_scale(0.1)
    drop = {name: T.dropout_mask(Bb * n * Cd, 0.1, seed, sid, gpu).view(Bb, n, Cd).cpu().double() * ks for name, sid in (("layer1", 1), ("layer2", 2))}
    assert abs(drop["layer1"].gt(0).double().mean().item() - 0.9) < 0.02
    ref_loss, ref_logits, ref_g, _ = _oracle(model.cpu(), bags, targets, drop=drop)
    model = model.to(gpu)
    assert abs(loss.item() - ref_loss) < 1e-4 * max(1.0, abs(ref_loss)), (loss.item(), ref_loss)
    for nme, p in model.named_parameters():
        assert _rel(p.grad.cpu(), ref_g[nme]) < 1e-4, nme
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        l = torch.nn.functional.cross_entropy(model(bags.to(gpu)), targets.to(gpu))
        l.backward()
        opt.step()
        losses.append(l.item())
    assert sum(losses[-3:]) < sum(losses[:3])


@pytest.mark.parametrize("Z,M,N,K", [(6, 256, 64, 1280), (3, 130, 70, 52), (1, 512, 1024, 1024), (16, 256, 256, 256), (4, 64, 512, 64), (2, 1280, 256, 64)])
@pytest.mark.parametrize("transa,transb", [(False, False), (False, True), (True, False), (True, True)])
def test_bgemm_f32_high_precision_mode(gpu, Z, M, N, K, transa, transb):
    """`torch.set_float32_matmul_precision("high")` (the reference's training setting, train.py:519; the library follows torch's flag): fp32 operands as hi + lo bf16, three bf16 MFMAs per product.
    Stated tolerance: 2e-5 relative L2 against the fp64 product (measured ~2e-6: 16+ mantissa bits per factor; TF32, torch's other form of "high", gives
    ~3e-4) on every tile shape and operand layout; values spanning 2^+-20 (the split has the exponent range of fp32, unlike an fp16 split); alpha / diag /
    accumulate epilogue unchanged; the level is restored by the context manager and "highest" stays bit-identical to the default."""
    from stamp_amd import ops
    from stamp_amd import transmil_core as tc
    g = torch.Generator().manual_seed(Z * 7 + M + K)
    A = (torch.randn(Z, K, M, generator=g) if transa else torch.randn(Z, M, K, generator=g))
    B = (torch.randn(Z, N, K, generator=g) if transb else torch.randn(Z, K, N, generator=g))
    A = (A * torch.exp2(torch.randint(-20, 21, (Z, 1, 1), generator=g).float())).to(gpu)
    B = B.to(gpu)
    Ad = A.double().transpose(1, 2) if transa else A.double()
    want = Ad @ (B.double().transpose(1, 2) if transb else B.double())
    exact = tc._mm(A, B, transb, transa=transa)
    assert torch.get_float32_matmul_precision() == "highest"
    with ops.float32_matmul_precision("high"):
        assert torch.get_float32_matmul_precision() == "high"
        out = tc._mm(A, B, transb, transa=transa)
        assert torch.equal(out, tc._mm(A, B, transb, transa=transa))          # deterministic
    assert torch.get_float32_matmul_precision() == "highest"
    assert torch.equal(exact, tc._mm(A, B, transb, transa=transa))
    for z in range(Z):                                                         # per batch: the batches differ by 2^40 in scale
        e = ((out[z].double() - want[z]).norm() / want[z].norm()).item()
        assert e < 2e-5, (z, e)
        assert ((exact[z].double() - want[z]).norm() / want[z].norm()).item() < 2e-6


@pytest.mark.parametrize("Z,M,N,K,transa,transb", [(7, 256, 256, 256, False, False), (3, 130, 70, 52, False, False), (2, 40, 40, 40, False, False),
                                                     (4, 256, 256, 64, False, True), (2, 128, 384, 96, True, False)])
@pytest.mark.parametrize("precision", ["highest", "high"])
def test_bgemm_f32_dual_is_two_products_bit_for_bit(gpu, Z, M, N, K, transa, transb, precision):
    """amds_bgemm_f32_dual (the pinv iteration's `xz = x @ z` and `7 I - xz` from one pass over the operands, reference trans_mil.py:31-33) against two
    amds_bgemm_f32 calls: the one-launch form (plain 128 x 128 tiles, both precisions), ragged tiles, the small-product fallback and the layouts that take two
    launches -- every output bit-identical, diagonal term included (square and non-square)."""
    from stamp_amd import ops
    from stamp_amd import transmil_core as tc
    g = torch.Generator().manual_seed(Z + M + K)
    A = (torch.randn(Z, K, M, generator=g) if transa else torch.randn(Z, M, K, generator=g)).to(gpu)
    B = (torch.randn(Z, N, K, generator=g) if transb else torch.randn(Z, K, N, generator=g)).to(gpu)
    with ops.float32_matmul_precision(precision):
        c1, c2 = tc._mm_dual(A, B, 1.0, 0.0, -1.0, 7.0, transb=transb, transa=transa)
        w1 = tc._mm(A, B, transb, transa=transa)
        w2 = tc._mm(A, B, transb, alpha=-1.0, diag=7.0, transa=transa)
    assert torch.equal(c1, w1) and torch.equal(c2, w2)
    Ad = A.double().transpose(1, 2) if transa else A.double()
    want = Ad @ (B.double().transpose(1, 2) if transb else B.double())
    eye = torch.eye(M, N, dtype=torch.float64, device=gpu)
    assert ((c2.double() - (7.0 * eye - want)).norm() / want.norm()).item() < 2e-5


@pytest.mark.parametrize("Bb,Tn,Fd,Cd", [(3, 300, 128, 128), (2, 1024, 1024, 512)])
def test_transmil_backward_high_precision_mode(gpu, Bb, Tn, Fd, Cd):
    """The TransMIL training step with the reference's own matmul setting ("high", train.py:519): logits and every gradient against fp64 autograd.
    Stated tolerance 2e-4 relative L2 per gradient (measured <= 3.5e-5; six cubic pinv iterations amplify the products' 2^-17 rounding; TF32 -- what the reference gets from
    the same setting on an NVIDIA part -- would be 30x coarser per product)."""
    from stamp_amd import ops
    model, bags, targets = _setup(Bb, Tn, Fd, Cd, 2, seed=Tn)
    ref_loss, ref_logits, ref_g, ref_dx = _oracle(model, bags, targets)
    model = model.to(gpu).eval()
    x = bags.to(gpu).requires_grad_(True)
    with ops.float32_matmul_precision("high"):
        logits = model(x)
        loss = torch.nn.functional.cross_entropy(logits, targets.to(gpu))
        loss.backward()
    assert abs(loss.item() - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))
    assert (logits.detach().cpu().double() - ref_logits).abs().max() < 2e-3 * max(1.0, ref_logits.abs().max().item())
    worst = sorted([(_rel(p.grad.cpu(), ref_g[n]), n) for n, p in model.named_parameters()] + [(_rel(x.grad.cpu(), ref_dx), "bags")], reverse=True)
    print(f"TransMIL high {Bb}x{Tn}x{Fd} hidden {Cd}: largest gradient errors", [(round(a, 7), b) for a, b in worst[:5]])
    for rel, n in worst:
        assert rel < 2e-4, (n, rel)


@pytest.mark.parametrize("Z,M,N,K", [(6, 256, 64, 1280), (3, 130, 70, 50), (1, 512, 1024, 1024)])
@pytest.mark.parametrize("transb", [False, True])
def test_bgemm_f32_transposed_a_and_split_k_wgrad(gpu, Z, M, N, K, transb):
    """amds_bgemm_f32 with A stored [K][M] (transb bit 1: the backward's x^T dy products without an explicit transpose), on shapes that take
    the vector loads and on shapes that do not; and transmil_core._wgrad (per-bag products + fixed-order sum) against one fp64 product."""
    from stamp_amd import transmil_core as tc
    g = torch.Generator().manual_seed(Z * 7 + M)
    A = torch.randn(Z, K, M, generator=g).to(gpu)
    B = (torch.randn(Z, N, K, generator=g) if transb else torch.randn(Z, K, N, generator=g)).to(gpu)
    out = tc._mm(A, B, transb, transa=True)
    want = A.double().transpose(1, 2) @ (B.double().transpose(1, 2) if transb else B.double())
    assert out.shape == (Z, M, N)
    assert ((out.double() - want).norm() / want.norm()).item() < 2e-6
    if not transb:
        gw = tc._wgrad(A, B)                                    # sum_z A_z^T B_z
        assert ((gw.double() - want.sum(0)).norm() / want.sum(0).norm()).item() < 2e-6
        assert torch.equal(gw, tc._wgrad(A, B))                  # fixed summation order: bit-reproducible


@pytest.mark.parametrize("Bb,Tn,Fd,Cd,train", [(2, 50, 96, 64, False), (3, 300, 128, 128, True), (2, 1024, 256, 512, True), (1, 255, 64, 256, False)])
def test_transmil_train_calls_equal_the_kernel_by_kernel_chain(gpu, Bb, Tn, Fd, Cd, train):
    """amds_transmil_train_forward / _backward (the whole step, one C call each) and amds_nystrom_attn_fwd / _bwd (one call per layer and direction under a
    host loop) against the same kernels launched one by one from the host: logits, every parameter gradient and d/d(bags) bit-identical -- with and
    without front padding (n = 65 < one landmark block; n = 1025 -> 1280; n = 257 on 128 landmarks -> 384), wrap padding, dropout live (same
    counter-based masks from the same seed) and off; a second, input-gradient-only backward from the same saved activations."""
    from stamp_amd import ops
    from stamp_amd import transmil_core as tc
    model, bags, targets = _setup(Bb, Tn, Fd, Cd, 2, seed=Tn + 1)
    model = model.to(gpu)
    get = model._get(torch.device(gpu))
    dlogits = torch.randn(Bb, 2, device=gpu)
    res = []
    from chains import transmil as chain
    tail_was = ops.set_mil_cls_tail(False)          # the chains run every row of layer2; the class-row tail has its own test below
    for level in (0, 1, 2):        # 0 = the whole-step C calls; 1 = host loop around amds_nystrom_attn_fwd / _bwd; 2 = every kernel from the host (tests/chains)
        fwd, bwd = (tc.forward_train, tc.backward) if level == 0 else (chain.forward_train_stepwise, chain.backward_stepwise)
        chain.NYSTROM_KERNEL_BY_KERNEL = level == 2
        try:
            logits, saved = fwd(get, bags.to(gpu), (Fd, Cd, 2), training=train, seed=4321)
            G, db = bwd(saved, dlogits, need_params=True, need_bags=True)
            G2, db2 = bwd(saved, dlogits, need_params=False, need_bags=True)        # input gradient only, saved activations untouched
        finally:
            chain.NYSTROM_KERNEL_BY_KERNEL = False
            if level == 2:
                ops.set_mil_cls_tail(tail_was)
        assert G2 == {} and torch.equal(db2, db)
        res.append((logits, G, db))
    l0, G0, d0 = res[2]
    names = {n: p.shape for n, p in model.named_parameters()}
    for l1, G1, d1 in res[:2]:
        assert torch.isfinite(l1).all() and torch.equal(l1, l0) and torch.equal(d1, d0)
        assert set(G1) == set(G0) == set(names)
        for k in G0:
            assert G1[k].shape == G0[k].shape == names[k], k
            assert torch.equal(G1[k], G0[k]), (k, (G1[k] - G0[k]).abs().max().item())
    model.train()                                   # the nn.Module under autograd runs on the same calls
    x = bags.to(gpu).requires_grad_(True)
    torch.nn.functional.cross_entropy(model(x), targets.to(gpu)).backward()
    assert x.grad is not None and all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("Bb,Tn,Fd,Cd,train", [(2, 50, 96, 64, True), (3, 300, 128, 128, True), (2, 1024, 256, 512, True), (1, 255, 64, 256, False)])
def test_transmil_train_class_row_tail_equals_every_row(gpu, Bb, Tn, Fd, Cd, train):
    """cfg.train_cls_tail (default from the context, amds_set_mil_cls_tail): layer2's attention output (a1 z a3 v on one query row), residual convolution, to_out,
    Dropout and the backward of all of them on the class rows alone, against the same step on every row -- same dropout seed, so the class rows draw the same
    mask.  Same mathematics, reordered fp32 sums (rank-1 products instead of n-row ones): logits <= 2e-6 abs, every gradient <= 2e-5 relative L2."""
    from stamp_amd import ops
    from stamp_amd import transmil_core as tc
    model, bags, _ = _setup(Bb, Tn, Fd, Cd, 2, seed=Tn + 3)
    model = model.to(gpu)
    get = model._get(torch.device(gpu))
    dlogits = torch.randn(Bb, 2, device=gpu)
    res = []
    was = ops.set_mil_cls_tail(True)
    try:
        for tail in (False, True):
            ops.set_mil_cls_tail(tail)
            logits, saved = tc.forward_train(get, bags.to(gpu), (Fd, Cd, 2), training=train, seed=99)
            assert saved["cfg"].train_cls_tail == int(tail)
            ops.set_mil_cls_tail(not tail)           # the backward follows the forward's cfg, not the context of the moment
            G, db = tc.backward(saved, dlogits, need_params=True, need_bags=True)
            res.append((logits, G, db))
    finally:
        ops.set_mil_cls_tail(was)
    (l0, G0, d0), (l1, G1, d1) = res
    assert (l1 - l0).abs().max().item() < 2e-6 * max(1.0, l0.abs().max().item())
    worst = sorted([(_rel(G1[k], G0[k]), k) for k in G0] + [(_rel(d1, d0), "bags")], reverse=True)
    print(f"TransMIL tail vs full {Bb}x{Tn}x{Fd} hidden {Cd}:", [(float(f"{a:.2e}"), b) for a, b in worst[:4]])
    for rel, k in worst:
        assert rel < 2e-5, (k, rel)


def test_nystrom_c_abi_guards(gpu):
    import ctypes as C
    lib = _lib.lib()
    assert lib.amds_nystrom_attn_saved_bytes(100, 2, 65) == 0 and b"multiple of 8" in lib.amds_last_error()
    assert lib.amds_nystrom_attn_saved_bytes(64, 2, 65) > 0 and lib.amds_nystrom_attn_workspace_bytes(64, 2, 65) > 0
    w = [torch.zeros(n, device=gpu) for n in (64, 64, 3 * 64 * 64, 64 * 64, 64, 8 * 33)]
    L = _lib.TransMilLayer(*[t.data_ptr() for t in w])
    y = torch.zeros(2, 65, 64, device=gpu)
    small = torch.empty(1024, dtype=torch.uint8, device=gpu)
    rc = lib.amds_nystrom_attn_fwd(C.byref(L), 64, y.data_ptr(), y.data_ptr(), 2, 65, 0.0, 0, 0, small.data_ptr(), small.numel(), None)
    assert rc == -2 and b"arena" in lib.amds_last_error()
