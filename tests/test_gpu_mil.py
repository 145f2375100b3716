"""MIL `vit` head forward on the HIP path vs the oracle (which is pinned to the reference's goldens)."""
import pytest
import torch

from oracle.mil_vit import mil_vit_forward
from stamp_amd.mil import VisionTransformer

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("Bb,T,F,C", [(2, 1024, 1024, 2), (3, 77, 768, 3), (1, 3000, 1024, 2), (2, 1, 512, 4)])
def test_mil_vit_forward_matches_oracle(gpu, Bb, T, F, C):
    torch.manual_seed(Bb * 1000 + T)
    model = VisionTransformer(dim_output=C, dim_input=F, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512,
                              dropout=0.0, use_alibi=False).eval()
    with torch.no_grad():          # make biases / norms non-trivial
        for n, p in model.named_parameters():
            if p.dim() == 1 and "class_token" not in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = model.state_dict()
    bags = torch.randn(Bb, T, F).half()              # features are fp16 on disk (reference preprocessing/__init__.py:325)
    coords = torch.rand(Bb, T, 2) * 1000
    ref = mil_vit_forward(bags.float(), coords, None, sd, n_heads=8, use_alibi=False)
    with torch.no_grad():
        out = model(bags.to(gpu), coords=coords.to(gpu), mask=None)
    assert out.shape == (Bb, C) and out.dtype == torch.float32
    err = (out.cpu() - ref).abs().max().item()
    assert err < 5e-3 * max(1.0, ref.abs().max().item()), (err, ref)
    with torch.no_grad():
        assert torch.equal(out, model(bags.to(gpu), coords=coords.to(gpu), mask=None))   # deterministic


def test_mil_vit_state_dict_roundtrip_and_guards(gpu):
    kw = dict(dim_output=2, dim_input=256, dim_model=128, n_layers=1, n_heads=2, dim_feedforward=128, dropout=0.0, use_alibi=False)
    m1, m2 = VisionTransformer(**kw).eval(), VisionTransformer(**kw).eval()
    m2.load_state_dict(m1.state_dict())
    assert set(m1.state_dict()) == {"class_token", "project_features.0.weight", "project_features.0.bias",
                                    "transformer.layers.0.0.norm.weight", "transformer.layers.0.0.norm.bias",
                                    "transformer.layers.0.0.mhsa.in_proj_weight", "transformer.layers.0.0.mhsa.in_proj_bias",
                                    "transformer.layers.0.0.mhsa.out_proj.weight", "transformer.layers.0.0.mhsa.out_proj.bias",
                                    "transformer.layers.0.1.0.weight", "transformer.layers.0.1.0.bias",
                                    "transformer.layers.0.1.1.weight", "transformer.layers.0.1.1.bias",
                                    "transformer.layers.0.1.4.weight", "transformer.layers.0.1.4.bias",
                                    "transformer.norm.weight", "transformer.norm.bias", "mlp_head.0.weight", "mlp_head.0.bias"}
    bags = torch.randn(2, 50, 256).to(gpu)           # fp32 bags are accepted too (cast on the device)
    with torch.no_grad():
        assert torch.equal(m1(bags, coords=None, mask=None), m2(bags, coords=None, mask=None))
    with pytest.raises(NotImplementedError):      # mask is an inference-path feature
        m1.train()(bags, coords=None, mask=torch.zeros(2, 50, dtype=torch.bool, device=gpu))
    m1.eval()
    with pytest.raises(RuntimeError, match="GPU"):
        with torch.no_grad():
            m1.eval()(bags.cpu(), coords=None, mask=None)


def test_fixed_size_bag_matches_reference_golden(gpu):
    """bit-exact against the fixtures captured from the reference's _to_fixed_size_bag (same RNG stream)."""
    from pathlib import Path

    import numpy as np

    from stamp_amd.mil import to_fixed_size_bag

    z = np.load(Path(__file__).parent / "golden" / "fixed_size_bag.npz")
    for n, bs in ((10, 16), (100, 16), (16, 16), (1000, 512), (1, 4)):
        bag, coords = torch.from_numpy(z[f"bag_{n}_{bs}"]).to(gpu), torch.from_numpy(z[f"coords_{n}_{bs}"]).to(gpu)
        b, c, l = to_fixed_size_bag(bag, coords, bs, deterministic=True)
        assert np.array_equal(b.cpu().numpy(), z[f"det_bag_{n}_{bs}"]) and np.array_equal(c.cpu().numpy(), z[f"det_coords_{n}_{bs}"])
        assert l == int(z[f"det_len_{n}_{bs}"])
        torch.manual_seed(1234)
        b, c, l = to_fixed_size_bag(bag, coords, bs, deterministic=False)
        assert np.array_equal(b.cpu().numpy(), z[f"rand_bag_{n}_{bs}"]) and np.array_equal(c.cpu().numpy(), z[f"rand_coords_{n}_{bs}"])
    # fp16 features on disk -> fp32 bag (".float()"), zero padding
    f16 = torch.randn(5, 64).half().to(gpu)
    b, _, l = to_fixed_size_bag(f16, torch.zeros(5, 2, device=gpu), 8, deterministic=True)
    assert b.dtype == torch.float32 and torch.equal(b[:5], f16.float()) and (b[5:] == 0).all() and l == 5


@pytest.mark.parametrize("name,tdt,idt", [("f32", torch.float32, torch.int32), ("f16", torch.float16, torch.int16), ("bf16", torch.bfloat16, torch.int16)])
def test_vary_precision_bit_exact_vs_reference_golden(gpu, name, tdt, idt):
    from pathlib import Path

    import numpy as np

    from stamp_amd.mil import vary_precision

    z = np.load(Path(__file__).parent / "golden" / "vary_precision.npz")
    data = torch.from_numpy(z[f"in_{name}"]).view(tdt).to(gpu)
    torch.manual_seed(22)                   # the draw the reference made (global CPU generator)
    out = vary_precision(data, min_fraction_bits=2)
    assert np.array_equal(out.cpu().view(idt).numpy(), z[f"out_{name}"])
    with pytest.raises(ValueError):
        vary_precision(data, min_fraction_bits=0)


def test_mlp_linear_heads_match_reference_golden(gpu):
    from pathlib import Path

    import numpy as np

    from stamp_amd.mil import MLP, Linear

    z = np.load(Path(__file__).parent / "golden" / "mlp.npz")
    m = MLP(dim_input=20, dim_hidden=16, dim_output=3, num_layers=3, dropout=0.0).eval()
    m.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp:")})
    lin = Linear(dim_input=20, dim_output=2).eval()
    lin.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("lin:")})
    m, lin = m.to(gpu), lin.to(gpu)
    with torch.no_grad():
        for key in ("3", "2"):
            x = torch.from_numpy(z[f"x{key}"]).to(gpu)
            np.testing.assert_allclose(m(x).cpu().numpy(), z[f"mlp_y{key}"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(lin(x).cpu().numpy(), z[f"lin_y{key}"], rtol=1e-5, atol=1e-6)
        with pytest.raises(ValueError):
            m(torch.zeros(2, device=gpu))


@pytest.mark.parametrize("shape", ["slide_table", "bags", "odd"])
def test_mlp_linear_heads_train_like_torch(gpu, shape):
    """H14 training (reference mlp.py:6-62 under LitSlide* / LitPatient* / LitTile*, models/__init__.py:778-937): `loss.backward()` through the
    HIP heads fills the same gradients torch autograd computes for the reference module on the same fp32 inputs -- every parameter, and
    the input (through the mean over tiles when bags are given) -- and one AdamW step leaves the same weights.  Shapes: a table of slide
    embeddings (1000 x 768 -> 256 -> 2, configs[3]'s consumer), bags [B, T, F], and dimensions that are multiples of nothing."""
    from stamp_amd.mil import MLP, Linear
    torch.manual_seed(31)
    if shape == "slide_table":
        x0, kw = torch.randn(1000, 768), dict(dim_input=768, dim_hidden=256, dim_output=2, num_layers=3)
    elif shape == "bags":
        x0, kw = torch.randn(16, 77, 512).half().float(), dict(dim_input=512, dim_hidden=128, dim_output=4, num_layers=2)
    else:
        x0, kw = torch.randn(13, 37), dict(dim_input=37, dim_hidden=29, dim_output=3, num_layers=4)
    C = kw["dim_output"]
    targets = torch.nn.functional.one_hot(torch.arange(x0.shape[0]) % C, C).float().to(gpu)
    hip = MLP(dropout=0.0, **kw).to(gpu).train()
    ref = torch.nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else type(m)() for m in hip.mlp
                                if not isinstance(m, torch.nn.Dropout)]).to(gpu)
    ref.load_state_dict({f"{i}.{n}": p.detach().clone() for i, k in enumerate(j for j, m in enumerate(hip.mlp) if not isinstance(m, torch.nn.Dropout))
                         for n, p in hip.mlp[k].named_parameters()})
    xa, xb = x0.to(gpu).requires_grad_(True), x0.to(gpu).requires_grad_(True)
    ya = hip(xa)
    yb = ref(xb.mean(1) if xb.dim() == 3 else xb)
    assert _rel(ya, yb) < 1e-5
    torch.nn.functional.cross_entropy(ya, targets).backward()
    torch.nn.functional.cross_entropy(yb, targets).backward()
    ga = [p.grad for p in hip.parameters()]
    gb = [p.grad for p in ref.parameters()]
    assert len(ga) == len(gb) and all(g is not None and g.shape == r.shape for g, r in zip(ga, gb))
    for g, r in zip(ga, gb):
        assert _rel(g, r) < 2e-5, (g.shape, _rel(g, r))
    assert xa.grad.shape == x0.shape and _rel(xa.grad, xb.grad) < 2e-5
    oa, ob = torch.optim.AdamW(hip.parameters(), lr=1e-2), torch.optim.AdamW(ref.parameters(), lr=1e-2)
    oa.step(); ob.step()
    for p, r in zip(hip.parameters(), ref.parameters()):
        assert _rel(p, r) < 1e-5
    # Linear head
    lin = Linear(dim_input=kw["dim_input"], dim_output=C).to(gpu).train()
    rl = torch.nn.Linear(kw["dim_input"], C).to(gpu)
    rl.load_state_dict(lin.fc.state_dict())
    xa2, xb2 = x0.to(gpu).requires_grad_(True), x0.to(gpu).requires_grad_(True)
    torch.nn.functional.cross_entropy(lin(xa2), targets).backward()
    torch.nn.functional.cross_entropy(rl(xb2.mean(1) if xb2.dim() == 3 else xb2), targets).backward()
    assert _rel(lin.fc.weight.grad, rl.weight.grad) < 2e-5 and _rel(lin.fc.bias.grad, rl.bias.grad) < 2e-5 and _rel(xa2.grad, xb2.grad) < 2e-5


def test_mlp_head_training_follows_torchs_matmul_precision(gpu):
    """Under `torch.set_float32_matmul_precision("high")` -- what the reference's `train_model_` sets (modeling/train.py:519) -- the heads' backward GEMMs
    (dx = dz W, dW = dz^T x on `amds_bgemm_f32`) run as bf16 x 3 products: gradients within 1e-4 relative L2 of an fp64 reference (measured ~1e-6; TF32, the other
    form of "high", would be ~1e-3), the forward (`amds_linear_f32`, not a tiled product) unchanged bit for bit, and "highest" restored on exit = the default's bits."""
    from stamp_amd import ops
    from stamp_amd.mil import MLP
    torch.manual_seed(5)
    x0 = torch.randn(1024, 768)
    hip = MLP(dropout=0.0, dim_input=768, dim_hidden=256, dim_output=2, num_layers=3).to(gpu).train()
    targets = torch.nn.functional.one_hot(torch.arange(1024) % 2, 2).float().to(gpu)

    def grads():
        for p in hip.parameters():
            p.grad = None
        x = x0.to(gpu).requires_grad_(True)
        y = hip(x)
        torch.nn.functional.cross_entropy(y, targets).backward()
        return y.detach(), [p.grad.clone() for p in hip.parameters()] + [x.grad.clone()]

    y_exact, g_exact = grads()
    with ops.float32_matmul_precision("high"):
        y_high, g_high = grads()
    y_again, g_again = grads()
    assert torch.equal(y_exact, y_high) and torch.equal(y_exact, y_again)
    assert all(torch.equal(a, b) for a, b in zip(g_exact, g_again))
    ref = torch.nn.Sequential(*[torch.nn.Linear(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else type(m)() for m in hip.mlp
                                if not isinstance(m, torch.nn.Dropout)]).double()
    ref.load_state_dict({f"{i}.{n}": p.detach().cpu().double() for i, k in enumerate(j for j, m in enumerate(hip.mlp) if not isinstance(m, torch.nn.Dropout))
                         for n, p in hip.mlp[k].named_parameters()})
    xr = x0.double().requires_grad_(True)
    torch.nn.functional.cross_entropy(ref(xr), targets.cpu().double()).backward()
    g_ref = [p.grad for p in ref.parameters()] + [xr.grad]
    for gh, ge, gr in zip(g_high, g_exact, g_ref):
        assert _rel(gh.cpu().double(), gr) < 1e-4 and _rel(ge.cpu().double(), gr) < 2e-5, (tuple(gr.shape), _rel(gh.cpu().double(), gr))
    assert any(not torch.equal(a, b) for a, b in zip(g_exact, g_high))          # the level did reach the kernels


def test_mlp_head_dropout_is_the_references_placement(gpu):
    """Train mode: Dropout(p) after every hidden ReLU (mlp.py:27-30), none after the last Linear.  With the keep masks the library drew
    (exported by `dropout_masks`) applied to a torch copy, output and gradients agree; eval mode has no dropout; seeds differ -> outputs differ;
    the realised drop rate is p."""
    from stamp_amd.mil import MLP
    torch.manual_seed(8)
    hip = MLP(dim_input=96, dim_hidden=512, dim_output=2, num_layers=3, dropout=0.25).to(gpu).train()
    x = torch.randn(200, 96, device=gpu)
    hip.dropout_seed = 4242
    y = hip(x)
    masks = hip.dropout_masks(200, 4242)
    assert len(masks) == 2 and abs(1.0 - masks[0].float().mean().item() - 0.25) < 0.01
    lins = [m for m in hip.mlp if isinstance(m, torch.nn.Linear)]
    params = [(m.weight.detach().clone().requires_grad_(True), m.bias.detach().clone().requires_grad_(True)) for m in lins]
    h = x
    for i, (w, b) in enumerate(params):
        h = torch.nn.functional.linear(h, w, b)
        if i < 2:
            h = torch.relu(h) * masks[i].float() / 0.75
    assert _rel(y, h) < 1e-5
    y.square().mean().backward()
    h.square().mean().backward()
    for m, (w, b) in zip(lins, params):
        assert _rel(m.weight.grad, w.grad) < 2e-5 and _rel(m.bias.grad, b.grad) < 2e-5
    hip.dropout_seed = 4243
    assert not torch.equal(hip(x), y)
    hip.eval()
    with torch.no_grad():
        e1, e2 = hip(x), hip(x)
    assert torch.equal(e1, e2) and not torch.equal(e1, y.detach())


@pytest.mark.parametrize("Bb,T", [(2, 1024), (1, 333)])
def test_mil_vit_alibi_forward_matches_oracle(gpu, Bb, T):
    """ALiBi-after-softmax (reference vision_tranformer.py:42-74), eval mode, post-training running means."""
    torch.manual_seed(T)
    F, C = 1024, 2
    model = VisionTransformer(dim_output=C, dim_input=F, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512,
                              dropout=0.0, use_alibi=True).eval()
    with torch.no_grad():
        for n, p in list(model.named_parameters()) + list(model.named_buffers()):
            if "running_mean" in n:
                p.fill_(900.0 + 200 * torch.rand(1).item())      # mean tile distance seen in training
            elif "items_so_far" in n:
                p.fill_(41.0)
            elif p.dim() == 1 and "class_token" not in n and "bias_scale" not in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = model.state_dict()
    assert "transformer.layers.0.0.mhsa.query_encoders.3.weight" in sd and "transformer.layers.1.0.mhsa.attentions.7.bias_scale" in sd
    bags = torch.randn(Bb, T, F).half()
    coords = (torch.rand(Bb, T, 2) * 40000 / 256).round() * 256          # tile grid in um
    ref = mil_vit_forward(bags.float(), coords, None, sd, n_heads=8, use_alibi=True)
    with torch.no_grad():
        out = model(bags.to(gpu), coords=coords.to(gpu), mask=None)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-2 * max(1.0, ref.abs().max().item()), (err, ref, out)


@pytest.mark.parametrize("tag", ["t50", "t300"])
def test_transmil_matches_reference_golden(gpu, tag):
    """logits against the fixture captured from the reference TransMIL (same state_dict, same bags)."""
    from pathlib import Path

    import numpy as np

    from stamp_amd.mil import TransMIL

    z = np.load(Path(__file__).parent / "golden" / f"transmil_{tag}.npz")
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    dim_out, dim_in, dim_h = (int(v) for v in z["hparams"])
    model = TransMIL(dim_output=dim_out, dim_input=dim_in, dim_hidden=dim_h).eval()
    model.load_state_dict(sd, strict=True)                   # identical keys to the reference
    model = model.to(gpu)
    with torch.no_grad():
        out = model(torch.from_numpy(z["bags"]).to(gpu))
    np.testing.assert_allclose(out.cpu().numpy(), z["logits"], rtol=2e-3, atol=2e-3)


def test_transmil_bag_1024_vs_oracle(gpu):
    from oracle.transmil import transmil_forward
    from stamp_amd.mil import TransMIL

    torch.manual_seed(3)
    model = TransMIL(dim_output=2, dim_input=1024, dim_hidden=512).eval()
    bags = torch.randn(2, 1024, 1024).half().float()
    ref = transmil_forward(bags, {k: v.detach() for k, v in model.state_dict().items()})
    with torch.no_grad():
        out = model.to(gpu)(bags.to(gpu))
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()), (err, ref, out)


@pytest.mark.parametrize("alibi,masked,dims,bdt", [(False, False, (512, 512, 8, 512), torch.float16), (True, False, (512, 512, 8, 512), torch.float16),
                                                   (False, True, (456, 132, 4, 135), torch.float32), (True, True, (456, 132, 4, 135), torch.float32),
                                                   (False, False, (300, 96, 3, 200), torch.float16), (True, False, (768, 256, 4, 512), torch.bfloat16)])
def test_mil_vit_forward_one_call_equals_the_kernel_by_kernel_chain(gpu, alibi, masked, dims, bdt):
    """amds_mil_vit_forward (one C call: staging, class token, L layers, final norm, head) against the same kernels launched one by one from
    the host -- bit-identical logits, with the reference's odd test shapes (tests/test_model.py:9-32), padding masks, ALiBi, any bag dtype."""
    from chains import mil_vit as chain
    from stamp_amd import mil_core
    F, D, H, FF = dims
    torch.manual_seed(F + D + int(alibi))
    model = VisionTransformer(dim_output=3, dim_input=F, dim_model=D, n_layers=2, n_heads=H, dim_feedforward=FF, dropout=0.0, use_alibi=alibi).eval()
    Bb, T = 3, 211
    bags = torch.randn(Bb, T, F).to(bdt).to(gpu)
    coords = (torch.rand(Bb, T, 2) * 2000).to(gpu)
    mask = None
    if masked:
        mask = torch.zeros(Bb, T, dtype=torch.bool)
        mask[0, 150:] = True
        mask[2, 7:] = True
        mask = mask.to(gpu)
    pk = model._infer_pack(bags.device)
    with torch.no_grad():
        one = mil_core.forward_infer(pk, bags, coords, mask)
        chain = chain.forward_infer_stepwise(pk, bags, coords, mask)
        again = model(bags, coords=coords, mask=mask)
    assert one.shape == (Bb, 3) and torch.isfinite(one).all()
    assert torch.equal(one, chain) and torch.equal(one, again)
    with pytest.raises(ValueError, match="bags must be"):
        mil_core.forward_infer(pk, bags[..., :-1], coords, mask)
    if alibi:
        with pytest.raises(ValueError, match="needs coords"):
            mil_core.forward_infer(pk, bags, None, mask)


def test_mil_vit_forward_c_abi_guards(gpu):
    import ctypes as C

    from stamp_amd import _lib
    lib = _lib.lib()
    cfg = _lib.MilVitCfg(512, 512, 8, 512, 2, 2, 0, _lib.F16)
    need = lib.amds_mil_vit_workspace_bytes(C.byref(cfg), 4, 512)
    assert need > 4 * 513 * 512 * 4
    assert lib.amds_mil_vit_workspace_bytes(C.byref(_lib.MilVitCfg(512, 520, 8, 512, 2, 2, 0, _lib.F16)), 4, 512) == 0      # head_dim 65
    assert b"head_dim" in lib.amds_last_error()
    model = VisionTransformer(dim_output=2, dim_input=512, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512, dropout=0.0, use_alibi=False).eval()
    pk = model._infer_pack(torch.device(gpu))
    cfg, wc = pk.c_structs()
    bags = torch.randn(4, 512, 512, device=gpu).half()
    out = torch.zeros(4, 2, device=gpu)
    small = torch.empty(1024, dtype=torch.uint8, device=gpu)
    rc = lib.amds_mil_vit_forward(C.byref(cfg), C.byref(wc), bags.data_ptr(), _lib.F16, None, None, out.data_ptr(), 4, 512, small.data_ptr(), small.numel(), None)
    assert rc == -2 and b"workspace" in lib.amds_last_error()      # AMDS_ERR_WORKSPACE
    rc = lib.amds_mil_vit_forward(C.byref(cfg), C.byref(wc), None, _lib.F16, None, None, out.data_ptr(), 4, 512, small.data_ptr(), small.numel(), None)
    assert rc == -1 and b"null" in lib.amds_last_error()              # AMDS_ERR_INVALID


@pytest.mark.parametrize("Bb,T,F,Cd,bdt", [(2, 1024, 1024, 512, torch.float16), (3, 50, 768, 256, torch.float32), (1, 300, 512, 512, torch.float16), (2, 1, 128, 64, torch.float32),
                                           (1, 4100, 256, 128, torch.bfloat16)])
def test_transmil_forward_one_call_equals_the_kernel_by_kernel_chain(gpu, Bb, T, F, Cd, bdt):
    """amds_transmil_forward (one C call) against the same kernels launched one by one from the host: bit-identical logits -- with and
    without the front padding of the Nystrom attention, a single tile, square and non-square tile counts, every bag dtype."""
    from stamp_amd.mil import TransMIL
    torch.manual_seed(T + Cd)
    model = TransMIL(dim_output=3, dim_input=F, dim_hidden=Cd).eval().to(gpu)
    bags = torch.randn(Bb, T, F).to(bdt).to(gpu)
    from stamp_amd import ops
    with torch.no_grad():
        tail = model(bags)                                 # default: the second layer's class row alone (csrc/transmil_fwd.hip)
        was = ops.set_mil_cls_tail(False)
        try:
            one = model(bags)
            from chains.transmil import transmil_forward_stepwise
            chain = transmil_forward_stepwise(model, bags)
        finally:
            ops.set_mil_cls_tail(was)
    assert one.shape == (Bb, 3) and torch.isfinite(one).all() and torch.equal(one, chain)
    # the head reads `self.norm(h)[:, 0]` behind the second TransLayer (trans_mil.py:319-323): attn1, attn1 pinv, the merge, the residual convolution and to_out for
    # that row alone give the full layer's logits to fp32 rounding (one-row products take another kernel of the same exact-fp32 family)
    assert torch.isfinite(tail).all() and (tail - one).abs().max().item() < 1e-4 * max(1.0, one.abs().max().item())
    with torch.no_grad():
        model._fc2.bias.add_(1.0)                          # the cached device weights follow the parameters
        assert torch.allclose(model(bags), one + 1.0, atol=1e-5)
    with pytest.raises(ValueError, match="bags must be"):
        with torch.no_grad():
            model(bags[..., :-1])
