"""End-to-end parity of the HIP tile encoder against the CPU oracle on identical u8 tiles and weights."""
import pytest
import torch

from oracle.vit_tile_encoder import extract_features
from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


@pytest.mark.parametrize("name", ["test_tiny", "test_tiny_swiglu", "test_tiny_hd80"])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_tiny_vit_matches_oracle(gpu, name, dt):
    cfg = PRESETS[name]
    sd = random_vit_state_dict(cfg, seed=1, init="moderate")
    tiles = torch.randint(0, 256, (5, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    model = HipViT(cfg, sd, device=gpu, act_dtype=dt, chunk=2)     # chunk < B: exercises the chunk loop
    f, t = model(tiles.to(gpu), return_tokens=True)
    tol = 2e-3 if dt == torch.float16 else 2e-2
    assert _rel(t.cpu(), ref_t) < tol, _rel(t.cpu(), ref_t)
    assert f.dtype == torch.float16 and _rel(f.cpu().float(), ref_f.float()) < tol
    # the product's default call (class features only: the last block's class-row tail, amds_vit_weights.cls_tail) against the same oracle
    f2 = model(tiles.to(gpu))
    assert f2.dtype == torch.float16 and _rel(f2.cpu().float(), ref_f.float()) < tol
    # determinism: same input twice -> bit-identical
    assert torch.equal(f2, model(tiles.to(gpu)))
    # float (already-normalised CHW) input takes the same path
    from oracle.vit_tile_encoder import tile_transform
    f3 = model(tile_transform(tiles, cfg.mean, cfg.std).to(gpu))
    assert torch.equal(f2, f3)


def test_vit_large_matches_oracle(gpu):
    """The headline shape (ViT-L/14, 257 tokens), fp16 operands / fp32 accumulate / fp32 residual stream.
    Stated tolerance: relative L2 error of the fp16 CLS features <= 1e-3 (BASELINE.json north_star)."""
    cfg = PRESETS["vit_large_patch14_224"]
    sd = random_vit_state_dict(cfg, seed=0, init="moderate")
    tiles = torch.randint(0, 256, (4, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    model = HipViT(cfg, sd, device=gpu, act_dtype=torch.float16, chunk=4)
    f, t = model(tiles.to(gpu), return_tokens=True)
    r_t, r_f = _rel(t.cpu(), ref_t), _rel(f.cpu().float(), ref_f.float())
    mx = ((f.cpu().float() - ref_f.float()).abs().max() / ref_f.float().abs().max()).item()
    r_d = _rel(model(tiles.to(gpu)).cpu().float(), ref_f.float())        # the default call: class features only (class-row tail in the last block)
    print(f"ViT-L/14 fp16 operands: rel-L2 tokens {r_t:.3e}, CLS features {r_f:.3e} (default call, class-row tail: {r_d:.3e}), max-abs/max {mx:.3e}")
    assert r_f < 1e-3 and r_t < 1e-3 and mx < 2e-3 and r_d < 1e-3


@pytest.mark.parametrize("name,tol_cls", [("uni2_h", 1e-3), ("virchow2", 1e-3), ("h_optimus_0", 1e-3), ("vit_large_patch16_224", 1e-3), ("dinobloom_s", 1e-3), ("gigapath", 1e-3)])
def test_full_size_presets_match_oracle(gpu, name, tol_cls):
    """Full-size parity for every preset DESIGN quotes a throughput for (reference uni2.py:17-37, virchow2.py:24-54 -- ViT-H/14
    with head_dim 80 --, h_optimus_0.py:15-30, uni.py:26-31): 2 tiles, fp16 operands / fp32 accumulate / fp32 residual stream.
    Stated tolerance (BASELINE.json north_star): relative L2 error <= 1e-3 of ALL final tokens (fp32) AND of the fp16 CLS feature row --
    the thing written to the .h5 -- rounded to fp16 on both sides, for every model.  Round 3: the patch-embedding weight is a 16-bit
    [hi | lo] pair (amds_vit_weights.patch_lo_shift); its single rounding was the largest error source of the path (UNI2-h tokens
    7.0e-4 -> 3.8e-4, CLS row 1.14e-3 -> ~8e-4; DESIGN.md section 5 lists the measured value per model)."""
    cfg = PRESETS[name]
    sd = random_vit_state_dict(cfg, seed=5, init="moderate")
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    model = HipViT(cfg, sd, device=gpu, act_dtype=torch.float16, chunk=2)
    f, t = model(tiles.to(gpu), return_tokens=True)
    r_t, r_f = _rel(t.cpu(), ref_t), _rel(f.cpu().float(), ref_f.float())
    r_d = _rel(model(tiles.to(gpu)).cpu().float(), ref_f.float())        # the default call: class features only (class-row tail in the last block)
    print(f"{name} fp16 operands: rel-L2 tokens {r_t:.3e}, CLS features {r_f:.3e} (default call, class-row tail: {r_d:.3e})")
    assert bool(torch.isfinite(f.float()).all()) and r_t < 1e-3 and r_f < tol_cls and r_d < tol_cls, (r_f, r_t, r_d)


@pytest.mark.parametrize("T,H,hd", [(257, 16, 64), (265, 24, 64), (261, 16, 80), (50, 2, 64), (288, 3, 80)])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_attention_cls_f32_matches_fp64(gpu, T, H, hd, dt):
    """The exact path's attention: one fp32 query row per (tile, head) against the stored 16-bit keys / values, everything else fp32.
    Bar 2e-6 relative L2 against fp64 on the same (already rounded) keys / values; spiky scores exercise the max subtraction."""
    from stamp_amd import ops
    g = torch.Generator().manual_seed(T + hd)
    B, D = 5, H * hd
    qkv = torch.randn(B * T, 3 * D, generator=g).to(dt)
    q = torch.randn(B, D, generator=g)
    q[1] *= 6.0                                                   # sharp softmax rows
    out = ops.attention_cls_f32(q.to(gpu), qkv.to(gpu), B, T, H, hd).cpu()
    k = qkv[:, D:2 * D].double().reshape(B, T, H, hd)
    v = qkv[:, 2 * D:].double().reshape(B, T, H, hd)
    s = torch.einsum("bhd,bthd->bht", q.double().reshape(B, H, hd), k) / hd ** 0.5
    ref = torch.einsum("bht,bthd->bhd", torch.softmax(s, -1), v).reshape(B, D)
    assert _rel(out, ref) < 2e-6, _rel(out, ref)
    assert torch.equal(out, ops.attention_cls_f32(q.to(gpu), qkv.to(gpu), B, T, H, hd).cpu())     # fixed summation order


def test_cls_scatter_gather_and_mlp_act(gpu):
    from stamp_amd import ops
    g = torch.Generator().manual_seed(3)
    B, T, D = 7, 11, 384
    x = torch.randn(B * T, D, generator=g).to(gpu)
    xc = ops.vit_cls_gather(x, B, T)
    assert torch.equal(xc, x.view(B, T, D)[:, 0])
    new = (torch.randn(B, D, generator=g) * 3 + 0.5).to(gpu)
    x2, xh, rs = x.clone(), torch.zeros(B * T, D, dtype=torch.float16, device=gpu), torch.zeros(B * T, 2, device=gpu)
    ops.vit_cls_scatter(new, x2, T, 1e-6, xh, rs)
    assert torch.equal(x2.view(B, T, D)[:, 0], new) and torch.equal(x2.view(B, T, D)[:, 1:], x.view(B, T, D)[:, 1:])
    assert torch.equal(xh.view(B, T, D)[:, 0], new.half()) and not xh.view(B, T, D)[:, 1:].any()
    mean, var = new.double().mean(-1), new.double().var(-1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    got = rs.view(B, T, 2)[:, 0].double()
    assert torch.allclose(got[:, 0], rstd, rtol=1e-5) and torch.allclose(got[:, 1], -mean * rstd, rtol=1e-5, atol=1e-6)
    x3 = x.clone()
    ops.vit_cls_scatter(new, x3, T, 1e-6)                              # rows only
    assert torch.equal(x3, x2)
    u = torch.randn(9, 2 * 136, generator=g).to(gpu)
    ug, us = u[:, :136].contiguous().clone(), u.clone()
    ops.mlp_act_f32(ug, 136, 0)
    ops.mlp_act_f32(us, 136, 1)
    assert _rel(ug.cpu(), torch.nn.functional.gelu(u[:, :136].double().cpu())) < 1e-6
    assert _rel(us[:, :136].cpu(), (torch.nn.functional.silu(u[:, :136].double()) * u[:, 136:].double()).cpu()) < 1e-6
    assert torch.equal(us[:, 136:], u[:, 136:])


@pytest.mark.parametrize("name", ["test_tiny", "test_tiny_swiglu", "test_tiny_hd80"])
@pytest.mark.parametrize("fold", [True, False])
def test_tiny_vit_exact_class_rows(gpu, name, fold):
    """exact=True: the class-token row on an exact-fp32 stream (csrc/vit_exact.hip).  The stored feature gets closer to the oracle, every
    other token keeps the 16-bit path's accuracy, the class row of the token tensor IS the stored feature's row, chunking and the
    LayerNorm-fold switch do not change what the mode means."""
    cfg = PRESETS[name]
    if fold and (cfg.dim % 256 or (cfg.hidden_pad * (2 if cfg.mlp == "swiglu" else 1)) % 256):
        pytest.skip("shape cannot fold")
    sd = random_vit_state_dict(cfg, seed=1, init="moderate")
    tiles = torch.randint(0, 256, (5, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    base = HipViT(cfg, sd, device=gpu, chunk=2, ln_fold=fold)
    f0, t0 = base(tiles.to(gpu), return_tokens=True)
    model = HipViT(cfg, sd, device=gpu, chunk=2, ln_fold=fold, exact=True)
    f, t = model(tiles.to(gpu), return_tokens=True)
    e0, e1 = _rel(t0[:, 0].cpu(), ref_t[:, 0]), _rel(t[:, 0].cpu(), ref_t[:, 0])
    print(f"{name} fold={fold}: class row rel-L2 {e0:.3e} -> {e1:.3e} (exact); tokens {_rel(t0.cpu(), ref_t):.3e} -> {_rel(t.cpu(), ref_t):.3e}")
    assert e1 < 0.8 * e0 and _rel(t.cpu(), ref_t) < 2e-3
    assert _rel(f.cpu().float(), ref_f.float()) < 1e-3
    assert torch.equal(f, t[:, 0].half())
    assert torch.equal(f, model(tiles.to(gpu)))                                         # deterministic
    assert torch.equal(f[:3], HipViT(cfg, sd, device=gpu, chunk=5, ln_fold=fold, exact=True)(tiles[:3].to(gpu)))    # batch / chunk invariant


@pytest.mark.parametrize("name", ["uni2_h", "virchow2", "h_optimus_0", "vit_large_patch14_224"])
def test_full_size_presets_exact_class_rows(gpu, name):
    """The opt-in exact mode at full size: the stored fp16 class feature (both sides rounded to fp16) within 6e-4 of the oracle for every
    preset -- measured 3-4e-4, against 7-9e-4 on the default path (bar 1e-3, test above); the fp32 class row within 4e-4."""
    cfg = PRESETS[name]
    sd = random_vit_state_dict(cfg, seed=5, init="moderate")
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    f, t = HipViT(cfg, sd, device=gpu, chunk=2, exact=True)(tiles.to(gpu), return_tokens=True)
    r_f, r_c, r_t = _rel(f.cpu().float(), ref_f.float()), _rel(t[:, 0].cpu(), ref_t[:, 0]), _rel(t.cpu(), ref_t)
    print(f"{name} exact class rows: fp16 CLS feature {r_f:.3e}, fp32 class row {r_c:.3e}, all tokens {r_t:.3e}")
    assert r_f < 6e-4 and r_c < 4e-4 and r_t < 1e-3, (r_f, r_c, r_t)


def test_folded_layernorm_range_diagnostics(gpu):
    """the folded path feeds the UN-normalised residual rows through a 16-bit copy.  The forward counts, for free (in the 9 us
    row-statistics kernel), the rows that could hold an element beyond fp16's range (sum of squares >= 65504^2: rows below provably cannot)
    and the rows with |mean| > 8 sigma.  Healthy weights: both 0.  A checkpoint with a massive-activation channel: counted, and
    `ln_fold=False` encodes the same tiles with finite features."""
    cfg = PRESETS["vit_large_patch14_224"]
    from dataclasses import replace
    cfg = replace(cfg, depth=2)
    sd = random_vit_state_dict(cfg, seed=2, init="moderate")
    tiles = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).to(gpu)
    m = HipViT(cfg, sd, device=gpu, chunk=3)
    f = m(tiles)
    assert m.range_diagnostics() == {"rows_beyond_fp16_range_possible": 0, "rows_mean_over_8_sigma": 0} and torch.isfinite(f.float()).all()
    bad = {k: v.clone() for k, v in sd.items()}
    bad["blocks.0.attn.proj.bias"][7] = 3.0e5 / 0.3          # one massive channel after the first attention branch (LayerScale ~0.3)
    mb = HipViT(cfg, bad, device=gpu, chunk=3, check="off")                                                    # (the guard off: what the default path would return)
    fb = mb(tiles)
    d = mb.range_diagnostics(reset=True)
    assert d["rows_beyond_fp16_range_possible"] >= 3 * cfg.tokens and not torch.isfinite(fb.float()).all()      # loud twice: counter AND non-finite features
    assert mb.range_diagnostics()["rows_beyond_fp16_range_possible"] == 0                                        # reset
    fu = HipViT(cfg, bad, device=gpu, chunk=3, ln_fold=False)(tiles)                                             # the stand-alone LayerNorm path is not affected
    assert torch.isfinite(fu.float()).all()
    import warnings
    with warnings.catch_warnings(record=True) as w:                                                              # the default: falls back to exactly that path
        warnings.simplefilter("always")
        md = HipViT(cfg, bad, device=gpu, chunk=3)
        assert torch.equal(md(tiles), fu) and md.safe_level == 1 and len(w) == 1
    shifted = {k: v.clone() for k, v in sd.items()}
    shifted["blocks.0.attn.proj.bias"] += 60.0 / shifted["blocks.0.ls1.gamma"]       # every channel shifted by +60 after LayerScale: |mean| >> sigma
    ms = HipViT(cfg, shifted, device=gpu, chunk=3, check="off")
    fs = ms(tiles)
    assert ms.range_diagnostics()["rows_mean_over_8_sigma"] >= cfg.tokens and torch.isfinite(fs.float()).all()
    # finite but imprecise (the folded form loses |mean| / sigma in relative precision): the guard treats it like an overflow
    from oracle.vit_tile_encoder import extract_features
    ref = extract_features(tiles.cpu(), shifted, cfg).float()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        mg = HipViT(cfg, shifted, device=gpu, chunk=3)
        fg = mg(tiles)
    e_fold, e_guard = _rel(fs.cpu().float(), ref), _rel(fg.cpu().float(), ref)
    print(f"|mean| ~ 60 sigma rows: folded path rel-L2 {e_fold:.3e}, guarded default (level {mg.safe_level}) {e_guard:.3e}")
    assert mg.safe_level == 1 and len(w) == 1 and "8 sigma" in str(w[0].message) and e_guard < 1e-3


def test_vit_large_bf16_matches_oracle(gpu):
    """BASELINE.json configs[1] says "bf16".  bf16 operands carry 8 mantissa bits (eps 3.9e-3): the 1e-3 feature bar of
    north_star is NOT reachable with them, which is why the product default is fp16 operands (same MFMA rate).
    Honest bf16 tolerance, stated: relative L2 <= 1e-2 on the CLS features and tokens of the full ViT-L/14 trunk."""
    cfg = PRESETS["vit_large_patch14_224"]
    sd = random_vit_state_dict(cfg, seed=0, init="moderate")
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    f, t = HipViT(cfg, sd, device=gpu, act_dtype=torch.bfloat16, chunk=2)(tiles.to(gpu), return_tokens=True)
    r_t, r_f = _rel(t.cpu(), ref_t), _rel(f.cpu().float(), ref_f.float())
    print(f"ViT-L/14 bf16 operands: rel-L2 tokens {r_t:.3e}, CLS features {r_f:.3e}")
    assert r_f < 1e-2 and r_t < 1e-2, (r_f, r_t)


def test_vit_large_chaotic_weights(gpu):
    """Ill-conditioned control ("stress" init: the fp32 and fp64 oracles already disagree by 1.1e-5, 180x the fp32
    epsilon).  fp16 operand rounding (2^-11) is amplified the same way; the bound below is that amplification with
    a 2x margin -- it documents the conditioning, it is not the product's tolerance claim."""
    cfg = PRESETS["vit_large_patch14_224"]
    sd = random_vit_state_dict(cfg, seed=0, init="stress")
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    ref_f = extract_features(tiles, sd, cfg)
    f = HipViT(cfg, sd, device=gpu, act_dtype=torch.float16, chunk=2)(tiles.to(gpu))
    assert _rel(f.cpu().float(), ref_f.float()) < 2.5e-2


def test_vit_large_batch_invariance_at_bench_size(gpu):
    """Size-independent properties at the bench shape (ViT-L/14, 1 100 tiles = 2 chunks of 510 + an 80-tile tail):
    a tile's feature does not depend on which batch / chunk it travels in (bit-exact), on the chunk size, or on the
    two-stream overlapped schedule; permuting tiles permutes features; every feature is finite."""
    cfg = PRESETS["vit_large_patch14_224"]
    sd = random_vit_state_dict(cfg, seed=0, init="moderate")
    g = torch.Generator().manual_seed(11)
    tiles = torch.randint(0, 256, (1100, 224, 224, 3), dtype=torch.uint8, generator=g).to(gpu)
    model = HipViT(cfg, sd, device=gpu, chunk=510)
    f = model(tiles)
    assert f.shape == (1100, cfg.dim) and torch.isfinite(f.float()).all()
    assert torch.equal(model(tiles[:37]), f[:37])                       # prefix batch
    assert torch.equal(model(tiles[600:700]), f[600:700])               # tiles that crossed a chunk boundary
    assert torch.equal(model.with_chunk(255)(tiles), f)                 # chunk size
    model.with_chunk(510).overlap = True
    assert torch.equal(model(tiles), f)                                 # two chunks in flight on two streams
    model.overlap = False
    perm = torch.randperm(1100, generator=g).to(gpu)
    assert torch.equal(model(tiles[perm].contiguous()), f[perm])        # permutation equivariance
    assert model(tiles[:0]).shape == (0, cfg.dim)                       # empty slide
    # distinct tiles give distinct features, identical tiles identical features
    assert not torch.equal(f[0], f[1])
    dup = tiles[:2].clone(); dup[1] = dup[0]
    fd = model(dup)
    assert torch.equal(fd[0], fd[1])


def test_lnfold_and_plain_paths_against_the_oracle_and_b64_tail(gpu):
    """The two forms of a block's LayerNorms (folded into the GEMMs: the default; separate kernels: ln_fold=False) are both within the
    stated tolerance of the fp32 oracle and of each other; and the reference's DataLoader batch of 64 tiles (64 x 257 rows = 64 row tiles
    + 64 rows: the ragged-tail schedule on the side stream) gives the bits the same tiles get inside a 1020-tile chunk."""
    cfg = PRESETS["vit_large_patch14_224"]
    sd = random_vit_state_dict(cfg, seed=3, init="moderate")
    tiles = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(21))
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    fold, plain = HipViT(cfg, sd, device=gpu, chunk=1020), HipViT(cfg, sd, device=gpu, chunk=1020, ln_fold=False)
    assert fold.ln_fold and not plain.ln_fold
    ff, tf = fold(tiles.to(gpu), return_tokens=True)
    fp, tp = plain(tiles.to(gpu), return_tokens=True)
    e_f, e_p, d = _rel(tf.cpu(), ref_t), _rel(tp.cpu(), ref_t), _rel(tf, tp)
    print(f"ViT-L/14 tokens vs oracle: LayerNorm folded {e_f:.3e}, separate kernels {e_p:.3e}; folded vs separate {d:.3e}")
    assert e_f < 1e-3 and e_p < 1e-3 and d < 1e-3
    assert _rel(ff.cpu().float(), ref_f.float()) < 1e-3 and _rel(fp.cpu().float(), ref_f.float()) < 1e-3
    big = torch.randint(0, 256, (1020, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(22)).to(gpu)
    whole_fold = fold(big)
    for m in (fold, plain):
        whole = m(big)
        assert torch.equal(m(big[128:192]), whole[128:192])        # a batch of 64: the last tiles run as their own chain on the side stream
        assert torch.equal(m(big[7:107]), whole[7:107])            # 100 tiles: 25 700 rows = 100 row tiles + 100 rows
    # every batch size around the tail-schedule decisions (none below 8 tiles; 1-4 tail tiles above), same bits as inside the big chunk
    for B in (1, 2, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 65, 96, 127, 128, 129, 255, 256):
        assert torch.equal(fold(big[3:3 + B]), whole_fold[3:3 + B]), B


def test_cls_plus_mean_patch_embedding(gpu):
    """`VirchowConcatenated` (reference virchow_full.py:25-35): cat(class token, mean of the remaining tokens)."""
    from stamp_amd.vit import HipViTClsMean
    cfg = PRESETS["test_tiny_hd80"]
    sd = random_vit_state_dict(cfg, seed=4, init="moderate")
    tiles = torch.randint(0, 256, (5, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(8))
    _, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    ref = torch.cat([ref_t[:, 0], ref_t[:, 1:].mean(1)], dim=-1)
    model = HipViTClsMean(HipViT(cfg, sd, device=gpu, chunk=2))
    out = model(tiles.to(gpu))
    assert out.dtype == torch.float16 and out.shape == (5, 2 * cfg.dim)
    assert _rel(out.cpu().float(), ref) < 2e-3
    assert model(tiles[:0].to(gpu)).shape == (0, 2 * cfg.dim)


@pytest.mark.parametrize("tag,kw", [("gelu", dict(hidden=256)), ("swiglu", dict(hidden=344, mlp="swiglu")), ("reg4", dict(hidden=344, mlp="swiglu", reg_tokens=4))])
def test_hip_vit_matches_transformers_dinov2_fixture(gpu, tag, kw):
    """The HIP tile encoder against outputs of an INDEPENDENT third-party implementation (`transformers`' Dinov2Model / Dinov2WithRegistersModel,
    tests/golden/dinov2_hf.npz; timm itself is not installed): GELU + LayerScale, SwiGLU (hidden 344 -> zero-padded), SwiGLU + 4 register tokens.
    Bar: the fp16-operand path's 1e-3 relative L2 on the compared tokens and on the stored class feature."""
    from pathlib import Path

    import numpy as np

    from stamp_amd.vit import ViTConfig, hf_dinov2_to_timm_names
    z = np.load(Path(__file__).parent / "golden" / "dinov2_hf.npz")
    sd = hf_dinov2_to_timm_names({k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_w:")})
    cfg = ViTConfig(dim=128, depth=2, heads=2, **kw)
    model = HipViT(cfg, sd, device=gpu, chunk=2)
    f, t = model(torch.from_numpy(z["tiles"]).to(gpu), return_tokens=True)
    sel = list(range(10)) + [-2, -1]
    ref = torch.from_numpy(z[f"{tag}_tokens"])
    r_t, r_f = _rel(t[:, sel].cpu(), ref), _rel(f.float().cpu(), ref[:, 0])
    r_d = _rel(model(torch.from_numpy(z["tiles"]).to(gpu)).float().cpu(), ref[:, 0])       # the default call (class-row tail)
    print(f"dinov2 ({tag}) vs transformers: tokens {r_t:.3e}, CLS feature {r_f:.3e} (default call {r_d:.3e})")
    assert r_t < 1e-3 and r_f < 1e-3 and r_d < 1e-3
    np.testing.assert_allclose(t.norm(dim=-1).cpu().numpy(), z[f"{tag}_token_norms"], rtol=3e-3)


@pytest.mark.parametrize("name", ["test_tiny", "test_tiny_swiglu", "test_tiny_hd80"])
@pytest.mark.parametrize("fold", [True, False])
def test_class_row_tail_of_the_last_block(gpu, name, fold):
    """cls_tail (the default): with only the class features requested -- what the reference stores, `model(tiles)[:, 0]`,
    src/stamp/preprocessing/__init__.py:324-325 -- the last block computes keys / values and the class row's own chain only
    (include/amdstamp.h, amds_vit_weights.cls_tail).  Same features as the full block up to the 16-bit path's rounding and no further from the
    oracle; asking for the token tensor runs the full block (the cls_tail=False bits); chunk / batch invariant and deterministic."""
    cfg = PRESETS[name]
    if fold and (cfg.dim % 256 or (cfg.hidden_pad * (2 if cfg.mlp == "swiglu" else 1)) % 256):
        pytest.skip("shape cannot fold")
    sd = random_vit_state_dict(cfg, seed=1, init="moderate")
    tiles = torch.randint(0, 256, (9, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    ref_f = extract_features(tiles, sd, cfg)
    full = HipViT(cfg, sd, device=gpu, chunk=4, ln_fold=fold, cls_tail=False)
    tail = HipViT(cfg, sd, device=gpu, chunk=4, ln_fold=fold)
    assert tail.cls_tail and not full.cls_tail
    f_full, f_tail = full(tiles.to(gpu)), tail(tiles.to(gpu))
    e_full, e_tail = _rel(f_full.cpu().float(), ref_f.float()), _rel(f_tail.cpu().float(), ref_f.float())
    print(f"{name} fold={fold}: CLS rel-L2 vs oracle: full last block {e_full:.3e}, class-row tail {e_tail:.3e}; tail vs full {_rel(f_tail.float(), f_full.float()):.3e}")
    assert e_tail < 1e-3 and e_tail < 1.1 * e_full + 1e-5
    assert _rel(f_tail.float(), f_full.float()) < 1e-3
    f_tok, t_tok = tail(tiles.to(gpu), return_tokens=True)
    assert torch.equal(f_tok, f_full) and torch.equal(f_tok, t_tok[:, 0].half())        # a token tensor needs the whole block
    assert torch.equal(f_tail, tail(tiles.to(gpu)))
    assert torch.equal(f_tail[:3], HipViT(cfg, sd, device=gpu, chunk=9, ln_fold=fold)(tiles[:3].to(gpu)))
    # with the exact class stream the tail IS that stream's last block: same bits with and without it
    ex_full = HipViT(cfg, sd, device=gpu, chunk=4, ln_fold=fold, exact=True, cls_tail=False)(tiles.to(gpu))
    assert torch.equal(HipViT(cfg, sd, device=gpu, chunk=4, ln_fold=fold, exact=True)(tiles.to(gpu)), ex_full)


def test_residual_planes_against_fp32_rows(gpu, monkeypatch):
    """The folded path keeps the residual stream as two fp16 planes (hi = the GEMMs' A operand, lo = x - hi; include/amdstamp.h,
    amds_gemm_lnfold_planes) instead of fp32 rows + a 16-bit copy.  Against the fp32-row form (AMDS_VIT_PLANES=0) on the full ViT-L/14:
    neither is further from the oracle than the other (measured: tokens 3.054e-4 / 3.053e-4, stored feature 5.18e-4 / 5.17e-4).  The two
    forms differ from EACH OTHER by ~2e-4 on the tokens: an fp32-eps change of a residual element flips the fp16 rounding of the next GEMM
    operand for ~2e-4 of the elements, and 48 such sites make the two runs two samples of the same 16-bit rounding noise -- which is why the
    bar is the oracle, not the other form.  Batch / chunk invariance holds in the plane form too."""
    cfg = PRESETS["vit_large_patch14_224"]
    sd = random_vit_state_dict(cfg, seed=0, init="moderate")
    tiles = torch.randint(0, 256, (6, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    ref_f, ref_t = extract_features(tiles[:4], sd, cfg, return_tokens=True)
    model = HipViT(cfg, sd, device=gpu, chunk=4)
    assert model.ln_fold
    f_pl, t_pl = model(tiles.to(gpu), return_tokens=True)
    d_pl = model(tiles.to(gpu))
    monkeypatch.setenv("AMDS_VIT_PLANES", "0")
    f_32, t_32 = model(tiles.to(gpu), return_tokens=True)
    d_32 = model(tiles.to(gpu))
    monkeypatch.delenv("AMDS_VIT_PLANES")
    assert torch.equal(d_pl, model(tiles.to(gpu)))
    r_ft, r_dd, r_tt = _rel(f_pl.float(), f_32.float()), _rel(d_pl.float(), d_32.float()), _rel(t_pl, t_32)
    e_pl, e_32 = _rel(t_pl[:4].cpu(), ref_t), _rel(t_32[:4].cpu(), ref_t)
    c_pl, c_32 = _rel(d_pl[:4].cpu().float(), ref_f.float()), _rel(d_32[:4].cpu().float(), ref_f.float())
    print(f"ViT-L/14 planes vs fp32 rows: features {r_ft:.2e} (default call {r_dd:.2e}), tokens {r_tt:.2e}; vs oracle: tokens {e_pl:.3e} / {e_32:.3e}, "
          f"stored feature {c_pl:.3e} / {c_32:.3e}")
    assert r_tt < 6e-4 and r_ft < 1e-3 and r_dd < 1e-3              # two samples of the same rounding noise (each ~3e-4 / 5e-4 from the oracle)
    assert e_pl < 1e-3 and c_pl < 1e-3 and e_pl < 1.05 * e_32 and c_pl < 1.1 * c_32
    assert torch.equal(d_pl[:3], HipViT(cfg, sd, device=gpu, chunk=6)(tiles[:3].to(gpu)))
