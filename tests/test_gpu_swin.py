"""HIP CTransPath (ConvStem + Swin-T) against the pinned oracle and, end to end, against the
golden vectors captured from the reference's own `_SwinTransformer` (tests/golden/ctranspath_*.npz)."""
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import swin_ctranspath as osw
from stamp_amd import _lib, ops
from stamp_amd.swin import (SWIN_PRESETS, HipSwin, pack_stem_params, random_swin_state_dict, rel_bias_lane_table,
                            shift_mask_bits)

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


@pytest.mark.parametrize("img", [112, 224])
def test_conv_stem(gpu, img):
    """fp32 FMA stem vs the fp32 oracle: conv + folded BatchNorm + ReLU + 1x1 + LayerNorm, incl. zero padding in the
    normalised domain at the tile border."""
    cfg = SWIN_PRESETS["ctranspath"]
    sd = random_swin_state_dict(cfg, seed=3)
    g = torch.Generator().manual_seed(5)
    tiles = torch.randint(0, 256, (3, img, img, 3), dtype=torch.uint8, generator=g)
    tiles[0] = 255                      # saturated tile: border padding is the only structure
    x = tiles.permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor(cfg.mean).view(1, 3, 1, 1)) / torch.tensor(cfg.std).view(1, 3, 1, 1)
    ref = osw.conv_stem(x, sd)
    got = ops.swin_stem(tiles.to(gpu), pack_stem_params(sd, cfg).to(gpu))
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max().item() < 5e-5, (got.cpu() - ref).abs().max().item()


def _window_attention_ref(qkv, table, B, grid, heads, shift):
    """fp64 evaluation of the window attention on the SAME (already rounded) qkv, via the oracle's index helpers."""
    C = heads * 32
    q3 = qkv.double().reshape(B, grid * grid, 3, heads, 32)
    ids = osw.window_token_ids(grid, grid, shift)
    nW = ids.shape[0]
    w = q3[:, ids.reshape(-1)].reshape(B * nW, 49, 3, heads, 32)
    q, k, v = (w[:, :, i].transpose(1, 2) for i in range(3))
    att = (q * 32 ** -0.5) @ k.transpose(-2, -1)
    bias = table.double()[osw.rel_pos_index().reshape(-1)].reshape(49, 49, heads).permute(2, 0, 1)
    att = att + bias.unsqueeze(0)
    if shift:
        lab = osw.window_region_labels(grid, grid, shift)
        mask = torch.where(lab[:, :, None] != lab[:, None, :], -100.0, 0.0).double()
        att = (att.reshape(B, nW, heads, 49, 49) + mask[None, :, None]).reshape(B * nW, heads, 49, 49)
    o = (torch.softmax(att, -1) @ v).transpose(1, 2).reshape(B, nW * 49, C)
    out = torch.empty(B, grid * grid, C, dtype=torch.float64)
    out[:, ids.reshape(-1)] = o
    return out.reshape(B * grid * grid, C)


@pytest.mark.parametrize("grid,heads,shift", [(56, 3, 0), (56, 3, 3), (28, 6, 3), (14, 12, 0), (14, 12, 3), (7, 24, 0)])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_window_attention(gpu, grid, heads, shift, dt):
    B = 2
    g = torch.Generator().manual_seed(grid * 10 + shift)
    qkv = (torch.randn(B * grid * grid, 3 * heads * 32, generator=g) * 1.5).to(dt)
    table = torch.randn(169, heads, generator=g)
    ref = _window_attention_ref(qkv.float(), table, B, grid, heads, shift)
    got = ops.window_attention(qkv.to(gpu), rel_bias_lane_table(table).to(gpu), shift_mask_bits().to(gpu), B, grid,
                               heads, shift)
    tol = 2e-3 if dt == torch.float16 else 1.2e-2        # P and the output are rounded to the operand type
    assert _rel(got.cpu().float(), ref) < tol, _rel(got.cpu().float(), ref)


@pytest.mark.parametrize("grid,C", [(56, 96), (28, 192), (14, 384)])
def test_patch_merge_ln(gpu, grid, C):
    g = torch.Generator().manual_seed(C)
    x = torch.randn(2, grid * grid, C, generator=g) * 2 + 0.3
    gam, bet = 1 + 0.2 * torch.randn(4 * C, generator=g), 0.1 * torch.randn(4 * C, generator=g)
    g4 = x.reshape(2, grid // 2, 2, grid // 2, 2, C)
    cat = torch.cat([g4[:, :, 0, :, 0], g4[:, :, 1, :, 0], g4[:, :, 0, :, 1], g4[:, :, 1, :, 1]], -1).reshape(2, -1, 4 * C)
    ref = F.layer_norm(cat, (4 * C,), gam, bet, 1e-5)
    got = ops.patch_merge_ln(x.to(gpu), gam.to(gpu), bet.to(gpu), grid)
    assert got.shape == ref.shape and _rel(got.cpu().float(), ref) < 4e-4


@pytest.mark.parametrize("L,C", [(49, 768), (196, 192)])
def test_layernorm_meanpool(gpu, L, C):
    g = torch.Generator().manual_seed(L)
    x = torch.randn(3, L, C, generator=g) * 3 + 1
    gam, bet = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5).mean(1)
    o16, o32 = ops.layernorm_meanpool(x.to(gpu), gam.to(gpu), bet.to(gpu))
    assert (o32.cpu().double() - ref).abs().max().item() < 2e-6
    assert torch.equal(o16.cpu(), o32.cpu().half())


@pytest.mark.parametrize("N,K", [(96, 128), (288, 128), (192, 768), (576, 192)])
def test_gemm_n96_tile(gpu, N, K):
    """The 128x96 tile (Swin widths that 128 does not divide) against fp64 on the same rounded operands."""
    g = torch.Generator().manual_seed(N + K)
    M = 300
    a = torch.randn(M, K, generator=g).half()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    out = ops.gemm(a.to(gpu), w.to(gpu), _lib.EPI_BIAS_F32, bias=b.to(gpu))
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5 * K ** 0.5
    res = torch.randn(M, N, generator=g)
    out2 = ops.gemm(a.to(gpu), w.to(gpu), _lib.EPI_RESIDUAL, bias=b.to(gpu), out=res.clone().to(gpu))
    assert (out2.cpu().double() - (res.double() + ref)).abs().max().item() < 2e-5 * K ** 0.5
    out3 = ops.gemm(a.to(gpu), w.to(gpu), _lib.EPI_BIAS_GELU, bias=b.to(gpu))
    assert _rel(out3.cpu().float(), F.gelu(ref)) < 6e-4


@pytest.mark.parametrize("tag,preset", [("tiny", "test_swin_tiny"), ("swin_t", "ctranspath")])
def test_swin_matches_reference_golden(gpu, tag, preset):
    """End to end against the REFERENCE's output (not only the oracle's): tiles + seeded weights -> 768-d features.
    fp16 MFMA operands, fp32 accumulation and residual stream.  Stated tolerance: relative L2 <= 1e-3 on the fp32
    features (the oracle itself sits 2e-7 from the reference), and the fp16 output is the rounding of the fp32 one."""
    z = np.load(G / f"ctranspath_{tag}.npz")
    cfg = SWIN_PRESETS[preset]
    sd = random_swin_state_dict(cfg, int(z["seed"]))
    tiles = torch.from_numpy(z["tiles"])
    model = HipSwin(cfg, sd, device=gpu, chunk=3)                # chunk < B: exercises the chunk loop and ws reuse
    f16, f32 = model(tiles.to(gpu), return_f32=True)
    ref = torch.from_numpy(z["feats"])
    r = _rel(f32.cpu(), ref)
    mx = ((f32.cpu() - ref).abs().max() / ref.abs().max()).item()
    print(f"CTransPath {tag}: rel-L2 vs reference {r:.3e}, max-abs/max {mx:.3e}")
    assert r < 1e-3 and mx < 3e-3
    assert torch.equal(f16.cpu(), f32.cpu().half())
    # determinism + float-input path + batch invariance
    f16b = model(tiles.to(gpu))
    assert torch.equal(f16, f16b)
    x = tiles.permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor(cfg.mean).view(1, 3, 1, 1)) / torch.tensor(cfg.std).view(1, 3, 1, 1)
    assert torch.equal(model(x.to(gpu)), f16)
    assert torch.equal(model(tiles[1:2].to(gpu)), f16[1:2])


def test_swin_bf16_and_oracle_taps(gpu):
    """bf16 operands (looser) and a second seed against the oracle."""
    cfg = SWIN_PRESETS["test_swin_tiny"]
    sd = random_swin_state_dict(cfg, seed=11)
    tiles = torch.randint(0, 256, (5, cfg.img, cfg.img, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(12))
    ref = osw.swin_features(tiles, sd, cfg)
    for dt, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        _, f32 = HipSwin(cfg, sd, device=gpu, act_dtype=dt, chunk=8)(tiles.to(gpu), return_f32=True)
        assert _rel(f32.cpu(), ref) < tol, (dt, _rel(f32.cpu(), ref))


def test_swin_errors(gpu):
    cfg = SWIN_PRESETS["test_swin_tiny"]
    sd = random_swin_state_dict(cfg, seed=0)
    model = HipSwin(cfg, sd, device=gpu)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 224, 224, 3, dtype=torch.uint8, device=gpu))
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 112, 112, 3, dtype=torch.uint8))
    bad = dict(sd)
    bad.pop("norm.weight")
    with pytest.raises(KeyError):
        HipSwin(cfg, bad, device=gpu)
    assert model(torch.zeros(0, 112, 112, 3, dtype=torch.uint8, device=gpu)).shape == (0, 192)


@pytest.mark.parametrize("K,N,epi,ln", [(96, 288, "bias", True), (96, 384, "gelu", True), (96, 96, "res", False),
                                        (192, 576, "bias", True), (192, 768, "gelu", True), (192, 192, "res", False),
                                        (384, 96, "res", False), (384, 192, "f32", False)])
@pytest.mark.parametrize("M", [77, 4096 + 33])
def test_gemm_rowstream(gpu, K, N, epi, ln, M):
    """Weights-stationary narrow GEMM (+ fused LayerNorm) against fp64 on the same inputs; ragged M exercises the
    row clamp and the predicated stores, N > one LDS slice exercises the slicing."""
    g = torch.Generator().manual_seed(K + N + M)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    b = torch.randn(N, generator=g) * 0.3
    code = {"bias": _lib.EPI_BIAS, "gelu": _lib.EPI_BIAS_GELU, "res": _lib.EPI_RESIDUAL, "f32": _lib.EPI_BIAS_F32}[epi]
    if ln:
        x = torch.randn(M, K, generator=g) * 2 + 0.5
        gam, bet = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
        a_ref = F.layer_norm(x.double(), (K,), gam.double(), bet.double(), 1e-5)
        y = a_ref @ w.double().t() + b.double()
        got = ops.gemm_rowstream(x.to(gpu), w.to(gpu), code, bias=b.to(gpu), ln_gamma=gam.to(gpu), ln_beta=bet.to(gpu))
        ref = F.gelu(y) if epi == "gelu" else y
        assert got.dtype == torch.float16 and got.shape == (M, N)
        assert _rel(got.cpu().float(), ref) < 1e-3           # operand (LN output) and result rounded to fp16
    else:
        a = torch.randn(M, K, generator=g).half()
        y = a.double() @ w.double().t()
        if epi == "res":
            res = torch.randn(M, N, generator=g)
            got = ops.gemm_rowstream(a.to(gpu), w.to(gpu), code, bias=b.to(gpu), out=res.clone().to(gpu))
            ref = res.double() + y + b.double()
        else:
            got = ops.gemm_rowstream(a.to(gpu), w.to(gpu), code, bias=None)
            ref = y
        assert got.dtype == torch.float32
        assert (got.cpu().double() - ref).abs().max().item() < 3e-5 * K ** 0.5


@pytest.mark.parametrize("M", [64, 3136 * 3 + 17])
@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-4), (torch.bfloat16, 5e-3)])
def test_swin_mlp96_fused(gpu, M, dt, tol):
    """x += fc2(gelu(fc1(LN(x)))) in one kernel (hidden activation in MFMA registers) vs fp64 on the same weights."""
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, 96, generator=g) * 1.5 + 0.2
    w1 = (torch.randn(384, 96, generator=g) / 96 ** 0.5).to(dt)
    w2 = (torch.randn(96, 384, generator=g) * 0.5 / 384 ** 0.5).to(dt)
    b1, b2 = torch.randn(384, generator=g) * 0.2, torch.randn(96, generator=g) * 0.2
    gam, bet = 1 + 0.2 * torch.randn(96, generator=g), 0.1 * torch.randn(96, generator=g)
    h = F.layer_norm(x.double(), (96,), gam.double(), bet.double(), 1e-5)
    ref = x.double() + F.gelu(h @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    got = ops.swin_mlp96(x.clone().to(gpu), w1.to(gpu), b1.to(gpu), w2.to(gpu), b2.to(gpu), gam.to(gpu), bet.to(gpu))
    assert _rel(got.cpu(), ref) < tol, _rel(got.cpu(), ref)
    # the branch itself (result minus residual) to the same relative tolerance x10: catches a wrong k-permutation
    assert _rel(got.cpu().double() - x.double(), ref - x.double()) < 10 * tol


@pytest.mark.parametrize("grid,shift,B", [(56, 0, 2), (56, 3, 2), (14, 3, 3), (7, 0, 5)])
@pytest.mark.parametrize("dt,tol", [(torch.float16, 1.5e-3), (torch.bfloat16, 1.2e-2)])
def test_swin_attn96_fused(gpu, grid, shift, B, dt, tol):
    """x += proj(window_attention(qkv(LN(x)))) in one kernel (q, k, v, P, head outputs only in registers) against the oracle's
    block arithmetic in fp64 on the same weights (oracle.swin_ctranspath.swin_block, attention half)."""
    g = torch.Generator().manual_seed(grid * 10 + shift)
    C, H = 96, 3
    x = torch.randn(B, grid * grid, C, generator=g) * 1.5 + 0.1
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dt)
    wproj = (torch.randn(C, C, generator=g) * 0.5 / C ** 0.5).to(dt)
    bqkv, bproj = torch.randn(3 * C, generator=g) * 0.2, torch.randn(C, generator=g) * 0.2
    gam, bet = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    table = torch.randn(169, H, generator=g) * 0.7
    # reference: the oracle's swin_block with an identity MLP half (fc2 = 0) in fp64
    sd = {"b.norm1.weight": gam, "b.norm1.bias": bet, "b.attn.qkv.weight": wqkv.float(), "b.attn.qkv.bias": bqkv,
          "b.attn.proj.weight": wproj.float(), "b.attn.proj.bias": bproj, "b.attn.relative_position_bias_table": table,
          "b.norm2.weight": torch.ones(C), "b.norm2.bias": torch.zeros(C), "b.mlp.fc1.weight": torch.zeros(4 * C, C),
          "b.mlp.fc1.bias": torch.zeros(4 * C), "b.mlp.fc2.weight": torch.zeros(C, 4 * C), "b.mlp.fc2.bias": torch.zeros(C)}
    ref = osw.swin_block(x.double(), {k: v.double() for k, v in sd.items()}, "b.", grid, grid, H, shift)
    got = ops.swin_attn96(x.clone().to(gpu), wqkv.to(gpu), bqkv.to(gpu), wproj.to(gpu), bproj.to(gpu), gam.to(gpu), bet.to(gpu),
                          rel_bias_lane_table(table).to(gpu), shift_mask_bits().to(gpu), grid, shift)
    assert _rel(got.cpu(), ref) < tol, _rel(got.cpu(), ref)
    assert _rel(got.cpu().double() - x.double(), ref - x.double()) < 10 * tol       # the branch alone


@pytest.mark.parametrize("M", [128, 784 * 5 + 9])
@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-4), (torch.bfloat16, 5e-3)])
def test_swin_mlp192_streamed(gpu, M, dt, tol):
    """Stage-2 MLP branch with weights streamed through LDS from the pre-packed fragment image, vs fp64."""
    g = torch.Generator().manual_seed(M + 1)
    C, Hd = 192, 768
    x = torch.randn(M, C, generator=g) * 1.5 + 0.2
    w1 = (torch.randn(Hd, C, generator=g) / C ** 0.5).to(dt)
    w2 = (torch.randn(C, Hd, generator=g) * 0.5 / Hd ** 0.5).to(dt)
    b1, b2 = torch.randn(Hd, generator=g) * 0.2, torch.randn(C, generator=g) * 0.2
    gam, bet = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    h = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-5)
    ref = x.double() + F.gelu(h @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    pk = ops.swin_mlp192_pack(w1.to(gpu), w2.to(gpu))
    got = ops.swin_mlp192(x.clone().to(gpu), pk, b1.to(gpu), b2.to(gpu), gam.to(gpu), bet.to(gpu))
    assert _rel(got.cpu(), ref) < tol, _rel(got.cpu(), ref)
    assert _rel(got.cpu().double() - x.double(), ref - x.double()) < 10 * tol
