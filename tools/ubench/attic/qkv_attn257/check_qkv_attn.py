"""amds_qkv_attention_vit257 (csrc/qkv_attn257.hip): the qkv Linear and the attention of a ViT block as one kernel, against the two launches it replaces
(amds_gemm_lnfold / amds_gemm + amds_attention_vit) -- k | v are the same bits (same operand rounding, same K order), q is rounded after the softmax
scale instead of before it, and the softmax is taken in two passes instead of online, so the outputs agree to rounding, not bit for bit -- and against a
plain fp32 torch restatement of timm's Attention.forward (what the reference runs inside `model(tiles)`, src/stamp/preprocessing/__init__.py:324-325)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, H, dtype, seed, fold=True):
    g = torch.Generator().manual_seed(seed)
    D = H * 64
    x = (torch.randn(B * 257, D, generator=g) * 1.5 + 0.3).to(dtype)
    w = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to(dtype)
    bias = torch.randn(3 * D, generator=g) * 0.2
    if fold:
        xf = x.float()
        mean, var = xf.mean(1), xf.var(1, unbiased=False)
        rstd = (var + 1e-6).rsqrt()
        rowstat = torch.stack([rstd, -mean * rstd], 1).contiguous()
        colsum = w.float().sum(1)
    else:
        rowstat = colsum = None
    return x, w, bias, rowstat, colsum


def _torch_ref(x, w, bias, rowstat, colsum, B, H):
    """fp32 throughout (the operands as given; NO rounding of q | k | v: the fused kernel rounds q * scale where the two launches round q -- two
    equally valid 16-bit roundings, and a reference that applied one of them would favour that path)."""
    D = H * 64
    xf, wf = x.float(), w.float()
    if rowstat is not None:
        xf = xf * rowstat[:, :1] + rowstat[:, 1:]               # (x - mean) rstd: gamma / beta live in w / bias
    qkv = (xf @ wf.t() + bias).view(B, 257, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) / 8.0, -1) @ qkv[2]
    return a.permute(0, 2, 1, 3).reshape(B * 257, D)


@pytest.mark.parametrize("B,H,dtype,fold", [(5, 4, torch.float16, True), (3, 16, torch.float16, True), (9, 8, torch.bfloat16, True), (2, 4, torch.float16, False),
                                            (33, 12, torch.float16, True), (1, 16, torch.float16, True)])
def test_fused_equals_the_two_launches_it_replaces(B, H, dtype, fold):
    from stamp_amd import _lib, ops
    dev = torch.device("cuda:0")
    x, w, bias, rowstat, colsum = [t.to(dev) if t is not None else None for t in _case(B, H, dtype, 100 + B + H, fold)]
    if fold:
        qkv = ops.gemm_lnfold(x, w, _lib.EPI_BIAS, bias=bias, rowstat=rowstat, colsum=colsum)
    else:
        qkv = ops.gemm(x, w, _lib.EPI_BIAS, bias=bias, cfg=12)
    want = ops.attention_vit(qkv, B, 257, H)
    got = ops.qkv_attention_vit257(x, w, bias, B, H, rowstat=rowstat, colsum=colsum)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    ref = _torch_ref(x.cpu(), w.cpu(), bias.cpu(), rowstat.cpu() if fold else None, colsum, B, H)
    rel = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < (2e-3 if dtype == torch.float16 else 1.5e-2), rel
    rel2 = ((got.float() - want.float()).norm() / want.float().norm()).item()
    assert rel2 < (1e-3 if dtype == torch.float16 else 8e-3), f"fused vs gemm + attention: rel-L2 {rel2:.3e}"
    want_rel = ((want.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 1.25 * want_rel + 1e-5, (rel, want_rel)          # no further from the fp32 restatement than the two launches are


def test_fused_is_deterministic_and_independent_of_the_batch_it_travels_in():
    from stamp_amd import ops
    dev = torch.device("cuda:0")
    B, H = 11, 16
    x, w, bias, rowstat, colsum = [t.to(dev) for t in _case(B, H, torch.float16, 7)]
    a = ops.qkv_attention_vit257(x, w, bias, B, H, rowstat=rowstat, colsum=colsum)
    b = ops.qkv_attention_vit257(x, w, bias, B, H, rowstat=rowstat, colsum=colsum)
    assert torch.equal(a, b)
    k = 4
    c = ops.qkv_attention_vit257(x[k * 257:(k + 3) * 257].contiguous(), w, bias, 3, H, rowstat=rowstat[k * 257:(k + 3) * 257].contiguous(), colsum=colsum)
    assert torch.equal(a[k * 257:(k + 3) * 257], c)


def test_tile_encoder_features_do_not_change_with_the_fused_kernel():
    """HipViT (ViT-L/14 shapes, reduced depth) with the fused kernel (opt-in: AMDS_VIT_QKVATTN=1) and without: the stored features agree to a fraction of
    the 1e-3 parity budget (two-pass vs online softmax rounding)."""
    import dataclasses
    from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict
    dev = torch.device("cuda:0")
    cfg = dataclasses.replace(PRESETS["vit_large_patch14_224"], depth=3)
    sd = random_vit_state_dict(cfg, seed=5)
    tiles = torch.randint(0, 256, (13, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(6)).to(dev)
    m = HipViT(cfg, sd, device=dev, chunk=8)
    old = os.environ.get("AMDS_VIT_QKVATTN")
    try:
        os.environ["AMDS_VIT_QKVATTN"] = "0"
        a = m(tiles).clone()
        os.environ["AMDS_VIT_QKVATTN"] = "1"
        b = m(tiles).clone()
    finally:
        if old is None:
            os.environ.pop("AMDS_VIT_QKVATTN", None)
        else:
            os.environ["AMDS_VIT_QKVATTN"] = old
    assert torch.isfinite(b.float()).all()
    rel = ((a.float() - b.float()).norm() / a.float().norm()).item()
    assert rel < 4e-4, rel
