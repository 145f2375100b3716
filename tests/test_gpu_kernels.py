"""Parity of each HIP building block (called through the C ABI) against a plain torch fp32 statement of the
same op on identical inputs.  Tolerances are stated per test; operands are rounded to the activation dtype
BEFORE the fp32 reference matmul so that only accumulation order / output rounding differ."""
import pytest
import torch

from stamp_amd import _lib, ops

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]


def _eps(dt):
    return 2 ** -10 if dt == torch.float16 else 2 ** -7


@pytest.mark.parametrize("rows,cols", [(1, 128), (257, 1024), (1000, 1536), (33, 1280), (5, 4096), (3, 8192)])
@pytest.mark.parametrize("odt", [torch.float16, torch.bfloat16, torch.float32])
def test_layernorm(gpu, rows, cols, odt):
    g = torch.Generator(device="cpu").manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, cols, generator=g) * 3 + 1.5).to(gpu)
    w = (1 + 0.3 * torch.randn(cols, generator=g)).to(gpu)
    b = (0.2 * torch.randn(cols, generator=g)).to(gpu)
    y = ops.layernorm(x, w, b, 1e-6, odt)
    ref = torch.nn.functional.layer_norm(x.double(), (cols,), w.double(), b.double(), 1e-6)
    tol = 2e-6 * 8 if odt == torch.float32 else _eps(odt)
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err


def _gemm_ref(a, w, bias):
    return a.double() @ w.double().t() + (bias.double() if bias is not None else 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [0, 8, 10, 12])
@pytest.mark.parametrize("M,N,K", [(257, 384, 128), (1000, 1024, 1024), (64, 128, 64), (513, 256, 640), (2570, 3072, 1024)])
def test_gemm_bias_asymmetric(gpu, dt, cfg, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(gpu, dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, dt)
    # asymmetric structure: a transposed / row<->col swapped C write cannot pass
    a[:, 0] += torch.arange(M, device=gpu).to(dt) * 0.01
    bias = torch.randn(N, generator=g).to(gpu)
    out = ops.gemm(a, w, _lib.EPI_BIAS_F32, bias=bias, cfg=cfg)
    ref = _gemm_ref(a, w, bias)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-3 * max(1.0, ref.abs().max().item()) * (1 if dt == torch.float16 else 1), err
    # fp32 accumulate of exactly-representable products: error is accumulation-order only
    assert (out.double() - ref).abs().mean().item() < 1e-5 * ref.abs().mean().item() + 1e-6


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 256, 128), (300, 256, 64), (255, 256, 192), (256, 512, 128), (257, 256, 256), (1000, 1024, 640),
                                   (4099, 768, 1024), (777, 256, 4096), (20000, 1024, 1024)])
@pytest.mark.parametrize("cfg", [8, 10, 12, 13])
def test_gemm_8phase_pipeline(gpu, dt, M, N, K, cfg):
    """The pipelined kernels (8: staggered two-group 8-wave; 10 / 12: four waves, 128x128 wave tiles):
    exact-shape sweep incl. the minimum K, ragged M (rows past M are out of range of the LDS-DMA buffer descriptor), and a
    race screen -- 6 launches must be bitwise identical and match an fp64 reference."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a = torch.randn(M, K, generator=g).to(gpu, dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, dt)
    a[:, 0] += torch.arange(M, device=gpu).to(dt) * 0.01
    bias = torch.randn(N, generator=g).to(gpu)
    ref = _gemm_ref(a, w, bias)
    outs = [ops.gemm(a, w, _lib.EPI_BIAS_F32, bias=bias, cfg=cfg) for _ in range(6)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    err = (outs[0].double() - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()) * (K / 1024 + 1), err


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [0, 8, 10, 12, 13])
def test_gemm_epilogues(gpu, dt, cfg):
    M, N, K = 771, 512, 256
    _g = ops.gemm
    class _O:      # route every call of this test through the chosen kernel family
        @staticmethod
        def gemm(*a, **k):
            k.setdefault("cfg", cfg)
            return _g(*a, **k)
    ops_ = _O
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(gpu, dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, dt)
    bias = torch.randn(N, generator=g).to(gpu)
    ref = _gemm_ref(a, w, bias)
    e = _eps(dt)
    out = ops_.gemm(a, w, _lib.EPI_BIAS, bias=bias)
    assert out.dtype == dt and (out.double() - ref).abs().max().item() < e * ref.abs().max().item() * 1.01
    out = ops_.gemm(a, w, _lib.EPI_BIAS_GELU, bias=bias)
    r = torch.nn.functional.gelu(ref)
    assert (out.double() - r).abs().max().item() < e * r.abs().max().item() * 1.01 + 1e-6
    out = ops_.gemm(a, w, _lib.EPI_BIAS_RELU, bias=bias)
    assert (out.double() - ref.clamp(min=0)).abs().max().item() < e * ref.abs().max().item() * 1.01
    out = ops_.gemm(a, w, _lib.EPI_BIAS_GELU_F32, bias=bias)
    assert (out.double() - r).abs().max().item() < 2e-5
    out = ops_.gemm(a, w, _lib.EPI_BIAS_RELU_F32, bias=bias)
    assert (out.double() - ref.clamp(min=0)).abs().max().item() < 2e-5
    # residual with LayerScale
    x0 = torch.randn(M, N, generator=g).to(gpu)
    scale = torch.rand(N, generator=g).to(gpu)
    x = x0.clone()
    ops_.gemm(a, w, _lib.EPI_RESIDUAL, bias=bias, scale=scale, out=x)
    r = x0.double() + scale.double() * ref
    assert (x.double() - r).abs().max().item() < 2e-5
    x = x0.clone()
    ops_.gemm(a, w, _lib.EPI_RESIDUAL, bias=None, scale=None, out=x)
    assert (x.double() - (x0.double() + _gemm_ref(a, w, None))).abs().max().item() < 2e-5
    # acc_scale
    out = ops_.gemm(a, w, _lib.EPI_BIAS_F32, bias=bias, acc_scale=0.25)
    assert (out.double() - (0.25 * _gemm_ref(a, w, None) + bias.double())).abs().max().item() < 2e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_kernels_bit_identical(gpu, dt):
    """The 256-wide kernels built on v_mfma 32x32x16 (ids 8, 10) accumulate k in the same order: bit-identical outputs.
    Id 12 (v_mfma 16x16x32, the library default) sums 32 products per instruction: last-bit differences only, and it is the
    kernel the default dispatch picks for this shape."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 1500, 768, 320
    a = torch.randn(M, K, generator=g).to(gpu, dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, dt)
    bias = torch.randn(N, generator=g).to(gpu)
    for epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_GELU, _lib.EPI_BIAS_F32):
        ref = ops.gemm(a, w, epi, bias=bias, cfg=8)
        for cfg in (10,):
            assert torch.equal(ops.gemm(a, w, epi, bias=bias, cfg=cfg), ref), (epi, cfg)
        o12 = ops.gemm(a, w, epi, bias=bias, cfg=12)
        ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10          # one unit in the last place at the top of the output range
        assert (o12.float() - ref.float()).abs().max().item() <= ulp * ref.float().abs().max().item(), epi
        for cfg in (13,):                                               # the other LDS-DMA schedules of id 12: same arithmetic, same bits
            assert torch.equal(ops.gemm(a, w, epi, bias=bias, cfg=cfg), o12), (epi, cfg)
    # a grid that fills the chip (>= 192 tiles of 256 x 256): the default dispatch is id 12
    M2, N2 = 4099, 3072
    a2 = torch.randn(M2, K, generator=g).to(gpu, dt)
    w2 = (torch.randn(N2, K, generator=g) / K ** 0.5).to(gpu, dt)
    b2 = torch.randn(N2, generator=g).to(gpu)
    assert torch.equal(ops.gemm(a2, w2, _lib.EPI_BIAS_F32, bias=b2), ops.gemm(a2, w2, _lib.EPI_BIAS_F32, bias=b2, cfg=12))


def test_gemm_bench_size_sampled_rows(gpu):
    """The four GEMMs of one ViT-L block at the bench chunk (M = 262 140, library-default kernel): 96 sampled rows -- the first
    and the last row tile included -- against an fp64 reference; the residual epilogue read-modify-writes its output."""
    g = torch.Generator().manual_seed(3)
    M = 262140
    rows = torch.cat([torch.arange(0, 32), torch.randint(0, M, (32,), generator=g), torch.arange(M - 32, M)]).to(gpu)
    for N, K, epi in ((3072, 1024, _lib.EPI_BIAS), (1024, 1024, _lib.EPI_RESIDUAL), (4096, 1024, _lib.EPI_BIAS_GELU), (1024, 4096, _lib.EPI_RESIDUAL)):
        a = torch.randn(M, K, device=gpu).half()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu).half()
        bias = torch.randn(N, generator=g).to(gpu)
        x0 = torch.randn(M, N, device=gpu) if epi == _lib.EPI_RESIDUAL else None
        out = ops.gemm(a, w, epi, bias=bias, out=x0.clone() if x0 is not None else None)
        ref = a[rows].double() @ w.double().t() + bias.double()
        if epi == _lib.EPI_BIAS_GELU:
            ref = torch.nn.functional.gelu(ref)
        if epi == _lib.EPI_RESIDUAL:
            ref = ref + x0[rows].double()
        tol = 2e-5 * (K / 1024) if epi == _lib.EPI_RESIDUAL else 1.5e-3
        err = (out[rows].double() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item()), (N, K, epi, err)
        del a, w, out, x0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg,M,H,K", [(-1, 300, 192, 128), (12, 777, 256, 192), (8, 777, 256, 192), (12, 5000, 1024, 64)])
def test_gemm_swiglu(gpu, dt, cfg, M, H, K):
    g = torch.Generator().manual_seed(9)
    a = torch.randn(M, K, generator=g).to(gpu, dt)
    w = (torch.randn(2 * H, K, generator=g) / K ** 0.5).to(gpu)
    bias = torch.randn(2 * H, generator=g).to(gpu)
    wp = ops.pack_swiglu_rows(w).to(dt)
    bp = ops.pack_swiglu_rows(bias.reshape(-1, 1)).reshape(-1)
    # padded N must be a multiple of 128: 384 ok
    out = ops.gemm(a, wp, _lib.EPI_SWIGLU, bias=bp, cfg=cfg)
    full = a.double() @ w.to(dt).double().t() + bias.double()
    x1, x2 = full[:, :H], full[:, H:]
    ref = torch.nn.functional.silu(x1) * x2
    assert out.shape == (M, H)
    assert (out.double() - ref).abs().max().item() < _eps(dt) * ref.abs().max().item() * 1.05 + 1e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_patch_epilogue(gpu, dt):
    B, np_, P, D, K = 3, 256, 5, 128, 640
    T = np_ + P
    g = torch.Generator().manual_seed(11)
    a = torch.randint(0, 256, (B * np_, K), generator=g).to(gpu, dt)
    w = (torch.randn(D, K, generator=g) / K ** 0.5).to(gpu, dt)
    bias = torch.randn(D, generator=g).to(gpu)
    pos = torch.randn(np_, D, generator=g).to(gpu)
    x = torch.full((B * T, D), 7.0, device=gpu)
    ops.gemm(a, w, _lib.EPI_PATCH, bias=bias, pos=pos, np_=np_, T=T, P=P, acc_scale=1 / 255.0, out=x)
    ref = (a.double() @ w.double().t()) / 255.0 + bias.double()
    ref = ref.reshape(B, np_, D) + pos.double()
    xx = x.reshape(B, T, D)
    assert (xx[:, :P] == 7.0).all()
    assert (xx[:, P:].double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,H", [(2, 257, 16), (3, 261, 2), (1, 265, 24), (2, 197, 4), (2, 64, 2), (1, 33, 1), (2, 288, 2),
                                   (40, 257, 16),       # 640 (tile, head) items: the persistent T = 257 kernel walks 2-3 items per workgroup
                                   (30, 265, 24), (45, 261, 16), (11, 265, 24)])      # the T = 256 + R pipeline (attention_vit26x.hip): 2.8 / 2.8 / 1.03 items per workgroup
def test_attention_vit(gpu, dt, B, T, H):
    g = torch.Generator().manual_seed(B * 1000 + T + H)
    D = H * 64
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(gpu, dt)
    out = ops.attention_vit(qkv, B, T, H)
    q, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(B * T, D)
    err = (out.double() - ref).abs().max().item()
    # P and the output are rounded to the act dtype: 2-3 ulp of the largest value
    assert err < 4 * _eps(dt) * max(1.0, ref.abs().max().item()), err
    if T > 256:          # the tail rows (class / register tokens past the 256 main tokens) on their own, and run-to-run determinism
        et = (out.double() - ref).reshape(B, T, D)[:, 256:].abs().max().item()
        assert et < 4 * _eps(dt) * max(1.0, ref.abs().max().item()), et
        assert torch.equal(out, ops.attention_vit(qkv, B, T, H))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,H", [(2, 1025, 8), (1, 300, 2), (1, 64, 1), (1, 63, 1), (3, 129, 4), (1, 5000, 2), (2, 257, 16), (1, 1, 1)])
def test_attention_streaming_any_length(gpu, dt, B, T, H):
    g = torch.Generator().manual_seed(B * 77 + T + H)
    D = H * 64
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(gpu, dt)
    out = ops.attention(qkv, B, T, H)
    q, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(B * T, D)
    err = (out.double() - ref).abs().max().item()
    assert err < 4 * _eps(dt) * max(1.0, ref.abs().max().item()), err
    assert torch.equal(out, ops.attention(qkv, B, T, H))
    if T <= 288:      # the two kernels implement the same op
        o2 = ops.attention_vit(qkv, B, T, H)
        assert (out.double() - o2.double()).abs().max().item() < 4 * _eps(dt) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,H", [(3, 1025, 8), (2, 212, 4), (1, 1, 1), (5, 63, 2), (2, 5000, 3)])
def test_attention_row_is_the_class_tokens_row_of_the_full_attention(gpu, dt, B, T, H):
    """amds_attention_row: ONE query per (bag, head) against all keys / values of the packed qkv -- the class token's attention in the last block of the MIL `vit`
    head, whose other rows nothing reads (vision_tranformer.py: the head takes x[:, 0]).  Against fp64 softmax(q k^T / 8) v on the same 16-bit inputs (its weights
    stay fp32: closer than the streaming kernel's row), and within the streaming kernel's own tolerance of that kernel's row 0; deterministic."""
    g = torch.Generator().manual_seed(B * 100 + T + H)
    D = H * 64
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(gpu, dt)
    q = qkv.view(B, T, 3 * D)[:, 0, :D]                                   # the class rows' queries, in place (row pitch T * 3 D)
    out = ops.attention_row(q, qkv, B, T, H)
    qq, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(qq[:, :, :1] @ k.transpose(-1, -2) / 8.0, -1) @ v).reshape(B, D)
    tol = 2 * _eps(dt) * max(1.0, ref.abs().max().item())
    assert (out.double() - ref).abs().max().item() < tol
    full = ops.attention(qkv, B, T, H).view(B, T, D)[:, 0]
    assert (out.double() - full.double()).abs().max().item() < 4 * _eps(dt) * max(1.0, ref.abs().max().item())
    assert torch.equal(out, ops.attention_row(q, qkv, B, T, H))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,T,H", [(2, 261, 16), (1, 257, 2), (3, 100, 1), (1, 288, 3), (2, 33, 2)])
def test_attention_vit_head_dim_80(gpu, dt, B, T, H):
    g = torch.Generator().manual_seed(B * 31 + T + H)
    D = H * 80
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(gpu, dt)
    out = ops.attention_vit(qkv, B, T, H, head_dim=80)
    q, k, v = qkv.double().reshape(B, T, 3, H, 80).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 80 ** 0.5, -1) @ v).transpose(1, 2).reshape(B * T, D)
    err = (out.double() - ref).abs().max().item()
    assert err < 4 * _eps(dt) * max(1.0, ref.abs().max().item()), err


def test_attention_softmax_spike(gpu):
    """one key dominating a row / large logits: exercises the running-max rescale across chunks"""
    B, T, H = 1, 257, 1
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * T, 192, generator=g)
    qkv[:, :64] *= 4
    qkv[200, 64:128] = qkv[5, :64] * 3       # key 200 (third chunk) spikes for query 5
    qkv[10, 64:128] = qkv[7, :64] * 3        # key 10 (first chunk) spikes for query 7
    qkv = qkv.to(gpu, torch.float16)
    out = ops.attention_vit(qkv, B, T, H)
    q, k, v = qkv.double().reshape(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(B * T, 64)
    assert torch.isfinite(out).all()
    assert (out.double() - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("patch", [14, 16])
def test_im2col(gpu, patch):
    B, img = 3, 224
    g = torch.Generator().manual_seed(patch)
    tiles = torch.randint(0, 256, (B, img, img, 3), dtype=torch.uint8, generator=g).to(gpu)
    kp = (3 * patch * patch + 63) // 64 * 64
    out = ops.tile_im2col_u8(tiles, patch, kp, torch.float16)
    x = tiles.permute(0, 3, 1, 2).float()
    ref = torch.nn.functional.unfold(x, patch, stride=patch).transpose(1, 2).reshape(-1, 3 * patch * patch)
    assert torch.equal(out[:, : 3 * patch * patch].float(), ref)      # integers 0..255: exact
    assert (out[:, 3 * patch * patch:] == 0).all()


def test_tile_normalize(gpu):
    from oracle.vit_tile_encoder import tile_transform

    mean, std = (0.707223, 0.578729, 0.703617), (0.211883, 0.230117, 0.177517)
    tiles = torch.randint(0, 256, (5, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    out = ops.tile_normalize_u8(tiles.to(gpu), mean, std).cpu()
    ref = tile_transform(tiles, mean, std)
    # one fp32 division + subtraction + multiply-by-reciprocal vs division: <= 2 ulp
    assert (out - ref).abs().max().item() <= 4e-7 * ref.abs().max().item()


def test_errors_are_values(gpu):
    a = torch.zeros(8, 100, dtype=torch.float16, device=gpu)   # K not a multiple of 64
    w = torch.zeros(128, 100, dtype=torch.float16, device=gpu)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm(a, w, _lib.EPI_BIAS_F32)
    with pytest.raises(RuntimeError, match="288"):
        ops.attention_vit(torch.zeros(400, 192, dtype=torch.float16, device=gpu), 1, 400, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("epi_name", ["EPI_BIAS", "EPI_BIAS_F32", "EPI_RESIDUAL"])
def test_gemm_ragged_row_tile_as_its_own_launch(gpu, epi_name):
    """cfg -2 (the MIL training step: M = 64 bags x 1025 tokens = 256.25 row tiles, N = 512 -> 514 workgroups): the last, partial row tile runs
    through the 128 x 128 kernel when that saves a wave.  Rows of the full tiles are the production kernel's bit for bit; the 64 rows of the
    ragged tile match the fp64 product to the output type's rounding."""
    epi = getattr(_lib, epi_name)
    M, N, K = 64 * 1025, 512, 512
    g = torch.Generator().manual_seed(11)
    a = torch.randn(M, K, generator=g).to(gpu, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, torch.bfloat16)
    b = torch.randn(N, generator=g).to(gpu)
    f32 = epi != _lib.EPI_BIAS
    base = torch.randn(M, N, generator=g).to(gpu) if epi == _lib.EPI_RESIDUAL else None
    outs = []
    for cfg in (12, -2):
        out = base.clone() if base is not None else None
        outs.append(ops.gemm(a, w, epi, bias=b, out=out, cfg=cfg))
    full = (M // 256) * 256
    assert torch.equal(outs[0][:full], outs[1][:full])
    ref = a[full:].double() @ w.double().T + b.double()
    if base is not None:
        ref = ref + base[full:].double()
    tol = (1e-4 if f32 else 2 * _eps(torch.bfloat16)) * max(1.0, ref.abs().max().item())
    for o in outs:
        assert (o[full:].double() - ref).abs().max().item() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,use_scale", [(1000, 1024, 320, True), (515, 512, 1024, False), (256, 256, 64, True)])
def test_gemm_lnfold_producer(gpu, dt, M, N, K, use_scale):
    """amds_gemm_lnfold, producer form: the fp32 rows are updated exactly as by the plain RESIDUAL epilogue; xh is their 16-bit
    rounding; rowpart holds (sum, sum of squares) per 128-column slab -- slab (tile tn, pass p) = columns tn*256 + {64p .. 64p+63} and
    tn*256 + 128 + {64p .. 64p+63} -- and amds_ln_rowstat turns them into (rstd, -mean*rstd)."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(gpu, dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, dt)
    bias = torch.randn(N, generator=g).to(gpu)
    scale = (0.5 + torch.rand(N, generator=g)).to(gpu) if use_scale else None
    x0 = torch.randn(M, N, generator=g).to(gpu)
    ref = x0.clone()
    ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=bias, scale=scale, out=ref, cfg=12)
    x = x0.clone()
    xh = torch.full((M, N), 7.0, dtype=dt, device=gpu)
    rowpart = torch.full((M, N // 128, 2), -1.0, device=gpu)
    ops.gemm_lnfold(a, w, _lib.EPI_RESIDUAL, out=x, bias=bias, scale=scale, xh=xh, rowpart=rowpart)
    assert torch.equal(x, ref)
    assert torch.equal(xh, ref.to(dt))
    cols = ref.view(M, N // 256, 2, 2, 64)                       # [tile, 128-half, pass, 64]
    slab = cols.permute(0, 1, 3, 2, 4).reshape(M, N // 128, 128).double()
    want = torch.stack([slab.sum(-1), (slab * slab).sum(-1)], -1)
    assert torch.allclose(rowpart.double(), want, rtol=2e-5, atol=1e-3)
    rs = ops.ln_rowstat(rowpart, N, 1e-6)
    mean = ref.double().mean(1)
    rstd = 1.0 / torch.sqrt(ref.double().var(1, unbiased=False) + 1e-6)
    assert torch.allclose(rs[:, 0].double(), rstd, rtol=1e-4)
    assert torch.allclose(rs[:, 1].double(), -mean * rstd, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [12, -1])
def test_gelu_epilogue_polynomial_over_its_whole_range(gpu, cfg):
    """The BIAS_GELU epilogue evaluates erf by a degree-7 polynomial in x^2 (common.h, tools/gelu_poly_fit.py; round 5: degree 8 -> 7, profiles/r05_gelu_degree_ab.txt).  A GEMM with an identity
    weight and the test values as bias puts exact pre-activations into the accumulators: a dense sweep of [-8, 8], the clamp points +-3 sqrt2,
    and massive values.  Bars: |error| <= 5e-5 |x| (measured 4.6e-5: a tenth of half an fp16 ulp where it is largest) + half an fp16 ulp of the result inside; beyond the
    clamp 0 (negative side, to 5e-8 |x|) or x."""
    import math
    N = K = 256
    xs = torch.cat([torch.linspace(-8, 8, 256 * 1021), torch.tensor([-3 * math.sqrt(2), 3 * math.sqrt(2), -4.2426, 4.2427, -6.0, 6.0, -30.0, 30.0, -1e3, 1e3, -6e4, 6e4, 0.0, -0.0])])
    pad = (-xs.numel()) % N
    xs = torch.cat([xs, torch.zeros(pad)]).reshape(-1, N).to(gpu)
    M = xs.shape[0]
    a = torch.zeros(M, K, dtype=torch.float16, device=gpu)
    w = torch.eye(N, K, dtype=torch.float16, device=gpu)
    # the value enters through an fp32 "residual-free" route: acc = 0 * w + bias is per column, so feed rows through A instead: A = hi + lo of x
    hi = xs.half()
    lo = (xs - hi.float()).half()
    a2 = torch.cat([hi, lo], 1)                         # [M, 2K]: x = hi + lo reproduces the fp32 value to 2^-22
    w2 = torch.cat([w, w], 1)
    out = ops.gemm(a2, w2, _lib.EPI_BIAS_GELU, bias=torch.zeros(N, device=gpu), cfg=cfg).double()
    x = hi.double() + lo.double()
    ref = torch.nn.functional.gelu(x)
    half_ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14, device=gpu, dtype=torch.float64)) * 2.0 ** -11
    err = (out - ref).abs()
    assert bool((err <= 5e-5 * x.abs() + 1.01 * half_ulp).all()), float((err - 5e-5 * x.abs() - half_ulp).max())
    neg_tail, pos_tail = x < -4.3, x > 4.3
    assert bool((out[neg_tail].abs() <= 5e-8 * x[neg_tail].abs() + 6e-8).all()) and torch.equal(out[pos_tail], x[pos_tail].half().double())
    assert torch.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,use_scale", [(1000, 1024, 320, True), (515, 512, 1024, False), (256, 256, 64, True)])
def test_gemm_lnfold_planes(gpu, M, N, K, use_scale):
    """amds_gemm_lnfold_planes: the residual rows as two fp16 planes.  The fp32 value the epilogue forms is bit for bit what the fp32-row
    RESIDUAL epilogue writes for the same old value hi + lo; the planes hold its fp16 rounding and the fp16 rounding of what that left over;
    the partial row sums are those of the fp32 value.  Then 48 updates in a row (a 24-block trunk): the pair drifts from an fp64 running sum
    by ~1e-7 of the row's magnitude -- the fp32 rows it replaces drift by about as much."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(gpu, torch.float16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu, torch.float16)
    bias = torch.randn(N, generator=g).to(gpu)
    scale = (0.5 + torch.rand(N, generator=g)).to(gpu) if use_scale else None
    x0 = (torch.randn(M, N, generator=g) * torch.logspace(-3, 1, N)).to(gpu)               # columns from 1e-3 to 10: lo reaches fp16's subnormals
    hi, lo, rs0 = ops.ln_stats_split(x0, 1e-6)
    assert torch.equal(hi, x0.half()) and torch.equal(lo, (x0 - hi.float()).half())
    xh_ref, rs_ref = ops.ln_stats_cast(x0, 1e-6, torch.float16)
    assert torch.equal(hi, xh_ref) and torch.equal(rs0, rs_ref)
    xin = ops.planes_to_f32(hi, lo)
    assert torch.equal(xin, hi.float() + lo.float())
    assert ((xin - x0).abs() <= 2.0 ** -21 * x0.abs() + 2.0 ** -25).all()                  # ~22 bits, or fp16's subnormal spacing / 2
    assert torch.equal(ops.planes_to_f32(hi, lo, 7), xin[::7])
    ref = xin.clone()
    ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=bias, scale=scale, out=ref, cfg=12)
    rowpart = ops.gemm_lnfold_planes(a, w, hi, lo, bias=bias, scale=scale)
    assert torch.equal(hi, ref.half()) and torch.equal(lo, (ref - hi.float()).half())
    cols = ref.view(M, N // 256, 2, 2, 64)
    slab = cols.permute(0, 1, 3, 2, 4).reshape(M, N // 128, 128).double()
    want = torch.stack([slab.sum(-1), (slab * slab).sum(-1)], -1)
    assert torch.allclose(rowpart.double(), want, rtol=2e-5, atol=1e-3)
    # drift over a trunk's worth of updates, against fp64 (the increments themselves taken from an fp32-row run of the same GEMM)
    x32 = ops.planes_to_f32(hi, lo)
    inc = ops.gemm(a, w, _lib.EPI_BIAS_F32, bias=bias, cfg=12).double() * (scale.double() if use_scale else 1.0)      # the same increment every time
    x64 = x32.double() + 48 * inc
    for _ in range(48):
        ops.gemm(a, w, _lib.EPI_RESIDUAL, bias=bias, scale=scale, out=x32, cfg=12)
        ops.gemm_lnfold_planes(a, w, hi, lo, bias=bias, scale=scale)
    xp = ops.planes_to_f32(hi, lo).double()
    d_pl = ((xp - x64).norm() / x64.norm()).item()
    d_32 = ((x32.double() - x64).norm() / x64.norm()).item()
    print(f"48 residual updates, M={M} N={N}: (hi | lo) planes drift {d_pl:.2e}, fp32 rows {d_32:.2e} (relative L2 vs an fp64 running sum)")
    assert d_pl < 2e-6
    with pytest.raises(RuntimeError, match="alias"):
        _lib.check(_lib.lib().amds_gemm_lnfold_planes(a.data_ptr(), K, w.data_ptr(), K, M, N, K, hi.data_ptr(), hi.data_ptr(), N, None, None, rowpart.data_ptr(), None), "planes")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["bias", "gelu", "swiglu"])
def test_gemm_lnfold_consumer(gpu, dt, epi):
    """amds_gemm_lnfold, consumer form, against Linear(LayerNorm(x)) in fp64: the folded GEMM reads the un-normalised 16-bit rows and
    W * gamma; it must be as close to the exact result as the LayerNorm-then-GEMM chain it replaces (both round to 16 bits once)."""
    g = torch.Generator().manual_seed(8)
    M, D, N = 777, 512, 1024
    x = (torch.randn(M, D, generator=g) * 3.0 + 0.4).to(gpu)
    x[:, 7] *= 40.0                                                      # one massive channel, as real ViT residual streams have
    gamma = (1.0 + 0.3 * torch.randn(D, generator=g)).to(gpu)
    beta = (0.2 * torch.randn(D, generator=g)).to(gpu)
    W = (torch.randn(N, D, generator=g) / D ** 0.5).to(gpu)
    b = (0.1 * torch.randn(N, generator=g)).to(gpu)
    y64 = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6) @ W.double().t() + b.double()
    if epi == "swiglu":
        H = N // 2
        want = torch.nn.functional.silu(y64[:, :H]) * y64[:, H:]
        Wp, bp = ops.pack_swiglu_rows(W), ops.pack_swiglu_rows(b.view(-1, 1)).view(-1)
    else:
        want = torch.nn.functional.gelu(y64) if epi == "gelu" else y64
        Wp, bp = W, b
    code = {"bias": _lib.EPI_BIAS, "gelu": _lib.EPI_BIAS_GELU, "swiglu": _lib.EPI_SWIGLU}[epi]
    # the chain it replaces
    h = ops.layernorm(x, gamma, beta, 1e-6, dt)
    chain = ops.gemm(h, Wp.to(dt), code, bias=bp, cfg=12).double()
    # folded
    Wf = (Wp * gamma[None, :]).to(dt)
    colsum = Wf.float().sum(1)
    bf = bp + Wp @ beta
    xh, rs = ops.ln_stats_cast(x, 1e-6, dt)
    assert torch.equal(xh, x.to(dt))
    got = ops.gemm_lnfold(xh, Wf, code, bias=bf, rowstat=rs, colsum=colsum).double()
    e_chain = ((chain - want).norm() / want.norm()).item()
    e_fold = ((got - want).norm() / want.norm()).item()
    tol = 6e-3 if dt == torch.bfloat16 else 8e-4
    assert e_fold < tol and e_fold < 1.5 * e_chain + 1e-4, (e_fold, e_chain)
