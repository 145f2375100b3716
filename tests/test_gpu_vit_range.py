"""The tile encoder on weights with the statistics real DINOv2-family checkpoints have (massive-activation channels, LayerScale gammas over
five decades), and the guard that keeps non-finite features out of the feature file (the reference writes
`model(tiles)[:, 0].half()` straight into the .h5, src/stamp/preprocessing/__init__.py:324-345)."""
import warnings

import pytest
import torch

from oracle.vit_tile_encoder import extract_features
from stamp_amd.vit import PRESETS, FeatureRangeError, HipViT, random_vit_state_dict

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def _rel_small(a, b):
    """relative L2 error over the entries that are NOT massive (|ref| < 10 x the median |ref|): a row's few giant channels otherwise carry the whole
    norm and hide what happens to the channels that hold the tile's content"""
    m = b.abs() < 10 * b.abs().median()
    return ((a.double() - b.double())[m].norm() / b.double()[m].norm()).item()


def _tiles(n, seed):
    return torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("name", ["test_tiny", "test_tiny_fold", "vit_large_patch14_224", "uni2_h"])
def test_massive_activation_checkpoint_on_the_default_path(gpu, name):
    """Residual channels at 1e2 ... 1e4 x the median (class token from the embedding on, every token from 2/3 depth on) and LayerScale gammas
    log-uniform over [1e-5, 1]: all representable in fp16, so the DEFAULT path (LayerNorm folded, hi | lo planes, class-row tail) must hold the
    stated 1e-3 on the stored features without falling back, and the range counters must stay 0."""
    cfg = PRESETS[name]
    sd = random_vit_state_dict(cfg, seed=11, init="massive")
    tiles = _tiles(3 if name.startswith("test_tiny") else 2, 12)
    ref_f, ref_t = extract_features(tiles, sd, cfg, return_tokens=True)
    assert bool(torch.isfinite(ref_f.float()).all())
    model = HipViT(cfg, sd, device=gpu, chunk=2, check="raise")
    f = model(tiles.to(gpu))
    f2, t = model(tiles.to(gpu), return_tokens=True)
    r_f, r_f2, r_t = _rel(f.cpu().float(), ref_f.float()), _rel(f2.cpu().float(), ref_f.float()), _rel(t.cpu(), ref_t)
    s_f, s_t = _rel_small(f.cpu().float(), ref_f.float()), _rel_small(t.cpu(), ref_t)
    d = model.range_diagnostics()
    print(f"{name} massive init: rel-L2 stored features {r_f:.3e} (token mode {r_f2:.3e}), all tokens {r_t:.3e}; over the non-massive entries only: "
          f"features {s_f:.3e}, tokens {s_t:.3e}; counters {d}")
    assert model.safe_level == 0 and d["rows_beyond_fp16_range_possible"] == 0
    tol = 2e-3 if name.startswith("test_tiny") else 1e-3
    assert r_f < tol and r_f2 < tol and r_t < tol, (r_f, r_f2, r_t)
    assert s_f < 2 * tol and s_t < 2 * tol, (s_f, s_t)


@pytest.mark.parametrize("name", ["test_tiny", "test_tiny_fold", "vit_large_patch14_224"])
def test_overflowing_checkpoint_falls_back_or_raises_never_nan(gpu, name):
    """One residual channel of every token beyond fp16's maximum (1e5 at 2/3 depth): the fast path cannot hold this stream.  check="raise" ->
    FeatureRangeError; the default check="fallback" re-packs with fp32 residual rows (one warning), returns finite features within the bar,
    and stays on that level for the following calls; check="off" shows what the guard is there for."""
    cfg = PRESETS[name]
    sd = random_vit_state_dict(cfg, seed=13, init="overflow")
    tiles = _tiles(2, 14)
    ref_f = extract_features(tiles, sd, cfg).float()
    assert bool(torch.isfinite(ref_f).all())
    raw = HipViT(cfg, sd, device=gpu, chunk=2, check="off")
    f_raw = raw(tiles.to(gpu)).float().cpu()
    if not raw.ln_fold:        # (widths the folded form does not take: fp32 residual rows + LayerNorm kernels from the start -- nothing to overflow)
        assert bool(torch.isfinite(f_raw).all()) and _rel(f_raw, ref_f) < 2e-3
        return
    assert not bool(torch.isfinite(f_raw).all())
    assert raw.range_diagnostics()["rows_beyond_fp16_range_possible"] > 0
    del raw
    with pytest.raises(FeatureRangeError):
        HipViT(cfg, sd, device=gpu, chunk=2, check="raise")(tiles.to(gpu))
    model = HipViT(cfg, sd, device=gpu, chunk=2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        f = model(tiles.to(gpu))
    assert model.safe_level == 1 and any("safe level 1" in str(x.message) for x in w)
    assert bool(torch.isfinite(f.float()).all())
    r = _rel(f.cpu().float(), ref_f)
    rs = _rel_small(f.cpu().float(), ref_f)
    print(f"{name} overflow init: fallback level {model.safe_level}, rel-L2 stored features {r:.3e} (non-massive entries {rs:.3e})")
    assert r < (2e-3 if name.startswith("test_tiny") else 1e-3) and rs < (4e-3 if name.startswith("test_tiny") else 2e-3), (r, rs)
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        assert torch.equal(model(tiles.to(gpu)), f) and not w2            # sticky: no second warning, same bits


def test_check_finite_entry(gpu):
    from stamp_amd import _lib
    import ctypes as C
    cnt = torch.zeros(1, dtype=torch.int32, device=gpu)
    host = (C.c_int * 1)()
    for dt, code in ((torch.float16, 0), (torch.bfloat16, 1), (torch.float32, 2)):
        x = torch.randn(100_003, device=gpu).to(dt)
        s = torch.cuda.current_stream().cuda_stream
        assert _lib.lib().amds_check_finite(x.data_ptr(), x.numel(), code, cnt.data_ptr(), host, s) == 0 and host[0] == 0
        x[5], x[77_777], x[100_002] = float("nan"), float("inf"), float("-inf")
        assert _lib.lib().amds_check_finite(x.data_ptr(), x.numel(), code, cnt.data_ptr(), host, s) == _lib.ERR_RANGE and host[0] == 3
        assert b"not finite" in _lib.lib().amds_last_error()
    assert _lib.lib().amds_check_finite(None, 0, 0, cnt.data_ptr(), host, torch.cuda.current_stream().cuda_stream) == 0
