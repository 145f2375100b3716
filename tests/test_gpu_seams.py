"""The Extractor / Encoder seams on the GPU: golden parity for gated-attention pooling (fixtures captured from the
reference's CHIEFModel), the Extractor object contract, empty / ragged inputs."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


@pytest.mark.parametrize("tag", ["small", "xs"])
def test_gated_attention_matches_reference_golden(gpu, tag):
    from stamp_amd.encoder import HipGatedAttentionEncoder

    z = np.load(G / f"chief_gated_attention_{tag}.npz")
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    enc = HipGatedAttentionEncoder(sd, device=gpu)
    x = torch.from_numpy(z["x"])
    emb = enc._generate_slide_embedding(x, gpu)
    assert emb.shape == (x.shape[1],) and emb.dtype == np.float32
    np.testing.assert_allclose(emb, z["wsi_feature"].reshape(-1), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(enc.attention_raw(x).cpu().numpy(), z["attention_raw"].reshape(-1), rtol=2e-5, atol=2e-5)
    # patient level = concatenation of slides (chief.py:129-135)
    emb2 = enc._generate_patient_embedding([x[:100], x[100:]], gpu)
    np.testing.assert_allclose(emb2, emb, rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        enc._generate_slide_embedding(torch.zeros(0, x.shape[1]), gpu)


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 1024, 5000])
def test_gated_attention_sizes_vs_oracle(gpu, N):
    from oracle.gated_attention import KEYS, gated_attention_pool
    from stamp_amd import ops

    g = torch.Generator().manual_seed(N)
    F, L, D = 768, 512, 256
    sd = {KEYS["fc_w"]: torch.randn(L, F, generator=g) / F ** 0.5, KEYS["fc_b"]: torch.randn(L, generator=g) * 0.1,
          KEYS["a_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["a_b"]: torch.randn(D, generator=g) * 0.1,
          KEYS["b_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["b_b"]: torch.randn(D, generator=g) * 0.1,
          KEYS["c_w"]: torch.randn(1, D, generator=g) * 2, KEYS["c_b"]: torch.randn(1, generator=g)}
    x = torch.randn(N, F, generator=g)
    ref = gated_attention_pool(x, sd)
    w = {k: sd[v].to(gpu).contiguous() for k, v in KEYS.items()}
    out, araw = ops.gated_attn_pool(x.to(gpu), w, return_attn=True)
    np.testing.assert_allclose(araw.cpu().numpy(), ref["attention_raw"].reshape(-1).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), ref["WSI_feature"].reshape(-1).numpy(), rtol=1e-4, atol=1e-5)
    out2 = ops.gated_attn_pool(x.to(gpu), w)
    assert torch.equal(out, out2)          # deterministic (no atomics)


def _gap_sd(F, L, D, seed):
    from oracle.gated_attention import KEYS

    g = torch.Generator().manual_seed(seed)
    return {KEYS["fc_w"]: torch.randn(L, F, generator=g) / F ** 0.5, KEYS["fc_b"]: torch.randn(L, generator=g) * 0.1,
            KEYS["a_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["a_b"]: torch.randn(D, generator=g) * 0.1,
            KEYS["b_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["b_b"]: torch.randn(D, generator=g) * 0.1,
            KEYS["c_w"]: torch.randn(1, D, generator=g) * 2, KEYS["c_b"]: torch.randn(1, generator=g)}, g


@pytest.mark.parametrize("mode", ["slab", "split", "auto"])
@pytest.mark.parametrize("dims", [(768, 512, 256), (384, 256, 256)])
def test_gated_attention_batched_ragged_vs_oracle(gpu, dims, mode):
    """Many bags in ONE launch (amds_gated_attn_pool_batched): ragged lengths around the 16- / 64-row unit edges, against the oracle per bag; with a
    forced decomposition bit-equal to the bag pooled alone (a bag's result must not depend on its neighbours); bit-stable run to run (the merge of a
    bag's partials is in unit order whichever workgroup arrives last).  "slab" / "auto" also take a slide-sized bag."""
    from oracle.gated_attention import KEYS, gated_attention_pool
    from stamp_amd import ops

    F, L, D = dims
    sd, g = _gap_sd(F, L, D, seed=F)
    lens = [1, 64, 2, 63, 65, 128, 300, 1, 1024, 129, 17, 7, 191, 640, 15, 16] + ([5000] if mode != "split" else [])
    xs = [torch.randn(n, F, generator=g) for n in lens]
    w = {k: sd[v].to(gpu).contiguous() for k, v in KEYS.items()}
    xcat = torch.cat(xs).to(gpu)
    out, araw = ops.gated_attn_pool_batched(xcat, lens, w, return_attn=True, mode=mode)
    assert out.shape == (len(lens), F) and araw.shape == (sum(lens),)
    o = 0
    for i, (n, x) in enumerate(zip(lens, xs)):
        ref = gated_attention_pool(x, sd)
        np.testing.assert_allclose(araw[o:o + n].cpu().numpy(), ref["attention_raw"].reshape(-1).numpy(), rtol=1e-4, atol=1e-4, err_msg=f"bag {i} (N={n})")
        # 3e-5: the oracle's own fp32 error on a 5000-row softmax (3.4e-5 against fp64, tools/scratch notes in DESIGN.md section 4.15)
        np.testing.assert_allclose(out[i].cpu().numpy(), ref["WSI_feature"].reshape(-1).numpy(), rtol=1e-4, atol=3e-5, err_msg=f"bag {i} (N={n})")
        if mode != "auto":
            one, a1 = ops.gated_attn_pool(xcat[o:o + n], w, return_attn=True, mode=mode)
            assert torch.equal(one, out[i]) and torch.equal(a1, araw[o:o + n]), f"bag {i} (N={n}) differs from the bag pooled alone"
        o += n
    for _ in range(3):
        assert torch.equal(ops.gated_attn_pool_batched(xcat, lens, w, mode=mode), out)
    # a [B, N, F] batch is the special case of equal lengths
    xb = torch.randn(5, 200, F, generator=g).to(gpu)
    ob = ops.gated_attn_pool_batched(xb, [200] * 5, w, mode=mode)
    for i in range(5):
        one = ops.gated_attn_pool(xb[i], w, mode="split" if mode == "auto" else mode)        # 1000 rows in total: "auto" is the split form
        assert torch.equal(ob[i], one)
    with pytest.raises(ValueError):
        ops.gated_attn_pool_batched(xcat[:10], [10, 0], w)
    if mode == "split":
        with pytest.raises(RuntimeError, match="split form"):
            ops.gated_attn_pool_batched(torch.zeros(13000, F, device=gpu), [13000], w, mode="split")


def test_gated_attention_fused_vs_six_launch_and_fallback_shape(gpu):
    """The fused launch against the six-launch form it replaces (same exact-fp32 products, different association), and a shape the fused kernel does
    not take (L = 320) still served -- by the six-launch form -- and still within the oracle's bars."""
    from oracle.gated_attention import KEYS, gated_attention_pool
    from stamp_amd import ops

    for (F, L, D), N in (((768, 512, 256), 1500), ((384, 256, 256), 90)):
        sd, g = _gap_sd(F, L, D, seed=N)
        x = torch.randn(N, F, generator=g).to(gpu)
        w = {k: sd[v].to(gpu).contiguous() for k, v in KEYS.items()}
        ou, au = ops.gated_attn_pool(x, w, return_attn=True, fused=False)
        for mode in ("slab", "split"):
            of, af = ops.gated_attn_pool(x, w, return_attn=True, mode=mode)
            assert ((of - ou).norm() / ou.norm()).item() < 5e-6 and (af - au).abs().max().item() < 2e-5, mode      # fp32 round-off class: same products, other association
    F, L, D, N = 200, 320, 72, 333
    assert not ops.gated_attn_pool_batched_supported(F, L, D)
    sd, g = _gap_sd(F, L, D, seed=5)
    x = torch.randn(N, F, generator=g)
    w = {k: sd[v].to(gpu).contiguous() for k, v in KEYS.items()}
    ref = gated_attention_pool(x, sd)
    out, araw = ops.gated_attn_pool(x.to(gpu), w, return_attn=True)
    np.testing.assert_allclose(araw.cpu().numpy(), ref["attention_raw"].reshape(-1).numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), ref["WSI_feature"].reshape(-1).numpy(), rtol=1e-4, atol=1e-5)


def test_extractor_seam(gpu):
    from oracle.vit_tile_encoder import extract_features
    from stamp_amd.extractor import Extractor, extract_tiles, hip_vit_extractor, u8_tile_transform
    from stamp_amd.vit import PRESETS, random_vit_state_dict

    cfg = PRESETS["test_tiny"]
    sd = random_vit_state_dict(cfg, seed=9)
    ex = hip_vit_extractor("test_tiny", sd, device=gpu, chunk=3)
    assert isinstance(ex, Extractor) and ex.identifier == "amdstamp-test_tiny"
    with pytest.raises(Exception):
        ex.identifier = "x"                 # frozen, like the reference dataclass
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (224, 224, 3), dtype=np.uint8) for _ in range(7)]
    tiles = torch.stack([ex.transform(im) for im in imgs])          # what the reference's DataLoader collates
    assert tiles.dtype == torch.uint8 and tiles.shape == (7, 224, 224, 3)
    model = ex.model.to(gpu).eval()                                   # preprocessing/__init__.py:243
    with torch.inference_mode():
        feats = model(tiles.to(gpu)).detach().half().cpu()            # :324-325
    ref = extract_features(tiles, sd, cfg)
    assert feats.dtype == torch.float16 and ((feats.float() - ref.float()).norm() / ref.float().norm()).item() < 2e-3
    assert torch.equal(extract_tiles(ex, tiles, batch_size=4, device=gpu), feats)
    assert extract_tiles(ex, tiles[:0], device=gpu).shape == (0, cfg.dim)      # slide without tiles
    with pytest.raises(ValueError):
        u8_tile_transform(np.zeros((224, 224), dtype=np.uint8))
    with pytest.raises(ValueError):
        model(torch.zeros(2, 100, 100, 3, dtype=torch.uint8, device=gpu))      # wrong tile size: an exception, not abort()


def test_ctranspath_extractor_seam(gpu):
    """`hip_ctranspath_extractor` = the reference's ctranspath()/chief_ctranspath() factories with the HIP model."""
    from oracle.swin_ctranspath import swin_encode_f16
    from stamp_amd.extractor import Extractor, extract_tiles, hip_ctranspath_extractor
    from stamp_amd.swin import SWIN_PRESETS, random_swin_state_dict

    cfg = SWIN_PRESETS["test_swin_tiny"]
    sd = random_swin_state_dict(cfg, seed=21)
    ex = hip_ctranspath_extractor(sd, identifier="chief-ctranspath", cfg=cfg, device=gpu, chunk=4)
    assert isinstance(ex, Extractor) and ex.identifier == "chief-ctranspath"
    rng = np.random.default_rng(1)
    tiles = torch.stack([ex.transform(rng.integers(0, 256, (cfg.img, cfg.img, 3), dtype=np.uint8)) for _ in range(9)])
    model = ex.model.to(gpu).eval()
    with torch.inference_mode():
        feats = model(tiles.to(gpu)).detach().half().cpu()
    ref = swin_encode_f16(tiles, sd, cfg)
    assert feats.shape == (9, cfg.out_dim) and ((feats.float() - ref.float()).norm() / ref.float().norm()).item() < 1.5e-3
    assert torch.equal(extract_tiles(ex, tiles, batch_size=5, device=gpu), feats)
    assert extract_tiles(ex, tiles[:0], device=gpu).shape == (0, cfg.out_dim)


def test_chief_pipeline_tiles_to_slide_embedding(gpu):
    """The reference's `chief-ctranspath` chain end to end on the HIP path: decoded tiles -> background filter ->
    CTransPath features (fp16, what the .h5 holds) -> CHIEF gated-attention pooling -> 768-d slide embedding, against the
    same chain through the oracle (both halves pinned to the reference: ctranspath.py, chief.py)."""
    from oracle.gated_attention import KEYS, gated_attention_pool
    from oracle.swin_ctranspath import swin_encode_f16
    from oracle import texture
    from stamp_amd.encoder import HipGatedAttentionEncoder
    from stamp_amd.extractor import extract_tiles, has_enough_texture, hip_ctranspath_extractor
    from stamp_amd.swin import SWIN_PRESETS, random_swin_state_dict

    cfg = SWIN_PRESETS["ctranspath"]
    sd = random_swin_state_dict(cfg, seed=31)
    rng = np.random.default_rng(3)
    tiles = rng.integers(0, 256, (12, 224, 224, 3), dtype=np.uint8)
    tiles[3] = 240                                           # two blank tiles: the filter must drop exactly these
    tiles[7] = 255
    tiles_t = torch.from_numpy(tiles)
    keep = has_enough_texture(tiles_t.to(gpu), cutoff=0.02).cpu()
    assert keep.tolist() == [texture.has_enough_texture(t, 0.02) for t in tiles] and keep.sum() == 10
    ex = hip_ctranspath_extractor(sd, identifier="chief-ctranspath", device=gpu, chunk=4)
    feats = extract_tiles(ex, tiles_t[keep], batch_size=6, device=gpu)               # fp16 [10, 768] on the host
    ref_feats = swin_encode_f16(tiles_t[keep], sd, cfg)
    assert ((feats.float() - ref_feats.float()).norm() / ref_feats.float().norm()).item() < 1.5e-3
    g = torch.Generator().manual_seed(9)
    F_, L, D = 768, 512, 256
    csd = {KEYS["fc_w"]: torch.randn(L, F_, generator=g) / F_ ** 0.5, KEYS["fc_b"]: torch.randn(L, generator=g) * 0.1,
           KEYS["a_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["a_b"]: torch.randn(D, generator=g) * 0.1,
           KEYS["b_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["b_b"]: torch.randn(D, generator=g) * 0.1,
           KEYS["c_w"]: torch.randn(1, D, generator=g), KEYS["c_b"]: torch.randn(1, generator=g)}
    enc = HipGatedAttentionEncoder(csd, device=gpu)
    assert "chief-ctranspath" in enc.required_extractors and ex.identifier in enc.required_extractors
    emb = enc._generate_slide_embedding(feats.float(), device=gpu)                    # the reference reads the h5 as fp32 (chief.py:117)
    ref_emb = gated_attention_pool(ref_feats.float(), csd)["WSI_feature"].reshape(-1).numpy()
    assert emb.shape == (768,) and np.linalg.norm(emb - ref_emb) / np.linalg.norm(ref_emb) < 2e-3


def test_encoder_file_loop_on_h5_files(gpu, tmp_path):
    """`Encoder.encode_slides_` / `encode_patients_` of the reference (encoding/encoder/__init__.py:42-162) on real .h5 files in STAMP's
    schema: tile files in -> slide / patient files out, extractor validated after stripping the hash suffix, existing outputs skipped."""
    import numpy as np
    from oracle.gated_attention import KEYS, gated_attention_pool
    from stamp_amd import h5io
    from stamp_amd.encoder import HipGatedAttentionEncoder, resolve_extractor_name

    assert resolve_extractor_name("chief-ctranspath-0a1b2c3d") == "chief-ctranspath" and resolve_extractor_name("chief-ctranspath") == "chief-ctranspath"
    g = torch.Generator().manual_seed(0)
    F_, L, Dd = 768, 512, 256
    sd = {KEYS["fc_w"]: torch.randn(L, F_, generator=g) / F_ ** 0.5, KEYS["fc_b"]: torch.randn(L, generator=g) * 0.1, KEYS["a_w"]: torch.randn(Dd, L, generator=g) / L ** 0.5,
          KEYS["a_b"]: torch.zeros(Dd), KEYS["b_w"]: torch.randn(Dd, L, generator=g) / L ** 0.5, KEYS["b_b"]: torch.zeros(Dd),
          KEYS["c_w"]: torch.randn(1, Dd, generator=g) / Dd ** 0.5, KEYS["c_b"]: torch.zeros(1)}
    feat_dir, out_dir = tmp_path / "feats", tmp_path / "out"
    slides = {}
    for name, n, ext in (("a/s1", 300, "chief-ctranspath-deadbeef"), ("s2", 77, "chief-ctranspath"), ("s3", 10, "uni2")):
        f = (torch.randn(n, F_, generator=g) * 0.5).half()
        slides[name] = f
        h5io.write_tile_features(feat_dir / f"{name}.h5", f, torch.zeros(n, 2), extractor=ext, tile_size_um=256.0, tile_size_px=224, code_hash="x", stamp_version="2.5.0")
    enc = HipGatedAttentionEncoder(sd, device=gpu)
    enc.encode_slides_(out_dir, feat_dir, gpu, generate_hash=False)
    outs = sorted(p.relative_to(out_dir).as_posix() for p in out_dir.rglob("*.h5"))
    assert outs == ["chief-slide/a/s1.h5", "chief-slide/s2.h5"]                      # s3 was extracted with a model this encoder does not accept
    d, a = h5io.read_file(out_dir / "chief-slide" / "a" / "s1.h5")
    ref = gated_attention_pool(slides["a/s1"].float(), sd)["WSI_feature"].reshape(-1)
    assert a["feat_type"] == "slide" and a["encoder"] == "chief" and a["precision"] == "torch.float32" and d["feats"].shape == (F_,)
    assert np.abs(d["feats"] - ref.numpy()).max() < 1e-4 * max(1.0, float(ref.abs().max()))
    # the slides above went through ONE batched launch; the reference's one-slide-per-iteration loop writes the same files
    enc.encode_slides_(tmp_path / "out1", feat_dir, gpu, generate_hash=False, batch_slides=1)
    for rel in outs:
        d1, a1 = h5io.read_file(tmp_path / "out1" / rel)
        db, ab = h5io.read_file(out_dir / rel)
        assert a1["feat_type"] == ab["feat_type"] and np.abs(d1["feats"] - db["feats"]).max() < 2e-6
    before = (out_dir / "chief-slide" / "s2.h5").stat().st_mtime_ns
    enc.encode_slides_(out_dir, feat_dir, gpu, generate_hash=False)                    # second run: everything exists, nothing rewritten
    assert (out_dir / "chief-slide" / "s2.h5").stat().st_mtime_ns == before
    enc.encode_patients_(out_dir, feat_dir, {"P1": ["a/s1.h5", "s2.h5"]}, gpu, generate_hash=False)
    d, a = h5io.read_file(out_dir / "chief-pat" / "P1.h5")
    refp = gated_attention_pool(torch.cat([slides["a/s1"], slides["s2"]]).float(), sd)["WSI_feature"].reshape(-1)
    assert a["feat_type"] == "patient" and np.abs(d["feats"] - refp.numpy()).max() < 1e-4 * max(1.0, float(refp.abs().max()))
    with pytest.raises(ValueError):
        enc.encode_patients_(out_dir, feat_dir, {"P2": ["s3.h5"]}, gpu, generate_hash=False)


def test_titan_shaped_stand_in_seam(gpu, tmp_path):
    """H19.  TITAN's arithmetic is remote code (titan.py:30) -- this encoder is a labelled STAND-IN of TITAN's interface and geometry (parity
    unpinned by construction; see its docstring).  What can be pinned is pinned: the seam (768-d CONCH1.5 features + coordinates in, one 768-d
    float32 vector out, coords required, the extractor check), that the stand-in's own arithmetic is this package's MIL `vit` + ALiBi
    (compared against the pinned oracle of THAT network on the stand-in's weights), and the reference's patient-level "virtual slide"
    construction (titan.py:87-179: slides side by side along x with a running offset)."""
    from oracle.mil_vit import mil_vit_forward
    from stamp_amd import h5io
    from stamp_amd.encoder import HipTitanShapedEncoder

    enc = HipTitanShapedEncoder(seed=3, device=gpu)
    assert enc.identifier == "titan-standin" and enc.required_extractors == ["conch1_5"] and enc.precision == torch.float32
    g = torch.Generator().manual_seed(1)
    N = 900
    feats = torch.randn(N, 768, generator=g).half().float()
    grid = torch.stack([torch.randint(0, 60, (N,), generator=g), torch.randint(0, 40, (N,), generator=g)], 1).double()
    coords = h5io.CoordsInfo((grid * 256.0).numpy(), 256.0, 512)                     # CONCH1.5 tiles: 512 px at 0.5 um / px
    emb = enc._generate_slide_embedding(feats, gpu, coords=coords)
    assert emb.shape == (768,) and emb.dtype == np.float32 and np.isfinite(emb).all()
    assert np.array_equal(emb, enc._generate_slide_embedding(feats, gpu, coords=coords))                 # deterministic
    with pytest.raises(ValueError, match="Coords must be provided"):
        enc._generate_slide_embedding(feats, gpu)
    # the stand-in's arithmetic == the pinned oracle of the MIL vit + ALiBi network on the same weights and the same tile-unit coordinates
    sd = {k: v.detach().cpu() for k, v in enc.net.state_dict().items()}
    cpx = (torch.tensor(coords.coords_um, dtype=torch.float32) / coords.mpp).to(torch.int64).float() / 512.0
    ref = mil_vit_forward(feats[None], cpx[None], None, sd, n_heads=12, use_alibi=True)[0].numpy()
    assert np.linalg.norm(emb - ref) / np.linalg.norm(ref) < 5e-3
    # patient level: two slides laid side by side == one slide with the second shifted by (max x + tile width) of the first
    a, b = slice(0, 500), slice(500, N)
    ca, cb = h5io.CoordsInfo(coords.coords_um[a].copy(), 256.0, 512), h5io.CoordsInfo(coords.coords_um[b].copy(), 256.0, 512)
    for nm, sl, ci in (("s1.h5", a, ca), ("s2.h5", b, cb)):
        h5io.write_tile_features(tmp_path / nm, feats[sl].half(), ci.coords_um.astype(np.float32), extractor="conch1_5-0a1b2c3d", tile_size_um=256.0,
                                 tile_size_px=512, code_hash="0a1b2c3d", stamp_version="2.5.0")
    enc.encode_patients_(tmp_path / "out", tmp_path, {"P1": ["s1.h5", "s2.h5", "notes.txt"]}, device=gpu, generate_hash=False)
    d, at = h5io.read_file(tmp_path / "out" / "titan-standin-pat" / "P1.h5")
    shifted = coords.coords_um[b].copy()
    shifted[:, 0] += coords.coords_um[a][:, 0].max() + 256.0
    virt = h5io.CoordsInfo(np.concatenate([coords.coords_um[a], shifted]), 256.0, 512)
    want = enc._generate_slide_embedding(torch.cat([feats[a], feats[b]]), gpu, coords=virt)
    assert at["encoder"] == "titan-standin" and at["feat_type"] == "patient" and np.allclose(d["feats"], want, rtol=1e-5, atol=1e-6)
    # slides extracted with another extractor are refused (encoder/__init__.py:173-180)
    h5io.write_tile_features(tmp_path / "other.h5", feats[:10].half(), coords.coords_um[:10].astype(np.float32), extractor="uni2", tile_size_um=256.0,
                             tile_size_px=512, code_hash="0a1b2c3d", stamp_version="2.5.0")
    with pytest.raises(ValueError, match="must be extracted with one of"):
        enc._validate_and_read_features(str(tmp_path / "other.h5"))
    # different mpp within one patient is an error, as in the reference
    h5io.write_tile_features(tmp_path / "s3.h5", feats[:10].half(), coords.coords_um[:10].astype(np.float32), extractor="conch1_5", tile_size_um=256.0,
                             tile_size_px=224, code_hash="0a1b2c3d", stamp_version="2.5.0")
    with pytest.raises(ValueError, match="same mpp"):
        enc.encode_patients_(tmp_path / "out2", tmp_path, {"P2": ["s1.h5", "s3.h5"]}, device=gpu, generate_hash=False)


def test_eagle_encoder_matches_reference_fixture_and_file_loop(gpu, tmp_path):
    """The reference's EAGLE encoder (encoding/encoder/eagle.py): HIP selection + mean against the fixture made by the reference's own
    `_generate_slide_embedding`, then its file loops on real .h5 pairs -- a permuted Virchow2 file is re-ordered by coordinates, a file from the
    wrong extractor or with foreign coordinates is reported and skipped, a patient = its slides concatenated."""
    from oracle import eagle
    from stamp_amd import h5io, ops
    from stamp_amd.encoder import HipEagleEncoder

    z = np.load(Path(__file__).parent / "golden" / "eagle.npz")
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    enc = HipEagleEncoder(sd, device=gpu)
    assert enc.identifier == "eagle" and enc.required_extractors == ["ctranspath", "chief-ctranspath"] and enc.required_agg_extractor == "virchow2"
    for tag in ("a", "b"):
        x, agg = torch.from_numpy(z[f"{tag}_x"]), torch.from_numpy(z[f"{tag}_agg"])
        emb = enc._generate_slide_embedding(x, gpu, agg_feats=agg)
        assert emb.dtype == np.float32 and emb.shape == (agg.shape[1],)
        np.testing.assert_allclose(emb, z[f"{tag}_emb"], rtol=1e-5, atol=1e-6)
        assert np.array_equal(enc.top_tiles(x).cpu().numpy(), z[f"{tag}_top"])                      # the same 25 (or N) tiles, in the same order
        emb16 = enc._generate_slide_embedding(x, gpu, agg_feats=agg.half())                         # fp16 aggregation features as stored on disk
        np.testing.assert_allclose(emb16, z[f"{tag}_emb"], rtol=1e-5, atol=1e-6)                    # (the fixture's values are fp16-exact)
    with pytest.raises(ValueError, match="agg_feats is required"):
        enc._generate_slide_embedding(torch.from_numpy(z["a_x"]), gpu)
    # ties: the lower index wins, k <= 32 enforced
    sc = torch.tensor([1.0, 3.0, 3.0, 2.0, 3.0], device=gpu)
    rows = torch.arange(5, dtype=torch.float32, device=gpu).unsqueeze(1).repeat(1, 4).contiguous()
    idx, mean = ops.topk_rows_mean(sc, rows, 3)
    assert idx.tolist() == [1, 2, 4] and torch.allclose(mean, torch.full((4,), 7.0 / 3.0, device=gpu))
    with pytest.raises(RuntimeError, match="1 <= k"):
        ops.topk_rows_mean(sc, rows, 6)
    # ---- file loops
    g = torch.Generator().manual_seed(5)
    ctp_dir, vir_dir, out_dir = tmp_path / "ctp", tmp_path / "vir", tmp_path / "out"
    Fd = z["a_x"].shape[1]
    slides = {}
    for name, n in (("s1", 120), ("s2", 40), ("s3", 30), ("s4", 20)):
        f = (torch.randn(n, Fd, generator=g) * 0.7).half()
        a = torch.randn(n, 96, generator=g).half()
        c = (torch.stack([torch.arange(n) % 13, torch.arange(n) // 13], 1) * 256.0).float()
        slides[name] = (f, a, c)
        h5io.write_tile_features(ctp_dir / f"{name}.h5", f, c, extractor="chief-ctranspath-0a1b2c3d" if name != "s3" else "uni2", tile_size_um=256.0, tile_size_px=224,
                                 code_hash="x", stamp_version="2.5.0")
        perm = torch.randperm(n, generator=g) if name == "s2" else torch.arange(n)                   # s2: the Virchow2 file lists the tiles in another order
        ca = c[perm] if name != "s4" else c[perm] + 1000.0                                             # s4: coordinates of some other slide
        h5io.write_tile_features(vir_dir / f"{name}.h5", a[perm], ca, extractor="virchow2", tile_size_um=256.0, tile_size_px=224, code_hash="x", stamp_version="2.5.0")
    with pytest.raises(ValueError, match="agg_feat_dir"):
        enc.encode_slides_(out_dir, ctp_dir, gpu, generate_hash=False)
    enc.encode_slides_(out_dir, ctp_dir, gpu, generate_hash=False, agg_feat_dir=vir_dir)
    assert sorted(p.name for p in (out_dir / "eagle-slide").glob("*.h5")) == ["s1.h5", "s2.h5"]     # s3: wrong extractor; s4: coordinates do not match
    for name in ("s1", "s2"):
        f, a, _ = slides[name]
        d, at = h5io.read_file(out_dir / "eagle-slide" / f"{name}.h5")
        ref, _ = eagle.eagle_slide_embedding(f.float(), a.float(), sd)
        assert at["feat_type"] == "slide" and at["encoder"] == "eagle" and d["feats"].shape == (96,)
        np.testing.assert_allclose(d["feats"], ref, rtol=1e-5, atol=1e-6)
    enc.encode_patients_(out_dir, ctp_dir, {"P1": ["s1.h5", "s2.h5", "missing.h5"], "P2": ["missing.h5"]}, gpu, generate_hash=False, agg_feat_dir=vir_dir)
    assert sorted(p.name for p in (out_dir / "eagle-pat").glob("*.h5")) == ["P1.h5"]
    d, at = h5io.read_file(out_dir / "eagle-pat" / "P1.h5")
    refp = eagle.eagle_patient_embedding([slides["s1"][0].float(), slides["s2"][0].float()], [slides["s1"][1].float(), slides["s2"][1].float()], sd)
    assert at["feat_type"] == "patient"
    np.testing.assert_allclose(d["feats"], refp, rtol=1e-5, atol=1e-6)


def test_ticon_tile_stage_and_extractor(gpu):
    """TICON as the reference's extractor runs it (ticon.py:691-718: every tile alone): the HIP stage (one C call, exact fp32) against the fixture made
    by the reference's own `EncoderDecoder`, both input projections; then the two-stage extractor (H-optimus trunk + TICON) at test size against
    the oracles of both stages."""
    from oracle import ticon as ot
    from oracle.vit_tile_encoder import extract_features
    from stamp_amd.extractor import Extractor
    from stamp_amd.ticon import HipHOptimusTicon, HipTiconTile
    from stamp_amd.vit import PRESETS, random_vit_state_dict

    z = np.load(Path(__file__).parent / "golden" / "ticon.npz")
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    for key in ("hoptimus1", "conchv15"):
        m = HipTiconTile(sd, key=key, device=gpu)
        emb = torch.from_numpy(z[f"emb_{key}"]).to(gpu)
        y = m(emb)
        assert y.dtype == torch.float32 and y.shape == z[f"out_{key}"].shape
        np.testing.assert_allclose(y.cpu().numpy(), z[f"out_{key}"], rtol=2e-5, atol=2e-5)
        assert torch.equal(y, m(emb.half()))                                 # fp16 embeddings (the fixture's are fp16-exact)
        assert m(emb[:0]).shape == (0, y.shape[1])
    with pytest.raises(KeyError, match="lacks"):
        HipTiconTile(sd, key="uni2h", device=gpu)                            # the fixture has no such input projection
    with pytest.raises(ValueError, match="tile embeddings"):
        m(torch.zeros(3, 7, device=gpu))
    # two stages: a test-size SwiGLU trunk with register tokens (H-optimus's structure) whose width is the fixture's hoptimus1 input width (128)
    cfg = PRESETS["test_tiny_swiglu"]
    vsd = random_vit_state_dict(cfg, seed=5)
    model = HipHOptimusTicon(vsd, sd, device=gpu, chunk=4, vit_cfg=cfg)
    tiles = torch.randint(0, 256, (6, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(6))
    out = model(tiles.to(gpu))
    assert out.dtype == torch.float16 and out.shape == (6, 96)                 # TICON's own width
    ref = ot.ticon_tile_forward(extract_features(tiles, vsd, cfg).float(), sd, "hoptimus1")          # fp16 trunk features, as the HIP trunk hands them over
    rel = ((out.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 2e-3, rel
    ex = Extractor(model=model, transform=lambda im: im, identifier="ticon")
    assert ex.identifier == "ticon" and ex.model is model


def test_keep_extractor_head_and_trunk(gpu):
    """KEEP (keep.py:25-50): ViT-L/16 trunk + `visual_head` + L2 normalisation.  The head (one exact-fp32 library call) against the fixture made by the
    reference's own class; the two stages together at test size against the oracles of both; the `.ls1.weight` key spelling of the checkpoint."""
    from dataclasses import replace

    from oracle import misc
    from oracle.vit_tile_encoder import extract_features
    from stamp_amd import _lib, ops
    from stamp_amd.extractor import HipKeep
    from stamp_amd.vit import PRESETS, random_vit_state_dict
    import ctypes as C
    z = np.load(Path(__file__).parent / "golden" / "keep_head.npz")
    hsd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    feats = torch.from_numpy(z["feats"]).to(gpu)
    lib = _lib.lib()
    w = [hsd[f"visual_head.{k}"].to(gpu).contiguous() for k in ("0.weight", "0.bias", "2.weight", "2.bias")]
    for x in (feats, feats.half()):
        out = torch.empty(11, 96, device=gpu)
        nb = lib.amds_proj_head_l2norm_workspace_bytes(11, 128, 96)
        ws = torch.empty(nb, dtype=torch.uint8, device=gpu)
        _lib.check(lib.amds_proj_head_l2norm(x.data_ptr(), ops._DT[x.dtype], *[t.data_ptr() for t in w], out.data_ptr(), 11, 128, 96, ws.data_ptr(), nb, None), "head")
        np.testing.assert_allclose(out.cpu().numpy(), z["out"], rtol=2e-5, atol=2e-6)
    cfg = replace(PRESETS["vit_large_patch16_224"], dim=128, depth=2, heads=2, hidden=256)
    vsd = random_vit_state_dict(cfg, seed=12)
    sd = {f"visual.{k.replace('.gamma', '.weight') if '.ls' in k else k}": v for k, v in vsd.items()}          # the checkpoint's LayerScale spelling (keep.py:53-59)
    sd.update(hsd)
    sd["logit_scale"] = torch.zeros(())                                                                       # entries outside visual.* are ignored (:83-88)
    model = HipKeep(sd, device=gpu, chunk=3, vit_cfg=cfg)
    tiles = torch.randint(0, 256, (5, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(13))
    out = model(tiles.to(gpu))
    assert out.dtype == torch.float32 and out.shape == (5, 96)
    ref = misc.keep_image_head(extract_features(tiles, vsd, cfg).float(), hsd)
    assert ((out.cpu() - ref).norm() / ref.norm()).item() < 1e-3
    np.testing.assert_allclose(out.norm(dim=1).cpu().numpy(), 1.0, rtol=1e-5)
    with pytest.raises(KeyError, match="visual_head"):
        HipKeep({k: v for k, v in sd.items() if not k.startswith("visual_head.2")}, device=gpu, vit_cfg=cfg)


def test_plip_extractor_matches_transformers_fixture(gpu):
    """PLIP (plip.py:16-36): CLIP's vision tower (32-pixel patches, pre-LayerNorm, quick_gelu MLP, no LayerScale) on the HIP tile encoder + the visual
    projection, against image features produced by the installed `transformers` CLIPModel itself (tests/golden/plip.npz).  Bar: the fp16-operand
    path's 1e-3 relative L2; then the full-size preset (ViT-B/32) against the oracle pinned to that fixture."""
    from dataclasses import replace

    from oracle import clip_vision as cv
    from stamp_amd.extractor import HipPlip, clip_vision_to_timm_names
    from stamp_amd.vit import PRESETS, HipViT
    z = np.load(Path(__file__).parent / "golden" / "plip.npz")
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    cfg = replace(PRESETS["plip"], dim=128, depth=2, heads=2, hidden=256)
    model = HipPlip(sd, device=gpu, chunk=2, vit_cfg=cfg)
    tiles = torch.from_numpy(z["tiles"])
    out = model(tiles.to(gpu))
    ref = torch.from_numpy(z["image_features"])
    rel = ((out.cpu() - ref).norm() / ref.norm()).item()
    assert out.dtype == torch.float32 and out.shape == ref.shape and rel < 1e-3, rel
    assert torch.equal(out, model(tiles.to(gpu)))
    with pytest.raises(ValueError, match="plain packing"):
        HipViT(cfg, clip_vision_to_timm_names(sd)[0], device=gpu, exact=True)
    # full size: CLIP ViT-B/32 (768 wide, 12 layers), random weights, 3 tiles
    g = torch.Generator().manual_seed(3)
    big = {}
    D, Hd, P = 768, 3072, 512
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc  # noqa: E731
    big["vision_model.embeddings.class_embedding"] = rn(D, sc=0.5)
    big["vision_model.embeddings.patch_embedding.weight"] = rn(D, 3, 32, 32, sc=(3 * 32 * 32) ** -0.5)
    big["vision_model.embeddings.position_embedding.weight"] = rn(50, D, sc=0.5)
    for n in ("pre_layrnorm", "post_layernorm"):
        big[f"vision_model.{n}.weight"], big[f"vision_model.{n}.bias"] = 1.0 + rn(D, sc=0.1), rn(D, sc=0.1)
    for l in range(12):
        q = f"vision_model.encoder.layers.{l}."
        for n in ("layer_norm1", "layer_norm2"):
            big[q + n + ".weight"], big[q + n + ".bias"] = 1.0 + rn(D, sc=0.1), rn(D, sc=0.1)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            big[q + f"self_attn.{n}.weight"], big[q + f"self_attn.{n}.bias"] = rn(D, D, sc=D ** -0.5 * (0.4 if n == "out_proj" else 1.0)), rn(D, sc=0.1)
        big[q + "mlp.fc1.weight"], big[q + "mlp.fc1.bias"] = rn(Hd, D, sc=D ** -0.5), rn(Hd, sc=0.1)
        big[q + "mlp.fc2.weight"], big[q + "mlp.fc2.bias"] = rn(D, Hd, sc=Hd ** -0.5 * 0.4), rn(D, sc=0.1)
    big["visual_projection.weight"] = rn(P, D, sc=D ** -0.5)
    t3 = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=g)
    refb = cv.clip_image_features(cv.tiles_to_pixels(t3), big, heads=12)
    outb = HipPlip(big, device=gpu, chunk=3)(t3.to(gpu)).cpu()
    relb = ((outb - refb).norm() / refb.norm()).item()
    print(f"PLIP (CLIP ViT-B/32) full size: image features vs the fp32 oracle {relb:.3e}")
    assert outb.shape == (3, 512) and relb < 1e-3, relb
