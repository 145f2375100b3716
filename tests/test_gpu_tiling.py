"""Supertile -> tiles on the GPU (amds_supertiles_to_tiles_u8) against the reference's own tiles (tests/golden/tiling_*.npz, produced by
the reference's `_supertiles` / `_tiles` on a synthetic slide) and against Pillow-pinned oracle resizes: BIT-EXACT."""
import zlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import tiling as ot
from stamp_amd import tiling as pt

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _read_region(slide_rgb, x, y, s):
    out = np.zeros((s, s, 4), dtype=np.uint8)
    sub = slide_rgb[y:y + s, x:x + s]
    out[:sub.shape[0], :sub.shape[1], :3] = sub
    out[:sub.shape[0], :sub.shape[1], 3] = 255
    return out


@pytest.mark.parametrize("tag", ["mpp050", "mpp025"])
def test_gpu_tiles_equal_the_references(gpu, tag):
    z = np.load(G / f"tiling_{tag}.npz")
    w, h, seed = (int(v) for v in z["slide"])
    mpp = float(z["mpp"])
    slide = ot.synthetic_slide(w, h, seed)
    geo = pt.supertile_geometry(mpp, 256.0, 224, 1024)
    fg = [tuple(int(v) for v in r) for r in z["foreground"]]
    regions = np.stack([_read_region(slide, x, y, geo.supertile_size_slide_px) for x, y in fg])           # what openslide.read_region hands over
    tiles = pt.supertiles_to_tiles(torch.from_numpy(regions).to(gpu), geo.tiles_per_side, 224).cpu().numpy()
    coords = np.concatenate([pt.tile_coords_um(o, mpp, geo.tiles_per_side, 256.0) for o in fg])
    order = np.lexsort((coords[:, 0], coords[:, 1]))
    assert np.array_equal(coords[order], z["coords_um"])
    crc = np.array([zlib.crc32(tiles[i].tobytes()) for i in order], dtype=np.uint32)
    assert np.array_equal(crc, z["crc32"])                                  # every tile, every byte, equals the reference's PIL pipeline
    for i, ref in zip(z["full_idx"], z["full_tiles"]):
        assert np.array_equal(tiles[order[int(i)]], ref)


@pytest.mark.parametrize("S,k", [(1024, 1), (1024, 2), (512, 1), (896, 4), (100, 1)])
def test_gpu_resize_random_rgba_vs_oracle(gpu, S, k):
    """Random pixels incl. partial transparency (premultiply / un-premultiply paths), down- and up-scaling."""
    rng = np.random.default_rng(S + k)
    a = rng.integers(0, 256, (3, S, S, 4), dtype=np.uint8)
    a[0, :, :, 3] = 255
    a[1, : S // 2, :, 3] = 255
    a[1, S // 2:, :, :] = 0
    t = 224 if S >= 512 else 64
    got = pt.supertiles_to_tiles(torch.from_numpy(a).to(gpu), k, t).cpu().numpy()
    assert got.shape == (3 * k * k, t, t, 3)
    for i in range(3):
        rgb = ot.supertile_to_rgb(a[i], k * t)
        ref = np.stack([tile for tile, _ in ot.tiles_of_supertile(rgb, (0.0, 0.0), 1.0, t)])
        assert np.array_equal(got[i * k * k:(i + 1) * k * k], ref), (S, k, i)
    assert pt.supertiles_to_tiles(torch.from_numpy(a[:0]).to(gpu), k, t).shape == (0, t, t, 3)
    with pytest.raises(RuntimeError, match="GPU"):
        pt.supertiles_to_tiles(torch.from_numpy(a), k, t)


def test_extract_slide_end_to_end_writes_stamp_h5(gpu, tmp_path):
    """A synthetic slide object through the whole path (thumbnail rejection -> threaded read_region -> GPU resize / crop -> GPU Canny
    filter -> tile encoder -> .h5 in STAMP's schema): tile coordinates are a subset of the reference fixture's, features equal the
    encoder applied to the reference's own tiles, attributes follow the schema."""
    from PIL import Image

    from oracle.vit_tile_encoder import extract_features
    from stamp_amd import h5io
    from stamp_amd.extractor import hip_vit_extractor
    from stamp_amd.preprocess import extract_slide
    from stamp_amd.vit import PRESETS, random_vit_state_dict

    z = np.load(G / "tiling_mpp050.npz")
    w, h, seed = (int(v) for v in z["slide"])
    rgb = ot.synthetic_slide(w, h, seed)

    class Slide:                                   # openslide's surface, as tools/make_golden.py::FakeSlide
        dimensions = (w, h)
        _im = Image.fromarray(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]), "RGBA")

        def read_region(self, loc, level, size):
            out = Image.new("RGBA", size, (0, 0, 0, 0))
            out.paste(self._im.crop((loc[0], loc[1], min(loc[0] + size[0], w), min(loc[1] + size[1], h))), (0, 0))
            return out

        def get_thumbnail(self, size):
            bg = Image.new("RGB", self._im.size, "#ffffff")
            t = Image.composite(self._im, bg, self._im)
            t.thumbnail(tuple(int(v) for v in size), Image.Resampling.LANCZOS)
            return t

    cfg = PRESETS["test_tiny"]
    sd = random_vit_state_dict(cfg, seed=3)
    ex = hip_vit_extractor("test_tiny", sd, device=gpu, chunk=8, identifier="amdstamp-test")
    out = tmp_path / "feats" / "slide.h5"
    stats = extract_slide(Slide(), ex, out, slide_mpp=0.5, brightness_cutoff=224, canny_cutoff=None, supertiles_per_batch=3, device=gpu)
    assert stats["supertiles"] == len(z["foreground"]) and stats["tiles_seen"] == stats["tiles_kept"] == len(z["coords_um"])
    feats, ci, attrs = h5io.read_tile_features(out)
    order = np.lexsort((ci.coords_um[:, 0], ci.coords_um[:, 1]))
    assert np.array_equal(ci.coords_um[order].astype(np.float64), z["coords_um"]) and feats.dtype == np.float16
    assert attrs["extractor"] == "amdstamp-test" and attrs["feat_type"] == "tile" and attrs["unit"] == "um" and attrs["tile_size_px"] == 224
    # features of the three tiles the fixture stores in full == the oracle encoder on the REFERENCE's tiles
    ref = extract_features(torch.from_numpy(z["full_tiles"]), sd, cfg).float().numpy()
    got = feats[order][z["full_idx"]].astype(np.float32)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-3
    # with the texture filter on, only a subset survives and every kept coordinate is one of the reference's
    out2 = tmp_path / "feats" / "slide_canny.h5"
    s2 = extract_slide(Slide(), ex, out2, slide_mpp=0.5, brightness_cutoff=224, canny_cutoff=0.02, device=gpu)
    _, ci2, _ = h5io.read_tile_features(out2)
    assert 0 < s2["tiles_kept"] <= s2["tiles_seen"] and {tuple(c) for c in ci2.coords_um.tolist()} <= {tuple(c) for c in z["coords_um"].astype(np.float32).tolist()}
    # the pipelined path (reader threads under the GPU work, keep-mask compacted on the device, encoder calls on accumulated chunks) writes what
    # the batch-by-batch path writes, bit for bit -- whatever the chunk / batch geometry: chunks smaller than a batch (several encoder calls per
    # flush + a remainder carried over), equal, larger than the slide
    from stamp_amd.preprocess import extract_slide_serial
    for canny in (None, 0.02):
        ref_path = tmp_path / "feats" / f"serial_{canny}.h5"
        extract_slide_serial(Slide(), ex, ref_path, slide_mpp=0.5, brightness_cutoff=224, canny_cutoff=canny, supertiles_per_batch=3, device=gpu)
        fr, cr, _ = h5io.read_tile_features(ref_path)
        for chunk, spb in ((5, 3), (12, 3), (7, 16), (10_000, 2)):
            pth = tmp_path / "feats" / f"pipe_{canny}_{chunk}_{spb}.h5"
            st = extract_slide(Slide(), ex, pth, slide_mpp=0.5, brightness_cutoff=224, canny_cutoff=canny, supertiles_per_batch=spb, encode_chunk=chunk,
                               max_workers=3, device=gpu)
            fp, cp, _ = h5io.read_tile_features(pth)
            assert np.array_equal(fp.view(np.uint16), fr.view(np.uint16)) and np.array_equal(cp.coords_um, cr.coords_um), (canny, chunk, spb)
            assert st["tiles_kept"] == fr.shape[0] and st["encoder_calls"] == -(-fr.shape[0] // chunk)

    class Broken(Slide):                               # a reader failure surfaces as an exception (the reference logs and skips the slide)
        def read_region(self, loc, level, size):
            raise OSError("cannot read region")
    with pytest.raises(OSError):
        extract_slide(Broken(), ex, tmp_path / "feats" / "broken.h5", slide_mpp=0.5, brightness_cutoff=224, device=gpu)
    assert not (tmp_path / "feats" / "broken.h5").exists()


def test_compact_rows_on_device(gpu):
    """amds_compact_rows_u8: kept rows appended in order behind a device-resident fill level; slots tell the host where each row went."""
    import ctypes as C  # noqa: F401

    from stamp_amd import _lib, ops
    torch.manual_seed(0)
    rb = 150528
    dst = torch.zeros(40, rb, dtype=torch.uint8, device=gpu)
    count = torch.zeros(1, dtype=torch.int32, device=gpu)
    expect = []
    for n, cutoff in ((13, 0.5), (1, 0.5), (17, 0.2), (9, None)):
        src = torch.randint(0, 256, (n, rb), dtype=torch.uint8, device=gpu)
        score = torch.rand(n, device=gpu)
        slots = torch.empty(n, dtype=torch.int32, device=gpu)
        _lib.check(_lib.lib().amds_compact_rows_u8(src.data_ptr(), rb, None if cutoff is None else score.data_ptr(), float(cutoff or 0), dst.data_ptr(), 40,
                                                   count.data_ptr(), slots.data_ptr(), n, ops._stream()), "compact")
        keep = torch.ones(n, dtype=torch.bool, device=gpu) if cutoff is None else score >= cutoff
        base = len(expect)
        want = torch.full((n,), -1, dtype=torch.int32)
        fit = 0
        for i in torch.nonzero(keep).flatten().tolist():
            if base + fit < 40:
                want[i] = base + fit
                expect.append(src[i].clone())
                fit += 1
            else:
                want[i] = -2
        assert torch.equal(slots.cpu(), want), (slots, want)
        assert count.item() == len(expect)
    assert torch.equal(dst[:len(expect)], torch.stack(expect))


@pytest.mark.parametrize("S,resized,crop", [(224, 256, 224), (224, 384, 384), (100, 257, 224), (256, 224, 200)])
def test_resize_center_crop_is_pillows_resize_plus_torchvisions_crop(gpu, S, resized, crop):
    """`Resize(resized, BICUBIC)` + `CenterCrop(crop)` as the reference's gigapath transform applies them to the PIL tile (gigapath.py:21-28): Pillow's
    bicubic `Image.resize` (what torchvision calls for PIL inputs) and torchvision's crop offset int(round((resized - crop) / 2.0)) -- bit for bit,
    up- and down-scaling, odd differences (Python's round half to even)."""
    from PIL import Image
    from stamp_amd.tiling import resize_center_crop
    rng = np.random.default_rng(S + resized)
    tiles = rng.integers(0, 256, (5, S, S, 3), dtype=np.uint8)
    tiles[1] = (np.indices((S, S)).sum(0) % 256)[..., None].astype(np.uint8)            # smooth ramp with wrap-arounds
    tiles[2, : S // 2] = 255
    out = resize_center_crop(torch.from_numpy(tiles).to(gpu), resized, crop).cpu().numpy()
    c0 = int(round((resized - crop) / 2.0))
    for i in range(5):
        ref = np.asarray(Image.fromarray(tiles[i], "RGB").resize((resized, resized), Image.BICUBIC))[c0:c0 + crop, c0:c0 + crop]
        assert out[i].shape == ref.shape and np.array_equal(out[i], ref), (i, np.abs(out[i].astype(int) - ref.astype(int)).max())
    assert resize_center_crop(torch.from_numpy(tiles[:0]).to(gpu), resized, crop).shape == (0, crop, crop, 3)
    with pytest.raises(ValueError, match="larger"):
        resize_center_crop(torch.from_numpy(tiles).to(gpu), 100, 224)


def test_gigapath_extractor_runs_the_transform_in_front_of_the_trunk(gpu):
    from dataclasses import replace

    from PIL import Image
    from oracle.vit_tile_encoder import extract_features
    from stamp_amd.extractor import ResizeCropThenModel
    from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict
    cfg = replace(PRESETS["gigapath"], dim=128, depth=2, heads=2, hidden=192)            # GigaPath's structure (16-pixel patches, SwiGLU, no register tokens) at test size
    sd = random_vit_state_dict(cfg, seed=8)
    tiles = torch.randint(0, 256, (4, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9))
    model = ResizeCropThenModel(HipViT(cfg, sd, device=gpu, chunk=4), 256, 224)
    out = model(tiles.to(gpu)).float().cpu()
    pil = np.stack([np.asarray(Image.fromarray(t.numpy(), "RGB").resize((256, 256), Image.BICUBIC))[16:240, 16:240] for t in tiles])
    ref = extract_features(torch.from_numpy(pil), sd, cfg).float()
    rel = ((out - ref).norm() / ref.norm()).item()
    assert out.shape == (4, 128) and rel < 1e-3, rel


def _fake_slide(w, h, seed):
    from PIL import Image
    rgb = ot.synthetic_slide(w, h, seed)

    class Slide:                                   # openslide's surface, as tools/make_golden.py::FakeSlide
        dimensions = (w, h)
        _im = Image.fromarray(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]), "RGBA")

        def read_region(self, loc, level, size):
            out = Image.new("RGBA", size, (0, 0, 0, 0))
            out.paste(self._im.crop((loc[0], loc[1], min(loc[0] + size[0], w), min(loc[1] + size[1], h))), (0, 0))
            return out

        def get_thumbnail(self, size):
            bg = Image.new("RGB", self._im.size, "#ffffff")
            t = Image.composite(self._im, bg, self._im)
            t.thumbnail(tuple(int(v) for v in size), Image.Resampling.LANCZOS)
            return t
    return Slide()


def test_extract_slide_low_resolution_slide_many_tiles_per_supertile(gpu, tmp_path):
    """mpp 4: k = floor(1024 * 4 / 256) = 16, 256 tiles per supertile -- the default 64 supertiles per batch would hand 16 384 tiles to one
    compaction call (limit 4096): the batch is clamped instead, and the file equals the serial path's."""
    from stamp_amd import h5io
    from stamp_amd.extractor import hip_vit_extractor
    from stamp_amd.preprocess import extract_slide, extract_slide_serial
    from stamp_amd.vit import PRESETS, random_vit_state_dict

    ex = hip_vit_extractor("test_tiny", random_vit_state_dict(PRESETS["test_tiny"], seed=3), device=gpu, chunk=64, identifier="amdstamp-test")
    slide = _fake_slide(2304, 2048, 5)
    a, b = tmp_path / "pipe.h5", tmp_path / "serial.h5"
    st = extract_slide(slide, ex, a, slide_mpp=4.0, brightness_cutoff=250, canny_cutoff=None, device=gpu)
    extract_slide_serial(slide, ex, b, slide_mpp=4.0, brightness_cutoff=250, canny_cutoff=None, supertiles_per_batch=2, device=gpu)
    fa, ca, _ = h5io.read_tile_features(a)
    fb, cb, _ = h5io.read_tile_features(b)
    assert st["tiles_seen"] >= 2 * 256 and st["tiles_kept"] == fa.shape[0] > 0
    assert np.array_equal(fa.view(np.uint16), fb.view(np.uint16)) and np.array_equal(ca.coords_um, cb.coords_um)


def test_extract_slide_never_writes_non_finite_features(gpu, tmp_path):
    """A checkpoint whose residual stream leaves fp16's range (random_vit_state_dict(init="overflow")): the pipelined path defers the encoder's
    per-call guard, checks the slide's features once, moves the encoder to its safe packing and runs the slide again (one warning); with
    check="raise" the slide raises and no file appears (STAMP's per-slide try/except skips it, preprocessing/__init__.py:328-336)."""
    import warnings

    from stamp_amd import h5io
    from stamp_amd.extractor import Extractor, hip_vit_extractor, u8_tile_transform
    from stamp_amd.preprocess import extract_slide, extract_slide_serial
    from stamp_amd.vit import PRESETS, FeatureRangeError, HipViT, random_vit_state_dict

    cfg = PRESETS["test_tiny_fold"]
    sd = random_vit_state_dict(cfg, seed=13, init="overflow")
    slide = _fake_slide(1536, 1024, 6)
    strict = Extractor(model=HipViT(cfg, sd, device=gpu, chunk=16, check="raise"), transform=u8_tile_transform, identifier="strict")
    with pytest.raises(FeatureRangeError):
        extract_slide(slide, strict, tmp_path / "strict.h5", slide_mpp=0.5, brightness_cutoff=250, canny_cutoff=None, device=gpu)
    assert not (tmp_path / "strict.h5").exists()
    ex = hip_vit_extractor("test_tiny_fold", sd, device=gpu, chunk=16, identifier="amdstamp-test")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        st = extract_slide(slide, ex, tmp_path / "a.h5", slide_mpp=0.5, brightness_cutoff=250, canny_cutoff=None, device=gpu)
    assert st["range_retries"] == 1 and ex.model.safe_level == 1 and sum("safe level 1" in str(x.message) for x in w) == 1
    fa, ca, _ = h5io.read_tile_features(tmp_path / "a.h5")
    assert fa.shape[0] == st["tiles_kept"] > 0 and np.isfinite(fa.astype(np.float32)).all()
    ex2 = hip_vit_extractor("test_tiny_fold", sd, device=gpu, chunk=16, identifier="amdstamp-test")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        extract_slide_serial(slide, ex2, tmp_path / "b.h5", slide_mpp=0.5, brightness_cutoff=250, canny_cutoff=None, supertiles_per_batch=2, device=gpu)
    fb, cb, _ = h5io.read_tile_features(tmp_path / "b.h5")
    assert np.array_equal(fa.view(np.uint16), fb.view(np.uint16)) and np.array_equal(ca.coords_um, cb.coords_um)


def _fake_slide_class(w, h, seed, broken_after=None):
    """openslide's surface over a synthetic image (tools/make_golden.py::FakeSlide); `broken_after`: read_region fails from that call on."""
    from PIL import Image
    rgb = ot.synthetic_slide(w, h, seed)
    im = Image.fromarray(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]), "RGBA")

    class Slide:
        dimensions = (w, h)
        calls = 0
        closed = False

        def read_region(self, loc, level, size):
            type(self).calls += 1
            if broken_after is not None and type(self).calls > broken_after:
                raise OSError("cannot read region")
            out = Image.new("RGBA", size, (0, 0, 0, 0))
            out.paste(im.crop((loc[0], loc[1], min(loc[0] + size[0], w), min(loc[1] + size[1], h))), (0, 0))
            return out

        def get_thumbnail(self, size):
            bg = Image.new("RGB", im.size, "#ffffff")
            t = Image.composite(im, bg, im)
            t.thumbnail(tuple(int(v) for v in size), Image.Resampling.LANCZOS)
            return t

        def close(self):
            type(self).closed = True
    return Slide


def test_extract_slides_one_pipeline_writes_the_files_of_per_slide_calls(gpu, tmp_path):
    """`extract_slides`: the rank's loop (reference preprocessing/__init__.py:269-286, 328-367) as one pipeline over all slides -- encoder chunks span
    slide boundaries, slide i + 1 is read under slide i's encoder calls -- writes, per slide, the file `extract_slide` writes: bit for bit, whatever
    the chunk / batch geometry and also when the slides differ in resolution; an existing output is skipped; a slide whose regions cannot be read is
    logged and skipped while its neighbours come out intact; a slide without foreground writes nothing."""
    from stamp_amd import h5io
    from stamp_amd.extractor import hip_vit_extractor
    from stamp_amd.preprocess import SlideJob, extract_slide, extract_slides
    from stamp_amd.vit import PRESETS, random_vit_state_dict

    cfg = PRESETS["test_tiny"]
    ex = hip_vit_extractor("test_tiny", random_vit_state_dict(cfg, seed=3), device=gpu, chunk=8, identifier="amdstamp-test")
    spec = [(2100, 1500, 11, 0.5), (1300, 2600, 12, 0.5), (4100, 3100, 13, 1.0), (900, 900, 14, 0.5), (2048, 2048, 15, 0.5)]
    classes = [_fake_slide_class(w, h, seed) for w, h, seed, _ in spec]
    ref_dir, out_dir = tmp_path / "ref", tmp_path / "out"
    for canny, chunk, spb in ((None, 7, 3), (0.02, 12, 2), (None, 10_000, 16), (0.02, 5, 64)):
        for d in (ref_dir, out_dir):
            if d.exists():
                for f in d.iterdir():
                    f.unlink()
        for i, (cls, (_, _, _, mpp)) in enumerate(zip(classes, spec)):
            extract_slide(cls(), ex, ref_dir / f"s{i}.h5", slide_mpp=mpp, brightness_cutoff=224, canny_cutoff=canny, device=gpu)
        broken = _fake_slide_class(1800, 1800, 21, broken_after=3)
        blank = type("Blank", (), {"dimensions": (1500, 1500), "read_region": None,
                                   "get_thumbnail": lambda self, size: __import__("PIL.Image").Image.new("RGB", tuple(int(v) for v in size), "#ffffff")})
        jobs = [SlideJob(classes[0], out_dir / "s0.h5", spec[0][3], "s0"),            # a callable (the class): opened by the pipeline, closed when read
                SlideJob(classes[1](), out_dir / "s1.h5", spec[1][3], "s1"),
                SlideJob(broken(), out_dir / "broken.h5", 0.5, "broken"),
                SlideJob(classes[2](), out_dir / "s2.h5", spec[2][3], "s2"),
                SlideJob(blank(), out_dir / "blank.h5", 0.5, "blank"),
                SlideJob(classes[3](), out_dir / "s3.h5", spec[3][3], "s3"),
                SlideJob(classes[4](), out_dir / "s4.h5", spec[4][3], "s4")]
        out_dir.mkdir(exist_ok=True)
        h5io.write_tile_features(out_dir / "s3.h5", torch.zeros(1, cfg.dim, dtype=torch.float16), np.zeros((1, 2), np.float32), extractor="x", tile_size_um=256.0,
                                 tile_size_px=224, code_hash="0", stamp_version="2.5.0", amdstamp_version="0")
        done = []
        res = extract_slides(jobs, ex, brightness_cutoff=224, canny_cutoff=canny, supertiles_per_batch=spb, encode_chunk=chunk, max_workers=3, device=gpu,
                             on_slide_done=lambda i, r: done.append(i))
        ref_status = ["written" if (ref_dir / f"s{i}.h5").exists() else "empty" for i in range(5)]          # (a slide without foreground writes nothing, in both forms)
        assert ref_status.count("written") >= 3
        assert [r["status"] for r in res] == [ref_status[0], ref_status[1], "failed", ref_status[2], "empty", "skipped", ref_status[4]], [(r["status"], r.get("error")) for r in res]
        assert "cannot read region" in res[2]["error"] and not (out_dir / "broken.h5").exists() and not (out_dir / "blank.h5").exists()
        assert sorted(done) == list(range(7)) and classes[0].closed
        for i in (0, 1, 2, 4):
            if ref_status[i] != "written":
                assert not (out_dir / f"s{i}.h5").exists()
                continue
            fr, cr, ar = h5io.read_tile_features(ref_dir / f"s{i}.h5")
            fp, cp, ap = h5io.read_tile_features(out_dir / f"s{i}.h5")
            assert np.array_equal(fp.view(np.uint16), fr.view(np.uint16)) and np.array_equal(cp.coords_um, cr.coords_um), (canny, chunk, spb, i)
            assert ap["extractor"] == ar["extractor"] and ap["tile_size_px"] == ar["tile_size_px"]
            assert res[[0, 1, 3, None, 6][i]]["tiles_kept"] == fr.shape[0]
        f3, _, _ = h5io.read_tile_features(out_dir / "s3.h5")
        assert f3.shape[0] == 1                                  # the existing file was left alone


def test_extract_slides_never_writes_non_finite_features(gpu, tmp_path):
    """The multi-slide pipeline keeps the single-slide guarantee: with an overflowing checkpoint (random_vit_state_dict(init="overflow")) every slide's rows
    fail the per-slide check, nothing is written by the pipeline itself, and each slide goes through `extract_slide` alone afterwards -- which moves the
    encoder to its safe packing once (one warning) and writes finite features equal to the serial path's; with check="raise" every slide is reported
    failed, no file appears, and the loop still returns (STAMP logs and goes on, preprocessing/__init__.py:328-336)."""
    import warnings

    from stamp_amd import h5io
    from stamp_amd.extractor import Extractor, hip_vit_extractor, u8_tile_transform
    from stamp_amd.preprocess import SlideJob, extract_slide_serial, extract_slides
    from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict

    cfg = PRESETS["test_tiny_fold"]
    sd = random_vit_state_dict(cfg, seed=13, init="overflow")
    slides = [_fake_slide(1536, 1024, 6), _fake_slide(1024, 2048, 7)]
    strict = Extractor(model=HipViT(cfg, sd, device=gpu, chunk=16, check="raise"), transform=u8_tile_transform, identifier="strict")
    res = extract_slides([SlideJob(s, tmp_path / f"strict{i}.h5", 0.5, f"strict{i}") for i, s in enumerate(slides)], strict, brightness_cutoff=250, canny_cutoff=None, device=gpu)
    assert [r["status"] for r in res] == ["failed", "failed"] and all("FeatureRangeError" in r["error"] for r in res)
    assert not any((tmp_path / f"strict{i}.h5").exists() for i in range(2))
    ex = hip_vit_extractor("test_tiny_fold", sd, device=gpu, chunk=16, identifier="amdstamp-test")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = extract_slides([SlideJob(s, tmp_path / f"a{i}.h5", 0.5, f"a{i}") for i, s in enumerate(slides)], ex, brightness_cutoff=250, canny_cutoff=None, device=gpu)
    assert [r["status"] for r in res] == ["written", "written"] and ex.model.safe_level == 1 and sum("safe level 1" in str(x.message) for x in w) == 1
    ex2 = hip_vit_extractor("test_tiny_fold", sd, device=gpu, chunk=16, identifier="amdstamp-test")
    for i, s in enumerate(slides):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            extract_slide_serial(s, ex2, tmp_path / f"b{i}.h5", slide_mpp=0.5, brightness_cutoff=250, canny_cutoff=None, supertiles_per_batch=2, device=gpu)
        fa, ca, _ = h5io.read_tile_features(tmp_path / f"a{i}.h5")
        fb, cb, _ = h5io.read_tile_features(tmp_path / f"b{i}.h5")
        assert fa.shape[0] > 0 and np.isfinite(fa.astype(np.float32)).all()
        assert np.array_equal(fa.view(np.uint16), fb.view(np.uint16)) and np.array_equal(ca.coords_um, cb.coords_um)
