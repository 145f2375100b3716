"""One slide through the whole extraction path on the GPU: supertiles -> tiles -> texture filter -> tile encoder -> feature .h5.

Mirrors the per-slide body of the reference's `extract_` (src/stamp/preprocessing/__init__.py:275-367) with the tile producer of
`tiling.py` (`_supertiles` :294-347, `_tiles` :196-247, `_tiles_with_tissue` :171-193) in front of it:

    slide.get_thumbnail -> foreground supertiles (host, a few hundred pixels)                         tiling.foreground_coords
    slide.read_region RGBA [S, S, 4], `max_workers` reader threads (host I/O, as the reference)      ThreadPoolExecutor
      -> pinned batch -> H2D -> PIL-exact bicubic resize + crop into 224 x 224 x 3 tiles (HIP)       tiling.supertiles_to_tiles
      -> Canny edge fraction >= canny_cutoff (HIP)                                                   extractor.has_enough_texture
      -> tile encoder (HIP) -> fp16 features                                                         Extractor.model
    feats fp16 [N, D] + coords f32 [N, 2] (um) + attributes -> .h5 in STAMP's schema                h5io.write_tile_features

`slide` is anything with openslide's `dimensions`, `read_region((x, y), 0, (w, h)) -> RGBA PIL image` and `get_thumbnail(size)`;
opening slide files is the caller's business (openslide / the reference's own reader).  Tiles come out in supertile order (the
reference yields in thread-completion order, :346, which is why its tests sort by coordinate before comparing).
"""
from __future__ import annotations

import os
from concurrent import futures
from pathlib import Path

import numpy as np
import torch

from . import h5io, tiling
from .encoder import AMDSTAMP_VERSION, STAMP_FORMAT_VERSION, code_hash
from .extractor import Extractor, has_enough_texture


_TIMELINE: list | None = None          # debugging aid (tools/slide_only.py timeline): host times of batch arrivals / encoder calls + GPU events


def _region_array(slide, x: int, y: int, s: int) -> np.ndarray:
    im = slide.read_region((x, y), 0, (s, s))
    if im.mode != "RGBA":              # openslide hands out RGBA already: no second 4 MB copy under the GIL
        im = im.convert("RGBA")
    return np.asarray(im, dtype=np.uint8)


# Pinned staging buffers are kept between calls: pinning is a driver call that walks the pages (~0.15 s for the 1 GB ring of one extract_slide call --
# 4 % of a 20 k-tile slide), and a rank calls these functions once per slide / once per list.  Buffers go back to the pool only after the call has
# synchronised its streams; at most _PIN_POOL_MAX bytes are kept.
_PIN_POOL: dict[int, list] = {}
_PIN_POOL_MAX = 6 << 30
_pin_lock = __import__("threading").Lock()


def _pinned_take(nbytes: int) -> torch.Tensor:
    """A pinned uint8 buffer of at least nbytes (rounded up to 1 MiB)."""
    size = max(1, -(-int(nbytes) // (1 << 20))) << 20
    with _pin_lock:
        lst = _PIN_POOL.get(size)
        if lst:
            return lst.pop()
    return torch.empty(size, dtype=torch.uint8).pin_memory()


def _pinned_give(bufs) -> None:
    with _pin_lock:
        held = sum(k * len(v) for k, v in _PIN_POOL.items())
        for b in bufs:
            if b is not None and held + b.numel() <= _PIN_POOL_MAX:
                _PIN_POOL.setdefault(b.numel(), []).append(b)
                held += b.numel()


def _nonfinite_rows(feats: torch.Tensor) -> int:
    """Rows of a host feature matrix that hold a NaN or an infinity.  fp16 is tested on its bits (exponent all ones): torch's CPU `isfinite` on a half tensor
    converts element by element -- 0.27 s for 2 304 x 1 024 values, on the thread that competes with the reader threads for the interpreter."""
    if feats.numel() == 0:
        return 0
    if feats.dtype == torch.float16 and feats.device.type == "cpu":
        mag = feats.contiguous().numpy().view(np.uint16) & 0x7FFF          # |x| as an integer: finite <=> below the all-ones exponent
        if int(mag.max()) < 0x7C00:
            return 0
        return int((mag.max(axis=1) >= 0x7C00).sum())
    return int((~torch.isfinite(feats)).any(dim=1).sum())


def _write(output_path, feats, coords, extractor, tile_size_um, tile_size_px):
    h5io.write_tile_features(Path(output_path), feats, coords.astype(np.float32), extractor=str(extractor.identifier), tile_size_um=tile_size_um,
                             tile_size_px=tile_size_px, code_hash=code_hash()[:8], stamp_version=STAMP_FORMAT_VERSION, amdstamp_version=AMDSTAMP_VERSION)


@torch.inference_mode()
def extract_slide(slide, extractor: Extractor, output_path, *, slide_mpp: float, tile_size_um: float = 256.0, tile_size_px: int = 224,
                  max_supertile_size_slide_px: int = 2 ** 10, brightness_cutoff: int | None = 240, canny_cutoff: float | None = 0.02,
                  max_workers: int = 8, supertiles_per_batch: int = 64, encode_chunk: int | None = None, device="cuda") -> dict:
    """Writes `output_path` (nothing if the slide has no tiles, like the reference :338-340) and returns counters.
    Defaults are the reference's (preprocessing/config.py:46-66; max_supertile_size_slide_px = 2**10 at __init__.py:307).

    A three-stage pipeline (the reference's own structure -- reader threads -> one consumer, tiling.py:326-346 -- with the consumer on the GPU):
      decode   `max_workers` reader threads fill a ring of six pinned supertile batches while the GPU works (a producer thread runs
               `read_region` for batch i+1, i+2 under batch i's GPU work);
      prepare  H2D on its own stream (DMA engines), then on the compute stream: PIL-exact resize + crop -> Canny edge fraction -> keep-mask
               compaction ON THE DEVICE (`amds_compact_rows_u8`: kept tiles are appended in order to an accumulation buffer whose fill level
               lives in device memory; no boolean index, no host round trip per batch -- the host reads the keep decisions asynchronously,
               whenever they arrive, only to know the coordinates and when a chunk is full);
      encode   whenever `encode_chunk` (default: the model's chunk, 1020) kept tiles are known to have accumulated, ONE encoder call on
               them, an asynchronous fp16 D2H of its features, and `amds_compact_shift_u8` moves the rest to the other buffer.
    The host never waits for the GPU before the end of the slide (unless the accumulation buffer -- two chunks -- would overflow).  Tiles come
    out in supertile order, so the file equals `extract_slide_serial`'s bit for bit (a tile's feature does not depend on its batch)."""
    import queue
    import threading

    from . import _lib, ops
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("extract_slide runs on the GPU only (no CPU fallback)")
    geo = tiling.supertile_geometry(slide_mpp, tile_size_um, tile_size_px, max_supertile_size_slide_px)
    S, k = geo.supertile_size_slide_px, geo.tiles_per_side
    dims = tuple(int(v) for v in slide.dimensions)
    gw, gh = tiling.thumbnail_size(dims, S)
    origins = tiling.foreground_coords(dims, slide.get_thumbnail((2 * gw, 2 * gh)), S, brightness_cutoff)
    import time as _time
    t_begin = _time.perf_counter()
    stats = {"supertiles": len(origins), "tiles_seen": 0, "tiles_kept": 0, "encoder_calls": 0, "host_syncs": 0, "wait_reader_s": 0.0, "wait_gpu_s": 0.0}
    if not origins:
        return stats
    model = extractor.model
    kk, t = k * k, int(tile_size_px)
    # amds_compact_rows_u8 takes at most 4096 tiles per call: a low-resolution slide (k = 9 from mpp 2.25 on: 81 tiles per supertile) gets
    # fewer supertiles per batch instead of an AMDS_ERR_INVALID on every batch
    spb = max(1, min(int(supertiles_per_batch), 4096 // kk))
    chunk = int(encode_chunk or getattr(model, "chunk", 1020))
    row_bytes = t * t * 3
    # pinned ring: four batches (one being consumed, up to three being decoded)
    n_buf = 4
    n_buf = max(2, min(n_buf, (4 << 30) // (spb * S * S * 4), -(-len(origins) // spb) + 1))       # <= 4 GB of pinned memory, not more slots than batches
    host_raw = [_pinned_take(spb * S * S * 4) for _ in range(n_buf)]
    host = [h[: spb * S * S * 4].view(spb, S, S, 4) for h in host_raw]
    host_np = [h.numpy() for h in host]               # the reader threads write through numpy views (no torch state in worker threads)
    buf_free: "queue.Queue[int]" = queue.Queue()
    for i in range(n_buf):
        buf_free.put(i)
    buf_ev: list = [None] * n_buf                      # H2D of the buffer's previous content has completed
    ready: "queue.Queue" = queue.Queue(maxsize=n_buf)
    stop = threading.Event()

    def producer():
        # up to n_buf batches are being decoded at any time: the reads of batch i+1, i+2 are submitted to the pool as soon as a pinned
        # buffer is free, batches are handed over in order as they complete
        try:
            inflight: list = []
            with futures.ThreadPoolExecutor(max_workers) as pool:
                def hand_over():
                    b0, batch0, futs0 = inflight.pop(0)
                    t_ = _time.perf_counter()
                    for fu in futs0:
                        fu.result()
                    stats["reader_wait_decode_s"] = stats.get("reader_wait_decode_s", 0.0) + _time.perf_counter() - t_
                    ready.put((b0, batch0))
                for i in range(0, len(origins), spb):
                    if stop.is_set():
                        return
                    while len(inflight) >= n_buf:
                        hand_over()
                    t_ = _time.perf_counter()
                    b = buf_free.get()
                    stats["reader_wait_free_buffer_s"] = stats.get("reader_wait_free_buffer_s", 0.0) + _time.perf_counter() - t_
                    t_ = _time.perf_counter()
                    if buf_ev[b] is not None:
                        buf_ev[b].synchronize()
                    stats["reader_wait_h2d_s"] = stats.get("reader_wait_h2d_s", 0.0) + _time.perf_counter() - t_
                    batch = origins[i:i + spb]

                    def fill(j, o, b=b):
                        host_np[b][j][...] = _region_array(slide, o[0], o[1], S)
                    inflight.append((b, batch, [pool.submit(fill, j, o) for j, o in enumerate(batch)]))
                while inflight:
                    hand_over()
            ready.put(None)
        except BaseException as e:      # surfaces in the consumer: the reference logs and skips the slide (__init__.py:328-336)
            ready.put(e)

    with torch.cuda.device(dev):
        # ONE compute stream (the caller's) for resize / Canny / compaction AND the encoder: side-stream kernels that need a CU's LDS
        # (Canny: 100 KB) were starved for milliseconds each behind the encoder's GEMM workgroups (128 KB LDS, thousands queued), so the
        # small kernels of the NEXT chunk simply run between two encoder calls (~10 ms per 1020 tiles).  What overlaps with the encoder is
        # everything that is not a kernel: the reader threads, the H2D copies (their own stream: DMA engines), the host's bookkeeping.
        cs = torch.cuda.current_stream()
        h2d = torch.cuda.Stream()
        cap = 2 * chunk + spb * kk                     # the host learns keep decisions late (it never waits for them): room for two chunks
        acc = [torch.empty(cap, t, t, 3, dtype=torch.uint8, device=dev) for _ in range(2)]
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        # device-side ring (supertiles and the tiles cut from them), allocated once and LONGER than the pinned ring (three chunks of tiles; HBM is not the scarce resource): a device slot
        # is recycled only when the compute stream has cut its tiles, and that stream is busy with the encoder for a chunk at a time, while a
        # pinned slot is free again as soon as its H2D copy is done -- so the readers keep going through an encoder call
        n_dev = max(n_buf, min(-(-3 * chunk // (spb * kk)) + 2, (8 << 30) // (spb * S * S * 4), -(-len(origins) // spb) + 1))
        d_rgba = [torch.empty(spb, S, S, 4, dtype=torch.uint8, device=dev) for _ in range(n_dev)]
        d_tiles = [torch.empty(spb * kk, t, t, 3, dtype=torch.uint8, device=dev) for _ in range(n_dev)]
        d_ws = torch.empty(max(_lib.lib().amds_supertiles_to_tiles_workspace_bytes(spb, S, k, t), 4), dtype=torch.uint8, device=dev)
        d_done: list = [None] * n_dev                  # the compute stream has finished with this slot's device buffers
        # host-side staging allocated ONCE (pinning memory is a driver call): per-batch keep slots, and the feature rows of the whole slide
        n_batches = (len(origins) + spb - 1) // spb
        slots_ring = torch.empty(n_batches, spb * kk, dtype=torch.int32).pin_memory()
        feat_dim = int(getattr(getattr(model, "cfg", None), "out_dim", 0) or getattr(getattr(model, "cfg", None), "dim", 0) or 0)
        feats_raw = _pinned_take(len(origins) * kk * feat_dim * 2) if feat_dim else None
        feats_host = feats_raw[: len(origins) * kk * feat_dim * 2].view(torch.float16).view(len(origins) * kk, feat_dim) if feat_dim else None
        # micrometre coordinates of every tile of every foreground supertile, in yield order (tiling.py:237-246), vectorised
        og = np.asarray(origins, dtype=np.float64) * slide_mpp                                  # [n, 2] (x, y)
        off = np.array([(x * tile_size_um, y * tile_size_um) for y in range(k) for x in range(k)], dtype=np.float64)
        all_coords = (og[:, None, :] + off[None, :, :]).reshape(-1, 2)
        cs.synchronize()
        cur = 0
        pending: list = []                             # batches whose keep decisions have not reached the host yet, in order
        kept_coords: list = []                         # coordinates of the kept tiles, in order (== the order of the feature rows)
        kept_known = 0                                 # kept tiles among the batches the host has heard from
        encoded = 0                                    # tiles handed to the encoder so far
        unknown = 0                                    # tiles of the batches in `pending` (each of them may or may not have been kept)
        feats_parts: list = []
        batch_idx = 0
        stats["setup_s"] = round(_time.perf_counter() - t_begin, 3)
        th = threading.Thread(target=producer, daemon=True)
        th.start()

        def absorb(block: bool) -> None:
            nonlocal kept_known, unknown
            while pending and (block or pending[0][1].query()):
                slots_h, ev, cu = pending.pop(0)
                unknown -= slots_h.shape[0]
                if block:
                    t_ = _time.perf_counter()
                    ev.synchronize()
                    stats["wait_gpu_s"] += _time.perf_counter() - t_
                    stats["host_syncs"] += 1
                sl = slots_h.numpy()
                if (sl == -2).any():
                    raise RuntimeError("extract_slide: accumulation buffer overflow")
                keep = sl >= 0
                kept_coords.append(cu[keep])
                kept_known += int(keep.sum())

        def encode(m: int) -> None:
            """The first m accumulated tiles -> encoder -> pinned feature rows; the rest of the buffer moves to the front of the other one."""
            nonlocal cur, encoded
            if _TIMELINE is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(cs)
                _TIMELINE.append(("enc_enqueue", _time.perf_counter() - t_begin, e0))
            f = model(acc[cur][:m]).detach().half()
            fh = feats_host[encoded:encoded + m] if feats_host is not None and f.shape[1] == feats_host.shape[1] else torch.empty(f.shape, dtype=torch.float16).pin_memory()
            feats_parts.append((fh, None, f))          # the D2H of the features is issued after the last encoder call (2 KB per tile stay on the device till then)
            stats["encoder_calls"] += 1
            other = 1 - cur
            _lib.check(_lib.lib().amds_compact_shift_u8(acc[cur].data_ptr(), acc[other].data_ptr(), row_bytes, m, cap - m, count.data_ptr(), cs.cuda_stream),
                       "compact_shift")
            cur = other
            encoded += m

        # the encoder's own per-call guard (vit.HipViT.check: one read-back + synchronisation per call) is deferred: the slide's features are
        # checked once, below, before anything is written
        guarded = [m for m in model.modules() if hasattr(m, "defer_check")] if hasattr(model, "modules") else []
        prev_defer = [m.defer_check for m in guarded]
        for m in guarded:
            m.defer_check = True
            if hasattr(m, "range_diagnostics"):
                m.range_diagnostics(reset=True)
        try:
            reader_done = False
            while not reader_done:
                absorb(False)
                while kept_known - encoded >= chunk:
                    encode(chunk)
                if kept_known - encoded + unknown + spb * kk > cap:      # the device buffer could overflow: wait for the oldest decisions
                    absorb(True)                                           # (kept_known - encoded < chunk here, so pending is not empty)
                    continue
                t_ = _time.perf_counter()
                try:
                    item = ready.get(timeout=0.002)
                except queue.Empty:
                    stats["wait_reader_s"] += _time.perf_counter() - t_
                    continue
                stats["wait_reader_s"] += _time.perf_counter() - t_
                if item is None:
                    reader_done = True
                    break
                if isinstance(item, BaseException):
                    raise item
                b, batch = item
                nb = len(batch)
                if _TIMELINE is not None:
                    _TIMELINE.append(("batch_ready", _time.perf_counter() - t_begin, None))
                ds = batch_idx % n_dev
                with torch.cuda.stream(h2d):        # its own stream: a pinned buffer is free again as soon as ITS copy is done
                    if d_done[ds] is not None:
                        h2d.wait_event(d_done[ds])
                    rgba = d_rgba[ds][:nb]
                    if _TIMELINE is not None:
                        e_s = torch.cuda.Event(enable_timing=True)
                        e_s.record(h2d)
                    rgba.copy_(host[b][:nb], non_blocking=True)
                    ev = torch.cuda.Event(enable_timing=_TIMELINE is not None)
                    ev.record(h2d)
                    if _TIMELINE is not None:
                        _TIMELINE.append(("h2d", _time.perf_counter() - t_begin, (e_s, ev)))
                    buf_ev[b] = ev
                    buf_free.put(b)
                cs.wait_event(ev)
                tiles = tiling.supertiles_to_tiles(rgba, k, t, out=d_tiles[ds], workspace=d_ws)
                frac = ops.tile_edge_fraction(tiles, 40, 100) if canny_cutoff is not None else None
                # the keep decisions go STRAIGHT into pinned host memory (device-accessible: the kernel stores over PCIe).  As D2H copy commands they
                # sat in the copy queue behind the encoder and held up the H2D copies submitted after them until the encoder call had finished
                slots_h = slots_ring[batch_idx, :tiles.shape[0]]
                _lib.check(_lib.lib().amds_compact_rows_u8(tiles.data_ptr(), row_bytes, None if frac is None else frac.data_ptr(),
                                                           float(canny_cutoff or 0.0), acc[cur].data_ptr(), cap, count.data_ptr(), slots_h.data_ptr(),
                                                           tiles.shape[0], cs.cuda_stream), "compact_rows")
                ev2 = torch.cuda.Event()
                ev2.record(cs)
                d_done[ds] = ev2
                pending.append((slots_h, ev2, all_coords[batch_idx * spb * kk: batch_idx * spb * kk + nb * kk]))
                batch_idx += 1
                stats["tiles_seen"] += nb * kk
                unknown += nb * kk
            absorb(True)
            while kept_known - encoded > 0:
                encode(min(chunk, kept_known - encoded))
        finally:
            for m, d in zip(guarded, prev_defer):
                m.defer_check = d
            stop.set()
            while th.is_alive():                        # unblock a producer waiting for a free buffer
                try:
                    ready.get_nowait()
                except queue.Empty:
                    pass
                buf_free.put(0)
                th.join(timeout=0.05)
        t_ = _time.perf_counter()
        for fh, _e, f in feats_parts:
            fh.copy_(f, non_blocking=True)
        cs.synchronize()
        stats["wait_gpu_s"] += _time.perf_counter() - t_
        coords_parts = kept_coords
        stats["pipeline_s"] = round(_time.perf_counter() - t_begin - stats["setup_s"], 3)
    if not feats_parts:
        _pinned_give(host_raw + [feats_raw])
        return stats
    feats = torch.cat([p[0] for p in feats_parts]) if len(feats_parts) > 1 else feats_parts[0][0].clone()
    _pinned_give(host_raw + [feats_raw])              # (every stream that touched them has been synchronised; `feats` is a copy)
    coords = np.concatenate(coords_parts)
    stats["tiles_kept"] = int(feats.shape[0])
    # Nothing non-finite reaches the file (the reference writes whatever its model returned, __init__.py:338-345).  A tile encoder with a safer
    # packing to offer (vit.HipViT: LayerNorm un-folded / bf16) is moved one level up and the slide is run again; otherwise the slide raises and
    # STAMP's per-slide try/except skips it (:328-336).
    why = None
    n_bad = _nonfinite_rows(feats)
    if n_bad:
        why = f"{n_bad} of {feats.shape[0]} tiles have non-finite features"
    else:
        for m in guarded:      # finite, but rows with |mean| > 8 sigma went through a folded LayerNorm (vit.HipViT.call_verdict)
            d = m.range_diagnostics(reset=True) if getattr(m, "safe_level", 1) == 0 and hasattr(m, "range_diagnostics") else {}
            if d.get("rows_mean_over_8_sigma"):
                why = f"{d['rows_mean_over_8_sigma']} rows with |mean| > 8 sigma entered a folded LayerNorm (precision loss ~ |mean| / sigma)"
    if why is not None:
        from .vit import FeatureRangeError
        moved = [m.enable_safe_mode(why) for m in guarded if getattr(m, "check", "raise") == "fallback"]
        if any(moved):
            stats_retry = extract_slide(slide, extractor, output_path, slide_mpp=slide_mpp, tile_size_um=tile_size_um, tile_size_px=tile_size_px,
                                        max_supertile_size_slide_px=max_supertile_size_slide_px, brightness_cutoff=brightness_cutoff, canny_cutoff=canny_cutoff,
                                        max_workers=max_workers, supertiles_per_batch=supertiles_per_batch, encode_chunk=encode_chunk, device=device)
            stats_retry["range_retries"] = stats_retry.get("range_retries", 0) + 1
            return stats_retry
        raise FeatureRangeError(f"extract_slide: {why} ({extractor.identifier}); nothing written")
    t_ = _time.perf_counter()
    _write(output_path, feats, coords, extractor, tile_size_um, tile_size_px)
    stats["write_s"] = round(_time.perf_counter() - t_, 3)
    return stats


@torch.inference_mode()
def extract_slide_serial(slide, extractor: Extractor, output_path, *, slide_mpp: float, tile_size_um: float = 256.0, tile_size_px: int = 224,
                  max_supertile_size_slide_px: int = 2 ** 10, brightness_cutoff: int | None = 240, canny_cutoff: float | None = 0.02,
                  max_workers: int = 8, supertiles_per_batch: int = 16, device="cuda") -> dict:
    """The un-pipelined form (round 2): decode a batch -> wait -> H2D -> resize -> Canny -> boolean index (host sync) -> encoder -> `.cpu()`,
    batch by batch.  Kept as the A/B partner of `extract_slide`: same files bit for bit (tests/test_gpu_tiling.py), a fraction of the rate."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("extract_slide runs on the GPU only (no CPU fallback)")
    geo = tiling.supertile_geometry(slide_mpp, tile_size_um, tile_size_px, max_supertile_size_slide_px)
    S, k = geo.supertile_size_slide_px, geo.tiles_per_side
    dims = tuple(int(v) for v in slide.dimensions)
    gw, gh = tiling.thumbnail_size(dims, S)
    origins = tiling.foreground_coords(dims, slide.get_thumbnail((2 * gw, 2 * gh)), S, brightness_cutoff)
    feats, coords = [], []
    n_tiles = 0
    model = extractor.model
    host = torch.empty(supertiles_per_batch, S, S, 4, dtype=torch.uint8).pin_memory()
    host_np = host.numpy()                         # (the regions may come as read-only arrays: copied through the numpy view)
    with futures.ThreadPoolExecutor(max_workers) as pool:
        for i in range(0, len(origins), supertiles_per_batch):
            batch = origins[i:i + supertiles_per_batch]
            for j, arr in enumerate(pool.map(lambda o: _region_array(slide, o[0], o[1], S), batch)):
                host_np[j] = arr
            tiles = tiling.supertiles_to_tiles(host[:len(batch)].to(dev, non_blocking=True), k, tile_size_px)
            cu = np.concatenate([tiling.tile_coords_um(o, slide_mpp, k, tile_size_um) for o in batch])
            n_tiles += tiles.shape[0]
            if canny_cutoff is not None:
                keep = has_enough_texture(tiles, canny_cutoff)
                tiles, cu = tiles[keep], cu[keep.cpu().numpy()]
            if tiles.shape[0]:
                feats.append(model(tiles.contiguous()).detach().half().cpu())
                coords.append(cu)
    stats = {"supertiles": len(origins), "tiles_seen": n_tiles, "tiles_kept": int(sum(f.shape[0] for f in feats))}
    if not feats:
        return stats
    h5io.write_tile_features(Path(output_path), torch.cat(feats), np.concatenate(coords).astype(np.float32), extractor=str(extractor.identifier),
                             tile_size_um=tile_size_um, tile_size_px=tile_size_px, code_hash=code_hash()[:8], stamp_version=STAMP_FORMAT_VERSION,
                             amdstamp_version=AMDSTAMP_VERSION)
    return stats


# ============================================================================================================================================
# The rank's loop over slides (reference src/stamp/preprocessing/__init__.py:269-286 loop + skip-existing, :328-336 per-slide try / except,
# :338-367 one .h5 per slide) as ONE pipeline over all of them.
# ============================================================================================================================================
import dataclasses as _dc
import logging as _logging
from typing import Any, Callable, Iterable

_log = _logging.getLogger("stamp_amd")


@_dc.dataclass
class SlideJob:
    """One slide of a rank's list.  `slide`: an object with openslide's `dimensions` / `read_region` / `get_thumbnail`, or a zero-argument
    callable (a function, a functools.partial, a class) that opens one -- called in the reader thread when the slide's turn comes, so opening slide i + 1 (and its thumbnail) happens
    under slide i's GPU work; an opened object that has `close()` is closed when its last region has been read."""
    slide: Any
    output_path: Any
    slide_mpp: float | None          # None: looked up in the slide's metadata (tiling.get_slide_mpp, the reference's get_slide_mpp_) before the pipeline starts
    name: str = ""


@_dc.dataclass
class _Plan:
    idx: int
    slide: Any
    S: int
    k: int
    spb: int
    origins: list
    coords: np.ndarray          # float64 [n_supertiles * k * k, 2], yield order
    opened_here: bool


def resolve_slide_mpps(jobs, default_mpp: float | None = None) -> tuple[list, dict]:
    """Jobs with `slide_mpp=None` get theirs from the slide's metadata (`tiling.get_slide_mpp`; a factory is opened, read and closed).  Returns (the job list
    with None in place of every job whose resolution could not be determined, {index: repr(error)})."""
    out, failed = [], {}
    for i, j in enumerate(jobs):
        if j.slide_mpp is not None:
            out.append(j)
            continue
        s = None
        try:
            s = j.slide() if isinstance(j.slide, type) or (callable(j.slide) and not hasattr(j.slide, "read_region")) else j.slide
            out.append(_dc.replace(j, slide_mpp=tiling.get_slide_mpp(s, default_mpp=default_mpp)))
        except Exception as e:
            _log.exception(f"Failed reading the resolution of {j.name or j.output_path}")
            out.append(None)
            failed[i] = repr(e)
        finally:
            if s is not None and s is not j.slide and hasattr(s, "close"):
                s.close()
    return out, failed


@torch.inference_mode()
def extract_slides(jobs: Iterable[SlideJob], extractor: Extractor, *, tile_size_um: float = 256.0, tile_size_px: int = 224,
                   max_supertile_size_slide_px: int = 2 ** 10, brightness_cutoff: int | None = 240, canny_cutoff: float | None = 0.02,
                   max_workers: int = 8, supertiles_per_batch: int = 64, encode_chunk: int | None = None, device="cuda", skip_existing: bool = True,
                   on_slide_done: Callable[[int, dict], None] | None = None, default_slide_mpp: float | None = None) -> list[dict]:
    """Every job's slide -> its feature `.h5`, the files `extract_slide` writes bit for bit, as ONE pipeline: the reader threads, the pinned
    ring, the device ring, the accumulation buffer and the encoder calls are shared by all slides, so slide i + 1 is opened, thumbnailed and read
    under slide i's last encoder calls, and an encoder chunk takes the tail of one slide together with the head of the next (a tile's features do
    not depend on the batch it travels in: tests/test_gpu_vit.py).  `extract_slide` per slide pays its start-up and its drain every time
    (0.82 of the encoder's rate on 20 k-tile slides, a third of it on 1 k-tile ones).

    Per slide, as the reference: an existing output is skipped (`skip_existing`, :277-283); a slide without foreground tiles writes nothing
    (:338-340); ANY exception while opening / reading / checking / writing it is logged and the loop goes on (:328-336) -- that includes
    `FeatureRangeError`: features are checked per slide before they are written (non-finite values; rows with |mean| > 8 sigma through a folded
    LayerNorm, counted per encoder call), and a slide that fails the check goes through `extract_slide` alone afterwards, which moves the
    encoder to a safer packing or raises.  Returns one dict per job, in job order: {"status": "written" | "skipped" | "empty" | "failed", ...}."""
    import queue
    import threading
    import time as _time

    from . import _lib, ops
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("extract_slides runs on the GPU only (no CPU fallback)")
    jobs = list(jobs)
    if any(j.slide_mpp is None for j in jobs):
        # `stamp preprocess` reads the resolution out of the slide (reference __init__.py:288-291 -> tiling.get_slide_mpp_); a slide whose metadata has none
        # (and no default_slide_mpp) fails like any other per-slide error and the rest goes through the pipeline
        known, failed = resolve_slide_mpps(jobs, default_slide_mpp)
        idx = [i for i, j in enumerate(known) if j is not None]
        sub = extract_slides([known[i] for i in idx], extractor, tile_size_um=tile_size_um, tile_size_px=tile_size_px,
                             max_supertile_size_slide_px=max_supertile_size_slide_px, brightness_cutoff=brightness_cutoff, canny_cutoff=canny_cutoff,
                             max_workers=max_workers, supertiles_per_batch=supertiles_per_batch, encode_chunk=encode_chunk, device=device, skip_existing=skip_existing,
                             on_slide_done=None if on_slide_done is None else (lambda k, r: on_slide_done(idx[k], r)))
        out = [{"status": "failed", "name": j.name or str(j.output_path), "error": failed.get(i, "")} for i, j in enumerate(jobs)]
        for k, i in enumerate(idx):
            out[i] = sub[k]
        if on_slide_done is not None:
            for i in failed:
                on_slide_done(i, out[i])
        return out
    results: list[dict] = [{"status": "pending", "name": j.name or str(j.output_path)} for j in jobs]
    if not jobs:
        return results
    model = extractor.model
    t = int(tile_size_px)
    row_bytes = t * t * 3
    chunk = int(encode_chunk or getattr(model, "chunk", 1020))
    geos = [tiling.supertile_geometry(j.slide_mpp, tile_size_um, tile_size_px, max_supertile_size_slide_px) for j in jobs]
    spbs = [max(1, min(int(supertiles_per_batch), 4096 // (g.tiles_per_side ** 2))) for g in geos]
    max_rgba = max(s * g.supertile_size_slide_px ** 2 * 4 for s, g in zip(spbs, geos))          # bytes of one batch of supertiles
    max_tiles = max(s * g.tiles_per_side ** 2 for s, g in zip(spbs, geos))                        # tiles one batch can yield
    n_buf = max(2, min(6, (4 << 30) // max_rgba))          # one batch being copied, up to five being decoded
    host = [_pinned_take(max_rgba) for _ in range(n_buf)]
    host_np = [h.numpy() for h in host]
    buf_free: "queue.Queue[int]" = queue.Queue()
    for i in range(n_buf):
        buf_free.put(i)
    buf_ev: list = [None] * n_buf
    ready: "queue.Queue" = queue.Queue(maxsize=n_buf + 4)
    stop = threading.Event()
    t_begin = _time.perf_counter()

    def plan_slide(idx: int) -> _Plan | None:
        job, geo = jobs[idx], geos[idx]
        slide = job.slide() if isinstance(job.slide, type) or (callable(job.slide) and not hasattr(job.slide, "read_region")) else job.slide
        S, k = geo.supertile_size_slide_px, geo.tiles_per_side
        dims = tuple(int(v) for v in slide.dimensions)
        gw, gh = tiling.thumbnail_size(dims, S)
        origins = tiling.foreground_coords(dims, slide.get_thumbnail((2 * gw, 2 * gh)), S, brightness_cutoff)
        og = np.asarray(origins, dtype=np.float64).reshape(-1, 2) * job.slide_mpp
        off = np.array([(x * tile_size_um, y * tile_size_um) for y in range(k) for x in range(k)], dtype=np.float64)
        coords = (og[:, None, :] + off[None, :, :]).reshape(-1, 2)
        return _Plan(idx, slide, S, k, spbs[idx], origins, coords, slide is not job.slide)

    def producer():
        # one pool of reader threads for the whole list; batches are handed over in order; at most n_buf of them are being decoded at any time
        try:
            inflight: list = []
            with futures.ThreadPoolExecutor(max_workers) as pool:
                def hand_over():
                    kind, payload, futs0 = inflight.pop(0)
                    err = None
                    for fu in futs0:
                        try:
                            fu.result()
                        except BaseException as e:          # a region that cannot be read fails ITS slide, not the loop
                            err = err or e
                    if kind == "batch" and err is not None:
                        buf_free.put(payload[2])
                        ready.put(("slide_error", payload[0], err))
                    else:
                        ready.put((kind,) + payload)
                for idx, job in enumerate(jobs):
                    if stop.is_set():
                        return
                    if skip_existing and Path(job.output_path).exists():
                        inflight.append(("skipped", (idx,), []))
                        continue
                    try:
                        t_pl = _time.perf_counter()
                        plan = plan_slide(idx)
                        results[idx]["plan_s"] = round(_time.perf_counter() - t_pl, 3)
                        results[idx]["planned_at_s"] = round(_time.perf_counter() - t_begin, 3)
                    except BaseException as e:
                        inflight.append(("slide_error", (idx, e), []))
                        continue
                    inflight.append(("slide", (plan,), []))
                    for bi, i in enumerate(range(0, len(plan.origins), plan.spb)):
                        if stop.is_set():
                            return
                        while len([x for x in inflight if x[0] == "batch"]) >= n_buf:
                            hand_over()
                        b = buf_free.get()
                        if stop.is_set():
                            return
                        if buf_ev[b] is not None:
                            buf_ev[b].synchronize()
                        batch = plan.origins[i:i + plan.spb]
                        view = host_np[b][: len(batch) * plan.S * plan.S * 4].reshape(len(batch), plan.S, plan.S, 4)

                        def fill(j, o, view=view, plan=plan):
                            view[j][...] = _region_array(plan.slide, o[0], o[1], plan.S)
                        inflight.append(("batch", (idx, bi, b, len(batch)), [pool.submit(fill, j, o) for j, o in enumerate(batch)]))
                    inflight.append(("slide_read", (idx,), []))
                while inflight:
                    hand_over()
            ready.put(("end",))
        except BaseException as e:
            ready.put(("fatal", e))

    with torch.cuda.device(dev):
        cs = torch.cuda.current_stream()
        h2d = torch.cuda.Stream()
        cap = 2 * chunk + max_tiles
        acc = [torch.empty(cap, t, t, 3, dtype=torch.uint8, device=dev) for _ in range(2)]
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        n_dev = max(n_buf, min(-(-3 * chunk // max_tiles) + 2, (8 << 30) // max_rgba))
        d_rgba = [torch.empty(max_rgba, dtype=torch.uint8, device=dev) for _ in range(n_dev)]
        d_tiles = [torch.empty(max_tiles, t, t, 3, dtype=torch.uint8, device=dev) for _ in range(n_dev)]
        ws_bytes = max(_lib.lib().amds_supertiles_to_tiles_workspace_bytes(s, g.supertile_size_slide_px, g.tiles_per_side, t) for s, g in zip(spbs, geos))
        d_ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        d_done: list = [None] * n_dev
        slot_ring = torch.empty(64, max_tiles, dtype=torch.int32).pin_memory()          # keep decisions of the batches in flight
        slot_busy = [False] * slot_ring.shape[0]
        guarded = [m for m in model.modules() if hasattr(m, "defer_check")] if hasattr(model, "modules") else []
        prev_defer = [m.defer_check for m in guarded]
        for m in guarded:
            m.defer_check = True
            if hasattr(m, "range_diagnostics"):
                m.range_diagnostics(reset=True)
        diag_ring = torch.zeros(256, max(1, len(guarded)), 2, dtype=torch.int32).pin_memory()          # [call slot][guarded module][2 counters]
        cs.synchronize()

        # ---- bookkeeping.  Kept tiles form ONE global sequence (slide after slide, supertile order inside a slide); a slide owns rows
        # [first, first + kept) of it; encoder call c covers rows [enc_lo[c], enc_lo[c] + m) and lands in its own pinned block.
        st: dict[int, dict] = {}                      # per announced slide
        order: list[int] = []                         # slides in pipeline order (announced, not finalised)
        pending: list = []                            # (slot index, rows, event, slide idx, coords slice): decisions not yet on the host
        kept_known = encoded = unknown = 0
        calls: list = []                              # (row_lo, m, pinned block, event, diag slot)
        n_calls = 0                                   # encoder calls so far: `calls` is pruned, its length is not a sequence number
        free_blocks: list = []
        batch_no = 0
        writer = futures.ThreadPoolExecutor(1)
        write_futs: list = []
        deferred: list[int] = []                      # slides whose features failed the check: through extract_slide afterwards

        def absorb(block: bool) -> None:
            nonlocal kept_known, unknown
            while pending and (block or pending[0][2].query()):
                slot, rows, ev, sidx, cu = pending.pop(0)
                if block:
                    ev.synchronize()
                sl = slot_ring[slot, :rows].numpy()
                if (sl == -2).any():
                    raise RuntimeError("extract_slides: accumulation buffer overflow")
                keep = sl >= 0
                s = st[sidx]
                if s["first"] is None:
                    s["first"] = kept_known
                s["coords"].append(cu[keep])
                nk = int(keep.sum())
                s["kept"] += nk
                s["decided"] += 1
                kept_known += nk
                unknown -= rows
                slot_busy[slot] = False

        trace_ev: list = [] if os.environ.get("AMDS_SLIDES_TRACE") else None       # (start, end) events around every encoder call: GPU time between calls

        def encode(m: int) -> None:
            nonlocal cur, encoded, n_calls
            if trace_ev is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(cs)
            f = model(acc[cur][:m]).detach()
            if trace_ev is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(cs)
                trace_ev.append((e0, e1, _time.perf_counter() - t_begin, m))
            if f.dtype != torch.float16:
                f = f.half()
            f = f.contiguous()
            blk = free_blocks.pop() if free_blocks and free_blocks[-1].shape[1] == f.shape[1] else torch.empty(chunk, f.shape[1], dtype=torch.float16).pin_memory()
            # amds_export_words moves 32-bit words: an odd fp16 count would leave the last feature of the last row behind (stale in a reused block)
            assert f.shape[1] % 2 == 0 and m <= blk.shape[0] and f.data_ptr() % 4 == 0, (tuple(f.shape), m, tuple(blk.shape))
            # the rows reach the host through a kernel's stores (amds_export_words): no copy command behind the encoder in front of the next H2D copies
            _lib.check(_lib.lib().amds_export_words(f.data_ptr(), blk.data_ptr(), m * f.shape[1] // 2, 0, cs.cuda_stream), "export_words")
            # one ring slot per call, by a MONOTONIC call number (len(calls) shrinks when finalize_ready prunes: a retained call's slot would be handed out
            # again and its counters overwritten -- a silent miss of that call's |mean| > 8 sigma count); a retained call must not be lapped either
            assert len(calls) < diag_ring.shape[0], "extract_slides: more encoder calls await finalisation than the range-counter ring holds"
            dslot = n_calls % diag_ring.shape[0]
            n_calls += 1
            diag_ring[dslot].zero_()
            # every guarded module exports (and resets) its own counters: a list, not any(), which stops at the first True
            has_diag = any([bool(getattr(g, "export_range_counters", None) and g.export_range_counters(diag_ring[dslot, gi])) for gi, g in enumerate(guarded)])
            ev = torch.cuda.Event()
            ev.record(cs)
            calls.append((encoded, m, blk, ev, dslot if has_diag else -1, f))
            other = 1 - cur
            _lib.check(_lib.lib().amds_compact_shift_u8(acc[cur].data_ptr(), acc[other].data_ptr(), row_bytes, m, cap - m, count.data_ptr(), cs.cuda_stream),
                       "compact_shift")
            cur = other
            encoded += m

        def finalize_ready(block: bool) -> None:
            """Slides (in order) whose decisions are all known and whose rows have all been encoded AND have arrived: check + write on the writer thread."""
            while order:
                sidx = order[0]
                s = st[sidx]
                if not s["read_done"] or s["decided"] < s["batches"]:
                    return
                if s["first"] is None:                 # no batch of it was ever absorbed (no foreground; failed before its first batch): it owns no rows
                    order.pop(0)
                    write_futs.append(writer.submit(_finish_slide, sidx, s, torch.empty(0, 0, dtype=torch.float16), np.zeros((0, 2)), 0))
                    continue
                lo = s["first"]
                hi = lo + s["kept"]
                if encoded < hi:
                    return
                mine = [c for c in calls if c[0] < hi and c[0] + c[1] > lo]
                for c in mine:
                    if not c[3].query():
                        if not block:
                            return
                        c[3].synchronize()
                order.pop(0)
                parts = [c[2][max(lo, c[0]) - c[0]: min(hi, c[0] + c[1]) - c[0]] for c in mine]
                feats = torch.cat(parts) if len(parts) > 1 else (parts[0].clone() if parts else torch.empty(0, 0, dtype=torch.float16))
                big_mean = sum(int(diag_ring[c[4], :, 1].sum()) for c in mine if c[4] >= 0)
                coords = np.concatenate(s["coords"]) if s["coords"] else np.zeros((0, 2))
                # blocks no unfinalised slide needs any more go back to the pool
                floor = hi
                keep_calls = []
                for c in calls:
                    if c[0] + c[1] <= floor:
                        free_blocks.append(c[2])
                    else:
                        keep_calls.append(c)
                calls[:] = keep_calls
                write_futs.append(writer.submit(_finish_slide, sidx, s, feats, coords, big_mean))

        def _finish_slide(sidx, s, feats, coords, big_mean):
            job = jobs[sidx]
            r = results[sidx]
            t_fin = _time.perf_counter()
            r.update(supertiles=s["supertiles"], tiles_seen=s["seen"], tiles_kept=int(feats.shape[0]))
            try:
                if s["error"] is not None:
                    raise s["error"]
                if feats.shape[0] == 0:
                    r["status"] = "empty"
                elif (n_bad := _nonfinite_rows(feats)) or big_mean:
                    r["status"] = "recheck"
                    r["why"] = (f"{n_bad} of {feats.shape[0]} tiles have non-finite features" if not big_mean
                                else f"{big_mean} rows with |mean| > 8 sigma entered a folded LayerNorm in the encoder calls of this slide")
                else:
                    _write(job.output_path, feats, coords, extractor, tile_size_um, tile_size_px)
                    r["status"] = "written"
            except BaseException as e:            # the reference logs and goes on (:328-336)
                r["status"] = "failed"
                r["error"] = repr(e)
                _log.exception(f"Failed extracting features from {r['name']}")
            finally:
                if s["plan"] is not None and s["plan"].opened_here and hasattr(s["plan"].slide, "close"):
                    try:
                        s["plan"].slide.close()
                    except Exception:
                        pass
                    s["plan"].slide = None
            r["finish_s"] = round(_time.perf_counter() - t_fin, 3)
            r["finished_at_s"] = round(_time.perf_counter() - t_begin, 3)
            if on_slide_done is not None:
                on_slide_done(sidx, r)

        cur = 0
        th = threading.Thread(target=producer, daemon=True)
        th.start()
        wait_reader = 0.0
        try:
            done = False
            while not done:
                absorb(False)
                while kept_known - encoded >= chunk:
                    encode(chunk)
                finalize_ready(False)
                if kept_known - encoded + unknown + max_tiles > cap or all(slot_busy):
                    absorb(True)
                    continue
                t_ = _time.perf_counter()
                try:
                    item = ready.get(timeout=0.002)
                except queue.Empty:
                    wait_reader += _time.perf_counter() - t_
                    continue
                wait_reader += _time.perf_counter() - t_
                kind = item[0]
                if kind == "end":
                    done = True
                elif kind == "fatal":
                    raise item[1]
                elif kind == "skipped":
                    results[item[1]]["status"] = "skipped"
                    if on_slide_done is not None:
                        on_slide_done(item[1], results[item[1]])
                elif kind == "slide":
                    plan = item[1]
                    st[plan.idx] = {"plan": plan, "first": None, "kept": 0, "seen": 0, "decided": 0, "batches": 0, "read_done": False, "error": None, "coords": [],
                                    "supertiles": len(plan.origins)}
                    order.append(plan.idx)
                elif kind == "slide_read":
                    st[item[1]]["read_done"] = True
                elif kind == "slide_error":
                    sidx, err = item[1], item[2]
                    if sidx not in st:
                        st[sidx] = {"plan": None, "first": None, "kept": 0, "seen": 0, "decided": 0, "batches": 0, "read_done": True, "error": err, "coords": [], "supertiles": 0}
                        order.append(sidx)
                    else:
                        st[sidx]["error"] = st[sidx]["error"] or err
                elif kind == "batch":
                    _, sidx, bi, b, nb = item
                    s = st[sidx]
                    plan = s["plan"]
                    kk, S, k = plan.k * plan.k, plan.S, plan.k
                    if s["error"] is not None:           # the slide already failed: its remaining batches are dropped
                        buf_free.put(b)
                        continue
                    ds = batch_no % n_dev
                    nbytes = nb * S * S * 4
                    with torch.cuda.stream(h2d):
                        if d_done[ds] is not None:
                            h2d.wait_event(d_done[ds])
                        rgba = d_rgba[ds][:nbytes].view(nb, S, S, 4)
                        rgba.copy_(host[b][:nbytes].view(nb, S, S, 4), non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(h2d)
                        buf_ev[b] = ev
                        buf_free.put(b)
                    cs.wait_event(ev)
                    tiles = tiling.supertiles_to_tiles(rgba, k, t, out=d_tiles[ds], workspace=d_ws)
                    frac = ops.tile_edge_fraction(tiles, 40, 100) if canny_cutoff is not None else None
                    slot = slot_busy.index(False)
                    slot_busy[slot] = True
                    rows = tiles.shape[0]
                    _lib.check(_lib.lib().amds_compact_rows_u8(tiles.data_ptr(), row_bytes, None if frac is None else frac.data_ptr(), float(canny_cutoff or 0.0),
                                                               acc[cur].data_ptr(), cap, count.data_ptr(), slot_ring[slot].data_ptr(), rows, cs.cuda_stream), "compact_rows")
                    ev2 = torch.cuda.Event()
                    ev2.record(cs)
                    d_done[ds] = ev2
                    pending.append((slot, rows, ev2, sidx, plan.coords[bi * plan.spb * kk: bi * plan.spb * kk + nb * kk]))
                    s["batches"] += 1
                    s["seen"] += rows
                    unknown += rows
                    batch_no += 1
            absorb(True)
            while kept_known - encoded > 0:
                encode(min(chunk, kept_known - encoded))
            finalize_ready(True)
            for sidx in list(order):                    # slides announced but never completed (a reader failure in the middle): their error is reported
                s = st[sidx]
                if s["error"] is None:
                    s["error"] = RuntimeError("slide left incomplete by the pipeline")
                s["read_done"] = True
                s["decided"] = s["batches"]
            finalize_ready(True)
        finally:
            for m, d in zip(guarded, prev_defer):
                m.defer_check = d
            stop.set()
            while th.is_alive():
                try:
                    ready.get_nowait()
                except queue.Empty:
                    pass
                buf_free.put(0)
                th.join(timeout=0.05)
            for fu in write_futs:
                fu.result()
            writer.shutdown()
        cs.synchronize()
        h2d.synchronize()
    _pinned_give(host)
    # slides whose features failed the check: alone through extract_slide, which moves the encoder to a safer packing (and re-runs) or raises
    for sidx, r in enumerate(results):
        if r["status"] != "recheck":
            continue
        job = jobs[sidx]
        try:
            slide = job.slide() if isinstance(job.slide, type) or (callable(job.slide) and not hasattr(job.slide, "read_region")) else job.slide
            r2 = extract_slide(slide, extractor, job.output_path, slide_mpp=job.slide_mpp, tile_size_um=tile_size_um, tile_size_px=tile_size_px,
                               max_supertile_size_slide_px=max_supertile_size_slide_px, brightness_cutoff=brightness_cutoff, canny_cutoff=canny_cutoff,
                               max_workers=max_workers, supertiles_per_batch=supertiles_per_batch, encode_chunk=encode_chunk, device=device)
            r.update(r2)
            r["status"] = "written" if Path(job.output_path).exists() else "empty"
            if slide is not job.slide and hasattr(slide, "close"):          # opened here (a factory): closed here
                slide.close()
        except BaseException as e:
            r["status"] = "failed"
            r["error"] = repr(e)
            _log.exception(f"Failed extracting features from {r['name']}")
        if on_slide_done is not None:
            on_slide_done(sidx, r)
    total = _time.perf_counter() - t_begin
    if trace_ev:
        enc = [a.elapsed_time(b) for a, b, _, _ in trace_ev]
        gaps = [trace_ev[i][1].elapsed_time(trace_ev[i + 1][0]) for i in range(len(trace_ev) - 1)]
        results[0]["trace"] = {"encoder_calls": len(enc), "encoder_ms": round(sum(enc), 1), "between_calls_ms": round(sum(gaps), 1),
                               "first_call_enqueued_at_s": round(trace_ev[0][2], 3), "largest_gaps_ms": sorted((round(g, 1), i) for i, g in enumerate(gaps))[-8:],
                               "median_gap_ms": round(sorted(gaps)[len(gaps) // 2], 2) if gaps else None}
    for r in results:
        r.setdefault("status", "failed")
    results[0]["pipeline_s"] = round(total, 3)
    results[0]["wait_reader_s"] = round(wait_reader, 3)
    return results
