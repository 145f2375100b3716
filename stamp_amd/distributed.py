"""Slide-parallel multi-GPU plumbing: one process per GPU, RCCL (torch.distributed "nccl" on ROCm) over xGMI.

The reference is strictly single-device (``devices=1``, src/stamp/modeling/train.py:541-547) and parallelises
across machines only by "shuffle the slide list and skip existing outputs"
(src/stamp/preprocessing/__init__.py:269-286).  The unit of work is a slide with no cross-slide state, so
slides shard embarrassingly: every rank runs the whole tile pipeline on its own slides and writes its own
feature files.  There is exactly ONE data-path collective: an all-gather of slide-level embeddings
(``[slides, D]``, a few MB at most) so that every rank holds the table patient-level MIL trains on.  Tile-level
features are never gathered.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class DistCtx:
    rank: int
    world: int
    local_rank: int
    device: torch.device

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def init_from_env(prefer_gpu: bool = True) -> DistCtx:
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run) and join the group if WORLD_SIZE>1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
    use_gpu = prefer_gpu and torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if use_gpu else {}
        dist.init_process_group("nccl" if use_gpu else "gloo", rank=rank, world_size=world, **kw)
    return DistCtx(rank, world, local, device)


def barrier(ctx: DistCtx) -> None:
    if ctx.world > 1:
        if ctx.device.type == "cuda":
            dist.barrier(device_ids=[ctx.local_rank])
        else:
            dist.barrier()


def shard_slides(tile_counts: list[int], world: int) -> list[list[int]]:
    """Longest-processing-time-first assignment of slides (by tile count) to ranks; deterministic.

    Returns, per rank, the slide indices it owns (each list in ascending slide order)."""
    loads = [0] * world
    owned: list[list[int]] = [[] for _ in range(world)]
    for idx in sorted(range(len(tile_counts)), key=lambda i: (-tile_counts[i], i)):
        r = min(range(world), key=lambda j: (loads[j], j))
        owned[r].append(idx)
        loads[r] += tile_counts[idx]
    return [sorted(o) for o in owned]


def gather_slide_embeddings(ctx: DistCtx, local_emb: torch.Tensor, local_ids: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather per-rank slide embeddings ``[n_local, D]`` (+ their global slide ids) into ``[n_total, D]``.

    Ranks may own different numbers of slides: buffers are padded to the maximum count (one equal-size
    all-gather instead of an all-gather-v); rows whose id is -1 are padding.  Slides nobody owns stay zero."""
    D = local_emb.shape[1] if local_emb.dim() == 2 else 0
    out = torch.zeros(n_total, D, dtype=local_emb.dtype, device=local_emb.device)
    if ctx.world == 1:
        if local_ids.numel():
            out[local_ids.long()] = local_emb
        return out
    n_local = torch.tensor([local_emb.shape[0]], dtype=torch.int64, device=local_emb.device)
    dist.all_reduce(n_local, op=dist.ReduceOp.MAX)
    cap = int(n_local.item())
    emb = torch.zeros(cap, D, dtype=local_emb.dtype, device=local_emb.device)
    ids = torch.full((cap,), -1, dtype=torch.int64, device=local_emb.device)
    emb[: local_emb.shape[0]] = local_emb
    ids[: local_ids.shape[0]] = local_ids.to(torch.int64)
    all_emb = torch.empty(ctx.world * cap, D, dtype=emb.dtype, device=emb.device)
    all_ids = torch.empty(ctx.world * cap, dtype=torch.int64, device=emb.device)
    dist.all_gather_into_tensor(all_emb, emb)
    dist.all_gather_into_tensor(all_ids, ids)
    valid = all_ids >= 0
    out[all_ids[valid]] = all_emb[valid]
    return out


def average_gradients(flat_grad: torch.Tensor) -> torch.Tensor:
    """Data-parallel MIL training (the reference is single-device, src/stamp/modeling/train.py:541-547): every rank holds
    a replica and its own bags; ONE all-reduce (RCCL over xGMI on the GPU box, 14.7 MB fp32 for the default `vit` head) averages the flat
    gradient buffer before the optimiser step.  With equal per-rank batch sizes and a mean-reduced loss the average of the ranks'
    gradients IS the gradient of the global batch, so N ranks x (B/N) bags train exactly like one rank with B bags."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad /= dist.get_world_size()
    return flat_grad


def average_buffers(values: torch.Tensor) -> torch.Tensor:
    """Replicas see different bags, so the ALiBi `_RunningMeanScaler` buffers (updated from each rank's own distance matrix) would drift
    apart: average them right after the update, before use (mean of the ranks' means = the mean over the global batch for equal
    per-rank batches)."""
    return average_gradients(values)


def max_over_ranks(ctx: DistCtx, value: float) -> float:
    if ctx.world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=ctx.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def slide_tile_counts(ctx: DistCtx, jobs, *, tile_size_um: float = 256.0, tile_size_px: int = 224, max_supertile_size_slide_px: int = 2 ** 10,
                      brightness_cutoff: int | None = 240, default_slide_mpp: float | None = None) -> list[int]:
    """Foreground tile count of every slide of the job list -- the cost LPT sharding balances (SURVEY.md 8e "Partitioning"): the number of supertiles that
    survive the thumbnail's brightness cut (`tiling.foreground_coords`, reference tiling.py:250-277) x tiles per supertile.  Rank r looks at slides
    r, r + world, ... (a thumbnail each: host I/O) and ONE control-plane all-reduce of the int64 counts (disjoint supports) makes the list complete on every rank; a slide
    that cannot be opened counts 0 (it fails again, and is logged, when its owner extracts it)."""
    from . import tiling
    counts = torch.zeros(len(jobs), dtype=torch.int64)
    for i in range(ctx.rank, len(jobs), ctx.world):
        job = jobs[i]
        try:
            slide = job.slide() if isinstance(job.slide, type) or (callable(job.slide) and not hasattr(job.slide, "read_region")) else job.slide
            mpp = job.slide_mpp if job.slide_mpp is not None else tiling.get_slide_mpp(slide, default_mpp=default_slide_mpp)     # (tiling.py:409-446)
            geo = tiling.supertile_geometry(mpp, tile_size_um, tile_size_px, max_supertile_size_slide_px)
            dims = tuple(int(v) for v in slide.dimensions)
            gw, gh = tiling.thumbnail_size(dims, geo.supertile_size_slide_px)
            n = len(tiling.foreground_coords(dims, slide.get_thumbnail((2 * gw, 2 * gh)), geo.supertile_size_slide_px, brightness_cutoff))
            counts[i] = n * geo.tiles_per_side ** 2
            if slide is not job.slide and hasattr(slide, "close"):
                slide.close()
        except Exception:
            counts[i] = 0
    if ctx.world > 1:
        counts = counts.to(ctx.device)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)          # disjoint supports: the sum IS the gathered list
        counts = counts.cpu()
    return [int(v) for v in counts]


def extract_slides_sharded(ctx: DistCtx, jobs, extractor, *, runner=None, tile_counts: list[int] | None = None, **kw):
    """A node's extraction job (BASELINE.json configs[3]): every rank takes its LPT share of the slide list and runs `preprocess.extract_slides` on it --
    no data-path collective; slide-level features are gathered afterwards, once, by `gather_slide_embeddings`.  Returns (the global indices this rank
    owned, its per-slide results in that order).  `runner(jobs, extractor, **kw) -> list[dict]` replaces `extract_slides` (tests on CPU ranks)."""
    if runner is None:
        from .preprocess import extract_slides as runner
    geo_kw = {k: kw[k] for k in ("tile_size_um", "tile_size_px", "max_supertile_size_slide_px", "brightness_cutoff", "default_slide_mpp") if k in kw}
    counts = tile_counts if tile_counts is not None else slide_tile_counts(ctx, jobs, **geo_kw)
    mine = shard_slides(counts, ctx.world)[ctx.rank]
    # largest first inside the share as well: the pipeline's tail (the last, partly filled encoder chunk) then belongs to a small slide
    mine_run = sorted(mine, key=lambda i: (-counts[i], i))
    res = runner([jobs[i] for i in mine_run], extractor, **kw)
    by_idx = dict(zip(mine_run, res))
    return mine, [by_idx[i] for i in mine]
