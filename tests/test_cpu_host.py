"""CPU-side tests: the C-ABI library loads and exports every declared symbol; host logic; sharding; the
world_size-2 gloo path of the slide-embedding collation."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from stamp_amd import _lib

    lib = _lib.lib()
    header = (ROOT / "include" / "amdstamp.h").read_text()
    declared = set(re.findall(r"\b(amds_[a-z0-9_]+)\s*\(", header))
    declared -= {"amds_status", "amds_dtype", "amds_epilogue"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/amdstamp.h but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.amds_version() == 2          # major * 100 + minor (include/amdstamp.h)


def test_ops_refuse_cpu_tensors():
    from stamp_amd import ops

    with pytest.raises(RuntimeError, match="GPU"):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8), 1e-6)
    from stamp_amd.vit import PRESETS, HipViT, random_vit_state_dict

    with pytest.raises(RuntimeError, match="GPU only"):
        HipViT(PRESETS["test_tiny"], random_vit_state_dict(PRESETS["test_tiny"]), device="cpu")


def test_product_never_imports_oracle():
    for p in (ROOT / "stamp_amd").rglob("*.py"):
        src = p.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{p} imports the oracle"


def test_flops_per_tile_matches_survey():
    from stamp_amd.vit import PRESETS

    assert abs(PRESETS["vit_large_patch14_224"].matmul_flops_per_tile() / 1e9 - 162.02) < 0.01
    assert abs(PRESETS["uni2_h"].matmul_flops_per_tile() / 1e9 - 370.94) < 0.01
    assert abs(PRESETS["virchow2"].matmul_flops_per_tile() / 1e9 - 340.13) < 0.01
    # what the class-row tail of the last block never computes: q / proj / fc1 / fc2 rows and the attention of T - 1 tokens
    c = PRESETS["vit_large_patch14_224"]
    T, D, H = c.tokens, c.dim, c.hidden
    assert c.matmul_flops_skipped_by_cls_tail() == 2 * (T - 1) * (2 * D * D + 2 * D * H) + 4 * T * (T - 1) * D
    assert abs(c.matmul_flops_skipped_by_cls_tail() / 1e9 - 5.638) < 0.001
    assert abs((PRESETS["uni2_h"].matmul_flops_per_tile() - PRESETS["uni2_h"].matmul_flops_skipped_by_cls_tail()) / 1e9 - 358.054) < 0.01


def test_shard_slides_lpt():
    from stamp_amd.distributed import shard_slides

    counts = [20000, 100, 15000, 15000, 300, 9000, 9000, 50]
    shards = shard_slides(counts, 4)
    assert sorted(i for s in shards for i in s) == list(range(len(counts)))
    loads = [sum(counts[i] for i in s) for s in shards]
    assert max(loads) == 20000 and min(loads) >= 15000
    assert shard_slides(counts, 4) == shards                      # deterministic
    assert shard_slides([], 3) == [[], [], []]
    assert shard_slides([5], 2) == [[0], []]


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["REPO"])
from stamp_amd import distributed as D
ctx = D.init_from_env(prefer_gpu=False)
counts = [7, 3, 9, 1, 4]
mine = D.shard_slides(counts, ctx.world)[ctx.rank]
emb = torch.stack([torch.full((6,), float(i + 1)) for i in mine]) if mine else torch.zeros(0, 6)
table = D.gather_slide_embeddings(ctx, emb, torch.tensor(mine, dtype=torch.int64), len(counts))
expect = torch.stack([torch.full((6,), float(i + 1)) for i in range(len(counts))])
assert torch.equal(table, expect), table
t = D.max_over_ranks(ctx, float(ctx.rank + 1))
assert t == float(ctx.world)
D.barrier(ctx)
print("rank", ctx.rank, "ok")
"""


def test_gather_slide_embeddings_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


_DP_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["REPO"])
from oracle.mil_vit import mil_vit_forward
from stamp_amd import distributed as D
from stamp_amd.mil import VisionTransformer
ctx = D.init_from_env(prefer_gpu=False)
torch.manual_seed(0)                                   # every rank builds the same replica and the same GLOBAL batch
model = VisionTransformer(dim_output=2, dim_input=32, dim_model=64, n_layers=1, n_heads=2, dim_feedforward=64, dropout=0.0, use_alibi=False)
names = [n for n, _ in model.named_parameters()]
bags, targets = torch.randn(8, 20, 32), torch.nn.functional.one_hot(torch.arange(8) % 2, 2).float()

def flat_grad(b, t):                                   # the oracle network stands in for the HIP step on this CPU-only host
    p = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    loss = torch.nn.functional.cross_entropy(mil_vit_forward(b, torch.zeros(b.shape[0], 20, 2), None, p, n_heads=2, use_alibi=False), t)
    loss.backward()
    return torch.cat([p[n].grad.reshape(-1) for n in names]), loss.detach()

full, full_loss = flat_grad(bags, targets)             # one rank, whole batch
per = 8 // ctx.world
mine = slice(ctx.rank * per, (ctx.rank + 1) * per)
g, l = flat_grad(bags[mine], targets[mine])            # this rank's shard
g = D.average_gradients(g)
l = D.average_gradients(l.reshape(1))
assert torch.allclose(g, full, rtol=1e-5, atol=1e-7), (g - full).abs().max()
assert abs(l.item() - full_loss.item()) < 1e-6
stats = D.average_buffers(torch.tensor([float(ctx.rank + 1), 10.0 * (ctx.rank + 1)]))
assert torch.allclose(stats, torch.tensor([1.5, 15.0])), stats
D.barrier(ctx)
print("rank", ctx.rank, "ok")
"""


def test_data_parallel_mil_gradients_equal_single_rank_gloo_world2(tmp_path):
    """N-GPU DP-MIL must equal 1-GPU with the same global batch.  The reduction recipe the trainer uses
    (stamp_amd.distributed.average_gradients / average_buffers: one all-reduce of the flat gradient buffer) on world-size-2 gloo."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, REPO=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


_MLP_WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["REPO"])
from oracle import misc
from stamp_amd import distributed as D
from stamp_amd.mil import MLP
ctx = D.init_from_env(prefer_gpu=False)
n_slides, dim = 11, 24
counts = [900, 40, 700, 700, 65, 300, 20, 1000, 5, 410, 77]

def embed(i):                                          # stands in for encode_slides_ on this rank's slides: a pure function of the slide
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(dim, generator=g)

mine = D.shard_slides(counts, ctx.world)[ctx.rank]
local = torch.stack([embed(i) for i in mine]) if mine else torch.zeros(0, dim)
table = D.gather_slide_embeddings(ctx, local, torch.tensor(mine, dtype=torch.int64), n_slides)
single = torch.stack([embed(i) for i in range(n_slides)])          # what ONE rank owning every slide would hold
assert torch.equal(table, single), (table - single).abs().max()
seen = torch.zeros(n_slides, dtype=torch.int64)
seen[torch.tensor(mine, dtype=torch.int64)] = 1
torch.distributed.all_reduce(seen)
assert bool((seen == 1).all()), seen                     # every slide id landed in the table exactly once

# the patient-level consumer of the table (LitSlide* / LitPatient* + MLP, models/__init__.py:778-937): one AdamW step on the gathered table.
# The head's parameter tree is the product's (stamp_amd.mil.MLP); its arithmetic on this CPU-only host is the pinned oracle's.
torch.manual_seed(5)
head = MLP(dim_input=dim, dim_hidden=16, dim_output=2, num_layers=3, dropout=0.0)
opt = torch.optim.AdamW(head.parameters(), lr=1e-2)
targets = torch.nn.functional.one_hot(torch.arange(n_slides) % 2, 2).float()

def step(x):
    opt.zero_grad()
    sd = dict(head.state_dict(keep_vars=True))
    loss = torch.nn.functional.cross_entropy(misc.mlp_forward(x, sd, 3), targets)
    loss.backward()
    opt.step()
    return loss.detach()

loss = step(table)
flat = torch.cat([p.detach().reshape(-1) for p in head.parameters()])
ref = [torch.zeros_like(flat) for _ in range(ctx.world)]
torch.distributed.all_gather(ref, flat)
assert all(torch.equal(r, flat) for r in ref)           # identical weights on every rank after the step

torch.manual_seed(5)                                   # and identical to a single-rank run on the single-rank table
head1 = MLP(dim_input=dim, dim_hidden=16, dim_output=2, num_layers=3, dropout=0.0)
opt1 = torch.optim.AdamW(head1.parameters(), lr=1e-2)
opt1.zero_grad()
l1 = torch.nn.functional.cross_entropy(misc.mlp_forward(single, dict(head1.state_dict(keep_vars=True)), 3), targets)
l1.backward()
opt1.step()
assert torch.equal(torch.cat([p.detach().reshape(-1) for p in head1.parameters()]), flat) and l1.item() == loss.item()
D.barrier(ctx)
print("rank", ctx.rank, "ok")
"""


def test_shard_encode_gather_then_one_mlp_step_gloo_world2(tmp_path):
    """SURVEY.md 8e end to end on two gloo ranks: LPT shard -> per-rank slide embeddings -> ONE padded all-gather -> the table every rank
    trains the patient-level MLP head on.  The table equals the single-rank one, every slide id lands exactly once, and one optimiser step
    leaves identical weights on both ranks and on a single-rank run."""
    script = tmp_path / "mlp.py"
    script.write_text(_MLP_WORKER)
    env = dict(os.environ, REPO=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_bench_multi_rank_branch_dry_run_gloo_world2():
    """`bench.py --gpus 2` launched exactly as the driver launches it (torch.distributed.run, one process per rank, 127.0.0.1 rendezvous), on
    CPU ranks over gloo with AMDS_BENCH_DRYRUN=1: the N > 1 branch (all-gather of slide embeddings every step, barrier, max over ranks,
    rank 0 prints ONE JSON line) runs to completion and every rank's slide lands in the gathered table."""
    import json
    env = dict(os.environ, AMDS_BENCH_DRYRUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29623",
           str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["dry_run"] is True and line["value"] is None
    assert line["scaling"] == "weak" and line["metric"].startswith("tiles/sec encoded")
    # a wrong --gpus is refused before any work
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120, cwd=str(ROOT))
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stdout + r2.stderr)


def test_gather_single_rank():
    from stamp_amd.distributed import DistCtx, gather_slide_embeddings

    ctx = DistCtx(0, 1, 0, torch.device("cpu"))
    t = gather_slide_embeddings(ctx, torch.ones(2, 3), torch.tensor([2, 0]), 4)
    assert t[0].sum() == 3 and t[2].sum() == 3 and t[1].sum() == 0


def test_swin_lane_tables_reproduce_dense_bias_and_mask():
    """The MFMA-lane-ordered relative-position-bias and shifted-window-mask tables (host packing for
    amds_window_attention) must be a pure re-indexing of the dense tensors the reference builds
    (ctranspath.py:478-496, 620-645), which the oracle restates."""
    from oracle import swin_ctranspath as osw
    from stamp_amd.swin import LOG2E, _lane_key_query, rel_bias_lane_table, shift_mask_bits, shift_mask_lane_table

    table = torch.randn(169, 6, generator=torch.Generator().manual_seed(0))
    dense = table[osw.rel_pos_index().reshape(-1)].reshape(49, 49, 6).permute(2, 0, 1)       # [h][q][k]
    lane = rel_bias_lane_table(table)
    key, query = _lane_key_query()
    ok = (key < 49) & (query < 49)
    got = lane[:, ok] / LOG2E
    want = dense[:, query[ok], key[ok]]
    assert torch.allclose(got, want, atol=1e-6)
    assert (lane[:, key >= 49] <= -20000).all()
    # every (query, key) pair of the window is covered exactly once
    assert torch.unique(query[ok] * 49 + key[ok]).numel() == 49 * 49 == int(ok.sum())
    masks = shift_mask_lane_table()
    for grid in (56, 14):
        lab = osw.window_region_labels(grid, grid, 3)
        n = grid // 7
        for wh, ww in ((0, 0), (n - 1, 0), (0, n - 1), (n - 1, n - 1)):
            typ = 2 * (wh == n - 1) + (ww == n - 1)
            l = lab[wh * n + ww]
            want = l[:, None] != l[None, :]                                                     # [q][k]: reference adds -100
            assert torch.equal(masks[typ][ok], want[query[ok], key[ok]])
    assert not masks[0].any()
    # the packed words the kernel reads: bit (kt*2+qt)*16 + r of lane's 64-bit word
    bits = shift_mask_bits()
    assert bits.shape == (4, 64) and bits.dtype == torch.int64
    for typ in range(4):
        for kt in range(2):
            for qt in range(2):
                for r in (0, 5, 15):
                    got = (bits[typ] >> ((kt * 2 + qt) * 16 + r)) & 1
                    assert torch.equal(got.bool(), masks[typ, kt, qt, :, r])


def test_swin_flops_and_state_dict_shapes():
    from stamp_amd.swin import SWIN_PRESETS, random_swin_state_dict, swin_param_shapes
    cfg = SWIN_PRESETS["ctranspath"]
    assert abs(cfg.matmul_flops_per_tile() / 1e9 - 8.99) < 0.01          # Swin-T: 4.5 GMAC (the published figure)
    sd = random_swin_state_dict(cfg, 0)
    assert sum(v.numel() for v in sd.values()) == sum(int(np.prod(s)) for _, s in swin_param_shapes(cfg))
    assert cfg.out_dim == 768 and sd["layers.3.blocks.1.attn.qkv.weight"].shape == (2304, 768)


def test_product_losses_match_oracle_and_reference_known_answers():
    """stamp_amd.losses (product, differentiable torch on [batch] scalars) vs the pinned oracle and the known answers of the
    reference's own docstring (src/stamp/modeling/models/cox.py:192-204)."""
    from oracle import misc
    from stamp_amd import losses
    g = torch.Generator().manual_seed(0)
    lh = torch.randn(40, generator=g)
    ev = torch.rand(40, generator=g) < 0.6
    t_notie = torch.randperm(40, generator=g).float()
    t_tie = torch.randint(0, 8, (40,), generator=g).float()
    for t, m in ((t_notie, "efron"), (t_tie, "efron"), (t_tie, "breslow")):
        a = losses.neg_partial_log_likelihood(lh, t, ev, m)
        b = misc.cox_neg_partial_log_likelihood(lh, t, ev, m)
        assert abs(a.item() - b.item()) < 1e-5, (m, a.item(), b.item())
    z = np.load(Path(__file__).parent / "golden" / "cox.npz")
    lhz, tt, evz = torch.from_numpy(z["log_hz"]), torch.from_numpy(z["time"]), torch.from_numpy(z["event"])
    assert abs(losses.neg_partial_log_likelihood(lhz, tt, evz, "efron").item() - float(z["efron"])) < 1e-4
    # differentiable, and the survival wrapper reads [time, event] columns
    p = lh.clone().requires_grad_(True)
    losses.cox_survival_loss(p.unsqueeze(-1), torch.stack([t_tie, ev.float()], 1)).backward()
    assert torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0
    assert abs(losses.l1_loss(torch.tensor([[1.0], [3.0]]), torch.tensor([2.0, 5.0])).item() - 1.5) < 1e-7


def test_vit_state_dict_validation():
    """Loader guard: a checkpoint must describe the preset -- names and shapes of timm's VisionTransformer."""
    from stamp_amd.vit import PRESETS, expected_state_dict_shapes, random_vit_state_dict, validate_state_dict

    for name in ("vit_large_patch14_224", "uni2_h", "virchow2", "h_optimus_0", "test_tiny_swiglu"):
        cfg = PRESETS[name]
        want = expected_state_dict_shapes(cfg)
        n_par = sum(int(np.prod(s)) for s in want.values())
        if name == "vit_large_patch14_224":
            assert abs(n_par / 1e6 - 303.2) < 0.1                       # ViT-L/14 with LayerScale: 303.2 M parameters (the HF Dinov2 probe of SURVEY.md 8c)
        if name == "uni2_h":
            assert want["blocks.0.mlp.fc1.weight"] == (8192, 1536) and want["reg_token"] == (1, 8, 1536) and want["pos_embed"] == (1, 256, 1536)
        if name == "virchow2":
            assert want["pos_embed"] == (1, 261, 1280) and want["blocks.31.mlp.fc2.weight"] == (1280, 3416)
    cfg = PRESETS["test_tiny_swiglu"]
    sd = random_vit_state_dict(cfg, seed=0)
    validate_state_dict(cfg, sd)
    validate_state_dict(cfg, dict(sd, **{"head.weight": torch.zeros(3, cfg.dim), "mask_token": torch.zeros(1, cfg.dim)}))     # ignored extras
    with pytest.raises(ValueError, match="missing"):
        validate_state_dict(cfg, {k: v for k, v in sd.items() if k != "blocks.1.ls2.gamma"})
    with pytest.raises(ValueError, match="wrong shape"):
        validate_state_dict(cfg, dict(sd, pos_embed=torch.zeros(1, 257, cfg.dim)))
    with pytest.raises(ValueError, match="unexpected"):
        validate_state_dict(cfg, dict(sd, **{"blocks.0.attn.q_norm.weight": torch.ones(64)}))
    with pytest.raises(ValueError):
        validate_state_dict(PRESETS["test_tiny"], sd)                    # a SwiGLU / register-token checkpoint is not a plain ViT


def test_docs_cite_existing_symbols_and_lines():
    """INTEGRATION.md names only symbols include/amdstamp.h declares; every `file.py:line` citation in the docs, the header and the sources
    points inside the cited reference file (checked where /root/reference exists, i.e. in the build container)."""
    import re
    hdr = set(re.findall(r"\b(amds_[a-z0-9_]+)\s*\(", (ROOT / "include" / "amdstamp.h").read_text()))
    types = {"amds_ctx", "amds_vit_cfg", "amds_vit_weights", "amds_vit_block", "amds_vit_host_weights", "amds_vit_host_block", "amds_vit_exact_block",
             "amds_swin_cfg", "amds_swin_weights", "amds_gap_weights", "amds_status", "amds_dtype", "amds_epilogue", "amds_mil_vit_cfg",
             "amds_mil_vit_weights", "amds_mil_vit_layer", "amds_mil_vit_grads", "amds_mil_vit_layer_grads", "amds_mil_vit_dropout",
             "amds_transmil_cfg", "amds_transmil_weights", "amds_transmil_layer", "amds_nystrom_grads", "amds_barspoon_cfg", "amds_barspoon_weights", "amds_barspoon_dec_layer", "amds_ticon_weights", "amds_ticon_block", "amds_transmil_grads"}
    assert all(re.search(r"\}\s*" + t + r"\s*;|typedef struct " + t + r"\b|\b" + t + r"\s*\*", (ROOT / "include" / "amdstamp.h").read_text()) for t in types), \
        [t for t in types if not re.search(r"\}\s*" + t + r"\s*;|typedef struct " + t + r"\b|\b" + t + r"\s*\*", (ROOT / "include" / "amdstamp.h").read_text())]
    doc = set(re.findall(r"\b(amds_[a-z0-9_]+)", (ROOT / "INTEGRATION.md").read_text()))
    unknown = sorted(d for d in doc if d not in hdr and d not in types and not any(h.startswith(d) for h in hdr))
    assert not unknown, unknown
    ref = Path("/root/reference")
    if not ref.is_dir():
        pytest.skip("reference tree not present on this machine")
    by_name: dict = {}
    for p in ref.rglob("*.py"):
        by_name.setdefault(p.name, []).append(p)
    bad = []
    docs = [ROOT / "INTEGRATION.md", ROOT / "DESIGN.md", ROOT / "README.md", ROOT / "include" / "amdstamp.h"] + sorted((ROOT / "docs" / "rounds").glob("*.md"))
    docs += sorted((ROOT / "stamp_amd").glob("*.py")) + sorted((ROOT / "oracle").glob("*.py")) + sorted((ROOT / "stamp_amd" / "csrc").glob("*.h*"))
    for d in docs:
        for m in re.finditer(r"([A-Za-z_][\w/\.]*\.py):(\d+)(?:[-–](\d+))?((?:,\s*\d+(?:[-–]\d+)?)*)", d.read_text()):
            cands = [p for p in by_name.get(Path(m.group(1)).name, []) if str(p).endswith(m.group(1))]
            if not cands:
                continue
            nums = [int(m.group(2))] + ([int(m.group(3))] if m.group(3) else []) + [int(x) for x in re.findall(r"\d+", m.group(4) or "")]
            n = max(len(p.read_text().splitlines()) for p in cands)
            if max(nums) > n:
                bad.append((d.name, m.group(0), n))
    assert not bad, bad


def test_code_hash_and_extractor_name_rules_equal_the_references_functions(tmp_path):
    """`encoder.code_hash` = the reference's `get_processing_code_hash` (utils/cache.py:42-55) and `encoder.resolve_extractor_name` = its
    `_resolve_extractor_name` (encoding/encoder/__init__.py:232-250): where the reference tree is present (the build container) its own functions are
    executed by name on the same inputs; elsewhere the recorded answers are used."""
    import ast
    import hashlib
    import re
    from functools import cache

    from stamp_amd.encoder import code_hash, resolve_extractor_name
    names = ["chief-ctranspath-0a1b2c3d", "chief-ctranspath", "virchow2", "virchow2-deadbeef", "uni2-12345", "h-optimus-0", "h-optimus-0-abcdef12", " ctranspath-ABCDEF ",
             "mstar-xyz123", "conch1_5-00000000", "a-b-c-123456"]
    recorded = ["chief-ctranspath", "chief-ctranspath", "virchow2", "virchow2", "uni2-12345", "h-optimus-0", "h-optimus-0", "ctranspath", "mstar-xyz123", "conch1_5", "a-b-c"]
    assert [resolve_extractor_name(n) for n in names] == recorded
    with pytest.raises(ValueError):
        resolve_extractor_name("")
    for i, body in enumerate((b"print(1)\n", b"x = 2\n" * 1000, b"")):
        (tmp_path / f"m{i}.py").write_bytes(body)
    (tmp_path / "notes.txt").write_text("ignored")
    h = hashlib.sha256()
    for i, body in enumerate((b"print(1)\n", b"x = 2\n" * 1000, b"")):
        h.update(hashlib.sha256(body).digest())
    assert code_hash(tmp_path) == h.hexdigest()
    ref = Path("/root/reference/src/stamp")
    if not ref.is_dir():
        return
    if not hasattr(hashlib, "file_digest"):                     # Python 3.10: the 3.11 helper the reference uses
        def file_digest(f, algo):
            hh = hashlib.new(algo)
            for chunk in iter(lambda: f.read(1 << 20), b""):
                hh.update(chunk)
            return hh
        hashlib.file_digest = file_digest
    glb = {"hashlib": hashlib, "Path": Path, "cache": cache, "re": re}
    for path, wanted in ((ref / "utils" / "cache.py", {"get_processing_code_hash"}), (ref / "encoding" / "encoder" / "__init__.py", {"_resolve_extractor_name"})):
        tree = ast.parse(path.read_text())
        body = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in wanted) or
                (isinstance(n, ast.Assign) and any(getattr(t, "id", "") == "_HASH_RE" for t in n.targets))]
        exec(compile(ast.Module(body=body, type_ignores=[]), str(path), "exec"), glb)
    assert glb["get_processing_code_hash"](tmp_path / "m0.py") == code_hash(tmp_path)
    assert [glb["_resolve_extractor_name"](n) for n in names] == recorded


def test_gelu_polynomial_in_the_kernel_header_is_what_the_fit_script_derives():
    """stamp_amd/csrc/common.h carries the erf polynomial of the GELU epilogue as literals; tools/gelu_poly_fit.py derives them (constrained
    near-minimax fit, re-expansion in x^2, the ulp search that makes the fp32 FMA chain saturate at the clamp).  Same numbers, and the fp32
    emulation of the kernel's arithmetic meets the bounds the header states: |error| <= 1.4e-5 |x|, <= 2.5e-8 |x| beyond the clamp."""
    import re
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "tools" / "gelu_poly_fit.py")], capture_output=True, text=True, check=True).stdout
    want = [float(v.rstrip("f")) for v in re.search(r"GELU_R = \{([^}]*)\}", out).group(1).split(",")]
    hdr = (root / "stamp_amd" / "csrc" / "common.h").read_text()
    got = [float(v.strip().rstrip("f")) for v in re.search(r"GELU_R\[GELU_DEG \+ 1\] = \{([^}]*)\}", hdr, re.S).group(1).split(",")]
    assert got == want and len(got) == 9
    assert float(re.search(r"GELU_XMAX = ([0-9.]+)f", hdr).group(1)) == float(re.search(r"GELU_XMAX = ([0-9.]+)", out).group(1))
    m = re.search(r"x\^2 form.*residue at the clamp ([0-9.e+-]+);.*max error / \|x\| = ([0-9.e+-]+);", out)
    assert float(m.group(1)) < 2.5e-8 and float(m.group(2)) < 1.4e-5


_SLIDES_WORKER = r"""
import os, sys, json
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, os.environ["REPO"])
from PIL import Image
from stamp_amd import distributed as D, h5io
from stamp_amd.preprocess import SlideJob
ctx = D.init_from_env(prefer_gpu=False)
out = Path(os.environ["OUT"])

def make_slide(i):                                        # a slide object whose foreground grows with i: thumbnails only (no GPU on this host)
    w = h = 1024 * (2 + i % 5)
    class S:
        dimensions = (w, h)
        def get_thumbnail(self, size):
            im = Image.new("RGB", tuple(int(v) for v in size), "#ffffff")
            px = im.load()
            for y in range(im.size[1]):
                for x in range(max(1, im.size[0] * (i % 5 + 1) // 6)):
                    px[x, y] = (120, 60, 140)
            return im
        def read_region(self, *a):
            raise AssertionError("the stand-in runner does not read regions")
    return S

jobs = [SlideJob(make_slide(i), out / f"slide{i:02d}.h5", 0.5, f"slide{i:02d}") for i in range(11)]
jobs[4] = SlideJob(lambda: (_ for _ in ()).throw(OSError("unreadable slide")), out / "slide04.h5", 0.5, "slide04")
(out / "slide07.h5").write_bytes(b"already there") if ctx.rank == 0 else None
D.barrier(ctx)

def runner(my_jobs, extractor, **kw):                     # stands in for preprocess.extract_slides on a host without a GPU: same contract
    res = []
    for j in my_jobs:
        r = {"name": j.name}
        try:
            if Path(j.output_path).exists():
                r["status"] = "skipped"
            else:
                s = j.slide()
                n = int(s.dimensions[0]) // 1024
                feats = torch.full((n, 8), float(int(j.name[-2:])), dtype=torch.float16)
                h5io.write_tile_features(Path(j.output_path), feats, np.zeros((n, 2), np.float32), extractor="standin", tile_size_um=256.0, tile_size_px=224,
                                         code_hash="0", stamp_version="2.5.0", amdstamp_version="0")
                r["status"] = "written"
        except Exception as e:
            r["status"] = "failed"; r["error"] = repr(e)
        res.append(r)
    return res

counts = D.slide_tile_counts(ctx, jobs, brightness_cutoff=224)
assert counts[4] == 0 and all(c > 0 for i, c in enumerate(counts) if i != 4), counts
mine, res = D.extract_slides_sharded(ctx, jobs, None, runner=runner, tile_counts=counts, brightness_cutoff=224)
shares = D.shard_slides(counts, ctx.world)
assert mine == shares[ctx.rank] and sorted(sum(shares, [])) == list(range(11))
loads = [sum(counts[i] for i in sh) for sh in shares]
assert max(loads) - min(loads) <= max(counts), loads       # LPT: no rank is more than one slide behind
status = {i: r["status"] for i, r in zip(mine, res)}
for i in mine:
    assert status[i] == ("failed" if i == 4 else "skipped" if i == 7 else "written"), (i, status)
# slide encoder stand-in (mean of the tile features) on the slides this rank wrote, then the ONE data-path collective
ids = [i for i in mine if status[i] == "written"]
emb = torch.stack([torch.from_numpy(h5io.read_tile_features(out / f"slide{i:02d}.h5")[0].astype(np.float32)).mean(0) for i in ids]) if ids else torch.zeros(0, 8)
table = D.gather_slide_embeddings(ctx, emb, torch.tensor(ids, dtype=torch.int64), len(jobs))
expect = torch.stack([torch.full((8,), float(i)) if i not in (4, 7) else torch.zeros(8) for i in range(11)])
assert torch.equal(table, expect), table
D.barrier(ctx)
print("rank", ctx.rank, "ok")
"""


def test_extract_slides_sharded_gloo_world2(tmp_path):
    """The node-level job of BASELINE.json configs[3] on two CPU ranks: foreground tile counts by thumbnail (one control-plane all-reduce), LPT shares, every
    rank runs its share through the `extract_slides` contract (a stand-in runner: no GPU here) with skip-existing and per-slide failures, slide
    embeddings of the written files, ONE all-gather -- the table is the same on both ranks and complete."""
    script = tmp_path / "sl.py"
    script.write_text(_SLIDES_WORKER)
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, REPO=str(ROOT), OUT=str(out), MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o
