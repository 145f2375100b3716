"""`amds_vit_pack_host` (csrc/vit_pack.hip): the library-side weight packer of the tile encoder, checked WITHOUT a GPU against a torch / numpy
restatement of every transformation it performs -- tile transform folded into the patch embedding (single-rounded and as a 16-bit hi | lo
pair), prefix tokens + position rows for both `no_embed_class` settings, LayerNorm folded into qkv / fc1 (W * gamma re-rounded, b + W beta,
row sums of the ROUNDED weights), SwiGLU gate / value padding + 32-row block interleave, fc2 K padding, LayerScale in the exact path's rows.
Host code only: no kernel is launched."""
import ctypes as C

import numpy as np
import pytest
import torch

from stamp_amd import _lib
from stamp_amd.vit import PRESETS, ViTConfig, host_weights, random_vit_state_dict


def _arr(ptr, n, ct):
    return np.ctypeslib.as_array((ct * n).from_address(ptr)).copy() if ptr else None


def _f16(ptr, rows, cols):
    return torch.from_numpy(_arr(ptr, rows * cols, C.c_uint16).view(np.float16).reshape(rows, cols))


def _bf16(ptr, rows, cols):
    return torch.from_numpy(_arr(ptr, rows * cols, C.c_uint16).astype(np.int16).reshape(rows, cols)).view(torch.bfloat16)


def _f32(ptr, n):
    return torch.from_numpy(_arr(ptr, n, C.c_float))


def _pack(cfg, sd, flags, dt=torch.float16):
    lib = _lib.lib()
    hw, keep = host_weights(cfg, sd)
    cc = _lib.VitCfg(cfg.img, cfg.patch, cfg.dim, cfg.depth, cfg.heads, cfg.hidden_pad, cfg.n_prefix, 1 if cfg.mlp == "swiglu" else 0,
                     1 if cfg.layerscale else 0, 0 if dt == torch.float16 else 1, cfg.ln_eps)
    need = lib.amds_vit_pack_bytes(C.byref(cc), C.byref(hw), flags)
    assert need > 0 and need % 256 == 0, lib.amds_last_error()
    raw = np.zeros(need + 256, np.uint8)
    base = (raw.ctypes.data + 255) & ~255
    w, blocks, exact = _lib.VitWeights(), (_lib.VitBlock * cfg.depth)(), (_lib.VitExactBlock * cfg.depth)()
    rc = lib.amds_vit_pack_host(C.byref(cc), C.byref(hw), flags, base, need, None, C.byref(w), blocks, exact)
    assert rc == 0, lib.amds_last_error()
    return w, blocks, exact, (raw, keep), need


CFG_FOLD = ViTConfig(dim=256, depth=2, heads=4, hidden=488, mlp="swiglu", reg_tokens=4, no_embed_class=True)       # Hp = 512, fc1 rows 1024


@pytest.mark.parametrize("name,flags", [("test_tiny", 0), ("test_tiny", 2), ("test_tiny_swiglu", 2), ("test_tiny_hd80", 2 | 4), ("fold", 1 | 2), ("fold", 1 | 2 | 4), ("fold", 1 | 2 | 8), ("test_tiny_swiglu", 2 | 4 | 8),
                                        ("fold_bf16", 1 | 2)])
def test_packed_image_matches_the_torch_restatement(name, flags):
    dt = torch.bfloat16 if name.endswith("bf16") else torch.float16
    cfg = CFG_FOLD if name.startswith("fold") else PRESETS[name]
    sd = random_vit_state_dict(cfg, seed=3)
    w, blocks, exact, keep, _ = _pack(cfg, sd, flags, dt)
    rd = _f16 if dt == torch.float16 else _bf16
    D, P, np_, kp, Hp, H = cfg.dim, cfg.n_prefix, cfg.n_patches, cfg.kp, cfg.hidden_pad, cfg.hidden
    fold, split, ex, tail = bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8)
    # ---- patch embedding
    mean, std = torch.tensor(cfg.mean, dtype=torch.float64), torch.tensor(cfg.std, dtype=torch.float64)
    pw = sd["patch_embed.proj.weight"].double()
    flat = (pw / std.view(1, 3, 1, 1)).reshape(D, -1)
    kreal = flat.shape[1]
    got = rd(w.patch_w, D, 2 * kp if split else kp)
    hi = flat.float().to(dt)
    assert torch.equal(got[:, :kreal], hi) and not got[:, kreal:kp].any()
    shift = (11 if dt == torch.float16 else 8) if split else 0
    assert w.patch_lo_shift == shift
    if split:
        lo = ((flat - hi.double()) * 2.0 ** shift).float().to(dt)
        assert torch.equal(got[:, kp:kp + kreal], lo) and not got[:, kp + kreal:].any()
        # what the split buys: hi + lo 2^-shift reproduces W / std to ~2^-20 instead of 2^-11
        err1 = ((hi.double() - flat).norm() / flat.norm()).item()
        err2 = ((hi.double() + lo.double() * 2.0 ** -shift - flat).norm() / flat.norm()).item()
        assert err2 < err1 * (2e-3 if dt == torch.float16 else 2e-2)
    pb = (sd["patch_embed.proj.bias"].double() - (pw * (mean / std).view(1, 3, 1, 1)).sum(dim=(1, 2, 3))).float()
    assert torch.allclose(_f32(w.patch_b, D), pb, rtol=0, atol=1e-6)
    # ---- tokens / position rows
    pos = sd["pos_embed"].float().reshape(-1, D)
    prefix = torch.cat([sd["cls_token"].reshape(1, D)] + ([sd["reg_token"].reshape(-1, D)] if cfg.reg_tokens else [])).float()
    if cfg.no_embed_class:
        pos_patch = pos
    else:
        prefix, pos_patch = prefix + pos[:P], pos[P:]
    assert torch.equal(_f32(w.prefix, P * D).view(P, D), prefix) and torch.equal(_f32(w.pos_patch, np_ * D).view(np_, D), pos_patch)
    assert torch.equal(_f32(w.norm_w, D), sd["norm.weight"].float()) and torch.equal(_f32(w.norm_b, D), sd["norm.bias"].float())
    assert w.exact_hidden == H and bool(w.exact_host) == ex
    # ---- blocks
    for i in range(cfg.depth):
        g = lambda n: sd[f"blocks.{i}.{n}"]  # noqa: E731
        b = blocks[i]
        assert torch.equal(_f32(b.ln1_w, D), g("norm1.weight").float()) and torch.equal(_f32(b.ln2_b, D), g("norm2.bias").float())

        def folded(wt, bias, gamma, beta):
            w64 = wt.double()
            wf = (w64 * gamma.double()[None, :]).float().to(dt)
            return wf, (bias.double() + w64 @ beta.double()).float(), wf.double().sum(1).float()

        if fold:
            wq, bq, cq = folded(g("attn.qkv.weight"), g("attn.qkv.bias"), g("norm1.weight"), g("norm1.bias"))
            assert torch.equal(rd(b.qkv_w, 3 * D, D), wq)
            assert torch.allclose(_f32(b.qkv_b, 3 * D), bq, rtol=0, atol=2e-6) and torch.allclose(_f32(b.qkv_colsum, 3 * D), cq, rtol=0, atol=1e-6)
        else:
            assert torch.equal(rd(b.qkv_w, 3 * D, D), g("attn.qkv.weight").float().to(dt)) and torch.equal(_f32(b.qkv_b, 3 * D), g("attn.qkv.bias").float())
            assert not b.qkv_colsum and not b.fc1_colsum
        assert torch.equal(rd(b.proj_w, D, D), g("attn.proj.weight").float().to(dt)) and torch.equal(_f32(b.proj_b, D), g("attn.proj.bias").float())
        w1, b1 = g("mlp.fc1.weight").float(), g("mlp.fc1.bias").float()
        if cfg.mlp == "swiglu":        # pad gate / value halves to Hp, then 32-row block interleave (timm SwiGLUPacked: chunk(2) -> silu(x1) * x2)
            w1p, b1p = w1.new_zeros(2 * Hp, D), b1.new_zeros(2 * Hp)
            w1p[:H], w1p[Hp:Hp + H], b1p[:H], b1p[Hp:Hp + H] = w1[:H], w1[H:], b1[:H], b1[H:]
            perm = torch.tensor([(r % 64 >= 32) * Hp + (r // 64) * 32 + r % 32 for r in range(2 * Hp)])
        else:
            w1p, b1p, perm = w1, b1, torch.arange(H)
        n1 = w1p.shape[0]
        if fold:
            wf, bf, cf = folded(w1p, b1p, g("norm2.weight"), g("norm2.bias"))
            assert torch.equal(rd(b.fc1_w, n1, D), wf[perm])
            assert torch.allclose(_f32(b.fc1_b, n1), bf[perm], rtol=0, atol=2e-6) and torch.allclose(_f32(b.fc1_colsum, n1), cf[perm], rtol=0, atol=1e-6)
        else:
            assert torch.equal(rd(b.fc1_w, n1, D), w1p.to(dt)[perm]) and torch.equal(_f32(b.fc1_b, n1), b1p[perm])
        w2 = rd(b.fc2_w, D, Hp)
        assert torch.equal(w2[:, :H], g("mlp.fc2.weight").float().to(dt)) and not w2[:, H:].any() and torch.equal(_f32(b.fc2_b, D), g("mlp.fc2.bias").float())
        if cfg.layerscale:
            assert torch.equal(_f32(b.ls1, D), g("ls1.gamma").float()) and torch.equal(_f32(b.ls2, D), g("ls2.gamma").float())
        if ex or (tail and i == cfg.depth - 1):      # AMDS_PACK_CLS_TAIL: the last block's fp32 rows only
            e = exact[i]
            f1 = 2 * H if cfg.mlp == "swiglu" else H
            ls1, ls2 = g("ls1.gamma").float(), g("ls2.gamma").float()
            assert torch.equal(_f32(e.q_w, D * D).view(D, D), g("attn.qkv.weight").float()[:D]) and torch.equal(_f32(e.q_b, D), g("attn.qkv.bias").float()[:D])
            assert torch.equal(_f32(e.proj_w, D * D).view(D, D), g("attn.proj.weight").float() * ls1[:, None]) and torch.equal(_f32(e.proj_b, D), g("attn.proj.bias").float() * ls1)
            assert torch.equal(_f32(e.fc1_w, f1 * D).view(f1, D), w1) and torch.equal(_f32(e.fc1_b, f1), b1)
            assert torch.equal(_f32(e.fc2_w, D * H).view(D, H), g("mlp.fc2.weight").float() * ls2[:, None]) and torch.equal(_f32(e.fc2_b, D), g("mlp.fc2.bias").float() * ls2)
    # amds_vit_weights.cls_tail points at the last block's struct (HOST pointer) exactly when the flag is set
    last = C.addressof(exact[cfg.depth - 1])
    assert (C.cast(w.cls_tail, C.c_void_p).value == last) if tail else not w.cls_tail
    if tail and not ex:
        assert all(not exact[i].q_w for i in range(cfg.depth - 1))
    del keep


def test_pack_refuses_what_it_cannot_pack():
    lib = _lib.lib()
    cfg = PRESETS["test_tiny"]
    sd = random_vit_state_dict(cfg, seed=0)
    hw, keep = host_weights(cfg, sd)
    cc = _lib.VitCfg(cfg.img, cfg.patch, cfg.dim, cfg.depth, cfg.heads, cfg.hidden_pad, cfg.n_prefix, 0, 1, 0, cfg.ln_eps)
    assert lib.amds_vit_pack_bytes(C.byref(cc), C.byref(hw), 1) == 0 and b"fold" in lib.amds_last_error()         # dim 128 cannot fold
    need = lib.amds_vit_pack_bytes(C.byref(cc), C.byref(hw), 0)
    buf = np.zeros(need + 256, np.uint8)
    base = (buf.ctypes.data + 255) & ~255
    w, blocks = _lib.VitWeights(), (_lib.VitBlock * cfg.depth)()
    assert lib.amds_vit_pack_host(C.byref(cc), C.byref(hw), 0, base, need - 256, None, C.byref(w), blocks, None) == -2     # AMDS_ERR_WORKSPACE
    assert lib.amds_vit_pack_host(C.byref(cc), C.byref(hw), 4, base, need, None, C.byref(w), blocks, None) == -1           # exact without out_exact
    assert lib.amds_vit_pack_host(C.byref(cc), C.byref(hw), 8, base, need, None, C.byref(w), blocks, None) == -1           # class-row tail without out_exact
    hw.hidden = cfg.hidden + 64
    assert lib.amds_vit_pack_bytes(C.byref(cc), C.byref(hw), 0) == 0
    del keep


def test_f16_rounding_of_the_packer_is_round_to_nearest_even():
    """The packer rounds on the HOST (integer code, csrc/vit_pack.hip); torch's fp32 -> fp16 conversion is the yardstick: halfway cases, the
    subnormal range, overflow."""
    cfg = ViTConfig(dim=128, depth=1, heads=2, hidden=128, layerscale=False)
    sd = random_vit_state_dict(cfg, seed=1)
    vals = torch.cat([torch.tensor([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 65504.0, 65519.9, 65520.0, 1e6, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8, 2.9e-8, -7e-6, 0.0, -0.0]),
                      torch.randn(128 * 128 - 14) * torch.logspace(-9, 5, 128 * 128 - 14)])
    sd["blocks.0.attn.proj.weight"] = vals.reshape(128, 128).clone()
    _, blocks, _, keep, _ = _pack(cfg, sd, 0)
    got = _f16(blocks[0].proj_w, 128, 128)
    want = vals.reshape(128, 128).half()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    _, blocks_b, _, keep_b, _ = _pack(cfg, sd, 0, torch.bfloat16)
    assert torch.equal(_bf16(blocks_b[0].proj_w, 128, 128).view(torch.int16), vals.reshape(128, 128).bfloat16().view(torch.int16))
