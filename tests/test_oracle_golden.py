"""Pin the CPU oracle against golden vectors captured from the reference's own code (tools/make_golden.py),
plus the known answers in the reference's docstrings.  CPU only."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import gated_attention, mil_vit, misc, transmil

G = Path(__file__).parent / "golden"


def _load(name):
    z = np.load(G / name)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    return z, sd


@pytest.mark.parametrize("tag", ["small", "xs"])
def test_chief_gated_attention(tag):
    z, sd = _load(f"chief_gated_attention_{tag}.npz")
    out = gated_attention.gated_attention_pool(torch.from_numpy(z["x"]), sd)
    np.testing.assert_allclose(out["attention_raw"].numpy(), z["attention_raw"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["WSI_feature"].numpy(), z["wsi_feature"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,alibi", [("plain", False), ("alibi", True)])
def test_mil_vit(tag, alibi):
    z, sd = _load(f"mil_vit_{tag}.npz")
    heads = int(z["hparams"][4])
    bags, coords, mask = (torch.from_numpy(z[k]) for k in ("bags", "coords", "mask"))
    y = mil_vit.mil_vit_forward(bags, coords, None, sd, n_heads=heads, use_alibi=alibi)
    np.testing.assert_allclose(y.numpy(), z["logits_nomask"], rtol=2e-5, atol=2e-5)
    y = mil_vit.mil_vit_forward(bags, coords, mask, sd, n_heads=heads, use_alibi=alibi)
    np.testing.assert_allclose(y.numpy(), z["logits_mask"], rtol=2e-5, atol=2e-5)


def test_alibi_running_mean_was_updated_in_golden():
    z = np.load(G / "mil_vit_alibi.npz")
    k = [f for f in z.files if f.startswith("w_after_train:") and "items_so_far" in f]
    assert k and all(z[f][0] == 3.0 for f in k)       # 1 + two train-mode forwards


@pytest.mark.parametrize("tag", ["t50", "t300"])
def test_transmil(tag):
    z, sd = _load(f"transmil_{tag}.npz")
    y = transmil.transmil_forward(torch.from_numpy(z["bags"]), sd)
    np.testing.assert_allclose(y.numpy(), z["logits"], rtol=1e-4, atol=1e-4)
    out = transmil.nystrom_attention(torch.from_numpy(z["nys_x"]), sd, "layer1.attn.", heads=8, landmarks=int(z["hparams"][2]) // 2)
    np.testing.assert_allclose(out.numpy(), z["nys_out"], rtol=1e-4, atol=1e-5)
    p = transmil.pinv_iter(torch.from_numpy(z["pinv_in"]), 6)
    np.testing.assert_allclose(p.numpy(), z["pinv_out"], rtol=1e-4, atol=1e-4)


def test_ppeg():
    z, sd = _load("transmil_ppeg.npz")
    sd = {"pos_layer." + k: v for k, v in sd.items()}
    y = transmil.ppeg(torch.from_numpy(z["x"]), sd, "pos_layer.", 8, 8)
    np.testing.assert_allclose(y.numpy(), z["out"], rtol=1e-5, atol=1e-5)


def test_mlp_linear():
    z = np.load(G / "mlp.npz")
    sm = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp:")}
    sl = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("lin:")}
    for key, x in (("3", torch.from_numpy(z["x3"])), ("2", torch.from_numpy(z["x2"]))):
        np.testing.assert_allclose(misc.mlp_forward(x, sm, 3).numpy(), z[f"mlp_y{key}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(misc.linear_forward(x, sl).numpy(), z[f"lin_y{key}"], rtol=1e-5, atol=1e-6)


def test_cox_known_answers_and_golden():
    # the reference's own docstring values (src/stamp/modeling/models/cox.py:192-204)
    log_hz = torch.tensor([0.1, 0.2, 0.3, 0.4, 0.5])
    event = torch.tensor([1, 0, 1, 0, 1], dtype=torch.bool)
    time = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0])
    assert abs(misc.cox_neg_partial_log_likelihood(log_hz, time, event).item() - 1.0071) < 1e-4
    assert abs(misc.cox_neg_partial_log_likelihood(log_hz, time, event, reduction="sum").item() - 3.0214) < 1e-4
    tie = torch.tensor([1.0, 2.0, 2.0, 4.0, 5.0])
    assert abs(misc.cox_neg_partial_log_likelihood(log_hz, tie, event, "efron").item() - 1.0873) < 1e-4
    assert abs(misc.cox_neg_partial_log_likelihood(log_hz, tie, event, "breslow").item() - 1.0873) < 1e-4
    z = np.load(G / "cox.npz")
    for k in ("doc_mean", "doc_sum", "doc_tie_efron", "doc_tie_breslow"):
        assert np.isfinite(z[k])
    lh, tt, ev = torch.from_numpy(z["log_hz"]), torch.from_numpy(z["time"]), torch.from_numpy(z["event"])
    assert abs(misc.cox_neg_partial_log_likelihood(lh, tt, ev, "efron").item() - float(z["efron"])) < 1e-5
    assert abs(misc.cox_neg_partial_log_likelihood(lh, tt, ev, "breslow").item() - float(z["breslow"])) < 1e-5
    assert abs(misc.cox_neg_partial_log_likelihood(lh, torch.arange(40.0), ev).item() - float(z["notie"])) < 1e-5


def test_cox_breslow_slide_loss_golden():
    """Slide / patient-level survival objective (`cox_loss`, models/__init__.py:625-659): the oracle's restatement AND the product's
    torch loss (stamp_amd.losses, K14: a few hundred scalars stay torch) against the reference function's value and gradient."""
    from stamp_amd import losses
    z = np.load(G / "cox_slide.npz")
    times, events = torch.from_numpy(z["times"]), torch.from_numpy(z["events"])
    for fn in (misc.cox_breslow_slide_loss, losses.cox_breslow_loss):
        sc = torch.from_numpy(z["scores"]).clone().requires_grad_(True)
        loss = fn(sc, times, events)
        assert abs(loss.item() - float(z["loss"])) < 1e-6
        g, = torch.autograd.grad(loss, sc)
        np.testing.assert_allclose(g.numpy(), z["grad"], rtol=1e-5, atol=1e-8)
        s0 = torch.randn(5, requires_grad=True)
        l0 = fn(s0, torch.arange(5.0), torch.zeros(5))
        assert l0.item() == 0.0 and l0.requires_grad == bool(z["no_event_requires_grad"])
    tg = torch.stack([times, events], 1)
    assert abs(losses.cox_slide_survival_loss(torch.from_numpy(z["scores"]), tg).item() - float(z["loss"])) < 1e-6


@pytest.mark.parametrize("name,tdt", [("f32", torch.float32), ("f16", torch.float16), ("bf16", torch.bfloat16)])
def test_vary_precision_bit_exact(name, tdt):
    z = np.load(G / "vary_precision.npz")
    bits_in, bits_out = z[f"in_{name}"], z[f"out_{name}"]
    torch.manual_seed(22)       # same global-generator draw as the reference made
    shifts = torch.randint(0, misc.vary_precision_shift_range(name, 2), bits_in.shape).numpy()
    assert np.array_equal(misc.vary_precision_bits(bits_in, shifts), bits_out)


def test_fixed_size_bag():
    z = np.load(G / "fixed_size_bag.npz")
    for n, bs in ((10, 16), (100, 16), (16, 16), (1000, 512), (1, 4)):
        bag, coords = torch.from_numpy(z[f"bag_{n}_{bs}"]), torch.from_numpy(z[f"coords_{n}_{bs}"])
        b, c, l = misc.to_fixed_size_bag(bag, coords, bs, deterministic=True)
        assert np.array_equal(b.numpy(), z[f"det_bag_{n}_{bs}"]) and np.array_equal(c.numpy(), z[f"det_coords_{n}_{bs}"])
        assert l == int(z[f"det_len_{n}_{bs}"])
        torch.manual_seed(1234)
        b, c, l = misc.to_fixed_size_bag(bag, coords, bs, deterministic=False)
        assert np.array_equal(b.numpy(), z[f"rand_bag_{n}_{bs}"]) and l == int(z[f"rand_len_{n}_{bs}"])


@pytest.mark.parametrize("tag,preset", [("tiny", "test_swin_tiny"), ("swin_t", "ctranspath")])
def test_ctranspath_swin(tag, preset):
    """The Swin/ConvStem restatement against the reference's own `_SwinTransformer` (ctranspath.py), stage by stage.
    The weights are the seeded `random_swin_state_dict` that tools/make_golden.py loaded into the reference module
    with strict=True (so names and shapes are the reference's)."""
    from oracle import swin_ctranspath
    from stamp_amd.swin import SWIN_PRESETS, random_swin_state_dict

    z = np.load(G / f"ctranspath_{tag}.npz")
    cfg = SWIN_PRESETS[preset]
    sd = random_swin_state_dict(cfg, int(z["seed"]))
    taps = {}
    feats = swin_ctranspath.swin_features(torch.from_numpy(z["tiles"]), sd, cfg, taps=taps)
    for k, v in taps.items():
        np.testing.assert_allclose(v[:, z["tap_idx_" + k]].numpy(), z["tap_" + k], rtol=0, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(feats.numpy(), z["feats"], rtol=0, atol=5e-6)
    assert feats.shape == (z["tiles"].shape[0], cfg.out_dim)


def test_texture_gray_matches_pillow_golden():
    """`tile.convert("L")` (tiling.py:284): fixtures produced by the installed Pillow in tools/make_golden.py."""
    from oracle import texture
    z = np.load(G / "texture_gray.npz")
    assert np.array_equal(texture.gray_L(z["tiles"]), z["gray"])


def test_texture_canny_known_answers():
    """The Canny restatement has no reference vectors (OpenCV is absent: parity unpinned); these are cases whose answer
    follows from the published algorithm by hand."""
    from oracle import texture
    S = 32
    flat = np.full((S, S), 77, np.uint8)
    assert texture.canny_l1(flat).sum() == 0
    step = np.zeros((S, S), np.uint8)
    step[:, 16:] = 255                      # |dx| = 1020 on columns 15 and 16; NMS (m > left, m >= right) keeps column 15 only
    e = texture.canny_l1(step)
    assert (e[:, 15] == 255).all() and e.sum() == 255 * S
    weak = np.zeros((S, S), np.uint8)
    weak[:, 16:] = 20                       # |dx| = 80: above low (40), below high (100): candidates without a seed -> nothing
    assert texture.canny_l1(weak).sum() == 0
    mixed = weak.copy()
    mixed[:8, 16:] = 255                    # a strong segment on top: hysteresis walks down the connected weak edge
    e = texture.canny_l1(mixed)
    assert (e[:, 15] == 255).sum() >= S - 2 and e[:, 15][-1] == 255
    rgb = np.stack([step] * 3, -1)
    assert abs(texture.edge_fraction(rgb) - 1 / S) < 1e-12 and texture.has_enough_texture(rgb, 0.02)


def _train_fixture(tag):
    z = np.load(G / f"mil_vit_train_{tag}.npz")
    sd0 = {k[len("w_before:"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_before:")}
    sd1 = {k[len("w_after:"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w_after:")}
    grads = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("g:")}
    L = int(z["hparams"][3])
    mult = lambda name: torch.from_numpy(z["mask:" + name]).float() / (1.0 - float(z["rate:" + name]))  # noqa: E731
    drop = {"proj": mult("project_features.2")}
    for l in range(L):
        drop[f"ff1_{l}"], drop[f"ff2_{l}"] = mult(f"transformer.layers.{l}.1.3"), mult(f"transformer.layers.{l}.1.5")
    return z, sd0, sd1, grads, drop


@pytest.mark.parametrize("tag,alibi", [("plain", False), ("alibi", True)])
def test_mil_vit_train_mode_matches_reference(tag, alibi):
    """TRAIN mode of the reference module (dropout sites + ALiBi scaler update), pinned by what the reference itself produced:
    with the keep masks its nn.Dropout modules drew, the oracle reproduces its logits, loss and every parameter gradient."""
    z, sd0, sd1, grads, drop = _train_fixture(tag)
    heads = int(z["hparams"][4])
    assert float(z["rate:transformer.layers.0.1.3"]) == 0.5 and float(z["rate:transformer.layers.0.1.5"]) == 0.5     # hard-coded (:160)
    assert abs(float(z["rate:project_features.2"]) - float(z["dropout"])) < 1e-7
    bags, coords = torch.from_numpy(z["bags"]), torch.from_numpy(z["coords"])
    sd = dict(sd0)
    if alibi:       # every scaler folds the batch's distances in BEFORE use
        cc = torch.cat([coords.new_zeros(3, 1, 2), coords], dim=1)
        dist = torch.cdist(cc, cc)
        for k in sd0:
            if k.endswith("running_mean"):
                n = k[: -len("running_mean")] + "items_so_far"
                sd[k], sd[n] = mil_vit.running_mean_update(sd0[k], sd0[n], dist)
                np.testing.assert_allclose(sd[k].numpy(), sd1[k].numpy(), rtol=1e-6)
                np.testing.assert_allclose(sd[n].numpy(), sd1[n].numpy())
    params = {k: v.clone().requires_grad_(k in grads) for k, v in sd.items()}
    logits = mil_vit.mil_vit_forward(bags, coords, None, params, n_heads=heads, use_alibi=alibi, drop=drop)
    np.testing.assert_allclose(logits.detach().numpy(), z["logits"], rtol=2e-5, atol=2e-5)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(z["targets"]), weight=torch.from_numpy(z["class_weights"]))
    np.testing.assert_allclose(loss.item(), float(z["loss"]), rtol=1e-5)
    loss.backward()
    for k, g in grads.items():
        np.testing.assert_allclose(params[k].grad.numpy(), g.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_eagle_slide_embedding_golden(tag):
    """oracle/eagle.py against the outputs of the reference's own `Eagle._generate_slide_embedding` (tools/make_golden.py::golden_eagle)."""
    from oracle import eagle
    z, sd = _load("eagle.npz")
    x, agg = torch.from_numpy(z[f"{tag}_x"]), torch.from_numpy(z[f"{tag}_agg"])
    emb, top = eagle.eagle_slide_embedding(x, agg, sd)
    assert np.array_equal(top, z[f"{tag}_top"]) and len(top) == min(25, x.shape[0])
    np.testing.assert_allclose(emb, z[f"{tag}_emb"], rtol=1e-6, atol=1e-7)
    assert emb.dtype == np.float32 and emb.shape == (agg.shape[1],)
    pe = eagle.eagle_patient_embedding([x[:5], x[5:]], [agg[:5], agg[5:]], sd)                 # a patient = its slides concatenated
    np.testing.assert_allclose(pe, z[f"{tag}_emb"], rtol=1e-6, atol=1e-7)


def test_eagle_coordinate_alignment_golden():
    from oracle import eagle
    from stamp_amd.encoder import align_by_coords
    z = np.load(G / "eagle.npz")
    # the fixture's `other` file holds the reference tiles in shuffled order (+ sub-rounding jitter, one duplicated coordinate); its feature rows
    # carry their own position in the ORIGINAL order, so the reference's aligned features read 0..59 except where duplicates may swap
    for fn in (eagle.align_by_coords, align_by_coords):            # the oracle and the product's host-side twin
        perm = fn(z["al_ref"], z["al_other"], 5)
        assert np.array_equal(z["al_other"][perm], z["al_coords"])
        assert sorted(perm.tolist()) == list(range(60)) and np.array_equal(z["al_ids"][perm], z["al_rows"])      # the same rows as the reference picked
        with pytest.raises(ValueError, match="Missing coord"):
            fn(z["al_ref"], z["al_other"][:-1], 5)
        with pytest.raises(ValueError, match="extra coords"):
            fn(z["al_ref"][:-1], z["al_other"], 5)
    assert np.array_equal(np.sort(z["al_rows"]), np.arange(60))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_barspoon_golden(tag):
    """oracle/barspoon.py against logits of the reference's own `EncDecTransformer` (tools/make_golden.py::golden_barspoon)."""
    from oracle import barspoon
    z = np.load(G / "barspoon.npz")
    sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_w:")}
    hp = z[f"{tag}_hparams"]
    targets = [str(t) for t in z[f"{tag}_targets"]]
    out = barspoon.barspoon_forward(torch.from_numpy(z[f"{tag}_x"]), torch.from_numpy(z[f"{tag}_pos"]), sd, targets, num_encoder_heads=int(hp[1]),
                                    num_decoder_heads=int(hp[2]), positional_encoding=bool(hp[6]))
    assert list(out) == targets
    for j, t in enumerate(targets):
        assert out[t].shape == (z[f"{tag}_x"].shape[0], int(z[f"{tag}_nout"][j]))
        np.testing.assert_allclose(out[t].numpy(), z[f"{tag}_logits_{j}"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("key", ["hoptimus1", "conchv15"])
def test_ticon_tile_golden(key):
    """oracle/ticon.py against outputs of the reference's own TICON `EncoderDecoder` called as its extractor calls it (one token per tile)."""
    from oracle import ticon
    z, sd = _load("ticon.npz")
    y = ticon.ticon_tile_forward(torch.from_numpy(z[f"emb_{key}"]), sd, key)
    np.testing.assert_allclose(y.numpy(), z[f"out_{key}"], rtol=1e-6, atol=1e-6)


def test_keep_image_head_golden():
    z, sd = _load("keep_head.npz")
    out = misc.keep_image_head(torch.from_numpy(z["feats"]), sd)
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.linalg.norm(out.numpy(), axis=1), 1.0, rtol=1e-5)


def test_plip_clip_vision_golden():
    """oracle/clip_vision.py against the installed `transformers` CLIPModel.get_image_features (the call of the reference's plip.py:16-22)."""
    from oracle import clip_vision as cv
    z, sd = _load("plip.npz")
    y = cv.clip_image_features(cv.tiles_to_pixels(torch.from_numpy(z["tiles"])), sd, heads=int(z["heads"]))
    np.testing.assert_allclose(y.numpy(), z["image_features"], rtol=1e-5, atol=2e-6)


_DINO_VARIANTS = {"gelu": dict(hidden=256), "swiglu": dict(hidden=344, mlp="swiglu"), "reg4": dict(hidden=344, mlp="swiglu", reg_tokens=4)}


@pytest.mark.parametrize("tag", list(_DINO_VARIANTS))
def test_vit_oracle_equals_transformers_dinov2(tag):
    """The tile-encoder oracle (a restatement of timm's VisionTransformer, which is NOT installed here) against an independent third-party
    implementation of the same architecture that IS: `transformers`' Dinov2Model / Dinov2WithRegistersModel -- GELU MLP + LayerScale (the reference's
    RedDino / DinoBloom / ViT-L/14 structure), SwiGLU, SwiGLU + 4 register tokens (its ViT-g family's).  Fixture: tools/make_golden.py::golden_dinov2_hf --
    the committed outputs the GPU test holds the HIP path to (tests/test_gpu_vit.py); the LIVE comparison on every branch and at full ViT-L/14 size is
    tests/test_oracle_vit.py."""
    from oracle.vit_tile_encoder import extract_features
    from stamp_amd.vit import ViTConfig, hf_dinov2_to_timm_names
    z = np.load(G / "dinov2_hf.npz")
    sd = hf_dinov2_to_timm_names({k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_w:")})
    cfg = ViTConfig(dim=128, depth=2, heads=2, **_DINO_VARIANTS[tag])
    feats, toks = extract_features(torch.from_numpy(z["tiles"]), sd, cfg, return_tokens=True)
    sel = list(range(10)) + [-2, -1]
    np.testing.assert_allclose(toks[:, sel].numpy(), z[f"{tag}_tokens"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(toks.norm(dim=-1).numpy(), z[f"{tag}_token_norms"], rtol=1e-5)
    np.testing.assert_allclose(feats.float().numpy(), z[f"{tag}_tokens"][:, 0], rtol=2e-3, atol=2e-3)       # the stored feature = fp16 of the class row
