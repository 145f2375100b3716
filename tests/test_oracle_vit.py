"""Cross-check of the ViT oracle (timm semantics restated from knowledge; timm itself is absent) against independent
third-party implementations that ARE installed: HF transformers `Dinov2Model` (CLS token, pos-embed on CLS+patches, pre-LN
blocks, LayerScale, exact GELU or SwiGLU FFN) and `Dinov2WithRegistersModel` (register tokens inserted behind the class
token after the position embedding).  Every branch the reference's extractors reach is covered: GELU Mlp (ViT-L/14: RedDino,
UNI), SwiGLUPacked (UNI2-h uni2.py:17-31, Virchow2 virchow2.py:34-39, H-optimus h_optimus_0.py:15-20), register tokens with
`no_embed_class=True` (UNI2-h, H-optimus: position embedding on patches only) and `no_embed_class=False` (Virchow2: one
position row per prefix token), head_dim 80 (Virchow2), and the full-size ViT-L/14 trunk.  CPU only."""
import pytest
import torch

from oracle.vit_tile_encoder import tile_transform, vit_tokens
from stamp_amd.vit import ViTConfig

transformers = pytest.importorskip("transformers")


def _hf_to_timm(hf_sd, depth, *, swiglu: bool, regs: int, no_embed_class: bool):
    """HF Dinov2(+Registers) state_dict -> timm VisionTransformer names.  Only re-labelling, except for the position
    embedding, where the two libraries place the class-token row differently (this is what timm's own checkpoint filter
    does for the reg4_dinov2 models): HF adds pos[0] to the class token and pos[1:] to the patches BEFORE inserting the
    registers.  timm `no_embed_class=True`: pos_embed covers patches only -> fold pos[0] into cls_token.  timm
    `no_embed_class=False` with registers: pos_embed has 1 + regs + n_patches rows added AFTER the concat -> register
    rows are zero."""
    pos = hf_sd["embeddings.position_embeddings"]
    cls = hf_sd["embeddings.cls_token"]
    sd = {"patch_embed.proj.weight": hf_sd["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.proj.bias": hf_sd["embeddings.patch_embeddings.projection.bias"],
          "norm.weight": hf_sd["layernorm.weight"], "norm.bias": hf_sd["layernorm.bias"]}
    if regs:
        sd["reg_token"] = hf_sd["embeddings.register_tokens"]
    if no_embed_class:
        sd["cls_token"], sd["pos_embed"] = cls + pos[:, :1], pos[:, 1:]
    elif regs:
        sd["cls_token"] = cls
        sd["pos_embed"] = torch.cat([pos[:, :1], pos.new_zeros(1, regs, pos.shape[-1]), pos[:, 1:]], dim=1)
    else:
        sd["cls_token"], sd["pos_embed"] = cls, pos
    for i in range(depth):
        h, t = f"encoder.layer.{i}.", f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[t + n + ".weight"], sd[t + n + ".bias"] = hf_sd[h + n + ".weight"], hf_sd[h + n + ".bias"]
        a = h + "attention.attention."
        sd[t + "attn.qkv.weight"] = torch.cat([hf_sd[a + "query.weight"], hf_sd[a + "key.weight"], hf_sd[a + "value.weight"]])
        sd[t + "attn.qkv.bias"] = torch.cat([hf_sd[a + "query.bias"], hf_sd[a + "key.bias"], hf_sd[a + "value.bias"]])
        sd[t + "attn.proj.weight"], sd[t + "attn.proj.bias"] = hf_sd[h + "attention.output.dense.weight"], hf_sd[h + "attention.output.dense.bias"]
        sd[t + "ls1.gamma"], sd[t + "ls2.gamma"] = hf_sd[h + "layer_scale1.lambda1"], hf_sd[h + "layer_scale2.lambda1"]
        f1, f2 = ("mlp.weights_in", "mlp.weights_out") if swiglu else ("mlp.fc1", "mlp.fc2")
        sd[t + "mlp.fc1.weight"], sd[t + "mlp.fc1.bias"] = hf_sd[h + f1 + ".weight"], hf_sd[h + f1 + ".bias"]
        sd[t + "mlp.fc2.weight"], sd[t + "mlp.fc2.bias"] = hf_sd[h + f2 + ".weight"], hf_sd[h + f2 + ".bias"]
    return sd


def _hf_model(dim, depth, heads, mlp_ratio, swiglu, regs, seed=0, perturb=0.05):
    torch.manual_seed(seed)
    kw = dict(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, mlp_ratio=mlp_ratio, image_size=224,
              patch_size=14, layerscale_value=0.7, layer_norm_eps=1e-6, hidden_act="gelu", use_swiglu_ffn=swiglu, qkv_bias=True)
    if regs:
        from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
        model = Dinov2WithRegistersModel(Dinov2WithRegistersConfig(num_register_tokens=regs, **kw)).eval()
    else:
        from transformers import Dinov2Config, Dinov2Model
        model = Dinov2Model(Dinov2Config(**kw)).eval()
    with torch.no_grad():      # HF zero-inits several tensors (biases, registers); make every parameter matter
        for p in model.parameters():
            p.add_(perturb * torch.randn_like(p))
    return model


def _swiglu_hidden(dim, mlp_ratio):      # HF Dinov2SwiGLUFFN and timm's SwiGLUPacked users size the FFN the same way
    return (int(int(dim * mlp_ratio) * 2 / 3) + 7) // 8 * 8


CASES = {
    # name: (dim, depth, heads, mlp_ratio, swiglu, regs, no_embed_class)
    "gelu": (128, 3, 2, 2, False, 0, False),                     # ViT-L/14 family (RedDino, UNI)
    "swiglu": (128, 3, 2, 4, True, 0, False),                    # Virchow v1 branch
    "swiglu_reg8_noembedclass": (192, 2, 3, 4, True, 8, True),   # UNI2-h / H-optimus branch
    "swiglu_reg4_embedclass_hd80": (160, 2, 2, 4, True, 4, False),   # Virchow2 branch, head_dim 80
    "gelu_reg4_noembedclass": (128, 2, 2, 2, False, 4, True),    # timm vit_*_reg4_dinov2 with the plain Mlp
}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_hf_dinov2(name):
    dim, depth, heads, mlp_ratio, swiglu, regs, nec = CASES[name]
    model = _hf_model(dim, depth, heads, mlp_ratio, swiglu, regs)
    hidden = _swiglu_hidden(dim, mlp_ratio) if swiglu else dim * mlp_ratio
    cfg = ViTConfig(dim=dim, depth=depth, heads=heads, hidden=hidden, mlp="swiglu" if swiglu else "gelu", reg_tokens=regs,
                    no_embed_class=nec, layerscale=True, ln_eps=1e-6)
    sd = _hf_to_timm({k: v.detach() for k, v in model.state_dict().items()}, depth, swiglu=swiglu, regs=regs, no_embed_class=nec)
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    x = tile_transform(tiles, cfg.mean, cfg.std)
    with torch.no_grad():
        ref = model(pixel_values=x).last_hidden_state
        got = vit_tokens(x, sd, cfg)
    assert got.shape == ref.shape == (2, 257 + regs, dim)
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)


def test_oracle_matches_hf_dinov2_full_size_vit_large():
    """The headline trunk at its real size: ViT-L/14, 24 blocks, 16 heads, 257 tokens, 303 M parameters, one tile."""
    model = _hf_model(1024, 24, 16, 4, False, 0, seed=3, perturb=0.02)
    cfg = ViTConfig()       # vit_large_patch14_224
    sd = _hf_to_timm({k: v.detach() for k, v in model.state_dict().items()}, 24, swiglu=False, regs=0, no_embed_class=False)
    tiles = torch.randint(0, 256, (1, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
    x = tile_transform(tiles, cfg.mean, cfg.std)
    with torch.no_grad():
        ref = model(pixel_values=x).last_hidden_state
        got = vit_tokens(x, sd, cfg)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert got.shape == ref.shape == (1, 257, 1024) and rel < 2e-5, rel


def test_oracle_sdpa_path_equals_explicit_attention():
    """bench.py times the oracle with F.scaled_dot_product_attention; it must be the same function as the explicit form the parity
    tests use."""
    from stamp_amd.vit import PRESETS, random_vit_state_dict
    from oracle.vit_tile_encoder import extract_features

    cfg = PRESETS["test_tiny_swiglu"]
    sd = random_vit_state_dict(cfg, seed=2, init="moderate")
    tiles = torch.randint(0, 256, (3, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    _, a = extract_features(tiles, sd, cfg, return_tokens=True)
    _, b = extract_features(tiles, sd, cfg, return_tokens=True, sdpa=True)
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def _hf_vit_to_timm(hf_sd, depth):
    """HF `ViTModel` (Google ViT: class token + one learned position row per token, pre-LN blocks, NO LayerScale, GELU MLP -- the
    architecture of timm `vit_large_patch16_224` behind the reference's uni.py:26-31 / keep.py:29-37) -> timm names.  Re-labelling only."""
    sd = {"patch_embed.proj.weight": hf_sd["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.proj.bias": hf_sd["embeddings.patch_embeddings.projection.bias"],
          "cls_token": hf_sd["embeddings.cls_token"], "pos_embed": hf_sd["embeddings.position_embeddings"],
          "norm.weight": hf_sd["layernorm.weight"], "norm.bias": hf_sd["layernorm.bias"]}
    for i in range(depth):
        h, t = f"layers.{i}.", f"blocks.{i}."
        sd[t + "norm1.weight"], sd[t + "norm1.bias"] = hf_sd[h + "layernorm_before.weight"], hf_sd[h + "layernorm_before.bias"]
        sd[t + "norm2.weight"], sd[t + "norm2.bias"] = hf_sd[h + "layernorm_after.weight"], hf_sd[h + "layernorm_after.bias"]
        a = h + "attention."
        sd[t + "attn.qkv.weight"] = torch.cat([hf_sd[a + "q_proj.weight"], hf_sd[a + "k_proj.weight"], hf_sd[a + "v_proj.weight"]])
        sd[t + "attn.qkv.bias"] = torch.cat([hf_sd[a + "q_proj.bias"], hf_sd[a + "k_proj.bias"], hf_sd[a + "v_proj.bias"]])
        sd[t + "attn.proj.weight"], sd[t + "attn.proj.bias"] = hf_sd[a + "o_proj.weight"], hf_sd[a + "o_proj.bias"]
        for n in ("fc1", "fc2"):
            sd[t + f"mlp.{n}.weight"], sd[t + f"mlp.{n}.bias"] = hf_sd[h + f"mlp.{n}.weight"], hf_sd[h + f"mlp.{n}.bias"]
    return sd


@pytest.mark.parametrize("shape", [(128, 3, 2, 256), (192, 2, 3, 768)])
def test_oracle_matches_hf_vit(shape):
    """A SECOND independent witness for the oracle, on the branch Dinov2 cannot reach: patch 16 (196 + 1 tokens) and no LayerScale -- HF `ViTModel`
    (transformers' Google-ViT implementation; timm's `vit_large_patch16_224` of uni.py:26-31 / keep.py:29-37 is this architecture)."""
    from transformers import ViTConfig as HFViTConfig
    from transformers import ViTModel

    dim, depth, heads, hidden = shape
    torch.manual_seed(11)
    model = ViTModel(HFViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=hidden, image_size=224, patch_size=16,
                                 hidden_act="gelu", layer_norm_eps=1e-6, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
                     add_pooling_layer=False).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    hf_sd = {k: v.detach() for k, v in model.state_dict().items()}
    if "layers.0.attention.q_proj.weight" not in hf_sd:
        pytest.skip("this transformers version names ViTModel's parameters differently")
    cfg = ViTConfig(patch=16, dim=dim, depth=depth, heads=heads, hidden=hidden, mlp="gelu", reg_tokens=0, no_embed_class=False, layerscale=False, ln_eps=1e-6)
    sd = _hf_vit_to_timm(hf_sd, depth)
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(12))
    x = tile_transform(tiles, cfg.mean, cfg.std)
    with torch.no_grad():
        ref = model(pixel_values=x).last_hidden_state
        got = vit_tokens(x, sd, cfg)
    assert got.shape == ref.shape == (2, 197, dim)
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)
