"""Cross-check of the ViT oracle (timm semantics restated from knowledge; timm itself is absent) against an
independent third-party implementation that is installed: HF transformers.Dinov2Model (ViT with CLS token,
pos-embed on CLS+patches, pre-LN blocks, LayerScale, exact GELU).  CPU only."""
import pytest
import torch

from oracle.vit_tile_encoder import tile_transform, vit_tokens
from stamp_amd.vit import ViTConfig

transformers = pytest.importorskip("transformers")


def _hf_to_timm(hf_sd, depth):
    sd = {"patch_embed.proj.weight": hf_sd["embeddings.patch_embeddings.projection.weight"],
          "patch_embed.proj.bias": hf_sd["embeddings.patch_embeddings.projection.bias"],
          "cls_token": hf_sd["embeddings.cls_token"], "pos_embed": hf_sd["embeddings.position_embeddings"],
          "norm.weight": hf_sd["layernorm.weight"], "norm.bias": hf_sd["layernorm.bias"]}
    for i in range(depth):
        h, t = f"encoder.layer.{i}.", f"blocks.{i}."
        for n in ("norm1", "norm2"):
            sd[t + n + ".weight"], sd[t + n + ".bias"] = hf_sd[h + n + ".weight"], hf_sd[h + n + ".bias"]
        a = h + "attention.attention."
        sd[t + "attn.qkv.weight"] = torch.cat([hf_sd[a + "query.weight"], hf_sd[a + "key.weight"], hf_sd[a + "value.weight"]])
        sd[t + "attn.qkv.bias"] = torch.cat([hf_sd[a + "query.bias"], hf_sd[a + "key.bias"], hf_sd[a + "value.bias"]])
        sd[t + "attn.proj.weight"], sd[t + "attn.proj.bias"] = hf_sd[h + "attention.output.dense.weight"], hf_sd[h + "attention.output.dense.bias"]
        sd[t + "ls1.gamma"], sd[t + "ls2.gamma"] = hf_sd[h + "layer_scale1.lambda1"], hf_sd[h + "layer_scale2.lambda1"]
        sd[t + "mlp.fc1.weight"], sd[t + "mlp.fc1.bias"] = hf_sd[h + "mlp.fc1.weight"], hf_sd[h + "mlp.fc1.bias"]
        sd[t + "mlp.fc2.weight"], sd[t + "mlp.fc2.bias"] = hf_sd[h + "mlp.fc2.weight"], hf_sd[h + "mlp.fc2.bias"]
    return sd


def test_oracle_matches_hf_dinov2():
    from transformers import Dinov2Config, Dinov2Model

    torch.manual_seed(0)
    hc = Dinov2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, mlp_ratio=2, image_size=224,
                      patch_size=14, layerscale_value=0.7, layer_norm_eps=1e-6, hidden_act="gelu",
                      use_swiglu_ffn=False, qkv_bias=True)
    model = Dinov2Model(hc).eval()
    with torch.no_grad():      # HF zero-inits several tensors; make every parameter matter
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    cfg = ViTConfig(dim=128, depth=3, heads=2, hidden=256, mlp="gelu", layerscale=True, ln_eps=1e-6)
    sd = _hf_to_timm({k: v.detach() for k, v in model.state_dict().items()}, 3)
    tiles = torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    x = tile_transform(tiles, cfg.mean, cfg.std)
    with torch.no_grad():
        ref = model(pixel_values=x).last_hidden_state
        got = vit_tokens(x, sd, cfg)
    assert got.shape == ref.shape == (2, 257, 128)
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)
