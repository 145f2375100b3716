"""CPU restatement of the vision tower the reference's PLIP extractor runs -- TEST INFRASTRUCTURE ONLY.

Reference call site: src/stamp/preprocessing/extractor/plip.py:16-22 (`PLIP.forward` = `CLIPModel.get_image_features(batch)`), :25-36 (the transform:
Resize(224), ToTensor, Normalize with CLIP's constants).  The arithmetic lives in the third-party `transformers` package (HF `CLIPVisionTransformer`):
restated here from its published structure -- patch convolution without bias, class embedding, learned position embedding, `pre_layrnorm`, pre-norm
encoder layers with separate q / k / v / out projections and the `quick_gelu` MLP (x * sigmoid(1.702 x)), `post_layernorm` on the class token,
`visual_projection` (no bias).  **Pinned** against the installed `transformers` itself (tests/golden/plip.npz from tools/make_golden.py::golden_plip,
a randomly initialised `CLIPModel` at a small width).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)              # plip.py:31-32
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_image_features(pixel_values: torch.Tensor, sd: dict, *, heads: int, eps: float = 1e-5) -> torch.Tensor:
    """pixel_values [B, 3, H, W] (normalised) -> image embeddings [B, projection_dim]."""
    sd = {k: v.to(pixel_values.dtype) for k, v in sd.items()}
    p = "vision_model."
    w = sd[p + "embeddings.patch_embedding.weight"]
    D, patch = w.shape[0], w.shape[-1]
    x = F.conv2d(pixel_values, w, stride=patch).flatten(2).transpose(1, 2)
    B = x.shape[0]
    x = torch.cat([sd[p + "embeddings.class_embedding"].expand(B, 1, D), x], dim=1) + sd[p + "embeddings.position_embedding.weight"]
    ln = lambda t, n: F.layer_norm(t, (D,), sd[p + n + ".weight"], sd[p + n + ".bias"], eps)  # noqa: E731
    x = ln(x, "pre_layrnorm")
    hd = D // heads
    l = 0
    while f"{p}encoder.layers.{l}.layer_norm1.weight" in sd:
        q_ = f"encoder.layers.{l}."
        h = ln(x, q_ + "layer_norm1")
        lin = lambda t, n: F.linear(t, sd[p + q_ + n + ".weight"], sd[p + q_ + n + ".bias"])  # noqa: E731
        q, k, v = (lin(h, f"self_attn.{n}_proj").view(B, -1, heads, hd).transpose(1, 2) for n in ("q", "k", "v"))
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
        x = x + lin(a.transpose(1, 2).reshape(B, -1, D), "self_attn.out_proj")
        u = lin(ln(x, q_ + "layer_norm2"), "mlp.fc1")
        x = x + lin(u * torch.sigmoid(1.702 * u), "mlp.fc2")
        l += 1
    return F.linear(ln(x[:, 0], "post_layernorm"), sd["visual_projection.weight"])


def tiles_to_pixels(tiles_u8: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """u8 [B, H, W, 3] -> what the reference's transform hands the model: ToTensor + Normalize (Resize(224) is the identity on 224-pixel tiles)."""
    x = tiles_u8.permute(0, 3, 1, 2).to(dtype) / 255.0
    return (x - torch.tensor(CLIP_MEAN, dtype=dtype).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD, dtype=dtype).view(1, 3, 1, 1)
