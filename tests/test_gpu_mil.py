"""MIL `vit` head forward on the HIP path vs the oracle (which is pinned to the reference's goldens)."""
import pytest
import torch

from oracle.mil_vit import mil_vit_forward
from stamp_amd.mil import VisionTransformer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Bb,T,F,C", [(2, 1024, 1024, 2), (3, 77, 768, 3), (1, 3000, 1024, 2), (2, 1, 512, 4)])
def test_mil_vit_forward_matches_oracle(gpu, Bb, T, F, C):
    torch.manual_seed(Bb * 1000 + T)
    model = VisionTransformer(dim_output=C, dim_input=F, dim_model=512, n_layers=2, n_heads=8, dim_feedforward=512,
                              dropout=0.0, use_alibi=False).eval()
    with torch.no_grad():          # make biases / norms non-trivial
        for n, p in model.named_parameters():
            if p.dim() == 1 and "class_token" not in n:
                p.add_(0.1 * torch.randn_like(p))
    sd = model.state_dict()
    bags = torch.randn(Bb, T, F).half()              # features are fp16 on disk (reference preprocessing/__init__.py:325)
    coords = torch.rand(Bb, T, 2) * 1000
    ref = mil_vit_forward(bags.float(), coords, None, sd, n_heads=8, use_alibi=False)
    with torch.no_grad():
        out = model(bags.to(gpu), coords=coords.to(gpu), mask=None)
    assert out.shape == (Bb, C) and out.dtype == torch.float32
    err = (out.cpu() - ref).abs().max().item()
    assert err < 5e-3 * max(1.0, ref.abs().max().item()), (err, ref)
    with torch.no_grad():
        assert torch.equal(out, model(bags.to(gpu), coords=coords.to(gpu), mask=None))   # deterministic


def test_mil_vit_state_dict_roundtrip_and_guards(gpu):
    kw = dict(dim_output=2, dim_input=256, dim_model=128, n_layers=1, n_heads=2, dim_feedforward=128, dropout=0.0, use_alibi=False)
    m1, m2 = VisionTransformer(**kw).eval(), VisionTransformer(**kw).eval()
    m2.load_state_dict(m1.state_dict())
    assert set(m1.state_dict()) == {"class_token", "project_features.0.weight", "project_features.0.bias",
                                    "transformer.layers.0.0.norm.weight", "transformer.layers.0.0.norm.bias",
                                    "transformer.layers.0.0.mhsa.in_proj_weight", "transformer.layers.0.0.mhsa.in_proj_bias",
                                    "transformer.layers.0.0.mhsa.out_proj.weight", "transformer.layers.0.0.mhsa.out_proj.bias",
                                    "transformer.layers.0.1.0.weight", "transformer.layers.0.1.0.bias",
                                    "transformer.layers.0.1.1.weight", "transformer.layers.0.1.1.bias",
                                    "transformer.layers.0.1.4.weight", "transformer.layers.0.1.4.bias",
                                    "transformer.norm.weight", "transformer.norm.bias", "mlp_head.0.weight", "mlp_head.0.bias"}
    bags = torch.randn(2, 50, 256).to(gpu)           # fp32 bags are accepted too (cast on the device)
    with torch.no_grad():
        assert torch.equal(m1(bags, coords=None, mask=None), m2(bags, coords=None, mask=None))
    with pytest.raises(NotImplementedError):
        m1(bags, coords=None, mask=torch.zeros(2, 50, dtype=torch.bool, device=gpu))
    with pytest.raises(NotImplementedError):
        VisionTransformer(**{**kw, "use_alibi": True})
    with pytest.raises(NotImplementedError):
        m1.train()(bags, coords=None, mask=None)
    with pytest.raises(RuntimeError, match="GPU"):
        with torch.no_grad():
            m1.eval()(bags.cpu(), coords=None, mask=None)
