"""The barspoon head (`EncDecTransformer`) on the HIP path: one C call (amds_barspoon_forward) against the fixture made by the reference's own
class and against the oracle (pinned to that fixture) at the defaults' geometry."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import barspoon as ob
from stamp_amd.barspoon import EncDecTransformer

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


@pytest.mark.parametrize("tag", ["a", "b"])
def test_barspoon_matches_reference_fixture(gpu, tag):
    """Stated tolerance: the tile side runs on fp16 MFMA operands with fp32 accumulation (like the MIL `vit` head: 5e-3 of the logit scale); the
    class-token side is exact fp32."""
    z = np.load(G / "barspoon.npz")
    sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_w:")}
    hp = [int(v) for v in z[f"{tag}_hparams"]]
    targets = {str(t): int(n) for t, n in zip(z[f"{tag}_targets"], z[f"{tag}_nout"])}
    model = EncDecTransformer(z[f"{tag}_x"].shape[2], targets, d_model=hp[0], num_encoder_heads=hp[1], num_decoder_heads=hp[2], num_encoder_layers=hp[3],
                              num_decoder_layers=hp[4], dim_feedforward=hp[5], positional_encoding=bool(hp[6])).eval()
    model.load_state_dict(sd, strict=True)                      # identical keys to the reference
    x, pos = torch.from_numpy(z[f"{tag}_x"]).to(gpu), torch.from_numpy(z[f"{tag}_pos"]).to(gpu)
    with torch.no_grad():
        out = model(x, pos)
        again = model(x.half(), pos)                            # fp16 bags as stored on disk (the fixture's values are fp16-exact)
    assert list(out) == list(targets)
    for j, t in enumerate(targets):
        ref = z[f"{tag}_logits_{j}"]
        assert out[t].shape == ref.shape and out[t].dtype == torch.float32
        err = np.abs(out[t].cpu().numpy() - ref).max()
        assert err < 5e-3 * max(1.0, np.abs(ref).max()), (t, err, ref)
        assert torch.equal(out[t], again[t])


def test_barspoon_default_geometry_vs_oracle_and_guards(gpu):
    torch.manual_seed(11)
    targets = {"A": 2, "B-1": 5}
    model = EncDecTransformer(768, targets).eval()              # d_model 512, 8 + 8 heads, 2 + 2 layers, dim_feedforward 2048
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    Bb, T = 2, 2500
    x = torch.randn(Bb, T, 768).half().float()
    pos = torch.rand(Bb, T, 2) * 50000.0
    ref = ob.barspoon_forward(x, pos, {k: v.detach() for k, v in model.state_dict().items()}, list(targets))
    with torch.no_grad():
        out = model(x.to(gpu), pos.to(gpu))
    for t in targets:
        err = (out[t].cpu() - ref[t]).abs().max().item()
        assert err < 5e-3 * max(1.0, ref[t].abs().max().item()), (t, err)
    with torch.no_grad():
        model.heads["A"].bias.add_(1.0)                         # the cached device weights follow the parameters
        out2 = model(x.to(gpu), pos.to(gpu))
    assert torch.allclose(out2["A"], out["A"] + 1.0, atol=1e-5) and torch.equal(out2["B-1"], out["B-1"])
    with pytest.raises(NotImplementedError, match="deploy / validation"):
        model.train()(x.to(gpu), pos.to(gpu))
    model.eval()
    with pytest.raises(RuntimeError, match="GPU"):
        with torch.no_grad():
            model(x, pos)
    with pytest.raises(ValueError, match="tile_positions"):
        with torch.no_grad():
            model(x.to(gpu), pos[:, :-1].to(gpu))
