"""Where does the tile encoder's fp16-operand error come from?  A torch (CPU or GPU, fp32 matmuls) emulation of the HIP path's rounding
sites, each of which can be switched on alone.  Used to choose what the opt-in `exact` mode of HipViT compensates.

  python tools/rounding_budget.py [preset] [tiles] [weight seed] [tile seed] [quick]

Sites (what the kernels round to the 16-bit act dtype; everything else is fp32 in the kernels too):
  a_qkv / a_fc1   the un-normalised residual row copy that qkv / fc1 read as their A operand (LayerNorm folded: statistics stay fp32)
  w_qkv / w_fc1   W * gamma         w_proj / w_fc2   W
  qkv             q, k, v as stored by the qkv epilogue        p   the softmax numerators fed to the P.V MFMAs
  o               attention output (A operand of proj)         u   MLP hidden activation (A operand of fc2)
  w_patch         the patch-embedding weight with the tile transform folded in (W / std, applied to the raw 0..255 values)
"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd.vit import PRESETS, random_vit_state_dict  # noqa: E402

SITES = ("w_patch", "a_qkv", "w_qkv", "qkv", "p", "o", "w_proj", "a_fc1", "w_fc1", "u", "w_fc2")


def _q8(t: torch.Tensor, dim: int) -> torch.Tensor:
    """e4m3 with one scale per slice along `dim` (rows of an activation, output channels of a weight): amds_quantize_rows_e4m3's arithmetic"""
    amax = t.abs().amax(dim=dim, keepdim=True)
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (t / s).to(torch.float8_e4m3fn).float() * s


def forward(tiles, sd, cfg, on: set, dt=torch.float16, cls_exact: bool = False, dev="cpu", fp8: bool = False):
    """fp8=True: the four GEMMs of every block take e4m3 operands (activations scaled per row, weights per output channel -- what
    amds_gemm_fp8 multiplies), everything else as the fp16 path."""
    def r(t, s):
        if s not in on:
            return t
        if fp8 and s in ("a_qkv", "a_fc1", "o", "u"):
            return _q8(t, -1)
        if fp8 and s in ("w_qkv", "w_fc1", "w_proj", "w_fc2"):
            return _q8(t, 1)
        return t.to(dt).float()
    D, p, H = cfg.dim, cfg.patch, cfg.heads
    hd = D // H
    B = tiles.shape[0]
    if "w_patch" in on:      # as packed: conv(W / std rounded, raw u8) / 255 + (b - sum W mean / std)
        mean, std = torch.tensor(cfg.mean, device=dev, dtype=torch.float64), torch.tensor(cfg.std, device=dev, dtype=torch.float64)
        pw = sd["patch_embed.proj.weight"].double()
        wf = r((pw / std.view(1, 3, 1, 1)).float(), "w_patch")
        bf = (sd["patch_embed.proj.bias"].double() - (pw * (mean / std).view(1, 3, 1, 1)).sum(dim=(1, 2, 3))).float()
        x = F.conv2d(tiles.permute(0, 3, 1, 2).float(), wf, None, stride=p) * (1.0 / 255.0) + bf.view(1, -1, 1, 1)
    else:
        x = tiles.permute(0, 3, 1, 2).float() / 255.0
        x = (x - torch.tensor(cfg.mean, device=dev).view(1, 3, 1, 1)) / torch.tensor(cfg.std, device=dev).view(1, 3, 1, 1)
        x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=p)
    x = x.flatten(2).transpose(1, 2)
    pos = sd["pos_embed"].reshape(1, -1, D)
    cat = [sd["cls_token"].reshape(1, 1, D).expand(B, -1, -1)]
    if cfg.reg_tokens:
        cat.append(sd["reg_token"].reshape(1, cfg.reg_tokens, D).expand(B, -1, -1))
    x = torch.cat(cat + [x + pos], 1) if cfg.no_embed_class else torch.cat(cat + [x], 1) + pos

    def folded_linear(x, w, b, gamma, beta, sa, sw):
        """Linear(LayerNorm(x)) the way the folded kernels do it: A = round(x), W' = round(W gamma), statistics in fp32."""
        mean = x.mean(-1, keepdim=True)
        rstd = (x.var(-1, unbiased=False, keepdim=True) + cfg.ln_eps).rsqrt()
        if fp8:      # no fold: the NORMALISED rows are quantised (LayerNorm -> quantise -> GEMM)
            h = (x - mean) * rstd * gamma + beta
            y = F.linear(r(h, sa), r(w, sw), b)
            if cls_exact:
                y[:, 0] = F.linear(h[:, 0], w, b)
            return y
        wf = r(w * gamma[None, :], sw)
        y = (F.linear(r(x, sa), wf) - mean * wf.sum(1)) * rstd + (b + w @ beta)
        if cls_exact:
            wf = w * gamma[None, :]
            y[:, 0] = (F.linear(x[:, 0], wf) - mean[:, 0] * wf.sum(1)) * rstd[:, 0] + (b + w @ beta)
        return y

    def lin(a, w, b, sa, sw):
        y = F.linear(r(a, sa), r(w, sw), b)
        if cls_exact:
            y[:, 0] = F.linear(a[:, 0], w, b)
        return y

    for i in range(cfg.depth):
        g = lambda n: sd[f"blocks.{i}.{n}"]  # noqa: E731
        qkv = folded_linear(x, g("attn.qkv.weight"), g("attn.qkv.bias"), g("norm1.weight"), g("norm1.bias"), "a_qkv", "w_qkv")
        qkv_r = r(qkv, "qkv").reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv_r[0], qkv_r[1], qkv_r[2]
        s = (q @ k.transpose(-2, -1)) * hd ** -0.5
        e = torch.exp(s - s.amax(-1, keepdim=True))
        att = (r(e, "p") @ v) / e.sum(-1, keepdim=True)
        if cls_exact:      # the class token's own query row, probabilities and output un-rounded (keys / values as stored)
            q0 = qkv.reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)[0][:, :, :1]
            s0 = (q0 @ k.transpose(-2, -1)) * hd ** -0.5
            att[:, :, :1] = torch.softmax(s0, -1) @ v
        att = att.transpose(1, 2).reshape(B, -1, D)
        y = lin(att, g("attn.proj.weight"), g("attn.proj.bias"), "o", "w_proj")
        x = x + (y * g("ls1.gamma") if cfg.layerscale else y)
        h = folded_linear(x, g("mlp.fc1.weight"), g("mlp.fc1.bias"), g("norm2.weight"), g("norm2.bias"), "a_fc1", "w_fc1")
        if cfg.mlp == "swiglu":
            x1, x2 = h.chunk(2, -1)
            h = F.silu(x1) * x2
        else:
            h = F.gelu(h)
        y = lin(h, g("mlp.fc2.weight"), g("mlp.fc2.bias"), "u", "w_fc2")
        x = x + (y * g("ls2.gamma") if cfg.layerscale else y)
    return F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], cfg.ln_eps)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "uni2_h"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    tseed = int(sys.argv[4]) if len(sys.argv) > 4 else seed + 1
    quick = len(sys.argv) > 5
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    cfg = PRESETS[name]
    sd = {k: v.float().to(dev) for k, v in random_vit_state_dict(cfg, seed=seed).items()}
    tiles = torch.randint(0, 256, (n, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(tseed)).to(dev)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    with torch.no_grad():
        ref = forward(tiles, sd, cfg, set(), dev=dev)

        def report(label, on, **kw):
            t = forward(tiles, sd, cfg, on, dev=dev, **kw)
            cls16 = rel(t[:, 0].half().float(), ref[:, 0].half().float())
            print(f"{label:28s} tokens {rel(t, ref):.3e}   cls row (fp32) {rel(t[:, 0], ref[:, 0]):.3e}   cls row (fp16 both) {cls16:.3e}", flush=True)

        print(f"{name}: {n} tiles, seed {seed}, device {dev}")
        report("all sites (round 2's path)", set(SITES))
        report("all but w_patch (the default path)", set(SITES) - {"w_patch"})
        report("same, class-token rows exact", set(SITES) - {"w_patch"}, cls_exact=True)
        if quick:
            report("fp8 GEMM operands (e4m3)", set(SITES) - {"w_patch"}, fp8=True)
            return
        for s in SITES:
            report("only " + s, {s})
        report("all but weights", set(SITES) - {"w_qkv", "w_proj", "w_fc1", "w_fc2"})
        report("all but activations a_*", set(SITES) - {"a_qkv", "a_fc1"})
        report("all but o, u", set(SITES) - {"o", "u"})
        report("all but a_*, o, u", set(SITES) - {"a_qkv", "a_fc1", "o", "u"})
        report("all but qkv, p", set(SITES) - {"qkv", "p"})
        report("fp8 GEMM operands (e4m3)", set(SITES) - {"w_patch"}, fp8=True)


if __name__ == "__main__":
    main()
