import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_qkv_attn import _case
from stamp_amd import _lib, ops
dev = torch.device("cuda:0")
for (B, H) in [(5, 4), (3, 16), (33, 8)]:
    x, w, bias, rowstat, colsum = [t.to(dev) for t in _case(B, H, torch.float16, 100 + B + H, True)]
    qkv = ops.gemm_lnfold(x, w, _lib.EPI_BIAS, bias=bias, rowstat=rowstat, colsum=colsum)
    want = ops.attention_vit(qkv, B, 257, H)
    for rep in range(2):
        got = ops.qkv_attention_vit257(x, w, bias, B, H, rowstat=rowstat, colsum=colsum)
        torch.cuda.synchronize()
        bad = ~torch.isfinite(got.float())
        g = got.float().view(B, 257, H, 64); wv = want.float().view(B, 257, H, 64)
        err = (g - wv).abs().amax(-1)          # [B, 257, H]
        err = torch.nan_to_num(err, nan=9.0)
        idx = (err > 0.02).nonzero()
        print(f"B={B} H={H} rep {rep}: nonfinite {int(bad.sum())}, bad (tile, token, head) count {len(idx)}; first: {idx[:12].tolist()}")
        if len(idx):
            print("   tokens:", sorted(set(idx[:, 1].tolist()))[:40], "tiles:", sorted(set(idx[:, 0].tolist()))[:20], "heads:", sorted(set(idx[:, 2].tolist())))
