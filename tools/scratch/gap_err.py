import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle.gated_attention import KEYS, gated_attention_pool
from stamp_amd import ops
gpu = torch.device('cuda:0')
def ref64(x, sd):
    w = {k: sd[v].double() for k, v in KEYS.items()}
    x = x.double()
    h = torch.relu(x @ w['fc_w'].T + w['fc_b'])
    a = torch.tanh(h @ w['a_w'].T + w['a_b']); b = torch.sigmoid(h @ w['b_w'].T + w['b_b'])
    A = (a * b) @ w['c_w'].T + w['c_b']
    P = torch.softmax(A.T, dim=1)
    return (P @ x).reshape(-1), A.reshape(-1)
for (F, L, D) in ((768, 512, 256), (384, 256, 256)):
    g = torch.Generator().manual_seed(F)
    sd = {KEYS["fc_w"]: torch.randn(L, F, generator=g) / F ** 0.5, KEYS["fc_b"]: torch.randn(L, generator=g) * 0.1,
          KEYS["a_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["a_b"]: torch.randn(D, generator=g) * 0.1,
          KEYS["b_w"]: torch.randn(D, L, generator=g) / L ** 0.5, KEYS["b_b"]: torch.randn(D, generator=g) * 0.1,
          KEYS["c_w"]: torch.randn(1, D, generator=g) * 2, KEYS["c_b"]: torch.randn(1, generator=g)}
    lens = [1, 64, 2, 63, 65, 128, 300, 1, 1024, 129, 5000, 7, 191, 640]
    xs = [torch.randn(n, F, generator=g) * (1 + (i % 3)) for i, n in enumerate(lens)]
    w = {k: sd[v].to(gpu).contiguous() for k, v in KEYS.items()}
    for i, x in enumerate(xs):
        r64, a64 = ref64(x, sd)
        ro = gated_attention_pool(x, sd)
        of, af = ops.gated_attn_pool(x.to(gpu), w, return_attn=True)
        ou, au = ops.gated_attn_pool(x.to(gpu), w, return_attn=True, fused=False)
        e = lambda t, r: float((t.double().cpu() - r).abs().max())
        print(F, 'N', len(x), 'scale', 1 + i % 3, 'out err vs f64: oracle %.2e fused %.2e six %.2e | araw: oracle %.2e fused %.2e six %.2e | max|A| %.1f' % (
            e(ro['WSI_feature'].reshape(-1), r64), e(of, r64), e(ou, r64), e(ro['attention_raw'].reshape(-1), a64), e(af, a64), e(au, a64), float(a64.abs().max())))
