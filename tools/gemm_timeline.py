"""Per-segment s_memtime timeline of the 8-phase GEMM (block 0, all 8 waves). Debug tool."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib
lib = _lib.lib(); f = lib.amds_gemm_ablate; f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
M, N, K = 65536, 1024, 4096
A = torch.randn(M, K, generator=g).to(dev, torch.float16); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, torch.float16)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
log = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
import os
ABLV = int(os.environ.get("ABL", "16"))
for _ in range(2):
    log.zero_()
    f(ABLV, A.data_ptr(), K, w.data_ptr(), K, M, N, K, out.data_ptr(), N, log.data_ptr(), st)
torch.cuda.synchronize()
L = log.cpu().numpy().reshape(8, 4096)
names = ["reads_issue", "glds_issue", "lgkm_wait", "vm_wait(g1)", "barrier1", "mfma", "vm_wait(g0)", "barrier2->next"]
for wv in (0, 1, 4, 5):
    t = L[wv]; n = (t > 0).sum() // 8
    ts = t[: n * 8].reshape(n, 8).astype(np.float64)
    seg = np.diff(np.concatenate([ts.reshape(-1), [ts[-1, -1]]])).reshape(n, 8)
    seg[:, 7] = np.concatenate([ts[1:, 0] - ts[:-1, 7], [0]])
    body = seg[8:-2]
    print(f"wave {wv}: phases {n}, mean ticks/phase {np.diff(ts[:,0]).mean():.0f}; per-segment mean:", " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, body.mean(0))))
    print("   p50:", " ".join(f"{v:.0f}" for v in np.percentile(body, 50, axis=0)), " p90:", " ".join(f"{v:.0f}" for v in np.percentile(body, 90, axis=0)))
print("memtime ticks are at a constant 100 MHz (s_memtime) or shader clock depending on the part; total loop ticks wave0:", L[0][(L[0] > 0).sum() - 1] - L[0][0])
