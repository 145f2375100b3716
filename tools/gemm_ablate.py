import ctypes as C, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stamp_amd import _lib
from tools.bench_kernels import timeit
lib = _lib.lib()
f = lib.amds_gemm_ablate
f.restype = C.c_int
f.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (M, N, K) in ((65536, 1024, 4096), (65536, 4096, 1024)):
    A = torch.randn(M, K, generator=g).to(dev, torch.float16); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, torch.float16)
    bias = torch.randn(N, generator=g).to(dev); out = torch.empty(M, N, dtype=torch.float16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    names = {0: "full", 32: "full, no setprio", 64: "full, LOAD wave prio 2", 128: "full, AGPR acc", 192: "full, AGPR acc + LOAD prio 2", 1: "no-copies", 2: "no-mfma", 4: "no-dsread", 8: "no-barrier", 3: "no-copies,no-mfma", 5: "no-copies,no-dsread", 6: "no-mfma,no-dsread (copies+barriers)",
             7: "barriers only", 9: "no-copies,no-barrier", 12: "no-dsread,no-barrier", 14: "copies only (no barrier)", 15: "nothing"}
    if len(sys.argv) > 1 and sys.argv[1] == "4w":
        names = {1000: "4w full", 1001: "4w no-copies", 1004: "4w no-dsread", 1005: "4w no-copies,no-dsread (MFMA+barrier)", 1008: "4w no-barrier",
                 1009: "4w no-copies,no-barrier", 1012: "4w no-dsread,no-barrier", 1013: "4w MFMA only"}
    if len(sys.argv) > 1 and sys.argv[1] == "4w64":
        names = {2000: "4w64 full", 2001: "4w64 no-copies", 2004: "4w64 no-dsread", 2005: "4w64 no-copies,no-dsread (MFMA+barrier)",
                 2008: "4w64 no-barrier", 2009: "4w64 no-copies,no-barrier", 2012: "4w64 no-dsread,no-barrier", 2013: "4w64 MFMA only"}
    best = {}
    for rnd in range(3):      # round-robin, best of 3: the clocks drift over a run
        for abl, nm in names.items():
            t = timeit(lambda: f(abl, A.data_ptr(), K, w.data_ptr(), K, M, N, K, out.data_ptr(), N, bias.data_ptr(), st), iters=10)
            best[abl] = min(best.get(abl, 1e9), t)
    for abl, nm in names.items():
        t = best[abl]
        print(f"M={M} N={N} K={K} abl={abl:2d} {nm:40s} {t*1e6:8.1f} us  ({2*M*N*K/t/1e12:7.1f} 'TF/s')", flush=True)
