import torch, time
h = torch.empty(256*1024*1024, dtype=torch.uint8).pin_memory()
d = torch.empty_like(h, device="cuda")
for n in (4, 64, 256):
    m = n*1024*1024
    d[:m].copy_(h[:m], non_blocking=True); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5): d[:m].copy_(h[:m], non_blocking=True)
    torch.cuda.synchronize(); el=(time.perf_counter()-t)/5
    print(f"H2D {n} MB pinned: {m/el/1e9:.1f} GB/s ({el*1e3:.2f} ms)")
    t=time.perf_counter()
    for _ in range(5): h[:m].copy_(d[:m], non_blocking=True)
    torch.cuda.synchronize(); el=(time.perf_counter()-t)/5
    print(f"D2H {n} MB pinned: {m/el/1e9:.1f} GB/s")
import numpy as np
a = np.random.randint(0,255,(1024,1024,4),dtype=np.uint8); hn = h.numpy()[:a.size].reshape(a.shape); p = np.empty_like(a)
t=time.perf_counter()
for _ in range(50): hn[...] = a
print("cpu write 4MB into pinned: %.2f ms" % ((time.perf_counter()-t)/50*1e3))
t=time.perf_counter()
for _ in range(50): p[...] = a
print("cpu write 4MB into pageable: %.2f ms" % ((time.perf_counter()-t)/50*1e3))
