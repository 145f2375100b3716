"""The collectives of the N > 1 path (stamp_amd/distributed.py: the padded all-gather of slide embeddings + ids, the MAX all-reduces of the
bench's timing, the SUM all-reduce of the DP-MIL gradient buffer, barrier(device_ids)) issued through RCCL on ONE MI355X at world size 1 --
the 8-GPU runs are the driver's; this only shows that RCCL initialises on the box and accepts every dtype / op the path uses.
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_world1_probe.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")), device_id=dev)
w = dist.get_world_size()
print(f"backend {dist.get_backend()}, world {w}, init {time.perf_counter() - t0:.2f} s, torch {torch.__version__}, hip {torch.version.hip}")
n_local = torch.tensor([125], dtype=torch.int64, device=dev)
dist.all_reduce(n_local, op=dist.ReduceOp.MAX)
emb, ids = torch.randn(125, 768, device=dev), torch.arange(125, device=dev)
all_emb, all_ids = torch.empty(w * 125, 768, device=dev), torch.empty(w * 125, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(all_emb, emb)
dist.all_gather_into_tensor(all_ids, ids)
assert torch.equal(all_emb[:125], emb) and torch.equal(all_ids[:125], ids) and int(n_local) == 125
g = torch.randn(3_683_330, device=dev)             # the default `vit` head's flat fp32 gradient buffer (14.7 MB)
ref = g.clone()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
dt = (time.perf_counter() - t1) / 20
assert torch.equal(g, ref)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[dev.index])
print(f"all_reduce MAX int64 / float64, all_gather_into_tensor fp32 [{125}x768] + int64, all_reduce SUM fp32 14.7 MB ({dt * 1e6:.0f} us per call at world 1), barrier: ok")
dist.destroy_process_group()
