#!/usr/bin/env python3
"""One-rank RCCL group on the GPU: the torch.distributed calls bench.py makes at N > 1 (barrier, MAX / SUM all-reduce on
device tensors, the host-side gather of per-image lengths) accept the nccl backend."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from seqoia_amd import shard
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.5], device="cuda", dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
tt = torch.tensor([1.0, 2.0, 3.0], device="cuda", dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.SUM)
print("gather:", shard.gather_lengths([2, 0, 1], [30, 10, 20], 3), float(t.item()), tt.tolist())
dist.destroy_process_group()
