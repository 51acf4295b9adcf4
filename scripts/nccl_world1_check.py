#!/usr/bin/env python3
"""RCCL sanity on a one-GPU box: process group "nccl" with world_size 1, the pipelined sample_and_gather read-back, an
all-reduce and an all-gather (the collectives the sharded path uses).  Multi-GPU runs are the driver's."""
import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
from torchebm_amd.utils import distributed as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = ta.LangevinDynamics(ta.DoubleWellModel(device="cuda"), step_size=0.01, device="cuda")
x = torch.randn(4096, 64, device="cuda")
local, gathered = D.sample_and_gather(s, x, 20, pieces=4)
t = torch.ones(8, device="cuda"); dist.all_reduce(t)
out = torch.empty(8, device="cuda"); dist.all_gather_into_tensor(out, t)
print("nccl world=1 ok", gathered.shape, float(t.sum()), dist.get_backend())
dist.destroy_process_group()
