"""Diagnostic: the RCCL transport of opal_amd.exchange with a single rank (both neighbours are the rank
itself), the only way to drive the nccl code path on a one-GPU box."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opal_amd.exchange import NeighbourTransport  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
tr = NeighbourTransport(0, 1)
n = 1 << 16
sl = torch.arange(n, dtype=torch.uint8, device=dev)
sr = (torch.arange(n, device=dev) * 3 % 251).to(torch.uint8)
rl = torch.zeros(n, dtype=torch.uint8, device=dev)
rr = torch.zeros(n, dtype=torch.uint8, device=dev)
for it in range(3):
    tr.exchange(sl, sr, rl, rr)
torch.cuda.synchronize()
print("recv_right == send_left:", bool(torch.equal(rr, sl)), " recv_left == send_right:", bool(torch.equal(rl, sr)))
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(200):
    tr.exchange(sl, sr, rl, rr)
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("host us/exchange %.1f  total us/exchange %.1f" % (host / 200 * 1e6, tot / 200 * 1e6))
dist.destroy_process_group()
