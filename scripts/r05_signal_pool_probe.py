#!/usr/bin/env python3
"""How far the host runs ahead of the device before a launch blocks (the HIP runtime's pool of completion signals), and whether
ROC_SIGNAL_POOL_SIZE set by the process itself -- before torch is imported -- is honoured.
usage: r05_signal_pool_probe.py [inproc]"""
import os
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "inproc":
    os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "1024")
import torch  # noqa: E402

x = torch.zeros(64 << 20, device="cuda")  # 256 MB: an add_ takes ~0.1 ms
for _ in range(10):
    x.add_(1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
marks = []
for i in range(600):
    x.add_(1.0)
    if i in (31, 63, 127, 255, 599):
        marks.append((i + 1, time.perf_counter() - t0))
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"ROC_SIGNAL_POOL_SIZE={os.environ.get('ROC_SIGNAL_POOL_SIZE')} ({'set by the process' if len(sys.argv) > 1 else 'as started'}): "
      + ", ".join(f"{n} launches enqueued after {1e3 * t:.2f} ms" for n, t in marks) + f"; all done after {1e3 * total:.1f} ms")
