"""Diagnostic: what a plain streaming kernel reaches on this GPU (the practical ceiling under the 8 TB/s
peak that the push kernel's 80 B/update are priced against)."""
import torch

n = 2 * 10 ** 7
x = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(5)]
y = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]


def run(fn, nbytes, name, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name}: {ms:.4f} ms, {nbytes / ms / 1e9:.2f} TB/s")


run(lambda: y[0].copy_(x[0]), 16 * n, "copy 1 array (8 R + 8 W per element)")
run(lambda: torch.add(x[0], x[1], out=y[0]), 24 * n, "add (16 R + 8 W)")
big = torch.rand(10 * n, dtype=torch.float64, device="cuda")
out = torch.empty_like(big)
run(lambda: out.copy_(big), 16 * 10 * n, "copy 1.6 GB + 1.6 GB")
run(lambda: big.mul_(1.0000001), 16 * 10 * n, "in-place scale of 1.6 GB (8 R + 8 W)")
run(lambda: big.sum(), 8 * 10 * n, "read-only sum of 1.6 GB")
