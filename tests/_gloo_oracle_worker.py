"""One rank of the decomposed ORACLE run over gloo (CPU only): exercises opal_amd.exchange's
NeighbourTransport with the reference's message pattern (particles per species, then halo)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _decomp_common import A_POSITRON, C, global_state, split  # noqa: E402
from opal_amd.exchange import NeighbourTransport  # noqa: E402
from oracle import oracle as o  # noqa: E402

CAP = 1 << 20  # bytes per direction for particle messages


def exchange_bytes(tr, left_payload, right_payload, to_left, to_right):
    """variable-length payloads inside fixed-size messages: [u64 length][bytes]"""
    def pack(b):
        t = torch.zeros(CAP, dtype=torch.uint8)
        t[:8] = torch.frombuffer(bytearray(np.uint64(len(b)).tobytes()), dtype=torch.uint8)
        if b:
            t[8:8 + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        return t

    def unpack(t):
        n = int(np.frombuffer(t[:8].numpy().tobytes(), dtype=np.uint64)[0])
        return t[8:8 + n].numpy().tobytes()

    sl = pack(left_payload) if to_left else None
    sr = pack(right_payload) if to_right else None
    rl = torch.zeros(CAP, dtype=torch.uint8) if to_left else None
    rr = torch.zeros(CAP, dtype=torch.uint8) if to_right else None
    tr.exchange(sl, sr, rl, rr)
    return (unpack(rl) if to_left else b""), (unpack(rr) if to_right else b"")


def main():
    outdir, nx, npc, steps, left, right = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = NeighbourTransport(rank, world)
    dx = 1.0e-8
    dt = 0.95 * dx / C
    nxs, offs = o.design_unbalanced(world, nx, dx)
    g = o.Grid(rank, world, left, right, 0.0, dx, nxs, offs)
    se, si = global_state(nx, npc, dx, 5), global_state(nx, npc, dx, 6)
    pe = o.Population.from_soa(o.ELECTRON, *split(se, nxs, rank), dx, dt)
    pi = o.Population.from_soa(o.ION, *split(si, nxs, rank), dx, dt)
    pi.set_charge_state(1.0, 1.0, A_POSITRON)
    to_left, to_right = g.to_left() >= 0, g.to_right() >= 0
    import ctypes
    t = 0.0
    for _ in range(steps):
        for p in (pe, pi):  # Population::advance, mod.rs:247-376
            assert p.push_and_sort(g, dt) == 0
            sl, sr = p.leavers()
            rl, rr = exchange_bytes(tr, sl, sr, to_left, to_right)
            p.splice(g, rl, rr)
        g.clear()
        g.deposit(pe, dt)
        g.deposit(pi, dt)
        # synchronize, yee.rs:351-452
        hl, hr = bytes(g.send_left()), bytes(g.send_right())
        rl, rr = exchange_bytes(tr, hl, hr, to_left, to_right)
        bl = (o.Cell * 8).from_buffer_copy(rl) if to_left else None
        br = (o.Cell * 8).from_buffer_copy(rr) if to_right else None
        g.synchronize_apply(bl, br, None, None, t, dt)
        g.advance(dt)
        t += dt
    out = {k: g.raw(k) for k in ("Ex", "Ey", "Ez", "By", "Bz", "jx", "jy", "jz", "rho")}
    for name, p in (("e", pe), ("i", pi)):
        for k, v in p.soa().items():
            out[f"{name}_{k}"] = v
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
