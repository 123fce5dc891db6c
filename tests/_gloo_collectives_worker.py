"""One rank of a CPU-only check of opal_amd.exchange.TorchCollectives: the two callbacks opal_sim_run uses for
its output-time reductions (opal_sim_set_collectives), driven through their C signatures over gloo."""
import ctypes as C
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opal_amd.exchange import TorchCollectives  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coll = TorchCollectives("cpu")
    dp = C.POINTER(C.c_double)
    # allreduce: sum / min / max, in place
    for op, want in ((0, lambda a: a.sum(0)), (1, lambda a: a.min(0)), (2, lambda a: a.max(0))):
        all_vals = np.array([[r + 1.0, -2.0 * r, 10.0 - r] for r in range(world)])
        mine = all_vals[rank].copy()
        assert coll.allreduce_cb(None, mine.ctypes.data_as(dp), len(mine), op) == 0
        assert np.array_equal(mine, want(all_vals)), (op, mine)
    # gather: ranks contribute different lengths; rank 0 receives them in rank order
    counts = [11 * (3 + r) for r in range(world)]
    mine = np.arange(counts[rank], dtype=np.float64) + 1000.0 * rank
    out = np.full(sum(counts), -1.0) if rank == 0 else None
    cnt = (C.c_size_t * world)(*counts)
    rc = coll.gather_cb(None, mine.ctypes.data_as(dp), len(mine), out.ctypes.data_as(dp) if rank == 0 else None, cnt, world)
    assert rc == 0
    if rank == 0:
        want = np.concatenate([np.arange(c, dtype=np.float64) + 1000.0 * r for r, c in enumerate(counts)])
        assert np.array_equal(out, want)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
