"""Run an opal input file on one or more MI355X: the launcher a user of the reference's
`mpirun -n N opal input.yaml` switches to.

    python -m opal_amd input.yaml                                                   # one GPU
    python -m torch.distributed.run --nproc-per-node 8 -m opal_amd input.yaml       # one subdomain per GPU

One process per GPU; the subdomains exchange through peer-mapped mailboxes over xGMI (or RCCL where the GPUs cannot
address each other; OPAL_LINK=peer|rccl|torch forces one; OPAL_RECUT=every[,threshold[,max_shift]] lets the cuts between
the subdomains follow the particles); outputs are written next to the input file by rank 0, as the
reference does (main.rs:60-61)."""
import os
import sys

import torch
import torch.distributed as dist

import opal_amd as oa
from opal_amd.exchange import make_link


def main(argv):
    if len(argv) != 2:
        print("Usage: python -m opal_amd input-file", file=sys.stderr)  # main.rs:56-59
        return 1
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("OPAL_BACKEND", "nccl")  # "gloo": host-staged exchange (rehearsals on one GPU)
    device = torch.device("cuda", local if backend == "nccl" else 0)
    torch.cuda.set_device(device)
    oa.set_device(device.index)
    link = None
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world,
                                **({"device_id": device} if backend == "nccl" else {}))
        cfg = oa.Config(path=argv[1])
        emission, absorption = cfg.read_bool("qed", "photon_emission"), cfg.read_bool("qed", "photon_absorption")
        # Room in the buffers for the leavers of one step through one cut, per species.  A peer link's edges GROW when more
        # particles stand within 8 cells of a cut than they have room for (link.cpp: link_plan / link_grow), from the zone
        # counts of earlier exchanges -- so a link only has to be created with room for what can cross before that history
        # exists: the particles within 8 cells on both sides of a cut at the initial density, 16 cells' worth of the densest
        # species (it used to be created for the worst pile-up ever measured, 128 cells' worth: 0.5 GB at npc = 10 000).
        npc = 0
        for sec in ("electrons", "ions"):
            try:
                npc = max(npc, cfg.read_usize(sec, "npc")) if cfg.contains(sec) else npc
            except oa.OpalError:
                pass  # (the run itself will report a malformed section)
        cap = int(os.environ.get("OPAL_LINK_CAPACITY", str(max(4096, 20 * npc))))
        # ... and a link that cannot grow (torch.distributed's, whose buffers are torch tensors) for the worst pile-up at a cut
        # measured so far (hole boring: 47 x / 93 x the initial density in a cell), as rounds 2-3 sized every link
        fixed = int(os.environ.get("OPAL_LINK_CAPACITY", str(max(65536, 128 * npc))))
        stream = torch.cuda.current_stream().cuda_stream
        open_ends = cfg.contains("laser")  # laser + absorbing boundaries: the first and last subdomain have one neighbour
        want = os.environ.get("OPAL_LINK")
        link, kind = make_link(rank, world, device if backend == "nccl" else torch.device("cpu"), cap, stream=stream,
                               photon_emission=emission, photon_absorption=absorption,
                               to_left=not (open_ends and rank == 0), to_right=not (open_ends and rank == world - 1),
                               prefer=(want,) if want else ("peer", "rccl", "torch"), fixed_capacity=fixed)
        if rank == 0:
            print(f"Neighbour exchange: {kind}")
    try:
        sim = oa.Simulation(yaml_path=argv[1], rank=rank, numtasks=world, link=link,
                            stream=torch.cuda.current_stream().cuda_stream)
        if world > 1 and os.environ.get("OPAL_RECUT"):
            # OPAL_RECUT=every[,threshold[,max_shift]]: the cuts follow the particles (Simulation.set_recut); unset: the
            # reference's partition, made once at t = 0 (grid/mod.rs:157-206)
            spec = os.environ["OPAL_RECUT"].split(",")
            if int(spec[0]):
                sim.set_recut(int(spec[0]), float(spec[1]) if len(spec) > 1 else 1.1, int(spec[2]) if len(spec) > 2 else 64)
        if rank == 0:
            print(f"Running {world} task{'s' if world > 1 else ''} on {world} GPU{'s' if world > 1 else ''}...")
        if world > 1:
            # the waits of the step's exchange are bounded (20 s): start together, however long each subdomain's
            # loader took
            torch.cuda.synchronize()
            dist.barrier()
        sim.run()
    except oa.OpalError as e:
        # No barrier here: the peers are still inside opal_sim_run issuing send/recv and all-reduces to this
        # rank, so a collective entered from the error path could never match.  Leave with a non-zero status
        # at once and let the launcher (torch.distributed.run) tear the other ranks down -- the reference's
        # panics end an MPI job the same way.
        print(e, file=sys.stderr)
        sys.stderr.flush()
        if world > 1:
            os._exit(1)
        return 1
    if world > 1:  # success path only
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
