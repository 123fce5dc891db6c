#!/usr/bin/env python3
"""Headline benchmark: particle-updates/s of the per-timestep PIC loop on N MI355X.

One "step" = one iteration of src/main.rs:238-267 (push + deposit of every charged species, halo/
boundary synchronisation, field solve) on synthetic two-stream input already resident in HBM.
Per GPU the workload is BASELINE.json configs[1]: 4096 cells, 2441 electrons + 2441 positrons per
cell (2 x 9 998 336 macro-particles), periodic, current deposition on, no QED; with N GPUs the
domain is N x 4096 cells split into N subdomains (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fused push+deposit
pass; one launch covers both species): algorithmic bytes = 80 B per particle-update x
particles per launch, divided by the kernel's mean duration measured with HIP events around a 1-in-5
sample of its launches in the timed region.
`cpu_baseline` times the CPU oracle (a port of the reference's algorithm, not the opal binary) on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CELLS_PER_GPU = 4096
NPC = 2441
BYTES_PER_UPDATE = 80.0   # read cell,x,u[3],w = 44 B; write cell,x,u[3] = 36 B
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

YAML = """---
control:
 dx: 500
 nx: {nx}
 xmin: 0.0
 start: 0.0
 end: 1.0
 current_deposition: true
 n_outputs: 20
 balance: false

qed:
 photon_emission: false
 photon_absorption: false

electrons:
 npc: {npc}
 ne: 20.0
 ux: drift_u * (1.0 + 0.001 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.0
 uz: 0.0
 output: x:px

ions:
 name: positron
 npc: {npc}
 Z: 1.0
 A: me/mp
 ni: 20.0
 ux: drift_u * (1.0 + 0.001 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.0
 uz: 0.0
 output: x:px

constants:
 drift_u: 2.5e-24 / (me * c)
"""


# BASELINE configs[3]: examples/hole_boring.yaml's physics (a0 = 300 laser on an overdense electron-carbon slab,
# laser + absorbing boundaries, balanced decomposition) with npc scaled; the whole 20 000-cell domain is divided
# among the GPUs (strong scaling).  The pulse reaches the slab after ~21 000 steps: --warmup 21000 times the
# interaction, the default warm-up the approach.
LASER_PLASMA_YAML = """---
control:
 dx: micro / 1000
 nx: 20000
 xmin: -10*micro
 start: -20.0e-6/c
 end: 10.0e-6/c
 current_deposition: true
 n_outputs: 30
 balance: true
qed:
 photon_emission: false
 photon_absorption: false
electrons:
 npc: {npc}
 ne: density * critical(omega) * step(x,xmin,xmax)
 ux: sqrt(kT/(m*c^2)) * nrand
 uy: sqrt(kT/(m*c^2)) * nrand
 uz: sqrt(kT/(m*c^2)) * nrand
 output: [x:px]
ions:
 name: carbon
 npc: {npc}
 Z: Z
 A: A
 ni: density * critical(omega) * step(x,xmin,xmax) / Z
 ux: sqrt(kT/(A*mp*c^2)) * nrand
 uy: sqrt(kT/(A*mp*c^2)) * nrand
 uz: sqrt(kT/(A*mp*c^2)) * nrand
 output: [x:px]
laser:
 Ey: (a0*me*c*omega/e) * gauss_pulse_re(t,x,omega,sigma)
 Ez: (a0*me*c*omega/e) * gauss_pulse_im(t,x,omega,sigma)
constants:
 density: 100.0
 a0: 300.0
 omega: 2*pi*c/0.8e-6
 sigma: pi * 4.0 / sqrt(ln(2.0))
 xmin: 0.0 * micro
 xmax: 5.0 * micro
 Z: 6.0
 A: 12.0
 kT: 500 * eV
"""


def cpu_baseline(oa, target_seconds=12.0):
    """Times the oracle (OpenMP over particle chunks, one subdomain) on a bounded sample of the
    workload: same grid, same physics, 1/10 of the particles per cell, a few steps."""
    import numpy as np
    from oracle import oracle as o

    o.build()
    npc_s = NPC // 10
    sim = oa.Simulation(yaml_text=YAML.format(nx=CELLS_PER_GPU, npc=npc_s))
    dx, dt = 500.0, sim.dt
    se, si = sim.electrons.soa(), sim.ions.soa()
    del sim
    threads = max(1, min(o.max_threads(), len(os.sched_getaffinity(0)), 16))  # a 1-GPU box gets a 16-core share
    grid = o.Grid.single(CELLS_PER_GPU, 0.0, dx)
    pe = o.Population.from_soa(o.ELECTRON, se["cell"], se["x"], se["ux"], se["uy"], se["uz"], se["w"], dx, dt)
    pi = o.Population.from_soa(o.ION, si["cell"], si["x"], si["ux"], si["uy"], si["uz"], si["w"], dx, dt)
    pi.set_charge_state(1.0, 1.0, 9.109383e-31 / 1.672622e-27)
    n = len(pe) + len(pi)
    t0 = time.perf_counter()
    o.world_run_periodic([grid], [pe], [pi], True, 0.0, dt, 2, threads)
    per_step = (time.perf_counter() - t0) / 2
    steps = int(max(3, min(200, target_seconds / max(per_step, 1e-6))))
    t0 = time.perf_counter()
    bad = o.world_run_periodic([grid], [pe], [pi], True, 2 * dt, dt, steps, threads)
    el = time.perf_counter() - t0
    assert bad == 0
    return {
        "value": n * steps / el,
        "unit": "particle-updates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"two_stream e-+e+, {CELLS_PER_GPU} cells, npc {npc_s}+{npc_s} ({n} particles), {steps} steps, "
                  f"C oracle with OpenMP over {threads} particle chunks (per-step counting sort, per-chunk private "
                  f"current arrays + serial reduce, serial field solve)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--npc", type=int, default=NPC)
    ap.add_argument("--cells-per-gpu", type=int, default=CELLS_PER_GPU,
                    help="other BASELINE configs, e.g. configs[2] on one GPU: --cells-per-gpu 32768 --npc 1526")
    ap.add_argument("--unfused", action="store_true", help="time the reference's operator sequence instead")
    ap.add_argument("--workload", choices=["two_stream", "laser_plasma"], default="two_stream",
                    help="two_stream (default; BASELINE configs[1] per GPU, weak scaling) or laser_plasma "
                         "(configs[3]'s physics, --npc per cell in 5000 of 20000 cells, strong scaling)")
    ap.add_argument("--self-exchange", action="store_true",
                    help="N=1 rehearsal of the multi-GPU step: the lone periodic subdomain sends its halo and leavers "
                         "to itself through RCCL (pack kernels, torch.distributed P2P, unpack) instead of wrapping in place")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: everything else that libraries print there (RCCL's version banner
    # at communicator creation, for one) is sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import opal_amd as oa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # OPAL_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share the
    # visible GPUs and exchange through host buffers.  The driver's runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("OPAL_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    oa.set_device(dev_index)
    link_kind = os.environ.get("OPAL_BENCH_LINK", "rccl")
    # leavers per species, neighbour and step that a message has room for: the two-stream flux through a cut is
    # npc * <|v_x|> dt / dx ~ 2441 * 0.009 ~ 25 per step, so 1024 is 40x headroom (an overflow is reported, not
    # silent); messages are fixed-size, 704 B + 2 species * 1024 * 52 B = 107 KB per neighbour per step
    LINK_CAPACITY = 1024

    def make_link(r, w, device):
        """Neighbour transport over RCCL: native (the library issues the send/recv itself on the simulation's
        stream) unless OPAL_BENCH_LINK=torch asks for the torch.distributed callback."""
        from opal_amd.exchange import RcclLink, TorchLink
        nonlocal link_kind
        if link_kind == "rccl":
            try:
                if os.environ.get("OPAL_BENCH_EXACT"):  # diagnostic: big buffers, messages sized to their content
                    return RcclLink(r, w, device, particle_capacity=1 << 18, exact=True)
                return RcclLink(r, w, device, particle_capacity=LINK_CAPACITY)
            except Exception as e:  # e.g. no librccl to open: the torch.distributed transport does the same job
                print(f"[bench] native RCCL link unavailable ({e}); using torch.distributed P2P", file=sys.stderr)
                link_kind = "torch"
        return TorchLink(r, w, device, particle_capacity=LINK_CAPACITY)

    link = None
    if args.self_exchange and world == 1:
        import torch.distributed as dist
        from opal_amd.exchange import TorchLink
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
        link = make_link(0, 1, torch.device("cuda", dev_index))
    if world > 1:
        import torch.distributed as dist
        from opal_amd.exchange import TorchLink
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
            link = make_link(rank, world, torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
            link = TorchLink(rank, world, torch.device("cpu"))
    red_dev = "cuda" if backend == "nccl" else "cpu"

    stream = torch.cuda.current_stream().cuda_stream
    cells_per_gpu = args.cells_per_gpu
    nx = cells_per_gpu * world
    if args.workload == "laser_plasma":
        nx = 20000
        sim = oa.Simulation(yaml_text=LASER_PLASMA_YAML.format(npc=args.npc), rank=rank, numtasks=world, link=link,
                            stream=stream)
    else:
        sim = oa.Simulation(yaml_text=YAML.format(nx=nx, npc=args.npc), rank=rank, numtasks=world, link=link,
                            stream=stream)
    sim.set_fused(not args.unfused)
    if args.self_exchange and world == 1:
        sim.grid.set_self_exchange(True)
    if os.environ.get("OPAL_BENCH_OVERLAP", "0") == "1":  # diagnostic: exchange on a second stream under the bulk push
        sim.set_overlap(True)
    n_local = len(sim.electrons) + len(sim.ions)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    sim.step(args.warmup)
    sim.sync()
    oa.prof_reset()
    # HIP events around every 5th launch of each tracked kernel (an odd stride samples both species);
    # timing every launch costs ~10 % of the step because each event record breaks back-to-back dispatch
    oa.prof_enable(5)
    barrier()
    t0 = time.perf_counter()
    sim.step(args.steps)
    sim.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    oa.prof_enable(0)

    n_total = n_local
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nn = torch.tensor([n_local], dtype=torch.float64, device=red_dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_total = int(nn.item())

    kname = "fused" if not args.unfused else "advance"
    launches, total_ms = oa.prof_get(kname)
    sort_launches, sort_ms = oa.prof_get("sort")
    field_launches, field_ms = oa.prof_get("field")
    # the dominant kernel covers this rank's particles once per step, in one launch (species paired) or two
    launches_total = oa.prof_seen(kname)
    per_launch_particles = n_local * args.steps / max(launches_total, 1)
    avg_ms = total_ms / max(launches, 1)
    achieved = BYTES_PER_UPDATE * per_launch_particles / (avg_ms * 1e-3) / 1e9 if launches else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            # PMC-measured HBM bytes per particle-update (profiles/r01_pmc_hbm.json) x updates per launch
            traffic = json.load(open(tpath)).get("hbm_bytes_per_particle_update") * per_launch_particles
        except Exception:
            traffic = None

    if rank == 0:
        value = n_total * args.steps / elapsed
        out = {
            "metric": "particle-updates/sec",
            "value": value,
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if args.workload == "two_stream" else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"two_stream, {args.npc} e- + {args.npc} e+ per cell, {cells_per_gpu} cells per GPU "
                             + ("(BASELINE configs[1] per GPU)" if (cells_per_gpu, args.npc) == (CELLS_PER_GPU, NPC) else "")
                             + ", periodic, push+deposit+field, no QED") if args.workload == "two_stream" else
                            (f"laser_plasma (hole_boring.yaml physics, BASELINE configs[3]), {args.npc} e- + {args.npc} C6+ per "
                             f"cell in 5000 of 20000 cells, laser + absorbing boundaries, t = {sim.t:.3e} s at the end of the "
                             f"timed steps"),
                "particles_total": n_total,
                "cells_total": nx,
                "path": "fused advance+deposit" if not args.unfused else "advance, clear, deposit (reference order)",
                "parallelism": f"{world} subdomain(s), 1 per GPU"
                               + (", self-exchange through RCCL (multi-GPU step rehearsed on one GPU)" if args.self_exchange else "")
                               + (f", neighbour exchange: {'native RCCL send/recv' if link_kind == 'rccl' else 'torch.distributed P2P'}"
                                  if link is not None and backend == "nccl" else ""),
                "frac_of_hbm_roofline_whole_step": value * BYTES_PER_UPDATE / (world * HBM_PEAK_GBS * 1e9),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_push<fused>" if not args.unfused else "k_push<advance>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launches_timed": launches,
                "launches_total": launches_total,
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": BYTES_PER_UPDATE * per_launch_particles,
                "other_kernels_avg_ms": {"field_solve": field_ms / max(field_launches, 1), "rebin": sort_ms / max(sort_launches, 1)},
                "rebins_in_timed_region": oa.prof_seen("sort"),
            },
        }
        if not args.no_cpu_baseline and world == 1 and args.workload == "two_stream":
            del sim
            out["cpu_baseline"] = cpu_baseline(oa)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.self_exchange:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
