"""One rank of a decomposed HIP run; ranks share cuda:0 and exchange through gloo (host staging).
mode "ops": operator-level calls with a link, from a global seeded state.
mode "sim": the YAML driver's fused step (one exchange per step)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import opal_amd as oa  # noqa: E402
from _decomp_common import A_POSITRON, C, global_state, split  # noqa: E402
from opal_amd.exchange import PeerLink, TorchLink  # noqa: E402

YAML = """---
control:
 dx: 1.0e-8
 nx: {nx}
 xmin: 0.0
 start: 0.0
 end: 1.0e-12
 current_deposition: true
 n_outputs: 2
 balance: false
qed:
 photon_emission: false
 photon_absorption: false
electrons:
 npc: {npc}
 ne: 1.0e27
 ux: 0.2 * (1.0 + 0.05 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.01 * nrand
 uz: 0.01 * nrand
 output: x:px
ions:
 name: positron
 npc: {npc}
 Z: 1.0
 A: me/mp
 ni: 1.0e27
 ux: 0.2 * (1.0 + 0.05 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.01 * nrand
 uz: 0.01 * nrand
 output: x:px
"""


# colliding_beams.yaml's physics on a 20 um domain cut into subdomains, plus a seeded photon population
# that straddles the cut and flies both ways (so that photons migrate too)
QED_YAML = """---
control:
 dx: 0.01*micro
 nx: 2000
 xmin: -10*micro
 start: -16.0e-6/c
 end: -2.0e-6/c
 current_deposition: false
 n_outputs: 2
 balance: false
qed:
 photon_emission: true
 photon_absorption: false
 photon_angle_max: 50 * milli
electrons:
 npc: 10
 ne: S * a0 * critical(omega) * step(x,xmin,xmax)
 ux: -1000.0 * (1.0 + 0.01 * nrand)
 uy: 0.0
 uz: 0.0
 output: [x]
ions:
 npc: 0
photons:
 npc: 4
 nph: 1.0e20 * step(x, -0.5*micro, 0.5*micro)
 ux: 10.0 * (2.0 * floor(urand - 0.5) + 1.0)
 uy: nrand
 uz: 0.0
 output: [x]
laser:
 Ey: >
  (a0*m*c*omega/e)
  *sin(omega*(t-x/c))
  *exp(-ln(2.0)*(omega*(t-x/c))^2/(2.0*pi^2*ncycles^2))
 Ez: 0.0
constants:
 S: 1.0e-6
 a0: 20.0
 omega: 2*pi*c/0.8e-6
 ncycles: 4.0
 xmin: (4 - 0.5) * micro
 xmax: (4 + 0.5) * micro
"""


def maybe_recut(sim, world):
    """OPAL_TEST_RECUT=every,threshold,max_shift: the run moves its cuts (Simulation.set_recut)"""
    spec = os.environ.get("OPAL_TEST_RECUT")
    if spec and world > 1:
        every, threshold, max_shift = spec.split(",")
        sim.set_recut(int(every), float(threshold), int(max_shift))


def recut_report(sim, world):
    st = sim.recut_stats(world)
    return {"sizes": np.array(st["sizes"]), "recuts": st["recuts"], "checks": st["checks"], "size": sim.grid.size,
            "L": sim.grid.left_bdy_size, "drift_steps": st["drift_steps"], "drift_v": np.array(st["drift_cells_per_step"])}


def dump(outdir, rank, g, pe, pi, extra=None):
    out = {k: g.raw(k) for k in ("Ex", "Ey", "Ez", "By", "Bz", "jx", "jy", "jz", "rho")}
    for name, p in (("e", pe), ("i", pi)):
        for k, v in p.soa().items():
            out[f"{name}_{k}"] = v
    out["fe"] = g.em_field_energy()
    out["ke_e"] = pe.total_kinetic_energy()
    out["ke_i"] = pi.total_kinetic_energy()
    if extra:
        out.update(extra)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    nx, npc, steps, fused = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oa.set_device(0)
    # OPAL_TEST_LINK=peer: peer-mapped mailboxes between the processes (hipIpc handles travel over gloo once; the
    # kernels of the step then exchange by themselves) instead of host-staged gloo messages
    kind = os.environ.get("OPAL_TEST_LINK", "gloo")
    peer = kind == "peer"
    cap = 128 if mode != "qed" else 1024
    if os.environ.get("OPAL_TEST_LINK_CAP"):
        cap = int(os.environ["OPAL_TEST_LINK_CAP"])
    if world == 1:
        link = None
    elif kind == "auto":
        # what bench.py and the launcher do: the best transport that comes up on EVERY subdomain (with
        # OPAL_PEER_ALLOC=none the mailboxes cannot be allocated anywhere, with OPAL_TEST_BREAK_RANK=r only there)
        from opal_amd.exchange import make_link
        if os.environ.get("OPAL_TEST_BREAK_RANK") == str(rank):
            os.environ["OPAL_PEER_ALLOC"] = "none"
        notes = []
        link, got = make_link(rank, world, torch.device("cpu"), cap, photon_emission=mode == "qed", log=notes.append)
        peer = got == "peer"
        with open(os.path.join(outdir, f"link_kind_{rank}.txt"), "w") as f:
            f.write(got + "\n" + "\n".join(notes))
    else:
        Link = PeerLink if kind == "peer" else TorchLink
        link = Link(rank, world, torch.device("cpu"), particle_capacity=cap, photon_emission=mode == "qed")
    dx = 1.0e-8
    dt = 0.95 * dx / C
    if mode == "grow":
        # everything starts on rank 0 and streams to the right through open boundaries: rank 1 starts
        # empty (minimum capacity) and has to enlarge its storage as the particles arrive
        left, right = oa.BDY_LASER, oa.BDY_ABSORBING
        nxs, offs = oa.design_unbalanced(world, nx, dx)
        g = oa.Grid(rank, world, left, right, 0.0, dx, nxs, offs)
        rng = np.random.default_rng(9)
        n = nx * npc
        # (OPAL_TEST_GROW_GAP: the bunch starts that many cells short of the cut, so that the link has the history of a few
        # exchanges -- what its growth is planned from -- before the first particle crosses)
        cell = rng.integers(0, nx // 2 - int(os.environ.get("OPAL_TEST_GROW_GAP", "0")), n)
        st = (cell, rng.random(n), 0.5 + 0.05 * rng.standard_normal(n), np.zeros(n), np.zeros(n),
              1.0e3 * (1.0 + 1e-9 * np.arange(n)))
        pe = oa.Population.from_soa(oa.ELECTRON, *split(st, nxs, rank), dx, dt, capacity=1024 if rank else None)
        pi = oa.Population.from_soa(oa.ION, *split(st, nxs, rank), dx, dt, capacity=1024 if rank else None)
        t = 0.0
        zero = lambda t_, x_: 0.0
        for _ in range(steps):
            g.clear()
            pe.advance_deposit(g, dt, link)
            pi.advance_deposit(g, dt, link)
            g.synchronize(zero, zero, t, dt, link)
            g.advance(dt)
            t += dt
        pe.total_kinetic_energy()  # (a status check: an overflow of the link surfaces here)
        dump(outdir, rank, g, pe, pi, {"link_growth": np.array(link.growth() if hasattr(link, "growth") else (0, 0, 0))})
    elif mode == "qed":
        deposit = nx == 1  # argv[3] doubles as the current_deposition switch in this mode
        text = QED_YAML.replace("current_deposition: false", "current_deposition: true") if deposit else QED_YAML
        sim = oa.Simulation(yaml_text=text, rank=rank, numtasks=world, link=link)
        out = {}
        tau, ids = sim.electrons.emission_state()
        for k, v in sim.electrons.soa().items():
            out[f"init_e_{k}"] = v
        out["init_e_tau"], out["init_e_id"] = tau, ids
        for k, v in sim.photons.soa().items():
            out[f"init_ph_{k}"] = v
        nsteps = sim.total_steps if steps <= 0 else steps
        out["init_size"] = sim.grid.size
        maybe_recut(sim, world)
        sim.step(nsteps)
        sim.sync()
        out.update(recut_report(sim, world))
        tau, ids = sim.electrons.emission_state()
        for k, v in sim.electrons.soa().items():
            out[f"e_{k}"] = v
        out["e_tau"], out["e_id"] = tau, ids
        for k, v in sim.photons.soa().items():
            out[f"ph_{k}"] = v
        out["ph_id"] = sim.photons.emission_state()[1]
        out["nsteps"] = nsteps
        out["ke_e"] = sim.electrons.total_kinetic_energy()
        out["ke_ph"] = sim.photons.total_kinetic_energy()
        out["fe"] = sim.grid.em_field_energy()
        out["Ey"] = sim.grid.raw("Ey")
        out["t"] = sim.t
        # leavers of all three species in one message; with deposition a second, halo-only one
        if not peer and not os.environ.get("OPAL_TEST_RECUT"):
            assert link.calls == nsteps * (2 if deposit else 1), (link.calls, nsteps)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    elif mode == "recut":
        # a slab that fills part of a periodic domain cut into equal subdomains: the inner cuts move onto it
        from test_gpu_recut import NONUNIFORM
        sim = oa.Simulation(yaml_text=NONUNIFORM, rank=rank, numtasks=world, link=link)
        sim.set_fused(bool(fused))
        init = {"init_size": sim.grid.size}
        for name, p in (("e", sim.electrons), ("i", sim.ions)):
            for k, v in p.soa().items():
                init[f"init_{name}_{k}"] = v
        maybe_recut(sim, world)
        sim.step(steps)
        sim.sync()
        init.update(recut_report(sim, world))
        init["dt"] = sim.dt
        if link is not None and hasattr(link, "growth"):
            init["link_growth"] = np.array(link.growth())
        dump(outdir, rank, sim.grid, sim.electrons, sim.ions, init)
        if link is not None and peer:
            link.status()  # (raises if a wait for a neighbour's message ever timed out)
    elif mode == "run":
        # the whole of main(): outputs included (grid rows gathered, energies and histograms reduced)
        sim = oa.Simulation(yaml_text=YAML.format(nx=nx, npc=npc).replace("n_outputs: 2", "n_outputs: 3")
                            .replace("end: 1.0e-12", "end: 30.5 * 0.95e-8 / c"),
                            output_dir=outdir, rank=rank, numtasks=world, link=link)
        init = {}
        for name, p in (("e", sim.electrons), ("i", sim.ions)):
            for k, v in p.soa().items():
                init[f"init_{name}_{k}"] = v
        init["steps"] = sim.total_steps
        maybe_recut(sim, world)
        sim.run()
        init.update(recut_report(sim, world))
        dump(outdir, rank, sim.grid, sim.electrons, sim.ions, init)
    elif mode == "ops":
        nxs, offs = oa.design_unbalanced(world, nx, dx)
        g = oa.Grid(rank, world, oa.BDY_INTERNAL, oa.BDY_INTERNAL, 0.0, dx, nxs, offs)
        se, si = global_state(nx, npc, dx, 5), global_state(nx, npc, dx, 6)
        pe = oa.Population.from_soa(oa.ELECTRON, *split(se, nxs, rank), dx, dt, capacity=4 * nx * npc)
        pi = oa.Population.from_soa(oa.ION, *split(si, nxs, rank), dx, dt, capacity=4 * nx * npc)
        pi.set_charge_state(1.0, 1.0, A_POSITRON)
        t = 0.0
        for _ in range(steps):
            if fused:
                g.clear()
                pe.advance_deposit(g, dt, link)
                pi.advance_deposit(g, dt, link)
            else:
                pe.advance(g, dt, link)
                pi.advance(g, dt, link)
                g.clear()
                g.deposit(pe, dt)
                g.deposit(pi, dt)
            g.synchronize(None, None, t, dt, link)
            g.advance(dt)
            t += dt
        dump(outdir, rank, g, pe, pi)
    else:
        text = YAML.format(nx=nx, npc=npc)
        if mode == "sim_es":  # nothing transverse: the linked step runs the electrostatic launch
            text = text.replace(" uy: 0.01 * nrand", " uy: 0.0").replace(" uz: 0.01 * nrand", " uz: 0.0")
        if mode == "sim_thin":  # a plasma too thin to act on itself, cold: every particle moves freely (test: ballistic)
            text = (text.replace(" ne: 1.0e27", " ne: 1.0e5").replace(" ni: 1.0e27", " ni: 1.0e5")
                    .replace(" uy: 0.01 * nrand", " uy: 0.0").replace(" uz: 0.01 * nrand", " uz: 0.0")
                    .replace("0.2 * (1.0 + 0.05 * nrand)", "0.5 * (1.0 + 0.3 * nrand)"))
        if mode == "sim_half":  # transverse momenta in the left half of the domain only: the other subdomains start in
            # the electrostatic launch and meet hot arrivals and the fields the left half radiates
            half = f"step(x, 0.0, {0.5 * nx * dx!r})"
            text = text.replace(" uy: 0.01 * nrand", f" uy: 0.01 * nrand * {half}").replace(" uz: 0.01 * nrand", f" uz: 0.01 * nrand * {half}")
        sim = oa.Simulation(yaml_text=text, rank=rank, numtasks=world, link=link)
        sim.set_fused(bool(fused))
        sim.set_overlap(nx >= 512)  # the optional overlapped step: edge chunks, exchange on a second stream, bulk chunks
        if mode != "sim_thin":  # (sim_thin keeps the adaptive policy)
            sim.electrons.set_sort_interval(16)  # re-bin on a fixed schedule so that the test crosses re-binnings
            sim.ions.set_sort_interval(16)
        init = {}
        for name, p in (("e", sim.electrons), ("i", sim.ions)):
            for k, v in p.soa().items():
                init[f"init_{name}_{k}"] = v
        if link is not None and hasattr(link, "growth"):
            init["link_growth_at_start"] = np.array(link.growth())
        sim.step(steps)
        sim.sync()
        if mode == "sim_stall":
            # subdomain 0 takes a step its neighbour never takes: its tail kernel polls its mailboxes for a message nobody
            # sends, gives up after 20 s of its own polling, and the next sync must say so -- and what it saw
            if rank == 0:
                try:
                    sim.step(1)
                    sim.sync()
                    said = "no error"
                except oa.OpalError as e:
                    said = str(e)
                with open(os.path.join(outdir, "stall_rank0.txt"), "w") as f:
                    f.write(said)
            dist.barrier()  # (the others keep their mailboxes mapped meanwhile)
            dist.destroy_process_group()
            return
        if link is not None and hasattr(link, "growth"):
            init["link_growth"] = np.array(link.growth())
        dump(outdir, rank, sim.grid, sim.electrons, sim.ions, init)
        if link is not None and fused and nx >= 512 and mode != "sim_thin" and not peer:
            # (not before the first re-binning: until then a chunk is 8 x 256 slots, half of such a small population)
            assert sim.overlapped_steps >= steps // 3, sim.overlapped_steps
        if link is not None and not peer:
            # one message per neighbour per step when fused; leavers (all species) + a halo-only message otherwise
            assert link.calls == (steps if fused else 2 * steps), link.calls
        if link is not None and peer:
            kind, n = link.status()  # (raises if a wait for a neighbour's message ever timed out)
            assert n == (steps if fused else 2 * steps), n
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
