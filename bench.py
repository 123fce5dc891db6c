#!/usr/bin/env python3
"""Headline benchmark: particle-updates/s of the per-timestep PIC loop on N MI355X.

One "step" = one iteration of src/main.rs:238-267 (push + deposit of every charged species, halo/
boundary synchronisation, field solve) on synthetic input already resident in HBM.

Default workload = north_star's target configuration, BASELINE.json configs[2]: two-stream, 32 768 cells,
1526 electrons + 1526 positrons per cell (100 007 936 macro-particles), periodic, current deposition on,
no QED.  N = 1 holds it on one GPU (4.4 GB); N > 1 splits the SAME domain into N subdomains, one per GPU,
exchanging with their neighbours through peer-mapped mailboxes over xGMI (RCCL where the GPUs cannot address each
other) -- strong scaling: `value` at N GPUs over `value` at 1 GPU is the speed-up north_star asks for.  `--weak`
gives every GPU its own `--cells-per-gpu` x `--npc` block instead (default configs[1]: 4096 cells, 2441 + 2441 per
cell).

What the line's top level measures: the CONTRACT RUN -- the general kernel with every attribute of every particle
streamed (SURVEY.md 8d's 80 B per update: read cell,x,ux,uy,uz,w; write cell,x,ux,uy,uz) and rho deposited on every
step, whatever the data would allow to skip.  The library's default path (which, for this electrostatic input, neither
streams uy/uz nor the uniform weight, and gives the same bits) is measured beside it on the same population as
`default_path`.  Every `roofline` block is priced with the algorithmic bytes of the kernel FORM that ran in its timed
region (`bytes_model`; the library reports the form of each launch, opal_prof_forms), so `frac` <= 1 by construction and
`algorithmic_GBps_over_whole_step` <= 8000.  The kernel's mean duration is measured with HIP events on the
simulation's stream around a 1-in-5 sample of its launches in the timed region; `traffic` is the PMC-measured HBM bytes
per launch -- measured ON THE BOX THE RUN IS ON at N = 1: before this process touches the GPU it runs two 5-step child
copies of its own command under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (measure_traffic_live, ~30 s; the
top level's kernel and `default_path`'s); for the other blocks, or where the passes cannot run (`config.traffic_live` says
why), the constant from profiles/traffic.json, used only if that entry was taken for this workload, this kernel instantiation
and these kernel sources.  `roofline.frac` prices a launch with the form's ALGORITHMIC bytes, `roofline.frac_physical` with the
bytes that crossed the HBM interface (an attribute no lane of a wave changed is not written back: fewer than 80).  A `--steps`
below 200 is followed by a second timed region of 200 steps (`steps_long`).  At N = 1 the line also carries configs[1]
(`secondary`), configs[3]'s physics inside the laser-plasma interaction (`laser_plasma_interaction`) and configs[4]
(`colliding_beams`: examples/colliding_beams.yaml's physics with photon emission through the collision, electrons + photons
pushed, each kernel priced with its own bytes), measured the same way in the same process.  At N > 1 the line carries
`laser_plasma_interaction` for configs[3] as named (1e8 particles, `balance: true`) and `colliding_beams` on the same N GPUs,
run after the two-stream measurement is complete and each under a deadline, so that nothing in them can cost the line;
`config.gpus_seen` / `config.ranks` say which cards took part (host, HIP ordinal, PCI bus id, link kind) whatever the
transport, `per_rank` what each rank did (its own clock, particles, push time, re-binnings, its link's room and growths).

`python bench.py --gpus N` needs no launcher: with WORLD_SIZE unset the parent -- which never touches a
GPU -- starts N ranks of itself as child processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, rendezvous on
127.0.0.1), relays rank 0's JSON line and exits non-zero if any rank failed.  Under torch.distributed.run
the same file is simply one rank.  With fewer GPUs than ranks (a 1-GPU box) the ranks share the card: a rehearsal of
the decomposed step, labelled as such in `config`.

`cpu_baseline` times the CPU oracle in its reference-algorithm mode (AoS structs, the population sorted by cell every
step with a restatement of rayon's par_sort_unstable_by_key, per-thread private current arrays + serial reduce, serial
field solve; a port of the reference's algorithm, not the opal binary) on a bounded sample of the workload on this
box's host cores: S subdomains x T threads = the cores this process may use (affinity AND cgroup quota), the
reference's MPI x Rayon layout, with the per-phase breakdown; plus the 1 x 1 point for BASELINE configs[0].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C3_CELLS, C3_NPC = 32768, 1526       # BASELINE configs[2] (north_star's 1e8-particle two-stream)
C2_CELLS, C2_NPC = 4096, 2441        # BASELINE configs[1]
BYTES_PER_UPDATE = 80.0   # read cell,x,u[3],w = 44 B; write cell,x,u[3] = 36 B
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PREHEAT_SECONDS = 0.3

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

# BASELINE configs[0]: the parameters of the reference's examples/two_stream.yaml (1000 cells, 100 electrons
# per cell, no ions) -- the reference's own CPU-runnable case, timed here on 1 subdomain x 1 thread
C1_YAML = """---
control:
 dx: 500
 nx: 1000
 xmin: 0.0
 start: 0.0
 end: 0.1
 current_deposition: true
 n_outputs: 20
qed:
 photon_emission: false
 photon_absorption: false
electrons:
 npc: 100
 ne: 20.0
 ux: drift_u * (1.0 + 0.001 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)
 uy: 0.0
 uz: 0.0
 output: x:px
ions:
 npc: 0
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


# BASELINE configs[4]: examples/colliding_beams.yaml as shipped (gamma = 1000 electron beam against an a0 = 20 four-cycle
# pulse, current_deposition off, stochastic photon emission on, photons kept within 50 mrad; interactions.rs:45-107,
# photon.rs:150-183) with npc scaled: 500 cells x 20 000 = 1e7 electrons, which radiate ~4.7e7 photons in the collision.
# The pulse meets the beam after ~1750 steps and the collision is over ~750 steps later: that stretch is the timed region.
COLLIDING_BEAMS_YAML = """---
control:
 dx: 0.01*micro
 nx: 4000
 xmin: -15*micro
 start: -24.0e-6/c
 end: 6.0e-6/c
 current_deposition: false
 n_outputs: 5
 balance: false
qed:
 photon_emission: true
 photon_absorption: false
 photon_angle_max: 50 * milli
electrons:
 npc: {npc}
 ne: S * a0 * critical(omega) * step(x,xmin,xmax)
 ux: -1000.0 * (1.0 + 0.01 * nrand)
 uy: 0.0
 uz: 0.0
 output: [x, chi, x:chi, energy, x:energy]
ions:
 npc: 0
photons:
 npc: 0
 output: [x, energy, energy:(log;energy), longitude:latitude, longitude:latitude:(energy)]
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
 xmin: (20 - 2.5) * micro
 xmax: (20 + 2.5) * micro
"""
CB_NPC, CB_WARM, CB_STEPS, CB_SEG = 20000, 1750, 750, 50
# algorithmic bytes per update: the electrons' advance with chi and the optical depth tracked reads cell,x,ux,uy,uz,w,tau
# (52 B; 44 B when the uniform weight comes from the launch arguments) and writes cell,x,ux,uy,uz,chi,tau (52 B): SURVEY.md
# 8d's 80 B + 8 B per extra read or written; a photon's push reads cell,x,kx,ky,kz (36 B) and writes cell,x (12 B)
CB_BYTES_ELECTRON, CB_BYTES_PHOTON = 104.0, 48.0


REHEAT_STEPS = int(os.environ.get("OPAL_BENCH_REHEAT", "40"))
REHEAT_SECONDS = 0.05
LONG_STEPS = 200   # a second timed region of this length follows a shorter --steps (the driver's 20): `steps_long`


def lp_link_capacity(npc):
    """Room a laser-plasma link is CREATED with for the leavers of one step through one cut, per species: what stands within
    8 cells of a cut on both sides at the initial density (16 cells' worth; 20 are reserved).  The hole-boring front piles the particles up
    (scripts/archive/lp_pileup_probe.py: up to 47 x the initial electrons and 93 x the ions in one cell); rounds 2-3 created
    every link for that (128 npc: 0.5 GB per subdomain at npc = 10 000) and still called an overflow an error.  Round 4:
    the edge a pile-up approaches GROWS -- from the zone counts both of its ends hold, before the push that could overflow
    it (link.cpp: link_grow) -- and the others stay small."""
    return int(os.environ.get("OPAL_BENCH_LP_CAPACITY") or max(4096, 20 * npc))


def lp_fixed_capacity(npc):
    """... and the room of a laser-plasma link that cannot grow (torch.distributed's: its buffers are torch tensors): the
    worst pile-up measured at a cut, as rounds 2-3 sized every link."""
    return int(os.environ.get("OPAL_BENCH_LP_CAPACITY") or max(65536, 128 * npc))


def src_sha():
    """Hash of the kernel sources the PMC traffic figures were taken with (profiles/traffic.json records it):
    the code of the push and field kernels with comments and white space taken out."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in ("particles.hip", "pic_device.hpp", "grid.hip", "grid_device.hpp"):
        text = open(os.path.join(ROOT, "opal_amd", "csrc", f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]

def bytes_model(forms, n_e, n_i, unfused):
    """Algorithmic bytes per particle-update of the kernel FORM(S) that ran in the timed region:
    cell, x, ux are read and written by every form (40 B); uy, uz are read and written (32 B) by every form but the
    electrostatic launch; the weight is read (8 B) unless the launch takes it from its arguments.  The general
    form with streamed weights is SURVEY.md 8d's 80 B.  Write-back elision and skipped rho are not byte-model
    changes (rho lives in LDS; an attribute no lane changed is still part of what the form has to handle)."""
    tot = sum(forms.values())
    if not tot:
        return None
    n = max(n_e + n_i, 1)
    acc, parts = 0.0, []
    for f, cnt in sorted(forms.items()):
        es, single = bool(f & 1), bool(f & 16)
        base = 40.0 + (0.0 if es else 32.0) + (8.0 if unfused else 0.0)   # unfused advance also writes prev_x
        if single:
            w = 0.0 if f & 12 else 8.0
        else:
            w = ((0.0 if f & 4 else 8.0) * n_e + (0.0 if f & 8 else 8.0) * n_i) / n
        acc += cnt * (base + w)
        parts.append({"form": ("electrostatic" if es else "general") + (", rho deposited" if f & 2 else ", rho skipped")
                              + ("" if w == 8.0 else ", weight from the launch arguments" if w == 0.0 else ", one species' weight from the launch arguments")
                              + (", single species" if single else ""),
                      "launches": cnt, "bytes_per_update": base + w})
    dom = max(forms.items(), key=lambda kv: kv[1])[0]
    mode = "0" if unfused else "1"
    name = (f"k_push<{'0' if n_e else '1'}, {mode}, true, 0, false>" if dom & 16 else
            f"k_push_pair<{mode}, {'true' if dom & 2 else 'false'}, {'true' if dom & 1 else 'false'}>")
    return {"bytes_per_update": acc / tot, "forms": parts, "kernel": name,
            "what": "read cell,x,ux + write cell,x,ux = 40 B; + uy,uz read and written = 32 B unless the launch is "
                    "electrostatic; + w read = 8 B unless it comes from the launch arguments (general form, streamed "
                    "weights: SURVEY.md 8d's 44 + 36 = 80 B)"}


def cpu_quota():
    """CPUs' worth of time the cgroup this process runs in may use (None: unlimited)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # cgroup v2: "max 100000" or "1600000 100000"
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def host_cpu():
    """(model name, physical cores this process may run on, logical cpus it may run on)."""
    allowed = sorted(os.sched_getaffinity(0))
    model, cores, cur = "unknown", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur.get("processor") in allowed:
                    cores.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
                cur = {}
                continue
            k, v = [s.strip() for s in line.split(":", 1)]
            if k == "model name":
                model = v
            elif k in ("processor", "physical id", "core id"):
                cur[k] = int(v)
        if cur.get("processor") in allowed:
            cores.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    return model, (len(cores) or len(allowed)), len(allowed)


def _oracle_world(oa, o, yaml_text, dx, nx, S):
    """Initial state from the product's loader, split into S in-process subdomains of AoS populations
    (what GridDesign::unbalanced + Population::new give each MPI rank of the reference)."""
    import numpy as np
    sim = oa.Simulation(yaml_text=yaml_text)
    dt = sim.dt
    se = sim.electrons.soa()
    si = sim.ions.soa() if sim.ions is not None and len(sim.ions) else None
    del sim
    nxs, off = o.design_unbalanced(S, nx, dx)
    grids, pes, pis = [], [], ([] if si is not None else None)
    lo = 0
    for r in range(S):
        grids.append(o.Grid(r, S, o.BDY_INTERNAL, o.BDY_INTERNAL, 0.0, dx, nxs, off))
        hi = lo + int(nxs[r])
        for src, species, dst in ((se, o.ELECTRON, pes), (si, o.ION, pis)):
            if src is None:
                continue
            m = (src["cell"] >= lo) & (src["cell"] < hi)
            p = o.Population.from_soa(species, src["cell"][m].astype(np.int64) - lo, src["x"][m], src["ux"][m],
                                      src["uy"][m], src["uz"][m], src["w"][m], dx, dt)
            if species == o.ION:
                p.set_charge_state(1.0, 1.0, 9.109383e-31 / 1.672622e-27)
            dst.append(p)
        lo = hi
    return grids, pes, pis, dt


def _time_oracle(o, grids, pes, pis, dt, threads, target_seconds, max_steps):
    n = sum(len(p) for p in pes) + (sum(len(p) for p in pis) if pis else 0)
    t0 = time.perf_counter()
    bad = o.world_run_periodic(grids, pes, pis, True, 0.0, dt, 2, threads)
    per_step = (time.perf_counter() - t0) / 2
    steps = int(max(3, min(max_steps, target_seconds / max(per_step, 1e-6))))
    o.phase_times()
    t0 = time.perf_counter()
    bad += o.world_run_periodic(grids, pes, pis, True, 2 * dt, dt, steps, threads)
    el = time.perf_counter() - t0
    assert bad == 0
    ph = o.phase_times()
    S = len(grids)
    # push, sort and the two deposit phases are summed over the subdomains, which run side by side
    per_step_ms = {k: 1e3 * v / steps / (S if k in ("push", "sort", "deposit_scatter", "deposit_reduce") else 1) for k, v in ph.items()}
    return n, steps, n * steps / el, per_step_ms, 1e3 * el / steps


def cpu_baseline(oa, cells, npc, target_seconds=14.0):
    """Times the oracle in reference-algorithm mode on a bounded sample of the workload: 4096 cells at the
    FULL particles-per-cell (what one of configs[2]'s 8 ranks holds, or all of configs[1]) as S subdomains x
    T threads = the cores this process may use (the reference's MPI x Rayon layout); then configs[0] on 1 x 1."""
    from oracle import oracle as o

    o.build()
    o.set_sort_mode(1)   # the AoS structs sorted by cell every step with the restatement of rayon's par_sort_unstable_by_key
    o.set_parallel_subdomains(True)
    try:
        model, phys, logical = host_cpu()
        quota = cpu_quota()
        cores = max(1, min(o.max_threads(), phys, int(quota) if quota else phys))
        S = 8 if cores >= 16 else 1
        T = max(1, cores // S)
        dx = 500.0
        cells_s = min(cells, 4096)
        grids, pes, pis, dt = _oracle_world(oa, o, YAML.format(nx=cells_s, npc=npc), dx, cells_s, S)
        n, steps, rate, phases, step_ms = _time_oracle(o, grids, pes, pis, dt, T, target_seconds, 200)
        del grids, pes, pis
        g1, pe1, _, dt1 = _oracle_world(oa, o, C1_YAML, dx, 1000, 1)
        n1, steps1, rate1, phases1, step1_ms = _time_oracle(o, g1, pe1, None, dt1, 1, 5.0, 2000)
    finally:
        o.set_sort_mode(0)
        o.set_parallel_subdomains(False)
    return {
        "value": rate,
        "unit": "particle-updates/s",
        "cores": S * T,
        "kind": "port",
        "cpu_model": model,
        "physical_cores": phys,
        "logical_cpus": logical,
        "cgroup_cpu_quota": quota,
        "S": S,
        "T": T,
        "ms_per_step": step_ms,
        "ms_per_step_by_phase": phases,
        "sort_share_of_step": phases["sort"] / step_ms,
        "sample": f"two_stream e-+e+, {cells_s} of {cells} cells at the full {npc}+{npc} per cell ({n} particles), "
                  f"{steps} steps; CPU restatement of opal's algorithm (C oracle, not the opal binary): AoS structs, "
                  f"the whole population sorted by cell every step with a restatement of rayon's par_sort_unstable_by_key "
                  f"(pattern-defeating quicksort, parallel joins), per-thread private current arrays + serial reduce, "
                  f"serial field solve per subdomain; {S} subdomains x {T} OpenMP threads (MPI x Rayon layout) = the "
                  f"{S * T} cores this process may use (affinity {logical} logical CPUs, cgroup quota {quota})",
        "config0_1x1": {
            "value": rate1,
            "unit": "particle-updates/s",
            "cores": 1,
            "S": 1,
            "T": 1,
            "ms_per_step_by_phase": phases1,
            "sort_share_of_step": phases1["sort"] / step1_ms,
            "sample": f"BASELINE configs[0] (examples/two_stream.yaml's parameters: 1000 cells, {n1} electrons), "
                      f"{steps1} steps, same restatement on 1 subdomain x 1 thread",
        },
    }


LIVE_TRAFFIC = {}  # "<workload_key>|<kernel instantiation>" -> what measure_traffic_live() found on THIS box, in THIS run


def measure_traffic_live(argv, deadline_s=150.0):
    """HBM bytes per launch of the push kernels, measured on the box this run is on, before the run itself touches the
    GPU: two child processes, `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, as
    MI355X_MICROARCH.md prescribes; the program after `--` is python3 itself), each a 5-step run of this very command
    with the side blocks off.  Counters cannot be read from inside the process, and a constant from profiles/ says
    nothing about the box the driver landed on -- boxes differ by a few per cent in clock, not in bytes, which is the
    point of measuring it: `roofline.traffic` then comes from the same silicon as `ms_per_step`.
    Returns a note for the line; fills LIVE_TRAFFIC.  Every failure (no rocprofv3, a refused profiler, a time-out) is a
    note and nothing else: bench.py then falls back to profiles/traffic.json, labelled as such."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None:
        return "live PMC passes skipped: no rocprofv3 on this box"
    drop_with_value = {"--steps", "--warmup", "--gpus"}
    child_args, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in drop_with_value:
            skip = True
            continue
        if a.split("=")[0] in drop_with_value:
            continue
        child_args.append(a)
    child_args += ["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--no-laser-plasma",
                   "--no-colliding-beams", "--no-live-traffic"]
    push = re.compile(r"k_push(?:_pair)?<[^>]*>")
    tmp = tempfile.mkdtemp(prefix="opal_pmc_")
    env = dict(os.environ, TMPDIR=tmp, OPAL_BENCH_PMC_CHILD="1")
    found, lines = {}, {}
    t0 = time.time()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-include-regex", "k_push", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "c",
                   "--", sys.executable, os.path.abspath(__file__)] + child_args
            left = deadline_s - (time.time() - t0)
            if left < 20.0:
                return f"live PMC passes gave up: {counter} pass not started, {deadline_s:.0f} s budget spent"
            # (its own process group: a pass that runs out of time is ended with everything it started -- the profiler AND
            # the copy of this program under it, which would otherwise keep the card busy beside the timed run)
            proc = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                stdout, _ = proc.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)  # exactly the group started above
                except ProcessLookupError:
                    pass
                proc.wait()
                raise
            if proc.returncode != 0:
                return f"live PMC passes failed: the {counter} pass exited with status {proc.returncode}"
            for ln in reversed(stdout.decode(errors="replace").strip().splitlines()):
                if ln.startswith("{"):
                    lines[counter] = json.loads(ln)
                    break
            path = None
            for root, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(root, f)
            if path is None or counter not in lines:
                return f"live PMC passes failed: the {counter} pass left no counter file"
            acc = {}
            for row in csv.DictReader(open(path)):
                m = push.search(row["Kernel_Name"])
                if m and row["Counter_Name"] == counter:
                    acc.setdefault(m.group(0), []).append(float(row["Counter_Value"]))
            # (only the launches of the child's timed steps: the last 5 of each instantiation -- its pre-heat and warm-up
            # launches run on arrays the loader has just written and alternate forms)
            found[counter] = {k: (sum(v[-5:]) / len(v[-5:]), len(v[-5:])) for k, v in acc.items()}
    except subprocess.TimeoutExpired:
        return f"live PMC passes gave up: a pass ran into the {deadline_s:.0f} s budget"
    except Exception as e:  # a profiler that is not allowed here, a changed file layout ...: never the run's problem
        return f"live PMC passes failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    cfg = lines["FETCH_SIZE"]["config"]
    wkey, n = cfg["workload_key"], cfg["particles_total"]
    for k in sorted(set(found["FETCH_SIZE"]) & set(found["WRITE_SIZE"])):
        # KiB; gfx950: FETCH_SIZE counts half of a coalesced streaming read, WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM)
        rd = 2.0 * found["FETCH_SIZE"][k][0] * 1024.0
        wr = found["WRITE_SIZE"][k][0] * 1024.0
        # a single-species launch (k_push<species, ...>) covers one of the two equal species of the workload
        n_cov = n / 2 if k.startswith("k_push<") else n
        LIVE_TRAFFIC[f"{wkey}|{k}"] = {"hbm_bytes_per_particle_update": (rd + wr) / n_cov, "read": rd / n_cov, "write": wr / n_cov,
                                       "launches": [found["FETCH_SIZE"][k][1], found["WRITE_SIZE"][k][1]]}
    return (f"live PMC passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two 5-step runs of this command on this box "
            f"before the timed run ({time.time() - t0:.0f} s): " +
            "; ".join(f"{k.split('|')[1]} {v['hbm_bytes_per_particle_update']:.1f} B/update" for k, v in sorted(LIVE_TRAFFIC.items())))


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank run.  Makes no GPU call (imports neither torch nor the library)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        time.sleep(0.05)
    if failed is None:
        failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:      # the exact children started above, nothing else
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        print(f"[bench] rank {failed} exited with status {procs[failed].returncode}", file=sys.stderr)
        return 1
    line = procs[0].stdout.read().decode().strip().splitlines()
    if not line:
        print("[bench] rank 0 printed no result", file=sys.stderr)
        return 1
    print(line[-1], flush=True)
    return 0


def main():
    # the CPU baseline's OpenMP teams (subdomains x threads, plus the sort's task teams) must not spin when idle: a
    # spinning team member takes the core another team's worker needs (read by the OpenMP runtime when it is loaded)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] measurement of the N=1 line")
    ap.add_argument("--cells", type=int, default=None, help="total cells of the (strong-scaling) two-stream domain")
    ap.add_argument("--npc", type=int, default=None, help="macro-particles per cell and species")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every GPU gets --cells-per-gpu cells")
    ap.add_argument("--cells-per-gpu", type=int, default=None,
                    help=f"implies --weak; default {C2_CELLS} (with npc {C2_NPC}: BASELINE configs[1] per GPU)")
    ap.add_argument("--unfused", action="store_true", help="time the reference's operator sequence instead")
    ap.add_argument("--elision", choices=["off", "on"], default="off",
                    help="off (default): the headline is the general kernel with every attribute of every particle "
                         "streamed (SURVEY.md 8d's 80 B per update, rho every step), and the library's default path is "
                         "reported beside it as `default_path`; on: the headline is the library's default path")
    ap.add_argument("--no-elision", action="store_true", help="same as --elision off without the default_path measurement")
    ap.add_argument("--no-default-path", action="store_true", help="skip the default_path measurement")
    ap.add_argument("--no-laser-plasma", action="store_true", help="skip the laser-plasma interaction block of the N=1 line")
    ap.add_argument("--no-colliding-beams", action="store_true", help="skip the colliding-beams (configs[4]) block")
    ap.add_argument("--workload", choices=["two_stream", "laser_plasma"], default="two_stream",
                    help="two_stream (default; BASELINE configs[2], strong scaling) or laser_plasma "
                         "(configs[3]'s physics, --npc per cell in 5000 of 20000 cells, strong scaling)")
    ap.add_argument("--self-exchange", action="store_true",
                    help="N=1 rehearsal of the multi-GPU step: the lone periodic subdomain sends its halo and leavers "
                         "to itself through RCCL (pack kernels, RCCL send/recv, unpack) instead of wrapping in place")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic on this box (N = 1)")
    args = ap.parse_args()

    # roofline.traffic from THIS box: two short child runs under rocprofv3's counters, before this process touches the GPU
    live_note = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_live_traffic and not under_profiler
            and not os.environ.get("OPAL_BENCH_PMC_CHILD") and not args.unfused and not args.self_exchange):
        live_note = measure_traffic_live(sys.argv[1:])
        print("[bench] " + live_note, file=sys.stderr)

    if args.no_elision:
        args.elision, args.no_default_path = "off", True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Fewer GPUs than ranks (a 1-GPU box): the ranks share the card and exchange through host buffers over
    # gloo -- a rehearsal of the decomposed step.  The driver's multi-GPU runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("OPAL_BENCH_BACKEND") or ("nccl" if ndev >= world else "gloo")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    oa.set_device(dev_index)
    # peer: mailboxes in the neighbours' device memory written by the step's own kernels (falls back to rccl, then to
    # torch.distributed); a rehearsal with fewer GPUs than ranks uses the same mailboxes between the processes that
    # share a card (falls back to host-staged gloo messages)
    link_kind = os.environ.get("OPAL_BENCH_LINK", "peer")

    weak = args.weak or args.cells_per_gpu is not None
    if args.workload == "laser_plasma":
        npc = args.npc if args.npc is not None else 2000
        nx = 20000
        link_capacity = lp_link_capacity(npc)
    else:
        if weak:
            cells_per_gpu = args.cells_per_gpu or C2_CELLS
            npc = args.npc if args.npc is not None else C2_NPC
            nx = cells_per_gpu * world
        else:
            nx = args.cells or C3_CELLS
            npc = args.npc if args.npc is not None else C3_NPC
        # two-stream flux through a cut: npc <|v_x|> dt / dx ~ 0.009 npc per step.  The room is what stands within 8 cells
        # of a cut on both sides (16 npc: the bound an edge is held to, link.cpp) and a margin, so that no edge ever has
        # to grow in this uniform plasma; what travels is the leavers only
        link_capacity = max(4096, 20 * npc)

    dist = None
    link = None
    if world > 1 or args.self_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        lp_open = args.workload == "laser_plasma"
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            link_device = torch.device("cuda", dev_index)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            link_device = torch.device("cpu")
        from opal_amd.exchange import make_link
        order = {"peer": ("peer", "rccl", "torch"), "rccl": ("rccl", "torch"), "torch": ("torch",), "gloo": ("torch",)}[link_kind]
        link, link_kind = make_link(rank, world, link_device, link_capacity, stream=torch.cuda.current_stream().cuda_stream,
                                    to_left=not (lp_open and rank == 0), to_right=not (lp_open and rank == world - 1),
                                    prefer=order, log=lambda m: print(m.replace("[opal_amd]", "[bench]"), file=sys.stderr),
                                    fixed_capacity=lp_fixed_capacity(npc) if lp_open else None)
        if link_kind == "rccl" and os.environ.get("OPAL_BENCH_FIXED"):  # diagnostic: ship all slots every step
            link.set_bounded(False)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    rccl_ranks = None
    if link is not None and hasattr(link, "rccl_ranks"):
        try:
            rccl_ranks = link.rccl_ranks()
        except Exception as e:
            print(f"[bench] {e}", file=sys.stderr)

    stream = torch.cuda.current_stream().cuda_stream

    # Which cards took part, whatever the transport: every rank reports the host it runs on, its HIP ordinal and that
    # device's PCI bus id (hipDeviceGetPCIBusId), the link kind it ended up with and -- peer links -- that its mailboxes'
    # checked self-test rounds passed (make_link keeps the kind only then) and how many exchanges they have made.
    import socket
    bus, dev_name = oa.device_identity(dev_index)
    me = {"rank": rank, "host": socket.gethostname(), "hip_device": dev_index, "pci_bus_id": bus, "device": dev_name,
          "link": link_kind if link is not None else None}
    if link is not None and link_kind == "peer":
        try:
            mem, nex = link.status()
            me.update(peer_self_test="passed (8 checked + 32 back-to-back + 1 checked halo exchanges per neighbour)",
                      peer_mailbox_memory=mem, peer_exchanges_so_far=nex)
        except Exception as e:  # noqa: BLE001
            me["peer_self_test"] = f"status failed: {e}"
    ranks_seen = [me]
    if world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
    gpus_seen = len({(r["host"], r["pci_bus_id"]) for r in ranks_seen})

    def link_kind_desc():
        return {"peer": "peer-mapped mailboxes over xGMI, written and awaited by the step's own kernels (no transport call)",
                "rccl": "native RCCL send/recv on the simulation stream", "torch": "torch.distributed P2P (RCCL)"}.get(link_kind, link_kind)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def measure(sim, steps, warmup, preheat=True, workload_key="two_stream"):
        """Pre-heat (untimed, >= PREHEAT_SECONDS of steps so that clocks and caches are those of a running job),
        W warm-up steps, then exactly K timed steps between barriers; max over ranks."""
        barrier()
        done = 0
        per_step = None
        if preheat:
            t0 = time.perf_counter()
            sim.step(10)
            sim.sync()
            per_step = all_max((time.perf_counter() - t0) / 10)
            done = 10
            more = int(min(5000, max(0.0, PREHEAT_SECONDS - 10 * per_step) / max(per_step, 1e-7)))  # same on every rank
            if more:
                sim.step(more)
                sim.sync()
                done += more
        tw0 = time.perf_counter()
        sim.step(warmup)
        sim.sync()
        if warmup >= 5:
            per_step = all_max((time.perf_counter() - tw0) / warmup)  # (same on every rank: it sizes the burst below)
        # the particles this rank holds NOW (open boundaries delete leavers, subdomains trade them): counted after the
        # warm-up, and again after the timed steps -- the smaller number is the one the rate is quoted for
        n_e = len(sim.electrons) if sim.electrons is not None else 0
        n_i = len(sim.ions) if sim.ions is not None else 0
        oa.prof_reset()
        if REHEAT_STEPS:
            # the counting above left the GPU idle for a few hundred microseconds, and its clocks take tens of
            # milliseconds to come back (a 20-step timed region measured 8 % slower per step than a 200-step one):
            # a last untimed burst of >= 50 ms, so that all that separates the timed steps from running ones is the
            # barrier itself
            burst = REHEAT_STEPS if per_step is None else min(2000, max(REHEAT_STEPS, int(REHEAT_SECONDS / max(per_step, 1e-7))))
            sim.step(burst)
            done += burst
        barrier()
        # HIP events around every 5th launch of each tracked kernel (an odd stride samples both species);
        # timing every launch costs ~10 % of the step because each event record breaks back-to-back dispatch
        oa.prof_enable(5)
        lk_wait = getattr(getattr(sim, "_bench_link", None), "wait", None)
        if lk_wait:
            lk_wait(reset=True)  # (drains the stream: in front of the barrier's own wait)
            barrier()
        t0 = time.perf_counter()
        sim.step(steps)
        sim.sync()
        barrier()
        elapsed_local = time.perf_counter() - t0
        elapsed = all_max(elapsed_local)
        oa.prof_enable(0)
        n_e_after = len(sim.electrons) if sim.electrons is not None else 0
        n_i_after = len(sim.ions) if sim.ions is not None else 0
        n_total = int(min(all_sum(n_e + n_i), all_sum(n_e_after + n_i_after)))
        if world == 1:
            n_e, n_i = min(n_e, n_e_after), min(n_i, n_i_after)
        n_local = n_e + n_i
        kname = "fused" if not args.unfused else "advance"
        launches, total_ms = oa.prof_get(kname)
        sort_launches, sort_ms = oa.prof_get("sort")
        field_launches, field_ms = oa.prof_get("field")
        forms = oa.prof_forms()
        model = bytes_model(forms, n_e, n_i, args.unfused)
        bpu = model["bytes_per_update"] if model else BYTES_PER_UPDATE
        # the dominant kernel covers this rank's particles once per step, in one launch (species paired) or two
        launches_total = oa.prof_seen(kname)
        per_launch = n_local * steps / max(launches_total, 1)
        avg_ms = total_ms / max(launches, 1)
        achieved = bpu * per_launch / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        value = n_total * steps / elapsed
        roof = {
            "bound": "hbm",
            "kernel": model["kernel"] if model else None,
            "bytes_model": model,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_physical": None,
            "traffic": None,
            "launches_timed": launches,
            "launches_total": launches_total,
            "avg_launch_ms": avg_ms,
            "particle_updates_per_launch": per_launch,
            "algorithmic_bytes_per_launch": bpu * per_launch,
            # the cross-check a reader can make without the kernel time: the same bytes over the WHOLE step
            "algorithmic_GBps_over_whole_step": bpu * n_total * steps / elapsed / 1e9 / world,
            "other_kernels_avg_ms": {"field_solve": field_ms / max(field_launches, 1), "rebin": sort_ms / max(sort_launches, 1)},
            "rebins_in_timed_region": oa.prof_seen("sort"),
        }
        # PMC-measured HBM bytes per particle-update: separate rocprofv3 --pmc passes of this command
        # (scripts/profile_bench.sh; the counters cannot be read from inside the process).  An entry is used only if
        # it was taken for THIS workload and THIS kernel instantiation with the kernel sources as they are now.
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            tj = json.load(open(tpath)).get(f"{workload_key}|{roof['kernel']}") if os.path.exists(tpath) else None
        except Exception:
            tj = None
        live = LIVE_TRAFFIC.get(f"{workload_key}|{roof['kernel']}")
        if live is not None:
            # measured minutes ago on this very box (measure_traffic_live): the committed constant is only the cross-check
            roof["traffic"] = live["hbm_bytes_per_particle_update"] * per_launch
            roof["traffic_source"] = ("measured by this run on this box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate "
                                      f"5-step passes of this command before the timed run ({live['launches'][0]} / "
                                      f"{live['launches'][1]} launches of this kernel counted; FETCH_SIZE doubled as the gfx950 guide prescribes)")
            roof["traffic_bytes_per_update"] = live["hbm_bytes_per_particle_update"]
            roof["traffic_read_write_bytes_per_update"] = [live["read"], live["write"]]
            if tj is not None and tj.get("src_sha") == src_sha():
                roof["traffic_committed_bytes_per_update"] = tj["hbm_bytes_per_particle_update"]  # (profiles/traffic.json)
                rel = abs(live["hbm_bytes_per_particle_update"] / tj["hbm_bytes_per_particle_update"] - 1.0)
                if rel > 0.03:
                    roof["traffic_disagreement"] = (f"live and committed bytes per update differ by {100 * rel:.1f} %: the live "
                                                    f"figure is used (same box as ms_per_step), the committed one is "
                                                    f"{tj.get('source')}")
            roof["traffic_GBps"] = roof["traffic"] / (avg_ms * 1e-3) / 1e9 if launches else None
            roof["frac_traffic"] = roof["traffic_GBps"] / HBM_PEAK_GBS if launches else None
            roof["frac_physical"] = roof["frac_traffic"]
            roof["traffic_note"] = ("bytes that crossed the HBM interface per launch; below the algorithmic bytes where the "
                                    "write-back of attributes no lane of a wave changed (uy, uz, most cell indices in an "
                                    "electrostatic run) is skipped")
        elif tj is None:
            roof["traffic_note"] = f"no PMC entry for {workload_key}|{roof['kernel']} in profiles/traffic.json"
        elif tj.get("src_sha") != src_sha():
            roof["traffic_note"] = (f"profiles/traffic.json entry for {workload_key}|{roof['kernel']} is stale (kernel "
                                    f"sources changed since {tj.get('source')}): not reported")
        else:
            roof["traffic"] = tj["hbm_bytes_per_particle_update"] * per_launch
            roof["traffic_source"] = tj["source"] + " (separate rocprofv3 --pmc passes of this command; not measured in this process)"
            roof["traffic_bytes_per_update"] = tj["hbm_bytes_per_particle_update"]
            roof["traffic_GBps"] = roof["traffic"] / (avg_ms * 1e-3) / 1e9 if launches else None
            roof["frac_traffic"] = roof["traffic_GBps"] / HBM_PEAK_GBS if launches else None
            # the same number under the name a reader looks for: `frac` prices the launch with the form's ALGORITHMIC bytes,
            # `frac_physical` with the bytes that really crossed the HBM interface
            roof["frac_physical"] = roof["frac_traffic"]
            roof["traffic_note"] = ("bytes that crossed the HBM interface per launch; below the algorithmic bytes where the "
                                    "write-back of attributes no lane of a wave changed (uy, uz, most cell indices in an "
                                    "electrostatic run) is skipped")
        per_rank = None
        if world > 1:
            # what each subdomain did: its own clock between the same two barriers (the slowest one IS the step), its
            # particles and its kernel -- imbalance (configs[3]) is otherwise invisible in the whole-job number
            per_rank = [None] * world
            mine = {"rank": rank, "ms_per_step_own_clock": 1e3 * elapsed_local / steps, "particles": n_local,
                    "push_avg_ms": avg_ms, "cells": int(sim.grid.size), "rebins": oa.prof_seen("sort")}
            lk = getattr(sim, "_bench_link", None)
            if lk is not None and hasattr(lk, "growth"):
                cl, cr, ng = lk.growth()
                mine["link_room_left_right"], mine["link_edges_grown"] = [cl, cr], ng
            if lk_wait:
                # how long this subdomain's tail kernel stood waiting for its neighbours' messages, per step: beside
                # push_avg_ms it tells link latency (the same few microseconds everywhere) from imbalance or skew (long on the
                # subdomains with few particles, nothing on the fullest)
                us, nw = lk_wait()
                mine["tail_wait_us_per_step"] = us / max(steps, 1)
                mine["tail_waits"] = nw
            dist.all_gather_object(per_rank, mine)
        out = {"value": value, "elapsed": elapsed, "n_total": n_total, "preheat": done, "roofline": roof, "per_rank": per_rank,
               "ms_per_step": 1e3 * elapsed / steps,
               "frac_of_hbm_roofline_whole_step": value * bpu / (world * HBM_PEAK_GBS * 1e9)}
        if per_rank:
            # the step takes as long as the fullest subdomain's: max / mean of the particles the subdomains held at the end of
            # the timed steps is the factor a perfectly balanced partition would be faster by
            held = [float(p["particles"]) for p in per_rank]
            out["imbalance_max_over_mean"] = max(held) * len(held) / max(sum(held), 1.0)
            st = sim.recut_stats(world)
            if st["checks"]:
                out["recut"] = {"checks": st["checks"], "recuts": st["recuts"], "seconds_in_checks_and_recuts_since_the_start": st["seconds"],
                                "steps_in_which_a_cut_drifted": st["drift_steps"], "drift_cells_per_step_of_the_interior_cuts": st["drift_cells_per_step"],
                                "imbalance_at_last_check": st["imbalance"],
                                "worst_imbalance_at_a_check": st["imbalance_worst"], "cells_per_subdomain": st["sizes"]}
        return out

    def regroup(sim):
        """A re-binning, so that the push workgroups' chunks are cut for the kernel form about to run (the electrostatic
        launch takes chunks twice as long as the general kernel: pop_max_ppt) -- what a run that started in this form
        would have had from its first re-binning on."""
        for p in (sim.electrons, sim.ions):
            if p is not None and len(p):
                p.sort(sim.grid)

    def block(m, what, steps, extra=None):
        out = {"what": what, "value": m["value"], "unit": "particle-updates/s", "particles_total": m["n_total"], "steps": steps,
               "ms_per_step": m["ms_per_step"], "frac_of_hbm_roofline_whole_step": m["frac_of_hbm_roofline_whole_step"],
               "roofline": m["roofline"]}
        if m.get("per_rank"):
            out["per_rank"] = m["per_rank"]
        for k in ("imbalance_max_over_mean", "recut"):
            if k in m:
                out[k] = m[k]
        if extra:
            out.update(extra)
        return out

    def runs_per_block():
        both = os.environ.get("OPAL_BENCH_BOTH_PARTITIONS")
        return 2 if ((backend == "nccl" and ndev >= world) if both is None else both == "1") else 1

    def both_partitions(run, recut):
        """A decomposed block with moving cuts.  Where every rank has a card of its own the block is run a second time on the
        reference's partition (made once, at t = 0) -- what balance is worth can only be measured there, and is not known
        before the first run on such a node: the block is the faster of the two, the other one is kept beside it
        (`other_partition`).  Ranks that share a card take turns on it, so that only the cost of moving cuts would show:
        one run there (OPAL_BENCH_BOTH_PARTITIONS=1 / 0 forces the one or the other).  Both runs share the block's deadline: with
        a card per rank a run takes seconds (23 000 steps of ~0.3 ms), and a block that hangs must not hold the line back longer
        than it did with one."""
        res = run(recut)
        if not recut or runs_per_block() == 1:
            return res
        brief = lambda r, name: {"cuts": name, **{k: r[k] for k in ("value", "unit", "ms_per_step", "imbalance_max_over_mean",
                                                                   "per_rank", "recut") if k in r}}
        static = run(None)
        faster_static = all_max(1.0 if static["value"] > res["value"] else 0.0) > 0.5  # (one verdict on every rank)
        keep, other = (static, res) if faster_static else (res, static)
        keep["cuts"] = "static" if faster_static else "moving"
        keep["other_partition"] = brief(other, "moving" if faster_static else "static")
        return keep

    def recut_spec(env, default):
        """(every, threshold, max_shift) of a block's moving cuts (Simulation.set_recut), or None: OPAL_BENCH_*_RECUT=0 keeps the
        reference's partition, made once at t = 0 (grid/mod.rs:157-206)"""
        v = os.environ.get(env, default)
        if not v or v == "0":
            return None
        e, t, m = v.split(",")
        return int(e), float(t), int(m)

    def timed_only(sim, steps):
        """One more timed region right behind another one (the device is hot): K steps between two barriers."""
        barrier()
        t0 = time.perf_counter()
        sim.step(steps)
        sim.sync()
        barrier()
        el = all_max(time.perf_counter() - t0)
        n = all_sum((len(sim.electrons) if sim.electrons is not None else 0) + (len(sim.ions) if sim.ions is not None else 0))
        return {"steps": steps, "ms_per_step": 1e3 * el / steps, "value": n * steps / el, "unit": "particle-updates/s",
                "what": "the same simulation, the same kernel form, a second timed region of this many steps right behind the "
                        "first: the figure a short `--steps` does not depend on"}

    def colliding_beams(link_cb, what, cb_recut=None):
        """BASELINE configs[4] on this run's ranks: untimed up to the collision, then CB_STEPS timed steps through it in
        segments of CB_SEG (the photon population grows by the step: each segment is credited with the electrons and the
        mean of the photons it began and ended with).  updates = electrons + photons pushed; every kernel of the step
        priced with its own bytes."""
        oa.set_elision(True)
        simc = oa.Simulation(yaml_text=COLLIDING_BEAMS_YAML.format(npc=CB_NPC), rank=rank, numtasks=world, link=link_cb, stream=stream)
        # The beam is 500 cells long and crosses the 4000-cell domain at c: under the reference's partition (made once, at
        # t = 0) it is on one or two subdomains at a time.  The cuts go with it (checked every 32 steps, in which it moves 30
        # cells; a cut moves 48 at the most).  OPAL_BENCH_CB_RECUT=0: the reference's partition.
        # (Not over a link whose buffers are torch's: they stay as created, every plan is held to them, the cuts fall behind the
        # beam and each step ships the whole bound through the host -- 60 ms a step in the rehearsal in which the mailboxes'
        # self-test once failed.)
        if cb_recut and getattr(link_cb, "transport_kind", None) in ("gloo", "torch"):
            cb_recut = None
        if cb_recut:
            simc.set_recut(*cb_recut)
        barrier()
        tw = time.perf_counter()
        if os.environ.get("OPAL_BENCH_CB_VERBOSE"):
            for k in range(CB_WARM // 250):
                simc.step(250)
                simc.sync()
                print(f"[bench] rank {rank} colliding beams warm-up: {250 * (k + 1)} steps, {time.perf_counter() - tw:.2f} s, "
                      f"{len(simc.electrons)} electrons here, link room {link_cb.growth() if link_cb is not None and hasattr(link_cb, 'growth') else None}",
                      file=sys.stderr, flush=True)
        else:
            simc.step(CB_WARM)
            simc.sync()
        oa.prof_reset()
        barrier()
        oa.prof_enable(5)
        upd_e = upd_p = 0.0
        t0 = time.perf_counter()
        n_p0 = simc.photons.live()
        for seg in range(CB_STEPS // CB_SEG):
            simc.step(CB_SEG)
            simc.sync()
            # (live counts from the device's counters: one small copy each; len() would bring the whole cell array to the host -- 40 MB and 1e7
            # comparisons per call, which round 4's first version of this block did inside the timed loop: 0.63 ms/step
            # where the run takes 0.40)
            n_e, n_p1 = simc.electrons.live(), simc.photons.live()
            if os.environ.get("OPAL_BENCH_CB_VERBOSE"):
                print(f"[bench] rank {rank} colliding beams segment {seg}: {n_e} electrons, {n_p1} photons, "
                      f"{time.perf_counter() - t0:.2f} s", file=sys.stderr, flush=True)
            upd_e += float(n_e) * CB_SEG
            upd_p += 0.5 * (n_p0 + n_p1) * CB_SEG
            n_p0 = n_p1
        barrier()
        el_local = time.perf_counter() - t0
        el = all_max(el_local)
        oa.prof_enable(0)
        tot_e, tot_p = all_sum(upd_e), all_sum(upd_p)
        kern = {}
        for key in ("advance", "photon", "radiate", "sort", "field"):
            n_t, ms = oa.prof_get(key)
            kern[key] = {"launches_timed": n_t, "launches_total": oa.prof_seen(key), "avg_launch_ms": ms / max(n_t, 1),
                         "gpu_ms_in_timed_region": ms / max(n_t, 1) * oa.prof_seen(key)}
        gpu_ms = sum(k["gpu_ms_in_timed_region"] for k in kern.values()) or 1.0
        forms = oa.prof_forms()
        w_arg = any(f & 4 for f, c in forms.items() if c)
        be = CB_BYTES_ELECTRON - (8.0 if w_arg else 0.0)
        ke, kp = kern["advance"], kern["photon"]
        # per launch: this rank's electrons once; the photons it held then (mean over the timed launches)
        e_per_launch = upd_e / max(ke["launches_total"], 1)
        p_per_launch = upd_p / max(kp["launches_total"], 1)
        for k, n_pl, b in ((ke, e_per_launch, be), (kp, p_per_launch, CB_BYTES_PHOTON)):
            k["bytes_per_update"] = b
            k["updates_per_launch"] = n_pl
            k["achieved"] = b * n_pl / (k["avg_launch_ms"] * 1e-3) / 1e9 if k["launches_timed"] else 0.0
            k["frac"] = k["achieved"] / HBM_PEAK_GBS
        dom = ke if ke["gpu_ms_in_timed_region"] >= kp["gpu_ms_in_timed_region"] else kp
        res = {"what": what, "value": (tot_e + tot_p) / el, "unit": "particle-updates/s (electrons + photons pushed)",
               "electron_updates_per_s": tot_e / el, "photon_updates_per_s": tot_p / el, "steps": CB_STEPS, "warmup": CB_WARM,
               "ms_per_step": 1e3 * el / CB_STEPS, "electrons_end": int(all_sum(float(len(simc.electrons)))),
               "photons_end": int(all_sum(float(simc.photons.live()))), "t_end": simc.t,
               "roofline": {"bound": "hbm", "kernel": "k_push<0, 0, true, 2, false> (electrons, optical depth + emitter list)"
                            if dom is ke else "k_push_photon<false>", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": dom["frac"], "traffic": None,
                            "traffic_note": "PMC passes of this block: profiles/r04_colliding_beams_* (scripts/profile_cb.sh)",
                            "algorithmic_GBps_over_whole_step": (be * tot_e + CB_BYTES_PHOTON * tot_p) / el / 1e9 / world},
               "kernels": {"electron_push": ke, "photon_push": kp, "k_radiate": kern["radiate"], "rebin": kern["sort"],
                           "field_solve": kern["field"]},
               "k_radiate_share_of_gpu_time": kern["radiate"]["gpu_ms_in_timed_region"] / gpu_ms,
               "bytes_model": f"electron push {be:.0f} B/update (cell,x,u[3],tau" + ("" if w_arg else ",w")
                              + " read; cell,x,u[3],chi,tau written), photon push 48 B/update (cell,x,k[3] read; cell,x written)"}
        if world > 1:
            pr = [None] * world
            dist.all_gather_object(pr, {"rank": rank, "ms_per_step_own_clock": 1e3 * el_local / CB_STEPS,
                                        "electrons": simc.electrons.live(), "photons": simc.photons.live(),
                                        "cells": int(simc.grid.size)})
            res["per_rank"] = pr
            # (an electron costs the step twice what a photon does: 96 against 48 bytes)
            load = [p["electrons"] + 0.5 * p["photons"] for p in pr]
            res["imbalance_max_over_mean"] = max(load) * world / max(sum(load), 1.0)
            res["subdomains_holding_a_tenth_of_their_share_of_the_electrons"] = sum(
                1 for p in pr if p["electrons"] >= 0.1 * sum(q["electrons"] for q in pr) / world)
            st = simc.recut_stats(world)
            res["recut"] = ({"every_threshold_max_shift": list(cb_recut), "checks": st["checks"], "recuts": st["recuts"],
                             "seconds_in_checks_and_recuts_since_the_start": st["seconds"],
                             "steps_in_which_a_cut_drifted": st["drift_steps"], "drift_cells_per_step_of_the_interior_cuts": st["drift_cells_per_step"],
                             "imbalance_at_last_check": st["imbalance"], "worst_imbalance_at_a_check": st["imbalance_worst"]}
                            if cb_recut else "off: the reference's partition, made once at t = 0")
        del simc
        return res

    CB_WHAT = (f"colliding_beams (examples/colliding_beams.yaml physics, BASELINE configs[4]): gamma = 1000 beam, {CB_NPC} per "
               f"cell in 500 of 4000 cells (1e7 electrons) against an a0 = 20 pulse, current deposition off, photon emission on "
               f"(photons within 50 mrad kept), {CB_WARM} untimed steps up to the collision, then {CB_STEPS} timed steps "
               f"through it (the photon population grows from ~0 to ~4.7e7); library default path")

    CONTRACT = ("general kernel, opal_set_elision(0): every attribute of every particle is read (44 B) and the kernel writes back "
                "cell, x, ux, uy, uz (36 B) -- SURVEY.md 8d's 80 algorithmic bytes per update, which `roofline.frac` is priced "
                "with -- except that an attribute no lane of a wave changed is not written back (results identical by "
                "construction): in this electrostatic two-stream run uy, uz and most cell indices are such, so fewer bytes "
                "than 80 cross the HBM interface (`roofline.traffic_bytes_per_update`, PMC-measured) and the PHYSICAL "
                "fraction of the HBM peak is `roofline.frac_physical`; rho deposited every step")
    DEFAULT = ("library default, opal_set_elision(1): what the data make unnecessary is not moved -- an electrostatic run "
               "(Ey = Ez = B = 0, uy = uz = 0, checked on the device) neither streams uy/uz nor gathers more than Ex, a "
               "uniform weight comes from the launch arguments, rho is deposited on the steps after which it can be "
               "observed; results are the general kernel's bit for bit (tests/test_gpu_fullsize.py); its roofline is "
               "priced with ITS form's bytes (bytes_model), not with 80 B")

    if args.workload == "laser_plasma":
        wkey = f"laser_plasma:{npc}"
        sim = oa.Simulation(yaml_text=LASER_PLASMA_YAML.format(npc=npc), rank=rank, numtasks=world, link=link, stream=stream)
        top_recut = recut_spec("OPAL_BENCH_LP_RECUT", "25,1.12,32") if world > 1 else None
        if top_recut:  # the cuts follow the particles (see the laser_plasma_interaction block)
            sim.set_recut(*top_recut)
    else:
        wkey = f"two_stream:{nx}x{npc}"
        text = YAML.format(nx=nx, npc=npc)
        if os.environ.get("OPAL_BENCH_HOT"):  # diagnostic: a hot plasma (u_th = that many mc) -- thousands of leavers per cut and step
            u = os.environ["OPAL_BENCH_HOT"]
            text = text.replace("ux: drift_u * (1.0 + 0.001 * nrand) * (2.0 * floor(urand - 0.5) + 1.0)", f"ux: {u} * nrand") \
                       .replace("uy: 0.0", f"uy: {u} * nrand").replace("uz: 0.0", f"uz: {u} * nrand")
        sim = oa.Simulation(yaml_text=text, rank=rank, numtasks=world, link=link, stream=stream)
    sim._bench_link = link
    sim.set_fused(not args.unfused)
    if args.self_exchange and world == 1:
        sim.grid.set_self_exchange(True)
    # Edges first, exchange on a second stream under the bulk push (opal_sim_set_overlap): worth it only where the exchange
    # takes longer than the small extra launch costs -- which one GPU cannot tell (there the exchange kernel and the bulk push
    # compete for the same CUs).  So the run measures it where it is: 40 steps each way once the job is warm, the slower of
    # the subdomains deciding, and keeps the faster.  OPAL_BENCH_OVERLAP=0 / 1 forces it off / on.
    overlap_note = None
    ov = os.environ.get("OPAL_BENCH_OVERLAP", "auto")
    if ov == "1":
        sim.set_overlap(True)
        overlap_note = {"kept": "on", "why": "OPAL_BENCH_OVERLAP=1"}
    elif ov == "auto" and world > 1 and not args.unfused and args.workload == "two_stream":
        sim.step(30)
        sim.sync()
        timing = {}
        for on in (False, True, False, True):
            sim.set_overlap(on)
            sim.step(20)  # (the first steps after the switch re-bin with host maps, or stop doing so)
            sim.sync()
            barrier()
            t0 = time.perf_counter()
            sim.step(40)
            sim.sync()
            barrier()
            timing.setdefault(on, []).append(all_max(time.perf_counter() - t0) / 40)
        off_ms, on_ms = 1e3 * min(timing[False]), 1e3 * min(timing[True])
        keep = on_ms < 0.98 * off_ms  # (every rank holds the same two numbers: all_max)
        sim.set_overlap(keep)
        overlap_note = {"kept": "on" if keep else "off", "ms_per_step_off": off_ms, "ms_per_step_on": on_ms,
                        "how": "2 x 40 steps each way after a warm-up, max over ranks, best of two; on is kept if 2 % faster"}

    # One population, two measurements.  First the library's default path (elision on: an electrostatic launch can
    # only serve a population nothing but electrostatic launches have pushed), then -- the headline -- the general
    # kernel with every byte of SURVEY.md 8d streamed.  `--elision on` makes the default path the headline instead.
    m_default = None
    half = max(20, args.steps // 2)
    if args.elision == "on":
        m = measure(sim, args.steps, args.warmup, workload_key=wkey)
        t_end = sim.t
    elif args.workload == "laser_plasma":
        # the physics has phases (the warm-up brings the pulse onto the slab): each form gets a simulation of its own,
        # warmed up to the same time -- the contract run first
        oa.set_elision(False)
        m = measure(sim, args.steps, args.warmup, preheat=args.warmup < 1000, workload_key=wkey)
        t_end = sim.t
        oa.set_elision(True)
        if not args.unfused and not args.no_default_path:
            del sim
            sim = oa.Simulation(yaml_text=LASER_PLASMA_YAML.format(npc=npc), rank=rank, numtasks=world, link=link, stream=stream)
            sim.set_fused(True)
            m_default = measure(sim, args.steps, args.warmup, preheat=args.warmup < 1000, workload_key=wkey)
            half = args.steps
    else:
        if not args.unfused and not args.no_default_path:
            m_default = measure(sim, half, args.warmup, workload_key=wkey)
        oa.set_elision(False)
        if m_default is not None or os.environ.get("OPAL_BENCH_REGROUP"):  # (the latter: a diagnostic)
            regroup(sim)
        m = measure(sim, args.steps, args.warmup, workload_key=wkey)  # (pre-heated again: another kernel, another power state)
        if args.steps < LONG_STEPS and not os.environ.get("OPAL_BENCH_NO_LONG"):
            m["steps_long"] = timed_only(sim, LONG_STEPS)
        t_end = sim.t

    if rank == 0:
        if args.workload == "two_stream":
            named = ("BASELINE configs[2]" if (nx, npc) == (C3_CELLS, C3_NPC) and not weak else
                     "BASELINE configs[1] per GPU" if weak and (nx // world, npc) == (C2_CELLS, C2_NPC) else "custom size")
            workload = (f"two_stream, {npc} e- + {npc} e+ per cell, {nx} cells"
                        + (f" ({nx // world} per GPU, weak scaling)" if weak else f" split into {world} subdomain(s)")
                        + f" ({named}), periodic, push+deposit+field, no QED")
        else:
            workload = (f"laser_plasma (hole_boring.yaml physics, BASELINE configs[3]), {npc} e- + {npc} C6+ per "
                        f"cell in 5000 of 20000 cells, laser + absorbing boundaries, t = {t_end:.3e} s at the end of the "
                        f"timed steps")
        if link is None:
            link_desc = "none (one periodic subdomain wraps in place)"
        elif backend != "nccl":
            link_desc = f"REHEARSAL: {world} ranks share {ndev} GPU(s), " + (
                "peer-mapped mailboxes between the processes, written and awaited by the step's own kernels"
                if link_kind == "peer" else "host-staged gloo exchange")
        else:
            link_desc = link_kind_desc() + (", self-exchange (multi-GPU step rehearsed on one GPU)" if args.self_exchange and world == 1 else "")
        out = {
            "metric": "particle-updates/sec",
            "value": m["value"],
            "unit": "particle-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat_steps": m["preheat"] + (m_default["preheat"] if m_default else 0),
            "ms_per_step": m["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak" if (weak and args.workload == "two_stream") else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "workload_key": wkey,
                "kernel_src_sha": src_sha(),
                "particles_total": m["n_total"],
                "cells_total": nx,
                "path": ("fused advance+deposit" if not args.unfused else "advance, clear, deposit (reference order)"),
                "kernel_form": DEFAULT if args.elision == "on" else CONTRACT,
                "parallelism": f"{world} subdomain(s), 1 per GPU" if backend == "nccl" or world == 1 else f"{world} subdomains on {ndev} GPU(s)",
                "link": link_desc,
                "link_particle_capacity": link_capacity if link is not None else None,
                "overlap": overlap_note,
                "rccl_ranks": rccl_ranks,
                "gpus_seen": gpus_seen,
                "ranks": ranks_seen,
                "frac_of_hbm_roofline_whole_step": m["frac_of_hbm_roofline_whole_step"],
                "traffic_live": live_note,
            },
            "roofline": m["roofline"],
        }
        if m.get("steps_long"):
            out["steps_long"] = m["steps_long"]
        if m.get("per_rank"):
            out["per_rank"] = m["per_rank"]
        for k_ in ("imbalance_max_over_mean", "recut"):
            if k_ in m:
                out[k_] = m[k_]
        if m_default is not None:
            out["default_path"] = block(m_default, DEFAULT, half)
    del sim

    def write_line():
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    def guarded(key, deadline, body):
        """Runs one more block of a complete line on all ranks under a watchdog.  The line above it is complete at this
        point: whatever happens in the block -- an error on some subdomain, a transport that never returns -- the line is
        written, with the failure in it as out[key] = {"error": ...}, and every rank leaves with status 0: at once where
        the failure is seen, within seconds elsewhere (the rendezvous store carries the news, which the others' watchdogs
        poll every 2 s: RCCL does not notice a peer that left a collective), at the deadline at the latest."""
        import threading
        finished = threading.Event()
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:  # noqa: BLE001
            store = None
        ABORT = f"opal_bench_abort_{key}"

        def give_up(why, tell=True):
            if not GIVE_UP_ONCE.acquire(blocking=False):
                return
            if tell and store is not None:
                try:
                    store.set(ABORT, f"rank {rank}: {why}")
                except Exception:  # noqa: BLE001
                    pass
            if rank == 0:
                out[key] = {"error": why}
                write_line()
            print(f"[bench] rank {rank}: {key} block abandoned: {why}", file=sys.stderr)
            sys.stderr.flush()
            if rank == 0 and tell:
                time.sleep(6.0)  # the store lives in this process: the others poll it every 2 s
            os._exit(0)

        def watch():
            t_end = time.monotonic() + deadline
            unreachable = 0
            while not finished.wait(2.0):
                if time.monotonic() > t_end:
                    give_up(f"did not finish within {deadline:.0f} s")
                if store is not None:
                    try:
                        if store.check([ABORT]):
                            give_up(store.get(ABORT).decode(), tell=False)
                        unreachable = 0
                    except Exception:  # noqa: BLE001
                        unreachable += 1
                        if unreachable >= 3 and rank != 0:
                            give_up("the rendezvous store is gone: rank 0 has left", tell=False)

        watchdog = threading.Thread(target=watch, daemon=True)
        watchdog.start()
        try:
            res = body()
            if rank == 0:
                out[key] = res
        except BaseException as e:  # noqa: BLE001
            # the other subdomains may be waiting in a collective of the block: no further collective from here
            give_up(f"{type(e).__name__}: {e}")
        finished.set()
        if GIVE_UP_ONCE.locked():  # the watchdog fired while the block was finishing: it ends the process
            time.sleep(3600)

    import threading
    GIVE_UP_ONCE = threading.Lock()
    full_line = (args.workload == "two_stream" and not weak and not args.self_exchange and not args.unfused
                 and (nx, npc) == (C3_CELLS, C3_NPC) and args.elision == "off")
    if world > 1 and full_line and not args.no_laser_plasma:
        # BASELINE configs[3] on the same N GPUs: hole_boring.yaml's physics with 1e8 particles, `balance: true`
        # (subdomains of different sizes, GridDesign::balanced), laser and absorbing ends, timed inside the interaction.
        # A second set of mailboxes (open ends, room for a relativistic front) is set up for it.
        lp_npc, lp_warm, lp_steps = 10000, 21000, 2000

        def lp_run(lp_recut):
            nonlocal link
            link = None
            link3, kind3 = make_link(rank, world, link_device, lp_link_capacity(lp_npc), stream=stream, to_left=rank != 0,
                                     to_right=rank != world - 1, prefer=order, fixed_capacity=lp_fixed_capacity(lp_npc),
                                     log=lambda m_: print(m_.replace("[opal_amd]", "[bench]"), file=sys.stderr))
            if os.environ.get("OPAL_BENCH_LP_FAIL_RANK") == str(rank):  # (rehearsing the failure path)
                raise RuntimeError("failure requested through OPAL_BENCH_LP_FAIL_RANK")
            oa.set_elision(False)
            sim3 = oa.Simulation(yaml_text=LASER_PLASMA_YAML.format(npc=lp_npc), rank=rank, numtasks=world, link=link3, stream=stream)
            sim3._bench_link = link3
            sizes = [None] * world
            dist.all_gather_object(sizes, int(sim3.grid.size))
            if lp_recut and kind3 in ("gloo", "torch"):
                lp_recut = None  # (buffers that belong to torch do not grow: a plan would be held to them -- see colliding_beams)
            if lp_recut:
                sim3.set_recut(*lp_recut)
            m3 = measure(sim3, lp_steps, lp_warm, preheat=False, workload_key=f"laser_plasma:{lp_npc}")
            t3 = sim3.t
            del sim3, link3
            oa.set_elision(True)
            return block(
                m3, f"laser_plasma (hole_boring.yaml physics, BASELINE configs[3]) on "
                    + (f"{world} GPUs" if backend == "nccl" else f"{world} subdomains sharing {ndev} GPU(s) (REHEARSAL)")
                    + f", {lp_npc} e- + {lp_npc} "
                    f"C6+ per cell in 5000 of 20000 cells, balanced decomposition (subdomain sizes at t = 0: {sizes}"
                    + (f"; the cuts then follow the particles: a check every {lp_recut[0]} steps, re-cut above max / mean = "
                       f"{lp_recut[1]}, {lp_recut[2]} cells at a time" if lp_recut else "; kept for the whole run, as the reference does")
                    + f"), {lp_warm} untimed steps to bring the pulse onto the slab, then {lp_steps} timed steps ending at "
                    f"t = {t3:.3e} s; " + CONTRACT, lp_steps, {"link": kind3, "n_gpus": world})

        def lp_body():
            # `balance: true` cuts on the electrons' density at t = 0 (grid/mod.rs:157-206); the pulse then piles the slab up in
            # front of it and the first subdomains empty into the next ones.  The cuts follow (a check every 25 steps, a cut 32
            # cells at a time).  OPAL_BENCH_LP_RECUT=0: the reference's partition.
            return both_partitions(lp_run, recut_spec("OPAL_BENCH_LP_RECUT", "25,1.12,32"))

        guarded("laser_plasma_interaction", float(os.environ.get("OPAL_BENCH_LP_DEADLINE", "300")), lp_body)

    if world > 1 and full_line and not args.no_colliding_beams:
        # BASELINE configs[4] on the same N GPUs: 4000 cells in N subdomains, the beam and its photons crossing the cuts at
        # c (links laid out for photon emission: 3 species slots, 12 fields per particle, open ends)
        def cb_run(cb_recut):
            nonlocal link
            link = None
            cap = int(os.environ.get("OPAL_BENCH_CB_CAPACITY") or (1 << 18))
            link4, kind4 = make_link(rank, world, link_device, cap, stream=stream, photon_emission=True, to_left=rank != 0,
                                     to_right=rank != world - 1, prefer=order,
                                     log=lambda m_: print(m_.replace("[opal_amd]", "[bench]"), file=sys.stderr))
            link4.transport_kind = kind4
            res = colliding_beams(link4, CB_WHAT + "; " + (f"{world} GPUs" if backend == "nccl" else
                                                             f"{world} subdomains sharing {ndev} GPU(s) (REHEARSAL)"), cb_recut)
            res.update(link=kind4, n_gpus=world, link_particle_capacity=cap)
            del link4
            return res

        def cb_body():
            return both_partitions(cb_run, recut_spec("OPAL_BENCH_CB_RECUT", "32,1.10,48"))

        guarded("colliding_beams", float(os.environ.get("OPAL_BENCH_CB_DEADLINE", "240")), cb_body)

    if rank == 0:
        if world == 1 and args.workload == "two_stream" and not weak and not args.no_secondary and not args.self_exchange \
                and (nx, npc) == (C3_CELLS, C3_NPC) and args.elision == "off":
            # BASELINE configs[1], measured the same way in the same process (default path first, then the contract run)
            oa.set_elision(True)
            sim2 = oa.Simulation(yaml_text=YAML.format(nx=C2_CELLS, npc=C2_NPC), stream=stream)
            sim2.set_fused(not args.unfused)
            k2 = f"two_stream:{C2_CELLS}x{C2_NPC}"
            m2d = measure(sim2, half, args.warmup, workload_key=k2) if not (args.unfused or args.no_default_path) else None
            oa.set_elision(False)
            if m2d is not None:
                regroup(sim2)
            m2 = measure(sim2, args.steps, args.warmup, workload_key=k2)
            del sim2
            out["secondary"] = block(m2, f"two_stream, {C2_NPC} e- + {C2_NPC} e+ per cell, {C2_CELLS} cells (BASELINE configs[1]), "
                                         f"1 GPU; " + CONTRACT, args.steps,
                                     {"default_path": block(m2d, DEFAULT, half)} if m2d else None)
            if not args.unfused and not args.no_laser_plasma:
                # BASELINE configs[3]'s physics on one GPU inside the laser-plasma INTERACTION (the pulse reaches the
                # slab after ~21 000 steps): the workload in which nothing can be elided and the particles outrun
                # their tiles -- general kernel, 80 B per update
                lp_npc, lp_warm, lp_steps = 2000, 21000, 2000
                sim3 = oa.Simulation(yaml_text=LASER_PLASMA_YAML.format(npc=lp_npc), stream=stream)
                m3 = measure(sim3, lp_steps, lp_warm, preheat=False, workload_key=f"laser_plasma:{lp_npc}")
                t3 = sim3.t
                del sim3
                out["laser_plasma_interaction"] = block(
                    m3, f"laser_plasma (hole_boring.yaml physics, BASELINE configs[3]) on 1 GPU, {lp_npc} e- + {lp_npc} C6+ "
                        f"per cell in 5000 of 20000 cells, {lp_warm} untimed steps to bring the pulse onto the slab, then "
                        f"{lp_steps} timed steps ending at t = {t3:.3e} s; " + CONTRACT, lp_steps)
            oa.set_elision(True)
            if not args.unfused and not args.no_colliding_beams:
                out["colliding_beams"] = colliding_beams(None, CB_WHAT + "; 1 GPU")
        if not args.no_cpu_baseline and world == 1 and args.workload == "two_stream":
            out["cpu_baseline"] = cpu_baseline(oa, nx, npc)
        write_line()
    if dist is not None:
        # success path only: a rank that failed above has already left with a non-zero status, and its peers
        # are torn down by the launcher instead of waiting in a collective that can never match
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)   # no barrier, no destroy_process_group: let the launcher tear the peers down
