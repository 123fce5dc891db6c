"""CPU-only tests of the product's host side: the C ABI exports what include/opal_amd.h declares,
the config layer reproduces src/setup.rs (incl. its own unit tests), GridDesign and the loader match
the oracle bit for bit, and the compute entry points fail loudly when there is no GPU."""
import ctypes as C
import math
import os
import sys
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXAMPLES = "/root/reference/examples"


@pytest.fixture(scope="module")
def oa():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "opal_amd", "lib", "libopal_amd.so")):
        ge.build()
    import opal_amd
    return opal_amd


def test_abi_exports_every_declared_symbol(oa):
    from opal_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "opal_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(opal_[a-z0-9_]+)\s*\(", header))
    declared -= {"opal_fn1", "opal_fn2", "opal_fn3", "opal_exchange_fn"}
    assert len(declared) > 60
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in opal_amd.h but not exported"
    # and the Python binding knows every one of them
    assert declared <= set(_lib.SIGNATURES), declared - set(_lib.SIGNATURES)


def test_link_plan_follows_its_specification(oa):
    """link.cpp: link_plan -- the sizes of the messages of called transports and the room populations reserve before a
    step's arrivals -- played on a link in host memory (no device): every exchange counting every species, exchanges
    alternating between species (an old ring entry stays valid there: the exchanges after it that did not count a species
    did not push it either), and a peer link that files every 4th exchange (peer.hip: opal_link_plan_selftest)."""
    from opal_amd import _lib
    lib = _lib.load()
    rc = lib.opal_link_plan_selftest()
    assert rc == 0, lib.opal_last_error().decode()


def test_two_mailboxes_per_direction_suffice_without_acknowledgements():
    """The hand-off rule of the peer-mapped mailboxes (peer_device.hpp), played as a model under random schedules: every
    subdomain, per exchange k, writes its two messages into its neighbours' mailboxes k & 1 and then waits for and reads
    its own two -- nothing else orders the subdomains.  Claim: a mailbox is never overwritten before its owner has read
    it, because a neighbour can write exchange k + 2 only after finishing exchange k + 1, which needed my message k + 1,
    which I sent after reading exchange k.  Checked for rings of 1, 2, 3 and 5 subdomains, open and periodic, with the
    scheduler free to run any subdomain that is not blocked.  (The same model with ONE mailbox per direction finds the
    overwrite within a few exchanges: the check has teeth.)"""
    import random
    for world, periodic, seed in [(1, True, 0), (2, True, 1), (3, True, 2), (5, True, 3), (3, False, 4), (5, False, 5)]:
        rng = random.Random(seed)
        K = 60
        # box[r][side][parity] = exchange number of the message lying there (or -1); read[r][side] = last exchange read
        box = [[[-1, -1], [-1, -1]] for _ in range(world)]
        read = [[-1, -1] for _ in range(world)]
        k = [0] * world            # exchange each subdomain is making
        stage = [0] * world        # 0: about to send; 1: sent, about to wait / read
        nb = lambda r, side: ((r - 1) % world if side == 0 else (r + 1) % world) if periodic or 0 <= r + (2 * side - 1) < world else None
        done = 0
        while done < world:
            r = rng.randrange(world)
            if k[r] >= K:
                continue
            if stage[r] == 0:
                for side in (0, 1):
                    n = nb(r, side)
                    if n is None:
                        continue
                    theirs = 1 - side  # what I send left lands in my left neighbour's from-right mailbox
                    # the hazard: the neighbour must have read what lay in this mailbox (exchange k - 2)
                    assert read[n][theirs] >= k[r] - 2, (world, periodic, r, side, k[r], read[n][theirs])
                    box[n][theirs][k[r] & 1] = k[r]
                stage[r] = 1
            else:
                ready = all(nb(r, side) is None or box[r][side][k[r] & 1] == k[r] for side in (0, 1))
                if not ready:
                    continue  # blocked in peer_wait: the scheduler picks somebody else
                for side in (0, 1):
                    if nb(r, side) is not None:
                        assert box[r][side][k[r] & 1] == k[r]  # not a stale or a future message
                        read[r][side] = k[r]
                k[r] += 1
                stage[r] = 0
                if k[r] == K:
                    done += 1


def test_no_gpu_fails_loudly(oa):
    """No silent CPU fallback: without a device the compute objects cannot even be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(oa.OpalError) as e:
        oa.Grid.single(64, 0.0, 1.0)
    assert e.value.status == 2
    with pytest.raises(oa.OpalError):
        oa.Population(oa.ELECTRON, 1024)
    with pytest.raises(oa.OpalError):
        oa.device_count()


CONFIG_TEXT = """---
        control:
          dx: 0.001
          nx: 4000
          ne: sin(a * x)
          ib: a * b^3

        extra:
          dx: 160

        constants:
          a: 2.0 * pi
          b: 17.0
        """


def test_config_parser(oa):
    """setup.rs:446-484 (config_parser)"""
    c = oa.Config(text=CONFIG_TEXT)
    assert c.read_f64("control", "dx") == 0.001
    assert c.read_usize("control", "nx") == 4000
    assert c.read_f64("control", "ib") == 2.0 * math.pi * 17.0 ** 3
    assert c.read_f64("extra", "dx") == 160.0
    ne = c.func("control", "ne", ["x"])
    assert ne(0.6) == math.sin(2.0 * math.pi * 0.6)


def test_time_format(oa):
    """setup.rs:486-492 (time_format) and the float formats of the output files"""
    lib = oa._lib.load()
    buf = C.create_string_buffer(64)

    def fmt(kind, v):
        assert lib.opal_format(kind, v, buf, 64) == 0
        return buf.value.decode()

    assert fmt(0, 2.6e-4) == "  260.00 μs"
    assert fmt(1, 1.0) == "1e0" and fmt(1, -2.5e-7) == "-2.5e-7" and fmt(1, 0.0) == "0e0" and fmt(1, 1234.5) == "1.2345e3"
    assert fmt(2, 1.2345678e-3) == "1.234568e-3" and fmt(2, 0.0) == "0.000000e0"
    assert fmt(3, 0.0) == "0" and fmt(3, 1.5) == "1.5" and fmt(3, 1e-6) == "0.000001" and fmt(3, 1e21) == "1" + "0" * 21


def test_config_errors_and_typing(oa):
    """ConfigError kinds (setup.rs:33-61) and the typed reads (setup.rs:287-370)."""
    c = oa.Config(text=CONFIG_TEXT)
    with pytest.raises(oa.OpalError) as e:
        c.read_f64("nope", "dx")
    assert 'Could not find section "nope"' in str(e.value)
    with pytest.raises(oa.OpalError) as e:
        c.read_f64("control", "nope")
    assert 'Could not find field "nope" in section "control"' in str(e.value)
    with pytest.raises(oa.OpalError) as e:
        c.read_usize("control", "dx")  # a Real is not a usize
    assert "Could not convert" in str(e.value)
    with pytest.raises(oa.OpalError):
        c.read_bool("control", "nx")
    # an Integer constant is silently ignored by with_context (setup.rs:165-175)
    c2 = oa.Config(text="control:\n a: k * 2.0\nconstants:\n k: 3\n")
    with pytest.raises(oa.OpalError):
        c2.read_f64("control", "a")
    c3 = oa.Config(text="control:\n a: k * 2.0\n b: -2^2\n c: 2^-1\n d: 7 % 4\nconstants:\n k: 3.0\n")
    assert c3.read_f64("control", "a") == 6.0
    assert c3.read_f64("control", "b") == -4.0   # unary minus binds looser than ^ (meval)
    assert c3.read_f64("control", "c") == 0.5
    assert c3.read_f64("control", "d") == 3.0
    # constants cannot depend on each other (setup.rs:160)
    c4 = oa.Config(text="control:\n a: q\nconstants:\n p: 2.0\n q: p * 2.0\n")
    with pytest.raises(oa.OpalError):
        c4.read_f64("control", "a")
    # built-ins of with_context: e is the elementary charge, gauss is un-normalised
    assert c3.eval("e") == 1.602177e-19
    assert c3.eval("gauss(1.0, 1.0, 3.0)") == 1.0
    assert c3.eval("step(0.5, 0.0, 1.0) + step(1.0, 0.0, 1.0)") == 1.0
    assert c3.eval("critical(1.0)") == 8.854188e-12 * 9.109383e-31 / 1.602177e-19 ** 2


def test_yaml_subset_accepts_what_input_files_contain(oa):
    """The library's own YAML reader (the reference uses yaml-rust) on the constructs input files use or might use:
    comments, CRLF line endings, blank lines, any indentation width, with and without the document markers, quoted
    and folded / literal block scalars (examples/colliding_beams.yaml folds its laser profile), integers read as reals."""
    cases = {
        "control:\n dx: 0.5 # half\n nx: 3\n": 0.5,
        "control:\r\n dx: 0.5\r\n nx: 3\r\n": 0.5,
        "control:\n\n dx: 0.5\n\n nx: 3\n": 0.5,
        "control:\n    dx: 0.5\n    nx: 3\n": 0.5,
        "---\ncontrol:\n dx: 0.5\n...\n": 0.5,
        "control: \n dx:   0.5   \n": 0.5,
        "control:\n dx: \"0.25 * 2\"\n": 0.5,
        "control:\n dx: '0.25 * 2'\n": 0.5,
        "control:\n dx: >\n  0.25\n  * 2\n": 0.5,
        "control:\n dx: |\n  0.25 * 2\n": 0.5,
        "# top\ncontrol:\n # inside\n dx: 0.5\n": 0.5,
        "control:\n dx: 160\n": 160.0,
        "control:\n dx: 2*micro\n": 2e-6,
    }
    for text, want in cases.items():
        assert oa.Config(text=text).read_f64("control", "dx") == want, text


def test_expression_grammar_and_builtins(oa):
    """What an input file may say: meval's grammar and built-ins (the crate behind setup.rs:287-300) and the context
    setup.rs:110-153 adds -- every function and constant once, against Python's math; precedence and associativity."""
    import math
    c = oa.Config(text="control:\n a: 1.0\nconstants:\n k: 3.0\n")
    cases = {
        "sqrt(2.0)": math.sqrt(2), "abs(-3.5)": 3.5, "exp(1.5)": math.exp(1.5), "ln(7.0)": math.log(7),
        "sin(0.3)": math.sin(.3), "cos(0.3)": math.cos(.3), "tan(0.3)": math.tan(.3), "asin(0.3)": math.asin(.3),
        "acos(0.3)": math.acos(.3), "atan(0.3)": math.atan(.3), "sinh(0.3)": math.sinh(.3), "cosh(0.3)": math.cosh(.3),
        "tanh(0.3)": math.tanh(.3), "asinh(0.3)": math.asinh(.3), "acosh(1.3)": math.acosh(1.3), "atanh(0.3)": math.atanh(.3),
        "floor(-2.5)": -3.0, "ceil(-2.5)": -2.0, "round(2.5)": 3.0, "round(-2.5)": -3.0, "signum(-2.0)": -1.0, "signum(0.0)": 1.0,
        "atan2(1.0, -2.0)": math.atan2(1, -2), "max(1.0, 5.0, 3.0)": 5.0, "min(4.0, 2.0, 3.0)": 2.0, "pi": math.pi,
        "2 + 3 * 4 ^ 2 / 8 - 1": 2 + 3 * 16 / 8 - 1, "2 ^ 3 ^ 2": 2.0 ** 9, "-(2 + 3)": -5.0, "10 % 4 * 2": 4.0, "1e3 + 2.5E-1": 1000.25,
        "micro / nano": 1e-6 / 1e-9, "k * milli": 3e-3, "MeV / keV": 1e3, "mp / me": 1.672622e-27 / 9.109383e-31, "c": 2.997925e8,
        "femto + pico": 1e-15 + 1e-12, "eV": 1.602177e-19, "m": 9.109383e-31,
    }
    for expr, want in cases.items():
        got = c.eval(expr)
        assert got == want or abs(got - want) <= 1e-15 * abs(want), (expr, got, want)
    # the laser profiles (setup.rs:113-133) as functions of (t, x)
    t, x, omega, sigma = 3.0e-15, 1.0e-6, 2.0 * math.pi * 2.997925e8 / 0.8e-6, 2.5
    phi = omega * (t - x / 2.997925e8)
    env = math.exp(-phi ** 2 / (2.0 * sigma ** 2))
    re = c.eval(f"gauss_pulse_re({t!r}, {x!r}, {omega!r}, {sigma!r})")
    im = c.eval(f"gauss_pulse_im({t!r}, {x!r}, {omega!r}, {sigma!r})")
    assert abs(re - (math.sin(phi) + phi * math.cos(phi) / sigma ** 2) * env) < 1e-12
    assert abs(im - (math.cos(phi) - phi * math.sin(phi) / sigma ** 2) * env) < 1e-12
    for bad in ("2 +", "sin(", "nosuchfn(1.0)", "undefined_name * 2", "1 2"):
        with pytest.raises(oa.OpalError):
            c.eval(bad)


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference examples not present")
def test_reference_examples_parse(oa):
    """The three shipped input files are accepted unchanged (docs/input.md)."""
    c = oa.Config(path=os.path.join(REF_EXAMPLES, "two_stream.yaml"))
    assert c.read_usize("control", "nx") == 1000 and c.read_f64("control", "dx") == 500.0
    assert not c.contains("laser") and c.read_bool("control", "current_deposition")
    ux = c.func("electrons", "ux", ["x", "urand", "nrand"])
    ud = 2.5e-24 / (9.109383e-31 * 2.997925e8)
    assert ux(0.0, 0.7, 0.5) == ud * (1.0 + 0.001 * 0.5) * (2.0 * math.floor(0.7 - 0.5) + 1.0)
    assert ux(0.0, 0.2, 0.0) == -ud

    c = oa.Config(path=os.path.join(REF_EXAMPLES, "hole_boring.yaml"))
    assert c.contains("laser") and c.read_usize("control", "nx") == 20000
    assert c.read_f64("control", "dx") == 1.0e-6 / 1000
    omega = 2 * math.pi * 2.997925e8 / 0.8e-6
    ne = c.func("electrons", "ne", ["x"])
    crit = 8.854188e-12 * 9.109383e-31 * omega ** 2 / 1.602177e-19 ** 2
    assert ne(1.0e-6) == 100.0 * crit * 1.0 and ne(-1.0e-6) == 0.0 and ne(5.0e-6) == 0.0
    assert c.read_f64("ions", "Z") == 6.0 and c.read_f64("ions", "A") == 12.0
    ey = c.func("laser", "Ey", ["t", "x"])
    assert math.isfinite(ey(1e-15, 0.0)) and ey(1e-15, 0.0) != 0.0

    c = oa.Config(path=os.path.join(REF_EXAMPLES, "colliding_beams.yaml"))
    ey = c.func("laser", "Ey", ["t", "x"])  # a folded block scalar
    a0, w = 20.0, 2 * math.pi * 2.997925e8 / 0.8e-6
    t = 3.0e-16
    phi = w * (t - 0.0 / 2.997925e8)
    want = (a0 * 9.109383e-31 * 2.997925e8 * w / 1.602177e-19) * math.sin(phi) * math.exp(-math.log(2.0) * phi ** 2 / (2.0 * math.pi ** 2 * 4.0 ** 2))
    assert abs(ey(t, 0.0) - want) <= 1e-12 * abs(want)
    assert c.read_f64("qed", "photon_angle_max") == 50 * 1e-3


def test_design_matches_oracle(oa, oracle):
    """GridDesign::unbalanced / balanced (grid/mod.rs:126-206) against the oracle, bit for bit."""
    o = oracle
    for ntasks, size in [(1, 1000), (3, 1000), (8, 32768), (7, 50)]:
        a, b = oa.design_unbalanced(ntasks, size, 0.25), o.design_unbalanced(ntasks, size, 0.25)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert oa.design_unbalanced(3, 1000, 1.0)[0].sum() == 999  # remainder cells are dropped (:128)
    ne = lambda x: 100.0 if 0.0 <= x < 5.0e-6 else 0.0  # hole_boring's profile
    for ntasks in (1, 2, 4, 8):
        a = oa.design_balanced(ntasks, 20000, -10e-6, 1e-9, ne)
        b = o.design_balanced(ntasks, 20000, -10e-6, 1e-9, ne)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[0].sum() == 20000 and a[0].min() >= 8
    with pytest.raises(oa.OpalError):
        oa.design_balanced(13, 100, 0.0, 1.0, lambda x: 1.0 if x < 10 else 0.0)  # reference panics (:179)


def test_loader_matches_oracle(oa, oracle):
    """Population::new (mod.rs:172-203): same RNG stream, draw order, weights and empty-cell rule."""
    o = oracle
    lib = oa._lib.load()
    from opal_amd._lib import FN1, FN3
    nx, npc, dx = 40, 7, 500.0
    ud = 2.5e-24 / (9.109383e-31 * 2.997925e8)
    dens = lambda x: 20.0 if x > 3000.0 else 0.0  # leaves the first cells empty
    fux = lambda x, ur, nr: ud * (1.0 + 0.001 * nr) * (2.0 * math.floor(ur - 0.5) + 1.0)
    fuy = lambda x, ur, nr: 0.1 * nr
    fuz = lambda x, ur, nr: x * 1e-9 + ur
    for species in (0, 1):
        cap = nx * npc
        cell = np.zeros(cap, np.int32)
        arrs = [np.zeros(cap) for _ in range(5)]
        n = C.c_size_t()
        cbs = (FN1(lambda x, u: dens(x)), FN3(lambda x, a, b, u: fux(x, a, b)), FN3(lambda x, a, b, u: fuy(x, a, b)),
               FN3(lambda x, a, b, u: fuz(x, a, b)))
        rc = lib.opal_loader_generate(species, npc, *cbs, None, 0.0, dx, nx, 3, cap, C.byref(n), cell.ctypes.data,
                                      *[a.ctypes.data for a in arrs])
        assert rc == 0
        g = o.Grid.single(nx, 0.0, dx)
        op = o.Population.new(species, npc, dens, fux, fuy, fuz, g, o.RNG(3), 0.95 * dx / 2.997925e8)
        ref = op.soa()
        assert n.value == len(op) == (nx - 6) * npc
        assert np.array_equal(cell[:n.value], ref["cell"].astype(np.int32))
        for a, k in zip(arrs, ("x", "ux", "uy", "uz", "w")):
            assert np.array_equal(a[:n.value], ref[k]), k


def test_parallel_loader_is_the_serial_loader_bit_for_bit(oa):
    """Population::new is one loop over one generator (mod.rs:172-203).  The loader may run it on several host threads
    (opal_set_loader_threads; a Simulation does by default): every particle takes a fixed number of draws -- 11 for an
    electron, 10 for an ion, 14 for a photon -- so the generator's state at any cell is known, one thread advances it, the
    others fill.  Same arrays bit for bit, whatever the species, with empty cells in between, and the generator is left where
    the serial loop leaves it (the next species continues from there)."""
    lib = oa._lib.load()
    from opal_amd._lib import FN1, FN3
    nx, npc, dx = 300, 800, 500.0
    dens = lambda x: 20.0 if (x > 3000.0 and not 60000.0 < x < 75000.0) else 0.0
    cbs = (FN1(lambda x, u: dens(x)), FN3(lambda x, a, b, u: 0.1 * b + a), FN3(lambda x, a, b, u: b), FN3(lambda x, a, b, u: x * 1e-9 + a * b))
    cap = nx * npc

    def generate(species, threads):
        assert lib.opal_set_loader_threads(threads) == 0
        cell = np.zeros(cap, np.int32)
        arrs = [np.zeros(cap) for _ in range(5)]
        n = C.c_size_t()
        try:
            rc = lib.opal_loader_generate(species, npc, *cbs, None, 0.0, dx, nx, 5, cap, C.byref(n), cell.ctypes.data,
                                          *[a.ctypes.data for a in arrs])
        finally:
            lib.opal_set_loader_threads(0)
        assert rc == 0
        return n.value, cell, arrs

    for species in (0, 1, 2):
        n1, c1, a1 = generate(species, 1)
        n4, c4, a4 = generate(species, 5)
        assert n1 == n4 > 200000          # (above the size below which the loader does not bother with threads)
        assert np.array_equal(c1, c4)
        for u, v in zip(a1, a4):
            assert np.array_equal(u, v)


def test_rng_known_answers(oracle):
    """xoshiro256** seeded via SplitMix64: first outputs for seed 0 (public reference values of the
    generator: splitmix64(0) state then xoshiro256starstar)."""
    r = oracle.RNG(0)
    # state after SplitMix64 seeding with 0
    assert [int(v) for v in r.s.s] == [0xe220a8397b1dcdaf, 0x6e789e6aa1b965f4, 0x06c45d188009454f, 0xf88bb8a8724c81ec]
    assert r.next_u64() == 0x99ec5f36cb75f2b4


def test_output_spec_file_names(oa):
    """File names of Population::write_data (mod.rs:527-559) and which specs the reference skips."""
    lib = oa._lib.load()
    buf = C.create_string_buffer(512)

    def name(spec, pop="electron", index=7):
        rc = lib.opal_hist_filename(b"out", index, pop.encode(), spec.encode(), buf, 512)
        return buf.value.decode() if rc == 0 else None

    assert name("x") == "out/7_electron_x.fits"
    assert name("x:px") == "out/7_electron_x-px.fits"
    assert name("energy:(log;energy)") == "out/7_electron_energy_energy_log.fits"
    assert name("x:energy:(energy)", "carbon") == "out/7_carbon_x-energy_energy.fits"
    assert name("longitude:latitude:(energy)", "photon") == "out/7_photon_longitude-latitude_energy.fits"
    assert name("x:(auto)") == "out/7_electron_x_auto.fits"          # weight "auto" is accepted, and named
    assert name("px:(32;weight)") == "out/7_electron_px.fits"
    assert name("x:px:pz") is None and name("bogus") is None and name("x:(volume)") is None
    assert name("x:(a;b;c)") == "out/7_electron_x.fits"               # three fields: defaults (mod.rs:463)


def test_committed_traffic_figures_point_at_committed_profiles():
    """bench.py's roofline.traffic comes from profiles/traffic.json (separate rocprofv3 --pmc passes of the bench
    command, scripts/profile_bench.sh + summarise_profile.py): every entry is keyed by the workload and the kernel
    instantiation it was measured on, names a committed PMC summary that holds the same bytes-per-update figure, and
    carries the hash of the kernel sources it was taken with (bench.py reports an entry only while that hash is current)."""
    import json
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    head = "two_stream:32768x1526|"
    assert head + "k_push_pair<1, true, false>" in t and head + "k_push_pair<1, false, true>" in t
    for key, entry in t.items():
        workload, kernel = key.split("|")
        assert re.fullmatch(r"k_push(_pair)?<[^>]*>", kernel), key
        src = os.path.join(root, entry["source"])
        assert os.path.exists(src), (key, entry["source"])
        pmc = json.load(open(src))
        assert pmc["workload_key"] == workload and pmc["kernel_src_sha"] == entry["src_sha"]
        assert abs(pmc["kernels"][kernel]["hbm_bytes_per_particle_update"] - entry["hbm_bytes_per_particle_update"]) < 1e-9
        assert 8.0 < entry["hbm_bytes_per_particle_update"] < 90.0
    # the electrostatic launch moves less than the general kernel, which moves less than the survey's 80 B
    assert t[head + "k_push_pair<1, false, true>"]["hbm_bytes_per_particle_update"] < \
        t[head + "k_push_pair<1, true, false>"]["hbm_bytes_per_particle_update"] < 80.0


def test_bench_prices_each_kernel_form_with_its_own_bytes():
    """bench.py's roofline: the algorithmic bytes per particle-update follow the FORM of the fused push that ran
    (opal_prof_forms: bit 0 electrostatic, 1 rho, 2 / 3 the electrons' / ions' weight from the launch arguments, 4 a
    single-species launch) -- 80 B only for the general form with streamed weights."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    general = bench.bytes_model({2: 200}, 5, 5, False)
    assert general["bytes_per_update"] == 80.0 and general["kernel"] == "k_push_pair<1, true, false>"
    es = bench.bytes_model({1 | 4 | 8: 19, 1 | 2 | 4 | 8: 1}, 5, 5, False)   # 19 steps without rho, the last with
    assert es["bytes_per_update"] == 40.0 and es["kernel"] == "k_push_pair<1, false, true>"
    assert bench.bytes_model({2 | 4 | 8: 3}, 5, 5, False)["bytes_per_update"] == 72.0      # general, uniform weights
    assert bench.bytes_model({1 | 2: 3}, 5, 5, False)["bytes_per_update"] == 48.0          # electrostatic, weights streamed
    assert bench.bytes_model({2 | 4: 3}, 6, 2, False)["bytes_per_update"] == 80.0 - 8.0 * 6 / 8   # one species' weight only
    assert bench.bytes_model({2: 5}, 5, 5, True)["bytes_per_update"] == 88.0                # unfused advance also writes prev_x
    assert bench.bytes_model({}, 5, 5, False) is None
    # the hash that guards profiles/traffic.json does not move with comments or white space
    assert len(bench.src_sha()) == 16


def test_native_launcher_without_a_gpu_fails_loudly_and_at_once(tmp_path):
    """`opal_amd/bin/opal_amd [-n N] input.yaml` (main.cpp: the reference's `mpirun -n N opal input.yaml`) has no CPU
    fallback: where no GPU is visible every rank says so, the launcher ends the others and leaves with a non-zero status --
    nobody waits in a socket for a rank that has gone.  Also its usage line (main.rs:56-59)."""
    import subprocess
    exe = os.path.join(ROOT, "opal_amd", "bin", "opal_amd")
    if not os.path.exists(exe):
        pytest.skip("opal_amd/bin/opal_amd is built by __graft_entry__.build()")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible: tests/test_gpu_multirank.py runs the launcher for real")
    except ImportError:
        pass
    yml = tmp_path / "in.yaml"
    yml.write_text("---\ncontrol:\n nx: 64\n")
    for n in (1, 3):
        res = subprocess.run([exe] + (["-n", str(n)] if n > 1 else []) + [str(yml)], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=60)
        assert res.returncode != 0
        assert "no HIP device visible" in res.stdout.decode()
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert res.returncode != 0 and "Usage: opal_amd [-n ranks] input-file" in res.stdout.decode()

