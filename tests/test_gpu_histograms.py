"""Distribution-function output (Population::write_data / Histogram): GPU histograms
against the oracle's restatement of hgram.rs, and the FITS files against what was binned."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C = 2.997925e8


def read_fits(path):
    """Minimal reader for a primary-HDU image of doubles."""
    raw = open(path, "rb").read()
    assert len(raw) % 2880 == 0
    cards, pos, done = {}, 0, False
    while not done:
        block = raw[pos:pos + 2880].decode("ascii")
        pos += 2880
        for k in range(36):
            c = block[80 * k:80 * (k + 1)]
            key = c[:8].strip()
            if key == "END":
                done = True
                break
            if c[8:10] == "= ":
                val = c[10:].split(" / ")[0].strip()
                cards[key] = val.strip("'").strip() if val.startswith("'") else val
    naxis = int(cards["NAXIS"])
    shape = [int(cards[f"NAXIS{k}"]) for k in range(1, naxis + 1)]
    n = int(np.prod(shape))
    data = np.frombuffer(raw[pos:pos + 8 * n], dtype=">f8").astype(np.float64)
    return cards, shape, data


def _state(nx, npc, seed):
    rng = np.random.default_rng(seed)
    n = nx * npc
    cell = np.repeat(np.arange(nx, dtype=np.int32), npc)
    x = rng.random(n)
    u = 0.7 * rng.standard_normal((3, n))
    w = 1.0e3 * (1.0 + rng.random(n))
    return cell, x, u[0], u[1], u[2], w


@pytest.mark.parametrize("species", [0, 1, 2])
@pytest.mark.parametrize("spec,names,bspec,weight", [
    ("x", ["x"], "auto", "weight"),
    ("x:px", ["x", "px"], "auto", "weight"),
    ("x:p_perp", ["x", "p_perp"], "auto", "weight"),
    ("py:pz", ["py", "pz"], "auto", "weight"),
    ("energy", ["energy"], "auto", "weight"),
    ("energy:(log;energy)", ["energy"], "log", "energy"),
    ("x:energy:(energy)", ["x", "energy"], "auto", "energy"),
    ("theta:phi:(17;weight)", ["theta", "phi"], 17, "weight"),
    ("longitude:latitude", ["longitude", "latitude"], "auto", "weight"),
    ("px:(0.25;weight)", ["px"], 0.25, "weight"),
])
def test_histogram_matches_oracle(oracle, species, spec, names, bspec, weight):
    import opal_amd as oa
    o = oracle
    nx, npc, dx = 40, 50, 1.0e-8
    dt = 0.95 * dx / C
    st = _state(nx, npc, 5)
    g = oa.Grid.single(nx, -2.0e-7, dx)
    og = o.Grid.single(nx, -2.0e-7, dx)
    p = oa.Population.from_soa(species, *st, dx, dt)
    op = o.Population.from_soa(species, *st, dx, dt)
    if species == 1:
        p.set_charge_state(6.0, 6.0, 12.0)
        op.set_charge_state(6.0, 6.0, 12.0)
    # species 2: photons (photon.rs:211-224: momentum k m_e c, energy |k| m_e c^2), as emission fills them
    h = p.histogram(g, spec)
    ref = o.histogram(op, og, names, bspec, weight)
    assert h["nbins"][:len(names)] == ref["nbins"][:len(names)]
    assert len(h["cts"]) == len(ref["cts"])
    for d in range(len(names)):
        assert abs(h["min"][d] - ref["min"][d]) <= 1e-12 * max(abs(ref["min"][d]), abs(ref["max"][d]))
        assert abs(h["bin_sz"][d] - ref["bin_sz"][d]) <= 1e-12 * abs(ref["bin_sz"][d])
    assert abs(h["total"] - ref["total"]) <= 1e-12 * abs(ref["total"])
    # a value within rounding of a bin edge may land in the neighbouring bin: compare the bulk tightly and
    # allow a handful of single-particle moves
    scale = np.max(ref["cts"])
    diff = np.abs(h["cts"] - ref["cts"])
    assert np.sum(diff > 1e-9 * scale) <= 4
    # (the particle holding the maximum sits exactly on the last edge: rounding decides whether the
    # reference's `fbin < len` keeps or drops it)
    assert abs(np.sum(h["cts"]) - np.sum(ref["cts"])) <= 1e-9 * np.sum(ref["cts"]) + 4 * scale


def test_write_data_fits_files(oracle, tmp_path):
    import opal_amd as oa
    o = oracle
    nx, npc, dx = 32, 40, 1.0e-8
    dt = 0.95 * dx / C
    st = _state(nx, npc, 7)
    g = oa.Grid.single(nx, 0.0, dx)
    og = o.Grid.single(nx, 0.0, dx)
    p = oa.Population.from_soa(oa.ELECTRON, *st, dx, dt)
    op = o.Population.from_soa(o.ELECTRON, *st, dx, dt)
    specs = ["x", "x:px", "energy:(log;energy)", "nonsense", "x:bogus", "x:px:pz"]
    p.write_data(g, str(tmp_path), 3, "electron", specs)
    files = sorted(os.listdir(tmp_path))
    # names of mod.rs:527-559; unmappable specs are skipped silently
    assert files == ["3_electron_energy_energy_log.fits", "3_electron_x-px.fits", "3_electron_x.fits"]
    cards, shape, data = read_fits(tmp_path / "3_electron_x-px.fits")
    ref = o.histogram(op, og, ["x", "px"])
    assert shape == [ref["nbins"][1], ref["nbins"][0]]
    assert cards["BITPIX"] == "-64" and cards["CNAME1"] == "x" and cards["CNAME2"] == "px"
    assert cards["CUNIT1"] == "m" and cards["CUNIT2"] == "MeV/c" and cards["BUNIT"] == "1/(m.MeV/c)"
    assert cards["OBJECT"] == "hgram/density/x_px"
    assert abs(float(cards["CDELT1"]) - ref["bin_sz"][0]) <= 1e-12 * ref["bin_sz"][0]
    assert abs(float(cards["CRVAL2"]) - (ref["min"][1] + 0.5 * ref["bin_sz"][1])) <= 1e-12 * abs(ref["bin_sz"][1]) + 1e-12 * abs(ref["min"][1])
    assert abs(float(cards["TOTAL"]) - ref["total"]) <= 1e-12 * ref["total"]
    assert abs(float(cards["DATAMAX"]) - data.max()) <= 1e-12 * data.max()
    scale = np.max(ref["cts"])
    assert np.sum(np.abs(data - ref["cts"]) > 1e-9 * scale) <= 4
    cards, shape, data = read_fits(tmp_path / "3_electron_x.fits")
    assert shape == [o.histogram(op, og, ["x"])["nbins"][0]] and cards["OBJECT"] == "hgram/density/x"


def test_photon_r_and_chi_histograms(oracle):
    """The accessors that read a photon's extras: r = hypot(y, z) and chi (mod.rs:393-395, 428-430) after a few
    pushes through a field."""
    import opal_amd as oa
    o = oracle
    nx, npc, dx = 40, 50, 1.0e-8
    dt = 0.9 * dx / C
    cell, x, kx, ky, kz, w = _state(nx, npc, 9)
    g = oa.Grid.single(nx, 0.0, dx)
    og = o.Grid.single(nx, 0.0, dx)
    for grid in (g, og):
        grid.set_raw("Ey", np.full(grid.len, 3.0e13))
        grid.set_raw("Bz", np.full(grid.len, 2.0e4))
    p = oa.Population(2, 2 * nx * npc)
    p.track_extras(1)
    p.upload(cell, x, 50 * kx, 50 * ky, 50 * kz, w, dx, dt)
    op = o.Population.from_soa(2, cell, x, 50 * kx, 50 * ky, 50 * kz, w, dx, dt)
    for _ in range(4):
        p.advance(g, dt)
        op.advance(og, dt)
    for spec, names in (("r", ["r"]), ("chi", ["chi"]), ("x:chi", ["x", "chi"]), ("work", ["work"])):
        h = p.histogram(g, spec)
        ref = o.histogram(op, og, names)
        assert h["nbins"][:len(names)] == ref["nbins"][:len(names)], spec
        assert abs(h["total"] - ref["total"]) <= 1e-12 * abs(ref["total"])
        for d in range(len(names)):
            assert abs(h["min"][d] - ref["min"][d]) <= 1e-9 * max(abs(ref["max"][d]), 1e-300), (spec, d)
        scale = np.max(ref["cts"])
        assert np.sum(np.abs(h["cts"] - ref["cts"]) > 1e-9 * scale) <= 6, spec
