"""bench.py's pieces that need no GPU: the byte model a launch is priced with, the link capacities, and that the PMC traffic
figures under profiles/ were taken with the kernel sources as they are now (bench.py refuses a stale entry: a line would then
carry `traffic: null`)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_bytes_model_by_form():
    # general form, weights streamed = 44 B read + 36 B written
    m = bench.bytes_model({2: 10}, 5, 5, False)
    assert m["bytes_per_update"] == 80.0 and m["kernel"] == "k_push_pair<1, true, false>"
    # electrostatic launch, both weights from the launch arguments, rho skipped: cell, x, ux read and written
    m = bench.bytes_model({1 | 4 | 8: 7}, 5, 5, False)
    assert m["bytes_per_update"] == 40.0 and m["kernel"] == "k_push_pair<1, false, true>"
    # general, uniform weights from the arguments
    assert bench.bytes_model({2 | 4 | 8: 3}, 5, 5, False)["bytes_per_update"] == 72.0
    # a mix of forms is priced launch by launch
    assert bench.bytes_model({2: 1, 1 | 4 | 8: 1}, 5, 5, False)["bytes_per_update"] == 60.0
    # the reference's operator order: the advance also writes prev_x
    assert bench.bytes_model({2: 1}, 5, 5, True)["bytes_per_update"] == 88.0
    assert bench.bytes_model({}, 5, 5, False) is None


def test_link_capacities_cover_the_initial_bound():
    # what stands within 8 cells of a cut on both sides at the initial density is 16 npc; links are created for 20
    for npc in (100, 1526, 2441, 10000):
        assert bench.lp_link_capacity(npc) >= 16 * npc + 64


def test_traffic_entries_were_taken_with_the_current_kernel_sources():
    sha = bench.src_sha()
    assert re.fullmatch(r"[0-9a-f]{16}", sha)
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    want = {f"two_stream:{bench.C3_CELLS}x{bench.C3_NPC}|k_push_pair<1, true, false>",
            f"two_stream:{bench.C3_CELLS}x{bench.C3_NPC}|k_push_pair<1, false, true>",
            f"two_stream:{bench.C2_CELLS}x{bench.C2_NPC}|k_push_pair<1, true, false>",
            "laser_plasma:2000|k_push_pair<1, true, false>"}
    assert want <= set(t), sorted(set(t))
    stale = [k for k in want if t[k]["src_sha"] != sha]
    assert not stale, f"profiles/traffic.json is stale for {stale}: re-run scripts/r04_profile.sh (kernel sources changed)"
    for k in want:
        b = t[k]["hbm_bytes_per_particle_update"]
        assert 30.0 < b < 90.0, (k, b)


def test_live_traffic_passes_fail_soft_without_a_gpu():
    """bench.py measures roofline.traffic on the box it runs on with two child runs under rocprofv3's counters; where that
    cannot work (here: no GPU, the child's own run refuses to start) it must come back with a note and nothing else, so
    that the line falls back to the committed constant instead of dying."""
    import bench
    bench.LIVE_TRAFFIC.clear()
    note = bench.measure_traffic_live(["--gpus", "1", "--steps", "20", "--warmup", "5"], deadline_s=120.0)
    assert isinstance(note, str) and note.startswith("live PMC passes"), note
    assert "failed" in note or "skipped" in note or "gave up" in note, note
    assert bench.LIVE_TRAFFIC == {}
