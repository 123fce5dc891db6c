"""Diagnostic: BASELINE configs[0] (examples/two_stream.yaml as shipped: 1e5 electrons, 1000 cells) is
launch-latency-bound on a GPU: time per step with everything queued asynchronously.
(Replaying four-step blocks from a captured hipGraph was tried here: 24.5 us/step against 22.2 us with plain
asynchronous launches on ROCm 7.2, so the driver does not use graphs.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import opal_amd as oa
from test_gpu_driver import TWO_STREAM

sim = oa.Simulation(yaml_text=TWO_STREAM)
n = len(sim.electrons)
sim.step(200); sim.sync()
for rep in range(3):
    t0 = time.perf_counter()
    sim.step(5000); sim.sync()
    el = time.perf_counter() - t0
    print("particles %d  us/step %.2f  updates/s %.3e" % (n, el / 5000 * 1e6, n * 5000 / el), flush=True)
