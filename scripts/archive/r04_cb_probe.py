import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import opal_amd as oa
import bench
stream = torch.cuda.current_stream().cuda_stream
for prof in (0, 5, 0):
    sim = oa.Simulation(yaml_text=bench.COLLIDING_BEAMS_YAML.format(npc=20000), stream=stream)
    sim.step(1750); sim.sync()
    oa.prof_reset(); oa.prof_enable(prof)
    t0 = time.perf_counter()
    for seg in range(15):
        sim.step(50); sim.sync()
        n = sim.photons.counters()["slots"]
    el = time.perf_counter() - t0
    oa.prof_enable(0)
    print("prof stride", prof, "ms/step %.4f" % (1e3 * el / 750), "photons", n, flush=True)
    t0 = time.perf_counter(); sim.step(300); sim.sync(); print("  after: ms/step %.4f" % (1e3*(time.perf_counter()-t0)/300), flush=True)
    del sim
