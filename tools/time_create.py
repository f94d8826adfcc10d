import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_amd import BatchSim, Model, workload
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = workload.with_ground_plane(Model.load(os.path.join(ROOT, "tests", "golden", "dexterous_model.npz")))
t0 = time.perf_counter(); sim = BatchSim(m, n_envs=64); sim.step(1); sim.sync(); t1 = time.perf_counter()
print(os.environ.get("BS_LIB", "default").split("/")[-2], "create + first step: %.2f s" % (t1 - t0), sim.footprint() if hasattr(sim, "footprint") else "")
