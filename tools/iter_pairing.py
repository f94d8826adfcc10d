"""Developer experiment: how well does an environment's Newton iteration count of the last step predict the next one's?
Mean per-wavefront maximum (two environments per wavefront) under adjacent pairing vs pairing sorted by the last count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mujoco_amd import BatchSim, Model, workload
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = workload.with_ground_plane(Model.load(os.path.join(ROOT, "tests", "golden", "dexterous_model.npz")))
E = 65536
q, v = workload.config2_state(m, E, seed=1234)
sim = BatchSim(m, n_envs=E, device_id=0)
sim.data.qpos.copy_(torch.as_tensor(q))
bank = [torch.as_tensor(workload.config2_ctrl(m, E, seed=1234, resample_idx=k)).to("cuda:0") for k in range(3)]
prev = None
rows = []
for p in range(3):
    sim.data.ctrl.copy_(bank[p])
    for s in range(50):
        sim.step(1)
        it = sim.data.solver_iter.cpu().numpy().astype(np.int64).ravel()
        nc = sim.data.ncon.cpu().numpy().astype(np.int64).ravel()
        if prev is not None:
            adj = np.maximum(it[0::2], it[1::2]).mean()
            o = np.argsort(prev, kind="stable")
            srt = np.maximum(it[o][0::2], it[o][1::2]).mean()
            o2 = np.argsort(it, kind="stable")
            ideal = np.maximum(it[o2][0::2], it[o2][1::2]).mean()
            o3 = np.argsort(prev * 64 + nc, kind="stable")
            srt3 = np.maximum(it[o3][0::2], it[o3][1::2]).mean()
            rows.append((p * 50 + s, it.mean(), adj, srt, srt3, ideal))
        prev = it
rows = np.array(rows)
for r in rows[::10]:
    print("step %3d mean %.3f  adjacent-pair max %.3f  sorted-by-prev %.3f  by-prev+ncon %.3f  ideal %.3f" % tuple(r))
print("average:", rows[:, 1:].mean(axis=0))
