import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mujoco_amd import BatchSim, Model, workload
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = workload.contact_free(Model.load(os.path.join(ROOT, "tests", "golden", "dexterous_model.npz")))
E = 130
q, v = workload.config2_state(m, E, seed=77)
c = workload.config2_ctrl(m, E, seed=77)
def run(env, n):
    for k in ("BS_NO_CHAIN", "BS_NO_PALM_SCALAR"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sim = BatchSim(m.copy(), n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.step(n); sim.sync()
    out = {f: getattr(sim.data, f).cpu().double().numpy().copy() for f in ("qpos", "qvel", "qacc", "solver_iter", "qfrc_bias")}
    out["path"] = sim.kernel_path()[1]
    sim.close()
    return out
for n in (1, 2, 60):
    a = run({}, n); b = run({"BS_NO_CHAIN": "1"}, n); o = run({"BS_NO_PALM_SCALAR": "1"}, n)
    print(n, a["path"], "|", b["path"])
    for f in ("qacc", "qvel", "qpos", "solver_iter"):
        d = np.abs(a[f] - b[f])
        cols = np.argwhere(d.reshape(E, -1).max(axis=0) > 0).ravel()
        print(f"  {f:12s} chain-vs-generic max {d.max():.3e} dofs differing {cols[:12]}   new-vs-old(chain) max {np.abs(a[f]-o[f]).max():.3e}")
a = run({"BS_NO_CHAIN": "1"}, 1); o = run({"BS_NO_CHAIN": "1", "BS_NO_PALM_SCALAR": "1"}, 1)
print("generic new-vs-old qacc", np.abs(a["qacc"] - o["qacc"]).max())
c1 = run({}, 1); c0 = run({"BS_NO_PALM_SCALAR": "1"}, 1)
print("old chain vs old generic qacc", np.abs(c0["qacc"] - o["qacc"]).max(), " new chain vs new generic", np.abs(c1["qacc"] - a["qacc"]).max())
print("bias dof0: chain new", c1["qfrc_bias"][:3, 0], "generic new", a["qfrc_bias"][:3, 0], "old", o["qfrc_bias"][:3, 0])
