import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mujoco_amd import BatchSim, Model, workload
from mujoco_amd.model import SOL_PGS
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = workload.with_ground_plane(Model.load(os.path.join(ROOT, "tests", "golden", "dexterous_model.npz")), z=-0.05).copy()
m.opt_solver = SOL_PGS
E = 133
q, v = workload.config2_state(m, E, seed=52)
v = np.random.default_rng(3).normal(size=v.shape) * 0.3
c = workload.config2_ctrl(m, E, seed=52)
full = ("xpos", "xquat", "cinert", "cdof", "cvel", "qM", "qLD", "qfrc_bias", "qacc_smooth", "efc_D", "efc_aref", "efc_force", "qfrc_constraint", "qacc", "solver_iter", "nefc", "ncon", "warning")
def run(env):
    for k in ("BS_NO_CHAIN", "BS_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward(); sim.sync()
    out = {f: getattr(sim.data, f).clone().cpu().double().numpy() for f in full}
    out["path"] = sim.kernel_path()[1]
    sim.close()
    return out
a = run({"BS_NO_CHAIN": "1"}); b = run({"BS_KERNEL": "wide"}); a2 = run({"BS_NO_CHAIN": "1"})
print(a["path"], "|", b["path"])
for f in full:
    d = np.abs(a[f] - b[f]); d2 = np.abs(a[f] - a2[f])
    bad = np.argwhere(d.reshape(E, -1).max(axis=1) > 0).ravel()
    print(f"{f:16s} max|32-wide| {d.max():.3e}  max|32-32 again| {d2.max():.3e}  envs differing {len(bad)} {bad[:10]}")
e = int(np.argwhere(np.abs(a['efc_force'] - b['efc_force']).reshape(E, -1).max(axis=1) > 0).ravel()[0]) if np.abs(a['efc_force'] - b['efc_force']).max() > 0 else 0
print("env", e, "iters", a["solver_iter"][e], b["solver_iter"][e], "ncon", a["ncon"][e], b["ncon"][e], "nefc", a["nefc"][e], b["nefc"][e])
print("efc_force 32  ", a["efc_force"][e][:30])
print("efc_force wide", b["efc_force"][e][:30])
from oracle import OracleSim
for e in (0, 1, 5):
    s = OracleSim(m)
    s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
    s.forward()
    print("env", e, "oracle qacc vs 32:", np.abs(s.qacc - a["qacc"][e]).max(), " vs wide:", np.abs(s.qacc - b["qacc"][e]).max(), "oracle iters", getattr(s, "solver_iter", None))
