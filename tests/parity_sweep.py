"""Wider GPU-vs-oracle sweep than the pytest cases (run by hand on the GPU box: `python tests/parity_sweep.py`);
the figures quoted in DESIGN.md section 2 come from it.  Not collected by pytest."""
import os, sys, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_amd import BatchSim, Model, workload
from oracle import OracleSim
base = Model.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dexterous_model.npz"))
def sweep(name, m, E, n, seed, every=1):
    q, v = workload.config2_state(m, E, seed=seed); c = workload.config2_ctrl(m, E, seed=seed)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.step(n); sim.sync()
    qp, qv, qa = (getattr(sim.data, f).cpu().numpy() for f in ("qpos", "qvel", "qacc"))
    nc = sim.data.ncon.cpu().numpy() if m.npair else None
    eq = eqv = eqa = 0.0; bad = 0; pairs_ok = 0; tot = 0
    for e in range(0, E, every):
        s = OracleSim(m); s.qpos[:] = q[e]; s.ctrl[:] = c[e]; s.step(n)
        eq = max(eq, np.abs(qp[e] - s.qpos).max()); eqv = max(eqv, np.abs(qv[e] - s.qvel).max())
        eqa = max(eqa, np.abs(qa[e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()))
        if nc is not None:
            tot += 1
            want = sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
            got = sorted((g1, g2) for g1, g2, _ in sim.data.contacts(e))
            pairs_ok += want == got
    print(f"{name}: {E} envs x {n} steps, checked {len(range(0, E, every))}: max|dqpos| {eq:.3g} max|dqvel| {eqv:.3g} max rel dqacc {eqa:.3g}"
          + (f" contact sets equal in {pairs_ok}/{tot}" if nc is not None else ""), flush=True)
sweep("contact-free euler", workload.contact_free(base), 256, 1000, 101, every=4)
m = workload.contact_free(base); m.opt_integrator = 1
sweep("contact-free rk4", m, 128, 300, 102, every=4)
sweep("contacts", base.copy(), 512, 100, 103, every=4)
sweep("contacts+plane", workload.with_ground_plane(base, z=-0.05), 512, 100, 104, every=4)
