"""The WIDE build of the tree-parallel kernels (physics_tp.h compiled with TP_LPE = 64: one environment per wavefront,
up to 63 bodies and 64 contacts per environment, kernels k_*_w) — VERDICT r02 item 7: the reference's tree builder accepts
any single-root joint tree (/root/reference/src/Fusion_Model.py:94-142), and a model or contact count beyond the 32-lane
kernels used to fall 40x to the generic ones.  Same arithmetic as the 32-lane build: on a model both can run the results
are bit-identical; a 40-body tree that only the wide build can run is compared with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(model, E, q, v, c, steps, env, monkeypatch, forward=False, fields=("qpos", "qvel", "qacc", "nefc", "solver_iter", "warning")):
    import torch
    from mujoco_amd import BatchSim
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    sim = BatchSim(model, n_envs=E)
    for k in env:
        monkeypatch.delenv(k)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    if steps:
        sim.step(steps)
    if forward:
        sim.forward()
    sim.sync()
    out = {f: getattr(sim.data, f).clone() for f in fields}
    out["path"] = sim.kernel_path()[1]
    if model.npair and not (model.opt_disableflags & 16):
        out["ncon"] = sim.data.ncon.clone()
        out["contacts"] = [sim.data.contact_list(e) for e in range(0, E, 7)]
    sim.close()
    return out


@pytest.mark.parametrize("variant", ["free-newton", "free-cg", "free-pgs", "free-rk4", "contacts-newton", "contacts-cg", "contacts-pgs"])
def test_wide_build_is_bitwise_equal_to_the_32_lane_build(hand_model, monkeypatch, variant):
    """The hand fits both builds: BS_KERNEL=wide against the generic 32-lane kernels (BS_NO_CHAIN=1: the wide build has no
    CHAIN instantiations, and those are bit-identical to the generic ones, test_chain_kernel_is_bitwise_neutral)."""
    import torch
    from mujoco_amd import workload
    from mujoco_amd.model import INT_RK4, SOL_CG, SOL_PGS
    kind, sol = variant.split("-")
    m = workload.contact_free(hand_model) if kind == "free" else workload.with_ground_plane(hand_model, z=-0.05)
    m = m.copy()
    if sol == "cg":
        m.opt_solver = SOL_CG
    elif sol == "pgs":
        m.opt_solver = SOL_PGS
    elif sol == "rk4":
        m.opt_integrator = INT_RK4
    E = 133  # ragged in both builds
    q, v = workload.config2_state(m, E, seed=52)
    v = np.random.default_rng(3).normal(size=v.shape) * 0.3
    c = workload.config2_ctrl(m, E, seed=52)
    steps = 12 if kind == "free" else 4
    a = _run(m, E, q, v, c, steps, {"BS_NO_CHAIN": "1"}, monkeypatch)
    b = _run(m, E, q, v, c, steps, {"BS_KERNEL": "wide"}, monkeypatch)
    assert "_w" not in a["path"] and "_w)" in b["path"], (a["path"], b["path"])
    for f in ("qpos", "qvel", "qacc", "nefc", "solver_iter", "warning"):
        assert torch.equal(a[f], b[f]), (variant, f, float((a[f].double() - b[f].double()).abs().max()))
    if kind == "contacts":
        assert torch.equal(a["ncon"], b["ncon"]) and int(a["ncon"].sum()) > 100
        for ca, cb in zip(a["contacts"], b["contacts"]):
            assert [(x.geom1, x.geom2, x.dist) for x in ca] == [(x.geom1, x.geom2, x.dist) for x in cb]
    # bs_forward: every intermediate the full-dump kernels write
    full = ("xpos", "xquat", "cinert", "cdof", "cvel", "qM", "qLD", "qfrc_bias", "qacc_smooth", "efc_D", "efc_aref", "efc_force", "qfrc_constraint", "qacc")
    a = _run(m, E, q, v, c, 0, {"BS_NO_CHAIN": "1"}, monkeypatch, forward=True, fields=full)
    b = _run(m, E, q, v, c, 0, {"BS_KERNEL": "wide"}, monkeypatch, forward=True, fields=full)
    for f in full:
        assert torch.equal(a[f], b[f]), (variant, "forward", f)


def forty_body_tree_xml(plane=True):
    """Eight five-link fingers of capsules on a fixed palm: 40 moving bodies + palm + world = 42 bodies, 40 hinge dofs of
    dof depth 5, limited joints, position actuators on the first four links of a finger, the fifth coupled to the fourth by a
    joint equality (the exporter's M-DP rule, /root/reference/src/Fusion_to_Mujoco.py:107-108)."""
    fingers, acts, eqs = [], [], []
    for f in range(8):
        ang = 2 * np.pi * f / 8
        x, y = 0.06 * np.cos(ang), 0.06 * np.sin(ang)
        body, close = "", ""
        for k in range(5):
            name = f"f{f}_{k}"
            pos = f"{x:.6f} {y:.6f} 0.02" if k == 0 else "0 0 0.045"
            axis = "1 0 0" if (k + f) % 2 == 0 else "0 1 0"
            body += (f'<body name="{name}" pos="{pos}"><joint name="{name}_j" type="hinge" axis="{axis}" range="-0.9 1.1" limited="true"/>'
                     f'<geom name="{name}_g" type="capsule" size="0.009 0.018" pos="0 0 0.0225"/>')
            close += "</body>"
            if k < 4:
                acts.append(f'<position name="{name}_a" joint="{name}_j" kp="2" ctrllimited="true" ctrlrange="-0.9 1.1" forcelimited="true" forcerange="-1 1"/>')
        eqs.append(f'<joint joint1="f{f}_4_j" joint2="f{f}_3_j" polycoef="0 0.58333 0 0 0"/>')
        fingers.append(body + close)
    floor = '<geom name="floor" type="plane" size="0 0 1" pos="0 0 -0.03"/>' if plane else ""
    return f"""<mujoco><compiler angle="radian"/>
    <default><geom margin="0.0005" condim="3" friction="1 0.005 0.0001"/>
      <joint damping="0.1" armature="0.001" frictionloss="0.001" margin="0.01"/></default>
    <worldbody>{floor}<body name="palm" pos="0 0 0.05"><geom name="palm_g" type="cylinder" size="0.075 0.01"/>{"".join(fingers)}</body></worldbody>
    <actuator>{"".join(acts)}</actuator><equality>{"".join(eqs)}</equality></mujoco>"""


@pytest.mark.parametrize("contacts", [False, True])
def test_forty_body_tree_runs_the_wide_kernels_and_matches_the_oracle(contacts):
    import torch
    import mujoco_amd
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = mujoco_amd.load_model_from_xml(forty_body_tree_xml(plane=contacts))
    assert m.nbody == 42 and m.nv == 40 and m.nu == 32 and m.neq == 8
    if not contacts:
        m = workload.contact_free(m)
    E = 67
    q, v = workload.config2_state(m, E, seed=7)
    v = np.random.default_rng(1).normal(size=v.shape) * 0.5
    c = workload.config2_ctrl(m, E, seed=7)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # no fallback warning: the model is on the tree-parallel path
        sim = BatchSim(m, n_envs=E, max_contacts=64 if contacts else 0)
    fast, path = sim.kernel_path()
    assert fast and (("k_tp_col_w" in path and "k_tpc_w" in path) if contacts else path == "tree-parallel (k_tp_w)"), path
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward(); sim.sync()
    xpos, qacc, nefc = sim.data.xpos.cpu().numpy(), sim.data.qacc.cpu().numpy(), sim.data.nefc.cpu().numpy()
    qM = sim.data.qM.cpu().numpy()
    ncon_tot = 0
    for e in range(0, E, 6):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.forward()
        assert np.abs(xpos[e] - s.xpos).max() <= 1e-12 * max(1.0, np.abs(s.xpos).max()), e
        assert np.abs(qM[e] - s.qM).max() <= 1e-12 * np.abs(s.qM).max(), e
        assert nefc[e] == s.nefc, (e, nefc[e], s.nefc)
        assert np.abs(qacc[e] - s.qacc).max() <= 1e-8 * max(1.0, np.abs(s.qacc).max()), e
        if contacts:
            assert sorted((a, b) for a, b, _ in sim.data.contacts(e, cap=128)) == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts()), e
            ncon_tot += s.ncon
    if contacts:
        assert ncon_tot > 20 and int((sim.data.warning & 8).sum()) == 0
    sim.step(20); sim.sync()
    qg, vg = sim.data.qpos.cpu().numpy(), sim.data.qvel.cpu().numpy()
    assert int((sim.data.warning != 0).sum()) == 0
    for e in (0, 33, 66):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.step(20)
        assert np.abs(qg[e] - s.qpos).max() < 1e-7 and np.abs(vg[e] - s.qvel).max() < 1e-5, e
    sim.close()


def test_more_than_32_contacts_per_environment_take_the_wide_kernels(hand_model):
    """`max_contacts` above 32 used to drop an eligible model onto the generic kernels; up to 64 it now selects the wide
    build, beyond 64 the generic path with its reason."""
    import warnings
    from mujoco_amd import BatchSim
    sim = BatchSim(hand_model, n_envs=8, max_contacts=48)
    assert sim.kernel_path() == (True, "tree-parallel (k_tp_col_w, k_c_narrow_list, k_tpc_w)")
    sim.close()
    with pytest.warns(RuntimeWarning, match="more than 64 contacts"):
        sim = BatchSim(hand_model, n_envs=8, max_contacts=100)
    assert not sim.kernel_path()[0]
    sim.close()
