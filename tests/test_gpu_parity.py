"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs.  PARITY UNPINNED vs MuJoCo 2.1.0: the oracle is the
in-repo restatement; tolerances are BASELINE.json's north_star tolerances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_forward(model, q, v, c, fields, steps=0, warm=None):
    from oracle import OracleSim
    s = OracleSim(model)
    s.qpos[:] = q
    s.qvel[:] = v
    s.ctrl[:] = c
    if warm is not None:
        s.qacc_warmstart[:] = warm
    if steps:
        s.step(steps)
    else:
        s.forward()
    return {f: np.array(getattr(s, f)).copy() for f in fields}, s


@pytest.fixture(scope="module")
def setup(hand_model):
    import torch
    from mujoco_amd import BatchSim, workload
    assert torch.cuda.is_available()
    m = workload.contact_free(hand_model)
    E = 256
    q, v = workload.config2_state(m, E, seed=1234)
    rng = np.random.default_rng(7)
    v = rng.normal(size=v.shape) * 0.5
    c = workload.config2_ctrl(m, E, seed=1234)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.qvel.copy_(torch.as_tensor(v))
    sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward()
    sim.sync()
    return m, sim, q, v, c


def test_kinematics_1e12(setup):
    """xpos / xquat to 1e-12 relative (north_star tolerance for kinematics)."""
    m, sim, q, v, c = setup
    xpos = sim.data.xpos.cpu().numpy()
    xquat = sim.data.xquat.cpu().numpy()
    xmat = sim.data.xmat.cpu().numpy()
    gx = sim.data.geom_xpos.cpu().numpy()
    for e in range(0, len(q), 17):
        ref, _ = _oracle_forward(m, q[e], v[e], c[e], ["xpos", "xquat", "xmat", "geom_xpos"])
        scale = np.abs(ref["xpos"]).max()
        assert np.abs(xpos[e] - ref["xpos"]).max() <= 1e-12 * scale
        assert np.abs(xquat[e] - ref["xquat"]).max() <= 1e-12
        assert np.abs(xmat[e] - ref["xmat"]).max() <= 1e-12
        assert np.abs(gx[e] - ref["geom_xpos"]).max() <= 1e-12 * scale


def test_mass_matrix_and_bias(setup):
    m, sim, q, v, c = setup
    qM = sim.data.qM.cpu().numpy()
    qLD = sim.data.qLD.cpu().numpy()
    bias = sim.data.qfrc_bias.cpu().numpy()
    smooth = sim.data.qacc_smooth.cpu().numpy()
    af = sim.data.actuator_force.cpu().numpy()
    for e in range(0, len(q), 17):
        ref, _ = _oracle_forward(m, q[e], v[e], c[e], ["qM", "qLD", "qfrc_bias", "qacc_smooth", "actuator_force"])
        assert np.abs(qM[e] - ref["qM"]).max() <= 1e-12 * np.abs(ref["qM"]).max()
        assert np.abs(qLD[e] - ref["qLD"]).max() <= 1e-10 * np.abs(ref["qLD"]).max()
        assert np.abs(bias[e] - ref["qfrc_bias"]).max() <= 1e-12 * max(1e-3, np.abs(ref["qfrc_bias"]).max())
        assert np.abs(af[e] - ref["actuator_force"]).max() <= 1e-12
        assert np.abs(smooth[e] - ref["qacc_smooth"]).max() <= 1e-9 * max(1.0, np.abs(ref["qacc_smooth"]).max())


def test_qacc_1e8(setup):
    """qacc after the constraint solve within 1e-8 (relative to max(1,|qacc|))."""
    m, sim, q, v, c = setup
    qacc = sim.data.qacc.cpu().numpy()
    nefc = sim.data.nefc.cpu().numpy()
    it = sim.data.solver_iter.cpu().numpy()
    worst = 0.0
    for e in range(0, len(q), 5):
        ref, s = _oracle_forward(m, q[e], v[e], c[e], ["qacc", "qfrc_constraint"])
        assert nefc[e] == s.nefc
        err = np.abs(qacc[e] - ref["qacc"]).max() / max(1.0, np.abs(ref["qacc"]).max())
        worst = max(worst, err)
        assert abs(int(it[e]) - s.solver_iter) <= 1
    assert worst <= 1e-8, worst


def test_trajectory_1e6(hand_model):
    """contact-free trajectory within 1e-6 of the oracle over 1000 steps, ctrl resampled
    every 50 steps (configs[1] protocol)."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    E = 64
    q, v = workload.config2_state(m, E, seed=99)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.qvel.copy_(torch.as_tensor(v))
    check = [0, 13, 63]
    orc = []
    for e in check:
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.qvel[:] = v[e]
        orc.append(s)
    for k in range(20):
        c = workload.config2_ctrl(m, E, seed=99, resample_idx=k)
        sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(50)
        for s, e in zip(orc, check):
            s.ctrl[:] = c[e]
            s.step(50)
    sim.sync()
    qg = sim.data.qpos.cpu().numpy()
    vg = sim.data.qvel.cpu().numpy()
    tg = sim.data.time.cpu().numpy()
    assert sim.data.warning.cpu().numpy().max() == 0
    for s, e in zip(orc, check):
        assert np.abs(qg[e] - s.qpos).max() <= 1e-6
        assert np.abs(vg[e] - s.qvel).max() <= 1e-5
        assert abs(tg[e] - s.time) < 1e-9


def test_default_state_golden(hand_model):
    """BASELINE.json configs[0]: default state, zero ctrl, 1000 steps, single env, against
    the committed oracle trajectory (self-generated fixture)."""
    import os
    from conftest import GOLDEN
    from mujoco_amd import BatchSim, workload
    g = np.load(os.path.join(GOLDEN, "dexterous_oracle_traj.npz"))
    m = workload.contact_free(hand_model)
    sim = BatchSim(m, n_envs=1)
    for k in range(10):
        sim.step(100)
        sim.sync()
        assert np.abs(sim.data.qpos.cpu().numpy() - g["qpos"][k]).max() <= 1e-6
        assert np.abs(sim.data.qvel.cpu().numpy() - g["qvel"][k]).max() <= 1e-5


def test_config0_as_shipped_contacts_on(hand_model):
    """BASELINE.json configs[0] AS THE REFERENCE SHIPS IT: /root/reference/output/DexterousDynamos.xml:4 gives
    every mesh geom contype = conaffinity = 1, so `MjSim(model).step()` x 1000 runs self-collision (no plane).
    Single environment (half-empty wavefront), default state, zero ctrl, default kernel path
    (k_tp_col -> k_c_narrow_list -> k_tpc_chain) against the committed oracle trajectory: qpos <= 1e-6,
    contact sets exact, the nine limit rows of SURVEY.md section 8d config 1 active at step 0."""
    import os
    from conftest import GOLDEN
    from mujoco_amd import BatchSim
    g = np.load(os.path.join(GOLDEN, "dexterous_default_contacts_traj.npz"))
    rows = {int(r[0]): r for r in g["rows"]}
    sim = BatchSim(hand_model, n_envs=1)
    assert sim.kernel_path()[0] and "k_tpc" in sim.kernel_path()[1]

    def check_rows(step):
        want = g[f"contacts_step{step}"]
        got = sorted(sim.data.contacts(0))
        assert [(a, b) for a, b, _ in got] == [(int(a), int(b)) for a, b, _ in want], step
        assert np.abs(np.array([d for _, _, d in got]) - want[:, 2]).max() < 1e-9
        nefc, ncon = int(sim.data.nefc.cpu().numpy().ravel()[0]), int(sim.data.ncon.cpu().numpy().ravel()[0])
        assert nefc == int(rows[step][1]) and ncon == int(rows[step][5])
        return nefc, ncon

    sim.step(1)
    sim.sync()
    nefc, ncon = check_rows(1)
    assert nefc - 4 - 21 - 4 * ncon == 9  # the nine joints that sit on a range end at qpos = 0
    sim.step(99)
    sim.sync()
    check_rows(100)
    for k in range(10):
        if k:
            sim.step(100)
            sim.sync()
        assert np.abs(sim.data.qpos.cpu().numpy() - g["qpos"][k]).max() <= 1e-6, k
        assert np.abs(sim.data.qvel.cpu().numpy() - g["qvel"][k]).max() <= 1e-5, k
    check_rows(1000)
    assert int(sim.data.warning.cpu().numpy().ravel()[0]) == 0
    assert abs(float(sim.data.time.cpu().numpy().ravel()[0]) - 2.0) < 1e-9


def test_rk4_matches_oracle(hand_model):
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    m.opt_integrator = 1
    E = 64
    q, v = workload.config2_state(m, E, seed=5)
    c = workload.config2_ctrl(m, E, seed=5)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.step(100)
    sim.sync()
    qg = sim.data.qpos.cpu().numpy()
    for e in (0, 31):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.ctrl[:] = c[e]
        s.step(100)
        assert np.abs(qg[e] - s.qpos).max() <= 1e-7


def test_ragged_env_count_and_reset(hand_model):
    """n_envs not a multiple of the 64-lane wave; masked reset; NaN guard (A14)."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.contact_free(hand_model)
    E = 70
    q, v = workload.config2_state(m, E, seed=3)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.step(10)
    mask = np.zeros(E, dtype=np.uint8)
    mask[[1, 69]] = 1
    sim.reset(mask)
    sim.sync()
    qp = sim.data.qpos.cpu().numpy()
    assert np.all(qp[1] == 0) and np.all(qp[69] == 0) and np.abs(qp[0]).max() > 0
    sim.data.qpos[5, 3] = float("nan")
    sim.step(1)
    sim.sync()
    w = sim.data.warning.cpu().numpy()
    assert w[5] & 1 and w[4] == 0
    assert np.isfinite(sim.data.qpos.cpu().numpy()).all()
    # host copies through bs_copy_field agree with the zero-copy views
    assert np.array_equal(sim.data.to_host("qpos").T, sim.data.qpos.cpu().numpy())


def test_error_behaviour(hand_model):
    from mujoco_amd import BatchSim, BatchSimError, workload
    m = workload.contact_free(hand_model)
    with pytest.raises(ValueError):
        BatchSim(m, n_envs=0)
    sim = BatchSim(m, n_envs=1)
    with pytest.raises(BatchSimError):
        sim.step(-1)
    bad = m.copy()
    bad.opt_solver = 7  # no such solver
    with pytest.raises(BatchSimError):
        BatchSim(bad, n_envs=1)


# ---------------------------------------------------------------------------------------------
# contacts (BASELINE.json configs[2] shape: mesh self-collision + the build's ground plane)
def _contact_model(hand_model):
    return hand_model.copy()


def test_contact_sets_exact(hand_model):
    """Contact sets (geom pair ids) equal the oracle's exactly after sorting; dist/qacc close."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = _contact_model(hand_model)
    E = 128
    q, v = workload.config2_state(m, E, seed=1235)
    c = workload.config2_ctrl(m, E, seed=1235)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward()
    sim.sync()
    ncon = sim.data.ncon.cpu().numpy()
    qacc = sim.data.qacc.cpu().numpy()
    nefc = sim.data.nefc.cpu().numpy()
    assert ncon.sum() > 100  # the workload really has contacts
    near = 0
    for e in range(0, E, 3):
        ref, s = _oracle_forward(m, q[e], v[e] * 0, c[e], ["qacc"])
        want = sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
        got = sim.data.contacts(e)
        assert sorted((g1, g2) for g1, g2, _ in got) == want, e
        assert ncon[e] == s.ncon and nefc[e] == s.nefc
        dref = {(cc["geom1"], cc["geom2"]): cc["dist"] for cc in s.contacts()}
        for g1, g2, dist in got:
            assert abs(dist - dref[(g1, g2)]) < 1e-9
            near += abs(dist - 0.0005) < 1e-6
        err = np.abs(qacc[e] - ref["qacc"]).max() / max(1.0, np.abs(ref["qacc"]).max())
        assert err < 1e-8, (e, err)
    st = sim.collision_stats()
    assert st["narrow_pairs"] > 0 and st["contacts"] == int(ncon.sum())


def test_contact_step_tree_parallel(hand_model, monkeypatch):
    """bs_step with contacts runs the tree-parallel kernel (k_tpc): one and three steps against the
    oracle (qacc <= 1e-8, equal contact sets / nefc), and against the serial contact kernel."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = _contact_model(hand_model)
    E = 131  # ragged: last wavefront half empty
    q, v = workload.config2_state(m, E, seed=4242)
    c = workload.config2_ctrl(m, E, seed=4242)
    out = {}
    for kernel in ("tp", "serial"):
        monkeypatch.setenv("BS_KERNEL", kernel)
        for nstep in (1, 3):
            sim = BatchSim(m, n_envs=E)
            sim.data.qpos.copy_(torch.as_tensor(q))
            sim.data.qvel.copy_(torch.as_tensor(v))
            sim.data.ctrl.copy_(torch.as_tensor(c))
            sim.step(nstep)
            sim.sync()
            out[kernel, nstep] = {k: getattr(sim.data, k).cpu().numpy().copy()
                                  for k in ("qpos", "qvel", "qacc", "ncon", "nefc", "solver_iter", "warning",
                                            "qfrc_constraint")}
            out[kernel, nstep]["contacts"] = [sorted((g1, g2) for g1, g2, _ in sim.data.contacts(e)) for e in range(0, E, 5)]
            del sim
    monkeypatch.delenv("BS_KERNEL")
    for nstep in (1, 3):
        a, b = out["tp", nstep], out["serial", nstep]
        assert a["ncon"].sum() > 100
        assert (a["warning"] == 0).all()
        assert (a["ncon"] == b["ncon"]).all() and (a["nefc"] == b["nefc"]).all()
        assert a["contacts"] == b["contacts"]
        sc = np.maximum(1.0, np.abs(b["qacc"]).max(axis=1, keepdims=True))
        assert (np.abs(a["qacc"] - b["qacc"]) / sc).max() < 1e-8
        assert np.abs(a["qpos"] - b["qpos"]).max() < 1e-10 and np.abs(a["qvel"] - b["qvel"]).max() < 1e-8
    for e in range(0, E, 5):
        for nstep in (1, 3):
            ref, s = _oracle_forward(m, q[e], v[e], c[e], ["qacc", "qpos", "qvel"], steps=nstep)
            a = out["tp", nstep]
            assert a["contacts"][e // 5] == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts()), (e, nstep)
            assert a["ncon"][e] == s.ncon and a["nefc"][e] == s.nefc
            err = np.abs(a["qacc"][e] - ref["qacc"]).max() / max(1.0, np.abs(ref["qacc"]).max())
            assert err < 1e-8, (e, nstep, err)
            assert np.abs(a["qpos"][e] - ref["qpos"]).max() < 1e-10
            assert np.abs(a["qvel"][e] - ref["qvel"]).max() < 1e-8


def test_contact_trajectory(hand_model):
    """200 steps with contacts: the trajectory follows the oracle while contact sets agree."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = _contact_model(hand_model)
    E = 64
    q, v = workload.config2_state(m, E, seed=77)
    c = workload.config2_ctrl(m, E, seed=77)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.step(200)
    sim.sync()
    qg = sim.data.qpos.cpu().numpy()
    assert sim.data.warning.cpu().numpy().max() == 0
    for e in (0, 21, 63):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.ctrl[:] = c[e]
        s.step(200)
        assert np.abs(qg[e] - s.qpos).max() < 1e-5, e


def test_ground_plane_contact(hand_model):
    """The build's own ground plane (not in the reference XML)."""
    import torch
    from mujoco_amd import BatchSim
    from mujoco_amd.model import GEOM_PLANE
    from oracle import OracleSim
    from mujoco_amd import workload
    m = workload.with_ground_plane(hand_model, z=-0.06)
    assert m.npair > 210
    sim = BatchSim(m, n_envs=2)
    s = OracleSim(m)
    # swing the wrist so the fingers sweep below the plane height
    qq = np.zeros(m.nq); qq[0] = -1.3
    sim.data.qpos.copy_(torch.as_tensor(np.stack([qq, qq])))
    s.qpos[:] = qq
    sim.forward(); sim.sync(); s.forward()
    want = sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
    got = sorted((g1, g2) for g1, g2, _ in sim.data.contacts(1))
    assert got == want
    assert any(g1 == m.ngeom - 1 for g1, _ in want)  # the plane really is touched
    # the same through bs_step (tree-parallel contact kernel: the plane's body is the world lane)
    sim.step(2); sim.sync(); s.step(2)
    assert np.abs(sim.data.qpos.cpu().numpy()[1] - s.qpos).max() < 1e-10
    err = np.abs(sim.data.qacc.cpu().numpy()[1] - s.qacc).max() / max(1.0, np.abs(s.qacc).max())
    assert err < 1e-8, err


@pytest.mark.parametrize("solver", ["newton", "cg"])
@pytest.mark.parametrize("kernel", ["tp", "serial"])
def test_rk4_with_contacts(hand_model, kernel, solver, monkeypatch):
    """A13 with contacts: four forward passes per step, collision included in each, against the
    oracle's RK4 (20 steps), through the tree-parallel and the serial contact kernels."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.model import INT_RK4
    from oracle import OracleSim
    monkeypatch.setenv("BS_KERNEL", kernel)
    m = _contact_model(hand_model)
    m.opt_integrator = INT_RK4
    tq, tv = 1e-8, 1e-6
    if solver == "cg":
        from mujoco_amd.model import SOL_CG
        m.opt_solver = SOL_CG
        tq, tv = 1e-6, 1e-4  # CG stops on a loose improvement test; a flipped stop moves the state this much
    E = 16
    q, v = workload.config2_state(m, E, seed=31)
    c = workload.config2_ctrl(m, E, seed=31)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.step(20); sim.sync()
    monkeypatch.delenv("BS_KERNEL")
    assert sim.data.warning.cpu().numpy().max() == 0
    assert sim.data.ncon.cpu().numpy().sum() > 10
    qg, vg, tg = sim.data.qpos.cpu().numpy(), sim.data.qvel.cpu().numpy(), sim.data.time.cpu().numpy()
    for e in (0, 7, 15):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.ctrl[:] = c[e]
        s.step(20)
        assert np.abs(qg[e] - s.qpos).max() < tq, e
        assert np.abs(vg[e] - s.qvel).max() < tv, e
        assert abs(float(np.ravel(tg[e])[0]) - s.time) < 1e-12
        assert sorted((g1, g2) for g1, g2, _ in sim.data.contacts(e)) == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())


@pytest.mark.parametrize("integrator", ["euler", "rk4"])
def test_two_stream_split_is_bitwise_neutral(hand_model, integrator, monkeypatch):
    """BS_SPLIT=n runs ranges of a contact batch on separate streams (overlap of the narrowphase with
    the dynamics kernel); it must not change a single bit of the result.  RK4: each range runs its own
    check / four passes / advances sequence (k_check_state and k_rk4_advance take the range)."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = _contact_model(hand_model)
    if integrator == "rk4":
        from mujoco_amd.model import INT_RK4
        m.opt_integrator = INT_RK4
    E = 389
    q, v = workload.config2_state(m, E, seed=123)
    c = workload.config2_ctrl(m, E, seed=123)
    res = {}
    for split in ("1", "2", "3"):
        monkeypatch.setenv("BS_SPLIT", split)
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(5); sim.sync()
        res[split] = [getattr(sim.data, k).cpu().numpy().copy() for k in ("qpos", "qvel", "qacc", "ncon", "xpos", "time")]
        del sim
    monkeypatch.delenv("BS_SPLIT")
    assert res["1"][3].sum() > 300
    for split in ("2", "3"):
        for a, b in zip(res["1"], res[split]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("contacts", [False, True])
def test_chain_kernel_is_bitwise_neutral(hand_model, monkeypatch, contacts):
    """k_tp_chain / k_tpc_chain (DPP shifts instead of ds_bpermute for the sweeps' packets on the single-child levels of
    the dof tree; chosen when TpModel::chain holds) and k_tp_chain4 (the chain sums by DPP quad permutes as
    well; chosen for hand-shaped models without contacts, TpModel::quad_top) move the same values between the same lanes as k_tp / k_tpc:
    not a bit of the trajectory may differ.  BS_NO_QUAD=1 selects the chain kernels, BS_NO_CHAIN=1 the generic ones."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.with_ground_plane(hand_model, z=-0.05) if contacts else workload.contact_free(hand_model)
    E = 130
    q, v = workload.config2_state(m, E, seed=77)
    c = workload.config2_ctrl(m, E, seed=77)
    res = {}
    # (the contact kernel has no chain4 instantiation: with contacts the default IS k_tpc_chain)
    for mode, var in ((("chain", None),) if contacts else (("chain4", None), ("chain", "BS_NO_QUAD"))) + (("generic", "BS_NO_CHAIN"),):
        if var:
            monkeypatch.setenv(var, "1")
        sim = BatchSim(m.copy(), n_envs=E)  # a fresh native model per mode: the flags are read when the model is built
        if var:
            monkeypatch.delenv(var)
        path = sim.kernel_path()[1]
        stem = "k_tpc" if contacts else "k_tp"
        assert ((stem + "_chain4") in path) == (mode == "chain4") and ((stem + "_chain") in path) == (mode != "generic"), (mode, path)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(20 if contacts else 60); sim.sync()
        res[mode] = [getattr(sim.data, k).cpu().numpy().copy() for k in ("qpos", "qvel", "qacc", "solver_iter", "xpos")]
        if contacts:
            res[mode].append(sim.data.ncon.cpu().numpy().copy())
        del sim
    assert np.abs(res["chain"][1]).max() > 1e-3 and (not contacts or res["chain"][5].sum() > 100)
    for mode in res:
        for a, b in zip(res[mode], res["generic"]):
            assert np.array_equal(a, b), mode


def test_work_stealing_narrowphase(hand_model, monkeypatch):
    """BS_NARROW=ws (opt-in variant of the narrowphase) finds the same contacts as the default."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = _contact_model(hand_model)
    E = 200
    q, v = workload.config2_state(m, E, seed=99)
    c = workload.config2_ctrl(m, E, seed=99)
    res = {}
    for mode in ("flat", "ws"):
        monkeypatch.setenv("BS_NARROW", mode)
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(2); sim.sync()
        res[mode] = (sim.data.qacc.cpu().numpy().copy(), sim.data.ncon.cpu().numpy().copy(),
                     [sorted(sim.data.contacts(e)) for e in range(0, E, 7)])
        del sim
    monkeypatch.delenv("BS_NARROW")
    assert (res["flat"][1] == res["ws"][1]).all() and res["flat"][1].sum() > 100
    for a, b in zip(res["flat"][2], res["ws"][2]):
        assert [(x[0], x[1]) for x in a] == [(x[0], x[1]) for x in b]
        assert all(abs(x[2] - y[2]) < 1e-9 for x, y in zip(a, b))
    assert np.abs(res["flat"][0] - res["ws"][0]).max() < 1e-6 * max(1.0, np.abs(res["flat"][0]).max())


def test_narrowphase_work_list_is_bitwise_neutral(hand_model, monkeypatch):
    """The default narrowphase takes its items from the work list the broadphase kernel appends to
    (order of the list depends on the append race); BS_NARROW=blocks scans the candidate counts per
    block of environments instead.  Every item writes its own slot, so the results are identical
    bit for bit, with and without the ground plane, over several steps (counters cleared in between)."""
    import torch
    from mujoco_amd import BatchSim, workload
    for m in (_contact_model(hand_model), workload.with_ground_plane(hand_model, z=-0.05)):
        E = 333
        q, v = workload.config2_state(m, E, seed=41)
        c = workload.config2_ctrl(m, E, seed=41)
        res = {}
        for mode in ("list", "blocks"):
            monkeypatch.setenv("BS_NARROW", mode)
            sim = BatchSim(m, n_envs=E)
            sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
            sim.step(5); sim.sync()
            res[mode] = [getattr(sim.data, f).cpu().numpy().copy() for f in ("qpos", "qvel", "qacc", "ncon")]
            st = sim.collision_stats()
            res[mode].append(np.array([st["narrow_pairs"], st["mpr_iterations"], st["contacts"]]))
            del sim
        monkeypatch.delenv("BS_NARROW")
        assert res["list"][3].sum() > 100
        for a, b in zip(res["list"], res["blocks"]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("kernel", ["tp", "serial"])
def test_contact_list_overflow(hand_model, kernel, monkeypatch):
    """More contacts than the cap: the list keeps the first `max_contacts` in geom-pair order (the
    oracle's order), the environment gets warning bit 8, environments below the cap are untouched."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    if kernel == "serial":
        monkeypatch.setenv("BS_KERNEL", "serial")
    m = workload.with_ground_plane(hand_model, z=-0.05)
    E, cap = 96, 2
    q, v = workload.config2_state(m, E, seed=77)
    sim = BatchSim(m, n_envs=E, max_contacts=cap)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.step(1); sim.sync()
    ncon = sim.data.ncon.cpu().numpy()
    warn = sim.data.warning.cpu().numpy()
    assert ncon.max() == cap
    over = 0
    for e in range(0, E, 5):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.forward()
        want = [(cc["geom1"], cc["geom2"]) for cc in s.contacts()]
        got = [(g1, g2) for g1, g2, _ in sim.data.contacts(e)]
        assert got == want[:cap]
        assert bool(warn[e] & 8) == (len(want) > cap)
        over += len(want) > cap
    assert over > 3
    assert np.isfinite(sim.data.qacc.cpu().numpy()).all()


@pytest.mark.parametrize("contacts", [False, True])
def test_cg_matches_oracle(hand_model, contacts):
    """solver=CG (mj_solCG: Polak-Ribiere with M^-1 as preconditioner, shared primal loop): device vs
    oracle, iteration counts equal within one, and both at the Newton fixed point."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.model import SOL_CG, SOL_NEWTON
    m = hand_model.copy() if contacts else workload.contact_free(hand_model)
    m.opt_solver = SOL_CG
    E = 64
    q, v = workload.config2_state(m, E, seed=1236)
    c = workload.config2_ctrl(m, E, seed=1236)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward()
    sim.sync()
    qacc = sim.data.qacc.cpu().numpy()
    it = sim.data.solver_iter.cpu().numpy()
    mn = m.copy()
    mn.opt_solver = SOL_NEWTON
    for e in range(0, E, 7):
        ref, s = _oracle_forward(m, q[e], v[e], c[e], ["qacc"])
        sc = max(1.0, np.abs(ref["qacc"]).max())
        assert np.abs(qacc[e] - ref["qacc"]).max() / sc < 1e-8, e
        assert abs(int(it[e]) - s.solver_iter) <= 1
        newton, _ = _oracle_forward(mn, q[e], v[e], c[e], ["qacc"])
        assert np.abs(qacc[e] - newton["qacc"]).max() / sc < 1e-4, e
    sim.step(100)
    sim.sync()
    assert np.isfinite(sim.data.qpos.cpu().numpy()).all()
    assert sim.data.warning.cpu().numpy().max() == 0


@pytest.mark.parametrize("contacts", [False, True])
def test_cg_step_tree_parallel(hand_model, contacts, monkeypatch):
    """bs_step with solver=CG runs in the tree-parallel kernels (k_tp_cg / k_tpc_cg): against the
    oracle stepping the same environments, and against the serial kernels (BS_KERNEL=serial)."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.model import SOL_CG
    from oracle import OracleSim
    m = hand_model.copy() if contacts else workload.contact_free(hand_model)
    m.opt_solver = SOL_CG
    E, n = 40, 5
    q, v = workload.config2_state(m, E, seed=1237)
    c = workload.config2_ctrl(m, E, seed=1237)
    res = {}
    for kern in ("tp", "serial"):
        if kern == "serial":
            monkeypatch.setenv("BS_KERNEL", "serial")
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(n); sim.sync()
        res[kern] = [getattr(sim.data, f).cpu().numpy().copy() for f in ("qpos", "qvel", "qacc", "solver_iter")]
        del sim
    monkeypatch.delenv("BS_KERNEL")
    tol = 1e-6 if contacts else 1e-8
    for a, b in zip(res["tp"][:3], res["serial"][:3]):
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())
    assert np.abs(res["tp"][3].astype(int) - res["serial"][3].astype(int)).max() <= 1
    for e in range(0, E, 9):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.ctrl[:] = c[e]
        s.step(n)
        assert np.abs(res["tp"][0][e] - s.qpos).max() < tol
        assert np.abs(res["tp"][1][e] - s.qvel).max() < tol * max(1.0, np.abs(s.qvel).max())
        assert np.abs(res["tp"][2][e] - s.qacc).max() < 1e-6 * max(1.0, np.abs(s.qacc).max())


@pytest.mark.parametrize("contacts", [False, True])
def test_pgs_matches_oracle(hand_model, contacts):
    """solver=PGS (north_star; BASELINE.json configs[3] 'PGS vs Newton'): device matrix-free PGS vs
    the oracle's explicit-dual PGS, and both near the Newton fixed point."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = hand_model.copy() if contacts else workload.contact_free(hand_model)
    m.opt_solver = 0
    E = 64
    q, v = workload.config2_state(m, E, seed=1235)
    c = workload.config2_ctrl(m, E, seed=1235)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward()
    sim.sync()
    qacc = sim.data.qacc.cpu().numpy()
    it = sim.data.solver_iter.cpu().numpy()
    for e in range(0, E, 7):
        ref, s = _oracle_forward(m, q[e], v[e], c[e], ["qacc"])
        err = np.abs(qacc[e] - ref["qacc"]).max() / max(1.0, np.abs(ref["qacc"]).max())
        assert err < 1e-8, (e, err)
        assert abs(int(it[e]) - s.solver_iter) <= 1
    sim.step(100)
    sim.sync()
    assert np.isfinite(sim.data.qpos.cpu().numpy()).all()


def test_pgs_step_tree_parallel(hand_model, monkeypatch):
    """bs_step with solver=PGS, contact-free, runs in the tree-parallel kernel (k_tp_pgs: M^-1 in
    LDS, sweep in efc order): against the oracle and against the serial kernels, warm starts included."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    m.opt_solver = 0
    E, n = 48, 6
    q, v = workload.config2_state(m, E, seed=1239)
    c = workload.config2_ctrl(m, E, seed=1239)
    res = {}
    for kern in ("tp", "serial"):
        if kern == "serial":
            monkeypatch.setenv("BS_KERNEL", "serial")
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(n); sim.sync()
        res[kern] = [getattr(sim.data, f).cpu().numpy().copy() for f in ("qpos", "qvel", "qacc", "solver_iter")]
        del sim
    monkeypatch.delenv("BS_KERNEL")
    for a, b in zip(res["tp"][:3], res["serial"][:3]):
        assert np.abs(a - b).max() <= 1e-7 * max(1.0, np.abs(b).max())
    assert np.abs(res["tp"][3].astype(int) - res["serial"][3].astype(int)).max() <= 1
    for e in range(0, E, 5):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.ctrl[:] = c[e]
        s.step(n)
        assert np.abs(res["tp"][0][e] - s.qpos).max() < 1e-8
        assert np.abs(res["tp"][1][e] - s.qvel).max() < 1e-7 * max(1.0, np.abs(s.qvel).max())
        assert np.abs(res["tp"][2][e] - s.qacc).max() < 1e-6 * max(1.0, np.abs(s.qacc).max())


def test_mixed_model_batch(hand_model):
    """BASELINE.json configs[3] shape: several models in one batch, environments sorted by model,
    one bs_step for all.  Three variants of the hand (contact-free Newton, self-collision Newton,
    contact-free PGS with heavier damping) against their own oracles."""
    import torch
    from mujoco_amd import MixedBatchSim, BatchSimError, workload
    from oracle import OracleSim
    m0 = workload.contact_free(hand_model)
    m1 = hand_model.copy()
    m2 = workload.contact_free(hand_model)
    m2.opt_solver = 0
    m2.dof_damping = m2.dof_damping * 3.0
    models, counts = [m0, m1, m2], [70, 33, 25]
    sim = MixedBatchSim(models, counts=counts)
    assert len(sim.parts) == 3 and [p.n_envs for p in sim.parts] == counts
    assert [p.env_start for p in sim.parts] == [0, 70, 103]
    init = []
    for k, p in enumerate(sim.parts):
        q, v = workload.config2_state(models[k], p.n_envs, seed=50 + k)
        c = workload.config2_ctrl(models[k], p.n_envs, seed=50 + k)
        p.data.qpos.copy_(torch.as_tensor(q)); p.data.ctrl.copy_(torch.as_tensor(c))
        init.append((q, c))
    sim.step(20)
    sim.sync()
    for k, p in enumerate(sim.parts):
        qg = p.data.qpos.cpu().numpy()
        assert p.data.warning.cpu().numpy().max() == 0
        for e in (0, p.n_envs - 1):
            s = OracleSim(models[k])
            s.qpos[:] = init[k][0][e]; s.ctrl[:] = init[k][1][e]
            s.step(20)
            assert np.abs(qg[e] - s.qpos).max() < 1e-8, (k, e)
            if k == 1:
                got = sorted((g1, g2) for g1, g2, _ in sim.contacts(p.env_start + e))
                assert got == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
    assert sim.collision_stats()["contacts"] > 0
    # masked reset addresses environments of the whole batch
    mask = np.zeros(sim.n_envs, dtype=np.uint8)
    mask[71] = 1
    sim.reset(mask)
    sim.sync()
    assert np.abs(sim.parts[1].data.qpos.cpu().numpy()[1] - m1.qpos0).max() == 0
    assert np.abs(sim.parts[1].data.qpos.cpu().numpy()[0] - m1.qpos0).max() > 0
    with pytest.raises(BatchSimError):
        sim.stage_times()
    with pytest.raises(BatchSimError):
        MixedBatchSim(models, env_model_id=[0, 2, 1])
    sim.close()


def test_three_exported_hands_in_one_batch():
    """BASELINE.json configs[3] with the real models: the exporter restatement's three hands
    (tests/golden/*.npz) stepped in one mixed batch, self-collision on, against their oracles."""
    import os
    import torch
    from mujoco_amd import MixedBatchSim, Model, workload
    from oracle import OracleSim
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    models = [Model.load(os.path.join(g, n)) for n in ("dexterous_model.npz", "hand_1120.npz", "hand_1206.npz")]
    sim = MixedBatchSim(models, counts=[6, 6, 6])
    init = []
    for k, p in enumerate(sim.parts):
        q, v = workload.config2_state(models[k], p.n_envs, seed=70 + k)
        c = workload.config2_ctrl(models[k], p.n_envs, seed=70 + k)
        p.data.qpos.copy_(torch.as_tensor(q)); p.data.ctrl.copy_(torch.as_tensor(c))
        init.append((q, c))
    sim.step(10)
    sim.sync()
    for k, p in enumerate(sim.parts):
        qg, ag = p.data.qpos.cpu().numpy(), p.data.qacc.cpu().numpy()
        for e in (0, 5):
            s = OracleSim(models[k])
            s.qpos[:] = init[k][0][e]; s.ctrl[:] = init[k][1][e]
            s.step(10)
            assert np.abs(qg[e] - s.qpos).max() < 1e-9, (k, e)
            assert np.abs(ag[e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()) < 1e-7, (k, e)
            assert sorted((a, b) for a, b, _ in sim.contacts(p.env_start + e)) == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
    sim.close()


def test_mixed_parts_concurrent_is_bitwise_neutral(monkeypatch):
    """The parts of a mixed-model sim with contacts step on their own streams (forked from and joined to
    the sim's stream); BS_MIXED_SERIAL=1 runs them one after the other.  Same bits, and work enqueued on
    the sim's stream afterwards (reset, a later step) sees the finished parts."""
    import os
    import torch
    from mujoco_amd import MixedBatchSim, Model, workload
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    models = [Model.load(os.path.join(g, n)) for n in ("dexterous_model.npz", "hand_1120.npz", "hand_1206.npz")]
    out = {}
    for mode in ("concurrent", "serial"):
        if mode == "serial":
            monkeypatch.setenv("BS_MIXED_SERIAL", "1")
        sim = MixedBatchSim(models, counts=[70, 130, 66])
        for k, p in enumerate(sim.parts):
            q, v = workload.config2_state(models[k], p.n_envs, seed=90 + k)
            c = workload.config2_ctrl(models[k], p.n_envs, seed=90 + k)
            p.data.qpos.copy_(torch.as_tensor(q)); p.data.ctrl.copy_(torch.as_tensor(c))
        for _ in range(3):
            sim.step(7)
            sim.step(1)
        sim.sync()
        out[mode] = [(p.data.qpos.cpu().numpy().copy(), p.data.qvel.cpu().numpy().copy(), p.data.qacc.cpu().numpy().copy(),
                      p.data.ncon.cpu().numpy().copy()) for p in sim.parts]
        sim.close()
    monkeypatch.delenv("BS_MIXED_SERIAL")
    for a, b in zip(out["concurrent"], out["serial"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert sum(int(a[3].sum()) for a in out["concurrent"]) > 0


def test_vec_env_rollout(hand_model):
    """RL glue: observation / reward definitions, frame skip, device-side
    auto-reset; the physics under it is the same bs_step as everywhere else."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.env import HandVecEnv
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    E = 32
    sim = BatchSim(m, n_envs=E)
    env = HandVecEnv(sim, frame_skip=10, episode_steps=3)
    obs = env.reset()
    assert obs.shape == (E, m.nq + m.nv) and obs.dtype == torch.float32
    assert torch.equal(obs[:, :m.nq].double(), torch.as_tensor(m.qpos0).to(obs.device).expand(E, -1).double().float().double())
    g = torch.Generator(device=obs.device); g.manual_seed(3)
    lo, hi = env.ctrl_lo, env.ctrl_hi
    act = lo + (hi - lo) * torch.rand((E, m.nu), generator=g, device=obs.device, dtype=torch.float64)
    o1, r1, d1, info = env.step(act * 3.0 - 1.0)  # out-of-range parts are clamped like MuJoCo clamps ctrl
    clamped = torch.minimum(torch.maximum(act * 3.0 - 1.0, lo), hi)
    s = OracleSim(m)
    s.ctrl[:] = clamped[5].cpu().numpy()
    s.step(10)
    assert np.abs(o1[5, :m.nq].cpu().numpy() - s.qpos).max() < 1e-6
    want_r = -float(((s.qpos[m.actuator_trnid] - s.ctrl) ** 2).sum())
    assert abs(float(r1[5]) - want_r) < 1e-5 * max(1.0, abs(want_r))
    assert not bool(d1.any())
    env.step(act)
    o3, r3, d3, info = env.step(act)
    assert bool(d3.all())  # episode_steps = 3
    assert torch.equal(o3[:, :m.nq].double(), obs[:, :m.nq].double()) and float(o3[:, m.nq:].abs().max()) == 0.0
    assert float(info["terminal_obs"][:, m.nq:].abs().max()) > 0
    assert float(sim.data.time.max()) == 0.0 and int(env.t.max()) == 0
    # a NaN in one environment: mj_check* resets it, the env reports done for that one only
    sim.data.qvel[7, 2] = float("nan")
    o4, r4, d4, info = env.step(act)
    assert bool(d4[7]) and int(d4.sum()) == 1 and bool(info["failed"][7])
    assert int(sim.data.warning.abs().max()) == 0


def test_simplified_hulls_keep_parity(hand_model):
    """The opt-in hull simplification changes the geometry, not the agreement: device and oracle on
    the SAME simplified model still find the same contacts and accelerations."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.meshproc import simplify_model_hulls
    m = simplify_model_hulls(hand_model, 64)
    E = 40
    q, v = workload.config2_state(m, E, seed=8)
    c = workload.config2_ctrl(m, E, seed=8)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.step(3); sim.sync()
    qacc = sim.data.qacc.cpu().numpy()
    total = 0
    for e in range(0, E, 4):
        ref, s = _oracle_forward(m, q[e], v[e] * 0, c[e], ["qacc"], steps=3)
        assert sorted((a, b) for a, b, _ in sim.data.contacts(e)) == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
        assert np.abs(qacc[e] - ref["qacc"]).max() / max(1.0, np.abs(ref["qacc"]).max()) < 1e-8
        total += s.ncon
    assert total > 5


def test_headless_run_simulation_records_trajectory(hand_model, tmp_path):
    """f-4: the viewer loop of `Mujoco_XML.run_simulation` replaced by a state recorder; the
    recorded frames equal the oracle's trajectory (contact-free tolerance 1e-6) and round-trip
    through .npz and .csv."""
    import torch
    from mujoco_amd import workload
    from mujoco_amd.recorder import run_simulation
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    E, n = 8, 40
    c = workload.config2_ctrl(m, E, seed=5)

    def ctrl_fn(i, sim):
        if i == 0:
            sim.data.ctrl.copy_(torch.as_tensor(c))

    rec = run_simulation(m, n, n_envs=E, ctrl_fn=ctrl_fn, record_every=10, envs=[1, 6], out=str(tmp_path / "traj.npz"))
    z = np.load(tmp_path / "traj.npz")
    assert list(z["frame_step"]) == [0, 10, 20, 30, 40] and list(z["env"]) == [1, 6]
    assert z["qpos"].shape == (5, 2, m.nq) and z["time"].shape == (5, 2)
    assert np.all(z["qpos"][0] == m.qpos0) and np.all(z["ctrl"][0] == 0)
    for k, e in enumerate([1, 6]):
        s = OracleSim(m)
        s.ctrl[:] = c[e]
        for f in range(1, 5):
            s.step(10)
            assert np.abs(z["qpos"][f, k] - s.qpos).max() < 1e-6
            assert np.abs(z["qvel"][f, k] - s.qvel).max() < 1e-6
            assert abs(z["time"][f, k] - s.time) < 1e-12
    rec.save(str(tmp_path / "traj.csv"))
    t = np.loadtxt(tmp_path / "traj.csv", delimiter=",", skiprows=1)
    head = open(tmp_path / "traj.csv").readline().strip().split(",")
    assert head[:4] == ["step", "env", "time", "qpos_0"] and t.shape == (10, 3 + 2 * m.nq + m.nu)
    assert np.array_equal(t[:, 3:3 + m.nq].reshape(5, 2, m.nq), z["qpos"])
    # a single environment keeps mujoco-py's shapes in sim.data; the recorder restores the env axis
    r1 = run_simulation(m, 5, n_envs=1)
    assert r1.arrays()["qpos"].shape == (6, 1, m.nq)


def test_pack_obs_kernel(hand_model):
    """bs_pack_obs (obs = qpos | qvel fp32, reward = -|qpos[actuated] - ctrl|^2) against the same
    thing written with torch ops (`mujoco_amd.dist.pack_obs`); row stride wider than the row."""
    import torch
    from mujoco_amd import BatchSim, BatchSimError, workload
    from mujoco_amd.dist import pack_obs
    m = workload.contact_free(hand_model)
    E = 130
    q, v = workload.config2_state(m, E, seed=8)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(workload.config2_ctrl(m, E, seed=8)))
    sim.step(7)
    d = sim.data
    want = pack_obs(d.qpos, d.qvel, d.ctrl, torch.as_tensor(m.actuator_trnid, dtype=torch.long, device=d.qpos.device))
    buf = torch.full((E, m.nq + m.nv + 5), -7.0, dtype=torch.float32, device=d.qpos.device)
    got = sim.pack_obs(buf)
    torch.cuda.synchronize()
    W = m.nq + m.nv + 1
    assert torch.equal(got[:, :W - 1], want[:, :W - 1])
    assert torch.allclose(got[:, W - 1], want[:, W - 1], rtol=1e-6, atol=0)
    assert bool((got[:, W:] == -7.0).all())
    with pytest.raises(ValueError):
        sim.pack_obs(torch.empty((E, W - 1), dtype=torch.float32, device=d.qpos.device))
    with pytest.raises(ValueError):
        sim.pack_obs(torch.empty((E, W), dtype=torch.float64, device=d.qpos.device))


@pytest.mark.parametrize("contacts", [False, True])
def test_long_rollout_invariants(hand_model, contacts):
    """Soak: thousands of steps with ctrl resampled every 50 (the bench's loop): no NaN, no
    mj_check* reset, joints stay inside their ranges up to the soft-constraint penetration."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.with_ground_plane(hand_model, z=-0.05) if contacts else workload.contact_free(hand_model)
    E, periods = 512, (20 if contacts else 80)
    q, v = workload.config2_state(m, E, seed=4321)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    bank = torch.as_tensor(np.stack([workload.config2_ctrl(m, E, seed=4321, resample_idx=k) for k in range(4)])).to(sim.data.ctrl.device)
    for p in range(periods):
        sim.data.ctrl.copy_(bank[p % 4])
        sim.step(50)
    sim.sync()
    qp, qv = sim.data.qpos.cpu().numpy(), sim.data.qvel.cpu().numpy()
    assert np.isfinite(qp).all() and np.isfinite(qv).all()
    assert int((sim.data.warning != 0).sum().item()) == 0
    assert np.maximum(m.jnt_range[:, 0] - qp, qp - m.jnt_range[:, 1]).max() < 0.05
    assert np.abs(qv).max() < 50.0
    assert abs(float(sim.data.time[0]) - periods * 50 * m.opt_timestep) < 1e-9


def test_by_name_accessors(hand_model):
    """mujoco-py's PyMjData helpers (get/set_joint_qpos, get_body_xpos, ...) over the batched views."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.contact_free(hand_model)
    jn, bn = m.joint_names[5], m.body_names[9]
    sim = BatchSim(m, n_envs=3)
    sim.data.set_joint_qpos(jn, torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64))
    sim.forward(); sim.sync()
    assert torch.equal(sim.data.get_joint_qpos(jn).cpu(), torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64))
    assert torch.equal(sim.data.qpos[:, m.get_joint_qpos_addr(jn)], sim.data.get_joint_qpos(jn))
    assert sim.data.get_body_xpos(bn).shape == (3, 3) and sim.data.get_body_xmat(bn).shape == (3, 3, 3)
    assert torch.equal(sim.data.get_body_xpos(bn), sim.data.xpos[:, m.body_name2id(bn)])
    one = BatchSim(m, n_envs=1)
    one.data.set_joint_qvel(jn, 0.5)
    assert float(one.data.get_joint_qvel(jn)) == 0.5 and one.data.get_body_xquat(bn).shape == (4,)
    assert one.data.get_body_xmat(bn).shape == (3, 3)


@pytest.mark.parametrize("kernel", ["tp", "serial"])
def test_nan_guard_with_contacts(hand_model, kernel, monkeypatch):
    """mj_checkPos on the contact path (k_tp_col treats the bad state as reset, k_tpc resets it): the
    poisoned environment is reset and flagged, its neighbours are bit-for-bit what they are without it."""
    import torch
    from mujoco_amd import BatchSim, workload
    if kernel == "serial":
        monkeypatch.setenv("BS_KERNEL", "serial")
    m = _contact_model(hand_model)
    E = 66
    q, v = workload.config2_state(m, E, seed=31)
    c = workload.config2_ctrl(m, E, seed=31)
    out = []
    for poison in (False, True):
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(3)
        if poison:
            sim.data.qpos[3, 7] = float("nan")
            sim.data.qvel[65, 2] = float("inf")
        sim.step(3); sim.sync()
        out.append((sim.data.qpos.cpu().numpy().copy(), sim.data.warning.cpu().numpy().copy(), sim.data.time.cpu().numpy().copy()))
    (q0, w0, t0), (q1, w1, t1) = out
    assert np.isfinite(q1).all()
    assert w1[3] & 1 and w1[65] & 2 and (w1[[0, 1, 2, 4, 64]] == 0).all() and (w0 == 0).all()
    keep = np.ones(E, bool); keep[[3, 65]] = False
    assert np.array_equal(q0[keep], q1[keep])
    assert abs(t1[3] - 3 * m.opt_timestep) < 1e-12 and abs(t1[0] - 6 * m.opt_timestep) < 1e-12  # reset restarts the clock


@pytest.mark.parametrize("solver", ["cg", "pgs"])
def test_rk4_with_cg_and_pgs_tree_parallel(hand_model, solver, monkeypatch):
    """RK4 with the non-default solvers, contact-free, in the tree-parallel kernels (k_tp_rk4_cg /
    k_tp_rk4_pgs): against the serial kernels and the oracle."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.model import SOL_CG, SOL_PGS
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    m.opt_solver = SOL_CG if solver == "cg" else SOL_PGS
    m.opt_integrator = 1
    E, n = 34, 4
    q, v = workload.config2_state(m, E, seed=1241)
    c = workload.config2_ctrl(m, E, seed=1241)
    res = {}
    for kern in ("tp", "serial"):
        if kern == "serial":
            monkeypatch.setenv("BS_KERNEL", "serial")
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(n); sim.sync()
        res[kern] = [getattr(sim.data, f).cpu().numpy().copy() for f in ("qpos", "qvel")]
        del sim
    monkeypatch.delenv("BS_KERNEL")
    for a, b in zip(res["tp"], res["serial"]):
        assert np.abs(a - b).max() <= 1e-7 * max(1.0, np.abs(b).max())
    for e in range(0, E, 8):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.ctrl[:] = c[e]
        s.step(n)
        assert np.abs(res["tp"][0][e] - s.qpos).max() < 1e-8
        assert np.abs(res["tp"][1][e] - s.qvel).max() < 1e-7 * max(1.0, np.abs(s.qvel).max())


@pytest.mark.parametrize("plane", [False, True])
def test_pgs_with_contacts_tree_parallel(hand_model, plane, monkeypatch):
    """bs_step with solver=PGS and contacts runs k_tpc_pgs (round 1: serial kernels only): contact rows
    swept in contact order with M^-1 in LDS, against the serial kernels and the oracle (warm starts
    included: several steps), ragged environment count."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.with_ground_plane(hand_model, z=-0.05) if plane else hand_model.copy()
    m.opt_solver = 0
    E, n = 67, 4
    q, v = workload.config2_state(m, E, seed=1242)
    c = workload.config2_ctrl(m, E, seed=1242)
    res = {}
    for kern in ("tp", "serial"):
        monkeypatch.setenv("BS_KERNEL", kern)
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(1); sim.sync()
        one = [getattr(sim.data, f).cpu().numpy().copy() for f in ("qacc", "solver_iter", "ncon", "nefc")]
        sim.step(n - 1); sim.sync()
        res[kern] = one + [getattr(sim.data, f).cpu().numpy().copy() for f in ("qpos", "qvel", "qacc", "solver_iter", "warning")]
        del sim
    monkeypatch.delenv("BS_KERNEL")
    a, b = res["tp"], res["serial"]
    assert a[2].sum() > 100 and (a[2] == b[2]).all() and (a[3] == b[3]).all() and (a[8] == 0).all()
    sc = np.maximum(1.0, np.abs(b[0]).max(axis=1, keepdims=True))
    assert (np.abs(a[0] - b[0]) / sc).max() < 1e-6            # one step: same sweep, rounding-level differences
    assert np.abs(a[1].astype(int) - b[1].astype(int)).max() <= 1
    assert np.abs(a[4] - b[4]).max() < 1e-7 and np.abs(a[5] - b[5]).max() < 1e-5
    for e in range(0, E, 6):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.ctrl[:] = c[e]
        s.step(1)
        assert np.abs(a[0][e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()) < 1e-6, e
        assert abs(int(a[1][e]) - s.solver_iter) <= 1 and a[2][e] == s.ncon
        s.step(n - 1)
        assert np.abs(a[4][e] - s.qpos).max() < 1e-7 and np.abs(a[5][e] - s.qvel).max() < 1e-5, e


def test_vec_env_fused_period_matches_torch_path(hand_model):
    """HandVecEnv.step: the fused control period (bs_env_pre, bs_step, bs_env_post) against the same period
    written with torch operations — observations, rewards, done / failed / overflow flags, terminal
    observations, episode clocks and the state left behind, over episode ends and a mj_check* failure."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.env import HandVecEnv
    for m in (workload.contact_free(hand_model), workload.with_ground_plane(hand_model, z=-0.05)):
        E = 96
        sims = [BatchSim(m, n_envs=E) for _ in range(2)]
        envs = [HandVecEnv(s, frame_skip=5, episode_steps=3) for s in sims]
        o0, o1 = envs[0].reset(), envs[1].reset()
        assert torch.equal(o0, o1)
        g = torch.Generator(device=o0.device); g.manual_seed(11)
        for period in range(8):
            a = torch.rand((E, m.nu), generator=g, device=o0.device, dtype=torch.float64) * 3.0 - 1.0   # beyond ctrlrange too
            if period == 4:  # poison one environment of both sims: mj_checkPos resets it inside the period
                for s in sims:
                    s.data.qpos[7, 2] = float("nan")
            of, rf, df, inf_ = envs[0].step(a)
            ot, rt, dt, int_ = envs[1]._step_torch(a)
            assert torch.equal(df, dt), period
            assert torch.equal(inf_["failed"], int_["failed"]) and torch.equal(inf_["contact_overflow"], int_["contact_overflow"])
            if period == 4:
                assert bool(inf_["failed"][7]) and bool(df[7])
            assert torch.equal(of, ot), period
            assert torch.equal(inf_["terminal_obs"], int_["terminal_obs"]), period
            assert torch.allclose(rf, rt, rtol=1e-6, atol=1e-9), period
            assert torch.equal(envs[0].t, envs[1].t)
            for f in ("qpos", "qvel", "ctrl", "qacc_warmstart", "time", "warning"):
                assert torch.equal(getattr(sims[0].data, f), getattr(sims[1].data, f)), (period, f)
        for s in sims:
            s.close()


def test_non_unit_model_quaternions_are_rescaled_on_the_host(hand_model):
    """The tree-parallel kernels' closed-form joint pose and single normalisation per chain pose assume unit body / inertial
    quaternions and joint axes (mjModel's are: MuJoCo's compiler normalises them).  Arrays a few digits short of unit — as a
    caller of bs_model_create might pass — are rescaled when the lane record is built: same poses and mass matrix."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.contact_free(hand_model)
    m2 = m.copy()
    m2.body_quat = m.body_quat * (1.0 + 1e-4 * np.arange(m.nbody))[:, None]
    m2.body_iquat = m.body_iquat * 0.999
    m2.jnt_axis = m.jnt_axis * 1.0005
    E = 64
    q, v = workload.config2_state(m, E, seed=5)
    c = workload.config2_ctrl(m, E, seed=5)
    out = []
    for mm in (m, m2):
        sim = BatchSim(mm, n_envs=E)
        assert sim.kernel_path()[0]
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.forward(); sim.sync()
        out.append({f: getattr(sim.data, f).cpu().numpy().copy() for f in ("xpos", "xquat", "qM", "qacc")})
        sim.close()
    for f in ("xpos", "xquat", "qM", "qacc"):
        a, b = out
        assert np.abs(a[f] - b[f]).max() <= 1e-12 * max(1.0, np.abs(a[f]).max()), f


_SLIDE_HINGE_XML = """<mujoco><compiler angle="radian"/><option timestep="0.002"/>
<default><joint damping="0.2" armature="0.01" frictionloss="0.002" margin="0.01"/><geom density="800" contype="0" conaffinity="0"/></default>
<worldbody>
 <body name="base" pos="0 0 0.5" quat="0.9 0.1 0.3 0.2">
  <joint name="s0" type="slide" axis="1 2 0.5" pos="0.01 0.02 0.03" range="-0.2 0.2" limited="true"/>
  <geom type="box" size="0.05 0.03 0.02" pos="0.01 0 0.02" quat="0.8 0.2 0.1 0.5"/>
  <body name="l1" pos="0.1 0 0.05" quat="0.7 0.5 0.1 0.4">
   <joint name="h1" type="hinge" axis="0.3 1 0.2" pos="0.02 -0.01 0.03" range="-1 1" limited="true"/>
   <geom type="capsule" size="0.02 0.06" pos="0 0.02 0.05" quat="0.9 0.3 0.2 0.1"/>
   <body name="l2" pos="0 0.05 0.12" quat="0.6 0.2 0.7 0.1">
    <joint name="s2" type="slide" axis="0 0.4 1" pos="0.01 0.01 0" range="-0.1 0.1" limited="true"/>
    <geom type="box" size="0.02 0.04 0.03" pos="0.01 0.01 0.02" quat="0.5 0.5 0.1 0.7"/>
    <body name="l3" pos="0.03 0 0.08" quat="0.2 0.9 0.3 0.1">
     <joint name="h3" type="hinge" axis="1 0 0.1" pos="0 0.01 -0.02" range="-1.2 0.8" limited="true"/>
     <geom type="capsule" size="0.015 0.05" pos="0.01 0 0.04" quat="0.7 0.1 0.6 0.3"/>
    </body>
   </body>
   <body name="m2" pos="0.04 -0.05 0.1" quat="0.3 0.1 0.2 0.9">
    <joint name="h4" type="hinge" axis="0 1 1" pos="0.01 0 0.01" range="-0.7 0.7" limited="true"/>
    <geom type="box" size="0.02 0.02 0.05" pos="0 0.01 0.05" quat="0.9 0.1 0.4 0.1"/>
   </body>
  </body>
 </body>
</worldbody>
<actuator><position joint="s0" kp="50"/><position joint="h1" kp="5"/><position joint="s2" kp="30"/><position joint="h3" kp="2"/></actuator>
</mujoco>"""


@pytest.mark.parametrize("env", [{}, {"BS_KERNEL": "wide"}])
def test_slide_and_hinge_tree_with_rotated_frames_matches_the_oracle(env, monkeypatch):
    """Slide and hinge joints mixed in one branching tree, every body / joint / inertial frame rotated and offset: the
    closed-form joint-local pose (TPK_P1 / P2 / Q1), the body-frame inertia tensor (TPK_IB) and the root-body reference
    point of the step kernels against the oracle's mj_kinematics / mj_comPos / mj_crb arithmetic."""
    import torch
    import mujoco_amd
    from mujoco_amd import BatchSim
    from oracle import OracleSim
    m = mujoco_amd.load_model_from_xml(_SLIDE_HINGE_XML)
    assert m.nv == 5 and m.nbody == 6
    E = 70
    rng = np.random.default_rng(11)
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    q = lo + (hi - lo) * rng.uniform(-0.1, 1.1, size=(E, m.nq))   # some beyond their limits
    v = rng.normal(size=(E, m.nv)) * 0.5
    c = rng.uniform(-0.5, 0.5, size=(E, m.nu))
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    sim = BatchSim(m, n_envs=E)
    for k in env:
        monkeypatch.delenv(k)
    fast, path = sim.kernel_path()
    assert fast and ("_w" in path) == bool(env), path
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward(); sim.sync()
    xpos, xquat, qM, qacc, bias = (getattr(sim.data, f).cpu().numpy() for f in ("xpos", "xquat", "qM", "qacc", "qfrc_bias"))
    for e in range(0, E, 3):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.forward()
        assert np.abs(xpos[e] - s.xpos).max() <= 1e-12 and np.abs(xquat[e] - s.xquat).max() <= 1e-12, e
        assert np.abs(qM[e] - s.qM).max() <= 1e-12 * np.abs(s.qM).max(), e
        assert np.abs(bias[e] - s.qfrc_bias).max() <= 1e-12 * max(1.0, np.abs(s.qfrc_bias).max()), e
        assert np.abs(qacc[e] - s.qacc).max() <= 1e-8 * max(1.0, np.abs(s.qacc).max()), e
    sim.step(50); sim.sync()   # the step kernels (root-body reference point)
    qg, vg = sim.data.qpos.cpu().numpy(), sim.data.qvel.cpu().numpy()
    assert int((sim.data.warning != 0).sum()) == 0
    for e in range(0, E, 7):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.step(50)
        assert np.abs(qg[e] - s.qpos).max() < 1e-9 and np.abs(vg[e] - s.qvel).max() < 1e-7, (e, np.abs(qg[e] - s.qpos).max(), np.abs(vg[e] - s.qvel).max())
    sim.close()
