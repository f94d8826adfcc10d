"""GPU tests of the mujoco-py-shaped surface and of the exported
intermediates (rows A2 / A7) against the oracle, through the C-ABI.  PARITY UNPINNED vs MuJoCo
2.1.0: the oracle is the in-repo restatement (oracle/oracle.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _efc_row_map(m, s):
    """Oracle efc rows (active rows only, efc order) -> row index in the device's fixed layout
    (active equalities | dofs with frictionloss | lower, upper limit of every limited joint)."""
    eq_ids = [e for e in range(m.neq) if m.eq_active[e]]
    fric = [i for i in range(m.nv) if m.dof_frictionloss[i] > 0]
    lim = [j for j in range(m.njnt) if m.jnt_limited[j]]
    typ, ids, J = s.efc("type"), s.efc("id"), s.efc("J")
    out = []
    for i in range(len(typ)):
        if typ[i] == 0:
            out.append(eq_ids.index(ids[i]))
        elif typ[i] == 1:
            out.append(len(eq_ids) + fric.index(ids[i]))
        elif typ[i] == 3:
            side = 0 if J[i, ids[i]] > 0 else 1
            out.append(len(eq_ids) + len(fric) + 2 * lim.index(ids[i]) + side)
        else:
            out.append(-1)  # contact rows: not part of the fixed layout
    return np.array(out)


@pytest.mark.parametrize("kernel", ["tp", "serial"])
def test_exported_intermediates_vs_oracle(hand_model, kernel, monkeypatch):
    """A2 / A7 / A8 intermediates that bs_forward exports and round 1 never compared: subtree_com,
    cinert, cdof, cvel, xipos, xanchor, xaxis, efc_D, efc_aref, efc_force, qfrc_constraint."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    monkeypatch.setenv("BS_KERNEL", kernel)
    m = workload.contact_free(hand_model)
    E = 96
    q, v = workload.config2_state(m, E, seed=21)
    v = np.random.default_rng(3).normal(size=v.shape) * 0.5
    c = workload.config2_ctrl(m, E, seed=21)
    q[:, 3] = m.jnt_range[3, 0] + 0.002   # inside the limit margin: active limit rows
    q[::2, 9] = m.jnt_range[9, 1] - 0.004
    sim = BatchSim(m, n_envs=E)
    monkeypatch.delenv("BS_KERNEL")
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward(); sim.sync()
    got = {f: getattr(sim.data, f).cpu().numpy() for f in
           ("rootcom", "cinert", "cdof", "cvel", "xipos", "xanchor", "xaxis", "efc_D", "efc_aref", "efc_force",
            "qfrc_constraint", "nefc", "qfrc_passive", "qfrc_actuator", "qfrc_smooth")}
    root = int(m.body_rootid[m.nbody - 1])
    for e in range(0, E, 7):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.forward()
        assert np.abs(got["rootcom"][e][root] - s.subtree_com[root]).max() <= 1e-12
        for f, tol in (("cinert", 1e-12), ("cdof", 1e-12), ("cvel", 1e-11), ("xipos", 1e-12), ("xanchor", 1e-12), ("xaxis", 1e-12)):
            ref = getattr(s, f)
            assert np.abs(got[f][e] - ref).max() <= tol * max(1.0, np.abs(ref).max()), (f, e)
        for f in ("qfrc_passive", "qfrc_actuator", "qfrc_smooth"):
            assert np.abs(got[f][e] - getattr(s, f)).max() <= 1e-12 * max(1.0, np.abs(getattr(s, f)).max()), (f, e)
        rows = _efc_row_map(m, s)
        assert (rows >= 0).all() and got["nefc"][e] == s.nefc == len(rows)
        assert s.nefc > 25  # limit rows are active
        for f, tol in (("efc_D", 1e-12), ("efc_aref", 1e-10), ("efc_force", 1e-8)):
            ref = s.efc(f[4:])
            dev = got[f][e][rows]
            assert np.abs(dev - ref).max() <= tol * max(1.0, np.abs(ref).max()), (f, e, np.abs(dev - ref).max())
        inactive = np.setdiff1d(np.arange(got["efc_D"].shape[1]), rows)
        assert (got["efc_D"][e][inactive] == 0).all()
        assert np.abs(got["qfrc_constraint"][e] - s.qfrc_constraint).max() <= 1e-8 * max(1.0, np.abs(s.qfrc_constraint).max())


@pytest.mark.parametrize("contacts", [False, True])
def test_state_round_trip_bit_exact(hand_model, contacts):
    """MjSimState (A15): 100 steps -> get_state -> 100 steps -> set_state -> 100 steps gives the same
    bits as the first continuation (qacc_warmstart is part of the snapshot), with and without contacts;
    a state without the warm start restores time/qpos/qvel only."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.with_ground_plane(hand_model, z=-0.05) if contacts else workload.contact_free(hand_model)
    E = 130
    q, v = workload.config2_state(m, E, seed=17)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.ctrl.copy_(torch.as_tensor(workload.config2_ctrl(m, E, seed=17)))
    sim.step(100)
    st = sim.get_state()
    assert st.qpos.shape == (E, m.nq) and st.qvel.shape == (E, m.nv) and st.time.shape == (E,) and st.act is None
    assert torch.equal(st.qpos, sim.data.qpos) and torch.equal(st.udd_state["qacc_warmstart"], sim.data.qacc_warmstart)
    sim.step(100); sim.sync()
    first = [getattr(sim.data, f).clone() for f in ("qpos", "qvel", "qacc", "time")]
    if contacts:
        assert int(sim.data.ncon.sum()) > 50
    sim.set_state(st)
    assert torch.equal(sim.data.qpos, st.qpos) and abs(float(sim.data.time[0]) - 100 * m.opt_timestep) < 1e-12
    sim.step(100); sim.sync()
    for a, f in zip(first, ("qpos", "qvel", "qacc", "time")):
        assert torch.equal(a, getattr(sim.data, f)), f
    # the snapshot is a copy: stepping did not change it; a hand-made state restores what it holds
    from mujoco_amd.sim import MjSimState
    sim.set_state(MjSimState(0.25, np.zeros(m.nq), np.zeros(m.nv), None, {}))
    sim.sync()
    assert float(sim.data.qpos.abs().max()) == 0 and float(sim.data.time.min()) == 0.25
    one = BatchSim(m, n_envs=1)
    one.step(3)
    s1 = one.get_state()
    assert s1.qpos.shape == (m.nq,) and isinstance(s1.time, float) and abs(s1.time - 3 * m.opt_timestep) < 1e-15
    one.step(2); one.set_state(s1); one.sync()
    assert torch.equal(one.data.qpos, s1.qpos)


def test_contact_accessor_single_env(hand_model):
    """`sim.data.contact[i].geom1 / geom2 / dist / pos / frame` for n_envs == 1 (mujoco-py's shape)."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.with_ground_plane(hand_model, z=-0.06)
    qq = np.zeros(m.nq); qq[0] = -1.3
    sim = BatchSim(m, n_envs=1)
    sim.data.qpos.copy_(torch.as_tensor(qq))
    sim.forward(); sim.sync()
    s = OracleSim(m)
    s.qpos[:] = qq
    s.forward()
    want = s.contacts()
    con = sim.data.contact
    assert len(con) == int(sim.data.ncon[0]) == len(want) > 0
    for a, b in zip(con, want):
        assert (a.geom1, a.geom2) == (b["geom1"], b["geom2"])
        assert abs(a.dist - b["dist"]) < 1e-9
        assert np.abs(a.pos - b["pos"]).max() < 1e-9 and np.abs(a.frame - b["frame"]).max() < 1e-9
    two = BatchSim(m, n_envs=2)
    with pytest.raises(AttributeError):
        two.data.contact
    assert two.data.contact_list(1) == []


def test_device_mask_reset_and_split4_ragged(hand_model, monkeypatch):
    """bs_reset_device_mask = bs_reset with the same mask; BS_SPLIT=4 with an environment count that
    does not fill four 64-multiple ranges (520 envs) is bitwise what one range gives."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = hand_model.copy()
    E = 520
    q, v = workload.config2_state(m, E, seed=5)
    c = workload.config2_ctrl(m, E, seed=5)
    res = {}
    for split in ("1", "4", "3"):
        monkeypatch.setenv("BS_SPLIT", split)
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(4); sim.sync()
        res[split] = [getattr(sim.data, k).cpu().numpy().copy() for k in ("qpos", "qvel", "qacc", "ncon", "warning")]
        if split != "1":
            del sim
        else:
            keep = sim
    monkeypatch.delenv("BS_SPLIT")
    assert res["1"][3].sum() > 300 and (res["1"][4] == 0).all()
    for split in ("4", "3"):
        for a, b in zip(res["1"], res[split]):
            assert np.array_equal(a, b)
    # masked reset: host mask and device mask agree, untouched environments keep their bits
    mask = np.zeros(E, dtype=np.uint8); mask[[0, 63, 64, 519]] = 1
    other = BatchSim(m, n_envs=E)
    other.set_state(keep.get_state())
    other.data.ctrl.copy_(keep.data.ctrl)
    keep.reset(mask)
    other.reset(torch.as_tensor(mask, device=other.data.qpos.device).bool())
    keep.sync(); other.sync()
    for f in ("qpos", "qvel", "qacc", "qacc_warmstart", "ctrl", "time", "warning"):
        assert torch.equal(getattr(keep.data, f), getattr(other.data, f)), f
    assert float(keep.data.qpos[63].abs().max()) == 0 and float(keep.data.qpos[62].abs().max()) > 0
    # two masked resets back to back reuse the staging buffer safely
    m2 = np.zeros(E, dtype=np.uint8); m2[5] = 1
    m3 = np.zeros(E, dtype=np.uint8); m3[6] = 1
    keep.reset(m2); keep.reset(m3); keep.sync()
    assert float(keep.data.qpos[5].abs().max()) == 0 and float(keep.data.qpos[6].abs().max()) == 0
    assert float(keep.data.qpos[7].abs().max()) > 0


def test_check_acc_paths_pinned(hand_model, monkeypatch):
    """mj_checkAcc (A14) on one poisoned environment, every kernel family.  Contact-free kernels do what
    mj_step does (reset, forward pass from the reset state, integrate: time = h afterwards); the
    contact kernels reset and skip that step's integration (the next step recomputes the collision
    from the reset pose), so time stays 0 — DESIGN.md §7 states this deviation; this test pins both."""
    import torch
    from mujoco_amd import BatchSim, workload
    h = hand_model.opt_timestep
    out = {}
    for label, model, kern in (("cf_tp", workload.contact_free(hand_model), "tp"), ("cf_serial", workload.contact_free(hand_model), "serial"),
                               ("con_tp", hand_model.copy(), "tp"), ("con_serial", hand_model.copy(), "serial")):
        monkeypatch.setenv("BS_KERNEL", kern)
        E = 8
        q, v = workload.config2_state(model, E, seed=2)
        sim = BatchSim(model, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q))
        sim.step(2)
        sim.data.qfrc_applied[3, 4] = float("inf")  # bad qacc in environment 3 only
        sim.step(1); sim.sync()
        out[label] = (sim.data.warning.cpu().numpy().copy(), sim.data.time.cpu().numpy().copy(),
                      sim.data.qpos.cpu().numpy().copy(), sim.data.qvel.cpu().numpy().copy())
    monkeypatch.delenv("BS_KERNEL")
    for label, (w, t, qp, qv) in out.items():
        assert w[3] & 4 and (np.delete(w, 3) == 0).all(), label
        assert np.isfinite(qp).all() and np.isfinite(qv).all(), label
        assert abs(t[2] - 3 * h) < 1e-12
        if label.startswith("cf"):
            assert abs(t[3] - h) < 1e-15, label       # reset, forward, integrate (mj_step)
        else:
            assert t[3] == 0.0 and np.abs(qv[3]).max() == 0, label  # reset, integration skipped
    assert np.allclose(out["cf_tp"][2][3], out["cf_serial"][2][3], atol=1e-12)
    assert np.array_equal(out["con_tp"][2][3], out["con_serial"][2][3])


@pytest.mark.parametrize("solver", ["cg", "pgs"])
def test_forward_cg_pgs_on_tree_parallel_kernels(hand_model, solver, monkeypatch):
    """bs_forward with CG / PGS runs k_tp_full_cg / k_tp_full_pgs (round 1: serial kernels only):
    the full dump against the serial kernels and the oracle."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.model import SOL_CG, SOL_PGS
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    m.opt_solver = SOL_CG if solver == "cg" else SOL_PGS
    E = 66
    q, v = workload.config2_state(m, E, seed=8)
    v = np.random.default_rng(8).normal(size=v.shape) * 0.3
    c = workload.config2_ctrl(m, E, seed=8)
    res = {}
    for kern in ("tp", "serial"):
        monkeypatch.setenv("BS_KERNEL", kern)
        sim = BatchSim(m, n_envs=E)
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.forward(); sim.sync()
        res[kern] = {f: getattr(sim.data, f).cpu().numpy().copy() for f in
                     ("qacc", "qfrc_constraint", "efc_force", "efc_D", "efc_aref", "qM", "qLD", "cinert", "solver_iter", "nefc")}
    monkeypatch.delenv("BS_KERNEL")
    a, b = res["tp"], res["serial"]
    assert (a["nefc"] == b["nefc"]).all() and np.abs(a["solver_iter"].astype(int) - b["solver_iter"].astype(int)).max() <= 1
    for f, tol in (("qM", 1e-12), ("qLD", 1e-10), ("cinert", 1e-12), ("efc_D", 1e-12), ("efc_aref", 1e-10), ("qacc", 1e-6),
                   ("qfrc_constraint", 1e-6), ("efc_force", 1e-6)):
        assert np.abs(a[f] - b[f]).max() <= tol * max(1.0, np.abs(b[f]).max()), f
    for e in range(0, E, 11):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.forward()
        assert np.abs(a["qacc"][e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()) < 1e-8, e


def test_kernel_path_is_reported_and_fallback_warns(hand_model, monkeypatch):
    """bs_sim_kernel_path: an eligible model runs the tree-parallel kernels; the generic fallback is
    not silent (BatchSim warns, the reason is readable)."""
    import warnings
    from mujoco_amd import BatchSim, workload

    free = workload.contact_free(hand_model)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sim = BatchSim(free, n_envs=4)
        fast, path = sim.kernel_path()
        assert fast and "k_tp" in path
        sim.close()
        sim = BatchSim(hand_model, n_envs=4)
        fast, path = sim.kernel_path()
        assert fast and "k_tpc" in path
        sim.close()
    monkeypatch.setenv("BS_KERNEL", "serial")
    with pytest.warns(RuntimeWarning, match="generic one-environment-per-lane"):
        sim = BatchSim(free, n_envs=4)
    fast, path = sim.kernel_path()
    assert not fast and "BS_KERNEL=serial" in path and "k_step" in path
    sim.close()
    monkeypatch.delenv("BS_KERNEL")


def test_solver_and_integrator_kwargs(hand_model):
    """`BatchSim(model, n_envs, solver=, integrator=)` override model.opt_* for this sim (the
    choice a user of the reference makes through Mujoco_XML.add_option, src/Mujoco_XML.py:137-145); the model passed
    in is untouched, the kernel that runs is the solver's, and the result equals a sim built from an edited model."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.model import INT_RK4, SOL_NEWTON, SOL_PGS
    free = workload.contact_free(hand_model)
    E = 66
    q, v = workload.config2_state(free, E, seed=8)
    c = workload.config2_ctrl(free, E, seed=8)

    def run(sim):
        sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
        sim.step(5); sim.sync()
        out = sim.data.qpos.clone(), sim.data.qacc.clone()
        sim.close()
        return out

    sim = BatchSim(free, n_envs=E, solver="pgs")
    assert sim.kernel_path() == (True, "tree-parallel (k_tp_pgs)") and sim.model.opt_solver == SOL_PGS
    assert free.opt_solver == SOL_NEWTON  # the caller's model is left as it was
    a = run(sim)
    edited = free.copy(); edited.opt_solver = SOL_PGS
    b = run(BatchSim(edited, n_envs=E))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    sim = BatchSim(free, n_envs=E, solver="CG", integrator="RK4")
    assert sim.kernel_path() == (True, "tree-parallel (k_tp_rk4_cg)") and sim.model.opt_integrator == INT_RK4
    sim.close()
    sim = BatchSim(hand_model, n_envs=E, solver=0)  # mjtSolver numbers work too: PGS with contacts
    assert sim.kernel_path() == (True, "tree-parallel (k_tp_col, k_c_narrow_list, k_tpc_pgs)")
    sim.close()
    sim = BatchSim(free, n_envs=E, device_ids=[0])
    assert sim.device_id == 0 and sim.kernel_path()[1] == "tree-parallel (k_tp_chain4)"
    sim.close()
    for bad in (dict(solver="sor"), dict(integrator="implicit"), dict(solver=5), dict(device_ids=[0, 0])):
        with pytest.raises(ValueError):
            BatchSim(free, n_envs=E, **bad)


def test_footprint_and_named_out_of_memory(hand_model):
    """bs_sim_footprint: bytes per arena (the figures of the DESIGN.md section 5 table); an arena that does not fit
    fails bs_sim_create with BS_ERR_NOMEM and a message naming it."""
    from mujoco_amd import BatchSim, BatchSimError, workload
    free = workload.contact_free(hand_model)
    sim = BatchSim(free, n_envs=4096)
    f = sim.footprint()
    assert f["contacts"] == 0 and f["geom_frames"] == 0 and f["serial_solver"] == 0
    assert f["state"] == f["total"] and 20_000 < f["bytes_per_env"] < 30_000
    sim.close()
    sim = BatchSim(workload.with_ground_plane(hand_model), n_envs=4096)
    g = sim.footprint()
    assert g["contacts"] > 0 and g["geom_frames"] == 4096 * 23 * 96 and g["serial_solver"] == 0
    assert 40_000 < g["bytes_per_env"] < 60_000
    sim.close()
    with pytest.raises(BatchSimError, match="state arena .* bytes per environment"):
        BatchSim(free, n_envs=40_000_000)  # ~ 1 TB of state: more than the 288 GB of an MI355X


def test_data_of_a_destroyed_sim_raises_clearly(hand_model):
    """`sim.data` is a borrowed view: after the sim is closed or collected an access raises BatchSimError, not a bare
    ReferenceError or a use of freed device memory."""
    from mujoco_amd import BatchSim, BatchSimError, workload
    free = workload.contact_free(hand_model)
    d = BatchSim(free, n_envs=4).data  # the sim is collected at once
    import gc
    gc.collect()
    with pytest.raises(BatchSimError, match="destroyed"):
        d.qpos
    sim = BatchSim(free, n_envs=4)
    d = sim.data
    assert d.qpos.shape == (4, free.nq)
    sim.close()
    with pytest.raises(BatchSimError, match="destroyed"):
        d.contacts(0)


def test_mixed_kernel_path_aggregates_parts(hand_model, monkeypatch):
    """bs_sim_kernel_path on a mixed-model handle (it dereferenced the top-level sim's null model): one
    description per part, fast only if every part is; MixedBatchSim warns when a part is on the serial kernels."""
    from mujoco_amd import MixedBatchSim, workload
    a = workload.contact_free(hand_model)
    b = a.copy(); b.dof_damping = b.dof_damping * 2.0
    sim = MixedBatchSim([a, b], counts=[6, 5])
    fast, path = sim.kernel_path()
    assert fast and path.count("k_tp_chain") == 2 and "part 0:" in path and "part 1:" in path
    sim.close()
    monkeypatch.setenv("BS_KERNEL", "serial")
    with pytest.warns(RuntimeWarning, match="generic one-environment-per-lane"):
        sim = MixedBatchSim([a, b], counts=[6, 5])
    fast, path = sim.kernel_path()
    assert not fast and path.count("BS_KERNEL=serial") == 2
    sim.close()
    monkeypatch.delenv("BS_KERNEL")


def test_sharded_batch_sim_two_shards_one_device(hand_model):
    """`device_ids=[...]` in one process (mujoco_amd.dist.ShardedBatchSim): contiguous shards, one sim per entry, no
    traffic between them.  One GPU here, so both shards sit on device 0 — the result must equal the unsharded batch."""
    import torch
    from mujoco_amd import BatchSim, workload
    from mujoco_amd.dist import ShardedBatchSim
    E = 131
    q, v = workload.config2_state(hand_model, E, seed=21)
    c = workload.config2_ctrl(hand_model, E, seed=21)
    sh = ShardedBatchSim(hand_model, E, device_ids=[0, 0])
    assert sh.env_range == [(0, 66), (66, 131)] and [s.n_envs for s in sh.shards] == [66, 65]
    sh.scatter("qpos", q); sh.scatter("ctrl", c)
    sh.step(4)
    one = BatchSim(hand_model, n_envs=E)
    one.data.qpos.copy_(torch.as_tensor(q)); one.data.ctrl.copy_(torch.as_tensor(c))
    one.step(4); one.sync()
    assert np.array_equal(sh.gather("qpos"), one.data.qpos.cpu().numpy())
    assert np.array_equal(sh.gather("ncon"), one.data.ncon.cpu().numpy()) and sh.gather("ncon").sum() > 0
    mask = np.zeros(E, dtype=np.uint8); mask[[0, 65, 66, 130]] = 1
    sh.reset(mask)
    t = sh.gather("time")
    assert (t[mask == 1] == 0).all() and (t[mask == 0] > 0).all()
    sh.close(); one.close()


@pytest.mark.parametrize("solver", ["newton", "cg", "pgs"])
def test_forward_with_contacts_writes_every_intermediate_on_tp_kernels(hand_model, solver):
    """bs_forward of a contact model on the tree-parallel path (k_tp_col, narrowphase, k_tpc_full[_cg|_pgs]): the
    intermediates a mj_forward leaves behind — poses, cinert / cdof / cvel, the fixed efc rows, forces, qacc and
    the contact list — against the oracle, and the same call again after steps."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.with_ground_plane(hand_model, z=-0.05)
    if solver != "newton":
        import copy
        from mujoco_amd.model import SOL_CG, SOL_PGS
        m = copy.deepcopy(m)
        m.opt_solver = SOL_CG if solver == "cg" else SOL_PGS
    E = 70
    q, v = workload.config2_state(m, E, seed=31)
    v = np.random.default_rng(5).normal(size=v.shape) * 0.3
    c = workload.config2_ctrl(m, E, seed=31)
    sim = BatchSim(m, n_envs=E)
    fast, path = sim.kernel_path()
    assert fast and "k_tpc" in path
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v)); sim.data.ctrl.copy_(torch.as_tensor(c))
    sim.forward(); sim.sync()
    # no eligible configuration allocates the serial solver's 43 KB per environment (PGS used to, round 2)
    assert sim.footprint()["serial_solver"] == 0
    fields = ("xpos", "xquat", "rootcom", "cinert", "cdof", "cvel", "xipos", "xanchor", "xaxis", "efc_D", "efc_aref",
              "qfrc_constraint", "nefc", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "qacc", "ncon")
    got = {f: getattr(sim.data, f).cpu().numpy() for f in fields}
    root = int(m.body_rootid[m.nbody - 1])
    ncon_tot = 0
    for e in range(0, E, 6):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = c[e]
        s.forward()
        assert np.abs(got["xpos"][e].reshape(-1) - np.asarray(s.xpos).reshape(-1)).max() <= 1e-12
        assert np.abs(got["rootcom"][e].reshape(-1, 3)[root] - s.subtree_com[root]).max() <= 1e-12
        for f, tol in (("cinert", 1e-12), ("cdof", 1e-12), ("cvel", 1e-11), ("xipos", 1e-12), ("xanchor", 1e-12), ("xaxis", 1e-12)):
            ref = np.asarray(getattr(s, f)).reshape(-1)
            assert np.abs(got[f][e].reshape(-1) - ref).max() <= tol * max(1.0, np.abs(ref).max()), (f, e)
        for f, tol in (("qfrc_bias", 1e-11), ("qfrc_passive", 1e-12), ("qfrc_actuator", 1e-12), ("qfrc_smooth", 1e-11), ("qacc_smooth", 1e-9)):
            ref = np.asarray(getattr(s, f))
            assert np.abs(got[f][e] - ref).max() <= tol * max(1.0, np.abs(ref).max()), (f, e)
        assert got["nefc"][e] == s.nefc and got["ncon"][e] == s.ncon
        ncon_tot += s.ncon
        assert sorted((a, b) for a, b, _ in sim.data.contacts(e)) == sorted((cc["geom1"], cc["geom2"]) for cc in s.contacts())
        rows = _efc_row_map(m, s)
        fixed = rows >= 0
        for f, tol in (("efc_D", 1e-12), ("efc_aref", 1e-10)):
            ref = s.efc(f[4:])[fixed]
            assert np.abs(got[f][e][rows[fixed]] - ref).max() <= tol * max(1.0, np.abs(ref).max()), (f, e)
        tol = 1e-8 if solver == "newton" else 1e-6
        if solver == "pgs":  # efc_force of the fixed rows, straight from the sweep
            ref = s.efc("force")[fixed]
            assert np.abs(sim.data.efc_force.cpu().numpy()[e][rows[fixed]] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), e
        assert np.abs(got["qacc"][e] - s.qacc).max() <= tol * max(1.0, np.abs(s.qacc).max()), e
        assert np.abs(got["qfrc_constraint"][e] - s.qfrc_constraint).max() <= tol * max(1.0, np.abs(s.qfrc_constraint).max()), e
    assert ncon_tot > 10
    # forward again after steps: same call, state advanced
    sim.step(5); sim.forward(); sim.sync()
    assert np.isfinite(sim.data.qacc.cpu().numpy()).all() and np.isfinite(sim.data.cinert.cpu().numpy()).all()
    sim.close()
