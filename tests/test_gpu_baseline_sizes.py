"""Every BASELINE.json config at its REAL size on the default code path:
whole-batch invariants (finite, no warnings, contact counters consistent) plus >= 32 sampled
environments against the oracle at the north-star tolerances, contact sets exact.  The samples sit
on the boundaries of the launch geometry: first / last environment of the batch, of the two-stream
ranges, of 64-lane wavefronts (two environments per tree-parallel block, 64 per serial block) and of
the narrowphase work list's sub-lists (block index modulo 64).

PARITY UNPINNED vs MuJoCo 2.1.0: "oracle" is oracle/liboracle.so (in-repo restatement)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _samples(E, extra=()):
    """>= 32 environment ids on the boundaries named above."""
    ids = {0, 1, 2, 3, 62, 63, 64, 65, 126, 127, 128, 129, E - 1, E - 2, E - 64, E - 65, E // 2 - 1, E // 2,
           E // 3, 2 * E // 3, E // 5, 4 * E // 5}
    for np_ in (2, 3):  # range boundaries of split_chunk(E, np) in batchsim.hip (default: 3 ranges from 8192 envs; BS_SPLIT=2 in the split test)
        chunk = (E // np_ + 63) // 64 * 64
        for q in range(1, np_):
            ids |= {q * chunk - 64, q * chunk - 1, q * chunk, q * chunk + 1}
    ids |= {2 * 64 * k + 2 * (k % 7) for k in range(1, 8)}   # different work-list sub-lists, later rounds of blocks
    ids |= set(extra)
    rng = np.random.default_rng(E)
    while len(ids) < 40:
        ids.add(int(rng.integers(0, E)))
    return sorted(i for i in ids if 0 <= i < E)


def _pairs(contacts):
    return sorted((c["geom1"], c["geom2"]) for c in contacts)


def test_config1_4096_envs_1000_steps(hand_model):
    """configs[1]: contacts disabled, 4096 lockstep envs, ctrl resampled every 50 steps, 1000 steps:
    trajectories within 1e-6 (qpos) / 1e-5 (qvel) of the oracle on 36 sampled environments."""
    import torch
    from mujoco_amd import BatchSim, workload
    from oracle import OracleSim
    m = workload.contact_free(hand_model)
    E, periods = 4096, 20
    q, v = workload.config2_state(m, E, seed=1234)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.qvel.copy_(torch.as_tensor(v))
    ctrl = [workload.config2_ctrl(m, E, seed=1234, resample_idx=k) for k in range(periods)]
    bank = torch.as_tensor(np.stack(ctrl)).to(sim.data.ctrl.device)
    for k in range(periods):
        sim.data.ctrl.copy_(bank[k])
        sim.step(50)
    sim.sync()
    qg, vg, tg = sim.data.qpos.cpu().numpy(), sim.data.qvel.cpu().numpy(), sim.data.time.cpu().numpy()
    assert np.isfinite(qg).all() and np.isfinite(vg).all()
    assert int((sim.data.warning != 0).sum()) == 0
    assert np.abs(tg - 1000 * m.opt_timestep).max() < 1e-9
    ids = _samples(E)
    assert len(ids) >= 32
    worst = 0.0
    for e in ids:
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]
        for k in range(periods):
            s.ctrl[:] = ctrl[k][e]
            s.step(50)
        worst = max(worst, np.abs(qg[e] - s.qpos).max())
        assert np.abs(qg[e] - s.qpos).max() <= 1e-6, e
        assert np.abs(vg[e] - s.qvel).max() <= 1e-5, e
    print(f"config1: worst |dqpos| over {len(ids)} sampled envs after 1000 steps = {worst:.2e}")


def _contact_run(m, E, seed, nsteps, gather=None):
    """Forward pass + `nsteps` steps of a contact model at full size on the default path; returns the sim
    and per-stage host copies for the sampled environments."""
    import torch
    from mujoco_amd import BatchSim, workload
    q, v = workload.config2_state(m, E, seed=seed)
    c = workload.config2_ctrl(m, E, seed=seed)
    sim = BatchSim(m, n_envs=E)
    sim.data.qpos.copy_(torch.as_tensor(q)); sim.data.ctrl.copy_(torch.as_tensor(c))
    return sim, q, v, c


def _check_contact_samples(sim, m, q, c, ids, nsteps):
    from oracle import OracleSim
    qg, vg, ag = sim.data.qpos.cpu().numpy(), sim.data.qvel.cpu().numpy(), sim.data.qacc.cpu().numpy()
    ncon, nefc = sim.data.ncon.cpu().numpy(), sim.data.nefc.cpu().numpy()
    tot = 0
    for e in ids:
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.ctrl[:] = c[e]
        s.step(nsteps)
        got = sim.data.contacts(e)
        assert _pairs(s.contacts()) == sorted((g1, g2) for g1, g2, _ in got), e
        assert ncon[e] == s.ncon and nefc[e] == s.nefc, e
        dref = {(cc["geom1"], cc["geom2"]): cc["dist"] for cc in s.contacts()}
        for g1, g2, dist in got:
            assert abs(dist - dref[(g1, g2)]) < 1e-8, e
        tol = 1e-8 if nsteps == 1 else 1e-6
        assert np.abs(ag[e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()) < tol, e
        assert np.abs(qg[e] - s.qpos).max() < (1e-12 if nsteps == 1 else 1e-8), e
        assert np.abs(vg[e] - s.qvel).max() < (1e-9 if nsteps == 1 else 1e-6), e
        tot += s.ncon
    return tot


def test_config2_65536_envs_contacts_default_path(hand_model):
    """configs[2]: ground plane + mesh self-collision, 65 536 envs on the DEFAULT path (three-range
    split engaged by size, work-list narrowphase, tree-parallel dynamics), one step at the one-step
    tolerances (qacc 1e-8, contact sets exact), then 19 more."""
    import os
    from mujoco_amd import workload
    assert "BS_SPLIT" not in os.environ and "BS_NARROW" not in os.environ and "BS_KERNEL" not in os.environ
    m = workload.with_ground_plane(hand_model, z=-0.05)
    E = 65536
    sim, q, v, c = _contact_run(m, E, 1235, 20)
    ids = _samples(E)
    assert len(ids) >= 32
    sim.collision_stats(reset=True)
    sim.step(1); sim.sync()
    st = sim.collision_stats()
    ncon = sim.data.ncon.cpu().numpy()
    assert st["contacts"] == int(ncon.sum()) and st["narrow_pairs"] > E
    assert int((sim.data.warning != 0).sum()) == 0
    assert _check_contact_samples(sim, m, q, c, ids, 1) > 32
    sim.step(19); sim.sync()
    assert np.isfinite(sim.data.qpos.cpu().numpy()).all() and np.isfinite(sim.data.qacc.cpu().numpy()).all()
    assert int((sim.data.warning != 0).sum()) == 0
    st = sim.collision_stats()
    assert st["contacts"] >= int(sim.data.ncon.sum())  # counters accumulate over the 20 steps
    _check_contact_samples(sim, m, q, c, ids, 20)
    print(f"config2: {ncon.mean():.2f} contacts / env, {st['narrow_pairs'] / (20 * E):.2f} narrowphase pairs / env-step")


def test_small_batch_default_path_9000_envs(hand_model):
    """The smallest batches that engage the three-range split (8192 envs and up), ragged: every
    wavefront of the persistent narrowphase is down to its least share of items here (one per lane),
    the last range is shorter than the others and ends inside a 64-environment block."""
    import os
    from mujoco_amd import workload
    assert "BS_SPLIT" not in os.environ and "BS_NL_SHARE" not in os.environ
    m = workload.with_ground_plane(hand_model, z=-0.05)
    E = 9001
    sim, q, v, c = _contact_run(m, E, 1240, 10)
    chunk = (E // 3 + 63) // 64 * 64
    ids = _samples(E, extra=(chunk - 1, chunk, 2 * chunk - 1, 2 * chunk, E - 1, E - 2, 8960, 8959))
    sim.collision_stats(reset=True)
    sim.step(1); sim.sync()
    st = sim.collision_stats()
    assert st["contacts"] == int(sim.data.ncon.sum())
    assert int((sim.data.warning != 0).sum()) == 0
    assert _check_contact_samples(sim, m, q, c, ids, 1) > 32
    sim.step(9); sim.sync()
    assert np.isfinite(sim.data.qpos.cpu().numpy()).all()
    assert int((sim.data.warning != 0).sum()) == 0
    _check_contact_samples(sim, m, q, c, ids, 10)


def test_rk4_contacts_default_path_9001_envs(hand_model):
    """RK4 with contacts on the default path at a size that engages the three-range split: every range
    runs its own state check, four collision + dynamics passes and RK4 advances on its stream."""
    import os
    from mujoco_amd import workload
    from mujoco_amd.model import INT_RK4
    assert "BS_SPLIT" not in os.environ
    m = workload.with_ground_plane(hand_model, z=-0.05).copy()
    m.opt_integrator = INT_RK4
    E = 9001
    sim, q, v, c = _contact_run(m, E, 1241, 4)
    chunk = (E // 3 + 63) // 64 * 64
    ids = _samples(E, extra=(chunk - 1, chunk, 2 * chunk - 1, 2 * chunk, E - 1))
    sim.step(4); sim.sync()
    assert np.isfinite(sim.data.qpos.cpu().numpy()).all()
    assert int((sim.data.warning != 0).sum()) == 0
    assert np.abs(sim.data.time.cpu().numpy() - 4 * m.opt_timestep).max() < 1e-12
    assert _check_contact_samples(sim, m, q, c, ids, 4) > 32


def test_config4_per_gpu_share_131072_envs_with_obs_gather(hand_model):
    """configs[4], one GPU's share: 131 072 envs with contacts + the per-control-period all-gather of
    obs / reward over RCCL (one rank: the collective, its overlap with the next steps and the pack
    kernel run exactly as on 8 GPUs; only the xGMI traffic is absent)."""
    import torch
    import torch.distributed as dist
    from mujoco_amd import workload
    from mujoco_amd import dist as bdist
    from oracle import OracleSim
    m = workload.with_ground_plane(hand_model, z=-0.05)
    E = 131072
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                                device_id=torch.device("cuda:0"))
    try:
        sim, q, v, c = _contact_run(m, E, 1236, 10)
        gather = bdist.ObsGather(sim, 1)
        ids = _samples(E)
        sim.step(5)
        out = gather.all_gather(overlap=True)   # snapshot after 5 steps, in flight under the next 5
        sim.step(5)
        gather.wait(); sim.sync()
        assert int((sim.data.warning != 0).sum()) == 0
        assert out.shape == (E, m.nq + m.nv + 1) and bool(torch.isfinite(out).all())
        _check_contact_samples(sim, m, q, c, ids, 10)
        got = out.cpu().numpy()
        for e in ids[::4]:
            s = OracleSim(m)
            s.qpos[:] = q[e]; s.ctrl[:] = c[e]
            s.step(5)
            assert np.abs(got[e, :m.nq] - s.qpos).max() < 1e-6 and np.abs(got[e, m.nq:m.nq + m.nv] - s.qvel).max() < 1e-4
            want_r = -float(((s.qpos[m.actuator_trnid] - s.ctrl) ** 2).sum())
            assert abs(got[e, -1] - want_r) < 1e-5 * max(1.0, abs(want_r))
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.parametrize("solver", ["newton", "pgs"])
def test_config3_three_hands_4096_envs(solver):
    """configs[3]: the three exported hands in one mixed-model batch, Newton and PGS, self-collision
    on, 4096 environments (1366 + 1365 + 1365), 10 steps, sampled environments of every part."""
    import os
    import torch
    from mujoco_amd import MixedBatchSim, Model, workload
    from mujoco_amd.model import SOL_NEWTON, SOL_PGS
    from oracle import OracleSim
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    models = [Model.load(os.path.join(g, n)) for n in ("dexterous_model.npz", "hand_1120.npz", "hand_1206.npz")]
    for mm in models:
        mm.opt_solver = SOL_NEWTON if solver == "newton" else SOL_PGS
    counts = [1366, 1365, 1365]
    # hand_1120's random poses reach more than 32 contacts in about 1 % of the environments (deep interpenetration of
    # its stand-in meshes): that part gets a 64-contact list, i.e. the WIDE tree-parallel kernels (one environment per
    # wavefront); the other two keep the 32-lane ones.  No environment overflows, so every sampled one is comparable.
    sim = MixedBatchSim(models, counts=counts, max_contacts=[0, 64, 0])
    path = sim.kernel_path()[1]
    assert sim.kernel_path()[0] and path.count("_w") == 2 and "part 1: tree-parallel (k_tp_col_w" in path, path
    init = []
    for k, p in enumerate(sim.parts):
        q, v = workload.config2_state(models[k], p.n_envs, seed=70 + k)
        c = workload.config2_ctrl(models[k], p.n_envs, seed=70 + k)
        p.data.qpos.copy_(torch.as_tensor(q)); p.data.ctrl.copy_(torch.as_tensor(c))
        init.append((q, c))
    sim.step(10)
    sim.sync()
    checked = 0
    for k, p in enumerate(sim.parts):
        qg, ag = p.data.qpos.cpu().numpy(), p.data.qacc.cpu().numpy()
        # bits 0-2 would be mj_check* resets; bit 3 (value 8) a truncated contact or candidate list
        assert np.isfinite(qg).all() and int(((p.data.warning & 7) != 0).sum()) == 0
        assert int(((p.data.warning & 8) != 0).sum()) == 0, k
        if k == 1:
            assert int(p.data.ncon.max()) > 32  # the longer list is really used
        n = p.n_envs
        for e in (0, 1, 63, 64, 65, n // 2, n - 65, n - 64, n - 2, n - 1, 127, 128):
            s = OracleSim(models[k])
            s.qpos[:] = init[k][0][e]; s.ctrl[:] = init[k][1][e]
            s.step(10)
            assert np.abs(qg[e] - s.qpos).max() < 1e-8, (k, e)
            tol = 1e-7 if solver == "newton" else 1e-5  # PGS stops on an improvement test: a flipped stop moves qacc this much
            assert np.abs(ag[e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()) < tol, (k, e)
            assert _pairs(s.contacts()) == sorted((a, b) for a, b, _ in sim.contacts(p.env_start + e)), (k, e)
            checked += 1
    assert checked >= 32
    sim.close()
