"""The CPU oracle (oracle/) against independent formulations: the reference holds no golden
vectors for dynamics (PARITY UNPINNED), so the oracle is pinned by identities
and third-opinion computations, plus its own committed trajectory fixture."""
import os

import numpy as np
import pytest

from mujoco_amd import workload
from mujoco_amd.model import host_fk, host_jac, host_mass_matrix
from oracle import OracleSim
from conftest import GOLDEN


@pytest.fixture(scope="module")
def cf_model(hand_model):
    return workload.contact_free(hand_model)


def _random_state(m, seed):
    rng = np.random.default_rng(seed)
    lo, hi = m.jnt_range[:, 0], m.jnt_range[:, 1]
    return lo + rng.random(m.nq) * (hi - lo), rng.normal(size=m.nv)


def test_fk_mass_matrix_vs_jacobian_formulation(cf_model):
    m = cf_model
    for seed in range(5):
        q, v = _random_state(m, seed)
        s = OracleSim(m)
        s.qpos[:] = q
        s.forward()
        fk = host_fk(m, q)
        assert np.abs(s.xpos - fk["xpos"]).max() < 1e-15
        assert np.abs(s.xquat - fk["xquat"]).max() < 1e-14
        assert np.abs(s.xmat.reshape(-1, 3, 3) - fk["xmat"]).max() < 1e-14
        M = host_mass_matrix(m, fk)          # sum_b m Jp^T Jp + Jr^T I Jr  (not CRB)
        Mo = s.full_m()
        assert np.abs(Mo - M).max() < 1e-15
        assert np.allclose(Mo, Mo.T) and np.linalg.eigvalsh(Mo).min() > 0
        x = np.random.default_rng(seed).normal(size=m.nv)
        assert np.abs(s.solve_m(x) - np.linalg.solve(M, x)).max() < 1e-9 * np.abs(np.linalg.solve(M, x)).max()
        assert np.abs(s.mul_m(x) - M @ x).max() < 1e-16 + 1e-14 * np.abs(M @ x).max()


def test_bias_forces_vs_lagrangian(cf_model):
    """qfrc_bias = C(q,v) + G(q): G from body Jacobians, C from finite-differenced M."""
    m = cf_model
    q, v = _random_state(m, 3)
    s = OracleSim(m)
    s.qpos[:] = q
    s.qvel[:] = v
    s.forward()
    fk = host_fk(m, q)
    G = np.zeros(m.nv)
    for b in range(1, m.nbody):
        jp, _ = host_jac(m, fk, b, fk["xipos"][b])
        G -= m.body_mass[b] * jp.T @ m.opt_gravity
    h = 1e-6
    dM = []
    for k in range(m.nv):
        qp, qm = q.copy(), q.copy()
        qp[k] += h
        qm[k] -= h
        dM.append((host_mass_matrix(m, host_fk(m, qp)) - host_mass_matrix(m, host_fk(m, qm))) / (2 * h))
    dM = np.array(dM)  # [k, i, j] = dM_ij/dq_k
    C = np.einsum("kij,k,j->i", dM, v, v) - 0.5 * np.einsum("iab,a,b->i", dM, v, v)
    # armature contributes nothing to C; compare
    assert np.abs(s.qfrc_bias - (C + G)).max() < 1e-9
    # rne with acceleration: M qacc + bias
    s.qacc[:] = np.linspace(-1, 1, m.nv)
    full = s.rne(1)
    want = (s.full_m() - np.diag(m.dof_armature)) @ s.qacc + s.qfrc_bias
    assert np.abs(full - want).max() < 1e-14


def test_actuation_and_passive(cf_model):
    m = cf_model
    q, v = _random_state(m, 4)
    s = OracleSim(m)
    s.qpos[:] = q
    s.qvel[:] = v
    s.ctrl[:] = 10.0  # far outside ctrlrange -> clamped
    s.forward()
    j = m.actuator_trnid
    want = np.clip(2.0 * np.clip(10.0, m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]) - 2.0 * q[j], -1, 1)
    assert np.abs(s.actuator_force - want).max() < 1e-15
    assert np.abs(s.qfrc_actuator[j] - want).max() < 1e-15
    assert np.abs(s.qfrc_passive + 0.1 * v).max() < 1e-16


def test_constraint_rows_and_kkt(cf_model):
    """Row order equality -> frictionloss -> limit; the Newton solution is the minimiser of
    the convex cost: checked against scipy's L-BFGS from a different start and by gradient."""
    from scipy.optimize import minimize
    m = cf_model
    s = OracleSim(m)
    # default pose: 9 joints sit on a limit -> 4 + 21 + 9 rows
    s.forward()
    assert s.nefc == 34
    t = s.efc("type")
    assert t.tolist() == [0] * 4 + [1] * 21 + [3] * 9
    q, v = _random_state(m, 8)
    s.qpos[:] = q
    s.qvel[:] = 0.3 * v
    s.forward()
    qa = s.qacc.copy()
    c0 = s.cost(qa)
    rng = np.random.default_rng(0)
    scale = np.abs(qa).max()
    for _ in range(20):
        assert s.cost(qa + 1e-4 * scale * rng.normal(size=m.nv)) >= c0 - 1e-9 * abs(c0)
    # gradient at the solution: M a - qfrc_smooth - J^T f = 0
    J, f = s.efc("J"), s.efc("force")
    g = s.full_m() @ qa - s.qfrc_smooth - J.T @ f
    assert np.abs(g).max() * (1.0 / (m.stat_meaninertia * m.nv)) < 1e-6
    assert np.abs(J.T @ f - s.qfrc_constraint).max() < 1e-12
    res = minimize(lambda a: s.cost(a), s.qacc_smooth.copy(), method="L-BFGS-B",
                   options=dict(maxiter=5000, ftol=1e-15, gtol=1e-12))
    assert s.cost(res.x) >= c0 - 1e-7 * max(1.0, abs(c0))


def test_limit_closed_form_single_dof(tmp_path):
    """1-DOF limit hit: the solver's acceleration equals the closed-form minimiser."""
    import mujoco_amd
    xml = """<mujoco><compiler angle="radian"/><option gravity="0 0 0"><flag contact="disable"/></option>
    <worldbody><body name="a"><joint name="j" axis="0 1 0" limited="true" range="-0.5 0.5" margin="0.01"/>
    <geom type="sphere" size="0.1" pos="0 0 0.3"/></body></worldbody></mujoco>"""
    m = mujoco_amd.load_model_from_xml(xml)
    s = OracleSim(m)
    s.qpos[0] = 0.52
    s.qvel[0] = 0.1
    s.forward()
    assert s.nefc == 1
    D, aref, J = s.efc("D")[0], s.efc("aref")[0], s.efc("J")[0, 0]
    M = s.full_m()[0, 0]
    a_s = s.qacc_smooth[0]
    # minimise 0.5 M (a-a_s)^2 + 0.5 D (J a - aref)^2  (active since J a - aref < 0)
    a = (M * a_s + D * J * aref) / (M + D * J * J)
    assert J * a - aref < 0
    assert abs(s.qacc[0] - a) < 1e-9 * abs(a)


def test_energy_drift_contact_free(tmp_path):
    """No damping/friction/limits/actuators: RK4 conserves energy to O(h^4); Euler drifts O(h)."""
    import mujoco_amd
    xml = """<mujoco><compiler angle="radian"/><option timestep="0.001" integrator="{I}"><flag contact="disable"/></option>
    <worldbody><body name="a" pos="0 0 1"><joint name="j1" axis="0 1 0"/><geom type="capsule" size="0.02 0.2" pos="0 0 -0.2"/>
      <body name="b" pos="0 0 -0.4"><joint name="j2" axis="0 1 0"/><geom type="capsule" size="0.02 0.15" pos="0 0 -0.15"/></body>
    </body></worldbody></mujoco>"""

    def energy(m, s):
        s.forward()
        T = 0.5 * s.qvel @ s.full_m() @ s.qvel
        V = -sum(m.body_mass[b] * m.opt_gravity @ s.xipos[b] for b in range(1, m.nbody))
        return T + V

    drift = {}
    for integ in ("RK4", "Euler"):
        m = mujoco_amd.load_model_from_xml(xml.replace("{I}", integ))
        s = OracleSim(m)
        s.qpos[:] = [1.0, 0.5]
        e0 = energy(m, s)
        s.step(2000)
        drift[integ] = abs(energy(m, s) - e0) / abs(e0)
    assert drift["RK4"] < 1e-8
    assert drift["Euler"] < 5e-2 and drift["Euler"] > drift["RK4"]


def test_rk4_convergence_order():
    """Smooth double pendulum (no clamps, no constraints): halving h cuts the error >= 12x."""
    import mujoco_amd
    xml = """<mujoco><compiler angle="radian"/><option timestep="{H}" integrator="RK4"><flag contact="disable"/></option>
    <worldbody><body name="a" pos="0 0 1"><joint name="j1" axis="0 1 0" damping="0.01"/><geom type="capsule" size="0.02 0.2" pos="0 0 -0.2"/>
      <body name="b" pos="0 0 -0.4"><joint name="j2" axis="0 1 0"/><geom type="capsule" size="0.02 0.15" pos="0 0 -0.15"/></body>
    </body></worldbody></mujoco>"""
    sols = []
    for h, n in ((0.02, 25), (0.01, 50), (0.00125, 400)):
        s = OracleSim(mujoco_amd.load_model_from_xml(xml.replace("{H}", repr(h))))
        s.qpos[:] = [1.0, 0.5]
        s.step(n)
        sols.append(s.qpos.copy())
    e1 = np.abs(sols[0] - sols[2]).max()
    e2 = np.abs(sols[1] - sols[2]).max()
    assert e2 < e1 / 12 and e1 > 1e-9


def test_committed_trajectory_fixture(cf_model):
    g = np.load(os.path.join(GOLDEN, "dexterous_oracle_traj.npz"))
    s = OracleSim(cf_model)
    for k in range(10):
        s.step(100)
        assert np.abs(s.qpos - g["qpos"][k]).max() < 1e-12
        assert np.abs(s.qvel - g["qvel"][k]).max() < 1e-11


def test_config0_as_shipped_fixture(hand_model):
    """BASELINE.json configs[0] as the reference ships it (output/DexterousDynamos.xml:4: every mesh geom
    contype = conaffinity = 1, so self-collision is ON): default state, zero ctrl, 1000 steps.  The oracle
    reproduces its committed trajectory; the structural facts of SURVEY.md section 8d config 1 hold: nine
    limit rows active at step 0, four equalities, 21 frictionloss rows."""
    g = np.load(os.path.join(GOLDEN, "dexterous_default_contacts_traj.npz"))
    assert hand_model.npair == 210 and not (hand_model.opt_disableflags & 16)
    s = OracleSim(hand_model)
    rows = {int(r[0]): r for r in g["rows"]}
    for k in range(1000):
        s.step()
        if k + 1 in rows:
            d = s._d.contents
            assert [d.nefc, d.ne, d.nf, d.nl, d.ncon] == [int(x) for x in rows[k + 1][1:6]]
            got = sorted((c["geom1"], c["geom2"], c["dist"]) for c in s.contacts())
            want = g[f"contacts_step{k + 1}"]
            assert [(a, b) for a, b, _ in got] == [(int(a), int(b)) for a, b, _ in want]
            assert np.abs(np.array([x[2] for x in got]) - want[:, 2]).max() < 1e-12
        if (k + 1) % 100 == 0:
            assert np.abs(s.qpos - g["qpos"][k // 100]).max() < 1e-12
            assert np.abs(s.qvel - g["qvel"][k // 100]).max() < 1e-11
    r1 = rows[1]
    assert (int(r1[2]), int(r1[3]), int(r1[4])) == (4, 21, 9)  # equalities, frictionloss, the nine limits at qpos = 0
    assert int(r1[1]) == 4 + 21 + 9 + 4 * int(r1[5])


def test_nan_guard_resets(cf_model):
    s = OracleSim(cf_model)
    s.qpos[2] = np.nan
    s.step()
    assert s.warning & 1 and np.isfinite(s.qpos).all()


def test_pgs_newton_same_fixed_point(hand_model):
    """Newton (primal) and PGS (dual, explicit A+R) minimise the same convex problem."""
    for contacts in (False, True):
        m = hand_model.copy() if contacts else workload.contact_free(hand_model)
        mp = m.copy()
        mp.opt_solver = 0
        mp.opt_iterations = 1000
        mp.opt_tolerance = 1e-12
        q, v = workload.config2_state(m, 3, seed=1235)
        for e in range(3):
            a, b = OracleSim(m), OracleSim(mp)
            for s in (a, b):
                s.qpos[:] = q[e]
                s.forward()
            assert np.abs(a.qacc - b.qacc).max() < 1e-5 * max(1.0, np.abs(a.qacc).max())


def test_cg_matches_newton(hand_model):
    """mj_solCG restatement: same minimiser as Newton on the same convex cost (tight tolerance),
    with and without contact rows; more iterations than Newton, as expected of a first-order method."""
    from mujoco_amd import workload
    from mujoco_amd.model import SOL_CG, SOL_NEWTON
    from oracle import OracleSim
    q, v = workload.config2_state(hand_model, 4, seed=5)
    c = workload.config2_ctrl(hand_model, 4, seed=5)
    for contacts in (False, True):
        base = hand_model if contacts else workload.contact_free(hand_model)
        out = {}
        for sol in (SOL_NEWTON, SOL_CG):
            m = base.copy()
            m.opt_solver = sol
            m.opt_tolerance = 1e-12
            m.opt_iterations = 500
            s = OracleSim(m)
            s.qpos[:] = q[1]
            s.ctrl[:] = c[1]
            s.forward()
            out[sol] = (np.array(s.qacc).copy(), s.solver_iter)
        err = np.abs(out[SOL_CG][0] - out[SOL_NEWTON][0]).max() / np.abs(out[SOL_NEWTON][0]).max()
        assert err < 1e-7, (contacts, err)
        assert out[SOL_CG][1] >= out[SOL_NEWTON][1]


def test_contact_rows_and_dual_qp_third_opinion(hand_model):
    """With contacts (self-collision + the ground plane): (a) the oracle's pyramid rows
    J_n +- mu J_t are rebuilt from an independent NumPy point Jacobian (host_jac) of the two bodies at
    the contact point and the contact frame; (b) the Newton solution equals the solution of the DUAL
    problem  min_f  1/2 f^T (J M^-1 J^T + R) f + f^T (J a_smooth - aref)  over the row-wise bounds
    (equality: free, frictionloss: |f| <= loss, limit / pyramid rows: f >= 0), solved by SciPy's
    bounded L-BFGS from zero — a formulation that shares nothing with the oracle's primal Newton."""
    from scipy.optimize import minimize
    m = workload.with_ground_plane(hand_model, z=-0.05)
    checked = 0
    for seed in (1235, 77, 4242, 31):
        q, v = workload.config2_state(m, 6, seed=seed)
        c = workload.config2_ctrl(m, 6, seed=seed)
        for e in range(6):
            s = OracleSim(m)
            s.qpos[:] = q[e]
            s.qvel[:] = 0.2 * np.random.default_rng(seed + e).normal(size=m.nv)
            s.ctrl[:] = c[e]
            s.forward()
            cons = s.contacts()
            if not cons:
                continue
            typ, ids, J, D, aref, f = s.efc("type"), s.efc("id"), s.efc("J"), s.efc("D"), s.efc("aref"), s.efc("force")
            fl = s.efc("frictionloss")
            fk = host_fk(m, s.qpos)
            # (a) contact rows from the independent Jacobian
            rows = np.flatnonzero(typ == 6)
            assert len(rows) == 4 * sum(1 for cc in cons if cc["dist"] < cc["includemargin"])
            r = 0
            for ci, cc in enumerate(cons):
                if not cc["dist"] < cc["includemargin"]:
                    continue
                b1, b2 = int(m.geom_bodyid[cc["geom1"]]), int(m.geom_bodyid[cc["geom2"]])
                j1, _ = host_jac(m, fk, b1, cc["pos"])
                j2, _ = host_jac(m, fk, b2, cc["pos"])
                fr = cc["frame"].reshape(3, 3)
                Jc = fr @ (j2 - j1)
                mu = max(m.geom_friction[cc["geom1"], 0], m.geom_friction[cc["geom2"], 0])
                want = np.stack([Jc[0] + mu * Jc[1], Jc[0] - mu * Jc[1], Jc[0] + mu * Jc[2], Jc[0] - mu * Jc[2]])
                got = J[rows[r:r + 4]]
                assert (ids[rows[r:r + 4]] == ci).all()
                assert np.abs(got - want).max() < 1e-12, (seed, e, ci)
                # all four rows of a contact share D; the frame is orthonormal with x = contact normal
                assert np.ptp(D[rows[r:r + 4]]) == 0 and np.abs(fr @ fr.T - np.eye(3)).max() < 1e-12
                r += 4
            # (b) dual QP
            M = host_mass_matrix(m, fk)
            assert np.abs(M - s.full_m()).max() < 1e-12 * np.abs(M).max()
            Minv = np.linalg.inv(M)
            A = J @ Minv @ J.T + np.diag(1.0 / D)
            b = J @ s.qacc_smooth - aref
            lo = np.where(typ == 0, -np.inf, np.where(typ == 1, -fl, 0.0))
            hi = np.where(typ == 1, fl, np.inf)
            sc = 1.0 / np.sqrt(np.diag(A))  # Jacobi scaling: the rows' regularisers span orders of magnitude
            As, bs = A * sc[:, None] * sc[None, :], b * sc
            res = minimize(lambda x: (0.5 * x @ As @ x + x @ bs, As @ x + bs), np.zeros(len(b)), jac=True, method="L-BFGS-B",
                           bounds=list(zip(lo / sc, hi / sc)), options=dict(maxiter=20000, ftol=1e-16, gtol=1e-13, maxcor=50))
            fd = res.x * sc
            qacc_dual = s.qacc_smooth + Minv @ (J.T @ fd)
            scale = max(1.0, np.abs(s.qacc).max())
            assert np.abs(qacc_dual - s.qacc).max() / scale < 2e-5, (seed, e, np.abs(qacc_dual - s.qacc).max() / scale)
            # the oracle's forces satisfy the dual KKT conditions: residual w = A f + b is >= 0 on rows at
            # their lower bound, <= 0 at the upper bound, ~0 in between (the rows' own scale)
            w = (A @ f + b) * sc
            tol = 1e-6 * max(1.0, np.abs(bs).max())
            at_lo, at_hi = f <= lo + 1e-12, f >= hi - 1e-12
            assert (w[at_lo & ~at_hi] > -tol).all() and (w[at_hi & ~at_lo] < tol).all()
            assert np.abs(w[~at_lo & ~at_hi]).max(initial=0.0) < tol
            checked += 1
    assert checked >= 12
