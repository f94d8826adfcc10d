"""Primitive-geom collision (north_star: "plus plane/primitive pairs"; verdict row N1): closed-form
known answers for the oracle on the CPU, and device vs oracle on a model that mixes every supported
pair kind.  The exporter itself only emits mesh geoms (reference src/Mujoco_XML.py:70); primitives
enter when a user edits the geom default, so these models are hand-written MJCF of the same dialect.
PARITY UNPINNED vs MuJoCo 2.1.0 (its native box-box / capsule-box routines return several contacts
per pair [EXT]; this build returns one per convex pair and says so in DESIGN.md)."""
import numpy as np
import pytest

import mujoco_amd
from oracle import OracleSim


def _xml(bodies, extra=""):
    return f"""<mujoco><compiler angle="radian"/><option gravity="0 0 -9.81"/>
    <default><geom margin="0.001" condim="3" friction="1 0.005 0.0001"/><joint damping="0.05" armature="0.01"/></default>
    <worldbody><geom name="floor" type="plane" size="0 0 1" pos="0 0 0"/>{extra}{bodies}</worldbody></mujoco>"""


def _slider(name, geom, x=0.0, z=1.0):
    return (f'<body name="{name}" pos="{x} 0 {z}"><joint name="{name}_z" type="slide" axis="0 0 1"/>'
            f'<geom name="{name}_g" {geom}/></body>')


def _contacts(s):
    return [(c["geom1"], c["geom2"], c["dist"], c["pos"], c["frame"][:3]) for c in s.contacts()]


def test_plane_sphere_capsule_box_closed_forms():
    m = mujoco_amd.load_model_from_xml(_xml(
        _slider("ball", 'type="sphere" size="0.1"', x=0.0)
        + _slider("cap", 'type="capsule" size="0.05 0.2" quat="0.7071067811865476 0 0.7071067811865476 0"', x=1.0)
        + _slider("box", 'type="box" size="0.1 0.2 0.3"', x=2.0)))
    s = OracleSim(m)
    s.qpos[:] = [-0.905, -0.9495, -0.7005]   # ball centre 0.095, capsule axis height 0.0505, box centre 0.2995
    s.forward()
    con = _contacts(s)
    ball = [c for c in con if c[1] == m.geom_name2id("ball_g")]
    assert len(ball) == 1 and abs(ball[0][2] - (-0.005)) < 1e-15
    assert np.allclose(ball[0][3], [0, 0, -0.0025], atol=1e-15) and np.allclose(ball[0][4], [0, 0, 1])
    cap = [c for c in con if c[1] == m.geom_name2id("cap_g")]
    assert len(cap) == 2  # the capsule lies on its side: both caps touch
    assert all(abs(c[2] - 0.0005) < 1e-12 for c in cap)
    assert sorted(round(c[3][0], 6) for c in cap) == [0.8, 1.2]
    box = [c for c in con if c[1] == m.geom_name2id("box_g")]
    assert len(box) == 4 and all(abs(c[2] - (-0.0005)) < 1e-15 for c in box)
    assert sorted((round(c[3][0] - 2.0, 6), round(c[3][1], 6)) for c in box) == [(-0.1, -0.2), (-0.1, 0.2), (0.1, -0.2), (0.1, 0.2)]
    assert s.nefc == 4 * 7  # 7 contacts x 4 pyramid rows (no frictionloss, no limits in this model)
    # above the margin: nothing
    s.qpos[:] = [-0.898, -0.94, -0.69]
    s.forward()
    assert s.ncon == 0


def test_sphere_sphere_capsule_closed_forms():
    m = mujoco_amd.load_model_from_xml(_xml(
        _slider("a", 'type="sphere" size="0.1"', x=0.0, z=1.0) + _slider("b", 'type="sphere" size="0.2"', x=0.25, z=1.0)
        + _slider("c", 'type="capsule" size="0.05 0.3"', x=0.0, z=2.0) + _slider("d", 'type="sphere" size="0.1"', x=0.14, z=2.1)
        + _slider("e", 'type="capsule" size="0.05 0.3" quat="0.7071067811865476 0.7071067811865476 0 0"', x=0.0, z=3.0)
        + _slider("f", 'type="capsule" size="0.04 0.3" quat="0.7071067811865476 0 0.7071067811865476 0"', x=0.0, z=3.08)))
    s = OracleSim(m)
    s.forward()
    con = {(c[0], c[1]): c for c in _contacts(s)}
    gid = m.geom_name2id
    c = con[(gid("a_g"), gid("b_g"))]
    assert abs(c[2] - (0.25 - 0.3)) < 1e-15 and np.allclose(c[4], [1, 0, 0]) and np.allclose(c[3], [0.075, 0, 1.0])
    # sphere d vs capsule c (g1 is the sphere: lower geom type first): closest point of the axis is at height 2.1
    c = con[(gid("d_g"), gid("c_g"))]
    assert abs(c[2] - (0.14 - 0.15)) < 1e-15 and np.allclose(c[4], [-1, 0, 0])
    # crossed capsules: axes along y and x, 0.08 apart
    c = con[(gid("e_g"), gid("f_g"))]
    assert abs(c[2] - (0.08 - 0.09)) < 1e-15 and np.allclose(c[4], [0, 0, 1]) and np.allclose(c[3], [0, 0, 3.0 + 0.05 - 0.005])


_ALONG_X = 'quat="0.7071067811865476 0 0.7071067811865476 0"'


def _parallel_model():
    """Two capsules along x, one resting 0.08 above the other and shifted by 0.15, and a box whose margin (0.5)
    reaches above its upper corners: the two cases where MuJoCo's native routines count contacts differently from
    'one per closest point' / 'one per corner below the margin' (mjc_CapsuleCapsule, mjc_PlaneBox [EXT])."""
    return mujoco_amd.load_model_from_xml(_xml(
        _slider("lo", f'type="capsule" size="0.05 0.3" {_ALONG_X}', x=0.0, z=3.0)
        + _slider("hi", f'type="capsule" size="0.04 0.2" {_ALONG_X}', x=0.15, z=3.08)
        + _slider("box", 'type="box" size="0.1 0.2 0.3" margin="0.5"', x=2.0, z=0.35)))


def test_parallel_capsules_two_contacts_and_plane_box_four():
    m = _parallel_model()
    gid = m.geom_name2id
    s = OracleSim(m)
    s.forward()
    con = _contacts(s)
    # segment 1 spans x in [-0.3, 0.3], segment 2 [-0.05, 0.35]: its +end (x = 0.3) and segment 2's -end (x = -0.05)
    # are 0.08 apart from the other axis (-0.01 after the radii); the other two ends are out of reach
    cc = [c for c in con if (c[0], c[1]) == (gid("lo_g"), gid("hi_g"))]
    assert len(cc) == 2
    assert [round(c[3][0], 9) for c in cc] == [0.3, -0.05]           # MuJoCo's order: ends of capsule 1, then of capsule 2
    for c in cc:
        assert abs(c[2] - (0.08 - 0.09)) < 1e-12 and np.allclose(c[4], [0, 0, 1], atol=1e-12)
        assert abs(c[3][2] - (3.0 + 0.05 - 0.005)) < 1e-12
    # the box: all 8 corners are below the margin (0.05 and 0.65 < 0.5 + ... no: 0.65 > 0.5), so take a closer look:
    # lower corners at 0.05 (contacts), upper corners at 0.65 (above the margin AND on the far side of the centre)
    box = [c for c in con if c[1] == gid("box_g")]
    assert len(box) == 4 and all(abs(c[2] - 0.05) < 1e-12 for c in box)
    assert [(round(c[3][0] - 2.0, 9), round(c[3][1], 9)) for c in box] == [(-0.1, -0.2), (0.1, -0.2), (-0.1, 0.2), (0.1, 0.2)]
    # lowered until the upper corners are below the margin too: still the four on the plane's side of the centre
    s.qpos[2] = -0.2   # centre 0.15: lower corners at -0.15, upper at 0.45 < 0.5
    s.forward()
    box = [c for c in _contacts(s) if c[1] == gid("box_g")]
    assert len(box) == 4 and all(abs(c[2] - (-0.15)) < 1e-12 for c in box)
    # capsules a hair off parallel take the general branch: one contact, between the closest points
    c, cz, sz = float(np.sqrt(0.5)), float(np.cos(0.005)), float(np.sin(0.005))     # yaw of 0.01 rad after the pitch that lays the axis along x
    s2 = OracleSim(mujoco_amd.load_model_from_xml(_xml(
        _slider("lo", f'type="capsule" size="0.05 0.3" {_ALONG_X}', x=0.0, z=3.0)
        + _slider("hi", f'type="capsule" size="0.04 0.2" quat="{cz * c!r} {-sz * c!r} {cz * c!r} {sz * c!r}"', x=0.15, z=3.08))))
    s2.forward()
    assert s2.ncon == 1 and abs(s2.contacts()[0]["dist"] - (-0.01)) < 1e-12


def test_parallel_capsule_rests_on_both_contacts():
    """A capsule dropped on a parallel fixed one stays level: with a single contact at a segment end it would tip."""
    m = mujoco_amd.load_model_from_xml(_xml(
        '<body name="lift" pos="0.1 0 0.181"><joint name="lift_z" type="slide" axis="0 0 1"/>'
        '<geom name="lift_g" type="sphere" size="0.01" contype="0" conaffinity="0"/>'
        '<body name="hi" pos="0 0 0"><joint name="pitch" type="hinge" axis="0 1 0"/>'
        f'<geom name="hi_g" type="capsule" size="0.04 0.2" {_ALONG_X}/></body></body>',
        extra=f'<geom name="lo_g" type="capsule" size="0.05 0.4" pos="0 0 0.05" {_ALONG_X}/>'))
    s = OracleSim(m)
    s.step(300)
    cc = [c for c in s.contacts() if c["geom2"] == m.geom_name2id("hi_g")]
    assert len(cc) == 2 and s.warning == 0
    assert abs(s.qpos[0] - (0.14 - 0.181)) < 2e-3   # resting on top: 0.05 + 0.05 + 0.04
    assert abs(s.qpos[1]) < 1e-3                     # level


def test_box_box_and_mixed_convex_pairs_by_mpr():
    """Convex pairs without a closed form go through MPR with the primitives' support functions:
    face-to-face boxes (depth known), sphere in a box corner region, capsule on a box, cylinder, ellipsoid."""
    m = mujoco_amd.load_model_from_xml(_xml(
        _slider("a", 'type="box" size="0.1 0.1 0.1"', x=0.0, z=1.0) + _slider("b", 'type="box" size="0.2 0.15 0.1"', x=0.05, z=1.19)
        + _slider("c", 'type="box" size="0.1 0.1 0.1"', x=1.0, z=1.0) + _slider("d", 'type="sphere" size="0.05"', x=1.0, z=1.145)
        + _slider("e", 'type="cylinder" size="0.1 0.1"', x=2.0, z=1.0) + _slider("f", 'type="ellipsoid" size="0.1 0.05 0.02"', x=2.0, z=1.115)))
    s = OracleSim(m)
    s.forward()
    con = {(c[0], c[1]): c for c in _contacts(s)}
    gid = m.geom_name2id
    c = con[(gid("a_g"), gid("b_g"))]
    assert abs(c[2] - (-0.01)) < 1e-6 and np.allclose(c[4], [0, 0, 1], atol=1e-6) and abs(c[3][2] - 1.095) < 1e-6
    c = con[(gid("d_g"), gid("c_g"))]          # sphere (type 2) before box (type 6)
    assert abs(c[2] - (-0.005)) < 1e-6 and np.allclose(c[4], [0, 0, -1], atol=1e-6)
    c = con[(gid("f_g"), gid("e_g"))]          # ellipsoid (4) before cylinder (5)
    assert abs(c[2] - (-0.005)) < 1e-6 and np.allclose(c[4], [0, 0, -1], atol=1e-5)


def _mixed_model():
    """Pendulum arms and sliders carrying every primitive type above a floor, with a mesh-free tree the
    tree-parallel kernels accept (one joint per body)."""
    arm = ('<body name="arm" pos="0 0 0.8"><joint name="sh" type="hinge" axis="0 1 0"/>'
           '<geom name="arm_g" type="capsule" size="0.04 0.2" pos="0 0 -0.2"/>'
           '<body name="fore" pos="0 0 -0.4"><joint name="el" type="hinge" axis="0 1 0" limited="true" range="-2 2"/>'
           '<geom name="fore_g" type="box" size="0.03 0.05 0.15" pos="0 0 -0.15"/>'
           '<body name="tip" pos="0 0 -0.3"><joint name="wr" type="hinge" axis="1 0 0"/>'
           '<geom name="tip_g" type="sphere" size="0.06" pos="0 0 -0.05"/></body></body></body>')
    def nested(name, geom, x, z, child=""):
        return (f'<body name="{name}" pos="{x} 0 {z}"><joint name="{name}_z" type="slide" axis="0 0 1"/>'
                f'<geom name="{name}_g" {geom}/>{child}</body>')

    # four kinematic trees (the tree-parallel kernels take at most four): the arm, cube -> egg, pill -> ball2, can;
    # a child slides relative to its parent, and parent-child geom pairs are filtered as MuJoCo filters them
    return mujoco_amd.load_model_from_xml(_xml(
        arm
        + nested("cube", 'type="box" size="0.1 0.12 0.08" quat="0.9238795 0.2209424 0.2209424 0.2209424"', 0.35, 0.12,
                 nested("egg", 'type="ellipsoid" size="0.09 0.06 0.05"', -0.2, 0.18))
        + nested("pill", 'type="capsule" size="0.05 0.15" quat="0.7071067811865476 0 0.7071067811865476 0"', -0.3, 0.06,
                 nested("ball2", 'type="sphere" size="0.08"', 0.0, 0.14))
        + _slider("can", 'type="cylinder" size="0.07 0.1"', x=0.0, z=0.105)))


def test_oracle_mixed_primitive_model_steps():
    m = _mixed_model()
    assert m.npair > 20
    s = OracleSim(m)
    s.step(300)
    assert np.isfinite(s.qpos).all() and s.ncon >= 5 and s.warning == 0
    # everything came to rest on the floor / on each other, nothing tunnelled below it
    assert min(float(s.geom_xpos[g][2]) for g in range(1, m.ngeom)) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["tp", "serial"])
def test_primitive_pairs_device_vs_oracle(kernel, monkeypatch):
    """Device (both kernel families, PRIM narrowphase instantiation) vs oracle on the mixed model:
    contact lists equal INCLUDING multiplicity and order (plane-capsule 2, plane-box up to 8 per pair),
    dist 1e-9, qacc 1e-8 after one step, trajectories after 50."""
    import torch
    from mujoco_amd import BatchSim
    monkeypatch.setenv("BS_KERNEL", kernel)
    m = _mixed_model()
    E = 70
    rng = np.random.default_rng(3)
    q = np.zeros((E, m.nq))
    q[:, :3] = rng.uniform(-1.2, 1.2, size=(E, 3))
    q[:, 3:] = rng.uniform(-0.06, 0.25, size=(E, m.nq - 3))
    sim = BatchSim(m, n_envs=E)
    monkeypatch.delenv("BS_KERNEL")
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.forward(); sim.sync()
    ncon = sim.data.ncon.cpu().numpy()
    qacc = sim.data.qacc.cpu().numpy()
    assert ncon.sum() > 3 * E and int((sim.data.warning != 0).sum()) == 0
    kinds, strict_envs = set(), 0
    curved = {2, 3, 4, 5}

    def by_mpr(t1, t2):  # pairs without a closed form
        return t1 != 0 and not (t1 == 2 and t2 in (2, 3)) and not (t1 == 3 and t2 == 3)

    for e in range(0, E, 3):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.forward()
        want = [(c["geom1"], c["geom2"]) for c in s.contacts()]
        got = sim.data.contact_list(e)
        assert [(c.geom1, c.geom2) for c in got] == want, e
        loose_env = False
        for a, b in zip(got, s.contacts()):
            t1, t2 = int(m.geom_type[a.geom1]), int(m.geom_type[a.geom2])
            kinds.add((t1, t2))
            # MPR on a curved primitive stops at mpr_tolerance (1e-6) along a path that depends on the last
            # bits of the arithmetic (the device contracts to FMA, the oracle does not): such contacts agree
            # to the tolerance of the algorithm, everything else (closed forms, polyhedra) to rounding
            loose = by_mpr(t1, t2) and (t1 in curved or t2 in curved)
            loose_env |= loose
            # (the normal of a curved contact found to 1e-6 in depth is only defined to sqrt(2e-6 / radius) ~ 1e-2 rad)
            td, tp, tf = (5e-5, 5e-4, 2e-2) if loose else (1e-9, 1e-8, 1e-7)
            assert abs(a.dist - b["dist"]) < td and np.abs(a.pos - b["pos"]).max() < tp, (e, t1, t2)
            assert np.abs(a.frame - b["frame"]).max() < tf, (e, t1, t2)
        assert ncon[e] == s.ncon
        err = np.abs(qacc[e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max())
        assert err < (2e-2 if loose_env else 1e-8), (e, err)
        strict_envs += not loose_env
    assert len(kinds) >= 8, kinds   # plane-x, sphere-x, capsule-x, box-x ... really exercised
    assert strict_envs >= 5
    sim.step(50); sim.sync()
    qg = sim.data.qpos.cpu().numpy()
    assert np.isfinite(qg).all() and int(((sim.data.warning & 7) != 0).sum()) == 0
    for e in (0, 33, 69):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.step(50)
        assert np.abs(qg[e] - s.qpos).max() < 5e-3, e


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["tp", "serial"])
def test_parallel_capsules_and_margin_box_device_vs_oracle(kernel, monkeypatch):
    """The two contact-count rules taken from MuJoCo's native routines (parallel capsule-capsule: up to 2; plane-box:
    the corners on the plane's side of the centre, at most 4), device vs oracle: same lists in the same order."""
    import torch
    from mujoco_amd import BatchSim
    monkeypatch.setenv("BS_KERNEL", kernel)
    m = _parallel_model()
    E = 48
    rng = np.random.default_rng(11)
    q = np.zeros((E, m.nq))
    q[:, 0] = rng.uniform(-0.02, 0.02, size=E)
    q[:, 1] = rng.uniform(-0.02, 0.02, size=E)
    q[:, 2] = rng.uniform(-0.3, 0.3, size=E)
    sim = BatchSim(m, n_envs=E)
    monkeypatch.delenv("BS_KERNEL")
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.forward(); sim.sync()
    assert int((sim.data.warning != 0).sum()) == 0
    qacc = sim.data.qacc.cpu().numpy()
    two = four = 0
    for e in range(E):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.forward()
        got = sim.data.contact_list(e)
        assert [(c.geom1, c.geom2) for c in got] == [(c["geom1"], c["geom2"]) for c in s.contacts()], e
        for a, b in zip(got, s.contacts()):
            assert abs(a.dist - b["dist"]) < 1e-12 and np.abs(a.pos - b["pos"]).max() < 1e-12, e
            assert np.abs(a.frame - b["frame"]).max() < 1e-12, e
        two += sum(1 for c in got if m.geom_type[c.geom1] == 3) == 2
        four += sum(1 for c in got if m.geom_type[c.geom2] == 6) == 4
        assert np.abs(qacc[e] - s.qacc).max() / max(1.0, np.abs(s.qacc).max()) < 1e-8, e
    assert two >= 10 and four >= 10
    sim.step(100); sim.sync()
    qg = sim.data.qpos.cpu().numpy()
    for e in (0, 17, 47):
        s = OracleSim(m)
        s.qpos[:] = q[e]
        s.step(100)
        assert np.abs(qg[e] - s.qpos).max() < 1e-7, e
