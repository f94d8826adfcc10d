"""Host-side model compiler: structural known-answer tests pinned by
the reference's own data file output/DexterousDynamos.xml (SURVEY.md §8c 'golden vectors'),
plus analytic checks of the mesh pipeline.  Dynamics numbers are NOT pinned by the reference."""
import os

import numpy as np
import pytest

import mujoco_amd
from mujoco_amd import meshproc
from mujoco_amd.mathutil import quat_mul, quat_to_mat
from mujoco_amd.model import host_fk
from conftest import REF_XML


def test_counts_from_committed_fixture(hand_model):
    m = hand_model
    # SURVEY.md §0.5 / BASELINE.md §2: 24 / 21 / 21 / 17 / 22 / 4 / 22 excludes, nM 71, 210 pairs
    assert (m.nbody, m.nq, m.nv, m.nu, m.ngeom, m.neq) == (24, 21, 21, 17, 22, 4)
    assert m.nM == 71 and m.npair == 210 and len(m.exclude_pair) == 22
    # ctrl -> joint map is the <actuator> document order (xml:119-135)
    assert m.actuator_trnid.tolist() == [0, 17, 18, 19, 13, 14, 15, 16, 9, 10, 11, 5, 6, 7, 1, 2, 3]
    # equalities (xml:162-165): (j20<-j19), (j12<-j11), (j8<-j7), (j4<-j3), coefficient 70/120
    assert list(zip(m.eq_obj1id.tolist(), m.eq_obj2id.tolist())) == [(20, 19), (12, 11), (8, 7), (4, 3)]
    assert np.allclose(m.eq_data[:, 1], 70 / 120) and np.all(m.eq_data[:, [0, 2, 3, 4]] == 0)
    # joint ranges == ctrl ranges for actuated joints (xml:35-111 vs :119-135)
    assert np.array_equal(m.jnt_range[m.actuator_trnid], m.actuator_ctrlrange)
    # defaults of reference src/Mujoco_XML.py:70-72
    assert np.all(m.dof_damping == 0.1) and np.all(m.dof_armature == 0.001)
    assert np.all(m.dof_frictionloss == 0.001) and np.all(m.jnt_margin == 0.01)
    assert np.all(m.geom_margin == 0.0005) and np.all(m.geom_condim == 3)
    assert np.allclose(m.geom_friction, [1, 0.005, 0.001])
    assert np.all(m.actuator_gainprm[:, 0] == 2.0) and np.all(m.actuator_biasprm[:, 1] == -2.0)
    assert np.all(m.actuator_forcerange == [-1, 1])
    # hull sizes computed in the survey
    assert sorted(m.mesh_vertnum.tolist()) == sorted([220, 340, 1342, 1179, 1086, 668, 1277, 1033, 895, 1493, 993, 714])
    # tree shape: dof 0 ancestor of all, fingers are 4-chains
    assert m.dof_parentid[0] == -1 and all(m.dof_parentid[1 + 4 * f] == 0 for f in range(5))
    assert m.opt_timestep == 0.002 and m.opt_integrator == 0 and m.opt_solver == 2


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference tree not present")
def test_compile_reference_xml_matches_fixture(hand_model):
    m = mujoco_amd.load_model_from_path(REF_XML)
    for f in ("body_pos", "body_quat", "body_mass", "body_inertia", "body_ipos", "jnt_axis", "jnt_range",
              "hull_vert", "dof_invweight0", "body_invweight0"):
        assert np.allclose(getattr(m, f), getattr(hand_model, f), rtol=1e-12, atol=0), f
    assert m.body_names == hand_model.body_names


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference tree not present")
def test_zero_pose_is_xml_composition(hand_model):
    """KAT (1) of SURVEY.md §8c: at qpos=0 world poses = composition of the pos/quat attributes
    down the XML nesting (parsed here independently with ElementTree)."""
    import xml.etree.ElementTree as ET
    root = ET.parse(REF_XML).getroot()
    poses = {}

    def walk(el, P, Q):
        for b in el.findall("body"):
            p = np.array([float(x) for x in b.get("pos", "0 0 0").split()])
            q = np.array([float(x) for x in b.get("quat", "1 0 0 0").split()])
            q = q / np.linalg.norm(q)
            P2 = P + quat_to_mat(Q) @ p
            Q2 = quat_mul(Q, q)
            poses[b.get("name")] = (P2, Q2 / np.linalg.norm(Q2))
            walk(b, P2, Q2)

    walk(root.find("worldbody"), np.zeros(3), np.array([1.0, 0, 0, 0]))
    fk = host_fk(hand_model, np.zeros(hand_model.nq))
    for i, name in enumerate(hand_model.body_names[1:], start=1):
        P, Q = poses[name]
        assert np.abs(fk["xpos"][i] - P).max() < 1e-15
        assert min(np.abs(fk["xquat"][i] - Q).max(), np.abs(fk["xquat"][i] + Q).max()) < 1e-14


def _box_tris(hx, hy, hz, offset=(0, 0, 0)):
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float)
    v += np.asarray(offset, float)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, c, d in quads:
        tris += [[v[a], v[b], v[c]], [v[a], v[c], v[d]]]
    return np.array(tris)


@pytest.mark.parametrize("mode", ["legacy", "exact"])
def test_box_mesh_inertia_analytic(mode):
    hx, hy, hz = 0.3, 0.2, 0.1
    vol, com, I = meshproc.mesh_inertia(_box_tris(hx, hy, hz, offset=(1, 2, 3)), mode)
    assert abs(vol - 8 * hx * hy * hz) < 1e-14
    assert np.abs(com - [1, 2, 3]).max() < 1e-13
    want = vol / 3 * np.diag([hy**2 + hz**2, hx**2 + hz**2, hx**2 + hy**2])
    assert np.abs(I - want).max() < 1e-14


def test_stl_roundtrip_hull_and_support(tmp_path):
    rng = np.random.default_rng(0)
    tris = _box_tris(1, 2, 3)
    p = tmp_path / "box.stl"
    meshproc.write_binary_stl(str(p), tris)
    back = meshproc.read_stl(str(p))
    assert np.array_equal(back, tris.astype(np.float32).astype(np.float64))
    cm = meshproc.load_mesh("box", str(p), [0.001] * 3)
    assert len(cm.hull_vert) == 8 and abs(cm.volume - 48e-9) < 1e-18
    # principal moments sorted descending, frame right handed
    assert np.all(np.diff(cm.inertia_unit) <= 0)
    # hill climbing over the adjacency graph reaches the exhaustive support vertex
    pts = rng.normal(size=(400, 3))
    hv, faces, start, adj = meshproc.convex_hull(pts)
    for _ in range(50):
        d = rng.normal(size=3)
        best = int(np.argmax(hv @ d))
        cur = 0
        while True:
            nb = adj[start[cur]:start[cur + 1]]
            j = nb[np.argmax(hv[nb] @ d)]
            if hv[j] @ d > hv[cur] @ d:
                cur = int(j)
            else:
                break
        assert abs(hv[cur] @ d - hv[best] @ d) < 1e-12


def test_synthetic_mjcf_dialect(tmp_path):
    """Same dialect as the exporter (reference src/Mujoco_XML.py:37-75), built from scratch:
    joint/geom after child bodies, ragged equality indentation, default classes, <option>."""
    meshproc.write_binary_stl(str(tmp_path / "link.stl"), _box_tris(5, 5, 20) )
    xml = f"""<mujoco model="two_link">
    <compiler angle="radian"/>
    <option timestep="0.001" integrator="RK4"><flag contact="disable"/></option>
    <default>
      <geom type="mesh" friction="1 0.005 0.001" margin="0.0005"/>
      <joint type="hinge" limited="true" damping="0.1" armature="0.001" margin="0.01" frictionloss="0.001"/>
      <position ctrllimited="true" forcelimited="true" forcerange="-1 1" kp="2.0"/>
      <mesh scale="0.001 0.001 0.001"/>
    </default>
    <asset><mesh name="link" file="link.stl"/></asset>
    <worldbody><body name="root" quat="1 0 0 0">
      <body name="a" pos="0 0 0.1" quat="1 0 0 0">
        <body name="b" pos="0 0 0.04">
          <joint name="b_joint" pos="0 0 -0.02" axis="0 1 0" range="0 1.3"/>
          <geom mesh="link"/>
        </body>
        <joint name="a_joint" pos="0 0 -0.02" axis="1 0 0" range="-1 1"/>
        <geom mesh="link"/>
      </body></body></worldbody>
    <actuator><position name="a_act" joint="a_joint" ctrlrange="-1 1"/></actuator>
    <contact><exclude body1="a" body2="b"/></contact>
    <equality>
            <joint joint1="b_joint" joint2="a_joint" polycoef="0 0.5 0 0 0"/>
    </equality></mujoco>"""
    p = tmp_path / "m.xml"
    p.write_text(xml)
    m = mujoco_amd.load_model_from_path(str(p))
    assert (m.nbody, m.nv, m.nu, m.neq, m.ngeom) == (4, 2, 1, 1, 2)
    assert m.joint_names == ["a_joint", "b_joint"] and m.dof_parentid.tolist() == [-1, 0]
    assert m.opt_timestep == 0.001 and m.opt_integrator == 1
    assert m.opt_disableflags & mujoco_amd.model.DSBL_CONTACT
    assert m.npair == 0  # only pair is parent-child and excluded
    mass = 1000 * 0.01 * 0.01 * 0.04
    assert np.allclose(m.body_mass[2:], mass)
    # npz round trip
    m.save(str(tmp_path / "m.npz"))
    m2 = mujoco_amd.Model.load(str(tmp_path / "m.npz"))
    assert np.array_equal(m2.body_inertia, m.body_inertia) and m2.joint_names == m.joint_names
    assert mujoco_amd.load_model_from_path(str(tmp_path / "m.npz")).nv == 2


def test_compiler_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        mujoco_amd.load_model_from_path(str(tmp_path / "nope.xml"))
    with pytest.raises(ValueError):
        mujoco_amd.load_model_from_xml("<notmujoco/>")
    # massless moving body -> MuJoCo's compile error
    with pytest.raises(ValueError):
        mujoco_amd.load_model_from_xml(
            "<mujoco><worldbody><body name='a'><joint name='j'/></body></worldbody></mujoco>")
    # geom referencing an undeclared mesh (what the reference exporter would emit for the
    # stripped Carpals.stl)
    with pytest.raises(ValueError):
        mujoco_amd.load_model_from_xml(
            "<mujoco><worldbody><body name='a'><geom type='mesh' mesh='Carpals'/></body></worldbody></mujoco>")
    with pytest.raises(FileNotFoundError):
        mujoco_amd.load_model_from_xml(
            "<mujoco><asset><mesh name='x' file='missing.stl'/></asset><worldbody><body name='a'>"
            "<geom type='mesh' mesh='x'/></body></worldbody></mujoco>", base_dir=str(tmp_path))
