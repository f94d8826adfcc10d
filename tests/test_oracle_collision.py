"""Oracle narrowphase against brute-force geometry (SURVEY.md §7.4 'GJK/EPA' row): the support
function of a convex hull is an exhaustive arg-max here, independent of the oracle's hill
climbing; separation / penetration are read off the Minkowski-difference support function
sampled over many directions and locally refined."""
import numpy as np
import pytest

import mujoco_amd
from mujoco_amd import meshproc, workload
from oracle import OracleSim


def _box_tris(hx, hy, hz):
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, c, d in quads:
        tris += [[v[a], v[b], v[c]], [v[a], v[c], v[d]]]
    return np.array(tris)


def _two_box_model(tmp_path, offset, margin=0.0):
    meshproc.write_binary_stl(str(tmp_path / "a.stl"), _box_tris(0.5, 0.3, 0.2))
    meshproc.write_binary_stl(str(tmp_path / "b.stl"), _box_tris(0.4, 0.4, 0.1))
    xml = f"""<mujoco><compiler angle="radian"/>
    <default><geom type="mesh" margin="{margin}"/></default>
    <asset><mesh name="a" file="a.stl"/><mesh name="b" file="b.stl"/></asset>
    <worldbody>
      <body name="A" pos="0 0 1"><joint name="ja" type="slide" axis="1 0 0"/><geom mesh="a"/></body>
      <body name="B" pos="{offset[0]} {offset[1]} {1 + offset[2]}"><joint name="jb" type="slide" axis="0 0 1"/><geom mesh="b"/></body>
    </worldbody></mujoco>"""
    return mujoco_amd.load_model_from_xml(xml, base_dir=str(tmp_path))


def test_box_box_known_depth(tmp_path):
    # overlap 0.05 along z: A half-height 0.2, B half-height 0.1, centres 0.25 apart
    m = _two_box_model(tmp_path, (0.05, 0.02, 0.25))
    s = OracleSim(m)
    s.forward()
    assert s.ncon == 1
    c = s.contacts()[0]
    assert abs(c["dist"] + 0.05) < 1e-6
    assert np.abs(np.abs(c["frame"][:3]) - [0, 0, 1]).max() < 1e-6 and c["frame"][2] > 0  # A -> B
    assert 1.15 - 1e-9 <= c["pos"][2] <= 1.2 + 1e-9  # MPR position: inside the overlap slab
    # separated by 0.01 > margin 0: no contact; with margin 0.02: contact with dist = +0.01
    m2 = _two_box_model(tmp_path, (0.0, 0.0, 0.31))
    s2 = OracleSim(m2)
    s2.forward()
    assert s2.ncon == 0
    m3 = _two_box_model(tmp_path, (0.0, 0.0, 0.31), margin=0.02)
    s3 = OracleSim(m3)
    s3.forward()
    assert s3.ncon == 1 and abs(s3.contacts()[0]["dist"] - 0.01) < 1e-6
    assert s3.nefc == 4  # pyramidal condim 3: four rows


def _world_hull(m, s, g):
    me = m.geom_dataid[g]
    v = m.hull_vert[m.mesh_vertadr[me]:m.mesh_vertadr[me] + m.mesh_vertnum[me]]
    return s.geom_xpos[g] + v @ s.geom_xmat[g].reshape(3, 3).T


def _sphere_dirs(n, seed=0):
    r = np.random.default_rng(seed).normal(size=(n, 3))
    return r / np.linalg.norm(r, axis=1, keepdims=True)


def test_hand_contacts_vs_bruteforce_support(hand_model):
    """For every reported contact: depth == h_{A-B}(normal) (the portal lies on the surface of
    the Minkowski difference), no sampled direction separates the inflated hulls; for candidate
    pairs NOT reported, some direction separates them by more than the margin."""
    m = hand_model
    s = OracleSim(m)
    q, _ = workload.config2_state(m, 6, seed=1235)
    dirs = _sphere_dirs(4000)
    margin = 0.0005
    checked_hit = checked_miss = 0
    for e in range(6):
        s.qpos[:] = q[e]
        s.forward()
        hits = {(c["geom1"], c["geom2"]): c for c in s.contacts()}
        hulls = {}
        for p in range(m.npair):
            g1, g2 = int(m.pair_geom1[p]), int(m.pair_geom2[p])
            if np.linalg.norm(s.geom_xpos[g1] - s.geom_xpos[g2]) > margin + m.geom_rbound[g1] + m.geom_rbound[g2]:
                assert (g1, g2) not in hits
                continue
            for g in (g1, g2):
                if g not in hulls:
                    hulls[g] = _world_hull(m, s, g)
            A, B = hulls[g1], hulls[g2]
            # overlap(d) = max_a d.a - min_b d.b  (support of A - B); < -margin somewhere => separated
            ov = (A @ dirs.T).max(axis=0) - (B @ dirs.T).min(axis=0)
            if (g1, g2) in hits:
                c = hits[(g1, g2)]
                n = c["frame"][:3]
                depth_n = (A @ n).max() - (B @ n).min() + margin  # inflated by margin/2 each
                # the portal triangle lies inside the Minkowski difference within mpr_tolerance of its
                # surface; the closest point may sit on a portal edge, so depth <= h(n)
                depth = margin - c["dist"]
                assert 0 <= depth <= depth_n + 2e-6, (g1, g2, depth, depth_n)
                assert ov.min() + margin > -1e-9      # nothing separates the inflated hulls
                checked_hit += 1
            else:
                # refine the best separating direction locally
                best = dirs[np.argmin(ov)]
                loc = best + 0.05 * _sphere_dirs(2000, seed=1)
                loc /= np.linalg.norm(loc, axis=1, keepdims=True)
                ov2 = (A @ loc.T).max(axis=0) - (B @ loc.T).min(axis=0)
                assert min(ov.min(), ov2.min()) < -margin + 2e-5, (g1, g2, ov.min())
                checked_miss += 1
    assert checked_hit > 10 and checked_miss > 100


def test_hill_climb_equals_exhaustive(hand_model):
    """Plane-hull contact uses a single support query: its vertex must be the exhaustive arg-min."""
    m = hand_model
    s = OracleSim(m)
    s.forward()
    rng = np.random.default_rng(3)
    for g in (0, 1, 5, 9, 15):
        W = _world_hull(m, s, g)
        for _ in range(20):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            # exhaustive support
            best = W[np.argmax(W @ d)]
            # oracle support through the plane routine is not exposed; emulate with adjacency walk
            me = m.geom_dataid[g]
            a0 = m.mesh_vertadr[me]
            V = m.hull_vert[a0:a0 + m.mesh_vertnum[me]]
            dl = s.geom_xmat[g].reshape(3, 3).T @ d
            cur, prev = 0, -1
            while cur != prev:
                prev = cur
                nb = m.hull_adj[m.hull_adj_start[a0 + prev]:m.hull_adj_start[a0 + prev + 1]]
                j = nb[np.argmax(V[nb] @ dl)]
                if V[j] @ dl > V[cur] @ dl:
                    cur = int(j)
            assert abs(V[cur] @ dl - (best - s.geom_xpos[g]) @ d) < 1e-12


def test_contact_rows_and_friction_cone(tmp_path):
    """A box resting on another: 4 pyramid rows, forces non-negative, solver removes penetration
    acceleration (normal acceleration >= aref)."""
    m = _two_box_model(tmp_path, (0.0, 0.0, 0.299))
    s = OracleSim(m)
    s.forward()
    assert s.ncon == 1 and s.nefc == 4
    f = s.efc("force")
    assert (f >= 0).all() and f.sum() > 0
    J = s.efc("J")
    # rows are J_n +- mu J_t: sums of opposite pairs are 2 J_n
    assert np.allclose(J[0] + J[1], J[2] + J[3])
    assert s.efc("type").tolist() == [6, 6, 6, 6]
    R = s.efc("R")
    assert np.allclose(R, R[0])


def test_hull_simplification_is_an_inner_approximation(hand_model):
    """SURVEY.md §8 f-3 knob: simplified hulls keep only original hull vertices (so they lie inside
    the original), respect the vertex budget and stay closed convex polytopes (Euler's formula)."""
    from mujoco_amd.meshproc import simplify_model_hulls
    s = simplify_model_hulls(hand_model, 96)
    assert (s.mesh_vertnum <= 96).all() and (s.mesh_vertnum >= 20).all()
    d = np.random.default_rng(1).normal(size=(300, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for i in range(len(s.mesh_vertadr)):
        a, n = hand_model.mesh_vertadr[i], hand_model.mesh_vertnum[i]
        b, k = s.mesh_vertadr[i], s.mesh_vertnum[i]
        full, simp = hand_model.hull_vert[a:a + n], s.hull_vert[b:b + k]
        assert ((simp @ d.T).max(0) <= (full @ d.T).max(0) + 1e-15).all()
        assert all((np.abs(full - v).sum(1) == 0).any() for v in simp)            # subset of the original vertices
        nedge = (s.hull_adj_start[b + k] - s.hull_adj_start[b]) // 2
        assert k - nedge + s.mesh_facenum[i] == 2                                   # V - E + F = 2
    assert np.array_equal(s.body_mass, hand_model.body_mass) and np.array_equal(s.geom_rbound, hand_model.geom_rbound)
