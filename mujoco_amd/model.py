"""Compiled model container (the build's `mjModel` counterpart, SURVEY.md §8a rows
A0/A15): flat NumPy arrays named after mjModel fields, plus the compile-time
constants MuJoCo derives at qpos0 (`mj_setConst` [EXT]: dof_invweight0,
body_invweight0, stat.meaninertia).

Serialises to a single .npz so a compiled model can travel without its STL
files (the GPU box has no /root/reference).
"""
import json

import numpy as np

from .mathutil import axis_angle_quat, quat_mul, quat_normalize, quat_to_mat

MINVAL = 1e-15

# enums (values follow mjtGeom / mjtJoint / mjtIntegrator / mjtSolver [EXT])
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = range(4)
INT_EULER, INT_RK4 = 0, 1
SOL_PGS, SOL_CG, SOL_NEWTON = 0, 1, 2
DSBL_CONSTRAINT = 1 << 0
DSBL_EQUALITY = 1 << 1
DSBL_FRICTIONLOSS = 1 << 2
DSBL_LIMIT = 1 << 3
DSBL_CONTACT = 1 << 4
DSBL_PASSIVE = 1 << 5
DSBL_GRAVITY = 1 << 6
DSBL_CLAMPCTRL = 1 << 7
DSBL_WARMSTART = 1 << 8
DSBL_FILTERPARENT = 1 << 9
DSBL_ACTUATION = 1 << 10
DSBL_REFSAFE = 1 << 11

_ARRAY_FIELDS = [
    # bodies
    ("body_parentid", np.int32), ("body_rootid", np.int32), ("body_weldid", np.int32),
    ("body_jntadr", np.int32), ("body_jntnum", np.int32), ("body_dofadr", np.int32),
    ("body_dofnum", np.int32), ("body_geomadr", np.int32), ("body_geomnum", np.int32),
    ("body_pos", np.float64), ("body_quat", np.float64), ("body_ipos", np.float64),
    ("body_iquat", np.float64), ("body_mass", np.float64), ("body_subtreemass", np.float64),
    ("body_inertia", np.float64), ("body_invweight0", np.float64),
    # joints / dofs
    ("jnt_type", np.int32), ("jnt_bodyid", np.int32), ("jnt_qposadr", np.int32),
    ("jnt_dofadr", np.int32), ("jnt_limited", np.int32), ("jnt_pos", np.float64),
    ("jnt_axis", np.float64), ("jnt_range", np.float64), ("jnt_margin", np.float64),
    ("jnt_stiffness", np.float64), ("jnt_solref", np.float64), ("jnt_solimp", np.float64),
    ("qpos0", np.float64), ("qpos_spring", np.float64),
    ("dof_bodyid", np.int32), ("dof_jntid", np.int32), ("dof_parentid", np.int32),
    ("dof_Madr", np.int32), ("dof_armature", np.float64), ("dof_damping", np.float64),
    ("dof_frictionloss", np.float64), ("dof_solref", np.float64), ("dof_solimp", np.float64),
    ("dof_invweight0", np.float64), ("dof_M0", np.float64),
    # geoms
    ("geom_type", np.int32), ("geom_bodyid", np.int32), ("geom_dataid", np.int32),
    ("geom_contype", np.int32), ("geom_conaffinity", np.int32), ("geom_condim", np.int32),
    ("geom_priority", np.int32), ("geom_pos", np.float64), ("geom_quat", np.float64),
    ("geom_size", np.float64), ("geom_friction", np.float64), ("geom_margin", np.float64),
    ("geom_gap", np.float64), ("geom_solref", np.float64), ("geom_solimp", np.float64),
    ("geom_solmix", np.float64), ("geom_rbound", np.float64),
    # meshes / hulls
    ("mesh_vertadr", np.int32), ("mesh_vertnum", np.int32), ("mesh_adjadr", np.int32),
    ("mesh_faceadr", np.int32), ("mesh_facenum", np.int32),
    ("hull_vert", np.float64), ("hull_adj_start", np.int32), ("hull_adj", np.int32),
    ("hull_face", np.int32),
    # actuators
    ("actuator_trnid", np.int32), ("actuator_ctrllimited", np.int32),
    ("actuator_forcelimited", np.int32), ("actuator_ctrlrange", np.float64),
    ("actuator_forcerange", np.float64), ("actuator_gear", np.float64),
    ("actuator_gainprm", np.float64), ("actuator_biasprm", np.float64),
    # contact filtering
    ("exclude_pair", np.int32), ("pair_geom1", np.int32), ("pair_geom2", np.int32),
    # equality
    ("eq_obj1id", np.int32), ("eq_obj2id", np.int32), ("eq_active", np.int32),
    ("eq_data", np.float64), ("eq_solref", np.float64), ("eq_solimp", np.float64),
]
_NAME_FIELDS = ["body_names", "joint_names", "geom_names", "actuator_names", "mesh_names"]
_OPT_FIELDS = ["opt_timestep", "opt_gravity", "opt_integrator", "opt_solver", "opt_iterations",
               "opt_tolerance", "opt_ls_iterations", "opt_ls_tolerance", "opt_impratio",
               "opt_mpr_iterations", "opt_mpr_tolerance", "opt_disableflags",
               "stat_meaninertia", "name"]


class _Opt:
    """`model.opt` view (mujoco-py exposes `sim.model.opt.timestep`, .gravity, ...)."""

    def __init__(self, m):
        object.__setattr__(self, "_m", m)

    def __getattr__(self, k):
        return getattr(self._m, "opt_" + k)

    def __setattr__(self, k, v):
        setattr(self._m, "opt_" + k, v)


class Model:
    def __init__(self):
        self.name = "model"
        self.opt_timestep = 0.002
        self.opt_gravity = np.array([0.0, 0.0, -9.81])
        self.opt_integrator = INT_EULER
        self.opt_solver = SOL_NEWTON
        self.opt_iterations = 100
        self.opt_tolerance = 1e-8
        self.opt_ls_iterations = 50
        self.opt_ls_tolerance = 0.01
        self.opt_impratio = 1.0
        self.opt_mpr_iterations = 50
        self.opt_mpr_tolerance = 1e-6
        self.opt_disableflags = 0
        self.stat_meaninertia = 1.0
        for n in _NAME_FIELDS:
            setattr(self, n, [])

    # ---- mujoco-py style accessors ------------------------------------------
    @property
    def opt(self):
        return _Opt(self)

    @property
    def nbody(self):
        return len(self.body_parentid)

    @property
    def njnt(self):
        return len(self.jnt_type)

    @property
    def nq(self):
        return len(self.jnt_type)

    @property
    def nv(self):
        return len(self.dof_bodyid)

    @property
    def nu(self):
        return len(self.actuator_trnid)

    @property
    def ngeom(self):
        return len(self.geom_type)

    @property
    def neq(self):
        return len(self.eq_obj1id)

    @property
    def nmesh(self):
        return len(self.mesh_vertadr)

    @property
    def npair(self):
        return len(self.pair_geom1)

    def body_name2id(self, name):
        return self.body_names.index(name)

    def joint_name2id(self, name):
        return self.joint_names.index(name)

    def geom_name2id(self, name):
        return self.geom_names.index(name)

    def actuator_name2id(self, name):
        return self.actuator_names.index(name)

    def get_joint_qpos_addr(self, name):
        return int(self.jnt_qposadr[self.joint_name2id(name)])

    def get_joint_qvel_addr(self, name):
        return int(self.jnt_dofadr[self.joint_name2id(name)])

    # ---- compile-time constants ------------------------------------------------
    def finalize(self):
        self._build_pairs()
        self._set_const()
        for name, dt in _ARRAY_FIELDS:
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=dt))

    def _build_pairs(self):
        """Static candidate geom pairs after MuJoCo's compile/broadphase filters
        [EXT]: contype/conaffinity, same weld body (covers both-static), weld
        parent-child (unless one is welded to world), explicit <exclude>
        (reference output/DexterousDynamos.xml:138-159)."""
        ex = {tuple(sorted(p)) for p in np.asarray(self.exclude_pair).reshape(-1, 2).tolist()}
        g1l, g2l = [], []
        fparent = not (self.opt_disableflags & DSBL_FILTERPARENT)
        for g1 in range(self.ngeom):
            for g2 in range(g1 + 1, self.ngeom):
                b1, b2 = int(self.geom_bodyid[g1]), int(self.geom_bodyid[g2])
                if b1 == b2:
                    continue
                if not ((self.geom_contype[g1] & self.geom_conaffinity[g2])
                        or (self.geom_contype[g2] & self.geom_conaffinity[g1])):
                    continue
                w1, w2 = int(self.body_weldid[b1]), int(self.body_weldid[b2])
                if w1 == w2:
                    continue
                wp1 = int(self.body_weldid[self.body_parentid[w1]])
                wp2 = int(self.body_weldid[self.body_parentid[w2]])
                if fparent and w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                    continue
                if tuple(sorted((b1, b2))) in ex:
                    continue
                # lower geom type first (plane before mesh), as MuJoCo orders a pair
                if self.geom_type[g1] > self.geom_type[g2]:
                    g1l.append(g2)
                    g2l.append(g1)
                else:
                    g1l.append(g1)
                    g2l.append(g2)
        self.pair_geom1 = np.array(g1l, dtype=np.int32)
        self.pair_geom2 = np.array(g2l, dtype=np.int32)

    def _set_const(self):
        nv, nbody = self.nv, self.nbody
        self.body_invweight0 = np.zeros((nbody, 2))
        self.dof_invweight0 = np.zeros(nv)
        self.dof_M0 = np.zeros(nv)
        if nv == 0:
            self.stat_meaninertia = 1.0
            return
        fk = host_fk(self, np.asarray(self.qpos0, dtype=np.float64))
        M = host_mass_matrix(self, fk)
        self.dof_M0 = np.diag(M).copy()
        self.stat_meaninertia = float(np.mean(np.diag(M)))
        Minv = np.linalg.inv(M)
        self.dof_invweight0 = np.diag(Minv).copy()
        for b in range(1, nbody):
            if self.body_weldid[b] == 0:
                continue
            jp, jr = host_jac(self, fk, b, fk["xipos"][b])
            self.body_invweight0[b, 0] = max(MINVAL, np.trace(jp @ Minv @ jp.T) / 3.0)
            self.body_invweight0[b, 1] = max(MINVAL, np.trace(jr @ Minv @ jr.T) / 3.0)

    # ---- (de)serialisation -----------------------------------------------------------
    def save(self, path):
        d = {name: np.asarray(getattr(self, name)) for name, _ in _ARRAY_FIELDS}
        meta = {n: getattr(self, n) for n in _NAME_FIELDS}
        meta.update({k: (np.asarray(getattr(self, k)).tolist()) for k in _OPT_FIELDS})
        meta["nM"] = int(self.nM)
        d["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(path, **d)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        m = cls()
        for name, dt in _ARRAY_FIELDS:
            setattr(m, name, np.ascontiguousarray(z[name], dtype=dt))
        meta = json.loads(bytes(z["meta_json"]).decode())
        for n in _NAME_FIELDS:
            setattr(m, n, list(meta[n]))
        for k in _OPT_FIELDS:
            v = meta[k]
            setattr(m, k, np.asarray(v, dtype=np.float64) if isinstance(v, list) else v)
        m.nM = int(meta["nM"])
        return m

    def copy(self):
        import copy
        return copy.deepcopy(self)


# ---------------------------------------------------------------------------
# Host-side NumPy kinematics / Jacobian mass matrix.  Used once at compile time
# for the qpos0 constants; independent of both the HIP kernels and oracle/ (it
# builds M from body Jacobians, not by CRB), so tests also use it as a third
# opinion on both.
# ---------------------------------------------------------------------------
def host_fk(m, qpos):
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.tile(np.array([1.0, 0, 0, 0]), (nb, 1))
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for b in range(1, nb):
        p = m.body_parentid[b]
        Rp = quat_to_mat(xquat[p])
        pos = xpos[p] + Rp @ m.body_pos[b]
        quat = quat_mul(xquat[p], m.body_quat[b])
        for k in range(int(m.body_jntnum[b])):
            j = int(m.body_jntadr[b]) + k
            R = quat_to_mat(quat)
            xanchor[j] = pos + R @ m.jnt_pos[j]
            xaxis[j] = R @ m.jnt_axis[j]
            dq = qpos[m.jnt_qposadr[j]] - m.qpos0[j]
            if m.jnt_type[j] == JNT_SLIDE:
                pos = pos + xaxis[j] * dq
            else:
                quat = quat_mul(quat, axis_angle_quat(m.jnt_axis[j], dq))
                pos = xanchor[j] - quat_to_mat(quat) @ m.jnt_pos[j]
        xquat[b] = quat_normalize(quat)
        xpos[b] = pos
    xmat = np.array([quat_to_mat(q) for q in xquat])
    xipos = xpos + np.einsum("bij,bj->bi", xmat, m.body_ipos)
    ximat = np.array([quat_to_mat(quat_mul(xquat[b], m.body_iquat[b])) for b in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xmat=xmat, xipos=xipos, ximat=ximat,
                xanchor=xanchor, xaxis=xaxis)


def host_jac(m, fk, body, point):
    """Translational / rotational Jacobian (3 x nv each) of `point` fixed to `body`."""
    jp = np.zeros((3, m.nv))
    jr = np.zeros((3, m.nv))
    b = body
    while b > 0:
        for k in range(int(m.body_jntnum[b])):
            j = int(m.body_jntadr[b]) + k
            d = int(m.jnt_dofadr[j])
            if m.jnt_type[j] == JNT_SLIDE:
                jp[:, d] = fk["xaxis"][j]
            else:
                jr[:, d] = fk["xaxis"][j]
                jp[:, d] = np.cross(fk["xaxis"][j], point - fk["xanchor"][j])
        b = m.body_parentid[b]
    return jp, jr


def host_mass_matrix(m, fk):
    M = np.diag(np.asarray(m.dof_armature, dtype=np.float64)).copy()
    for b in range(1, m.nbody):
        if m.body_mass[b] <= 0:
            continue
        jp, jr = host_jac(m, fk, b, fk["xipos"][b])
        Iw = fk["ximat"][b] @ np.diag(m.body_inertia[b]) @ fk["ximat"][b].T
        M += m.body_mass[b] * jp.T @ jp + jr.T @ Iw @ jr
    return M
