"""ctypes wrapper around liboracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see oracle.h): may be imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by mujoco_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

_I, _D = C.c_int32, C.c_double
_IP, _DP = C.POINTER(C.c_int32), C.POINTER(C.c_double)

# (name, kind) in the exact order of `struct orc_model` in oracle.h
_MODEL_FIELDS = (
    [(n, _I) for n in "nbody njnt nv nu ngeom neq nmesh nM npair nhullvert nhulladj".split()]
    + [(n, _I) for n in "integrator solver iterations ls_iterations disableflags mpr_iterations".split()]
    + [("timestep", _D), ("gravity", _D * 3), ("tolerance", _D), ("ls_tolerance", _D),
       ("impratio", _D), ("mpr_tolerance", _D), ("meaninertia", _D)]
    + [(n, _IP) for n in "body_parentid body_rootid body_weldid body_jntadr body_jntnum body_dofadr body_dofnum".split()]
    + [(n, _DP) for n in "body_pos body_quat body_ipos body_iquat body_mass body_subtreemass body_inertia body_invweight0".split()]
    + [(n, _IP) for n in "jnt_type jnt_bodyid jnt_qposadr jnt_dofadr jnt_limited".split()]
    + [(n, _DP) for n in "jnt_pos jnt_axis jnt_range jnt_margin jnt_stiffness jnt_solref jnt_solimp qpos0 qpos_spring".split()]
    + [(n, _IP) for n in "dof_bodyid dof_jntid dof_parentid dof_Madr".split()]
    + [(n, _DP) for n in "dof_armature dof_damping dof_frictionloss dof_solref dof_solimp dof_invweight0".split()]
    + [(n, _IP) for n in "geom_type geom_bodyid geom_dataid geom_condim geom_priority".split()]
    + [(n, _DP) for n in "geom_pos geom_quat geom_size geom_friction geom_margin geom_gap geom_solref geom_solimp geom_solmix geom_rbound".split()]
    + [(n, _IP) for n in "mesh_vertadr mesh_vertnum hull_adj_start hull_adj".split()]
    + [("hull_vert", _DP)]
    + [(n, _IP) for n in "actuator_trnid actuator_ctrllimited actuator_forcelimited".split()]
    + [(n, _DP) for n in "actuator_ctrlrange actuator_forcerange actuator_gear actuator_gainprm actuator_biasprm".split()]
    + [(n, _IP) for n in "pair_geom1 pair_geom2".split()]
    + [(n, _IP) for n in "eq_obj1id eq_obj2id eq_active".split()]
    + [(n, _DP) for n in "eq_data eq_solref eq_solimp".split()]
)


class _CModel(C.Structure):
    _fields_ = _MODEL_FIELDS


class _CContact(C.Structure):
    _fields_ = [("dist", _D), ("pos", _D * 3), ("frame", _D * 9), ("includemargin", _D),
                ("friction", _D * 5), ("solref", _D * 2), ("solimp", _D * 5),
                ("dim", _I), ("geom1", _I), ("geom2", _I), ("efc_address", _I)]


MAXCON = 128

_DATA_D1 = "qpos qvel ctrl qfrc_applied qacc qacc_warmstart".split()
_DATA_D2 = "xpos xquat xmat xipos ximat xanchor xaxis geom_xpos geom_xmat".split()
_DATA_D3 = "subtree_com cinert cdof crb qM qLD qLDiagInv".split()
_DATA_D4 = ("cvel cdof_dot qfrc_bias qfrc_passive actuator_force qfrc_actuator qfrc_smooth "
            "qacc_smooth qfrc_constraint").split()
_DATA_E = ("efc_J efc_pos efc_margin efc_frictionloss efc_diagApprox efc_KBIP efc_D efc_R "
           "efc_vel efc_aref efc_force").split()


class _CData(C.Structure):
    _fields_ = (
        [("time", _D)]
        + [(n, _DP) for n in _DATA_D1 + _DATA_D2 + _DATA_D3 + _DATA_D4]
        + [(n, _I) for n in "nefc nefc_max ncon ne nf nl".split()]
        + [(n, _IP) for n in "efc_type efc_id efc_state".split()]
        + [(n, _DP) for n in _DATA_E]
        + [("contact", _CContact * MAXCON)]
        + [("solver_iter", _I), ("warning", _I)]
        + [("stat_npair_narrow", C.c_int64), ("stat_mpr_iter", C.c_int64)]
        + [("scratch", _DP)]
    )


def build(force=False):
    """Compile liboracle.so with gcc (called by __graft_entry__.build())."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_collide.c", "oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_model_create.restype = C.POINTER(_CModel)
        L.orc_model_create.argtypes = [C.POINTER(_CModel)]
        L.orc_model_destroy.argtypes = [C.POINTER(_CModel)]
        L.orc_data_create.restype = C.POINTER(_CData)
        L.orc_data_create.argtypes = [C.POINTER(_CModel)]
        L.orc_data_destroy.argtypes = [C.POINTER(_CData)]
        for fn in ("orc_reset orc_kinematics orc_com_pos orc_crb orc_factor_m orc_collision "
                   "orc_make_constraint orc_fwd_velocity orc_fwd_actuation orc_fwd_acceleration "
                   "orc_fwd_constraint orc_forward orc_step").split():
            getattr(L, fn).argtypes = [C.POINTER(_CModel), C.POINTER(_CData)]
            getattr(L, fn).restype = None
        L.orc_solve_m.argtypes = [C.POINTER(_CModel), C.POINTER(_CData), _DP]
        L.orc_mul_m.argtypes = [C.POINTER(_CModel), C.POINTER(_CData), _DP, _DP]
        L.orc_full_m.argtypes = [C.POINTER(_CModel), C.POINTER(_CData), _DP]
        L.orc_rne.argtypes = [C.POINTER(_CModel), C.POINTER(_CData), C.c_int, _DP]
        L.orc_cost.argtypes = [C.POINTER(_CModel), C.POINTER(_CData), _DP]
        L.orc_cost.restype = _D
        L.orc_narrow_pair.argtypes = [C.POINTER(_CModel), C.POINTER(_CData), C.c_int, C.c_int, _D,
                                      _DP, _DP, _DP]
        L.orc_narrow_pair.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(_DP)


class OracleSim:
    """Single-environment CPU simulator with a mujoco-py `MjSim`-like surface."""

    def __init__(self, model):
        L = lib()
        self.model = model
        desc = _CModel()
        keep = []
        for name, kind in _MODEL_FIELDS:
            if kind is _IP or kind is _DP:
                arr = np.ascontiguousarray(getattr(model, name),
                                           dtype=np.int32 if kind is _IP else np.float64)
                keep.append(arr)
                setattr(desc, name, arr.ctypes.data_as(kind))
        for n in "nbody njnt nv nu ngeom neq nmesh nM npair".split():
            setattr(desc, n, int(getattr(model, n)))
        desc.nhullvert = len(model.hull_vert)
        desc.nhulladj = len(model.hull_adj)
        desc.integrator = int(model.opt_integrator)
        desc.solver = int(model.opt_solver)
        desc.iterations = int(model.opt_iterations)
        desc.ls_iterations = int(model.opt_ls_iterations)
        desc.disableflags = int(model.opt_disableflags)
        desc.mpr_iterations = int(model.opt_mpr_iterations)
        desc.timestep = float(model.opt_timestep)
        desc.gravity = (_D * 3)(*[float(x) for x in model.opt_gravity])
        desc.tolerance = float(model.opt_tolerance)
        desc.ls_tolerance = float(model.opt_ls_tolerance)
        desc.impratio = float(model.opt_impratio)
        desc.mpr_tolerance = float(model.opt_mpr_tolerance)
        desc.meaninertia = float(model.stat_meaninertia)
        self._m = L.orc_model_create(C.byref(desc))
        del keep
        self._d = L.orc_data_create(self._m)
        m = model
        shapes = dict(
            qpos=(m.nq,), qvel=(m.nv,), ctrl=(m.nu,), qfrc_applied=(m.nv,), qacc=(m.nv,),
            qacc_warmstart=(m.nv,), xpos=(m.nbody, 3), xquat=(m.nbody, 4), xmat=(m.nbody, 9),
            xipos=(m.nbody, 3), ximat=(m.nbody, 9), xanchor=(m.njnt, 3), xaxis=(m.njnt, 3),
            geom_xpos=(m.ngeom, 3), geom_xmat=(m.ngeom, 9), subtree_com=(m.nbody, 3),
            cinert=(m.nbody, 10), cdof=(m.nv, 6), crb=(m.nbody, 10), qM=(m.nM,), qLD=(m.nM,),
            qLDiagInv=(m.nv,), cvel=(m.nbody, 6), cdof_dot=(m.nv, 6), qfrc_bias=(m.nv,),
            qfrc_passive=(m.nv,), actuator_force=(m.nu,), qfrc_actuator=(m.nv,),
            qfrc_smooth=(m.nv,), qacc_smooth=(m.nv,), qfrc_constraint=(m.nv,))
        d = self._d.contents
        for k, shp in shapes.items():
            n = int(np.prod(shp))
            if n == 0:
                setattr(self, k, np.zeros(shp))
            else:
                setattr(self, k, np.ctypeslib.as_array(getattr(d, k), shape=(n,)).reshape(shp))

    def __del__(self):
        try:
            lib().orc_data_destroy(self._d)
            lib().orc_model_destroy(self._m)
        except Exception:
            pass

    # -- scalar fields
    @property
    def time(self):
        return self._d.contents.time

    @time.setter
    def time(self, v):
        self._d.contents.time = v

    @property
    def nefc(self):
        return self._d.contents.nefc

    @property
    def ncon(self):
        return self._d.contents.ncon

    @property
    def solver_iter(self):
        return self._d.contents.solver_iter

    @property
    def warning(self):
        return self._d.contents.warning

    def efc(self, name):
        d = self._d.contents
        n = d.nefc
        if name in ("type", "id", "state"):
            return np.ctypeslib.as_array(getattr(d, "efc_" + name), shape=(d.nefc_max,))[:n].copy()
        w = {"J": self.model.nv, "KBIP": 4}.get(name, 1)
        a = np.ctypeslib.as_array(getattr(d, "efc_" + name), shape=(d.nefc_max * w,))[:n * w].copy()
        return a.reshape(n, w) if name in ("J", "KBIP") else a

    def contacts(self):
        d = self._d.contents
        out = []
        for i in range(d.ncon):
            c = d.contact[i]
            out.append(dict(geom1=c.geom1, geom2=c.geom2, dist=c.dist, pos=np.array(c.pos[:]),
                            frame=np.array(c.frame[:]), includemargin=c.includemargin, dim=c.dim))
        return out

    # -- pipeline
    def _call(self, fn):
        getattr(lib(), fn)(self._m, self._d)

    def step(self, n=1):
        for _ in range(n):
            self._call("orc_step")

    def forward(self):
        self._call("orc_forward")

    def reset(self):
        self._call("orc_reset")

    def stage(self, name):
        self._call("orc_" + name)

    def full_m(self):
        out = np.zeros((self.model.nv, self.model.nv))
        lib().orc_full_m(self._m, self._d, _dp(out))
        return out

    def solve_m(self, x):
        x = np.array(x, dtype=np.float64)
        lib().orc_solve_m(self._m, self._d, _dp(x))
        return x

    def mul_m(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros_like(v)
        lib().orc_mul_m(self._m, self._d, _dp(out), _dp(v))
        return out

    def rne(self, flg_acc):
        out = np.zeros(self.model.nv)
        lib().orc_rne(self._m, self._d, int(flg_acc), _dp(out))
        return out

    def cost(self, qacc):
        q = np.ascontiguousarray(qacc, dtype=np.float64)
        return lib().orc_cost(self._m, self._d, _dp(q))

    def narrow_pair(self, g1, g2, margin):
        dist = np.zeros(1)
        pos = np.zeros(3)
        nrm = np.zeros(3)
        n = lib().orc_narrow_pair(self._m, self._d, g1, g2, margin, _dp(dist), _dp(pos), _dp(nrm))
        return n, float(dist[0]), pos, nrm
