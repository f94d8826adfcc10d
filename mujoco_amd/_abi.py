"""ctypes binding of libbatchsim.so (include/batchsim.h).  No CPU fallback: if the
library is missing or no HIP device is present the product path raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BS_LIB") or os.path.join(_HERE, "csrc", "libbatchsim.so")

_I, _D = C.c_int32, C.c_double
_IP, _DP = C.POINTER(C.c_int32), C.POINTER(C.c_double)

# order == struct bs_model_desc in include/batchsim.h
MODEL_FIELDS = (
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


class ModelDesc(C.Structure):
    _fields_ = MODEL_FIELDS


class Options(C.Structure):
    _fields_ = [("max_contacts_per_env", _I), ("nsubsteps_hint", _I), ("max_contacts_per_model", _IP), ("reserved", _I * 4)]


# enum bs_field
FIELD_NAMES = ("qpos qvel ctrl qfrc_applied qacc qacc_warmstart time xpos xquat xipos xanchor xaxis "
               "rootcom cinert crb cdof qM qLD qLDiagInv cvel qfrc_bias qfrc_passive actuator_force "
               "qfrc_actuator qfrc_smooth qacc_smooth qfrc_constraint efc_D efc_aref efc_force "
               "solver_iter warning nefc xmat geom_xpos geom_xmat ncon").split()
FIELD_ID = {n: i for i, n in enumerate(FIELD_NAMES)}

# every symbol include/batchsim.h declares
SYMBOLS = ("bs_version bs_last_error bs_device_count bs_model_create bs_model_destroy bs_sim_create "
           "bs_sim_destroy bs_set_stream bs_step bs_forward bs_reset bs_get_ptr bs_copy_field "
           "bs_get_contacts bs_collision_stats bs_sync bs_stage_times bs_timed_steps bs_sim_num_parts bs_sim_part bs_pack_obs "
           "bs_reset_device_mask bs_state_rows bs_get_state bs_set_state bs_get_stream bs_get_contact_geometry bs_reset_noalloc bs_sim_kernel_path bs_env_pre bs_env_post bs_sim_footprint").split()


class BatchSimError(RuntimeError):
    """Raised for every non-zero bs_status (the wrapper's MujocoException counterpart)."""


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BatchSimError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    # torch first: it ships its own HIP runtime, and the zero-copy views need torch and the library to share ONE runtime.
    # Loaded the other way round (libbatchsim.so pulls in /opt/rocm's libamdhip64 before torch loads its bundled copy)
    # torch's lazy initialisation reports "No HIP GPUs are available".
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.bs_version.restype = C.c_int
    L.bs_last_error.restype = C.c_char_p
    L.bs_device_count.argtypes = [C.POINTER(C.c_int)]
    L.bs_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    L.bs_model_destroy.argtypes = [vp]
    L.bs_sim_create.argtypes = [C.POINTER(vp), C.c_int, _IP, C.c_int64, C.c_int, C.POINTER(Options),
                                C.POINTER(vp)]
    L.bs_sim_destroy.argtypes = [vp]
    L.bs_set_stream.argtypes = [vp, vp, C.c_int]
    L.bs_step.argtypes = [vp, C.c_int]
    L.bs_forward.argtypes = [vp]
    L.bs_reset.argtypes = [vp, C.POINTER(C.c_uint8)]
    L.bs_get_ptr.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                             C.POINTER(C.c_int)]
    L.bs_copy_field.argtypes = [vp, C.c_int, vp, C.c_int]
    L.bs_get_contacts.argtypes = [vp, C.c_int64, _IP, _IP, _DP, C.c_int, C.POINTER(C.c_int)]
    L.bs_collision_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.bs_sync.argtypes = [vp]
    L.bs_stage_times.argtypes = [vp, _DP, C.c_int]
    L.bs_timed_steps.argtypes = [vp, C.c_int, _DP]
    L.bs_pack_obs.argtypes = [vp, vp, C.c_int64]
    L.bs_reset_device_mask.argtypes = [vp, vp]
    L.bs_reset_noalloc.argtypes = [vp, vp]
    L.bs_state_rows.argtypes = [vp, C.POINTER(C.c_int)]
    L.bs_get_state.argtypes = [vp, vp, C.c_int64]
    L.bs_set_state.argtypes = [vp, vp, C.c_int64]
    L.bs_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.bs_sim_kernel_path.argtypes = [vp, C.c_char_p, C.c_int]
    L.bs_sim_footprint.argtypes = [vp, C.POINTER(C.c_int64)]
    L.bs_env_pre.argtypes = [vp, vp, C.c_int64]
    L.bs_env_post.argtypes = [vp, vp, C.c_int64, vp, C.c_int, vp, vp, C.c_int64, vp, vp]
    L.bs_get_contact_geometry.argtypes = [vp, C.c_int64, _DP, _DP, C.c_int, C.POINTER(C.c_int)]
    for s in SYMBOLS:
        if s not in ("bs_last_error",):
            getattr(L, s).restype = C.c_int
    L.bs_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise BatchSimError(f"[bs_status {rc}] {lib().bs_last_error().decode(errors='replace')}")


def make_desc(model):
    """Fill a bs_model_desc from a compiled `Model`; returns (desc, keepalive list)."""
    desc = ModelDesc()
    keep = []
    for name, kind in MODEL_FIELDS:
        if kind is _IP or kind is _DP:
            arr = np.ascontiguousarray(getattr(model, name), dtype=np.int32 if kind is _IP else np.float64)
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
    return desc, keep
