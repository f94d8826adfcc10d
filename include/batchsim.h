/* batchsim.h — C-ABI of libbatchsim.so: the MI355X-native batched physics step.
 *
 * This is the drop-in boundary of SURVEY.md §8b.  The reference reaches its physics step
 * through mujoco-py's Cython binding of libmujoco210.so; its only call sites are
 *   reference src/Mujoco_XML.py:4    from mujoco_py import load_model_from_xml, MjSim, MjViewer
 *   reference src/Mujoco_XML.py:341  model = load_model_from_xml(model_str)
 *   reference src/Mujoco_XML.py:342  sim = MjSim(model)
 * and BASELINE.json's north_star names the rest of the surface (sim.step(),
 * sim.data.qpos/qvel/ctrl).  Each entry point below cites the mujoco-py / MuJoCo call it
 * replaces.  Plain pointers and sizes only; no torch types.  Every function returns 0 or a
 * negative bs_status; bs_last_error() gives the message (thread-local).
 *
 * Device layout: every per-environment field is a row-major [rows][E_pad] array of its
 * element type (env-minor, so a wavefront's 64 lanes read 512 contiguous bytes); E_pad is
 * n_envs rounded up to 64.  bs_get_ptr reports (rows, n_envs) with strides (E_pad, 1).
 *
 * Threading: one bs_sim = one device + one HIP stream; calls on a handle must be serialised
 * by the caller; bs_step / bs_forward are asynchronous until bs_sync or a host copy.
 */
#ifndef BATCHSIM_H
#define BATCHSIM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BS_VERSION 200

typedef enum bs_status {
  BS_OK = 0,
  BS_ERR_INVALID = -1,     /* bad argument / inconsistent model description */
  BS_ERR_NODEVICE = -2,    /* no usable HIP device: there is NO CPU fallback */
  BS_ERR_HIP = -3,         /* a HIP runtime call failed */
  BS_ERR_UNSUPPORTED = -4, /* feature outside the implemented subset */
  BS_ERR_NOMEM = -5
} bs_status;

/* Flattened model (the build's mjModel): arrays produced by the host-side MJCF compiler
 * (mujoco_amd/mjcf.py), copied to the device once by bs_model_create.  Field names follow
 * mjModel.  Replaces the result of mujoco-py load_model_from_xml (reference
 * src/Mujoco_XML.py:341).  body_quat, body_iquat and jnt_axis are unit vectors in an mjModel
 * (MuJoCo's compiler normalises them); the tree-parallel kernels' per-lane tables rescale ones
 * that are not, the generic kernels use the arrays as given. */
typedef struct bs_model_desc {
  int32_t nbody, njnt, nv, nu, ngeom, neq, nmesh, nM, npair, nhullvert, nhulladj;
  int32_t integrator; /* 0 Euler, 1 RK4 */
  int32_t solver;     /* 0 PGS, 1 CG, 2 Newton */
  int32_t iterations, ls_iterations, disableflags, mpr_iterations;
  double timestep, gravity[3], tolerance, ls_tolerance, impratio, mpr_tolerance, meaninertia;
  const int32_t *body_parentid, *body_rootid, *body_weldid, *body_jntadr, *body_jntnum,
      *body_dofadr, *body_dofnum;
  const double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_subtreemass,
      *body_inertia, *body_invweight0;
  const int32_t *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_stiffness, *jnt_solref,
      *jnt_solimp, *qpos0, *qpos_spring;
  const int32_t *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr;
  const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_solref, *dof_solimp,
      *dof_invweight0;
  const int32_t *geom_type, *geom_bodyid, *geom_dataid, *geom_condim, *geom_priority;
  const double *geom_pos, *geom_quat, *geom_size, *geom_friction, *geom_margin, *geom_gap,
      *geom_solref, *geom_solimp, *geom_solmix, *geom_rbound;
  const int32_t *mesh_vertadr, *mesh_vertnum, *hull_adj_start, *hull_adj;
  const double *hull_vert;
  const int32_t *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited;
  const double *actuator_ctrlrange, *actuator_forcerange, *actuator_gear, *actuator_gainprm,
      *actuator_biasprm;
  const int32_t *pair_geom1, *pair_geom2;
  const int32_t *eq_obj1id, *eq_obj2id, *eq_active;
  const double *eq_data, *eq_solref, *eq_solimp;
} bs_model_desc;

typedef struct bs_options {
  int32_t max_contacts_per_env; /* cap for the compacted contact list (0 = default 32; up to 32 the 32-lane tree-parallel
                                   kernels run, up to 64 the wide ones — one environment per wavefront, about half the
                                   throughput —, beyond that the generic one-environment-per-lane kernels) */
  int32_t nsubsteps_hint;       /* informational */
  const int32_t *max_contacts_per_model; /* nullable; mixed-model sims: one cap per model (0 = max_contacts_per_env), so
                                            that only the model that needs a longer list pays for the wide kernels */
  int32_t reserved[4];
} bs_options;

typedef struct bs_model bs_model;
typedef struct bs_sim bs_sim;

/* Field ids for bs_get_ptr / bs_copy_field: the mjData members mujoco-py exposes as
 * sim.data.<name> (BASELINE.json north_star: qpos/qvel/ctrl). */
typedef enum bs_field {
  BS_F_QPOS = 0, BS_F_QVEL, BS_F_CTRL, BS_F_QFRC_APPLIED, BS_F_QACC, BS_F_QACC_WARMSTART,
  BS_F_TIME,
  BS_F_XPOS, BS_F_XQUAT, BS_F_XIPOS, BS_F_XANCHOR, BS_F_XAXIS, BS_F_ROOTCOM,
  BS_F_CINERT, BS_F_CRB, BS_F_CDOF, BS_F_QM, BS_F_QLD, BS_F_QLDIAGINV,
  BS_F_CVEL, BS_F_QFRC_BIAS, BS_F_QFRC_PASSIVE, BS_F_ACTUATOR_FORCE, BS_F_QFRC_ACTUATOR,
  BS_F_QFRC_SMOOTH, BS_F_QACC_SMOOTH, BS_F_QFRC_CONSTRAINT,
  BS_F_EFC_D, BS_F_EFC_AREF, BS_F_EFC_FORCE,
  BS_F_SOLVER_ITER, /* int32 [1][E] */
  BS_F_WARNING,     /* int32 [1][E]: bit0 bad qpos, bit1 bad qvel, bit2 bad qacc, bit3 contact overflow */
  BS_F_NEFC,        /* int32 [1][E]: active constraint rows */
  BS_F_XMAT,        /* derived on demand from xquat */
  BS_F_GEOM_XPOS, BS_F_GEOM_XMAT,
  BS_F_NCON,        /* int32 [1][E]: contacts of the last collision pass (0 rows when contacts are disabled) */
  BS_F_COUNT
} bs_field;

/* mujoco-py has no counterpart (library metadata). */
int bs_version(void);
const char *bs_last_error(void);
int bs_device_count(int *n);

/* Replaces mujoco_py.load_model_from_xml's native half: mj_loadXML's result as device arrays
 * (reference src/Mujoco_XML.py:341).  Host arrays may be released after the call. */
int bs_model_create(const bs_model_desc *desc, bs_model **out);
int bs_model_destroy(bs_model *m);

/* Replaces mujoco_py.MjSim(model) = mj_makeData + mj_resetData (reference
 * src/Mujoco_XML.py:342), for n_envs lockstep environments on HIP device `device`.
 * env_model_id (nullable) selects a model per environment and must be non-decreasing. */
int bs_sim_create(const bs_model *const *models, int n_models, const int32_t *env_model_id,
                  int64_t n_envs, int device, const bs_options *opt, bs_sim **out);
int bs_sim_destroy(bs_sim *s);

/* Mixed-model batches (BASELINE.json configs[3]; no counterpart in mujoco-py, where one MjSim holds
 * one model): with n_models > 1 the sim is a set of parts, one per model, over contiguous ranges
 * of environments.  bs_step / bs_forward / bs_reset / bs_sync / bs_set_stream / bs_get_contacts /
 * bs_collision_stats act on all of them; fields are addressed per part (row counts differ between
 * models): bs_sim_part returns a borrowed handle that takes bs_get_ptr / bs_copy_field and dies
 * with the parent.  A single-model sim has 0 parts. */
int bs_sim_num_parts(const bs_sim *s);
int bs_sim_part(bs_sim *s, int part, bs_sim **out_part, int64_t *env_start, int64_t *n_envs);

/* Run on a caller-owned hipStream_t (e.g. PyTorch's current stream; 0 is the legacy null
 * stream) instead of the sim's own; use_own_stream != 0 switches back to the sim's stream. */
int bs_set_stream(bs_sim *s, void *hip_stream, int use_own_stream);

/* Replaces MjSim.step() = mj_step, n_steps times (mujoco-py's nsubsteps), all envs in lockstep. */
int bs_step(bs_sim *s, int n_steps);
/* Replaces MjSim.forward() = mj_forward. */
int bs_forward(bs_sim *s);
/* Replaces MjSim.reset() = mj_resetData; env_mask (host, n_envs bytes, nullable = all).  The mask is
 * staged through buffers the sim owns: no allocation and no wait for the device per call. */
int bs_reset(bs_sim *s, const uint8_t *env_mask);
/* The same with the mask already in device memory of the sim's GPU (n_envs bytes, non-zero = reset):
 * one kernel on the sim's stream, for rollout loops that compute `done` on the device (no mujoco-py
 * counterpart: there one MjSim is one environment). */
int bs_reset_device_mask(bs_sim *s, const uint8_t *dev_mask);

/* Masked reset that ALLOCATES NOTHING.  No counterpart in the reference or in mujoco-py: it exists for
 * callers that cannot call hipMalloc — code under HIP graph capture, tight serving / rollout loops,
 * harnesses that forbid allocation.  Same result as bs_reset(s, env_mask) with a non-null mask, but the
 * device copy of the mask is staged in a file-scope `static __device__` array of the library
 * (BS_NOALLOC_MAX_ENVS bytes, held by every GPU the library runs on until the process ends) instead of
 * memory the sim allocated; the array's device address is looked up once per device
 * (hipGetSymbolAddress).  Sims with more than BS_NOALLOC_MAX_ENVS environments get BS_ERR_UNSUPPORTED.
 * env_mask: host memory, n_envs bytes; if it is pinned / registered the call is fully asynchronous, a
 * pageable buffer is waited for (stream synchronisation) before the call returns.  Calls no hipMalloc /
 * hipFree / allocating routine (the first call on a device looks up the array's address and creates one
 * event; make one warm-up call before a loop that must not allocate).  Every call on a GPU shares that GPU's
 * array: the library orders the calls it is given — a call's copy into the array waits (hipStreamWaitEvent, on
 * the device) for the previous call's reset kernels on that GPU, whichever sim and stream they ran on, and the
 * host side of a call is under a mutex — so back-to-back calls on different sims / streams / host threads do not
 * overwrite a mask that has not been read.  Under HIP graph capture the mask must be pinned and the call must not
 * be the first on its device; the cross-call event is not part of a captured graph, so REPLAYS that may overlap
 * with other resets on the same GPU are the caller's to serialise. */
#define BS_NOALLOC_MAX_ENVS (1 << 19)
#define BS_NOALLOC_MAX_DEVICES 64
int bs_reset_noalloc(bs_sim *s, const uint8_t *env_mask);

/* Replaces MjSim.get_state() / set_state() (MjSimState: time, qpos, qvel, act, udd_state; reference call
 * site of the sim object src/Mujoco_XML.py:341-345): the state as ONE device buffer [rows][ld] of the
 * caller, rows = 1 + nq + 3 nv in the order time | qpos | qvel | qacc | qacc_warmstart (the last two make
 * a restored run bit-identical: the solvers warm-start from qacc_warmstart), ld >= n_envs.  Copies are
 * enqueued on the sim's stream. */
int bs_state_rows(const bs_sim *s, int *rows);
int bs_get_state(bs_sim *s, double *dev_buf, int64_t ld);
int bs_set_state(bs_sim *s, const double *dev_buf, int64_t ld);
/* Which kernel family bs_step runs for this sim: returns 1 and writes "tree-parallel (...)" when the model is
 * eligible for the one-lane-per-body kernels, 0 and "serial (...): <reason>" when it runs the generic
 * one-environment-per-lane kernels (about 40x slower: more than 31 bodies, two joints on a body, dof depth
 * above 6, ...), < 0 (a bs_status) on error.  No counterpart in the reference (MuJoCo has one code path);
 * mujoco_amd.BatchSim warns once when the answer is 0. */
int bs_sim_kernel_path(const bs_sim *s, char *buf, int cap);
/* Device memory the sim holds, in bytes per arena: out[0] state arena (every [rows][E_pad] field of bs_get_ptr
 * and the intermediates of the generic kernels), out[1] contact arena (candidate + contact lists, narrowphase
 * work list), out[2] geom-frame records of the persistent narrowphase, out[3] buffers of the serial contact solver
 * (0 until bs_stage_times or an ineligible model needs them), out[4] = E_pad (n_envs rounded up to 64; a mixed
 * sim: summed over its parts, like the bytes).  When an arena does not fit, bs_sim_create fails with BS_ERR_NOMEM
 * and a message that names the arena, its bytes per environment and the total.  No counterpart in mujoco-py
 * (mjData is one host allocation per environment); used to size batches against the 288 GB of an MI355X. */
int bs_sim_footprint(const bs_sim *s, int64_t out[5]);
/* The stream the sim's kernels are enqueued on (its own, or the one given to bs_set_stream). */
int bs_get_stream(bs_sim *s, void **hip_stream);

/* Zero-copy device view of a field (what sim.data.<field> wraps). shape/stride in elements. */
int bs_get_ptr(bs_sim *s, int field_id, void **dev_ptr, int64_t shape[2], int64_t stride[2],
               int *is_int32);
/* direction 0: device -> host (dense [rows][n_envs]); 1: host -> device. Synchronises. */
int bs_copy_field(bs_sim *s, int field_id, void *host, int direction);

/* sim.data.ncon / sim.data.contact[i].geom1, geom2, dist for one environment. */
int bs_get_contacts(bs_sim *s, int64_t env, int32_t *geom1, int32_t *geom2, double *dist, int cap,
                    int *n);

/* sim.data.contact[i].pos (3) and .frame (9, rows = normal, tangent1, tangent2) for one environment. */
int bs_get_contact_geometry(bs_sim *s, int64_t env, double *pos, double *frame, int cap, int *n);

/* Work-model counters of the collision stage since creation / last reset:
 * out[0] narrowphase pairs, out[1] MPR support iterations, out[2] contacts.  No mujoco-py
 * counterpart (mjData has no such counters); used by bench.py to report N_np and ncon. */
int bs_collision_stats(bs_sim *s, uint64_t out[3], int reset);

int bs_sync(bs_sim *s);
/* Runs one step stage by stage with HIP events between the stages and returns the elapsed
 * ms per stage: [0] position (A1-A4) [1] collision (A5-A6) [2] velocity+actuation+accel (A8-A10)
 * [3] constraints (A7+A11) [4] integrate (A12/A13).  Advances the state by one step (Euler for a sim
 * with contacts; RK4 with contacts returns BS_ERR_UNSUPPORTED). */
int bs_stage_times(bs_sim *s, double *ms, int n);
/* Times n_steps of bs_step with HIP events on the sim's stream; returns elapsed ms. */
int bs_timed_steps(bs_sim *s, int n_steps, double *elapsed_ms);
/* Observation / reward kernel for the learner glue (the reference has no
 * environment code, src/Mujoco_XML.py:327-330 is a TODO): one launch on the sim's stream writes, for
 * every environment e, out[e * row_stride + 0 .. nq) = qpos, [nq .. nq+nv) = qvel (fp32) and
 * out[e * row_stride + nq + nv] = -sum_u (qpos[joint of actuator u] - ctrl[u])^2, the placeholder
 * reward of BASELINE.json configs[4].  `out_dev` is device memory of the sim's GPU with room for
 * n_envs * row_stride floats; row_stride >= nq + nv + 1. */
int bs_pack_obs(bs_sim *s, float *out_dev, int64_t row_stride);
/* The two ends of a control period for the learner glue, each ONE launch on the sim's stream (the reference has
 * no environment code, src/Mujoco_XML.py:327-330; definitions in mujoco_amd/env.py).
 * bs_env_pre: ctrl[u] = action[e * ld + u] clamped to actuator u's ctrlrange, for every environment e
 * (action_dev: n_envs rows of ld >= nu doubles in device memory).
 * bs_env_post, after bs_step, with the same action: reward[e] = -sum_u (qpos[joint of actuator u] - clamped action[u])^2; t[e] += 1;
 * flags[e] = 1 (done: failed or t[e] >= episode_steps) | 2 (failed: mj_check* reset the environment, warning
 * bits 0-2) | 4 (contact / candidate list overflowed, warning bit 3, cleared here); term_obs[e] = qpos | qvel as
 * fp32 as the period left them; finished environments are reset (bs_reset's arithmetic, t[e] = 0) and obs[e] is
 * what the next action is chosen from (= term_obs[e] for the others).  obs / term_obs: n_envs rows of
 * row_stride >= nq + nv floats; t: int32[n_envs]; reward: float[n_envs]; flags: uint8[n_envs], all device memory. */
int bs_env_pre(bs_sim *s, const double *action_dev, int64_t ld);
int bs_env_post(bs_sim *s, const double *action_dev, int64_t ld, int32_t *t_dev, int episode_steps, float *obs_dev,
                float *term_obs_dev, int64_t row_stride, float *reward_dev, uint8_t *flags_dev);

#ifdef __cplusplus
}
#endif
#endif
