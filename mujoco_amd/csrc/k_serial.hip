// k_serial.hip — kernels of the serial layout (one environment per wavefront lane, intermediates in
// the [row][E] arena; physics_dev.h, physics_contact.h) and their launchers.  Fallback for models the
// tree-parallel kernels cannot take, the stage-by-stage timing, reset / frames / observation packing.
#include "kernels.h"

__global__ __launch_bounds__(BS_BLOCK) void k_step(DevModel m, DevData d, int nsteps) {
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  for (int s = 0; s < nsteps; s++) env_step(m, d, e);
}
__global__ __launch_bounds__(BS_BLOCK) void k_forward(DevModel m, DevData d) {
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  env_forward(m, d, e);
}
// single stage, for per-stage timing: 0 check+position+factor, 2 velocity, 3 constraint, 4 integrate
__global__ __launch_bounds__(BS_BLOCK) void k_stage(DevModel m, DevData d, int stage) {
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  switch (stage) {
    case 0:
      env_check(m, d, e, d.qpos, m.njnt, 0);
      env_check(m, d, e, d.qvel, m.nv, 1);
      stage_position(m, d, e);
      stage_factor(m, d, e);
      break;
    case 2: stage_velocity(m, d, e); break;
    case 3:
      if (m.solver == 0) stage_constraint_pgs(m, d, e);
      else stage_constraint(m, d, e);
      break;
    case 4:
      if (env_check(m, d, e, d.qacc, m.nv, 2)) env_forward(m, d, e);
      if (m.integrator == 1) stage_rk4(m, d, e);
      else stage_euler(m, d, e);
      break;
    default: break;
  }
}
__global__ __launch_bounds__(BS_BLOCK) void k_reset(DevModel m, DevData d, const uint8_t* mask) {
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  if (mask && !mask[e]) return;
  env_reset(m, d, e);
  const long E = d.E;
  AT(d.warning, 0) = 0;
}
// A14 mj_checkPos / mj_checkVel on their own (the RK4 step with contacts starts with it)
__global__ __launch_bounds__(BS_BLOCK) void k_check_state(DevModel m, DevData d) {
  const long e = d.env0 + (long)blockIdx.x * BS_BLOCK + threadIdx.x;  // a range [env0, n_envs) of the batch
  if (e >= d.n_envs) return;
  env_check(m, d, e, d.qpos, m.njnt, 0);
  env_check(m, d, e, d.qvel, m.nv, 1);
}
// RK4 across launches (contacts): see rk4_advance
__global__ __launch_bounds__(BS_BLOCK) void k_rk4_advance(DevModel m, DevData d, int i) {
  const long e = d.env0 + (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  rk4_advance(m, d, e, i);
}
// on-demand frames: xmat[9*nbody], geom_xpos[3*ngeom], geom_xmat[9*ngeom] from xpos/xquat
__global__ __launch_bounds__(BS_BLOCK) void k_frames(DevModel m, DevData d, double* xmat, double* gpos,
                                                     double* gmat) {
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  const long E = d.E;
  for (int b = 0; b < m.nbody; b++) {
    Q4 q = {AT(d.xquat, 4 * b), AT(d.xquat, 4 * b + 1), AT(d.xquat, 4 * b + 2), AT(d.xquat, 4 * b + 3)};
    M3 R = q2m(q);
    for (int k = 0; k < 9; k++) AT(xmat, 9 * b + k) = R.a[k];
  }
  for (int g = 0; g < m.ngeom; g++) {
    const int b = m.geom_bodyid[g];
    Q4 q = {AT(d.xquat, 4 * b), AT(d.xquat, 4 * b + 1), AT(d.xquat, 4 * b + 2), AT(d.xquat, 4 * b + 3)};
    M3 R = q2m(q);
    V3 p = mrot(R, ld3(m.geom_pos + 3 * g));
    AT(gpos, 3 * g) = AT(d.xpos, 3 * b) + p.x;
    AT(gpos, 3 * g + 1) = AT(d.xpos, 3 * b + 1) + p.y;
    AT(gpos, 3 * g + 2) = AT(d.xpos, 3 * b + 2) + p.z;
    M3 G = q2m(qmul(q, ld4(m.geom_quat + 4 * g)));
    for (int k = 0; k < 9; k++) AT(gmat, 9 * g + k) = G.a[k];
  }
}

// ---- contact pipeline (serial layout): position -> broadphase -> narrowphase -> dynamics
__global__ __launch_bounds__(BS_BLOCK) void k_c_pos(DevModel m, DevData d, ColData c, int check) {
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  if (check) {
    env_check(m, d, e, d.qpos, m.njnt, 0);
    env_check(m, d, e, d.qvel, m.nv, 1);
  }
  stage_position(m, d, e);
  stage_factor(m, d, e);
  col_geom_frames(m, d, c, e);
}
__global__ __launch_bounds__(64) void k_c_broad(ColModel cm, DevData d, ColData c) {
  const long e = blockIdx.x;  // one wavefront per environment
  if (e >= d.n_envs) return;
  col_broadphase(cm, d, c, e, threadIdx.x);
}
__global__ __launch_bounds__(BS_BLOCK) void k_c_dyn(DevModel m, ColModel cm, DevData d, ColData c, ConData k,
                                                    int integrate, int lds_hessian) {
  extern __shared__ double hs_lds[];  // dense Hessian + one Jacobian row per lane, [entry][64]
  const long e = (long)blockIdx.x * BS_BLOCK + threadIdx.x;
  if (e >= d.n_envs) return;
  col_assemble(cm, d, c, e);
  stage_velocity(m, d, e);
  stage_constraint_c(m, cm, d, c, k, e, lds_hessian ? hs_lds : nullptr);
  if (integrate) {
    // mj_checkAcc: a bad acceleration resets the environment; the next step recomputes everything
    if (!env_check(m, d, e, d.qacc, m.nv, 2)) stage_euler(m, d, e);
  }
}
// obs = qpos | qvel as fp32, reward = -|qpos[actuated joints] - ctrl|^2; lane = environment
// (coalesced reads of the [row][E] fields; rows of the output are written by their own lane)
__global__ __launch_bounds__(256) void k_pack_obs(DevModel m, DevData d, float* out, long stride, int nq) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x, E = d.E;
  if (e >= d.n_envs) return;
  float* o = out + e * stride;
  for (int i = 0; i < nq; i++) o[i] = (float)AT(d.qpos, i);
  for (int i = 0; i < m.nv; i++) o[nq + i] = (float)AT(d.qvel, i);
  double r = 0;
  for (int u = 0; u < m.nu; u++) {
    const double err = AT(d.qpos, m.jnt_qposadr[m.actuator_trnid[u]]) - AT(d.ctrl, u);
    r += err * err;
  }
  o[nq + m.nv] = (float)(-r);
}

// ---- learner glue around bs_step (mujoco_amd/env.py::HandVecEnv.step), one lane per environment
// before the steps: ctrl = action clamped to the actuators' ctrlrange (action: [n_envs][ld] doubles, row = environment)
__global__ __launch_bounds__(256) void k_env_pre(DevModel m, DevData d, const double* action, long ld) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x, E = d.E;
  if (e >= d.n_envs) return;
  for (int u = 0; u < m.nu; u++) {
    const double a = action[e * ld + u];
    AT(d.ctrl, u) = fmin(fmax(a, m.actuator_ctrlrange[2 * u]), m.actuator_ctrlrange[2 * u + 1]);
  }
}
// after the steps: reward, episode clock, done / failed / overflow flags, the observation of the finished
// episodes (terminal), reset of the finished environments (k_reset's arithmetic), the observation to act on
__global__ __launch_bounds__(256) void k_env_post(DevModel m, DevData d, const double* action, long ld, int* t, int episode_steps,
                                                  float* obs, float* term_obs, long stride, float* reward, uint8_t* flags, int nq) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x, E = d.E;
  if (e >= d.n_envs) return;
  double r = 0;
  for (int u = 0; u < m.nu; u++) {
    // against the clamped ACTION, not ctrl: a mj_check* reset during the period has zeroed ctrl
    const double a = fmin(fmax(action[e * ld + u], m.actuator_ctrlrange[2 * u]), m.actuator_ctrlrange[2 * u + 1]);
    const double err = AT(d.qpos, m.jnt_qposadr[m.actuator_trnid[u]]) - a;
    r += err * err;
  }
  reward[e] = (float)(-r);
  const int w = AT(d.warning, 0), tt = t[e] + 1;
  const bool failed = (w & 7) != 0, overflow = (w & 8) != 0, done = failed || tt >= episode_steps;
  flags[e] = (uint8_t)((done ? 1 : 0) | (failed ? 2 : 0) | (overflow ? 4 : 0));
  float* to = term_obs + e * stride;
  float* o = obs + e * stride;
  for (int i = 0; i < nq; i++) to[i] = (float)AT(d.qpos, i);
  for (int i = 0; i < m.nv; i++) to[nq + i] = (float)AT(d.qvel, i);
  if (done) {
    env_reset(m, d, e);
    AT(d.warning, 0) = 0;
    t[e] = 0;
    for (int i = 0; i < nq; i++) o[i] = (float)AT(d.qpos, i);
    for (int i = 0; i < m.nv; i++) o[nq + i] = 0.0f;
  } else {
    AT(d.warning, 0) = w & 7;
    t[e] = tt;
    for (int i = 0; i < nq + m.nv; i++) o[i] = to[i];
  }
}

// ---- launchers
void bs_launch_env_pre(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const double* action, long ld) {
  hipLaunchKernelGGL(k_env_pre, g, dim3(256), 0, st, m, d, action, ld);
}
void bs_launch_env_post(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const double* action, long ld, int* t,
                        int episode_steps, float* obs, float* term_obs, long stride, float* reward, uint8_t* flags, int nq) {
  hipLaunchKernelGGL(k_env_post, g, dim3(256), 0, st, m, d, action, ld, t, episode_steps, obs, term_obs, stride, reward, flags, nq);
}
void bs_launch_step(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, int nsteps) {
  hipLaunchKernelGGL(k_step, g, dim3(BS_BLOCK), 0, st, m, d, nsteps);
}
void bs_launch_forward(dim3 g, hipStream_t st, const DevModel& m, const DevData& d) {
  hipLaunchKernelGGL(k_forward, g, dim3(BS_BLOCK), 0, st, m, d);
}
void bs_launch_stage(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, int stage) {
  hipLaunchKernelGGL(k_stage, g, dim3(BS_BLOCK), 0, st, m, d, stage);
}
void bs_launch_reset(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const uint8_t* mask) {
  hipLaunchKernelGGL(k_reset, g, dim3(BS_BLOCK), 0, st, m, d, mask);
}
void bs_launch_check_state(dim3 g, hipStream_t st, const DevModel& m, const DevData& d) {
  hipLaunchKernelGGL(k_check_state, g, dim3(BS_BLOCK), 0, st, m, d);
}
void bs_launch_rk4_advance(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, int i) {
  hipLaunchKernelGGL(k_rk4_advance, g, dim3(BS_BLOCK), 0, st, m, d, i);
}
void bs_launch_frames(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, double* xmat, double* gpos,
                      double* gmat) {
  hipLaunchKernelGGL(k_frames, g, dim3(BS_BLOCK), 0, st, m, d, xmat, gpos, gmat);
}
void bs_launch_pack_obs(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, float* out, long stride, int nq) {
  hipLaunchKernelGGL(k_pack_obs, g, dim3(256), 0, st, m, d, out, stride, nq);
}
void bs_launch_c_pos(dim3 g, hipStream_t st, const DevModel& m, const DevData& d, const ColData& c, int check) {
  hipLaunchKernelGGL(k_c_pos, g, dim3(BS_BLOCK), 0, st, m, d, c, check);
}
void bs_launch_c_broad(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c) {
  hipLaunchKernelGGL(k_c_broad, g, dim3(64), 0, st, cm, d, c);
}
void bs_launch_c_dyn(dim3 g, size_t lds, hipStream_t st, const DevModel& m, const ColModel& cm, const DevData& d,
                     const ColData& c, const ConData& k, int integrate, int lds_hessian) {
  hipLaunchKernelGGL(k_c_dyn, g, dim3(BS_BLOCK), lds, st, m, cm, d, c, k, integrate, lds_hessian);
}
hipError_t bs_c_dyn_set_lds(int bytes) {
  return hipFuncSetAttribute((const void*)k_c_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
