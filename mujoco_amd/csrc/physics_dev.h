// physics_dev.h — device-side physics for libbatchsim.so (gfx950 / MI355X).
//
// One environment per wavefront lane; every per-environment array is [row][E] (env-minor),
// so lane e touches element row*E + e and a wave's access is one contiguous 512-byte run.
// Model arrays are wave-uniform and come in through scalar loads.  All arithmetic is fp64.
//
// Stage names follow the MuJoCo routine each one replaces; the
// reference lines that fix each stage's inputs are cited per stage.  The constraint rows have
// a FIXED layout per model (equalities | dof friction | 2 limit rows per limited joint), with
// inactive limit rows carrying D = 0, so all lanes of a wave run the same row loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BS_MINVAL 1e-15
#define BS_MAXVAL 1e10
#define BS_MINIMP 0.0001
#define BS_MAXIMP 0.9999

enum { BS_JNT_SLIDE = 2, BS_JNT_HINGE = 3 };
enum {
  BS_DSBL_CONSTRAINT = 1 << 0, BS_DSBL_EQUALITY = 1 << 1, BS_DSBL_FRICTIONLOSS = 1 << 2,
  BS_DSBL_LIMIT = 1 << 3, BS_DSBL_CONTACT = 1 << 4, BS_DSBL_PASSIVE = 1 << 5,
  BS_DSBL_GRAVITY = 1 << 6, BS_DSBL_CLAMPCTRL = 1 << 7, BS_DSBL_WARMSTART = 1 << 8,
  BS_DSBL_ACTUATION = 1 << 10, BS_DSBL_REFSAFE = 1 << 11
};

struct DevModel {
  int nbody, njnt, nv, nu, ngeom, neq, nM, npair;
  int nfric, nlim, nrow;  // fixed constraint-row layout: neq | nfric | 2*nlim
  int integrator, solver, iterations, ls_iterations, disableflags, damped, hess_sparse;
  double timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  const int *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  const double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_subtreemass,
      *body_inertia, *body_invweight0;
  const int *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *jnt_margin, *jnt_stiffness, *jnt_solref,
      *jnt_solimp, *qpos0, *qpos_spring;
  const int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_Madr;
  const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_solref, *dof_solimp,
      *dof_invweight0;
  const int *geom_bodyid;
  const double *geom_pos, *geom_quat;
  const int *actuator_trnid, *actuator_ctrllimited, *actuator_forcelimited;
  const double *actuator_ctrlrange, *actuator_forcerange, *actuator_gear, *actuator_gainprm,
      *actuator_biasprm;
  const int *eq_dof1, *eq_dof2, *eq_jnt1, *eq_jnt2, *eq_Hadr;  // active equalities only
  const double *eq_data, *eq_solref, *eq_solimp;
  const int *fric_dof, *lim_jnt;
};

struct DevData {
  long E;  // padded environment count = row stride
  long n_envs;  // end of the environment range a launch covers
  long env0;    // its start (tree-parallel contact kernels only: two halves of a batch on two streams)
  int range;    // index of that range (selects its narrowphase item counter)
  double *qpos, *qvel, *ctrl, *qfrc_applied, *qacc, *qacc_warmstart, *time;
  double *xpos, *xquat, *xipos, *xanchor, *xaxis, *rootcom, *cinert, *crb, *cdof, *qM, *qLD,
      *qLDiagInv;
  double *cvel, *cacc, *cfrc, *qfrc_bias, *qfrc_passive, *actuator_force, *qfrc_actuator,
      *qfrc_smooth, *qacc_smooth, *qfrc_constraint;
  double *efc_D, *efc_aref, *efc_force, *jar, *jv, *eq_coef, *fric_rf;
  double *Ma, *grad, *Mgrad, *search, *Mv, *H, *Hdinv;
  double *rk_q0, *rk_v0, *rk_Fv, *rk_Fa;
  double *rk_t0;  // [2][E]: start time of the RK4 step; 1.0 while the step is abandoned (mj_checkAcc reset)
  double *pgs_W;  // PGS only: rows of J M^-1 for the fixed rows, [(row*nv + dof)][E]
  int *solver_iter, *warning, *nefc;
};

#define AT(p, i) (p)[(size_t)(i) * (size_t)E + (size_t)e]

// ------------------------------------------------------------------ small math (registers)
struct V3 { double x, y, z; };
struct Q4 { double w, x, y, z; };
struct M3 { double a[9]; };

__host__ __device__ __forceinline__ Q4 qmul(const Q4& a, const Q4& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__host__ __device__ __forceinline__ M3 q2m(const Q4& q) {
  double q00 = q.w * q.w, q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
  double q11 = q.x * q.x, q12 = q.x * q.y, q13 = q.x * q.z, q22 = q.y * q.y, q23 = q.y * q.z,
         q33 = q.z * q.z;
  M3 R;
  R.a[0] = q00 + q11 - q22 - q33; R.a[4] = q00 - q11 + q22 - q33; R.a[8] = q00 - q11 - q22 + q33;
  R.a[1] = 2 * (q12 - q03); R.a[2] = 2 * (q13 + q02);
  R.a[3] = 2 * (q12 + q03); R.a[5] = 2 * (q23 - q01);
  R.a[6] = 2 * (q13 - q02); R.a[7] = 2 * (q23 + q01);
  return R;
}
__host__ __device__ __forceinline__ V3 mrot(const M3& R, const V3& v) {
  return {R.a[0] * v.x + R.a[1] * v.y + R.a[2] * v.z, R.a[3] * v.x + R.a[4] * v.y + R.a[5] * v.z,
          R.a[6] * v.x + R.a[7] * v.y + R.a[8] * v.z};
}
__host__ __device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ Q4 ld4(const double* p) { return {p[0], p[1], p[2], p[3]}; }

__device__ __forceinline__ void mul_inert_vec(double* r, const double* I, const double* v) {
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
__device__ __forceinline__ void cross_motion(double* r, const double* vel, const double* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void cross_force(double* r, const double* vel, const double* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

// ------------------------------------------------------------------ A14 reset / checks
__device__ inline void env_reset(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  for (int j = 0; j < m.njnt; j++) AT(d.qpos, j) = m.qpos0[j];
  for (int i = 0; i < m.nv; i++) {
    AT(d.qvel, i) = 0; AT(d.qacc, i) = 0; AT(d.qacc_warmstart, i) = 0; AT(d.qfrc_applied, i) = 0;
  }
  for (int u = 0; u < m.nu; u++) AT(d.ctrl, u) = 0;
  AT(d.time, 0) = 0;
  AT(d.solver_iter, 0) = 0;
  AT(d.nefc, 0) = 0;
}
__device__ __forceinline__ bool bad_num(double x) { return !(x <= BS_MAXVAL && x >= -BS_MAXVAL); }
__device__ inline bool env_check(const DevModel& m, const DevData& d, long e, const double* x, int n,
                                 int bit) {
  const long E = d.E;
  bool bad = false;
  for (int i = 0; i < n; i++) bad |= bad_num(AT(x, i));
  if (bad) {
    int w = AT(d.warning, 0) | (1 << bit);
    env_reset(m, d, e);
    AT(d.warning, 0) = w;
  }
  return bad;
}

// ------------------------------------------------------------------ A1-A4 position stage
// mj_kinematics + mj_comPos + mj_crb + mj_factorM, one tree walk down and one back up.
// Inputs fixed by body pos/quat and joint pos/axis attributes (reference
// output/DexterousDynamos.xml:29-111; src/Mujoco_XML.py:212,255), mesh-derived inertias
// (xml:7,10-25) and armature (src/Mujoco_XML.py:71).
__device__ inline void stage_position(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  for (int k = 0; k < 3; k++) { AT(d.xpos, k) = 0; AT(d.xipos, k) = 0; }
  AT(d.xquat, 0) = 1; AT(d.xquat, 1) = 0; AT(d.xquat, 2) = 0; AT(d.xquat, 3) = 0;
  for (int b = 0; b < m.nbody; b++)
    if (m.body_rootid[b] == b)
      for (int k = 0; k < 3; k++) AT(d.rootcom, 3 * b + k) = 0;

  for (int b = 1; b < m.nbody; b++) {
    const int p = m.body_parentid[b];
    V3 pp = {AT(d.xpos, 3 * p), AT(d.xpos, 3 * p + 1), AT(d.xpos, 3 * p + 2)};
    Q4 pq = {AT(d.xquat, 4 * p), AT(d.xquat, 4 * p + 1), AT(d.xquat, 4 * p + 2), AT(d.xquat, 4 * p + 3)};
    M3 R = q2m(pq);
    V3 t = mrot(R, ld3(m.body_pos + 3 * b));
    V3 pos = {pp.x + t.x, pp.y + t.y, pp.z + t.z};
    Q4 quat = qmul(pq, ld4(m.body_quat + 4 * b));
    const int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
    for (int jj = 0; jj < jn; jj++) {
      const int j = ja + jj;
      M3 Rq = q2m(quat);
      V3 jp = ld3(m.jnt_pos + 3 * j), jax = ld3(m.jnt_axis + 3 * j);
      V3 a = mrot(Rq, jp);
      V3 anchor = {a.x + pos.x, a.y + pos.y, a.z + pos.z};
      V3 axis = mrot(Rq, jax);
      AT(d.xanchor, 3 * j) = anchor.x; AT(d.xanchor, 3 * j + 1) = anchor.y; AT(d.xanchor, 3 * j + 2) = anchor.z;
      AT(d.xaxis, 3 * j) = axis.x; AT(d.xaxis, 3 * j + 1) = axis.y; AT(d.xaxis, 3 * j + 2) = axis.z;
      const double dq = AT(d.qpos, m.jnt_qposadr[j]) - m.qpos0[j];
      if (m.jnt_type[j] == BS_JNT_SLIDE) {
        pos.x += axis.x * dq; pos.y += axis.y * dq; pos.z += axis.z * dq;
      } else {
        double s, c;
        sincos(0.5 * dq, &s, &c);
        quat = qmul(quat, Q4{c, jax.x * s, jax.y * s, jax.z * s});
        V3 v = mrot(q2m(quat), jp);
        pos = {anchor.x - v.x, anchor.y - v.y, anchor.z - v.z};
      }
    }
    double n = sqrt(quat.w * quat.w + quat.x * quat.x + quat.y * quat.y + quat.z * quat.z);
    if (n < BS_MINVAL) quat = {1, 0, 0, 0};
    else { quat.w /= n; quat.x /= n; quat.y /= n; quat.z /= n; }
    AT(d.xpos, 3 * b) = pos.x; AT(d.xpos, 3 * b + 1) = pos.y; AT(d.xpos, 3 * b + 2) = pos.z;
    AT(d.xquat, 4 * b) = quat.w; AT(d.xquat, 4 * b + 1) = quat.x; AT(d.xquat, 4 * b + 2) = quat.y;
    AT(d.xquat, 4 * b + 3) = quat.z;
    V3 ip = mrot(q2m(quat), ld3(m.body_ipos + 3 * b));
    V3 xi = {pos.x + ip.x, pos.y + ip.y, pos.z + ip.z};
    AT(d.xipos, 3 * b) = xi.x; AT(d.xipos, 3 * b + 1) = xi.y; AT(d.xipos, 3 * b + 2) = xi.z;
    const double mass = m.body_mass[b];
    const int r = m.body_rootid[b];
    AT(d.rootcom, 3 * r) += mass * xi.x;
    AT(d.rootcom, 3 * r + 1) += mass * xi.y;
    AT(d.rootcom, 3 * r + 2) += mass * xi.z;
  }
  for (int b = 1; b < m.nbody; b++)
    if (m.body_rootid[b] == b) {
      const double sm = m.body_subtreemass[b];
      for (int k = 0; k < 3; k++) {
        if (sm < BS_MINVAL) AT(d.rootcom, 3 * b + k) = AT(d.xipos, 3 * b + k);
        else AT(d.rootcom, 3 * b + k) /= sm;
      }
    }

  // cinert (body inertia about its tree's COM, world axes) and cdof (motion axes)
  for (int k = 0; k < 10; k++) { AT(d.cinert, k) = 0; AT(d.crb, k) = 0; }
  for (int b = 1; b < m.nbody; b++) {
    const int r = m.body_rootid[b];
    Q4 q = {AT(d.xquat, 4 * b), AT(d.xquat, 4 * b + 1), AT(d.xquat, 4 * b + 2), AT(d.xquat, 4 * b + 3)};
    M3 R = q2m(qmul(q, ld4(m.body_iquat + 4 * b)));
    const double* I = m.body_inertia + 3 * b;
    const double mass = m.body_mass[b];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = AT(d.xipos, 3 * b + k) - AT(d.rootcom, 3 * r + k);
    double t[9], ci[10];
    for (int rr = 0; rr < 3; rr++)
      for (int k = 0; k < 3; k++) t[3 * rr + k] = R.a[3 * rr + k] * I[k];
    ci[0] = t[0] * R.a[0] + t[1] * R.a[1] + t[2] * R.a[2];
    ci[1] = t[3] * R.a[3] + t[4] * R.a[4] + t[5] * R.a[5];
    ci[2] = t[6] * R.a[6] + t[7] * R.a[7] + t[8] * R.a[8];
    ci[3] = t[0] * R.a[3] + t[1] * R.a[4] + t[2] * R.a[5];
    ci[4] = t[0] * R.a[6] + t[1] * R.a[7] + t[2] * R.a[8];
    ci[5] = t[3] * R.a[6] + t[4] * R.a[7] + t[5] * R.a[8];
    ci[0] += mass * (off[1] * off[1] + off[2] * off[2]);
    ci[1] += mass * (off[0] * off[0] + off[2] * off[2]);
    ci[2] += mass * (off[0] * off[0] + off[1] * off[1]);
    ci[3] -= mass * off[0] * off[1];
    ci[4] -= mass * off[0] * off[2];
    ci[5] -= mass * off[1] * off[2];
    ci[6] = mass * off[0]; ci[7] = mass * off[1]; ci[8] = mass * off[2]; ci[9] = mass;
    for (int k = 0; k < 10; k++) { AT(d.cinert, 10 * b + k) = ci[k]; AT(d.crb, 10 * b + k) = ci[k]; }
  }
  for (int j = 0; j < m.njnt; j++) {
    const int r = m.body_rootid[m.jnt_bodyid[j]], i = m.jnt_dofadr[j];
    V3 axis = {AT(d.xaxis, 3 * j), AT(d.xaxis, 3 * j + 1), AT(d.xaxis, 3 * j + 2)};
    if (m.jnt_type[j] == BS_JNT_SLIDE) {
      AT(d.cdof, 6 * i) = 0; AT(d.cdof, 6 * i + 1) = 0; AT(d.cdof, 6 * i + 2) = 0;
      AT(d.cdof, 6 * i + 3) = axis.x; AT(d.cdof, 6 * i + 4) = axis.y; AT(d.cdof, 6 * i + 5) = axis.z;
    } else {
      V3 off = {AT(d.rootcom, 3 * r) - AT(d.xanchor, 3 * j), AT(d.rootcom, 3 * r + 1) - AT(d.xanchor, 3 * j + 1),
                AT(d.rootcom, 3 * r + 2) - AT(d.xanchor, 3 * j + 2)};
      V3 c = cross(axis, off);
      AT(d.cdof, 6 * i) = axis.x; AT(d.cdof, 6 * i + 1) = axis.y; AT(d.cdof, 6 * i + 2) = axis.z;
      AT(d.cdof, 6 * i + 3) = c.x; AT(d.cdof, 6 * i + 4) = c.y; AT(d.cdof, 6 * i + 5) = c.z;
    }
  }

  // composite inertia + mass matrix, leaves to root: crb[b] is final once all children
  // (higher ids) have been folded in, so the M rows of b's dofs are produced on the way up.
  for (int b = m.nbody - 1; b > 0; b--) {
    double cb[10];
    for (int k = 0; k < 10; k++) cb[k] = AT(d.crb, 10 * b + k);
    const int p = m.body_parentid[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) AT(d.crb, 10 * p + k) += cb[k];
    const int dn = m.body_dofnum[b], da = m.body_dofadr[b];
    for (int ii = 0; ii < dn; ii++) {
      const int i = da + ii;
      double cd[6], buf[6];
      for (int k = 0; k < 6; k++) cd[k] = AT(d.cdof, 6 * i + k);
      mul_inert_vec(buf, cb, cd);
      int adr = m.dof_Madr[i];
      double s = 0;
      for (int k = 0; k < 6; k++) s += cd[k] * buf[k];
      AT(d.qM, adr) = m.dof_armature[i] + s;
      adr++;
      for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j], adr++) {
        double sj = 0;
        for (int k = 0; k < 6; k++) sj += AT(d.cdof, 6 * j + k) * buf[k];
        AT(d.qM, adr) = sj;
      }
    }
  }
}

// sparse L^T D L factorisation / solve on the dof tree (mj_factorM / mj_solveM layout)
__device__ inline void factor_sparse(const DevModel& m, double* LD, double* dinv, long E, long e) {
  for (int k = m.nv - 1; k >= 0; k--) {
    const int akk = m.dof_Madr[k];
    const double dkk = AT(LD, akk);
    int aki = akk + 1;
    for (int i = m.dof_parentid[k]; i >= 0; i = m.dof_parentid[i], aki++) {
      const double tmp = AT(LD, aki) / dkk;
      int aij = m.dof_Madr[i], akj = aki;
      for (int j = i; j >= 0; j = m.dof_parentid[j], aij++, akj++) AT(LD, aij) -= tmp * AT(LD, akj);
      AT(LD, aki) = tmp;
    }
    AT(dinv, k) = 1.0 / dkk;
  }
}
__device__ inline void solve_sparse(const DevModel& m, const double* LD, const double* dinv, double* x,
                                    long E, long e) {
  for (int i = m.nv - 1; i >= 0; i--) {
    const double xi = AT(x, i);
    int adr = m.dof_Madr[i] + 1;
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j], adr++) AT(x, j) -= AT(LD, adr) * xi;
  }
  for (int i = 0; i < m.nv; i++) AT(x, i) *= AT(dinv, i);
  for (int i = 0; i < m.nv; i++) {
    double xi = AT(x, i);
    int adr = m.dof_Madr[i] + 1;
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j], adr++) xi -= AT(LD, adr) * AT(x, j);
    AT(x, i) = xi;
  }
}
__device__ inline void mul_m(const DevModel& m, const double* qM, double* res, const double* v, long E,
                             long e) {
  for (int i = 0; i < m.nv; i++) AT(res, i) = 0;
  for (int i = 0; i < m.nv; i++) {
    int adr = m.dof_Madr[i];
    const double vi = AT(v, i);
    double ri = AT(res, i) + AT(qM, adr) * vi;
    adr++;
    for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j], adr++) {
      const double mij = AT(qM, adr);
      ri += mij * AT(v, j);
      AT(res, j) += mij * vi;
    }
    AT(res, i) = ri;
  }
}
__device__ inline void stage_factor(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  for (int k = 0; k < m.nM; k++) AT(d.qLD, k) = AT(d.qM, k);
  factor_sparse(m, d.qLD, d.qLDiagInv, E, e);
}

// ------------------------------------------------------------------ A8-A10 velocity / force stage
// mj_comVel + mj_passive + mj_rne(flg_acc=0) + mj_fwdActuation + mj_fwdAcceleration.
// damping: reference src/Mujoco_XML.py:71; position servos kp/forcerange/ctrlrange:
// src/Mujoco_XML.py:72,261, src/Fusion_to_Mujoco.py:110, xml:6,119-135.
__device__ inline void stage_velocity(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  for (int k = 0; k < 6; k++) { AT(d.cvel, k) = 0; AT(d.cfrc, k) = 0; }
  AT(d.cacc, 0) = 0; AT(d.cacc, 1) = 0; AT(d.cacc, 2) = 0;
  const bool grav = !(m.disableflags & BS_DSBL_GRAVITY);
  for (int k = 0; k < 3; k++) AT(d.cacc, 3 + k) = grav ? -m.gravity[k] : 0.0;
  for (int b = 1; b < m.nbody; b++) {
    const int p = m.body_parentid[b];
    double v[6], a[6], ci[10];
    for (int k = 0; k < 6; k++) { v[k] = AT(d.cvel, 6 * p + k); a[k] = AT(d.cacc, 6 * p + k); }
    const int dn = m.body_dofnum[b], da = m.body_dofadr[b];
    for (int ii = 0; ii < dn; ii++) {
      const int i = da + ii;
      double cd[6], cdd[6];
      for (int k = 0; k < 6; k++) cd[k] = AT(d.cdof, 6 * i + k);
      cross_motion(cdd, v, cd);
      const double qv = AT(d.qvel, i);
      for (int k = 0; k < 6; k++) { a[k] += cdd[k] * qv; v[k] += cd[k] * qv; }
    }
    for (int k = 0; k < 6; k++) { AT(d.cvel, 6 * b + k) = v[k]; AT(d.cacc, 6 * b + k) = a[k]; }
    for (int k = 0; k < 10; k++) ci[k] = AT(d.cinert, 10 * b + k);
    double t0[6], t1[6], t2[6];
    mul_inert_vec(t0, ci, a);
    mul_inert_vec(t1, ci, v);
    cross_force(t2, v, t1);
    for (int k = 0; k < 6; k++) AT(d.cfrc, 6 * b + k) = t0[k] + t2[k];
  }
  for (int b = m.nbody - 1; b > 0; b--) {
    double f[6];
    for (int k = 0; k < 6; k++) f[k] = AT(d.cfrc, 6 * b + k);
    const int p = m.body_parentid[b];
    if (p > 0)
      for (int k = 0; k < 6; k++) AT(d.cfrc, 6 * p + k) += f[k];
    const int dn = m.body_dofnum[b], da = m.body_dofadr[b];
    for (int ii = 0; ii < dn; ii++) {
      const int i = da + ii;
      double s = 0;
      for (int k = 0; k < 6; k++) s += AT(d.cdof, 6 * i + k) * f[k];
      AT(d.qfrc_bias, i) = s;
    }
  }
  const bool passive = !(m.disableflags & BS_DSBL_PASSIVE);
  for (int i = 0; i < m.nv; i++) {
    double f = 0;
    if (passive) {
      const int j = m.dof_jntid[i];
      f = -m.jnt_stiffness[j] * (AT(d.qpos, m.jnt_qposadr[j]) - m.qpos_spring[j]) -
          m.dof_damping[i] * AT(d.qvel, i);
    }
    AT(d.qfrc_passive, i) = f;
    AT(d.qfrc_actuator, i) = 0;
  }
  const bool act = !(m.disableflags & BS_DSBL_ACTUATION);
  for (int u = 0; u < m.nu; u++) {
    double f = 0;
    if (act) {
      const int j = m.actuator_trnid[u];
      const double gear = m.actuator_gear[u];
      const double len = gear * AT(d.qpos, m.jnt_qposadr[j]), vel = gear * AT(d.qvel, m.jnt_dofadr[j]);
      double c = AT(d.ctrl, u);
      if (m.actuator_ctrllimited[u] && !(m.disableflags & BS_DSBL_CLAMPCTRL)) {
        c = fmax(c, m.actuator_ctrlrange[2 * u]);
        c = fmin(c, m.actuator_ctrlrange[2 * u + 1]);
      }
      const double* bp = m.actuator_biasprm + 3 * u;
      f = m.actuator_gainprm[3 * u] * c + bp[0] + bp[1] * len + bp[2] * vel;
      if (m.actuator_forcelimited[u]) {
        f = fmax(f, m.actuator_forcerange[2 * u]);
        f = fmin(f, m.actuator_forcerange[2 * u + 1]);
      }
      AT(d.qfrc_actuator, m.jnt_dofadr[j]) += gear * f;
    }
    AT(d.actuator_force, u) = f;
  }
  for (int i = 0; i < m.nv; i++) {
    const double f = AT(d.qfrc_passive, i) - AT(d.qfrc_bias, i) + AT(d.qfrc_applied, i) + AT(d.qfrc_actuator, i);
    AT(d.qfrc_smooth, i) = f;
    AT(d.qacc_smooth, i) = f;
  }
  solve_sparse(m, d.qLD, d.qLDiagInv, d.qacc_smooth, E, e);
}

// ------------------------------------------------------------------ A7 constraint rows
__device__ inline void impedance(const double* solimp, double pos, double margin, double& imp) {
  double dmin = fmin(fmax(solimp[0], BS_MINIMP), BS_MAXIMP), dmax = fmin(fmax(solimp[1], BS_MINIMP), BS_MAXIMP);
  const double width = fmax(solimp[2], BS_MINVAL), mid = fmin(fmax(solimp[3], BS_MINIMP), BS_MAXIMP);
  const double power = fmax(solimp[4], 1.0);
  const double x = fabs(pos - margin) / width;
  if (dmin == dmax || x >= 1.0) { imp = (x >= 1.0) ? dmax : 0.5 * (dmin + dmax); return; }
  if (x <= 0) { imp = dmin; return; }
  double y;
  if (power == 1.0) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  imp = dmin + y * (dmax - dmin);
}
__host__ __device__ inline void kb_from_solref(const DevModel& m, const double* solref, const double* solimp,
                                               double& K, double& B) {
  const double dmax = fmin(fmax(solimp[1], BS_MINIMP), BS_MAXIMP);
  if (solref[0] > 0) {
    double tc = solref[0];
    const double dr = solref[1];
    if (!(m.disableflags & BS_DSBL_REFSAFE)) tc = fmax(tc, 2 * m.timestep);
    K = 1.0 / fmax(BS_MINVAL, dmax * dmax * tc * tc * dr * dr);
    B = 2.0 / fmax(BS_MINVAL, dmax * tc);
  } else {
    K = -solref[0] / fmax(BS_MINVAL, dmax * dmax);
    B = -solref[1] / fmax(BS_MINVAL, dmax);
  }
}

// mj_makeConstraint + mj_makeImpedance + mj_referenceConstraint on the fixed row layout.
// equality coefficient 70/120: reference src/Fusion_to_Mujoco.py:107-108, src/Mujoco_XML.py:274;
// frictionloss 0.001 / limit margin 0.01: src/Mujoco_XML.py:71.  Returns the active row count.
__device__ inline int make_rows(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  int nact = 0;
  for (int r = 0; r < m.neq; r++) {
    const int j1 = m.eq_jnt1[r], j2 = m.eq_jnt2[r];
    const double* c = m.eq_data + 5 * r;
    const double p1 = AT(d.qpos, m.jnt_qposadr[j1]) - m.qpos0[j1];
    double pos, deriv = 0, diag = m.dof_invweight0[m.eq_dof1[r]], vel = AT(d.qvel, m.eq_dof1[r]);
    if (j2 >= 0) {
      const double x = AT(d.qpos, m.jnt_qposadr[j2]) - m.qpos0[j2];
      pos = p1 - (c[0] + x * (c[1] + x * (c[2] + x * (c[3] + x * c[4]))));
      deriv = c[1] + x * (2 * c[2] + x * (3 * c[3] + x * 4 * c[4]));
      diag += m.dof_invweight0[m.eq_dof2[r]];
      vel -= deriv * AT(d.qvel, m.eq_dof2[r]);
    } else pos = p1 - c[0];
    double imp, K, B;
    impedance(m.eq_solimp + 5 * r, pos, 0.0, imp);
    kb_from_solref(m, m.eq_solref + 2 * r, m.eq_solimp + 5 * r, K, B);
    const double R = fmax(BS_MINVAL, (1 - imp) * diag / imp);
    AT(d.eq_coef, r) = deriv;
    AT(d.efc_D, r) = 1.0 / R;
    AT(d.efc_aref, r) = -B * vel - K * imp * pos;
    nact++;
  }
  for (int k = 0; k < m.nfric; k++) {
    const int r = m.neq + k, i = m.fric_dof[k];
    double imp, K, B;
    impedance(m.dof_solimp + 5 * i, 0.0, 0.0, imp);
    kb_from_solref(m, m.dof_solref + 2 * i, m.dof_solimp + 5 * i, K, B);
    const double R = fmax(BS_MINVAL, (1 - imp) * m.dof_invweight0[i] / imp);
    AT(d.efc_D, r) = 1.0 / R;
    AT(d.fric_rf, k) = R * m.dof_frictionloss[i];
    AT(d.efc_aref, r) = -B * AT(d.qvel, i);
    nact++;
  }
  for (int k = 0; k < m.nlim; k++) {
    const int j = m.lim_jnt[k], i = m.jnt_dofadr[j];
    const double v = AT(d.qpos, m.jnt_qposadr[j]), margin = m.jnt_margin[j], qv = AT(d.qvel, i);
    double K, B;
    kb_from_solref(m, m.jnt_solref + 2 * j, m.jnt_solimp + 5 * j, K, B);
    for (int s = 0; s < 2; s++) {
      const int r = m.neq + m.nfric + 2 * k + s;
      const double dist = s ? (m.jnt_range[2 * j + 1] - v) : (v - m.jnt_range[2 * j]);
      double D = 0, aref = 0;
      if (dist < margin) {
        double imp;
        impedance(m.jnt_solimp + 5 * j, dist, margin, imp);
        const double R = fmax(BS_MINVAL, (1 - imp) * m.dof_invweight0[i] / imp);
        D = 1.0 / R;
        const double vel = s ? -qv : qv;
        aref = -B * vel - K * imp * (dist - margin);
        nact++;
      }
      AT(d.efc_D, r) = D;
      AT(d.efc_aref, r) = aref;
    }
  }
  return nact;
}

// jar = J*a - aref for the fixed rows
__device__ inline void rows_residual(const DevModel& m, const DevData& d, long e, const double* a,
                                     double* out, bool subtract_aref) {
  const long E = d.E;
  for (int r = 0; r < m.neq; r++) {
    double s = AT(a, m.eq_dof1[r]);
    if (m.eq_dof2[r] >= 0) s -= AT(d.eq_coef, r) * AT(a, m.eq_dof2[r]);
    AT(out, r) = subtract_aref ? s - AT(d.efc_aref, r) : s;
  }
  for (int k = 0; k < m.nfric; k++) {
    const int r = m.neq + k;
    const double s = AT(a, m.fric_dof[k]);
    AT(out, r) = subtract_aref ? s - AT(d.efc_aref, r) : s;
  }
  for (int k = 0; k < m.nlim; k++) {
    const double s = AT(a, m.jnt_dofadr[m.lim_jnt[k]]);
    const int r = m.neq + m.nfric + 2 * k;
    AT(out, r) = subtract_aref ? s - AT(d.efc_aref, r) : s;
    AT(out, r + 1) = subtract_aref ? -s - AT(d.efc_aref, r + 1) : -s;
  }
}

// cost of the rows at residual `jar`; optionally writes efc_force and accumulates
// qfrc_constraint = J^T force
template <bool WRITE>
__device__ inline double rows_update(const DevModel& m, const DevData& d, long e, const double* jar) {
  const long E = d.E;
  double cost = 0;
  if (WRITE)
    for (int i = 0; i < m.nv; i++) AT(d.qfrc_constraint, i) = 0;
  for (int r = 0; r < m.neq; r++) {
    const double x = AT(jar, r), D = AT(d.efc_D, r), f = -D * x;
    cost += 0.5 * D * x * x;
    if (WRITE) {
      AT(d.efc_force, r) = f;
      AT(d.qfrc_constraint, m.eq_dof1[r]) += f;
      if (m.eq_dof2[r] >= 0) AT(d.qfrc_constraint, m.eq_dof2[r]) -= AT(d.eq_coef, r) * f;
    }
  }
  for (int k = 0; k < m.nfric; k++) {
    const int r = m.neq + k, i = m.fric_dof[k];
    const double x = AT(jar, r), D = AT(d.efc_D, r), rf = AT(d.fric_rf, k), fl = m.dof_frictionloss[i];
    double f;
    if (x <= -rf) { f = fl; cost += -fl * (0.5 * rf + x); }
    else if (x >= rf) { f = -fl; cost += -fl * (0.5 * rf - x); }
    else { f = -D * x; cost += 0.5 * D * x * x; }
    if (WRITE) { AT(d.efc_force, r) = f; AT(d.qfrc_constraint, i) += f; }
  }
  for (int k = 0; k < m.nlim; k++) {
    const int i = m.jnt_dofadr[m.lim_jnt[k]];
    for (int s = 0; s < 2; s++) {
      const int r = m.neq + m.nfric + 2 * k + s;
      const double x = AT(jar, r), D = AT(d.efc_D, r);
      double f = 0;
      if (x < 0) { f = -D * x; cost += 0.5 * D * x * x; }
      if (WRITE) { AT(d.efc_force, r) = f; AT(d.qfrc_constraint, i) += s ? -f : f; }
    }
  }
  return cost;
}

// Gauss + constraint cost at acceleration `a`; leaves Ma and jar filled for `a`
__device__ inline double cost_at(const DevModel& m, const DevData& d, long e, const double* a) {
  const long E = d.E;
  mul_m(m, d.qM, d.Ma, a, E, e);
  rows_residual(m, d, e, a, d.jar, true);
  double g = 0;
  for (int i = 0; i < m.nv; i++) g += (AT(d.Ma, i) - AT(d.qfrc_smooth, i)) * (AT(a, i) - AT(d.qacc_smooth, i));
  return 0.5 * g + rows_update<false>(m, d, e, d.jar);
}

// grad = Ma - qfrc_smooth - qfrc_constraint;  H = M + J^T D_active J on the dof-tree sparsity
// (every fixed row touches dofs on one root path), factorised like M;  search = -H^-1 grad
// mj_solCG's direction: Mgrad = M^-1 grad (the preconditioner is M), Polak-Ribiere beta clamped
// at 0 in the M^-1 inner product; shares the primal loop, line search and stopping rule with
// Newton.  d.Mv holds the previous Mgrad while the new one is formed (it is free outside the
// line search).
__device__ inline double cg_gradient(const DevModel& m, const DevData& d, long e, bool first) {
  const long E = d.E;
  double gg = 0, gg_old = 0;
  for (int i = 0; i < m.nv; i++) {
    if (!first) { gg_old += AT(d.grad, i) * AT(d.Mgrad, i); AT(d.Mv, i) = AT(d.Mgrad, i); }
    const double g = AT(d.Ma, i) - AT(d.qfrc_smooth, i) - AT(d.qfrc_constraint, i);
    AT(d.grad, i) = g;
    AT(d.Mgrad, i) = g;
    gg += g * g;
  }
  solve_sparse(m, d.qLD, d.qLDiagInv, d.Mgrad, E, e);
  double beta = 0;
  if (!first) {
    double num = 0;
    for (int i = 0; i < m.nv; i++) num += AT(d.grad, i) * (AT(d.Mgrad, i) - AT(d.Mv, i));
    beta = num / (gg_old > BS_MINVAL ? gg_old : BS_MINVAL);
    if (beta < 0) beta = 0;
  }
  for (int i = 0; i < m.nv; i++) AT(d.search, i) = -AT(d.Mgrad, i) + (first ? 0.0 : beta * AT(d.search, i));
  return gg;
}

__device__ inline double update_gradient(const DevModel& m, const DevData& d, long e, bool first = true) {
  const long E = d.E;
  if (m.solver == 1) return cg_gradient(m, d, e, first);
  double gg = 0;
  for (int i = 0; i < m.nv; i++) {
    const double g = AT(d.Ma, i) - AT(d.qfrc_smooth, i) - AT(d.qfrc_constraint, i);
    AT(d.grad, i) = g;
    AT(d.Mgrad, i) = g;
    gg += g * g;
  }
  for (int k = 0; k < m.nM; k++) AT(d.H, k) = AT(d.qM, k);
  for (int r = 0; r < m.neq; r++) {
    const double D = AT(d.efc_D, r);
    AT(d.H, m.dof_Madr[m.eq_dof1[r]]) += D;
    if (m.eq_dof2[r] >= 0) {
      const double c = AT(d.eq_coef, r);
      AT(d.H, m.dof_Madr[m.eq_dof2[r]]) += D * c * c;
      AT(d.H, m.eq_Hadr[r]) -= D * c;
    }
  }
  for (int k = 0; k < m.nfric; k++) {
    const int r = m.neq + k;
    const double x = AT(d.jar, r), rf = AT(d.fric_rf, k);
    if (x > -rf && x < rf) AT(d.H, m.dof_Madr[m.fric_dof[k]]) += AT(d.efc_D, r);
  }
  for (int k = 0; k < m.nlim; k++) {
    const int r = m.neq + m.nfric + 2 * k, a = m.dof_Madr[m.jnt_dofadr[m.lim_jnt[k]]];
    double add = 0;
    if (AT(d.jar, r) < 0) add += AT(d.efc_D, r);
    if (AT(d.jar, r + 1) < 0) add += AT(d.efc_D, r + 1);
    AT(d.H, a) += add;
  }
  factor_sparse(m, d.H, d.Hdinv, E, e);
  solve_sparse(m, d.H, d.Hdinv, d.Mgrad, E, e);
  for (int i = 0; i < m.nv; i++) AT(d.search, i) = -AT(d.Mgrad, i);
  return gg;
}

struct LsPoint { double alpha, cost, d0, d1; };

__device__ inline LsPoint ls_eval(const DevModel& m, const DevData& d, long e, double alpha, double g0,
                                  double g1, double g2) {
  const long E = d.E;
  double q0 = g0, q1 = g1, q2 = g2;
  for (int r = 0; r < m.neq; r++) {
    const double jv = AT(d.jv, r), x = AT(d.jar, r) + alpha * jv, D = AT(d.efc_D, r);
    q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv;
  }
  for (int k = 0; k < m.nfric; k++) {
    const int r = m.neq + k;
    const double jv = AT(d.jv, r), x = AT(d.jar, r) + alpha * jv, D = AT(d.efc_D, r);
    const double rf = AT(d.fric_rf, k), fl = m.dof_frictionloss[m.fric_dof[k]];
    if (x <= -rf) { q0 += -fl * (0.5 * rf + x); q1 += -fl * jv; }
    else if (x >= rf) { q0 += -fl * (0.5 * rf - x); q1 += fl * jv; }
    else { q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv; }
  }
  for (int r = m.neq + m.nfric; r < m.nrow; r++) {
    const double jv = AT(d.jv, r), x = AT(d.jar, r) + alpha * jv, D = AT(d.efc_D, r);
    if (x < 0) { q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv; }
  }
  LsPoint p;
  p.alpha = alpha;
  p.cost = q0 + alpha * g1 + alpha * alpha * g2;
  p.d0 = q1 + 2 * alpha * g2;
  p.d1 = 2 * q2;
  if (p.d1 <= 0) p.d1 = BS_MINVAL;
  return p;
}

// exact line search on the piecewise-quadratic cost along `search`: Newton steps until the
// derivative changes sign, then bracketed Newton/bisection (MuJoCo's primal line search shape)
__device__ inline double line_search(const DevModel& m, const DevData& d, long e, double gauss,
                                     double scale) {
  const long E = d.E;
  double ss = 0;
  for (int i = 0; i < m.nv; i++) ss += AT(d.search, i) * AT(d.search, i);
  const double snorm = sqrt(ss);
  if (snorm < BS_MINVAL) return 0;
  const double gtol = m.tolerance * m.ls_tolerance * snorm / scale;
  mul_m(m, d.qM, d.Mv, d.search, E, e);
  rows_residual(m, d, e, d.search, d.jv, false);
  double g1 = 0, g2 = 0;
  for (int i = 0; i < m.nv; i++) {
    const double s = AT(d.search, i);
    g1 += s * (AT(d.Ma, i) - AT(d.qfrc_smooth, i));
    g2 += s * AT(d.Mv, i);
  }
  g2 *= 0.5;
  int it = 0;
  const int maxit = m.ls_iterations;
  LsPoint p0 = ls_eval(m, d, e, 0.0, gauss, g1, g2);
  LsPoint p1 = ls_eval(m, d, e, p0.alpha - p0.d0 / p0.d1, gauss, g1, g2);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  const double dir = p1.d0 < 0 ? 1.0 : -1.0;
  bool p2update = false;
  LsPoint p2 = p1;
  while (p1.d0 * dir <= -gtol && it < maxit) {
    p2 = p1; p2update = true;
    p1 = ls_eval(m, d, e, p1.alpha - p1.d0 / p1.d1, gauss, g1, g2);
    it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= maxit || !p2update) return p1.alpha;
  LsPoint p2next = p1;
  LsPoint p1next = ls_eval(m, d, e, p1.alpha - p1.d0 / p1.d1, gauss, g1, g2);
  while (it < maxit) {
    LsPoint pmid = ls_eval(m, d, e, 0.5 * (p1.alpha + p2.alpha), gauss, g1, g2);
    it++;
    bool b1 = false, b2 = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const LsPoint c = k == 0 ? p1next : (k == 1 ? p2next : pmid);
      if (fabs(c.d0) < gtol) return c.alpha;
      if (c.d0 * dir < 0) {
        if (c.alpha * dir > p2.alpha * dir && c.alpha * dir < p1.alpha * dir) { p2 = c; b2 = true; }
      } else {
        if (c.alpha * dir < p1.alpha * dir && c.alpha * dir > p2.alpha * dir) { p1 = c; b1 = true; }
      }
    }
    if (!b1 && !b2) break;
    if (b1) p1next = ls_eval(m, d, e, p1.alpha - p1.d0 / p1.d1, gauss, g1, g2);
    if (b2) p2next = ls_eval(m, d, e, p2.alpha - p2.d0 / p2.d1, gauss, g1, g2);
  }
  const LsPoint best = p1.cost <= p2.cost ? p1 : p2;
  return best.cost < p0.cost ? best.alpha : 0.0;
}

// ------------------------------------------------------------------ A7 + A11 constraint stage
// mj_makeConstraint ... mj_fwdConstraint with the Newton solver (reference default: no <option>
// element in output/DexterousDynamos.xml, so solver=Newton, iterations=100, tolerance=1e-8 [EXT]).
__device__ inline void stage_constraint(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  int nact = 0;
  if (!(m.disableflags & BS_DSBL_CONSTRAINT)) nact = make_rows(m, d, e);
  AT(d.nefc, 0) = nact;
  if (nact == 0) {
    for (int i = 0; i < m.nv; i++) {
      const double a = AT(d.qacc_smooth, i);
      AT(d.qacc, i) = a; AT(d.qacc_warmstart, i) = a; AT(d.qfrc_constraint, i) = 0;
    }
    AT(d.solver_iter, 0) = 0;
    return;
  }
  // warm start: previous acceleration only if cheaper than the unconstrained one
  bool use_smooth = true;
  if (!(m.disableflags & BS_DSBL_WARMSTART)) {
    const double cw = cost_at(m, d, e, d.qacc_warmstart);
    const double cs = cost_at(m, d, e, d.qacc_smooth);
    use_smooth = cw > cs;
  }
  for (int i = 0; i < m.nv; i++) AT(d.qacc, i) = use_smooth ? AT(d.qacc_smooth, i) : AT(d.qacc_warmstart, i);
  const double scale = 1.0 / (m.meaninertia * (m.nv > 1 ? m.nv : 1));
  mul_m(m, d.qM, d.Ma, d.qacc, E, e);
  rows_residual(m, d, e, d.qacc, d.jar, true);
  double gauss = 0;
  for (int i = 0; i < m.nv; i++)
    gauss += (AT(d.Ma, i) - AT(d.qfrc_smooth, i)) * (AT(d.qacc, i) - AT(d.qacc_smooth, i));
  gauss *= 0.5;
  double cost = gauss + rows_update<true>(m, d, e, d.jar);
  update_gradient(m, d, e);
  int iter = 0;
  while (iter < m.iterations) {
    const double alpha = line_search(m, d, e, gauss, scale);
    if (alpha == 0) break;
    for (int i = 0; i < m.nv; i++) {
      AT(d.qacc, i) += alpha * AT(d.search, i);
      AT(d.Ma, i) += alpha * AT(d.Mv, i);
    }
    for (int r = 0; r < m.nrow; r++) AT(d.jar, r) += alpha * AT(d.jv, r);
    const double oldcost = cost;
    gauss = 0;
    for (int i = 0; i < m.nv; i++)
      gauss += (AT(d.Ma, i) - AT(d.qfrc_smooth, i)) * (AT(d.qacc, i) - AT(d.qacc_smooth, i));
    gauss *= 0.5;
    cost = gauss + rows_update<true>(m, d, e, d.jar);
    const double gg = update_gradient(m, d, e, false);
    iter++;
    const double improvement = scale * (oldcost - cost), gradient = scale * sqrt(gg);
    if (improvement < m.tolerance || gradient < m.tolerance) break;
  }
  AT(d.solver_iter, 0) = iter;
  for (int i = 0; i < m.nv; i++) AT(d.qacc_warmstart, i) = AT(d.qacc, i);
}

__device__ inline void stage_constraint_pgs(const DevModel& m, const DevData& d, long e);  // physics_contact.h
__device__ inline void env_forward(const DevModel& m, const DevData& d, long e) {
  stage_position(m, d, e);
  stage_factor(m, d, e);
  stage_velocity(m, d, e);
  if (m.solver == 0) stage_constraint_pgs(m, d, e);
  else stage_constraint(m, d, e);
}

// ------------------------------------------------------------------ A12 mj_Euler
// semi-implicit Euler with joint damping integrated implicitly:
// (M + h diag(damping)) a = qfrc_smooth + qfrc_constraint  (damping: reference src/Mujoco_XML.py:71)
__device__ inline void stage_euler(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  const double h = m.timestep;
  const double* acc = d.qacc;
  if (m.damped) {
    for (int k = 0; k < m.nM; k++) AT(d.H, k) = AT(d.qM, k);
    for (int i = 0; i < m.nv; i++) {
      AT(d.H, m.dof_Madr[i]) += h * m.dof_damping[i];
      AT(d.Mgrad, i) = AT(d.qfrc_smooth, i) + AT(d.qfrc_constraint, i);
    }
    factor_sparse(m, d.H, d.Hdinv, E, e);
    solve_sparse(m, d.H, d.Hdinv, d.Mgrad, E, e);
    acc = d.Mgrad;
  }
  for (int i = 0; i < m.nv; i++) AT(d.qvel, i) += h * AT(acc, i);
  for (int j = 0; j < m.njnt; j++) AT(d.qpos, m.jnt_qposadr[j]) += h * AT(d.qvel, m.jnt_dofadr[j]);
  AT(d.time, 0) += h;
}

// ------------------------------------------------------------------ A13 mj_RungeKutta(4)
// classical tableau; contemplated by the reference only as a comment (src/Mujoco_XML.py:56)
__device__ inline void stage_rk4(const DevModel& m, const DevData& d, long e) {
  const long E = d.E;
  const int nv = m.nv;
  const double h = m.timestep, t0 = AT(d.time, 0);
  const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}};
  const double B[4] = {1 / 6., 1 / 3., 1 / 3., 1 / 6.}, C[3] = {0.5, 0.5, 1.0};
  for (int k = 0; k < nv; k++) {
    AT(d.rk_q0, k) = AT(d.qpos, k); AT(d.rk_v0, k) = AT(d.qvel, k);
    AT(d.rk_Fv, k) = AT(d.qvel, k); AT(d.rk_Fa, k) = AT(d.qacc, k);
  }
  for (int i = 1; i < 4; i++) {
    for (int k = 0; k < nv; k++) {
      double dv = 0, da = 0;
      for (int j = 0; j < i; j++) {
        dv += A[i - 1][j] * AT(d.rk_Fv, j * nv + k);
        da += A[i - 1][j] * AT(d.rk_Fa, j * nv + k);
      }
      AT(d.qpos, k) = AT(d.rk_q0, k) + h * dv;
      AT(d.qvel, k) = AT(d.rk_v0, k) + h * da;
    }
    AT(d.time, 0) = t0 + C[i - 1] * h;
    env_forward(m, d, e);
    for (int k = 0; k < nv; k++) {
      AT(d.rk_Fv, i * nv + k) = AT(d.qvel, k); AT(d.rk_Fa, i * nv + k) = AT(d.qacc, k);
    }
  }
  for (int k = 0; k < nv; k++) {
    double dv = 0, da = 0;
    for (int j = 0; j < 4; j++) { dv += B[j] * AT(d.rk_Fv, j * nv + k); da += B[j] * AT(d.rk_Fa, j * nv + k); }
    AT(d.qvel, k) = AT(d.rk_v0, k) + h * da;
    AT(d.qpos, k) = AT(d.rk_q0, k) + h * dv;
  }
  AT(d.time, 0) = t0 + h;
}

// A13 with one launch per forward pass (needed when a forward pass spans several kernels, i.e.
// with contacts): called after forward pass i-1, i = 1..4.  Stores F[i-1] = (qvel, qacc) and
// moves the state to stage i (i < 4) or to the end of the step (i == 4).  mj_checkAcc is applied
// to the first forward pass; a reset abandons the step (the next one starts from the reset state).
__device__ inline void rk4_advance(const DevModel& m, const DevData& d, long e, int i) {
  const long E = d.E;
  const int nv = m.nv;
  const double h = m.timestep;
  const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}};
  const double B[4] = {1 / 6., 1 / 3., 1 / 3., 1 / 6.}, C[3] = {0.5, 0.5, 1.0};
  if (i == 1) {
    if (env_check(m, d, e, d.qacc, nv, 2)) { AT(d.rk_t0, 1) = 1.0; return; }
    AT(d.rk_t0, 1) = 0.0;
    AT(d.rk_t0, 0) = AT(d.time, 0);
    for (int k = 0; k < nv; k++) { AT(d.rk_q0, k) = AT(d.qpos, k); AT(d.rk_v0, k) = AT(d.qvel, k); }
  }
  if (AT(d.rk_t0, 1) != 0.0) return;
  for (int k = 0; k < nv; k++) {
    AT(d.rk_Fv, (i - 1) * nv + k) = AT(d.qvel, k); AT(d.rk_Fa, (i - 1) * nv + k) = AT(d.qacc, k);
  }
  const double t0 = AT(d.rk_t0, 0);
  for (int k = 0; k < nv; k++) {
    double dv = 0, da = 0;
    if (i < 4)
      for (int j = 0; j < i; j++) { dv += A[i - 1][j] * AT(d.rk_Fv, j * nv + k); da += A[i - 1][j] * AT(d.rk_Fa, j * nv + k); }
    else
      for (int j = 0; j < 4; j++) { dv += B[j] * AT(d.rk_Fv, j * nv + k); da += B[j] * AT(d.rk_Fa, j * nv + k); }
    if (i < 4) { AT(d.qpos, k) = AT(d.rk_q0, k) + h * dv; AT(d.qvel, k) = AT(d.rk_v0, k) + h * da; }
    else { AT(d.qvel, k) = AT(d.rk_v0, k) + h * da; AT(d.qpos, k) = AT(d.rk_q0, k) + h * dv; }
  }
  AT(d.time, 0) = i < 4 ? t0 + C[i - 1] * h : t0 + h;
}

// one mj_step for one environment
__device__ inline void env_step(const DevModel& m, const DevData& d, long e) {
  env_check(m, d, e, d.qpos, m.njnt, 0);
  env_check(m, d, e, d.qvel, m.nv, 1);
  env_forward(m, d, e);
  if (env_check(m, d, e, d.qacc, m.nv, 2)) env_forward(m, d, e);
  if (m.integrator == 1) stage_rk4(m, d, e);
  else stage_euler(m, d, e);
}
