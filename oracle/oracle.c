/* oracle.c — CPU restatement of MuJoCo 2.1.0's mj_step for the exporter's MJCF dialect.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h).
 *
 * Each routine names the MuJoCo routine it restates and the reference line that fixes
 * its inputs.  MuJoCo's source is not in /root/reference: the algorithms
 * follow MuJoCo's public documentation and are [EXT].
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MAXVAL 1e10
#define MINIMP 0.0001
#define MAXIMP 0.9999

enum { JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { INT_EULER = 0, INT_RK4 = 1 };
enum { SOL_PGS = 0, SOL_CG = 1, SOL_NEWTON = 2 };
enum {
  DSBL_CONSTRAINT = 1 << 0, DSBL_EQUALITY = 1 << 1, DSBL_FRICTIONLOSS = 1 << 2,
  DSBL_LIMIT = 1 << 3, DSBL_CONTACT = 1 << 4, DSBL_PASSIVE = 1 << 5, DSBL_GRAVITY = 1 << 6,
  DSBL_CLAMPCTRL = 1 << 7, DSBL_WARMSTART = 1 << 8, DSBL_ACTUATION = 1 << 10,
  DSBL_REFSAFE = 1 << 11
};
enum { CT_EQUALITY = 0, CT_FRICTION_DOF = 1, CT_LIMIT_JOINT = 3, CT_CONTACT_PYRAMIDAL = 6 };
enum { ST_SATISFIED = 0, ST_QUADRATIC = 1, ST_LINEARNEG = 2, ST_LINEARPOS = 3 };

/* ------------------------------------------------------------------ small math */
static void quat_mul(double *r, const double *a, const double *b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                 a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                 a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void quat2mat(double *R, const double *q) {
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
static void rot_vec_mat(double *r, const double *R, const double *v) {
  double t[3] = {R[0] * v[0] + R[1] * v[1] + R[2] * v[2], R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
                 R[6] * v[0] + R[7] * v[1] + R[8] * v[2]};
  memcpy(r, t, sizeof t);
}
static void rot_vec_quat(double *r, const double *v, const double *q) {
  double R[9];
  quat2mat(R, q);
  rot_vec_mat(r, R, v);
}
static void cross3(double *r, const double *a, const double *b) {
  double t[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  memcpy(r, t, sizeof t);
}
static double dotn(const double *a, const double *b, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
static void quat_normalize(double *q) {
  double n = sqrt(dotn(q, q, 4));
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
/* spatial inertia (10 numbers) times motion vector [rot; lin] */
static void mul_inert_vec(double *r, const double *I, const double *v) {
  r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
  r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
  r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
  r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
  r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
  r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}
static void cross_motion(double *r, const double *vel, const double *v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
static void cross_force(double *r, const double *vel, const double *f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

/* ------------------------------------------------------------------ alloc */
#define DUP(field, count, type)                                              \
  do {                                                                       \
    size_t nb_ = (size_t)(count) * sizeof(type);                             \
    type *p_ = (type *)malloc(nb_ ? nb_ : sizeof(type));                     \
    if (nb_) memcpy(p_, desc->field, nb_);                                   \
    m->field = p_;                                                           \
  } while (0)

orc_model *orc_model_create(const orc_model *desc) {
  orc_model *m = (orc_model *)malloc(sizeof *m);
  *m = *desc;
  int nb = m->nbody, nj = m->njnt, nv = m->nv, nu = m->nu, ng = m->ngeom, ne = m->neq;
  DUP(body_parentid, nb, int32_t); DUP(body_rootid, nb, int32_t); DUP(body_weldid, nb, int32_t);
  DUP(body_jntadr, nb, int32_t); DUP(body_jntnum, nb, int32_t); DUP(body_dofadr, nb, int32_t);
  DUP(body_dofnum, nb, int32_t);
  DUP(body_pos, nb * 3, double); DUP(body_quat, nb * 4, double); DUP(body_ipos, nb * 3, double);
  DUP(body_iquat, nb * 4, double); DUP(body_mass, nb, double); DUP(body_subtreemass, nb, double);
  DUP(body_inertia, nb * 3, double); DUP(body_invweight0, nb * 2, double);
  DUP(jnt_type, nj, int32_t); DUP(jnt_bodyid, nj, int32_t); DUP(jnt_qposadr, nj, int32_t);
  DUP(jnt_dofadr, nj, int32_t); DUP(jnt_limited, nj, int32_t);
  DUP(jnt_pos, nj * 3, double); DUP(jnt_axis, nj * 3, double); DUP(jnt_range, nj * 2, double);
  DUP(jnt_margin, nj, double); DUP(jnt_stiffness, nj, double); DUP(jnt_solref, nj * 2, double);
  DUP(jnt_solimp, nj * 5, double); DUP(qpos0, nj, double); DUP(qpos_spring, nj, double);
  DUP(dof_bodyid, nv, int32_t); DUP(dof_jntid, nv, int32_t); DUP(dof_parentid, nv, int32_t);
  DUP(dof_Madr, nv, int32_t);
  DUP(dof_armature, nv, double); DUP(dof_damping, nv, double); DUP(dof_frictionloss, nv, double);
  DUP(dof_solref, nv * 2, double); DUP(dof_solimp, nv * 5, double); DUP(dof_invweight0, nv, double);
  DUP(geom_type, ng, int32_t); DUP(geom_bodyid, ng, int32_t); DUP(geom_dataid, ng, int32_t);
  DUP(geom_condim, ng, int32_t); DUP(geom_priority, ng, int32_t);
  DUP(geom_pos, ng * 3, double); DUP(geom_quat, ng * 4, double); DUP(geom_size, ng * 3, double);
  DUP(geom_friction, ng * 3, double); DUP(geom_margin, ng, double); DUP(geom_gap, ng, double);
  DUP(geom_solref, ng * 2, double); DUP(geom_solimp, ng * 5, double); DUP(geom_solmix, ng, double);
  DUP(geom_rbound, ng, double);
  DUP(mesh_vertadr, m->nmesh, int32_t); DUP(mesh_vertnum, m->nmesh, int32_t);
  DUP(hull_adj_start, m->nhullvert + 1, int32_t); DUP(hull_adj, m->nhulladj, int32_t);
  DUP(hull_vert, m->nhullvert * 3, double);
  DUP(actuator_trnid, nu, int32_t); DUP(actuator_ctrllimited, nu, int32_t);
  DUP(actuator_forcelimited, nu, int32_t);
  DUP(actuator_ctrlrange, nu * 2, double); DUP(actuator_forcerange, nu * 2, double);
  DUP(actuator_gear, nu, double); DUP(actuator_gainprm, nu * 3, double);
  DUP(actuator_biasprm, nu * 3, double);
  DUP(pair_geom1, m->npair, int32_t); DUP(pair_geom2, m->npair, int32_t);
  DUP(eq_obj1id, ne, int32_t); DUP(eq_obj2id, ne, int32_t); DUP(eq_active, ne, int32_t);
  DUP(eq_data, ne * 5, double); DUP(eq_solref, ne * 2, double); DUP(eq_solimp, ne * 5, double);
  return m;
}

void orc_model_destroy(orc_model *m) {
  if (!m) return;
  /* every pointer member was malloc'ed by DUP; walk them via a table of addresses */
  const void *ptrs[] = {
      m->body_parentid, m->body_rootid, m->body_weldid, m->body_jntadr, m->body_jntnum,
      m->body_dofadr, m->body_dofnum, m->body_pos, m->body_quat, m->body_ipos, m->body_iquat,
      m->body_mass, m->body_subtreemass, m->body_inertia, m->body_invweight0, m->jnt_type,
      m->jnt_bodyid, m->jnt_qposadr, m->jnt_dofadr, m->jnt_limited, m->jnt_pos, m->jnt_axis,
      m->jnt_range, m->jnt_margin, m->jnt_stiffness, m->jnt_solref, m->jnt_solimp, m->qpos0,
      m->qpos_spring, m->dof_bodyid, m->dof_jntid, m->dof_parentid, m->dof_Madr, m->dof_armature,
      m->dof_damping, m->dof_frictionloss, m->dof_solref, m->dof_solimp, m->dof_invweight0,
      m->geom_type, m->geom_bodyid, m->geom_dataid, m->geom_condim, m->geom_priority,
      m->geom_pos, m->geom_quat, m->geom_size, m->geom_friction, m->geom_margin, m->geom_gap,
      m->geom_solref, m->geom_solimp, m->geom_solmix, m->geom_rbound, m->mesh_vertadr,
      m->mesh_vertnum, m->hull_adj_start, m->hull_adj, m->hull_vert, m->actuator_trnid,
      m->actuator_ctrllimited, m->actuator_forcelimited, m->actuator_ctrlrange,
      m->actuator_forcerange, m->actuator_gear, m->actuator_gainprm, m->actuator_biasprm,
      m->pair_geom1, m->pair_geom2, m->eq_obj1id, m->eq_obj2id, m->eq_active, m->eq_data,
      m->eq_solref, m->eq_solimp};
  for (size_t i = 0; i < sizeof ptrs / sizeof ptrs[0]; i++) free((void *)ptrs[i]);
  free(m);
}

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static int32_t *ialloc(size_t n) { return (int32_t *)calloc(n ? n : 1, sizeof(int32_t)); }

orc_data *orc_data_create(const orc_model *m) {
  orc_data *d = (orc_data *)calloc(1, sizeof *d);
  int nb = m->nbody, nj = m->njnt, nv = m->nv, nu = m->nu, ng = m->ngeom;
  d->qpos = dalloc(nj); d->qvel = dalloc(nv); d->ctrl = dalloc(nu);
  d->qfrc_applied = dalloc(nv); d->qacc = dalloc(nv); d->qacc_warmstart = dalloc(nv);
  d->xpos = dalloc(nb * 3); d->xquat = dalloc(nb * 4); d->xmat = dalloc(nb * 9);
  d->xipos = dalloc(nb * 3); d->ximat = dalloc(nb * 9); d->xanchor = dalloc(nj * 3);
  d->xaxis = dalloc(nj * 3); d->geom_xpos = dalloc(ng * 3); d->geom_xmat = dalloc(ng * 9);
  d->subtree_com = dalloc(nb * 3); d->cinert = dalloc(nb * 10); d->cdof = dalloc(nv * 6);
  d->crb = dalloc(nb * 10); d->qM = dalloc(m->nM); d->qLD = dalloc(m->nM);
  d->qLDiagInv = dalloc(nv);
  d->cvel = dalloc(nb * 6); d->cdof_dot = dalloc(nv * 6); d->qfrc_bias = dalloc(nv);
  d->qfrc_passive = dalloc(nv); d->actuator_force = dalloc(nu); d->qfrc_actuator = dalloc(nv);
  d->qfrc_smooth = dalloc(nv); d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv);
  int nefc_max = m->neq + nv + 2 * nj + 4 * ORC_MAXCON;
  d->nefc_max = nefc_max;
  d->efc_type = ialloc(nefc_max); d->efc_id = ialloc(nefc_max); d->efc_state = ialloc(nefc_max);
  d->efc_J = dalloc((size_t)nefc_max * nv); d->efc_pos = dalloc(nefc_max);
  d->efc_margin = dalloc(nefc_max); d->efc_frictionloss = dalloc(nefc_max);
  d->efc_diagApprox = dalloc(nefc_max); d->efc_KBIP = dalloc(4 * (size_t)nefc_max);
  d->efc_D = dalloc(nefc_max); d->efc_R = dalloc(nefc_max); d->efc_vel = dalloc(nefc_max);
  d->efc_aref = dalloc(nefc_max); d->efc_force = dalloc(nefc_max);
  d->scratch = dalloc((size_t)nefc_max * 8 + (size_t)nv * nv * 2 + (size_t)nv * 32 + nb * 32);
  orc_reset(m, d);
  return d;
}

void orc_data_destroy(orc_data *d) {
  if (!d) return;
  void *ptrs[] = {d->qpos, d->qvel, d->ctrl, d->qfrc_applied, d->qacc, d->qacc_warmstart, d->xpos,
                  d->xquat, d->xmat, d->xipos, d->ximat, d->xanchor, d->xaxis, d->geom_xpos,
                  d->geom_xmat, d->subtree_com, d->cinert, d->cdof, d->crb, d->qM, d->qLD,
                  d->qLDiagInv, d->cvel, d->cdof_dot, d->qfrc_bias, d->qfrc_passive,
                  d->actuator_force, d->qfrc_actuator, d->qfrc_smooth, d->qacc_smooth,
                  d->qfrc_constraint, d->efc_type, d->efc_id, d->efc_state, d->efc_J, d->efc_pos,
                  d->efc_margin, d->efc_frictionloss, d->efc_diagApprox, d->efc_KBIP, d->efc_D,
                  d->efc_R, d->efc_vel, d->efc_aref, d->efc_force, d->scratch};
  for (size_t i = 0; i < sizeof ptrs / sizeof ptrs[0]; i++) free(ptrs[i]);
  free(d);
}

/* mj_resetData: state to the model default (qpos0, zero velocity/control) */
void orc_reset(const orc_model *m, orc_data *d) {
  d->time = 0;
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->njnt);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memset(d->qfrc_applied, 0, sizeof(double) * m->nv);
  memset(d->qacc, 0, sizeof(double) * m->nv);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  d->nefc = d->ncon = 0;
  d->solver_iter = 0;
  d->warning = 0;
}

/* ------------------------------------------------------------------ A1 mj_kinematics
 * inputs fixed by body pos/quat attrs (reference output/DexterousDynamos.xml:29-98) and
 * joint pos/axis (xml:35-111), emitted by src/Mujoco_XML.py:212,255 */
void orc_kinematics(const orc_model *m, orc_data *d) {
  memset(d->xpos, 0, 3 * sizeof(double));
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  memset(d->xmat, 0, 9 * sizeof(double));
  d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
  memset(d->xipos, 0, 3 * sizeof(double));
  memcpy(d->ximat, d->xmat, 9 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    double pos[3], quat[4];
    rot_vec_mat(pos, d->xmat + 9 * p, m->body_pos + 3 * b);
    for (int k = 0; k < 3; k++) pos[k] += d->xpos[3 * p + k];
    quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
    for (int jj = 0; jj < m->body_jntnum[b]; jj++) {
      int j = m->body_jntadr[b] + jj;
      double *anchor = d->xanchor + 3 * j, *axis = d->xaxis + 3 * j;
      rot_vec_quat(anchor, m->jnt_pos + 3 * j, quat);
      for (int k = 0; k < 3; k++) anchor[k] += pos[k];
      rot_vec_quat(axis, m->jnt_axis + 3 * j, quat);
      double dq = d->qpos[m->jnt_qposadr[j]] - m->qpos0[j];
      if (m->jnt_type[j] == JNT_SLIDE) {
        for (int k = 0; k < 3; k++) pos[k] += axis[k] * dq;
      } else {
        double s = sin(0.5 * dq), ql[4] = {cos(0.5 * dq), 0, 0, 0}, v[3];
        for (int k = 0; k < 3; k++) ql[1 + k] = m->jnt_axis[3 * j + k] * s;
        quat_mul(quat, quat, ql);
        rot_vec_quat(v, m->jnt_pos + 3 * j, quat);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - v[k];
      }
    }
    quat_normalize(quat);
    memcpy(d->xpos + 3 * b, pos, sizeof pos);
    memcpy(d->xquat + 4 * b, quat, sizeof quat);
    quat2mat(d->xmat + 9 * b, quat);
    double iq[4], ip[3];
    rot_vec_mat(ip, d->xmat + 9 * b, m->body_ipos + 3 * b);
    for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = pos[k] + ip[k];
    quat_mul(iq, quat, m->body_iquat + 4 * b);
    quat2mat(d->ximat + 9 * b, iq);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double gp[3], gq[4];
    rot_vec_mat(gp, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + gp[k];
    quat_mul(gq, d->xquat + 4 * b, m->geom_quat + 4 * g);
    quat2mat(d->geom_xmat + 9 * g, gq);
  }
}

/* ------------------------------------------------------------------ A2 mj_comPos
 * masses / inertias come from the meshes (xml:10-25) x scale (xml:7) */
void orc_com_pos(const orc_model *m, orc_data *d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = m->body_mass[b] * d->xipos[3 * b + k];
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * b + k];
  }
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < MINVAL)
      memcpy(d->subtree_com + 3 * b, d->xipos + 3 * b, 3 * sizeof(double));
    else
      for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] /= m->body_subtreemass[b];
  }
  memset(d->cinert, 0, 10 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    const double *R = d->ximat + 9 * b, *I = m->body_inertia + 3 * b;
    const double *c = d->subtree_com + 3 * m->body_rootid[b];
    double mass = m->body_mass[b], off[3], *ci = d->cinert + 10 * b;
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3 * b + k] - c[k];
    /* R diag(I) R^T, upper triangle in the order XX YY ZZ XY XZ YZ */
    double t[9];
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < 3; k++) t[3 * r + k] = R[3 * r + k] * I[k];
    ci[0] = t[0] * R[0] + t[1] * R[1] + t[2] * R[2];
    ci[1] = t[3] * R[3] + t[4] * R[4] + t[5] * R[5];
    ci[2] = t[6] * R[6] + t[7] * R[7] + t[8] * R[8];
    ci[3] = t[0] * R[3] + t[1] * R[4] + t[2] * R[5];
    ci[4] = t[0] * R[6] + t[1] * R[7] + t[2] * R[8];
    ci[5] = t[3] * R[6] + t[4] * R[7] + t[5] * R[8];
    ci[0] += mass * (off[1] * off[1] + off[2] * off[2]);
    ci[1] += mass * (off[0] * off[0] + off[2] * off[2]);
    ci[2] += mass * (off[0] * off[0] + off[1] * off[1]);
    ci[3] -= mass * off[0] * off[1];
    ci[4] -= mass * off[0] * off[2];
    ci[5] -= mass * off[1] * off[2];
    ci[6] = mass * off[0]; ci[7] = mass * off[1]; ci[8] = mass * off[2];
    ci[9] = mass;
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j];
    const double *c = d->subtree_com + 3 * m->body_rootid[b];
    double off[3], *cd = d->cdof + 6 * m->jnt_dofadr[j];
    const double *axis = d->xaxis + 3 * j;
    for (int k = 0; k < 3; k++) off[k] = c[k] - d->xanchor[3 * j + k];
    if (m->jnt_type[j] == JNT_SLIDE) {
      cd[0] = cd[1] = cd[2] = 0;
      memcpy(cd + 3, axis, 3 * sizeof(double));
    } else {
      memcpy(cd, axis, 3 * sizeof(double));
      cross3(cd + 3, axis, off);
    }
  }
}

/* ------------------------------------------------------------------ A3 mj_crb
 * armature from reference src/Mujoco_XML.py:71 */
void orc_crb(const orc_model *m, orc_data *d) {
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * nb);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  memset(d->qM, 0, sizeof(double) * m->nM);
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i];
    double buf[6];
    d->qM[adr] += m->dof_armature[i];
    mul_inert_vec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) d->qM[adr++] += dotn(d->cdof + 6 * j, buf, 6);
  }
}

/* ------------------------------------------------------------------ A4 mj_factorM */
static void factor_sparse(const orc_model *m, double *qLD, double *diaginv) {
  int nv = m->nv;
  for (int k = nv - 1; k >= 0; k--) {
    int akk = m->dof_Madr[k], aki = akk + 1;
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i], aki++) {
      double tmp = qLD[aki] / qLD[akk];
      int aij = m->dof_Madr[i], akj = aki;
      for (int j = i; j >= 0; j = m->dof_parentid[j]) qLD[aij++] -= tmp * qLD[akj++];
      qLD[aki] = tmp;
    }
    diaginv[k] = 1.0 / qLD[akk];
  }
}
static void solve_sparse(const orc_model *m, const double *qLD, const double *diaginv, double *x) {
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--) {
    int adr = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= qLD[adr++] * x[i];
  }
  for (int i = 0; i < nv; i++) x[i] *= diaginv[i];
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[i] -= qLD[adr++] * x[j];
  }
}
void orc_factor_m(const orc_model *m, orc_data *d) {
  memcpy(d->qLD, d->qM, sizeof(double) * m->nM);
  factor_sparse(m, d->qLD, d->qLDiagInv);
}
void orc_solve_m(const orc_model *m, const orc_data *d, double *x) {
  solve_sparse(m, d->qLD, d->qLDiagInv, x);
}
void orc_mul_m(const orc_model *m, const orc_data *d, double *res, const double *v) {
  for (int i = 0; i < m->nv; i++) res[i] = 0;
  for (int i = 0; i < m->nv; i++) {
    int adr = m->dof_Madr[i];
    res[i] += d->qM[adr++] * v[i];
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j], adr++) {
      res[i] += d->qM[adr] * v[j];
      res[j] += d->qM[adr] * v[i];
    }
  }
}
void orc_full_m(const orc_model *m, const orc_data *d, double *dense) {
  int nv = m->nv;
  memset(dense, 0, sizeof(double) * nv * nv);
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j], adr++)
      dense[i * nv + j] = dense[j * nv + i] = d->qM[adr];
  }
}

/* ------------------------------------------------------------------ A8 mj_comVel, mj_passive, mj_rne
 * damping from reference src/Mujoco_XML.py:71 */
static void com_vel(const orc_model *m, orc_data *d) {
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double v[6];
    memcpy(v, d->cvel + 6 * m->body_parentid[b], sizeof v);
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int i = m->body_dofadr[b] + k;
      cross_motion(d->cdof_dot + 6 * i, v, d->cdof + 6 * i);
      for (int c = 0; c < 6; c++) v[c] += d->cdof[6 * i + c] * d->qvel[i];
    }
    memcpy(d->cvel + 6 * b, v, sizeof v);
  }
}
void orc_rne(const orc_model *m, orc_data *d, int flg_acc, double *result) {
  int nb = m->nbody;
  double *cacc = d->scratch, *cfrc = d->scratch + 6 * nb;
  memset(cacc, 0, 6 * sizeof(double));
  if (!(m->disableflags & DSBL_GRAVITY))
    for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
  memset(cfrc, 0, 6 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    double *a = cacc + 6 * b, t0[6], t1[6], t2[6];
    memcpy(a, cacc + 6 * m->body_parentid[b], 6 * sizeof(double));
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int i = m->body_dofadr[b] + k;
      for (int c = 0; c < 6; c++) {
        a[c] += d->cdof_dot[6 * i + c] * d->qvel[i];
        if (flg_acc) a[c] += d->cdof[6 * i + c] * d->qacc[i];
      }
    }
    mul_inert_vec(t0, d->cinert + 10 * b, a);
    mul_inert_vec(t1, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(t2, d->cvel + 6 * b, t1);
    for (int c = 0; c < 6; c++) cfrc[6 * b + c] = t0[c] + t2[c];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int c = 0; c < 6; c++) cfrc[6 * p + c] += cfrc[6 * b + c];
  }
  for (int i = 0; i < m->nv; i++) result[i] = dotn(d->cdof + 6 * i, cfrc + 6 * m->dof_bodyid[i], 6);
}
void orc_fwd_velocity(const orc_model *m, orc_data *d) {
  com_vel(m, d);
  for (int i = 0; i < m->nv; i++) {
    double f = 0;
    if (!(m->disableflags & DSBL_PASSIVE)) {
      int j = m->dof_jntid[i];
      f = -m->jnt_stiffness[j] * (d->qpos[m->jnt_qposadr[j]] - m->qpos_spring[j]) -
          m->dof_damping[i] * d->qvel[i];
    }
    d->qfrc_passive[i] = f;
  }
  orc_rne(m, d, 0, d->qfrc_bias);
}

/* ------------------------------------------------------------------ A9 mj_fwdActuation
 * position servo kp / forcerange / ctrlrange: reference src/Mujoco_XML.py:72,261,
 * src/Fusion_to_Mujoco.py:110, xml:6,119-135 */
void orc_fwd_actuation(const orc_model *m, orc_data *d) {
  memset(d->qfrc_actuator, 0, sizeof(double) * m->nv);
  if (m->disableflags & DSBL_ACTUATION) {
    memset(d->actuator_force, 0, sizeof(double) * m->nu);
    return;
  }
  for (int u = 0; u < m->nu; u++) {
    int j = m->actuator_trnid[u];
    double gear = m->actuator_gear[u];
    double len = gear * d->qpos[m->jnt_qposadr[j]], vel = gear * d->qvel[m->jnt_dofadr[j]];
    double c = d->ctrl[u];
    if (m->actuator_ctrllimited[u] && !(m->disableflags & DSBL_CLAMPCTRL)) {
      if (c < m->actuator_ctrlrange[2 * u]) c = m->actuator_ctrlrange[2 * u];
      if (c > m->actuator_ctrlrange[2 * u + 1]) c = m->actuator_ctrlrange[2 * u + 1];
    }
    const double *bp = m->actuator_biasprm + 3 * u;
    double f = m->actuator_gainprm[3 * u] * c + bp[0] + bp[1] * len + bp[2] * vel;
    if (m->actuator_forcelimited[u]) {
      if (f < m->actuator_forcerange[2 * u]) f = m->actuator_forcerange[2 * u];
      if (f > m->actuator_forcerange[2 * u + 1]) f = m->actuator_forcerange[2 * u + 1];
    }
    d->actuator_force[u] = f;
    d->qfrc_actuator[m->jnt_dofadr[j]] += gear * f;
  }
}

/* ------------------------------------------------------------------ A10 mj_fwdAcceleration */
void orc_fwd_acceleration(const orc_model *m, orc_data *d) {
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_smooth[i] =
        d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  orc_solve_m(m, d, d->qacc_smooth);
}

/* ------------------------------------------------------------------ A7 mj_makeConstraint
 * equality coefficient 70/120: reference src/Fusion_to_Mujoco.py:107-108, src/Mujoco_XML.py:274;
 * frictionloss / limit margin: src/Mujoco_XML.py:71; condim/friction: src/Mujoco_XML.py:70 */
static int add_row(const orc_model *m, orc_data *d, int type, int id, double pos, double margin,
                   double floss, double diag) {
  int i = d->nefc++;
  memset(d->efc_J + (size_t)i * m->nv, 0, sizeof(double) * m->nv);
  d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin;
  d->efc_frictionloss[i] = floss; d->efc_diagApprox[i] = diag;
  return i;
}

/* translational Jacobian of a world point attached to body b (mj_jac, jacp only) */
static void jac_point(const orc_model *m, const orc_data *d, double *jacp, int b, const double *pt) {
  int nv = m->nv;
  memset(jacp, 0, sizeof(double) * 3 * nv);
  const double *c = d->subtree_com + 3 * m->body_rootid[b];
  double off[3] = {pt[0] - c[0], pt[1] - c[1], pt[2] - c[2]};
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b == 0) return;
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
    const double *cd = d->cdof + 6 * i;
    double t[3];
    cross3(t, cd, off);
    for (int k = 0; k < 3; k++) jacp[k * nv + i] = cd[3 + k] + t[k];
  }
}

static void impedance(const double *solimp, double pos, double margin, double *imp, double *impP) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(fmax(dmin, MINIMP), MAXIMP); dmax = fmin(fmax(dmax, MINIMP), MAXIMP);
  width = fmax(width, MINVAL); mid = fmin(fmax(mid, MINIMP), MAXIMP); power = fmax(power, 1.0);
  double x = fabs(pos - margin) / width;
  double sgn = (pos - margin) > 0 ? 1.0 : ((pos - margin) < 0 ? -1.0 : 0.0);
  if (dmin == dmax || x >= 1.0) {
    *imp = (x >= 1.0) ? dmax : 0.5 * (dmin + dmax);
    *impP = 0;
    return;
  }
  if (x <= 0) { *imp = dmin; *impP = 0; return; }
  double y, yP;
  if (power == 1.0) { y = x; yP = 1; }
  else if (x <= mid) {
    double a = 1.0 / pow(mid, power - 1);
    y = a * pow(x, power); yP = power * a * pow(x, power - 1);
  } else {
    double b = 1.0 / pow(1 - mid, power - 1);
    y = 1 - b * pow(1 - x, power); yP = power * b * pow(1 - x, power - 1);
  }
  *imp = dmin + y * (dmax - dmin);
  *impP = yP * sgn * (dmax - dmin) / width;
}

void orc_make_constraint(const orc_model *m, orc_data *d) {
  int nv = m->nv;
  d->nefc = d->ne = d->nf = d->nl = 0;
  if (m->disableflags & DSBL_CONSTRAINT) return;
  /* --- equality (joint polynomial coupling) */
  if (!(m->disableflags & DSBL_EQUALITY))
    for (int e = 0; e < m->neq; e++) {
      if (!m->eq_active[e]) continue;
      int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
      const double *c = m->eq_data + 5 * e;
      double p1 = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[j1], pos, deriv = 0;
      double diag = m->dof_invweight0[m->jnt_dofadr[j1]];
      if (j2 >= 0) {
        double x = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[j2];
        pos = p1 - (c[0] + x * (c[1] + x * (c[2] + x * (c[3] + x * c[4]))));
        deriv = c[1] + x * (2 * c[2] + x * (3 * c[3] + x * 4 * c[4]));
        diag += m->dof_invweight0[m->jnt_dofadr[j2]];
      } else pos = p1 - c[0];
      int i = add_row(m, d, CT_EQUALITY, e, pos, 0, 0, diag);
      d->efc_J[(size_t)i * nv + m->jnt_dofadr[j1]] = 1;
      if (j2 >= 0) d->efc_J[(size_t)i * nv + m->jnt_dofadr[j2]] = -deriv;
      d->ne++;
    }
  /* --- dof friction loss */
  if (!(m->disableflags & DSBL_FRICTIONLOSS))
    for (int i = 0; i < nv; i++)
      if (m->dof_frictionloss[i] > 0) {
        int r = add_row(m, d, CT_FRICTION_DOF, i, 0, 0, m->dof_frictionloss[i], m->dof_invweight0[i]);
        d->efc_J[(size_t)r * nv + i] = 1;
        d->nf++;
      }
  /* --- joint limits */
  if (!(m->disableflags & DSBL_LIMIT))
    for (int j = 0; j < m->njnt; j++)
      if (m->jnt_limited[j]) {
        double v = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
        for (int side = -1; side <= 1; side += 2) {
          double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - v);
          if (dist < margin) {
            int r = add_row(m, d, CT_LIMIT_JOINT, j, dist, margin, 0,
                            m->dof_invweight0[m->jnt_dofadr[j]]);
            d->efc_J[(size_t)r * nv + m->jnt_dofadr[j]] = -(double)side;
            d->nl++;
          }
        }
      }
  /* --- contacts: pyramidal cone, 2*(dim-1) rows each */
  if (!(m->disableflags & DSBL_CONTACT)) {
    double *j1 = d->scratch + 12 * m->nbody, *j2 = j1 + 3 * nv, *jc = j2 + 3 * nv;
    for (int c = 0; c < d->ncon; c++) {
      orc_contact *con = d->contact + c;
      con->efc_address = -1;
      if (con->dist >= con->includemargin) continue;
      int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
      jac_point(m, d, j1, b1, con->pos);
      jac_point(m, d, j2, b2, con->pos);
      /* difference (body2 - body1) rotated into the contact frame */
      for (int r = 0; r < con->dim; r++)
        for (int k = 0; k < nv; k++) {
          double s = 0;
          for (int a = 0; a < 3; a++) s += con->frame[3 * r + a] * (j2[a * nv + k] - j1[a * nv + k]);
          jc[r * nv + k] = s;
        }
      double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
      double rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
      con->efc_address = d->nefc;
      if (con->dim == 1) {
        int r = add_row(m, d, CT_CONTACT_PYRAMIDAL, c, con->dist, con->includemargin, 0, tran);
        memcpy(d->efc_J + (size_t)r * nv, jc, sizeof(double) * nv);
        continue;
      }
      for (int k = 1; k < con->dim; k++) {
        double fri = con->friction[k - 1];
        double diag = tran + fri * fri * (k < 3 ? tran : rot);
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
          int r = add_row(m, d, CT_CONTACT_PYRAMIDAL, c, con->dist, con->includemargin, 0, diag);
          for (int q = 0; q < nv; q++)
            d->efc_J[(size_t)r * nv + q] = jc[q] + sgn * fri * jc[k * nv + q];
        }
      }
    }
  }
  /* --- mj_makeImpedance: R, D, KBIP */
  for (int i = 0; i < d->nefc; i++) {
    const double *solref, *solimp;
    int id = d->efc_id[i];
    switch (d->efc_type[i]) {
      case CT_EQUALITY: solref = m->eq_solref + 2 * id; solimp = m->eq_solimp + 5 * id; break;
      case CT_FRICTION_DOF: solref = m->dof_solref + 2 * id; solimp = m->dof_solimp + 5 * id; break;
      case CT_LIMIT_JOINT: solref = m->jnt_solref + 2 * id; solimp = m->jnt_solimp + 5 * id; break;
      default: solref = d->contact[id].solref; solimp = d->contact[id].solimp; break;
    }
    double imp, impP;
    impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp, &impP);
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
    double K, B, dmax = fmin(fmax(solimp[1], MINIMP), MAXIMP);
    if (solref[0] > 0) {
      double tc = solref[0], dr = solref[1];
      if (!(m->disableflags & DSBL_REFSAFE)) tc = fmax(tc, 2 * m->timestep);
      K = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2.0 / fmax(MINVAL, dmax * tc);
    } else {
      K = -solref[0] / fmax(MINVAL, dmax * dmax);
      B = -solref[1] / fmax(MINVAL, dmax);
    }
    if (d->efc_type[i] == CT_FRICTION_DOF) K = 0;
    d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B;
    d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = impP;
  }
  /* pyramidal contacts: every edge row gets R = 2 mu^2 R_normal, mu = friction[0]/sqrt(impratio) */
  for (int c = 0; c < d->ncon; c++) {
    orc_contact *con = d->contact + c;
    if (con->efc_address < 0 || con->dim < 2) continue;
    double mu = con->friction[0] / sqrt(fmax(MINVAL, m->impratio));
    int a = con->efc_address;
    double Rpy = fmax(MINVAL, 2 * mu * mu * d->efc_R[a]);
    for (int k = 0; k < 2 * (con->dim - 1); k++) d->efc_R[a + k] = Rpy;
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1.0 / d->efc_R[i];
}

/* mj_referenceConstraint: aref = -B*vel - K*imp*(pos - margin) */
static void reference_constraint(const orc_model *m, orc_data *d) {
  for (int i = 0; i < d->nefc; i++) {
    d->efc_vel[i] = dotn(d->efc_J + (size_t)i * m->nv, d->qvel, m->nv);
    const double *k = d->efc_KBIP + 4 * i;
    d->efc_aref[i] = -k[1] * d->efc_vel[i] - k[0] * k[2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* ------------------------------------------------------------------ A11 mj_fwdConstraint (Newton) */
typedef struct {
  const orc_model *m;
  orc_data *d;
  int nv, nefc;
  double *jar, *Ma, *grad, *Mgrad, *search, *Mv, *jv, *H, *qacc;
  double cost, gauss, scale;
  double quadGauss[3];
} solctx;

/* per-row force / state / cost at residual jar; returns the constraint cost sum */
static double constraint_update(const orc_model *m, orc_data *d, const double *jar, int write) {
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    double x = jar[i], D = d->efc_D[i], f;
    int st;
    switch (d->efc_type[i]) {
      case CT_EQUALITY: st = ST_QUADRATIC; f = -D * x; cost += 0.5 * D * x * x; break;
      case CT_FRICTION_DOF: {
        double fl = d->efc_frictionloss[i], rf = d->efc_R[i] * fl;
        if (x <= -rf) { st = ST_LINEARNEG; f = fl; cost += -fl * (0.5 * rf + x); }
        else if (x >= rf) { st = ST_LINEARPOS; f = -fl; cost += -fl * (0.5 * rf - x); }
        else { st = ST_QUADRATIC; f = -D * x; cost += 0.5 * D * x * x; }
        break;
      }
      default:
        if (x < 0) { st = ST_QUADRATIC; f = -D * x; cost += 0.5 * D * x * x; }
        else { st = ST_SATISFIED; f = 0; }
    }
    if (write) { d->efc_force[i] = f; d->efc_state[i] = st; }
  }
  return cost;
}

double orc_cost(const orc_model *m, orc_data *d, const double *qacc) {
  int nv = m->nv;
  double *jar = d->scratch + 12 * m->nbody, *Ma = jar + d->nefc_max;
  for (int i = 0; i < d->nefc; i++)
    jar[i] = dotn(d->efc_J + (size_t)i * nv, qacc, nv) - d->efc_aref[i];
  orc_mul_m(m, d, Ma, qacc);
  double g = 0;
  for (int i = 0; i < nv; i++) g += (Ma[i] - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
  return 0.5 * g + constraint_update(m, d, jar, 0);
}

static void sol_update_constraint(solctx *c) {
  const orc_model *m = c->m;
  orc_data *d = c->d;
  int nv = c->nv;
  double s = constraint_update(m, d, c->jar, 1);
  for (int k = 0; k < nv; k++) d->qfrc_constraint[k] = 0;
  for (int i = 0; i < d->nefc; i++) {
    double f = d->efc_force[i];
    if (f != 0)
      for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += d->efc_J[(size_t)i * nv + k] * f;
  }
  double g = 0;
  for (int k = 0; k < nv; k++) g += (c->Ma[k] - d->qfrc_smooth[k]) * (c->qacc[k] - d->qacc_smooth[k]);
  c->gauss = 0.5 * g;
  c->cost = c->gauss + s;
}

/* dense Cholesky H = L L^T in place (lower), with MuJoCo's MINVAL floor on the pivots */
static void chol_factor(double *H, int n) {
  for (int j = 0; j < n; j++) {
    double s = H[j * n + j];
    for (int k = 0; k < j; k++) s -= H[j * n + k] * H[j * n + k];
    if (s < MINVAL) s = MINVAL;
    double l = sqrt(s);
    H[j * n + j] = l;
    for (int i = j + 1; i < n; i++) {
      double t = H[i * n + j];
      for (int k = 0; k < j; k++) t -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = t / l;
    }
  }
}
static void chol_solve(const double *L, int n, double *x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

static void sol_update_gradient(solctx *c) {
  const orc_model *m = c->m;
  orc_data *d = c->d;
  int nv = c->nv;
  for (int k = 0; k < nv; k++) c->grad[k] = c->Ma[k] - d->qfrc_smooth[k] - d->qfrc_constraint[k];
  if (m->solver == SOL_CG) { /* mj_solCG: the preconditioner is M itself, Mgrad = M^-1 grad */
    memcpy(c->Mgrad, c->grad, sizeof(double) * nv);
    orc_solve_m(m, d, c->Mgrad);
    return;
  }
  /* H = M + J^T diag(D_active) J */
  orc_full_m(m, d, c->H);
  for (int i = 0; i < d->nefc; i++) {
    if (d->efc_state[i] != ST_QUADRATIC) continue;
    const double *J = d->efc_J + (size_t)i * nv;
    double D = d->efc_D[i];
    for (int a = 0; a < nv; a++) {
      if (J[a] == 0) continue;
      double t = D * J[a];
      for (int b = 0; b <= a; b++) c->H[a * nv + b] += t * J[b];
    }
  }
  chol_factor(c->H, nv);
  memcpy(c->Mgrad, c->grad, sizeof(double) * nv);
  chol_solve(c->H, nv, c->Mgrad);
}

typedef struct { double alpha, cost, d0, d1; } lspoint;

static lspoint ls_eval(const solctx *c, double alpha) {
  const orc_data *d = c->d;
  double q0 = c->quadGauss[0], q1 = c->quadGauss[1], q2 = c->quadGauss[2];
  for (int i = 0; i < d->nefc; i++) {
    double x = c->jar[i] + alpha * c->jv[i], D = d->efc_D[i], jv = c->jv[i];
    /* row contributes a0 + a1*alpha-free form directly in (cost, d0, d1) at this alpha */
    switch (d->efc_type[i]) {
      case CT_EQUALITY: q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv; break;
      case CT_FRICTION_DOF: {
        double fl = d->efc_frictionloss[i], rf = d->efc_R[i] * fl;
        if (x <= -rf) { q0 += -fl * (0.5 * rf + x); q1 += -fl * jv; }
        else if (x >= rf) { q0 += -fl * (0.5 * rf - x); q1 += fl * jv; }
        else { q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv; }
        break;
      }
      default:
        if (x < 0) { q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv; }
    }
  }
  /* q0/q1/q2 were accumulated with the constraint part already evaluated AT alpha; the
     Gauss part is a polynomial in alpha */
  lspoint p;
  p.alpha = alpha;
  p.cost = q0 + alpha * c->quadGauss[1] + alpha * alpha * c->quadGauss[2];
  p.d0 = q1 + 2 * alpha * c->quadGauss[2];
  p.d1 = 2 * q2;
  if (p.d1 <= 0) p.d1 = MINVAL;
  return p;
}

static double line_search(solctx *c) {
  const orc_model *m = c->m;
  orc_data *d = c->d;
  int nv = c->nv;
  double snorm = sqrt(dotn(c->search, c->search, nv));
  if (snorm < MINVAL) return 0;
  double gtol = m->tolerance * m->ls_tolerance * snorm / c->scale;
  orc_mul_m(m, d, c->Mv, c->search);
  for (int i = 0; i < d->nefc; i++) c->jv[i] = dotn(d->efc_J + (size_t)i * nv, c->search, nv);
  c->quadGauss[0] = c->gauss;
  double g1 = 0;
  for (int k = 0; k < nv; k++) g1 += c->search[k] * (c->Ma[k] - d->qfrc_smooth[k]);
  c->quadGauss[1] = g1;
  c->quadGauss[2] = 0.5 * dotn(c->search, c->Mv, nv);

  int it = 0, maxit = m->ls_iterations;
  lspoint p0 = ls_eval(c, 0), p1, p2, pmid, p1next, p2next;
  p1 = ls_eval(c, p0.alpha - p0.d0 / p0.d1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  double dir = p1.d0 < 0 ? 1.0 : -1.0;
  int p2update = 0;
  p2 = p1;
  while (p1.d0 * dir <= -gtol && it < maxit) {
    p2 = p1; p2update = 1;
    p1 = ls_eval(c, p1.alpha - p1.d0 / p1.d1);
    it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= maxit || !p2update) return p1.alpha;
  /* bracketed: p2 before the minimum, p1 past it (along dir) */
  p2next = p1;
  p1next = ls_eval(c, p1.alpha - p1.d0 / p1.d1);
  while (it < maxit) {
    pmid = ls_eval(c, 0.5 * (p1.alpha + p2.alpha));
    it++;
    lspoint cand[3] = {p1next, p2next, pmid};
    int b1 = 0, b2 = 0;
    for (int k = 0; k < 3; k++) {
      if (fabs(cand[k].d0) < gtol) return cand[k].alpha;
      if (cand[k].d0 * dir < 0) {
        if (cand[k].alpha * dir > p2.alpha * dir && cand[k].alpha * dir < p1.alpha * dir) { p2 = cand[k]; b2 = 1; }
      } else {
        if (cand[k].alpha * dir < p1.alpha * dir && cand[k].alpha * dir > p2.alpha * dir) { p1 = cand[k]; b1 = 1; }
      }
    }
    if (!b1 && !b2) break;
    if (b1) p1next = ls_eval(c, p1.alpha - p1.d0 / p1.d1);
    if (b2) p2next = ls_eval(c, p2.alpha - p2.d0 / p2.d1);
  }
  lspoint best = p1.cost <= p2.cost ? p1 : p2;
  return best.cost < p0.cost ? best.alpha : 0.0;
}

/* mj_solNewton and mj_solCG share the primal loop (mj_solPrimal shape [EXT]): Newton searches
 * along -H^-1 grad, CG along -M^-1 grad + beta * previous direction with the Polak-Ribiere beta
 * (clamped at 0) in the M^-1 inner product */
static void sol_newton(const orc_model *m, orc_data *d) {
  int nv = m->nv, nefc = d->nefc;
  solctx c;
  c.m = m; c.d = d; c.nv = nv; c.nefc = nefc;
  double *w = d->scratch + 12 * m->nbody;
  c.jar = w; w += d->nefc_max; c.jv = w; w += d->nefc_max;
  c.Ma = w; w += nv; c.grad = w; w += nv; c.Mgrad = w; w += nv; c.search = w; w += nv;
  c.Mv = w; w += nv; c.H = w; w += nv * nv;
  c.qacc = d->qacc;
  c.scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  orc_mul_m(m, d, c.Ma, d->qacc);
  for (int i = 0; i < nefc; i++)
    c.jar[i] = dotn(d->efc_J + (size_t)i * nv, d->qacc, nv) - d->efc_aref[i];
  sol_update_constraint(&c);
  sol_update_gradient(&c);
  for (int k = 0; k < nv; k++) c.search[k] = -c.Mgrad[k];
  int iter = 0;
  double oldgrad[nv > 0 ? nv : 1], oldMgrad[nv > 0 ? nv : 1]; /* CG only */
  while (iter < m->iterations) {
    double alpha = line_search(&c);
    if (alpha == 0) break;
    for (int k = 0; k < nv; k++) { d->qacc[k] += alpha * c.search[k]; c.Ma[k] += alpha * c.Mv[k]; }
    for (int i = 0; i < nefc; i++) c.jar[i] += alpha * c.jv[i];
    double oldcost = c.cost;
    double gg_old = 0;
    if (m->solver == SOL_CG) {
      memcpy(oldgrad, c.grad, sizeof(double) * nv);
      memcpy(oldMgrad, c.Mgrad, sizeof(double) * nv);
      gg_old = dotn(oldgrad, oldMgrad, nv);
    }
    sol_update_constraint(&c);
    sol_update_gradient(&c);
    if (m->solver == SOL_CG) {
      double num = 0;
      for (int k = 0; k < nv; k++) num += c.grad[k] * (c.Mgrad[k] - oldMgrad[k]);
      double beta = num / (gg_old > MINVAL ? gg_old : MINVAL);
      if (beta < 0) beta = 0;
      for (int k = 0; k < nv; k++) c.search[k] = -c.Mgrad[k] + beta * c.search[k];
    } else
      for (int k = 0; k < nv; k++) c.search[k] = -c.Mgrad[k];
    iter++;
    double improvement = c.scale * (oldcost - c.cost);
    double gradient = c.scale * sqrt(dotn(c.grad, c.grad, nv));
    if (improvement < m->tolerance || gradient < m->tolerance) break;
  }
  d->solver_iter = iter;
}


/* ------------------------------------------------------------------ A11 mj_solPGS
 * projected Gauss-Seidel on the dual  min 0.5 f'(A+R)f + f'b,  A = J M^-1 J', b = J qacc_smooth - aref,
 * with the explicit dense matrix (mj_projectConstraint), scalar projections: equality free,
 * friction loss in [-floss, floss], limits / pyramidal contact edges >= 0 */
static void sol_pgs(const orc_model *m, orc_data *d) {
  int nv = m->nv, n = d->nefc;
  double *AR = (double *)malloc(sizeof(double) * ((size_t)n * n + (size_t)n * nv + 2 * (size_t)n + nv));
  double *W = AR + (size_t)n * n, *b = W + (size_t)n * nv, *jar = b + n, *tmp = jar + n;
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  for (int i = 0; i < n; i++) {
    memcpy(W + (size_t)i * nv, d->efc_J + (size_t)i * nv, sizeof(double) * nv);
    orc_solve_m(m, d, W + (size_t)i * nv); /* row i of J M^-1 */
  }
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) AR[(size_t)i * n + j] = dotn(W + (size_t)i * nv, d->efc_J + (size_t)j * nv, nv);
    AR[(size_t)i * n + i] += d->efc_R[i];
    b[i] = dotn(d->efc_J + (size_t)i * nv, d->qacc_smooth, nv) - d->efc_aref[i];
  }
  /* warm start: forces implied by the previous acceleration, kept only if the dual cost is < 0 */
  if (!(m->disableflags & DSBL_WARMSTART)) {
    for (int i = 0; i < n; i++)
      jar[i] = dotn(d->efc_J + (size_t)i * nv, d->qacc_warmstart, nv) - d->efc_aref[i];
    constraint_update(m, d, jar, 1);
    double c = 0;
    for (int i = 0; i < n; i++) c += d->efc_force[i] * (0.5 * dotn(AR + (size_t)i * n, d->efc_force, n) + b[i]);
    if (c > 0) memset(d->efc_force, 0, sizeof(double) * n);
  } else
    memset(d->efc_force, 0, sizeof(double) * n);
  int iter = 0;
  while (iter < m->iterations) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      double res = b[i] + dotn(AR + (size_t)i * n, d->efc_force, n), old = d->efc_force[i];
      double f = old - res / AR[(size_t)i * n + i];
      if (d->efc_type[i] == CT_FRICTION_DOF) {
        double fl = d->efc_frictionloss[i];
        f = f < -fl ? -fl : (f > fl ? fl : f);
      } else if (d->efc_type[i] != CT_EQUALITY) {
        if (f < 0) f = 0;
      }
      double delta = f - old, change = 0.5 * delta * delta * AR[(size_t)i * n + i] + delta * res;
      if (change > 1e-10) { f = old; change = 0; }
      d->efc_force[i] = f;
      improvement -= change;
    }
    iter++;
    if (improvement * scale < m->tolerance) break;
  }
  d->solver_iter = iter;
  for (int k = 0; k < nv; k++) tmp[k] = 0;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < nv; k++) tmp[k] += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
  memcpy(d->qfrc_constraint, tmp, sizeof(double) * nv);
  orc_solve_m(m, d, tmp);
  for (int k = 0; k < nv; k++) d->qacc[k] = d->qacc_smooth[k] + tmp[k];
  free(AR);
}

void orc_fwd_constraint(const orc_model *m, orc_data *d) {
  int nv = m->nv;
  if (d->nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    d->solver_iter = 0;
    return;
  }
  if (m->solver == SOL_PGS) {
    sol_pgs(m, d);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
    return;
  }
  /* warm start: keep the previous acceleration only if it is cheaper than qacc_smooth */
  if (!(m->disableflags & DSBL_WARMSTART)) {
    double cw = orc_cost(m, d, d->qacc_warmstart), cs = orc_cost(m, d, d->qacc_smooth);
    memcpy(d->qacc, cw > cs ? d->qacc_smooth : d->qacc_warmstart, sizeof(double) * nv);
  } else
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
  sol_newton(m, d); /* Newton is the reference default (no <option> in the XML); SOL_CG takes the CG branches */
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
}

/* ------------------------------------------------------------------ pipeline */
static void forward_skip_sensor(const orc_model *m, orc_data *d) {
  orc_kinematics(m, d);
  orc_com_pos(m, d);
  orc_crb(m, d);
  orc_factor_m(m, d);
  orc_collision(m, d);
  orc_make_constraint(m, d);
  orc_fwd_velocity(m, d);
  reference_constraint(m, d);
  orc_fwd_actuation(m, d);
  orc_fwd_acceleration(m, d);
  orc_fwd_constraint(m, d);
}
void orc_forward(const orc_model *m, orc_data *d) { forward_skip_sensor(m, d); }

static int bad(double x) { return isnan(x) || x > MAXVAL || x < -MAXVAL; }
/* A14 mj_checkPos / Vel / Acc: on NaN or |x|>1e10 record a warning and reset */
static int check(const orc_model *m, orc_data *d, const double *x, int n, int bit) {
  for (int i = 0; i < n; i++)
    if (bad(x[i])) {
      int w = d->warning | (1 << bit);
      orc_reset(m, d);
      d->warning = w;
      return 1;
    }
  return 0;
}

/* A12 mj_Euler: joint damping integrated implicitly in velocity */
static void euler(const orc_model *m, orc_data *d) {
  int nv = m->nv;
  double h = m->timestep, *qacc = d->scratch + 12 * m->nbody, *MM = qacc + nv, *dinv = MM + m->nM;
  int damped = 0;
  for (int i = 0; i < nv; i++) damped |= m->dof_damping[i] > 0;
  if (damped) {
    memcpy(MM, d->qM, sizeof(double) * m->nM);
    for (int i = 0; i < nv; i++) {
      MM[m->dof_Madr[i]] += h * m->dof_damping[i];
      qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    }
    factor_sparse(m, MM, dinv);
    solve_sparse(m, MM, dinv, qacc);
  } else
    memcpy(qacc, d->qacc, sizeof(double) * nv);
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) d->qpos[m->jnt_qposadr[j]] += h * d->qvel[m->jnt_dofadr[j]];
  d->time += h;
}

/* A13 mj_RungeKutta(4): classical tableau, no implicit damping */
static void rk4(const orc_model *m, orc_data *d) {
  int nv = m->nv, nq = m->njnt;
  double h = m->timestep, t0 = d->time;
  static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}}, B[4] = {1 / 6., 1 / 3., 1 / 3., 1 / 6.},
                      C[3] = {0.5, 0.5, 1.0};
  double *buf = (double *)malloc(sizeof(double) * (nq + nv + 8 * nv));
  double *q0 = buf, *v0 = q0 + nq, *Fv = v0 + nv, *Fa = Fv + 4 * nv;
  memcpy(q0, d->qpos, sizeof(double) * nq);
  memcpy(v0, d->qvel, sizeof(double) * nv);
  memcpy(Fv, d->qvel, sizeof(double) * nv);
  memcpy(Fa, d->qacc, sizeof(double) * nv);
  for (int i = 1; i < 4; i++) {
    for (int k = 0; k < nv; k++) {
      double dv = 0, da = 0;
      for (int j = 0; j < i; j++) { dv += A[i - 1][j] * Fv[j * nv + k]; da += A[i - 1][j] * Fa[j * nv + k]; }
      d->qpos[k] = q0[k] + h * dv; /* nq == nv: hinge / slide only */
      d->qvel[k] = v0[k] + h * da;
    }
    d->time = t0 + C[i - 1] * h;
    forward_skip_sensor(m, d);
    memcpy(Fv + i * nv, d->qvel, sizeof(double) * nv);
    memcpy(Fa + i * nv, d->qacc, sizeof(double) * nv);
  }
  for (int k = 0; k < nv; k++) {
    double dv = 0, da = 0;
    for (int j = 0; j < 4; j++) { dv += B[j] * Fv[j * nv + k]; da += B[j] * Fa[j * nv + k]; }
    d->qvel[k] = v0[k] + h * da;
    d->qpos[k] = q0[k] + h * dv;
  }
  d->time = t0 + h;
  free(buf);
}

void orc_step(const orc_model *m, orc_data *d) {
  check(m, d, d->qpos, m->njnt, 0);
  check(m, d, d->qvel, m->nv, 1);
  orc_forward(m, d);
  if (check(m, d, d->qacc, m->nv, 2)) orc_forward(m, d);
  if (m->integrator == INT_RK4) rk4(m, d);
  else euler(m, d);
}
