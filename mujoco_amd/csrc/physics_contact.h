// physics_contact.h — contact rows (A7) and the Newton solve with a dense Hessian (A11) for
// environments that have contacts.  Serial layout of physics_dev.h (one environment per lane,
// [row][E] arrays); correctness-first: this is the path BASELINE.json configs[2] runs on.
//
// Contact rows: pyramidal cone, condim 3 -> 4 rows per contact  J_n +- mu J_t1, J_n +- mu J_t2
// with J = frame * (Jac(body2, pos) - Jac(body1, pos)); all four rows share
// R = 2 mu^2 R_n (impratio = 1); condim/friction/margin: reference src/Mujoco_XML.py:70.
// Contacts with dist >= margin - gap stay in the contact list but get no rows.
#pragma once
#include "physics_collide.h"

struct ConData {
  // [4*maxcon][...] contact rows
  double *cJ;      // [(row*nv + dof)][E]
  double *cD, *caref, *cjar, *cjv, *cforce;
  double *Hd;      // dense lower-triangular Hessian [(nv*(nv+1)/2)][E]
  double *cW;      // PGS only: rows of J_c M^-1, [(row*nv + dof)][E]
  int *ncrow;      // [1][E] contact rows of this env
  int *con_efc;    // [maxcon][E] first row of each contact or -1
};

__device__ __forceinline__ int tri_idx(int i, int j) { return i * (i + 1) / 2 + j; }  // j <= i

// add sign * (n,t1,t2 projections of the point Jacobian of body b at `pos`) into the 4 rows
__device__ inline void contact_jac_body(const DevModel& m, const DevData& d, const ConData& k, long e, int b,
                                        const V3& pos, const double* fr, double mu, int row0, int dim, double sign) {
  const long E = d.E;
  const int r = m.body_rootid[b];
  const V3 off = {pos.x - AT(d.rootcom, 3 * r), pos.y - AT(d.rootcom, 3 * r + 1), pos.z - AT(d.rootcom, 3 * r + 2)};
  while (b > 0 && m.body_dofnum[b] == 0) b = m.body_parentid[b];
  if (b == 0) return;
  for (int i = m.body_dofadr[b] + m.body_dofnum[b] - 1; i >= 0; i = m.dof_parentid[i]) {
    const V3 ang = {AT(d.cdof, 6 * i), AT(d.cdof, 6 * i + 1), AT(d.cdof, 6 * i + 2)};
    const V3 lin = {AT(d.cdof, 6 * i + 3), AT(d.cdof, 6 * i + 4), AT(d.cdof, 6 * i + 5)};
    const V3 c = cross(ang, off);
    const V3 w = {lin.x + c.x, lin.y + c.y, lin.z + c.z};
    const double jn = fr[0] * w.x + fr[1] * w.y + fr[2] * w.z;
    if (dim == 1) {
      AT(k.cJ, (size_t)row0 * m.nv + i) += sign * jn;
    } else {
      const double j1 = fr[3] * w.x + fr[4] * w.y + fr[5] * w.z, j2 = fr[6] * w.x + fr[7] * w.y + fr[8] * w.z;
      AT(k.cJ, (size_t)(row0 + 0) * m.nv + i) += sign * (jn + mu * j1);
      AT(k.cJ, (size_t)(row0 + 1) * m.nv + i) += sign * (jn - mu * j1);
      AT(k.cJ, (size_t)(row0 + 2) * m.nv + i) += sign * (jn + mu * j2);
      AT(k.cJ, (size_t)(row0 + 3) * m.nv + i) += sign * (jn - mu * j2);
    }
  }
}

// mj_makeConstraint (contact part) + impedance + reference acceleration; returns #contact rows
__device__ inline int make_contact_rows(const DevModel& m, const ColModel& cm, const DevData& d, const ColData& c,
                                        const ConData& k, long e) {
  const long E = d.E;
  const int ncon = AT(c.ncon, 0);
  int nrow = 0;
  for (int ci = 0; ci < ncon; ci++) {
    const int p = AT(c.con_pair, ci), g1 = cm.pair_geom1[p], g2 = cm.pair_geom2[p];
    const double margin = fmax(cm.geom_margin[g1], cm.geom_margin[g2]) - fmax(cm.geom_gap[g1], cm.geom_gap[g2]);
    const double dist = AT(c.con_dist, ci);
    AT(k.con_efc, ci) = -1;
    if (dist >= margin) continue;
    const int dim = cm.geom_condim[g1] > cm.geom_condim[g2] ? cm.geom_condim[g1] : cm.geom_condim[g2];
    const int nr = dim == 1 ? 1 : 4;
    const double mu = fmax(cm.geom_friction[3 * g1], cm.geom_friction[3 * g2]);
    const int b1 = cm.geom_bodyid[g1], b2 = cm.geom_bodyid[g2];
    double fr[9];
    for (int q = 0; q < 9; q++) fr[q] = AT(c.con_frame, 9 * ci + q);
    const V3 pos = {AT(c.con_pos, 3 * ci), AT(c.con_pos, 3 * ci + 1), AT(c.con_pos, 3 * ci + 2)};
    for (int r = 0; r < nr; r++)
      for (int i = 0; i < m.nv; i++) AT(k.cJ, (size_t)(nrow + r) * m.nv + i) = 0;
    contact_jac_body(m, d, k, e, b2, pos, fr, mu, nrow, dim, 1.0);
    contact_jac_body(m, d, k, e, b1, pos, fr, mu, nrow, dim, -1.0);
    // solref / solimp mixed by solmix
    const double s1 = cm.geom_solmix[g1], s2 = cm.geom_solmix[g2];
    double mix;
    if (s1 >= BS_MINVAL && s2 >= BS_MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < BS_MINVAL && s2 < BS_MINVAL) mix = 0.5;
    else mix = s1 < BS_MINVAL ? 0.0 : 1.0;
    double solref[2], solimp[5];
    for (int q = 0; q < 2; q++) solref[q] = mix * cm.geom_solref[2 * g1 + q] + (1 - mix) * cm.geom_solref[2 * g2 + q];
    for (int q = 0; q < 5; q++) solimp[q] = mix * cm.geom_solimp[5 * g1 + q] + (1 - mix) * cm.geom_solimp[5 * g2 + q];
    double imp, K, B;
    impedance(solimp, dist, margin, imp);
    kb_from_solref(m, solref, solimp, K, B);
    const double tran = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
    const double diag = dim == 1 ? tran : tran + mu * mu * tran;
    double R = fmax(BS_MINVAL, (1 - imp) * diag / imp);
    if (dim > 1) {
      const double mue = mu / sqrt(fmax(BS_MINVAL, m.impratio));
      R = fmax(BS_MINVAL, 2 * mue * mue * R);
    }
    for (int r = 0; r < nr; r++) {
      double vel = 0;
      for (int i = 0; i < m.nv; i++) vel += AT(k.cJ, (size_t)(nrow + r) * m.nv + i) * AT(d.qvel, i);
      AT(k.cD, nrow + r) = 1.0 / R;
      AT(k.caref, nrow + r) = -B * vel - K * imp * (dist - margin);
    }
    AT(k.con_efc, ci) = nrow;
    nrow += nr;
  }
  AT(k.ncrow, 0) = nrow;
  return nrow;
}

__device__ inline void crows_residual(const DevModel& m, const DevData& d, const ConData& k, long e, int ncrow,
                                      const double* a, double* out, bool sub_aref) {
  const long E = d.E;
  for (int r = 0; r < ncrow; r++) {
    double s = 0;
    for (int i = 0; i < m.nv; i++) s += AT(k.cJ, (size_t)r * m.nv + i) * AT(a, i);
    AT(out, r) = sub_aref ? s - AT(k.caref, r) : s;
  }
}
template <bool WRITE>
__device__ inline double crows_update(const DevModel& m, const DevData& d, const ConData& k, long e, int ncrow,
                                      const double* jar) {
  const long E = d.E;
  double cost = 0;
  for (int r = 0; r < ncrow; r++) {
    const double x = AT(jar, r), D = AT(k.cD, r);
    double f = 0;
    if (x < 0) { f = -D * x; cost += 0.5 * D * x * x; }
    if (WRITE) {
      AT(k.cforce, r) = f;
      if (f != 0)
        for (int i = 0; i < m.nv; i++) AT(d.qfrc_constraint, i) += AT(k.cJ, (size_t)r * m.nv + i) * f;
    }
  }
  return cost;
}
__device__ inline double cost_at_c(const DevModel& m, const DevData& d, const ConData& k, long e, int ncrow,
                                   const double* a) {
  const double c0 = cost_at(m, d, e, a);  // fills Ma, jar (fixed rows)
  crows_residual(m, d, k, e, ncrow, a, k.cjar, true);
  return c0 + crows_update<false>(m, d, k, e, ncrow, k.cjar);
}

// dense H = M + J^T D_active J, Cholesky, search = -H^-1 grad
// `hs`: optional LDS block of (nv(nv+1)/2 + nv) x 64 doubles ([entry][lane]); when given, the
// Hessian and the staged Jacobian row live there instead of the global arena (the dense
// factorisation is otherwise bound by L2/HBM traffic).
#define HDX(q) (*(hs ? hs + (size_t)(q) * 64 + threadIdx.x : &AT(k.Hd, q)))
__device__ inline double update_gradient_dense(const DevModel& m, const DevData& d, const ConData& k, long e,
                                               int ncrow, double* hs, bool first = true) {
  const long E = d.E;
  const int nv = m.nv;
  if (m.solver == 1) return cg_gradient(m, d, e, first);
  double gg = 0;
  for (int i = 0; i < nv; i++) {
    const double g = AT(d.Ma, i) - AT(d.qfrc_smooth, i) - AT(d.qfrc_constraint, i);
    AT(d.grad, i) = g;
    AT(d.Mgrad, i) = g;
    gg += g * g;
  }
  for (int q = 0; q < nv * (nv + 1) / 2; q++) HDX(q) = 0;
  for (int i = 0; i < nv; i++) {
    int adr = m.dof_Madr[i];
    for (int j = i; j >= 0; j = m.dof_parentid[j], adr++) HDX(tri_idx(i, j)) = AT(d.qM, adr);
  }
  for (int r = 0; r < m.neq; r++) {
    const double D = AT(d.efc_D, r);
    const int a = m.eq_dof1[r], b = m.eq_dof2[r];
    HDX(tri_idx(a, a)) += D;
    if (b >= 0) {
      const double c = AT(d.eq_coef, r);
      HDX(tri_idx(b, b)) += D * c * c;
      HDX(a > b ? tri_idx(a, b) : tri_idx(b, a)) -= D * c;
    }
  }
  for (int q = 0; q < m.nfric; q++) {
    const int r = m.neq + q, i = m.fric_dof[q];
    const double x = AT(d.jar, r), rf = AT(d.fric_rf, q);
    if (x > -rf && x < rf) HDX(tri_idx(i, i)) += AT(d.efc_D, r);
  }
  for (int q = 0; q < m.nlim; q++) {
    const int r = m.neq + m.nfric + 2 * q, i = m.jnt_dofadr[m.lim_jnt[q]];
    double add = 0;
    if (AT(d.jar, r) < 0) add += AT(d.efc_D, r);
    if (AT(d.jar, r + 1) < 0) add += AT(d.efc_D, r + 1);
    HDX(tri_idx(i, i)) += add;
  }
  for (int r = 0; r < ncrow; r++) {
    if (!(AT(k.cjar, r) < 0)) continue;
    const double D = AT(k.cD, r);
    if (hs) {
      double* row = hs + (size_t)(nv * (nv + 1) / 2) * 64 + threadIdx.x;
      for (int a = 0; a < nv; a++) row[(size_t)a * 64] = AT(k.cJ, (size_t)r * nv + a);
      for (int a = 0; a < nv; a++) {
        const double Ja = row[(size_t)a * 64];
        if (Ja == 0) continue;
        const double t = D * Ja;
        for (int b = 0; b <= a; b++) HDX(tri_idx(a, b)) += t * row[(size_t)b * 64];
      }
      continue;
    }
    for (int a = 0; a < nv; a++) {
      const double Ja = AT(k.cJ, (size_t)r * nv + a);
      if (Ja == 0) continue;
      const double t = D * Ja;
      for (int b = 0; b <= a; b++) HDX(tri_idx(a, b)) += t * AT(k.cJ, (size_t)r * nv + b);
    }
  }
  // Cholesky H = L L^T in place (MuJoCo's MINVAL floor on the pivots)
  for (int j = 0; j < nv; j++) {
    double s = HDX(tri_idx(j, j));
    for (int q = 0; q < j; q++) { const double l = HDX(tri_idx(j, q)); s -= l * l; }
    if (s < BS_MINVAL) s = BS_MINVAL;
    const double l = sqrt(s);
    HDX(tri_idx(j, j)) = l;
    for (int i = j + 1; i < nv; i++) {
      double t = HDX(tri_idx(i, j));
      for (int q = 0; q < j; q++) t -= HDX(tri_idx(i, q)) * HDX(tri_idx(j, q));
      HDX(tri_idx(i, j)) = t / l;
    }
  }
  for (int i = 0; i < nv; i++) {
    double s = AT(d.Mgrad, i);
    for (int q = 0; q < i; q++) s -= HDX(tri_idx(i, q)) * AT(d.Mgrad, q);
    AT(d.Mgrad, i) = s / HDX(tri_idx(i, i));
  }
  for (int i = nv - 1; i >= 0; i--) {
    double s = AT(d.Mgrad, i);
    for (int q = i + 1; q < nv; q++) s -= HDX(tri_idx(q, i)) * AT(d.Mgrad, q);
    AT(d.Mgrad, i) = s / HDX(tri_idx(i, i));
  }
  for (int i = 0; i < nv; i++) AT(d.search, i) = -AT(d.Mgrad, i);
  return gg;
}

__device__ inline LsPoint ls_eval_c(const DevModel& m, const DevData& d, const ConData& k, long e, int ncrow,
                                    double alpha, double g0, double g1, double g2) {
  const long E = d.E;
  // fixed rows evaluated by the contact-free routine with zero Gauss part, then contact rows
  LsPoint p = ls_eval(m, d, e, alpha, 0.0, 0.0, 0.0);
  double q0 = p.cost, q1 = p.d0, q2 = p.d1 > BS_MINVAL ? 0.5 * p.d1 : 0.0;
  for (int r = 0; r < ncrow; r++) {
    const double jv = AT(k.cjv, r), x = AT(k.cjar, r) + alpha * jv, D = AT(k.cD, r);
    if (x < 0) { q0 += 0.5 * D * x * x; q1 += D * x * jv; q2 += 0.5 * D * jv * jv; }
  }
  p.alpha = alpha;
  p.cost = g0 + q0 + alpha * g1 + alpha * alpha * g2;
  p.d0 = g1 + q1 + 2 * alpha * g2;
  p.d1 = 2 * (g2 + q2);
  if (p.d1 <= 0) p.d1 = BS_MINVAL;
  return p;
}

__device__ inline double line_search_c(const DevModel& m, const DevData& d, const ConData& k, long e, int ncrow,
                                       double gauss, double scale) {
  const long E = d.E;
  double ss = 0;
  for (int i = 0; i < m.nv; i++) ss += AT(d.search, i) * AT(d.search, i);
  const double snorm = sqrt(ss);
  if (snorm < BS_MINVAL) return 0;
  const double gtol = m.tolerance * m.ls_tolerance * snorm / scale;
  mul_m(m, d.qM, d.Mv, d.search, E, e);
  rows_residual(m, d, e, d.search, d.jv, false);
  crows_residual(m, d, k, e, ncrow, d.search, k.cjv, false);
  double g1 = 0, g2 = 0;
  for (int i = 0; i < m.nv; i++) {
    const double s = AT(d.search, i);
    g1 += s * (AT(d.Ma, i) - AT(d.qfrc_smooth, i));
    g2 += s * AT(d.Mv, i);
  }
  g2 *= 0.5;
  int it = 0;
  const int maxit = m.ls_iterations;
  const LsPoint p0 = ls_eval_c(m, d, k, e, ncrow, 0.0, gauss, g1, g2);
  LsPoint p1 = ls_eval_c(m, d, k, e, ncrow, p0.alpha - p0.d0 / p0.d1, gauss, g1, g2);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  const double dir = p1.d0 < 0 ? 1.0 : -1.0;
  bool p2update = false;
  LsPoint p2 = p1;
  while (p1.d0 * dir <= -gtol && it < maxit) {
    p2 = p1; p2update = true;
    p1 = ls_eval_c(m, d, k, e, ncrow, p1.alpha - p1.d0 / p1.d1, gauss, g1, g2);
    it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= maxit || !p2update) return p1.alpha;
  LsPoint p2next = p1;
  LsPoint p1next = ls_eval_c(m, d, k, e, ncrow, p1.alpha - p1.d0 / p1.d1, gauss, g1, g2);
  while (it < maxit) {
    const LsPoint pmid = ls_eval_c(m, d, k, e, ncrow, 0.5 * (p1.alpha + p2.alpha), gauss, g1, g2);
    it++;
    bool b1 = false, b2 = false;
    for (int q = 0; q < 3; q++) {
      const LsPoint c = q == 0 ? p1next : (q == 1 ? p2next : pmid);
      if (fabs(c.d0) < gtol) return c.alpha;
      if (c.d0 * dir < 0) {
        if (c.alpha * dir > p2.alpha * dir && c.alpha * dir < p1.alpha * dir) { p2 = c; b2 = true; }
      } else {
        if (c.alpha * dir < p1.alpha * dir && c.alpha * dir > p2.alpha * dir) { p1 = c; b1 = true; }
      }
    }
    if (!b1 && !b2) break;
    if (b1) p1next = ls_eval_c(m, d, k, e, ncrow, p1.alpha - p1.d0 / p1.d1, gauss, g1, g2);
    if (b2) p2next = ls_eval_c(m, d, k, e, ncrow, p2.alpha - p2.d0 / p2.d1, gauss, g1, g2);
  }
  const LsPoint best = p1.cost <= p2.cost ? p1 : p2;
  return best.cost < p0.cost ? best.alpha : 0.0;
}

__device__ inline void stage_pgs(const DevModel& m, const DevData& d, const ConData* kp, long e, int ncrow);

// A7 + A11 with contacts
__device__ inline void stage_constraint_c(const DevModel& m, const ColModel& cm, const DevData& d, const ColData& c,
                                          const ConData& k, long e, double* hs) {
  const long E = d.E;
  int nact = 0, ncrow = 0;
  if (!(m.disableflags & BS_DSBL_CONSTRAINT)) {
    nact = make_rows(m, d, e);
    ncrow = make_contact_rows(m, cm, d, c, k, e);
  } else AT(k.ncrow, 0) = 0;
  AT(d.nefc, 0) = nact + ncrow;
  if (nact + ncrow == 0) {
    for (int i = 0; i < m.nv; i++) {
      const double a = AT(d.qacc_smooth, i);
      AT(d.qacc, i) = a; AT(d.qacc_warmstart, i) = a; AT(d.qfrc_constraint, i) = 0;
    }
    AT(d.solver_iter, 0) = 0;
    return;
  }
  if (m.solver == 0) {
    stage_pgs(m, d, &k, e, ncrow);
    return;
  }
  bool use_smooth = true;
  if (!(m.disableflags & BS_DSBL_WARMSTART)) {
    const double cw = cost_at_c(m, d, k, e, ncrow, d.qacc_warmstart);
    const double cs = cost_at_c(m, d, k, e, ncrow, d.qacc_smooth);
    use_smooth = cw > cs;
  }
  for (int i = 0; i < m.nv; i++) AT(d.qacc, i) = use_smooth ? AT(d.qacc_smooth, i) : AT(d.qacc_warmstart, i);
  const double scale = 1.0 / (m.meaninertia * (m.nv > 1 ? m.nv : 1));
  mul_m(m, d.qM, d.Ma, d.qacc, E, e);
  rows_residual(m, d, e, d.qacc, d.jar, true);
  crows_residual(m, d, k, e, ncrow, d.qacc, k.cjar, true);
  auto update = [&](double& gauss) {
    double g = 0;
    for (int i = 0; i < m.nv; i++)
      g += (AT(d.Ma, i) - AT(d.qfrc_smooth, i)) * (AT(d.qacc, i) - AT(d.qacc_smooth, i));
    gauss = 0.5 * g;
    double cost = gauss + rows_update<true>(m, d, e, d.jar);
    cost += crows_update<true>(m, d, k, e, ncrow, k.cjar);
    return cost;
  };
  double gauss;
  double cost = update(gauss);
  update_gradient_dense(m, d, k, e, ncrow, hs);
  int iter = 0;
  while (iter < m.iterations) {
    const double alpha = line_search_c(m, d, k, e, ncrow, gauss, scale);
    if (alpha == 0) break;
    for (int i = 0; i < m.nv; i++) {
      AT(d.qacc, i) += alpha * AT(d.search, i);
      AT(d.Ma, i) += alpha * AT(d.Mv, i);
    }
    for (int r = 0; r < m.nrow; r++) AT(d.jar, r) += alpha * AT(d.jv, r);
    for (int r = 0; r < ncrow; r++) AT(k.cjar, r) += alpha * AT(k.cjv, r);
    const double oldcost = cost;
    cost = update(gauss);
    const double gg = update_gradient_dense(m, d, k, e, ncrow, hs, false);
    iter++;
    const double improvement = scale * (oldcost - cost), gradient = scale * sqrt(gg);
    if (improvement < m.tolerance || gradient < m.tolerance) break;
  }
  AT(d.solver_iter, 0) = iter;
  for (int i = 0; i < m.nv; i++) AT(d.qacc_warmstart, i) = AT(d.qacc, i);
}


// ------------------------------------------------------------------ A11 mj_solPGS (matrix-free)
// Projected Gauss-Seidel on the dual.  Instead of the dense nefc x nefc matrix A + R the kernel
// keeps w_i = M^-1 J_i^T per row and the running acceleration a = qacc_smooth + sum_j f_j w_j, so
// the row residual is  J_i a - aref_i + R_i f_i  and a force change costs one axpy of length nv.
// Same iteration as mj_solPGS (row order: equality, friction loss, limits, contacts).
__device__ inline double frow_dot(const DevModel& m, const DevData& d, long e, int r, const double* v) {
  const long E = d.E;
  if (r < m.neq) {
    double s = AT(v, m.eq_dof1[r]);
    if (m.eq_dof2[r] >= 0) s -= AT(d.eq_coef, r) * AT(v, m.eq_dof2[r]);
    return s;
  }
  if (r < m.neq + m.nfric) return AT(v, m.fric_dof[r - m.neq]);
  const int q = r - m.neq - m.nfric;
  const double s = AT(v, m.jnt_dofadr[m.lim_jnt[q >> 1]]);
  return (q & 1) ? -s : s;
}
__device__ inline void frow_scatter(const DevModel& m, const DevData& d, long e, int r, double* v, double f) {
  const long E = d.E;  // v += f * J_r^T
  if (r < m.neq) {
    AT(v, m.eq_dof1[r]) += f;
    if (m.eq_dof2[r] >= 0) AT(v, m.eq_dof2[r]) -= AT(d.eq_coef, r) * f;
  } else if (r < m.neq + m.nfric) AT(v, m.fric_dof[r - m.neq]) += f;
  else {
    const int q = r - m.neq - m.nfric;
    AT(v, m.jnt_dofadr[m.lim_jnt[q >> 1]]) += (q & 1) ? -f : f;
  }
}

__device__ inline void stage_pgs(const DevModel& m, const DevData& d, const ConData* kp, long e, int ncrow) {
  const long E = d.E;
  const int nv = m.nv, nrow = m.nrow;
  const double scale = 1.0 / (m.meaninertia * (nv > 1 ? nv : 1));
  // w_i = M^-1 J_i^T and diagonal of A + R
  for (int r = 0; r < nrow; r++) {
    if (!(AT(d.efc_D, r) > 0)) continue;
    for (int i = 0; i < nv; i++) AT(d.Mgrad, i) = 0;
    frow_scatter(m, d, e, r, d.Mgrad, 1.0);
    solve_sparse(m, d.qLD, d.qLDiagInv, d.Mgrad, E, e);
    for (int i = 0; i < nv; i++) AT(d.pgs_W, (size_t)r * nv + i) = AT(d.Mgrad, i);
    AT(d.jv, r) = frow_dot(m, d, e, r, d.Mgrad) + 1.0 / AT(d.efc_D, r);
  }
  for (int r = 0; r < ncrow; r++) {
    for (int i = 0; i < nv; i++) AT(d.Mgrad, i) = AT(kp->cJ, (size_t)r * nv + i);
    solve_sparse(m, d.qLD, d.qLDiagInv, d.Mgrad, E, e);
    double s = 0;
    for (int i = 0; i < nv; i++) {
      AT(kp->cW, (size_t)r * nv + i) = AT(d.Mgrad, i);
      s += AT(kp->cJ, (size_t)r * nv + i) * AT(d.Mgrad, i);
    }
    AT(kp->cjv, r) = s + 1.0 / AT(kp->cD, r);
  }
  // warm start from the previous acceleration's implied forces, kept if the dual cost is <= 0
  bool zero = true;
  if (!(m.disableflags & BS_DSBL_WARMSTART)) {
    rows_residual(m, d, e, d.qacc_warmstart, d.jar, true);
    rows_update<true>(m, d, e, d.jar);
    if (ncrow) {
      crows_residual(m, d, *kp, e, ncrow, d.qacc_warmstart, kp->cjar, true);
      crows_update<true>(m, d, *kp, e, ncrow, kp->cjar);
    }
    for (int i = 0; i < nv; i++) AT(d.Mgrad, i) = AT(d.qfrc_constraint, i);
    solve_sparse(m, d.qLD, d.qLDiagInv, d.Mgrad, E, e);  // M^-1 J^T f
    double c = 0;
    for (int r = 0; r < nrow; r++) {
      if (!(AT(d.efc_D, r) > 0)) { AT(d.efc_force, r) = 0; continue; }
      const double f = AT(d.efc_force, r);
      c += f * (0.5 * (frow_dot(m, d, e, r, d.Mgrad) + f / AT(d.efc_D, r)) +
                (frow_dot(m, d, e, r, d.qacc_smooth) - AT(d.efc_aref, r)));
    }
    for (int r = 0; r < ncrow; r++) {
      const double f = AT(kp->cforce, r);
      double Af = 0, Ja = 0;
      for (int i = 0; i < nv; i++) {
        const double J = AT(kp->cJ, (size_t)r * nv + i);
        Af += J * AT(d.Mgrad, i); Ja += J * AT(d.qacc_smooth, i);
      }
      c += f * (0.5 * (Af + f / AT(kp->cD, r)) + (Ja - AT(kp->caref, r)));
    }
    zero = c > 0;
    if (!zero)
      for (int i = 0; i < nv; i++) AT(d.qacc, i) = AT(d.qacc_smooth, i) + AT(d.Mgrad, i);
  }
  if (zero) {
    for (int r = 0; r < nrow; r++) AT(d.efc_force, r) = 0;
    for (int r = 0; r < ncrow; r++) AT(kp->cforce, r) = 0;
    for (int i = 0; i < nv; i++) AT(d.qacc, i) = AT(d.qacc_smooth, i);
  }
  int iter = 0;
  while (iter < m.iterations) {
    double improvement = 0;
    for (int r = 0; r < nrow; r++) {
      const double D = AT(d.efc_D, r);
      if (!(D > 0)) continue;
      const double old = AT(d.efc_force, r), ARii = AT(d.jv, r);
      const double res = frow_dot(m, d, e, r, d.qacc) - AT(d.efc_aref, r) + old / D;
      double f = old - res / ARii;
      if (r >= m.neq && r < m.neq + m.nfric) {
        const double fl = m.dof_frictionloss[m.fric_dof[r - m.neq]];
        f = f < -fl ? -fl : (f > fl ? fl : f);
      } else if (r >= m.neq + m.nfric) {
        if (f < 0) f = 0;
      }
      double delta = f - old, change = 0.5 * delta * delta * ARii + delta * res;
      if (change > 1e-10) { f = old; delta = 0; change = 0; }
      if (delta != 0) {
        AT(d.efc_force, r) = f;
        for (int i = 0; i < nv; i++) AT(d.qacc, i) += delta * AT(d.pgs_W, (size_t)r * nv + i);
      }
      improvement -= change;
    }
    for (int r = 0; r < ncrow; r++) {
      const double D = AT(kp->cD, r), old = AT(kp->cforce, r), ARii = AT(kp->cjv, r);
      double Ja = 0;
      for (int i = 0; i < nv; i++) Ja += AT(kp->cJ, (size_t)r * nv + i) * AT(d.qacc, i);
      const double res = Ja - AT(kp->caref, r) + old / D;
      double f = old - res / ARii;
      if (f < 0) f = 0;
      double delta = f - old, change = 0.5 * delta * delta * ARii + delta * res;
      if (change > 1e-10) { f = old; delta = 0; change = 0; }
      if (delta != 0) {
        AT(kp->cforce, r) = f;
        for (int i = 0; i < nv; i++) AT(d.qacc, i) += delta * AT(kp->cW, (size_t)r * nv + i);
      }
      improvement -= change;
    }
    iter++;
    if (improvement * scale < m.tolerance) break;
  }
  AT(d.solver_iter, 0) = iter;
  // map back: qfrc_constraint = J^T f, qacc = qacc_smooth + M^-1 qfrc_constraint
  for (int i = 0; i < nv; i++) AT(d.qfrc_constraint, i) = 0;
  for (int r = 0; r < nrow; r++) {
    const double f = AT(d.efc_force, r);
    if (f != 0) frow_scatter(m, d, e, r, d.qfrc_constraint, f);
  }
  for (int r = 0; r < ncrow; r++) {
    const double f = AT(kp->cforce, r);
    if (f != 0)
      for (int i = 0; i < nv; i++) AT(d.qfrc_constraint, i) += AT(kp->cJ, (size_t)r * nv + i) * f;
  }
  for (int i = 0; i < nv; i++) AT(d.Mgrad, i) = AT(d.qfrc_constraint, i);
  solve_sparse(m, d.qLD, d.qLDiagInv, d.Mgrad, E, e);
  for (int i = 0; i < nv; i++) {
    const double a = AT(d.qacc_smooth, i) + AT(d.Mgrad, i);
    AT(d.qacc, i) = a;
    AT(d.qacc_warmstart, i) = a;
  }
}

// contact-free sims with solver = PGS (serial kernels)
__device__ inline void stage_constraint_pgs(const DevModel& m, const DevData& d, long e) {
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
  stage_pgs(m, d, nullptr, e, 0);
}
