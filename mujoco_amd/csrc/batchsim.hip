// batchsim.hip — host side of libbatchsim.so: the C-ABI of include/batchsim.h over the
// gfx950 kernels in physics_dev.h.  There is no CPU fallback: without a HIP device every
// entry point that needs one fails with BS_ERR_NODEVICE.
#include "../../include/batchsim.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include <cstdlib>

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(BS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                 \
  } while (0)

extern "C" int bs_version(void) { return BS_VERSION; }
extern "C" const char* bs_last_error(void) { return g_err.c_str(); }
extern "C" int bs_device_count(int* n) {
  if (!n) return fail(BS_ERR_INVALID, "bs_device_count: null argument");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
  *n = c;
  return BS_OK;
}

// ------------------------------------------------------------------ model
// Host copy of everything (validated), uploaded lazily per device.
struct HostModel {
  bs_model_desc s;  // scalars only are meaningful; pointers are replaced by the vectors below
  std::vector<std::vector<int32_t>> iv;
  std::vector<std::vector<double>> dv;
};

struct DeviceImage {
  int device = -1;
  void* blob = nullptr;
  DevModel dm;
  TpModel tp[2];  // 32-lane and wide tables
  ColModel cm;
};

struct bs_model {
  bs_model_desc d;  // pointers point into the vectors below
  std::vector<int32_t> ibuf;
  std::vector<double> dbuf;
  // derived fixed constraint-row layout
  std::vector<int32_t> eq_dof1, eq_dof2, eq_jnt1, eq_jnt2, eq_Hadr, fric_dof, lim_jnt;
  std::vector<double> eq_data, eq_solref, eq_solimp;
  int damped = 0, hess_sparse = 1;
  // tree-parallel kernels (physics_tp.h): per-lane tables and eligibility, for the 32-lane build (two environments
  // per wavefront, nbody <= 31) and the wide 64-lane build (one environment per wavefront, nbody <= 63)
  struct TpHost {
    int lpe = 32;
    int tp_ok = 0;
    std::string tp_why;
    std::vector<int32_t> tp_table;  // [TPI_TOTAL][lpe]
    std::vector<double> tp_dtable;  // [TPD_TOTAL][lpe]
    std::vector<double> tp_ktable;  // [lpe][TPK_N]
    std::vector<int32_t> tp_pgs;    // PGS sweep order (TpModel::pgs_order)
    TpModel tp;
  } tpv[2];
  std::vector<DeviceImage> images;
};

// Decide whether the tree-parallel kernel can run this model and build its lane table.
static void build_tp(bs_model* mdl, int variant) {
  const bs_model_desc& s = mdl->d;
  bs_model::TpHost* const m = &mdl->tpv[variant];
  const int LPE = variant ? 64 : 32;  // lanes per environment of this variant's kernels (their compile-time TP_LPE)
  m->lpe = LPE;
  const int nb = s.nbody, nv = s.nv, nu = s.nu;
  m->tp_ok = 0;
  memset(&m->tp, 0, sizeof m->tp);
  auto no = [&](const char* why) { m->tp_why = why; };
  if (nb > LPE - 1)
    return no(variant ? "more than 63 bodies (the last lane of an environment must stay empty)"
                      : "more than 31 bodies (lane 31 of an environment must stay empty)");
  std::vector<int> depth(nb, 0), dd(nv, 1), jnt_of(nb, -1), act_of(nv, -1);
  for (int b = 1; b < nb; b++) {
    depth[b] = depth[s.body_parentid[b]] + 1;
    if (s.body_jntnum[b] > 1) return no("a body has more than one joint");
    if (s.body_jntnum[b] == 1) jnt_of[b] = s.body_jntadr[b];
  }
  int maxdepth = 0, maxdd = 1;
  for (int b = 0; b < nb; b++) maxdepth = depth[b] > maxdepth ? depth[b] : maxdepth;
  for (int i = 0; i < nv; i++) {
    dd[i] = s.dof_parentid[i] >= 0 ? dd[s.dof_parentid[i]] + 1 : 1;
    maxdd = dd[i] > maxdd ? dd[i] : maxdd;
  }
  if (maxdd > TP_MAXD) return no("dof tree deeper than TP_MAXD");
  if (maxdepth > LPE) return no("body tree too deep");
  for (int u = 0; u < nu; u++) {
    if (act_of[s.actuator_trnid[u]] >= 0) return no("two actuators on one joint");
    act_of[s.actuator_trnid[u]] = u;
  }
  std::vector<int32_t>& T = m->tp_table;
  T.assign((size_t)TPI_TOTAL * LPE, -1);
  auto at = [&](int f, int l) -> int32_t& { return T[(size_t)f * LPE + l]; };
  for (int l = 0; l < LPE; l++) {
    at(TPI_PARENT, l) = l; at(TPI_DEPTH, l) = -1; at(TPI_DD, l) = 0; at(TPI_ROOT, l) = -1;
  }
  std::vector<int> nbc(nb, 0), ndc(nb, 0);
  int nroots = 0;
  for (int b = 0; b < nb; b++) {
    at(TPI_PARENT, b) = b == 0 ? 0 : s.body_parentid[b];
    at(TPI_DEPTH, b) = depth[b];
    at(TPI_ROOT, b) = s.body_rootid[b];
    if (b > 0 && s.body_parentid[b] == 0) {
      if (nroots >= 4) return no("more than 4 kinematic trees");
      m->tp.roots[nroots++] = b;
    }
    const int p = s.body_parentid[b];
    if (b > 0 && p > 0) {
      if (nbc[p] >= TP_MAXC) return no("a body has more than 8 children");
      at(TPI_BCHILD + nbc[p]++, p) = b;
      if (nbc[p] > m->tp.lvl_bchild[depth[b]]) m->tp.lvl_bchild[depth[b]] = nbc[p];
    }
    const int j = jnt_of[b];
    if (j >= 0) {
      at(TPI_JNT, b) = j;
      at(TPI_DD, b) = dd[j];
      at(TPI_ACT, b) = act_of[j];
      int a = 1;
      for (int k = s.dof_parentid[j]; k >= 0; k = s.dof_parentid[k], a++) at(TPI_ANC + a, b) = s.dof_bodyid[k];
      const int pj = s.dof_parentid[j];
      if (pj >= 0) {
        const int pb = s.dof_bodyid[pj];
        if (ndc[pb] >= TP_MAXC) return no("a dof has more than 8 child dofs");
        at(TPI_DCHILD + ndc[pb]++, pb) = b;
        if (ndc[pb] > m->tp.lvl_dchild[dd[j]]) m->tp.lvl_dchild[dd[j]] = ndc[pb];
      }
    }
  }
  // dofs below the first branching as single-child chains on consecutive lanes (k_tp_chain)
  m->tp.chain = maxdd >= 3;
  for (int j = 0; j < nv; j++)
    if (dd[j] >= 3 && s.dof_bodyid[s.dof_parentid[j]] != s.dof_bodyid[j] - 1) m->tp.chain = 0;
  if (getenv("BS_NO_CHAIN") || variant) m->tp.chain = 0;  // the wide build has no CHAIN instantiations
  if (m->tp.lvl_dchild[2] > TP_CHAIN_MAXC) m->tp.chain = 0;  // more child chains than the CHAIN kernels unroll
  // levels whose bodies all sit on the lane after their parent's
  for (int L = 0; L < TP_MAXLVL; L++) m->tp.lvl_next[L] = 0;
  for (int L = 2; L <= maxdepth && L < TP_MAXLVL; L++) {
    bool all = true, any = false;
    for (int b = 1; b < nb; b++)
      if (depth[b] == L) { any = true; if (s.body_parentid[b] != b - 1) all = false; }
    m->tp.lvl_next[L] = any && all;
  }
  // levels whose bodies (two or more) all hang off one body
  for (int L = 0; L < TP_MAXLVL; L++) m->tp.lvl_parent[L] = -1;
  for (int L = 2; L <= maxdepth && L < TP_MAXLVL; L++) {
    int par = -1, cnt = 0;
    bool one = true;
    for (int b = 1; b < nb; b++)
      if (depth[b] == L) {
        cnt++;
        if (par < 0) par = s.body_parentid[b];
        else if (par != s.body_parentid[b]) one = false;
      }
    if (one && cnt >= 2 && par > 0) m->tp.lvl_parent[L] = par;
  }
  // equalities: owner = dof1's lane, other = dof2 (must be an ancestor dof of dof1)
  for (size_t r = 0; r < mdl->eq_dof1.size(); r++) {
    const int d1 = mdl->eq_dof1[r], d2 = mdl->eq_dof2[r];
    if (d2 < 0) return no("single-joint equality");
    const int b1 = s.dof_bodyid[d1], b2 = s.dof_bodyid[d2];
    int pos = -1, a = 1;
    for (int k = s.dof_parentid[d1]; k >= 0; k = s.dof_parentid[k], a++)
      if (k == d2) pos = a;
    if (pos < 0) return no("equality joint2 is not an ancestor of joint1");
    if (at(TPI_EQ_OWN, b1) >= 0 || at(TPI_EQ_SRC, b2) >= 0) return no("a dof takes part in two equalities of the same role");
    at(TPI_EQ_OWN, b1) = (int)r; at(TPI_EQ_OTHER, b1) = b2; at(TPI_EQ_POS, b1) = pos; at(TPI_EQ_SRC, b2) = b1;
  }
  const int neq = (int)mdl->eq_dof1.size(), nfric = (int)mdl->fric_dof.size();
  for (int k = 0; k < nfric; k++) at(TPI_FRIC_ROW, s.dof_bodyid[mdl->fric_dof[k]]) = neq + k;
  for (size_t k = 0; k < mdl->lim_jnt.size(); k++) at(TPI_LIM_ROW, s.jnt_bodyid[mdl->lim_jnt[k]]) = neq + nfric + 2 * (int)k;
  for (int l = 0; l < LPE; l++) { at(TPI_ANCMASK, l) = 0; at(TPI_ANCMASK_HI, l) = 0; }
  for (int b = 1; b < nb; b++) {
    uint64_t mask = 0;
    for (int a = b; a > 0; a = s.body_parentid[a])
      if (jnt_of[a] >= 0) mask |= (uint64_t)1 << a;
    at(TPI_ANCMASK, b) = (int32_t)(uint32_t)mask;
    at(TPI_ANCMASK_HI, b) = (int32_t)(uint32_t)(mask >> 32);
  }
  // derived rows (TPI_SA_*, TPI_CW): environment-relative permute byte offsets with the fallbacks resolved, child words
  for (int l = 0; l < LPE; l++) {
    auto rel4 = [&](int v, int fallback) { return 4 * (v >= 0 ? v : fallback); };
    for (int a = 0; a < TP_MAXD; a++) at(TPI_SA_ANC + a, l) = rel4(at(TPI_ANC + a, l), l);
    at(TPI_SA_EQO, l) = rel4(at(TPI_EQ_OTHER, l), l);
    at(TPI_SA_EQS, l) = rel4(at(TPI_EQ_SRC, l), l);
    at(TPI_SA_DC0, l) = rel4(at(TPI_DCHILD, l), LPE - 1);
    at(TPI_SA_BC0, l) = rel4(at(TPI_BCHILD, l), LPE - 1);
    for (int r = 0; r < 2 * TPL_CROWS; r++) {
      const int first = (r < TPL_CROWS ? TPI_BCHILD : TPI_DCHILD) + 1 + 4 * (r % TPL_CROWS);
      int w = 0;
      for (int k = 0; k < 4; k++) {
        const int ch = 1 + 4 * (r % TPL_CROWS) + k;
        const int v = ch < TP_MAXC ? at(first + k, l) : -1;
        w |= (v >= 0 ? v : LPE - 1) << (8 * k);
      }
      at(TPI_CW + r, l) = w;
    }
  }
  // state-independent parts of the constraint rows, with the kernels' own functions
  std::vector<double>& Dt = m->tp_dtable;
  Dt.assign((size_t)TPD_TOTAL * LPE, 0.0);
  auto dat = [&](int f, int l) -> double& { return Dt[(size_t)f * LPE + l]; };
  DevModel hm;
  memset(&hm, 0, sizeof hm);
  hm.disableflags = s.disableflags; hm.timestep = s.timestep;
  for (int l = 0; l < LPE; l++) dat(TPD_LIM_K, l) = -1.0;
  for (int b = 1; b < nb; b++) {
    const int j = jnt_of[b];
    if (j < 0) continue;
    const double iw = s.dof_invweight0[j];
    dat(TPD_IW, b) = iw;
    double K, B;
    if (!(s.disableflags & BS_DSBL_FRICTIONLOSS) && s.dof_frictionloss[j] > 0) {
      const double imp = tp_impedance(s.dof_solimp + 5 * j, 0.0, 0.0);
      kb_from_solref(hm, s.dof_solref + 2 * j, s.dof_solimp + 5 * j, K, B);
      const double R = fmax(BS_MINVAL, (1 - imp) * iw / imp);
      dat(TPD_FRIC_D, b) = 1.0 / R; dat(TPD_FRIC_FL, b) = s.dof_frictionloss[j];
      dat(TPD_FRIC_RF, b) = R * s.dof_frictionloss[j]; dat(TPD_FRIC_B, b) = B;
    }
    if (!(s.disableflags & BS_DSBL_LIMIT) && s.jnt_limited[j]) {
      kb_from_solref(hm, s.jnt_solref + 2 * j, s.jnt_solimp + 5 * j, K, B);
      dat(TPD_LIM_K, b) = K; dat(TPD_LIM_B, b) = B;
    }
    dat(TPD_STIFF, b) = s.jnt_stiffness[j]; dat(TPD_SPRING, b) = s.qpos_spring[j];
    const int u = act_of[j];
    if (u >= 0) {
      const bool cl = s.actuator_ctrllimited[u] && !(s.disableflags & BS_DSBL_CLAMPCTRL), fl = s.actuator_forcelimited[u];
      dat(TPD_ACT_GEAR, b) = s.actuator_gear[u]; dat(TPD_ACT_GAIN, b) = s.actuator_gainprm[3 * u];
      dat(TPD_ACT_B0, b) = s.actuator_biasprm[3 * u]; dat(TPD_ACT_B1, b) = s.actuator_biasprm[3 * u + 1];
      dat(TPD_ACT_B2, b) = s.actuator_biasprm[3 * u + 2];
      dat(TPD_ACT_CLO, b) = cl ? s.actuator_ctrlrange[2 * u] : -1e300; dat(TPD_ACT_CHI, b) = cl ? s.actuator_ctrlrange[2 * u + 1] : 1e300;
      dat(TPD_ACT_FLO, b) = fl ? s.actuator_forcerange[2 * u] : -1e300; dat(TPD_ACT_FHI, b) = fl ? s.actuator_forcerange[2 * u + 1] : 1e300;
    }
    const int r = at(TPI_EQ_OWN, b);
    if (r >= 0) {
      kb_from_solref(hm, mdl->eq_solref.data() + 2 * r, mdl->eq_solimp.data() + 5 * r, K, B);
      dat(TPD_EQ_K, b) = K; dat(TPD_EQ_B, b) = B;
      dat(TPD_EQ_DIAG, b) = iw + s.dof_invweight0[mdl->eq_dof2[r]];
    }
  }
  // the lane-major record of everything a lane reads per step (TPK_*)
  std::vector<double>& Kt = m->tp_ktable;
  Kt.assign((size_t)LPE * TPK_N, 0.0);
  for (int l = 0; l < LPE; l++) {
    double* r = &Kt[(size_t)l * TPK_N];
    const int b = l < nb ? l : 0;  // lanes beyond the bodies read body 0, lanes without a joint read joint 0 (as the kernels index)
    const int j = l < nb ? jnt_of[b] : -1;
    for (int q = 0; q < 3; q++) { r[TPK_BPOS + q] = s.body_pos[3 * b + q]; r[TPK_IPOS + q] = s.body_ipos[3 * b + q]; r[TPK_INERTIA + q] = s.body_inertia[3 * b + q]; }
    for (int q = 0; q < 4; q++) { r[TPK_BQUAT + q] = s.body_quat[4 * b + q]; r[TPK_IQUAT + q] = s.body_iquat[4 * b + q]; }
    for (int f = 0; f < TPD_TOTAL; f++) r[TPK_D0 + f] = dat(f, l);
    // the closed forms below and the kernels' single normalisation per chain pose take unit quaternions (mjModel's are;
    // arrays handed straight to bs_model_create may be a few digits short): rescaled here when they are not
    auto unit = [](double* v, int n) {
      double n2 = 0;
      for (int q = 0; q < n; q++) n2 += v[q] * v[q];
      if (n2 > 0 && std::fabs(n2 - 1.0) > 1e-15) { const double sc = 1.0 / std::sqrt(n2); for (int q = 0; q < n; q++) v[q] *= sc; }
    };
    unit(r + TPK_BQUAT, 4); unit(r + TPK_IQUAT, 4);
    {  // Ib = R_i diag(I) R_i^T
      const M3 Ri = q2m(Q4{r[TPK_IQUAT], r[TPK_IQUAT + 1], r[TPK_IQUAT + 2], r[TPK_IQUAT + 3]});
      const int ij[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
      for (int q = 0; q < 6; q++) {
        double v = 0;
        for (int k = 0; k < 3; k++) v += Ri.a[3 * ij[q][0] + k] * r[TPK_INERTIA + k] * Ri.a[3 * ij[q][1] + k];
        r[TPK_IB + q] = v;
      }
    }
    const int jj = j >= 0 ? j : 0;
    if (s.njnt > 0) {
      for (int q = 0; q < 3; q++) { r[TPK_JPOS + q] = s.jnt_pos[3 * jj + q]; r[TPK_JAX + q] = s.jnt_axis[3 * jj + q]; }
      unit(r + TPK_JAX, 3);
      r[TPK_ARMATURE] = s.dof_armature[jj];
      {  // closed-form joint-local pose constants (physics_tp.h, TPK_P1)
        const Q4 bq = {r[TPK_BQUAT], r[TPK_BQUAT + 1], r[TPK_BQUAT + 2], r[TPK_BQUAT + 3]};
        const M3 Rb = q2m(bq);
        const V3 k = {r[TPK_JAX], r[TPK_JAX + 1], r[TPK_JAX + 2]}, jp = {r[TPK_JPOS], r[TPK_JPOS + 1], r[TPK_JPOS + 2]};
        V3 p1, p2;
        if (s.jnt_type[jj] == BS_JNT_SLIDE) { p1 = {0, 0, 0}; p2 = mrot(Rb, k); }
        else {
          const double kj = k.x * jp.x + k.y * jp.y + k.z * jp.z;
          p1 = mrot(Rb, V3{jp.x - k.x * kj, jp.y - k.y * kj, jp.z - k.z * kj});
          p2 = mrot(Rb, cross(k, jp));
        }
        const Q4 q1 = qmul(bq, Q4{0.0, k.x, k.y, k.z});
        r[TPK_P1] = p1.x; r[TPK_P1 + 1] = p1.y; r[TPK_P1 + 2] = p1.z;
        r[TPK_P2] = p2.x; r[TPK_P2 + 1] = p2.y; r[TPK_P2 + 2] = p2.z;
        r[TPK_Q1] = q1.w; r[TPK_Q1 + 1] = q1.x; r[TPK_Q1 + 2] = q1.y; r[TPK_Q1 + 3] = q1.z;
      }
      r[TPK_JMARGIN] = s.jnt_margin[jj]; r[TPK_JRANGE] = s.jnt_range[2 * jj]; r[TPK_JRANGE + 1] = s.jnt_range[2 * jj + 1];
      for (int q = 0; q < 5; q++) r[TPK_JSOLIMP + q] = s.jnt_solimp[5 * jj + q];
    }
    const int e = at(TPI_EQ_OWN, l);
    if (e >= 0) {
      for (int q = 0; q < 5; q++) { r[TPK_EQDATA + q] = mdl->eq_data[5 * e + q]; r[TPK_EQSOLIMP + q] = mdl->eq_solimp[5 * e + q]; }
      r[TPK_EQQPOS0] = s.qpos0[mdl->eq_jnt2[e]];
    }
  }
  // PGS sweep order = efc order: equalities (owner lanes listed here), then frictionloss rows and limits in
  // dof order, which is the lane order (bodies and dofs are both numbered depth-first)
  m->tp_pgs.clear();
  for (size_t r = 0; r < mdl->eq_dof1.size(); r++) m->tp_pgs.push_back(s.dof_bodyid[mdl->eq_dof1[r]]);
  m->tp.npgs = (int)m->tp_pgs.size();
  m->tp.maxdepth = maxdepth; m->tp.maxdd = maxdd; m->tp.nroots = nroots;
  int mind = maxdepth;
  for (int b = 1; b < nb; b++)
    if (jnt_of[b] >= 0 && depth[b] < mind) mind = depth[b];
  m->tp.fold_minlvl = mind + 1 < 2 ? 2 : mind + 1;
  {
    // hands: every body after the first dof body ("palm") in aligned groups of four lanes, each a chain off the palm
    int top = -1;
    for (int b = 1; b < nb; b++)
      if (jnt_of[b] >= 0) { top = b; break; }
    bool ok = m->tp.chain && top > 0 && (top + 1) % 4 == 0 && nb <= 32 && LPE == 32 && !getenv("BS_NO_QUAD");
    for (int b = top + 1; ok && b < nb; b++) {
      const int q = (b - top - 1) % 4;
      if (s.body_parentid[b] != (q == 0 ? top : b - 1)) ok = false;
    }
    m->tp.quad_top = ok ? top : -1;
  }
  m->tp.first_dof = 0;
  for (int b = 1; b < nb; b++)
    if (jnt_of[b] >= 0) { m->tp.first_dof = b; break; }
  if (getenv("BS_NO_FIRST_DOF")) m->tp.first_dof = 0;
  {
    // the fold at fold_minlvl goes into one body (lvl_parent) that carries the shallowest dof, and that dof has no
    // ancestor dof: the kernels need only two scalars of it (physics_tp.h, "palm")
    const int L = m->tp.fold_minlvl, top = L < TP_MAXLVL ? m->tp.lvl_parent[L] : -1;
    m->tp.palm_scalar = top > 0 && jnt_of[top] >= 0 && depth[top] == mind && L == mind + 1 && at(TPI_DD, top) == 1 ? 1 : 0;
    if (getenv("BS_NO_PALM_SCALAR")) m->tp.palm_scalar = 0;
  }
  m->tp.jump_rounds = 0;
  while ((1 << m->tp.jump_rounds) < maxdepth) m->tp.jump_rounds++;
  m->tp_ok = 1;
}

namespace {
inline int64_t pack_rec(int id, int deg, int k0) {
  return (int64_t)(((uint64_t)(uint32_t)k0 << 32) | ((uint64_t)(deg & 0xFFF) << 20) | (uint64_t)(id & 0xFFFFF));
}
struct Packer {
  std::vector<char> bytes;
  size_t add(const void* p, size_t n) {
    size_t off = (bytes.size() + 31) & ~size_t(31);  // 32 bytes: double4 records
    bytes.resize(off + (n ? n : 16));
    if (n) memcpy(bytes.data() + off, p, n);
    return off;
  }
};
}  // namespace

#define CHECK_PTR(field, count)                                                        \
  if ((count) > 0 && !desc->field) return fail(BS_ERR_INVALID, "bs_model_create: %s is null", #field)

extern "C" int bs_model_create(const bs_model_desc* desc, bs_model** out) {
  if (!desc || !out) return fail(BS_ERR_INVALID, "bs_model_create: null argument");
  const int nb = desc->nbody, nj = desc->njnt, nv = desc->nv, nu = desc->nu, ng = desc->ngeom,
            ne = desc->neq;
  if (nb < 1 || nj < 0 || nv != nj || nu < 0 || ng < 0 || ne < 0 || desc->nM < nv)
    return fail(BS_ERR_INVALID, "bs_model_create: inconsistent sizes (nbody=%d njnt=%d nv=%d nM=%d)", nb,
                nj, nv, desc->nM);
  if (!(desc->timestep > 0)) return fail(BS_ERR_INVALID, "bs_model_create: timestep must be > 0");
  if (desc->solver < 0 || desc->solver > 2)
    return fail(BS_ERR_UNSUPPORTED, "bs_model_create: solver %d is not implemented (0 = PGS, 1 = CG, 2 = Newton)",
                desc->solver);
  if (desc->integrator != 0 && desc->integrator != 1)
    return fail(BS_ERR_INVALID, "bs_model_create: integrator must be 0 (Euler) or 1 (RK4)");
  CHECK_PTR(body_parentid, nb); CHECK_PTR(body_rootid, nb); CHECK_PTR(body_jntadr, nb);
  CHECK_PTR(body_jntnum, nb); CHECK_PTR(body_dofadr, nb); CHECK_PTR(body_dofnum, nb);
  CHECK_PTR(body_pos, nb); CHECK_PTR(body_quat, nb); CHECK_PTR(body_ipos, nb); CHECK_PTR(body_iquat, nb);
  CHECK_PTR(body_mass, nb); CHECK_PTR(body_subtreemass, nb); CHECK_PTR(body_inertia, nb);
  CHECK_PTR(body_invweight0, nb);
  CHECK_PTR(jnt_type, nj); CHECK_PTR(jnt_bodyid, nj); CHECK_PTR(jnt_qposadr, nj); CHECK_PTR(jnt_dofadr, nj);
  CHECK_PTR(jnt_limited, nj); CHECK_PTR(jnt_pos, nj); CHECK_PTR(jnt_axis, nj); CHECK_PTR(jnt_range, nj);
  CHECK_PTR(jnt_margin, nj); CHECK_PTR(jnt_stiffness, nj); CHECK_PTR(jnt_solref, nj);
  CHECK_PTR(jnt_solimp, nj); CHECK_PTR(qpos0, nj); CHECK_PTR(qpos_spring, nj);
  CHECK_PTR(dof_bodyid, nv); CHECK_PTR(dof_jntid, nv); CHECK_PTR(dof_parentid, nv); CHECK_PTR(dof_Madr, nv);
  CHECK_PTR(dof_armature, nv); CHECK_PTR(dof_damping, nv); CHECK_PTR(dof_frictionloss, nv);
  CHECK_PTR(dof_solref, nv); CHECK_PTR(dof_solimp, nv); CHECK_PTR(dof_invweight0, nv);
  CHECK_PTR(geom_bodyid, ng); CHECK_PTR(geom_pos, ng); CHECK_PTR(geom_quat, ng);
  CHECK_PTR(geom_type, ng); CHECK_PTR(geom_dataid, ng); CHECK_PTR(geom_condim, ng); CHECK_PTR(geom_rbound, ng);
  CHECK_PTR(geom_margin, ng); CHECK_PTR(geom_gap, ng); CHECK_PTR(geom_friction, ng); CHECK_PTR(geom_solref, ng);
  CHECK_PTR(geom_solimp, ng); CHECK_PTR(geom_solmix, ng);
  CHECK_PTR(pair_geom1, desc->npair); CHECK_PTR(pair_geom2, desc->npair);
  CHECK_PTR(mesh_vertadr, desc->nmesh); CHECK_PTR(mesh_vertnum, desc->nmesh);
  CHECK_PTR(hull_vert, desc->nhullvert); CHECK_PTR(hull_adj_start, desc->nhullvert); CHECK_PTR(hull_adj, desc->nhulladj);
  if (desc->npair < 0 || desc->nmesh < 0 || desc->nhullvert < 0 || desc->nhulladj < 0)
    return fail(BS_ERR_INVALID, "bs_model_create: negative collision sizes");
  for (int p = 0; p < desc->npair; p++) {
    const int g1 = desc->pair_geom1[p], g2 = desc->pair_geom2[p];
    if (g1 < 0 || g1 >= ng || g2 < 0 || g2 >= ng) return fail(BS_ERR_INVALID, "pair %d: bad geom id", p);
  }
  for (int g = 0; g < ng; g++) {
    if (desc->geom_type[g] == BS_GEOM_MESH) {
      const int me = desc->geom_dataid[g];
      if (me < 0 || me >= desc->nmesh) return fail(BS_ERR_INVALID, "geom %d: bad mesh id", g);
      if (desc->mesh_vertnum[me] < 1 || desc->mesh_vertadr[me] < 0 ||
          desc->mesh_vertadr[me] + desc->mesh_vertnum[me] > desc->nhullvert)
        return fail(BS_ERR_INVALID, "mesh %d: bad hull vertex range", me);
    }
    if (desc->npair > 0 && !(desc->disableflags & BS_DSBL_CONTACT) && desc->geom_condim[g] != 1 && desc->geom_condim[g] != 3)
      return fail(BS_ERR_UNSUPPORTED, "geom %d: condim %d (only 1 and 3 are implemented)", g, desc->geom_condim[g]);
  }
  if (desc->npair > 0 && desc->ngeom > 0xFFF)
    return fail(BS_ERR_UNSUPPORTED, "bs_model_create: %d geoms (the per-pair records hold 12-bit geom ids)", desc->ngeom);
  for (int me = 0; me < desc->nmesh; me++)
    for (int v = desc->mesh_vertadr[me]; v < desc->mesh_vertadr[me] + desc->mesh_vertnum[me]; v++) {
      if (desc->hull_adj_start[v] < 0 || desc->hull_adj_start[v + 1] < desc->hull_adj_start[v] ||
          desc->hull_adj_start[v + 1] > desc->nhulladj)
        return fail(BS_ERR_INVALID, "hull vertex %d: bad adjacency range", v);
      if (desc->hull_adj_start[v + 1] - desc->hull_adj_start[v] > 0xFFF || desc->mesh_vertnum[me] > 0xFFFFF)
        return fail(BS_ERR_UNSUPPORTED, "bs_model_create: hull %d exceeds 2^20 vertices or a vertex has more than 4095 neighbours", me);
      for (int q = desc->hull_adj_start[v]; q < desc->hull_adj_start[v + 1]; q++)
        if (desc->hull_adj[q] < 0 || desc->hull_adj[q] >= desc->mesh_vertnum[me])
          return fail(BS_ERR_INVALID, "hull vertex %d: neighbour out of range", v);
    }
  CHECK_PTR(actuator_trnid, nu); CHECK_PTR(actuator_ctrllimited, nu); CHECK_PTR(actuator_forcelimited, nu);
  CHECK_PTR(actuator_ctrlrange, nu); CHECK_PTR(actuator_forcerange, nu); CHECK_PTR(actuator_gear, nu);
  CHECK_PTR(actuator_gainprm, nu); CHECK_PTR(actuator_biasprm, nu);
  CHECK_PTR(eq_obj1id, ne); CHECK_PTR(eq_obj2id, ne); CHECK_PTR(eq_active, ne); CHECK_PTR(eq_data, ne);
  CHECK_PTR(eq_solref, ne); CHECK_PTR(eq_solimp, ne);

  // structural validation: the kernels index with these, so reject anything out of range
  for (int b = 0; b < nb; b++) {
    if (b > 0 && (desc->body_parentid[b] < 0 || desc->body_parentid[b] >= b))
      return fail(BS_ERR_INVALID, "body %d: parent id %d must precede it", b, desc->body_parentid[b]);
    if (desc->body_rootid[b] < 0 || desc->body_rootid[b] > b)
      return fail(BS_ERR_INVALID, "body %d: bad root id", b);
    if (desc->body_jntnum[b] < 0 || (desc->body_jntnum[b] > 0 &&
        (desc->body_jntadr[b] < 0 || desc->body_jntadr[b] + desc->body_jntnum[b] > nj)))
      return fail(BS_ERR_INVALID, "body %d: bad joint range", b);
    if (desc->body_dofnum[b] != desc->body_jntnum[b] ||
        (desc->body_dofnum[b] > 0 && desc->body_dofadr[b] != desc->body_jntadr[b]))
      return fail(BS_ERR_INVALID, "body %d: dof range must equal joint range (hinge/slide only)", b);
  }
  int nM = 0;
  for (int i = 0; i < nv; i++) {
    if (desc->jnt_type[i] != BS_JNT_HINGE && desc->jnt_type[i] != BS_JNT_SLIDE)
      return fail(BS_ERR_UNSUPPORTED, "joint %d: only hinge and slide joints are supported", i);
    if (desc->jnt_qposadr[i] != i || desc->jnt_dofadr[i] != i || desc->dof_jntid[i] != i)
      return fail(BS_ERR_INVALID, "joint %d: qpos/dof address must equal the joint id", i);
    if (desc->dof_parentid[i] >= i || desc->dof_parentid[i] < -1)
      return fail(BS_ERR_INVALID, "dof %d: bad parent", i);
    if (desc->jnt_bodyid[i] < 1 || desc->jnt_bodyid[i] >= nb || desc->dof_bodyid[i] != desc->jnt_bodyid[i])
      return fail(BS_ERR_INVALID, "joint %d: bad body id", i);
    if (desc->dof_Madr[i] != nM) return fail(BS_ERR_INVALID, "dof %d: bad dof_Madr", i);
    for (int k = i; k >= 0; k = desc->dof_parentid[k]) nM++;
  }
  if (nM != desc->nM) return fail(BS_ERR_INVALID, "nM=%d does not match the dof tree (%d)", desc->nM, nM);
  for (int u = 0; u < nu; u++)
    if (desc->actuator_trnid[u] < 0 || desc->actuator_trnid[u] >= nj)
      return fail(BS_ERR_INVALID, "actuator %d: bad joint id", u);
  for (int g = 0; g < ng; g++)
    if (desc->geom_bodyid[g] < 0 || desc->geom_bodyid[g] >= nb)
      return fail(BS_ERR_INVALID, "geom %d: bad body id", g);

  bs_model* m = new bs_model();
  m->d = *desc;
  // deep copy of the arrays the device image needs (sizes in elements)
  struct IA { const int32_t** p; size_t n; };
  struct DA { const double** p; size_t n; };
  IA ia[] = {{&m->d.body_parentid, (size_t)nb}, {&m->d.body_rootid, (size_t)nb}, {&m->d.body_jntadr, (size_t)nb},
             {&m->d.body_jntnum, (size_t)nb}, {&m->d.body_dofadr, (size_t)nb}, {&m->d.body_dofnum, (size_t)nb},
             {&m->d.jnt_type, (size_t)nj}, {&m->d.jnt_bodyid, (size_t)nj}, {&m->d.jnt_qposadr, (size_t)nj},
             {&m->d.jnt_dofadr, (size_t)nj}, {&m->d.jnt_limited, (size_t)nj}, {&m->d.dof_bodyid, (size_t)nv},
             {&m->d.dof_jntid, (size_t)nv}, {&m->d.dof_parentid, (size_t)nv}, {&m->d.dof_Madr, (size_t)nv},
             {&m->d.geom_bodyid, (size_t)ng}, {&m->d.geom_type, (size_t)ng}, {&m->d.geom_dataid, (size_t)ng},
             {&m->d.geom_condim, (size_t)ng}, {&m->d.pair_geom1, (size_t)desc->npair},
             {&m->d.pair_geom2, (size_t)desc->npair}, {&m->d.mesh_vertadr, (size_t)desc->nmesh},
             {&m->d.mesh_vertnum, (size_t)desc->nmesh},
             {&m->d.hull_adj_start, (size_t)(desc->nhullvert > 0 ? desc->nhullvert + 1 : 0)},
             {&m->d.hull_adj, (size_t)desc->nhulladj}, {&m->d.actuator_trnid, (size_t)nu},
             {&m->d.actuator_ctrllimited, (size_t)nu}, {&m->d.actuator_forcelimited, (size_t)nu},
             {&m->d.eq_obj1id, (size_t)ne}, {&m->d.eq_obj2id, (size_t)ne}, {&m->d.eq_active, (size_t)ne}};
  DA da[] = {{&m->d.body_pos, (size_t)nb * 3}, {&m->d.body_quat, (size_t)nb * 4}, {&m->d.body_ipos, (size_t)nb * 3},
             {&m->d.body_iquat, (size_t)nb * 4}, {&m->d.body_mass, (size_t)nb}, {&m->d.body_subtreemass, (size_t)nb},
             {&m->d.body_inertia, (size_t)nb * 3}, {&m->d.body_invweight0, (size_t)nb * 2},
             {&m->d.jnt_pos, (size_t)nj * 3}, {&m->d.jnt_axis, (size_t)nj * 3}, {&m->d.jnt_range, (size_t)nj * 2},
             {&m->d.jnt_margin, (size_t)nj}, {&m->d.jnt_stiffness, (size_t)nj}, {&m->d.jnt_solref, (size_t)nj * 2},
             {&m->d.jnt_solimp, (size_t)nj * 5}, {&m->d.qpos0, (size_t)nj}, {&m->d.qpos_spring, (size_t)nj},
             {&m->d.dof_armature, (size_t)nv}, {&m->d.dof_damping, (size_t)nv}, {&m->d.dof_frictionloss, (size_t)nv},
             {&m->d.dof_solref, (size_t)nv * 2}, {&m->d.dof_solimp, (size_t)nv * 5}, {&m->d.dof_invweight0, (size_t)nv},
             {&m->d.geom_pos, (size_t)ng * 3}, {&m->d.geom_quat, (size_t)ng * 4},
             {&m->d.geom_rbound, (size_t)ng}, {&m->d.geom_margin, (size_t)ng}, {&m->d.geom_gap, (size_t)ng},
             {&m->d.geom_friction, (size_t)ng * 3}, {&m->d.geom_solref, (size_t)ng * 2},
             {&m->d.geom_solimp, (size_t)ng * 5}, {&m->d.geom_solmix, (size_t)ng},
             {&m->d.hull_vert, (size_t)desc->nhullvert * 3}, {&m->d.geom_size, (size_t)(desc->geom_size ? ng * 3 : 0)},
             {&m->d.actuator_ctrlrange, (size_t)nu * 2}, {&m->d.actuator_forcerange, (size_t)nu * 2},
             {&m->d.actuator_gear, (size_t)nu}, {&m->d.actuator_gainprm, (size_t)nu * 3},
             {&m->d.actuator_biasprm, (size_t)nu * 3}, {&m->d.eq_data, (size_t)ne * 5},
             {&m->d.eq_solref, (size_t)ne * 2}, {&m->d.eq_solimp, (size_t)ne * 5}};
  size_t ni = 0, nd = 0;
  for (auto& a : ia) ni += a.n;
  for (auto& a : da) nd += a.n;
  m->ibuf.resize(ni + 1);
  m->dbuf.resize(nd + 1);
  size_t oi = 0, od = 0;
  for (auto& a : ia) {
    if (a.n) memcpy(m->ibuf.data() + oi, *a.p, a.n * sizeof(int32_t));
    *a.p = m->ibuf.data() + oi;
    oi += a.n;
  }
  for (auto& a : da) {
    if (a.n) memcpy(m->dbuf.data() + od, *a.p, a.n * sizeof(double));
    *a.p = m->dbuf.data() + od;
    od += a.n;
  }

  // fixed constraint-row layout
  const int flags = desc->disableflags;
  if (!(flags & BS_DSBL_EQUALITY))
    for (int e = 0; e < ne; e++) {
      if (!m->d.eq_active[e]) continue;
      const int j1 = m->d.eq_obj1id[e], j2 = m->d.eq_obj2id[e];
      if (j1 < 0 || j1 >= nj || j2 >= nj) {
        delete m;
        return fail(BS_ERR_INVALID, "equality %d: bad joint id", e);
      }
      m->eq_jnt1.push_back(j1); m->eq_jnt2.push_back(j2);
      const int d1 = j1, d2 = j2 >= 0 ? j2 : -1;
      m->eq_dof1.push_back(d1); m->eq_dof2.push_back(d2);
      int hadr = -1;
      if (d2 >= 0 && d1 != d2) {
        const int hi = d1 > d2 ? d1 : d2, lo = d1 > d2 ? d2 : d1;
        int adr = m->d.dof_Madr[hi];
        for (int k = hi; k >= 0; k = m->d.dof_parentid[k], adr++)
          if (k == lo) { hadr = adr; break; }
        if (hadr < 0) m->hess_sparse = 0;
      } else if (d2 >= 0) {
        delete m;
        return fail(BS_ERR_INVALID, "equality %d couples a joint with itself", e);
      }
      m->eq_Hadr.push_back(hadr);
      for (int k = 0; k < 5; k++) m->eq_data.push_back(m->d.eq_data[5 * e + k]);
      for (int k = 0; k < 2; k++) m->eq_solref.push_back(m->d.eq_solref[2 * e + k]);
      for (int k = 0; k < 5; k++) m->eq_solimp.push_back(m->d.eq_solimp[5 * e + k]);
    }
  if (!m->hess_sparse) {
    delete m;
    return fail(BS_ERR_UNSUPPORTED,
                "a joint equality couples dofs on different branches of the tree; the dense-Hessian "
                "solver path is not implemented yet");
  }
  if (!(flags & BS_DSBL_FRICTIONLOSS))
    for (int i = 0; i < nv; i++)
      if (m->d.dof_frictionloss[i] > 0) m->fric_dof.push_back(i);
  if (!(flags & BS_DSBL_LIMIT))
    for (int j = 0; j < nj; j++)
      if (m->d.jnt_limited[j]) m->lim_jnt.push_back(j);
  for (int i = 0; i < nv; i++) m->damped |= m->d.dof_damping[i] > 0;
  build_tp(m, 0);
  build_tp(m, 1);
  *out = m;
  return BS_OK;
}

static void free_images(bs_model* m) {
  for (auto& im : m->images)
    if (im.blob) {
      (void)hipSetDevice(im.device);
      (void)hipFree(im.blob);
    }
  m->images.clear();
}

extern "C" int bs_model_destroy(bs_model* m) {
  if (!m) return BS_OK;
  free_images(m);
  delete m;
  return BS_OK;
}

// upload (once per device) and return the device-side view
static int model_image(bs_model* m, int device, const DeviceImage** out) {
  for (auto& im : m->images)
    if (im.device == device) { *out = &im; return BS_OK; }
  Packer pk;
  const bs_model_desc& s = m->d;
  const int nb = s.nbody, nj = s.njnt, nv = s.nv, nu = s.nu, ng = s.ngeom;
  DevModel dm;
  memset(&dm, 0, sizeof dm);
  dm.nbody = nb; dm.njnt = nj; dm.nv = nv; dm.nu = nu; dm.ngeom = ng; dm.nM = s.nM; dm.npair = s.npair;
  dm.neq = (int)m->eq_dof1.size(); dm.nfric = (int)m->fric_dof.size(); dm.nlim = (int)m->lim_jnt.size();
  dm.nrow = dm.neq + dm.nfric + 2 * dm.nlim;
  dm.integrator = s.integrator; dm.solver = s.solver; dm.iterations = s.iterations; dm.ls_iterations = s.ls_iterations;
  dm.disableflags = s.disableflags; dm.damped = m->damped; dm.hess_sparse = m->hess_sparse;
  dm.timestep = s.timestep; memcpy(dm.gravity, s.gravity, sizeof dm.gravity);
  dm.tolerance = s.tolerance; dm.ls_tolerance = s.ls_tolerance; dm.impratio = s.impratio;
  dm.meaninertia = s.meaninertia;
  struct Fix { size_t off; void** slot; };
  std::vector<Fix> fix;
#define PUT_I(dst, src, n) fix.push_back({pk.add(src, (size_t)(n) * sizeof(int32_t)), (void**)&dm.dst})
#define PUT_D(dst, src, n) fix.push_back({pk.add(src, (size_t)(n) * sizeof(double)), (void**)&dm.dst})
  PUT_I(body_parentid, s.body_parentid, nb); PUT_I(body_rootid, s.body_rootid, nb);
  PUT_I(body_jntadr, s.body_jntadr, nb); PUT_I(body_jntnum, s.body_jntnum, nb);
  PUT_I(body_dofadr, s.body_dofadr, nb); PUT_I(body_dofnum, s.body_dofnum, nb);
  PUT_D(body_pos, s.body_pos, nb * 3); PUT_D(body_quat, s.body_quat, nb * 4);
  PUT_D(body_ipos, s.body_ipos, nb * 3); PUT_D(body_iquat, s.body_iquat, nb * 4);
  PUT_D(body_mass, s.body_mass, nb); PUT_D(body_subtreemass, s.body_subtreemass, nb);
  PUT_D(body_inertia, s.body_inertia, nb * 3); PUT_D(body_invweight0, s.body_invweight0, nb * 2);
  PUT_I(jnt_type, s.jnt_type, nj); PUT_I(jnt_bodyid, s.jnt_bodyid, nj); PUT_I(jnt_qposadr, s.jnt_qposadr, nj);
  PUT_I(jnt_dofadr, s.jnt_dofadr, nj); PUT_I(jnt_limited, s.jnt_limited, nj);
  PUT_D(jnt_pos, s.jnt_pos, nj * 3); PUT_D(jnt_axis, s.jnt_axis, nj * 3); PUT_D(jnt_range, s.jnt_range, nj * 2);
  PUT_D(jnt_margin, s.jnt_margin, nj); PUT_D(jnt_stiffness, s.jnt_stiffness, nj);
  PUT_D(jnt_solref, s.jnt_solref, nj * 2); PUT_D(jnt_solimp, s.jnt_solimp, nj * 5);
  PUT_D(qpos0, s.qpos0, nj); PUT_D(qpos_spring, s.qpos_spring, nj);
  PUT_I(dof_bodyid, s.dof_bodyid, nv); PUT_I(dof_jntid, s.dof_jntid, nv);
  PUT_I(dof_parentid, s.dof_parentid, nv); PUT_I(dof_Madr, s.dof_Madr, nv);
  PUT_D(dof_armature, s.dof_armature, nv); PUT_D(dof_damping, s.dof_damping, nv);
  PUT_D(dof_frictionloss, s.dof_frictionloss, nv); PUT_D(dof_solref, s.dof_solref, nv * 2);
  PUT_D(dof_solimp, s.dof_solimp, nv * 5); PUT_D(dof_invweight0, s.dof_invweight0, nv);
  PUT_I(geom_bodyid, s.geom_bodyid, ng); PUT_D(geom_pos, s.geom_pos, ng * 3); PUT_D(geom_quat, s.geom_quat, ng * 4);
  PUT_I(actuator_trnid, s.actuator_trnid, nu); PUT_I(actuator_ctrllimited, s.actuator_ctrllimited, nu);
  PUT_I(actuator_forcelimited, s.actuator_forcelimited, nu);
  PUT_D(actuator_ctrlrange, s.actuator_ctrlrange, nu * 2); PUT_D(actuator_forcerange, s.actuator_forcerange, nu * 2);
  PUT_D(actuator_gear, s.actuator_gear, nu); PUT_D(actuator_gainprm, s.actuator_gainprm, nu * 3);
  PUT_D(actuator_biasprm, s.actuator_biasprm, nu * 3);
  PUT_I(eq_dof1, m->eq_dof1.data(), dm.neq); PUT_I(eq_dof2, m->eq_dof2.data(), dm.neq);
  PUT_I(eq_jnt1, m->eq_jnt1.data(), dm.neq); PUT_I(eq_jnt2, m->eq_jnt2.data(), dm.neq);
  PUT_I(eq_Hadr, m->eq_Hadr.data(), dm.neq);
  PUT_D(eq_data, m->eq_data.data(), dm.neq * 5); PUT_D(eq_solref, m->eq_solref.data(), dm.neq * 2);
  PUT_D(eq_solimp, m->eq_solimp.data(), dm.neq * 5);
  PUT_I(fric_dof, m->fric_dof.data(), dm.nfric); PUT_I(lim_jnt, m->lim_jnt.data(), dm.nlim);
  ColModel cm;
  memset(&cm, 0, sizeof cm);
  cm.npair = s.npair; cm.ngeom = ng; cm.mpr_iterations = s.mpr_iterations; cm.mpr_tolerance = s.mpr_tolerance;
#define PUT_CI(dst, src, n) fix.push_back({pk.add(src, (size_t)(n) * sizeof(int32_t)), (void**)&cm.dst})
#define PUT_CD(dst, src, n) fix.push_back({pk.add(src, (size_t)(n) * sizeof(double)), (void**)&cm.dst})
  PUT_CI(pair_geom1, s.pair_geom1, s.npair); PUT_CI(pair_geom2, s.pair_geom2, s.npair);
  PUT_CI(geom_type, s.geom_type, ng); PUT_CI(geom_dataid, s.geom_dataid, ng); PUT_CI(geom_condim, s.geom_condim, ng);
  PUT_CI(geom_bodyid, s.geom_bodyid, ng);
  PUT_CD(geom_rbound, s.geom_rbound, ng); PUT_CD(geom_margin, s.geom_margin, ng); PUT_CD(geom_gap, s.geom_gap, ng);
  PUT_CD(geom_friction, s.geom_friction, ng * 3); PUT_CD(geom_solref, s.geom_solref, ng * 2);
  PUT_CD(geom_solimp, s.geom_solimp, ng * 5); PUT_CD(geom_solmix, s.geom_solmix, ng);
  // per-pair contact-row constants: the arithmetic of physics_contact.h::make_contact_rows that does not depend
  // on the state (parameter mixing, mj_makeImpedance's K / B, diagApprox, the pyramid's R scale)
  std::vector<double> pcon((size_t)(s.npair > 0 ? s.npair : 1) * BS_PAIR_CON, 0.0);
  std::vector<int32_t> pbody((size_t)(s.npair > 0 ? s.npair : 1) * 2, 0);
  for (int p = 0; p < s.npair; p++) {
    const int g1 = s.pair_geom1[p], g2 = s.pair_geom2[p];
    const int b1 = s.geom_bodyid[g1], b2 = s.geom_bodyid[g2];
    double* r = &pcon[(size_t)p * BS_PAIR_CON];
    const int dim = s.geom_condim[g1] > s.geom_condim[g2] ? s.geom_condim[g1] : s.geom_condim[g2];
    const double mu = fmax(s.geom_friction[3 * g1], s.geom_friction[3 * g2]);
    const double s1 = s.geom_solmix[g1], s2 = s.geom_solmix[g2];
    double mix;
    if (s1 >= BS_MINVAL && s2 >= BS_MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < BS_MINVAL && s2 < BS_MINVAL) mix = 0.5;
    else mix = s1 < BS_MINVAL ? 0.0 : 1.0;
    double solref[2], solimp[5];
    for (int q = 0; q < 2; q++) solref[q] = mix * s.geom_solref[2 * g1 + q] + (1 - mix) * s.geom_solref[2 * g2 + q];
    for (int q = 0; q < 5; q++) solimp[q] = mix * s.geom_solimp[5 * g1 + q] + (1 - mix) * s.geom_solimp[5 * g2 + q];
    double K, B;
    kb_from_solref(dm, solref, solimp, K, B);
    const double tran = s.body_invweight0[2 * b1] + s.body_invweight0[2 * b2];
    const double mue = mu / sqrt(fmax(BS_MINVAL, s.impratio));
    r[0] = fmax(s.geom_margin[g1], s.geom_margin[g2]) - fmax(s.geom_gap[g1], s.geom_gap[g2]);
    r[1] = mu; r[2] = K; r[3] = B;
    for (int q = 0; q < 5; q++) r[4 + q] = solimp[q];
    r[9] = dim == 1 ? tran : tran + mu * mu * tran;
    r[10] = 2 * mue * mue;
    pbody[2 * p] = b1 | (dim << 16); pbody[2 * p + 1] = b2;
  }
  std::vector<int32_t> pnp((size_t)(s.npair > 0 ? s.npair : 1) * 8, 0);
  for (int p = 0; p < s.npair; p++) {
    const int g1 = s.pair_geom1[p], g2 = s.pair_geom2[p];
    int32_t* r = &pnp[(size_t)p * 8];
    const bool mesh1 = s.geom_type[g1] == BS_GEOM_MESH, mesh2 = s.geom_type[g2] == BS_GEOM_MESH;
    const int m1 = mesh1 ? s.geom_dataid[g1] : 0, m2 = mesh2 ? s.geom_dataid[g2] : 0;
    r[0] = g1 | ((s.geom_type[g1] == BS_GEOM_PLANE ? 1 : 0) << 16); r[1] = g2;
    r[2] = s.nmesh > 0 ? s.mesh_vertadr[m1] : 0; r[3] = m1 * BS_NSTART;
    r[4] = s.nmesh > 0 ? s.mesh_vertadr[m2] : 0; r[5] = m2 * BS_NSTART;
    const double mg = fmax(s.geom_margin[g1], s.geom_margin[g2]);
    memcpy(r + 6, &mg, sizeof mg);
  }
  PUT_CI(pair_np, pnp.data(), pnp.size());
  PUT_CD(pair_con, pcon.data(), pcon.size());
  PUT_CI(pair_body, pbody.data(), pbody.size());
  PUT_CD(hull_vert, s.hull_vert, s.nhullvert * 3);
  PUT_CI(mesh_vertadr, s.mesh_vertadr, s.nmesh);
  PUT_CI(hull_adj_start, s.hull_adj_start, s.nhullvert > 0 ? s.nhullvert + 1 : 0);
  PUT_CI(hull_adj, s.hull_adj, s.nhulladj);
  {
    // the direction-bin tables are per hull and live on the host while they are built: refuse a model whose tables
    // would not fit rather than die in an allocation (the hand's 24 hulls: 0.23 GB)
    const size_t per_hull = (size_t)BS_NSTART * (32 + 32 * BS_CAND_K);
    if ((size_t)s.nmesh * per_hull > ((size_t)16 << 30))
      return fail(BS_ERR_NOMEM, "bs_model_create: the collision tables of %d hulls (%zu bytes per hull: 6 x %d x %d direction bins, start record + %d "
                  "support candidates each) exceed 16 GiB; compile with a smaller -DBS_SG", s.nmesh, per_hull, BS_SG, BS_SG, BS_CAND_K);
  }
  // climb warm-start table: exhaustive support vertex for the centre direction of every bin
  std::vector<int32_t> hstart((size_t)(s.nmesh > 0 ? s.nmesh : 1) * BS_NSTART, 0);
  // support candidates: per bin the vertices that are the hull's support vertex for SOME direction of the bin (the
  // vertices whose normal cone meets the bin's cone of directions).  md_support takes the best of them — no climb,
  // no verification of the neighbours.  cand[bin] = the list, centre direction's support vertex first.
  std::vector<std::vector<int32_t>> cand((size_t)(s.nmesh > 0 ? s.nmesh : 1) * BS_NSTART);
  // does the normal cone of vertex y (directions d with d . (y - n) >= 0 for every neighbour n) meet the bin?  On a
  // cube face a direction is (s, u, v) up to the order of the components, the bin a rectangle in (u, v) and every
  // neighbour a half-plane there: the rectangle is clipped by them one after the other.  Tolerances make the answer
  // err towards "meets" (a candidate too many costs a comparison, one too few a wrong contact).
  auto cone_meets_bin = [&](int me, int y, int bin) -> bool {
    const int va = s.mesh_vertadr[me];
    const double* V = s.hull_vert + 3 * (size_t)va;
    const int iv = bin % BS_SG, iu = (bin / BS_SG) % BS_SG, face = bin / (BS_SG * BS_SG);
    const double eps = 1e-9, sg = (face & 1) ? -1.0 : 1.0;
    const double u0 = ((double)iu / BS_SG - 0.5) * 2 - eps, u1 = ((double)(iu + 1) / BS_SG - 0.5) * 2 + eps;
    const double v0 = ((double)iv / BS_SG - 0.5) * 2 - eps, v1 = ((double)(iv + 1) / BS_SG - 0.5) * 2 + eps;
    constexpr int CAP = 96;
    double pu[CAP] = {u0, u1, u1, u0}, pv[CAP] = {v0, v0, v1, v1}, qu[CAP], qv[CAP];
    int n = 4;
    for (int k = s.hull_adj_start[va + y]; k < s.hull_adj_start[va + y + 1]; k++) {
      const int nb = s.hull_adj[k];
      const double e[3] = {V[3 * y] - V[3 * nb], V[3 * y + 1] - V[3 * nb + 1], V[3 * y + 2] - V[3 * nb + 2]};
      double A, B, C;
      if (face < 2) { C = sg * e[0]; A = e[1]; B = e[2]; }
      else if (face < 4) { A = e[0]; C = sg * e[1]; B = e[2]; }
      else { A = e[0]; B = e[1]; C = sg * e[2]; }
      const double tol = 1e-11 * sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + 1e-300;
      if (n + 2 > CAP) return true;
      int mq = 0;
      for (int i = 0; i < n; i++) {
        const int j = i + 1 < n ? i + 1 : 0;
        const double fi = A * pu[i] + B * pv[i] + C + tol, fj = A * pu[j] + B * pv[j] + C + tol;
        if (fi >= 0) { qu[mq] = pu[i]; qv[mq] = pv[i]; mq++; }
        if ((fi >= 0) != (fj >= 0)) {
          const double t = fi / (fi - fj);
          qu[mq] = pu[i] + t * (pu[j] - pu[i]); qv[mq] = pv[i] + t * (pv[j] - pv[i]); mq++;
        }
      }
      if (mq == 0) return false;
      n = mq;
      for (int i = 0; i < n; i++) { pu[i] = qu[i]; pv[i] = qv[i]; }
    }
    return true;
  };
  {
    // exhaustive (lowest index wins a tie), 6 x 64 x 64 bins per hull: a few hundred million multiply-adds
    // per model, spread over up to eight host threads
    auto fill = [&](int me, int b0, int b1) {
      const double* V = s.hull_vert + 3 * (size_t)s.mesh_vertadr[me];
      for (int bin = b0; bin < b1; bin++) {
        double dd[3];
        bin_dir(bin, dd);
        int best = 0;
        double bv = -1e300;
        for (int v = 0; v < s.mesh_vertnum[me]; v++) {
          const double val = V[3 * v] * dd[0] + V[3 * v + 1] * dd[1] + V[3 * v + 2] * dd[2];
          if (val > bv) { bv = val; best = v; }
        }
        hstart[(size_t)me * BS_NSTART + bin] = best;
        // the vertices of the bin form a connected piece of the hull's graph around `best`: breadth-first from there
        std::vector<int32_t>& cl = cand[(size_t)me * BS_NSTART + bin];
        cl.assign(1, best);
        const int va = s.mesh_vertadr[me];
        for (size_t at = 0; at < cl.size() && cl.size() < BS_CAND_MAX; at++)
          for (int k = s.hull_adj_start[va + cl[at]]; k < s.hull_adj_start[va + cl[at] + 1] && cl.size() < BS_CAND_MAX; k++) {
            const int y = s.hull_adj[k];
            bool seen = false;
            for (int32_t z : cl) seen |= z == y;
            if (!seen && cone_meets_bin(me, y, bin)) cl.push_back(y);
          }
      }
    };
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : nt > 8 ? 8 : nt;
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < nt; k++)
      pool.emplace_back([&, k]() {
        for (int me = 0; me < s.nmesh; me++) {
          const int per = (BS_NSTART + (int)nt - 1) / (int)nt, b0 = (int)k * per, b1 = b0 + per < BS_NSTART ? b0 + per : BS_NSTART;
          if (b0 < b1) fill(me, b0, b1);
        }
      });
    for (auto& th : pool) th.join();
  }
  PUT_CI(hull_start, hstart.data(), hstart.size());
  std::vector<double> start4(hstart.size() * 4, 0.0);
  for (int me = 0; me < s.nmesh; me++)
    for (int bin = 0; bin < BS_NSTART; bin++) {
      const int va = s.mesh_vertadr[me], v = hstart[(size_t)me * BS_NSTART + bin];
      double* r = &start4[4 * ((size_t)me * BS_NSTART + bin)];
      for (int q = 0; q < 3; q++) r[q] = s.hull_vert[3 * ((size_t)va + v) + q];
      const int64_t bits = pack_rec(v, s.hull_adj_start[va + v + 1] - s.hull_adj_start[va + v], s.hull_adj_start[va + v]);
      memcpy(r + 3, &bits, sizeof bits);
    }
  fix.push_back({pk.add(start4.data(), start4.size() * sizeof(double)), (void**)&cm.hull_start4});
  // candidate records: BS_CAND_K per bin {x, y, z, -} (short lists repeat their first vertex), the first one's
  // fourth word = number of candidates | first record of the rest in hull_more4 << 32
  std::vector<double> cand4(hstart.size() * 4 * BS_CAND_K, 0.0), more4;
  for (int me = 0; me < s.nmesh; me++)
    for (int bin = 0; bin < BS_NSTART; bin++) {
      const std::vector<int32_t>& cl = cand[(size_t)me * BS_NSTART + bin];
      const int va = s.mesh_vertadr[me];
      double* r = &cand4[4 * BS_CAND_K * ((size_t)me * BS_NSTART + bin)];
      for (int q = 0; q < BS_CAND_K; q++) {
        const int v = cl[q < (int)cl.size() ? q : 0];
        for (int k = 0; k < 3; k++) r[4 * q + k] = s.hull_vert[3 * ((size_t)va + v) + k];
      }
      const int64_t bits = (int64_t)cl.size() | ((int64_t)(more4.size() / 4) << 32);
      memcpy(r + 3, &bits, sizeof bits);
      for (size_t q = BS_CAND_K; q < cl.size(); q++) {
        for (int k = 0; k < 3; k++) more4.push_back(s.hull_vert[3 * ((size_t)va + cl[q]) + k]);
        more4.push_back(0.0);
      }
    }
  for (int q = 0; q < 4 * BS_CAND_K; q++) more4.push_back(0.0);  // a clamped read beyond the last list stays inside
  fix.push_back({pk.add(cand4.data(), cand4.size() * sizeof(double)), (void**)&cm.hull_cand4});
  fix.push_back({pk.add(more4.data(), more4.size() * sizeof(double)), (void**)&cm.hull_more4});
  // hill-climbing records: neighbour {x, y, z, id} per adjacency entry, padded vertices, ranges
  std::vector<double> nbr4((size_t)(s.nhulladj > 0 ? s.nhulladj : 1) * 4, 0.0);
  std::vector<double> vert4((size_t)(s.nhullvert > 0 ? s.nhullvert : 1) * 4, 0.0);
  std::vector<int32_t> range((size_t)(s.nhullvert > 0 ? s.nhullvert : 1) * 2, 0);
  for (int me = 0; me < s.nmesh; me++) {
    const int va = s.mesh_vertadr[me];
    for (int v = 0; v < s.mesh_vertnum[me]; v++) {
      for (int q = 0; q < 3; q++) vert4[4 * ((size_t)va + v) + q] = s.hull_vert[3 * ((size_t)va + v) + q];
      {
        const int64_t bits = pack_rec(v, s.hull_adj_start[va + v + 1] - s.hull_adj_start[va + v], s.hull_adj_start[va + v]);
        memcpy(&vert4[4 * ((size_t)va + v) + 3], &bits, sizeof bits);
      }
      range[2 * ((size_t)va + v)] = s.hull_adj_start[va + v];
      range[2 * ((size_t)va + v) + 1] = s.hull_adj_start[va + v + 1];
      for (int k = s.hull_adj_start[va + v]; k < s.hull_adj_start[va + v + 1]; k++) {
        for (int q = 0; q < 3; q++) nbr4[4 * (size_t)k + q] = s.hull_vert[3 * ((size_t)va + s.hull_adj[k]) + q];
        const int nb = s.hull_adj[k];  // named vertex: id | degree << 20 | first adjacency entry << 32
        const int64_t bits = pack_rec(nb, s.hull_adj_start[va + nb + 1] - s.hull_adj_start[va + nb], s.hull_adj_start[va + nb]);
        memcpy(&nbr4[4 * (size_t)k + 3], &bits, sizeof bits);
      }
    }
  }
  fix.push_back({pk.add(nbr4.data(), nbr4.size() * sizeof(double)), (void**)&cm.hull_nbr4});
  fix.push_back({pk.add(vert4.data(), vert4.size() * sizeof(double)), (void**)&cm.hull_vert4});
  fix.push_back({pk.add(range.data(), range.size() * sizeof(int32_t)), (void**)&cm.hull_range});
  // box around each mesh geom's hull in the geom frame (broadphase separating-axis filter)
  std::vector<double> obb((size_t)(ng > 0 ? ng : 1) * 6, -1.0);
  for (int g = 0; g < ng; g++) {
    const int me = s.geom_dataid[g];
    if (s.geom_type[g] != BS_GEOM_MESH || me < 0 || me >= s.nmesh || s.mesh_vertnum[me] < 1) continue;
    const double* V = s.hull_vert + 3 * (size_t)s.mesh_vertadr[me];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < s.mesh_vertnum[me]; v++)
      for (int k = 0; k < 3; k++) {
        lo[k] = V[3 * v + k] < lo[k] ? V[3 * v + k] : lo[k];
        hi[k] = V[3 * v + k] > hi[k] ? V[3 * v + k] : hi[k];
      }
    for (int k = 0; k < 3; k++) { obb[6 * g + k] = 0.5 * (lo[k] + hi[k]); obb[6 * g + 3 + k] = 0.5 * (hi[k] - lo[k]); }
  }
  PUT_CD(geom_obb, obb.data(), obb.size());
  // broadphase record of a pair: both geoms, which bounding test applies and its threshold (k_tp_col's first pass)
  std::vector<int32_t> pbp((size_t)(s.npair > 0 ? s.npair : 1) * 4, 0);
  for (int p = 0; p < s.npair; p++) {
    const int g1 = s.pair_geom1[p], g2 = s.pair_geom2[p];
    const double margin = fmax(s.geom_margin[g1], s.geom_margin[g2]);
    const double r1 = s.geom_rbound[g1], r2 = s.geom_rbound[g2];
    int kind = 0, has_box = 0;
    double thresh = 0.0;
    if (r1 > 0 && r2 > 0) {
      kind = 1; thresh = (margin + r1 + r2) * (margin + r1 + r2);  // the SQUARED distance is compared (no square root per pair)
      has_box = obb[6 * g1 + 3] >= 0 && obb[6 * g2 + 3] >= 0;
    } else if (s.geom_type[g1] == BS_GEOM_PLANE && r2 > 0) {
      kind = 2; thresh = margin + r2;
    }
    int32_t* r = &pbp[(size_t)p * 4];
    r[0] = g1 | (g2 << 12) | (kind << 24) | (has_box << 26);
    memcpy(r + 2, &thresh, sizeof thresh);
  }
  PUT_CI(pair_bp, pbp.data(), pbp.size());
  std::vector<double> gsize((size_t)(ng > 0 ? ng : 1) * 3, 0.0);
  if (s.geom_size) memcpy(gsize.data(), s.geom_size, (size_t)ng * 3 * sizeof(double));
  PUT_CD(geom_size, gsize.data(), gsize.size());
  // sub-contacts per candidate pair (0: geom types no collision routine handles, e.g. height fields)
  std::vector<int32_t> nsub((size_t)(s.npair > 0 ? s.npair : 1), 0);
  cm.has_prim = 0;
  for (int p = 0; p < s.npair; p++) {
    const int t1 = s.geom_type[s.pair_geom1[p]], t2 = s.geom_type[s.pair_geom2[p]];
    nsub[p] = pair_supported_types(t1, t2) ? pair_nsub_types(t1, t2) : 0;
    if (nsub[p] && !((t1 == BS_GEOM_PLANE || t1 == BS_GEOM_MESH) && t2 == BS_GEOM_MESH)) cm.has_prim = 1;
  }
  PUT_CI(pair_nsub, nsub.data(), nsub.size());
#undef PUT_CI
#undef PUT_CD
  TpModel tpd[2];
  // row of the packed lower-triangle entry q = l + 32 t (k_tpc's dense Cholesky deals the entries round-robin);
  // the wide build reads a byte per entry instead (33 chunks of 6 bits do not fit two words)
  std::vector<uint64_t> tri_pk(2 * 32, 0);
  for (int l = 0; l < 32; l++)
    for (int tt = 0; tt < TPC_NTRI; tt++) {
      const int q = l + tt * 32;
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= q) i++;
      if (tt < 12) tri_pk[l] |= (uint64_t)(i & 31) << (5 * tt);
      else tri_pk[32 + l] |= (uint64_t)(i & 31) << (5 * (tt - 12));
    }
  std::vector<uint8_t> tri_row(64 * 65 / 2 + 64, 0);
  for (int i = 0, q = 0; i < 64; i++)
    for (int k = 0; k <= i; k++) tri_row[q++] = (uint8_t)i;
  for (int v = 0; v < 2; v++) {
    const bs_model::TpHost& h = m->tpv[v];
    TpModel& tp = tpd[v];
    tp = h.tp;
    tp.lane = nullptr; tp.lane_d = nullptr; tp.lane_k = nullptr; tp.pgs_order = nullptr; tp.tri_pk = nullptr; tp.tri_row = nullptr;
    if (!h.tp_ok) continue;
    fix.push_back({pk.add(tri_pk.data(), tri_pk.size() * sizeof(uint64_t)), (void**)&tp.tri_pk});
    fix.push_back({pk.add(tri_row.data(), tri_row.size()), (void**)&tp.tri_row});
    fix.push_back({pk.add(h.tp_pgs.data(), h.tp_pgs.size() * sizeof(int32_t)), (void**)&tp.pgs_order});
    fix.push_back({pk.add(h.tp_table.data(), h.tp_table.size() * sizeof(int32_t)), (void**)&tp.lane});
    fix.push_back({pk.add(h.tp_dtable.data(), h.tp_dtable.size() * sizeof(double)), (void**)&tp.lane_d});
    fix.push_back({pk.add(h.tp_ktable.data(), h.tp_ktable.size() * sizeof(double)), (void**)&tp.lane_k});
  }
#undef PUT_I
#undef PUT_D
  HIPCHK(hipSetDevice(device));
  void* blob = nullptr;
  HIPCHK(hipMalloc(&blob, pk.bytes.size()));
  {
    const hipError_t ce = hipMemcpy(blob, pk.bytes.data(), pk.bytes.size(), hipMemcpyHostToDevice);
    if (ce != hipSuccess) {
      (void)hipFree(blob);
      return fail(BS_ERR_HIP, "model upload (%zu bytes) failed: %s", pk.bytes.size(), hipGetErrorString(ce));
    }
  }
  for (auto& f : fix) *f.slot = (char*)blob + f.off;
  DeviceImage im;
  im.device = device; im.blob = blob; im.dm = dm; im.tp[0] = tpd[0]; im.tp[1] = tpd[1]; im.cm = cm;
  m->images.push_back(im);
  *out = &m->images.back();
  return BS_OK;
}

// ------------------------------------------------------------------ sim
struct FieldInfo {
  void* ptr = nullptr;
  int rows = 0;
  bool is_int = false;
};

struct bs_sim {
  bs_model* model = nullptr;
  DevModel dm;
  TpModel tp;
  int wide = 0, lpe = 32;  // which tree-parallel build runs this sim: 32 lanes per environment, or the wide one (64)
  bool use_tp = false;
  bool narrow_ws = false, narrow_blocks = false;
  int nl_grid = 2048;  // wavefronts of the persistent narrowphase (k_c_narrow_list): one resident set at two per SIMD
  bool use_tpc = false;  // contacts: tree-parallel dynamics kernel instead of the serial k_c_dyn (steps only)
  bool contacts = false;
  ColModel cm;
  ColData cd;
  ConData kd;
  void* carena = nullptr;
  size_t hs_bytes = 0;  // dynamic LDS of k_c_dyn (0 = Hessian in the global arena)
  DevData dd;
  int device = 0;
  long n_envs = 0, E = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  void* arena = nullptr;
  size_t arena_bytes = 0, carena_bytes = 0, gframe_bytes = 0, sarena_bytes = 0;
  FieldInfo fields[BS_F_COUNT];
  double* derived = nullptr;  // scratch for on-demand fields (xmat, geom_xpos, geom_xmat)
  size_t derived_rows = 0;
  hipEvent_t ev[8] = {};
  bool events = false;
  // bs_reset with a host mask: pinned staging + device copy, allocated once (no malloc / sync per call);
  // mask_ev marks the end of the last upload, so the staging buffer is not overwritten under it
  uint8_t* mask_pinned = nullptr;
  uint8_t* mask_dev = nullptr;
  hipEvent_t mask_ev = nullptr;
  bool mask_pending = false;
  size_t serial_rows = 0;   // rows of the serial contact solver's buffers (allocated on first use)
  void* sarena = nullptr;   // cJ / Hd / cW of the serial contact solver (ensure_serial_contact)
  void* gframe = nullptr;   // ColData::gframe: geom frames as [E][ngeom][12] for the persistent narrowphase
  // mixed-model batch: one child sim per model over a contiguous range of environments (each with
  // its own arenas, since row counts differ between models); all run on this sim's stream
  // two halves of a contact batch on two streams: kernels of different kinds overlap (both the
  // narrowphase and k_tpc leave the SIMDs half idle on their own)
  hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr, join_ev[3] = {nullptr, nullptr, nullptr};
  int split = 1;  // number of ranges (1 = no split)
  std::vector<bs_sim*> parts;
  std::vector<int64_t> part_start;
};
#define NO_PARTS(s, what)                                                                             \
  if (!(s)->parts.empty())                                                                            \
    return fail(BS_ERR_UNSUPPORTED, what ": a mixed-model sim has one field set per model; use bs_sim_part")


extern "C" int bs_sim_destroy(bs_sim* s);
// size of one range of a split contact batch: a whole number of 64-environment blocks
static inline long split_chunk(long n_envs, int np) { return (n_envs / np + 63) / 64 * 64; }
// Jacobian rows, dense Hessian and PGS rows of the serial contact solver (k_c_dyn): 43 KB per
// environment, so they are allocated when the serial dynamics kernel is first needed (sims on the
// serial path: at creation; sims on the tree-parallel path: the first bs_forward / bs_stage_times)
static int ensure_serial_contact(bs_sim* s) {
  if (!s->contacts || s->sarena) return BS_OK;
  const size_t nv = (size_t)s->dm.nv, nr = 4 * (size_t)s->cm.maxcon, nh = nv * (nv + 1) / 2, E = (size_t)s->E;
  const bool pgs = s->dm.solver == 0;
  const size_t rows = nr * nv + nr * 5 + nh + (pgs ? nr * nv : 0);
  const hipError_t he = hipMalloc(&s->sarena, rows * E * sizeof(double));
  if (he != hipSuccess) {
    (void)hipGetLastError();  // a failed allocation leaves its error behind: the next call must not trip over it
    s->sarena = nullptr;
    return fail(BS_ERR_NOMEM, "serial contact solver arena (Jacobian rows + dense Hessian, %zu bytes per environment): hipMalloc of %zu bytes for %ld environments failed: %s",
                rows * sizeof(double), rows * E * sizeof(double), s->n_envs, hipGetErrorString(he));
  }
  s->sarena_bytes = rows * E * sizeof(double);
  HIPCHK(hipMemsetAsync(s->sarena, 0, rows * E * sizeof(double), s->stream));
  double* dp = (double*)s->sarena;
  auto take = [&](size_t rows_) { double* q = dp; dp += rows_ * E; return q; };
  ConData& k = s->kd;
  k.cJ = take(nr * nv); k.cD = take(nr); k.caref = take(nr); k.cjar = take(nr); k.cjv = take(nr); k.cforce = take(nr);
  k.Hd = take(nh);
  k.cW = pgs ? take(nr * nv) : nullptr;
  // dense Hessian in LDS when (nv(nv+1)/2 + nv) x 64 lanes x 8 B fits one CU's 160 KB
  const size_t want = (nh + nv) * BS_BLOCK * sizeof(double);
  if (want <= 160 * 1024 && getenv("BS_LDS_HESSIAN")) {  // opt-in: lowers occupancy to one wave per CU
    if (bs_c_dyn_set_lds((int)want) == hipSuccess) s->hs_bytes = want;
    else (void)hipGetLastError();
  }
  return BS_OK;
}
static inline dim3 grid_for(const bs_sim* s) { return dim3((unsigned)((s->n_envs + BS_BLOCK - 1) / BS_BLOCK)); }
// tree-parallel kernels: one wavefront per pair of environments, the grid rounded up to whole rounds of the 8 XCDs
// (physics_tp.h::tp_block deals each XCD a contiguous run of pairs)
static inline dim3 grid_tp_n(long n_envs, int lpe) {
  const long per = 64 / lpe;  // environments per wavefront
  return dim3((unsigned)(((n_envs + per - 1) / per + 7) / 8 * 8));
}
static inline dim3 grid_tp(const bs_sim* s) { return grid_tp_n(s->n_envs, s->lpe); }
// narrowphase: flattened work items by default; BS_NARROW=ws selects the work-stealing variant
// (measured slower on MI355X at 65536 envs: 2.6 ms vs 1.9 ms)
static void launch_narrow(bs_sim* s, const DevData& dd, hipStream_t st) {
  const dim3 g((unsigned)((dd.n_envs - dd.env0 + NB_ENVS - 1) / NB_ENVS));
  if (s->narrow_ws)
    bs_launch_narrow_ws(g, st, s->cm, dd, s->cd);  // mesh / plane pairs only (bs_sim_create falls back to blocks otherwise)
  else
    bs_launch_narrow_flat(g, st, s->cm, dd, s->cd, s->cm.has_prim);
}
// one forward pass (mode 1) or step (mode 0) with contacts through the tree-parallel kernels, for
// the environments [dd.env0, dd.n_envs) on stream st
static void launch_tpc_pass(bs_sim* s, const DevData& dd, hipStream_t st, int mode, int dump, bool full = false) {
#ifdef TPC_CUTS
  if (const char* e = getenv("BS_CUT")) dump |= (atoi(e) + 1) << 8;
  extern void bs_set_colcut(int);
  if (const char* e = getenv("BS_COLCUT")) bs_set_colcut(atoi(e));
#endif
  const dim3 g = grid_tp_n(dd.n_envs - dd.env0, s->lpe);
  const size_t nw = (size_t)(64 / s->lpe);  // environments per wavefront
  const size_t lds_col = nw * s->cm.ngeom * TPG_SLOT * sizeof(double) + nw * s->cm.npair * sizeof(int)  // frames, pair lists,
                         + nw * 2 * s->cm.maxcand * sizeof(int);                                         // slot records
  (s->wide ? bs_launch_tp_col_w : bs_launch_tp_col)(g, lds_col, st, s->dm, s->tp, dd, s->cm, s->cd);
  if (s->narrow_ws || s->narrow_blocks) {
    launch_narrow(s, dd, st);
  } else {
    size_t nb = ((size_t)(dd.n_envs - dd.env0) * s->cm.maxcand + NL_THREADS - 1) / NL_THREADS;
    const size_t cap = s->cm.has_prim ? 16384 : (size_t)s->nl_grid;
    nb = nb < 1 ? 1 : nb > cap ? cap : nb;
    bs_launch_narrow_list(dim3((unsigned)nb), st, s->cm, dd, s->cd, s->cm.has_prim);
  }
  const int ntri = s->dm.nbody * (s->dm.nbody + 1) / 2;
  // LDS per wavefront: 64 contact slots and U rows whatever the build (2 x 32 or 1 x 64 lanes), Hessians / M^-1 per environment
  if (s->dm.solver == 0) {
    const size_t lds = (size_t)(64 * tpc_slot_pgs(s->lpe) + 64 * 3 + nw * s->dm.nbody * tp_minv_stride(s->dm.nbody)) * sizeof(double);
    auto* launch = s->wide ? (full ? bs_launch_tpc_full_pgs_w : bs_launch_tpc_pgs_w) : (full ? bs_launch_tpc_full_pgs : bs_launch_tpc_pgs);
    launch(g, lds, st, s->dm, s->tp, dd, s->cm, s->cd, mode, dump);
    return;
  }
  const size_t lds = (size_t)(nw * ntri + 64 * tpc_slot(s->lpe) + 64 * 3) * sizeof(double);
  auto* launch = s->dm.solver == 1 ? (full ? bs_launch_tpc_full_cg : bs_launch_tpc_cg) : (full ? bs_launch_tpc_full : (s->tp.chain ? bs_launch_tpc_chain : bs_launch_tpc));
  if (s->wide) launch = s->dm.solver == 1 ? (full ? bs_launch_tpc_full_cg_w : bs_launch_tpc_cg_w) : (full ? bs_launch_tpc_full_w : bs_launch_tpc_w);
  launch(g, lds, st, s->dm, s->tp, dd, s->cm, s->cd, mode, dump);
}
// check: mj_checkPos/Vel first; integrate: mj_step (Euler) rather than mj_forward; fast: let the
// tree-parallel kernels run a forward pass too (they leave the position-stage intermediates of
// bs_forward unwritten, which the RK4 stages do not need)
static void launch_contact_pass(bs_sim* s, int check, int integrate, int dump = 1, int fast = 0) {
  if (s->use_tpc && (integrate || fast)) {
    launch_tpc_pass(s, s->dd, s->stream, integrate ? 0 : 1, dump);
    return;
  }
  const dim3 g = dim3((unsigned)((s->n_envs + BS_BLOCK - 1) / BS_BLOCK));
  bs_launch_c_pos(g, s->stream, s->dm, s->dd, s->cd, check);
  bs_launch_c_broad(dim3((unsigned)s->n_envs), s->stream, s->cm, s->dd, s->cd);
  launch_narrow(s, s->dd, s->stream);
  bs_launch_c_dyn(g, s->hs_bytes, s->stream, s->dm, s->cm, s->dd, s->cd, s->kd, integrate,
                     s->hs_bytes ? 1 : 0);
}
static void launch_step(bs_sim* s, int n_steps) {
  if (s->contacts) {
    // outputs of the forward pass (poses, generalized forces) are only observable after the last step
    const bool rk = s->dm.integrator == 1;  // RK4: four forward passes (collision included), state moved in between
    if (s->use_tpc && s->split > 1) {
      // range 0 on the sim's stream, the others on auxiliary streams, joined at the end
      const int np = s->split;
      DevData part[4];
      const long chunk = split_chunk(s->n_envs, np);  // bs_sim_create made sure every range is non-empty
      for (int q = 0; q < np; q++) {
        part[q] = s->dd;
        part[q].env0 = q * chunk;
        part[q].range = q;
        part[q].n_envs = q == np - 1 || (q + 1) * chunk > s->n_envs ? s->n_envs : (q + 1) * chunk;
      }
      (void)hipEventRecord(s->fork_ev, s->stream);
      for (int q = 1; q < np; q++) (void)hipStreamWaitEvent(s->aux_stream[q - 1], s->fork_ev, 0);
      for (int k = 0; k < n_steps; k++)
        for (int i = 1; i <= (rk ? 4 : 1); i++)
          for (int q = 0; q < np; q++) {
            hipStream_t st = q == 0 ? s->stream : s->aux_stream[q - 1];
            if (!rk) {
              launch_tpc_pass(s, part[q], st, 0, k == n_steps - 1);
              continue;
            }
            const dim3 gq((unsigned)((part[q].n_envs - part[q].env0 + BS_BLOCK - 1) / BS_BLOCK));
            if (i == 1) bs_launch_check_state(gq, st, s->dm, part[q]);
            launch_tpc_pass(s, part[q], st, 1, i == 4 && k == n_steps - 1);
            bs_launch_rk4_advance(gq, st, s->dm, part[q], i);
          }
      for (int q = 1; q < np; q++) {
        (void)hipEventRecord(s->join_ev[q - 1], s->aux_stream[q - 1]);
        (void)hipStreamWaitEvent(s->stream, s->join_ev[q - 1], 0);
      }
      return;
    }
    if (rk) {
      for (int k = 0; k < n_steps; k++)
        for (int i = 1; i <= 4; i++) {
          if (i == 1) bs_launch_check_state(grid_for(s), s->stream, s->dm, s->dd);
          launch_contact_pass(s, 0, 0, i == 4 && k == n_steps - 1, 1);
          bs_launch_rk4_advance(grid_for(s), s->stream, s->dm, s->dd, i);
        }
      return;
    }
    for (int k = 0; k < n_steps; k++) launch_contact_pass(s, 1, 1, k == n_steps - 1);
    return;
  }
  if (s->use_tp) {
    const bool rk = s->dm.integrator == 1;
    auto* launch = s->dm.solver == 0 ? (rk ? bs_launch_tp_rk4_pgs : bs_launch_tp_pgs)
                   : s->dm.solver == 1 ? (rk ? bs_launch_tp_rk4_cg : bs_launch_tp_cg)
                                       : (rk ? bs_launch_tp_rk4 : (s->tp.chain ? (s->tp.quad_top >= 0 ? bs_launch_tp_chain4 : bs_launch_tp_chain) : bs_launch_tp));
    if (s->wide)
      launch = s->dm.solver == 0 ? (rk ? bs_launch_tp_rk4_pgs_w : bs_launch_tp_pgs_w)
               : s->dm.solver == 1 ? (rk ? bs_launch_tp_rk4_cg_w : bs_launch_tp_cg_w) : (rk ? bs_launch_tp_rk4_w : bs_launch_tp_w);
    int dump = 1;
#ifdef TPC_CUTS
    if (const char* e = getenv("BS_CUT")) dump |= (atoi(e) + 1) << 8;
#endif
    launch(grid_tp(s), s->dm.solver == 0 ? (size_t)(64 / s->lpe) * s->dm.nbody * tp_minv_stride(s->dm.nbody) * sizeof(double) : 0, s->stream,
           s->dm, s->tp, s->dd, n_steps, 0, dump);
  } else {
    bs_launch_step(grid_for(s), s->stream, s->dm, s->dd, n_steps);
  }
}

static thread_local int tl_part_of_mixed = 0;  // bs_sim_create is creating a part of a mixed-model sim
extern "C" int bs_sim_create(const bs_model* const* models, int n_models, const int32_t* env_model_id,
                             int64_t n_envs, int device, const bs_options* opt, bs_sim** out) {
  (void)opt;
  if (!models || !out || n_models < 1 || !models[0]) return fail(BS_ERR_INVALID, "bs_sim_create: null argument");
  if (n_envs < 1) return fail(BS_ERR_INVALID, "bs_sim_create: n_envs must be >= 1 (got %lld)", (long long)n_envs);
  if (env_model_id)
    for (int64_t i = 0; i < n_envs; i++) {
      if (env_model_id[i] < 0 || env_model_id[i] >= n_models)
        return fail(BS_ERR_INVALID, "bs_sim_create: env_model_id[%lld] out of range", (long long)i);
      if (i > 0 && env_model_id[i] < env_model_id[i - 1])
        return fail(BS_ERR_INVALID, "bs_sim_create: env_model_id must be non-decreasing (environments sorted by model)");
    }
  if (n_models > 1) {
    if (!env_model_id) return fail(BS_ERR_INVALID, "bs_sim_create: n_models > 1 needs env_model_id");
    for (int k = 0; k < n_models; k++)
      if (!models[k]) return fail(BS_ERR_INVALID, "bs_sim_create: models[%d] is null", k);
    bs_sim* top = new bs_sim();
    top->device = device; top->n_envs = n_envs;
    int64_t i = 0;
    for (int k = 0; k < n_models; k++) {
      const int64_t start = i;
      while (i < n_envs && env_model_id[i] == k) i++;
      if (i == start) continue;  // no environment uses this model
      bs_sim* child = nullptr;
      tl_part_of_mixed = 1;  // the parts already step concurrently on three streams: a part keeps the old split threshold
      bs_options po;
      memset(&po, 0, sizeof po);
      if (opt) po = *opt;
      if (opt && opt->max_contacts_per_model && opt->max_contacts_per_model[k] > 0) po.max_contacts_per_env = opt->max_contacts_per_model[k];
      po.max_contacts_per_model = nullptr;
      const int rc = bs_sim_create(models + k, 1, nullptr, i - start, device, &po, &child);
      tl_part_of_mixed = 0;
      if (rc) { bs_sim_destroy(top); return rc; }
      top->parts.push_back(child);
      top->part_start.push_back(start);
    }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&top->own_stream, hipStreamNonBlocking) != hipSuccess) {
      bs_sim_destroy(top);
      return fail(BS_ERR_HIP, "bs_sim_create: cannot create the stream of the mixed-model sim");
    }
    top->stream = top->own_stream;
    for (bs_sim* c : top->parts) c->stream = top->stream;
    // parts 1 and 2 step on their own streams, forked from / joined to the sim's stream by these events: a
    // part of a few thousand environments leaves most of the machine idle (BS_MIXED_SERIAL=1: one after the other)
    const char* ms = getenv("BS_MIXED_SERIAL");
    if (top->parts.size() > 1 && !(ms && atoi(ms))) {
      bool ok = hipEventCreateWithFlags(&top->fork_ev, hipEventDisableTiming) == hipSuccess;
      for (int q = 0; q < 2 && ok; q++) ok = hipEventCreateWithFlags(&top->join_ev[q], hipEventDisableTiming) == hipSuccess;
      if (!ok) {
        bs_sim_destroy(top);
        return fail(BS_ERR_HIP, "bs_sim_create: cannot create the events of the mixed-model sim");
      }
      top->split = 2;  // marks the concurrent mode
    }
    *out = top;
    return BS_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    (void)hipGetLastError();
    return fail(BS_ERR_NODEVICE, "bs_sim_create: no HIP device available (libbatchsim has no CPU fallback)");
  }
  if (device < 0 || device >= ndev) return fail(BS_ERR_INVALID, "bs_sim_create: device %d out of range [0,%d)", device, ndev);
  bs_model* model = const_cast<bs_model*>(models[0]);
  const bool contacts = !(model->d.disableflags & BS_DSBL_CONTACT) && !(model->d.disableflags & BS_DSBL_CONSTRAINT) &&
                        model->d.npair > 0;
  const DeviceImage* img = nullptr;
  int rc = model_image(model, device, &img);
  if (rc) return rc;

  bs_sim* s = new bs_sim();
  s->model = model; s->dm = img->dm; s->device = device; s->n_envs = n_envs;
  // every failure below goes through bs_sim_destroy, which tolerates a partially built sim
#define HIPCHK_S(call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      bs_sim_destroy(s);                                                                            \
      return fail(BS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                               \
  } while (0)
  const char* kv = getenv("BS_KERNEL");
  s->contacts = contacts;
  s->cm = img->cm;
  s->cm.maxcon = (opt && opt->max_contacts_per_env > 0) ? opt->max_contacts_per_env : 32;
  {
    // candidate slots: one per sub-contact of every pair (plane-capsule 2, plane-box 8, else 1), at most 256
    // (the work list packs the slot into 8 bits)
    long slots = 0;
    for (int p = 0; p < model->d.npair; p++) {
      const int t1 = model->d.geom_type[model->d.pair_geom1[p]], t2 = model->d.geom_type[model->d.pair_geom2[p]];
      slots += pair_supported_types(t1, t2) ? pair_nsub_types(t1, t2) : 0;
    }
    s->cm.maxcand = slots < 1 ? 1 : (slots < 256 ? (int)slots : 256);
  }
  // every solver with either integrator, with and without contacts (PGS with contacts: Euler)
  const bool serial = kv && strcmp(kv, "serial") == 0;
  const bool tp_solver = true;  // Newton, CG and PGS all have tree-parallel contact kernels
  // which tree-parallel build: the 32-lane kernels when the model fits them and (with contacts) at most 32 contacts per
  // environment are asked for; else the wide ones (nbody <= 63, up to 64 contacts; BS_KERNEL=wide forces them); else serial
  {
    const bool force_wide = kv && strcmp(kv, "wide") == 0;
    const bool fits32 = model->tpv[0].tp_ok && (!contacts || s->cm.maxcon <= 32) && !force_wide;
    const bool fits64 = model->tpv[1].tp_ok && (!contacts || s->cm.maxcon <= 64);
    s->wide = !fits32 && fits64;
    s->lpe = s->wide ? 64 : 32;
    s->tp = img->tp[s->wide];
    const bool ok = fits32 || fits64;
    s->use_tp = !contacts && ok && !serial;
    s->use_tpc = contacts && tp_solver && ok && !serial;
  }
  {
    const char* nk = getenv("BS_NARROW");
    s->narrow_ws = nk && strcmp(nk, "ws") == 0 && !s->cm.has_prim;  // the work-stealing variant knows hulls and planes only
    s->narrow_blocks = nk && strcmp(nk, "blocks") == 0;
    const char* sh = getenv("BS_NL_SHARE");  // tuning knobs of the persistent narrowphase
    s->cm.nl_min_share = sh && atoi(sh) > 0 ? atoi(sh) : NL_MIN_SHARE;
    const char* gr = getenv("BS_NL_GRID");
    if (gr && atoi(gr) > 0) s->nl_grid = atoi(gr);
  }
  if (kv && (strcmp(kv, "tp") == 0 || strcmp(kv, "wide") == 0) && !(contacts ? s->use_tpc : s->use_tp)) {
    const std::string why = model->tpv[1].tp_ok ? "more contacts per environment requested than a lane group holds" : model->tpv[1].tp_why;
    delete s;
    return fail(BS_ERR_UNSUPPORTED, "BS_KERNEL=%s but the model is not eligible: %s", kv, why.c_str());
  }
  s->E = (n_envs + 63) / 64 * 64;
  const DevModel& m = s->dm;
  const long E = s->E;
  HIPCHK_S(hipSetDevice(device));
  HIPCHK_S(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
  s->stream = s->own_stream;

  // arena layout: all double rows, then the int rows
  struct Slot { double** p; int rows; int field; };
  DevData& d = s->dd;
  memset(&d, 0, sizeof d);
  d.E = E; d.n_envs = n_envs;
  const int nb = m.nbody, nj = m.njnt, nv = m.nv, nu = m.nu, nM = m.nM, nrow = m.nrow;
  Slot slots[] = {
      {&d.qpos, nj, BS_F_QPOS}, {&d.qvel, nv, BS_F_QVEL}, {&d.ctrl, nu, BS_F_CTRL},
      {&d.qfrc_applied, nv, BS_F_QFRC_APPLIED}, {&d.qacc, nv, BS_F_QACC},
      {&d.qacc_warmstart, nv, BS_F_QACC_WARMSTART}, {&d.time, 1, BS_F_TIME},
      {&d.xpos, 3 * nb, BS_F_XPOS}, {&d.xquat, 4 * nb, BS_F_XQUAT}, {&d.xipos, 3 * nb, BS_F_XIPOS},
      {&d.xanchor, 3 * nj, BS_F_XANCHOR}, {&d.xaxis, 3 * nj, BS_F_XAXIS}, {&d.rootcom, 3 * nb, BS_F_ROOTCOM},
      {&d.cinert, 10 * nb, BS_F_CINERT}, {&d.crb, 10 * nb, BS_F_CRB}, {&d.cdof, 6 * nv, BS_F_CDOF},
      {&d.qM, nM, BS_F_QM}, {&d.qLD, nM, BS_F_QLD}, {&d.qLDiagInv, nv, BS_F_QLDIAGINV},
      {&d.cvel, 6 * nb, BS_F_CVEL}, {&d.cacc, 6 * nb, -1}, {&d.cfrc, 6 * nb, -1},
      {&d.qfrc_bias, nv, BS_F_QFRC_BIAS}, {&d.qfrc_passive, nv, BS_F_QFRC_PASSIVE},
      {&d.actuator_force, nu, BS_F_ACTUATOR_FORCE}, {&d.qfrc_actuator, nv, BS_F_QFRC_ACTUATOR},
      {&d.qfrc_smooth, nv, BS_F_QFRC_SMOOTH}, {&d.qacc_smooth, nv, BS_F_QACC_SMOOTH},
      {&d.qfrc_constraint, nv, BS_F_QFRC_CONSTRAINT},
      {&d.efc_D, nrow, BS_F_EFC_D}, {&d.efc_aref, nrow, BS_F_EFC_AREF}, {&d.efc_force, nrow, BS_F_EFC_FORCE},
      {&d.jar, nrow, -1}, {&d.jv, nrow, -1}, {&d.eq_coef, m.neq, -1}, {&d.fric_rf, m.nfric, -1},
      {&d.Ma, nv, -1}, {&d.grad, nv, -1}, {&d.Mgrad, nv, -1}, {&d.search, nv, -1}, {&d.Mv, nv, -1},
      {&d.H, nM, -1}, {&d.Hdinv, nv, -1},
      {&d.rk_q0, nj, -1}, {&d.rk_v0, nv, -1}, {&d.rk_Fv, 4 * nv, -1}, {&d.rk_Fa, 4 * nv, -1}, {&d.rk_t0, 2, -1},
      {&d.pgs_W, m.solver == 0 ? nrow * nv : 0, -1}};
  size_t rows = 0;
  for (auto& sl : slots) rows += (size_t)(sl.rows > 0 ? sl.rows : 1);
  s->derived_rows = (size_t)9 * nb + (size_t)12 * (m.ngeom > 0 ? m.ngeom : 1);
  const size_t dbl_bytes = (rows + s->derived_rows) * (size_t)E * sizeof(double);
  const size_t int_bytes = 3 * (size_t)E * sizeof(int);
  s->arena_bytes = dbl_bytes + int_bytes + (size_t)E;  // + the device copy of bs_reset's mask
  hipError_t he = hipMalloc(&s->arena, s->arena_bytes);
  if (he != hipSuccess) {
    (void)hipGetLastError();  // a failed allocation leaves its error behind: the next call must not trip over it
    const size_t want = s->arena_bytes;
    s->arena = nullptr;
    bs_sim_destroy(s);
    return fail(BS_ERR_NOMEM, "bs_sim_create: state arena (every [rows][E] field, %zu bytes per environment): hipMalloc of %zu bytes for %lld environments failed: %s",
                want / (size_t)E, want, (long long)n_envs, hipGetErrorString(he));
  }
  HIPCHK_S(hipMemset(s->arena, 0, s->arena_bytes));
  double* base = (double*)s->arena;
  size_t r = 0;
  for (auto& sl : slots) {
    *sl.p = base + r * (size_t)E;
    if (sl.field >= 0) { s->fields[sl.field].ptr = *sl.p; s->fields[sl.field].rows = sl.rows; }
    r += (size_t)(sl.rows > 0 ? sl.rows : 1);
  }
  s->derived = base + r * (size_t)E;
  s->fields[BS_F_XMAT] = {s->derived, 9 * nb, false};
  s->fields[BS_F_GEOM_XPOS] = {s->derived + (size_t)9 * nb * E, 3 * m.ngeom, false};
  s->fields[BS_F_GEOM_XMAT] = {s->derived + ((size_t)9 * nb + 3 * (size_t)m.ngeom) * E, 9 * m.ngeom, false};
  int* ibase = (int*)((char*)s->arena + dbl_bytes);
  d.solver_iter = ibase; d.warning = ibase + E; d.nefc = ibase + 2 * E;
  s->mask_dev = (uint8_t*)(ibase + 3 * E);
  s->fields[BS_F_SOLVER_ITER] = {d.solver_iter, 1, true};
  s->fields[BS_F_WARNING] = {d.warning, 1, true};
  s->fields[BS_F_NEFC] = {d.nefc, 1, true};
  memset(&s->cd, 0, sizeof s->cd);
  memset(&s->kd, 0, sizeof s->kd);
  s->cd.geom_xpos = (double*)s->fields[BS_F_GEOM_XPOS].ptr;
  s->cd.geom_xmat = (double*)s->fields[BS_F_GEOM_XMAT].ptr;
  if (contacts) {
    // candidate / contact lists and the narrowphase work list; the serial solver's Jacobian rows and
    // dense Hessian (43 KB per environment) are allocated by ensure_serial_contact() only when the
    // serial dynamics kernel is going to run
    const size_t mc = (size_t)s->cm.maxcand, mk = (size_t)s->cm.maxcon;
    const size_t drows = mc * 8 + mk * 13;  // candidates: 7 double rows + pair | hit as two int rows in one more: 64 bytes per slot,
                                            // contiguous, so that the same memory can hold CandRec[E][maxcand] (ColData::crec)
    const size_t irows = 1 + 1 + mk + 1 + mk;
    const size_t nstat = BS_STAT_CON + BS_NSUB;                        // counters (8 bytes each)
    const size_t nitems = mc * ((size_t)E + 4 * 2 * BS_NSUB);         // narrowphase work list, 4 ranges at most
    const size_t bytes = drows * E * sizeof(double) + irows * E * sizeof(int) + 16 + nstat * 8 + 16 + nitems * sizeof(int4);
    he = hipMalloc(&s->carena, bytes);
    if (he != hipSuccess) {
    (void)hipGetLastError();  // a failed allocation leaves its error behind: the next call must not trip over it
      s->carena = nullptr;
      bs_sim_destroy(s);
      return fail(BS_ERR_NOMEM, "bs_sim_create: contact arena (candidate + contact lists + narrowphase work list, %zu bytes per environment): hipMalloc of %zu bytes for %lld environments failed: %s",
                  bytes / (size_t)E, bytes, (long long)n_envs, hipGetErrorString(he));
    }
    s->carena_bytes = bytes;
    HIPCHK_S(hipMemset(s->carena, 0, bytes));
    double* dp = (double*)s->carena;
    auto take = [&](size_t rows_) { double* q = dp; dp += rows_ * (size_t)E; return q; };
    ColData& c = s->cd;
    ConData& k = s->kd;
    c.cand_dist = take(mc); c.cand_pos = take(3 * mc); c.cand_nrm = take(3 * mc);
    c.cand_pair = (int*)take(mc); c.cand_hit = c.cand_pair + mc * (size_t)E;
    c.con_dist = take(mk); c.con_pos = take(3 * mk); c.con_frame = take(9 * mk);
    int* ip = (int*)dp;
    auto takei = [&](size_t rows_) { int* q = ip; ip += rows_ * (size_t)E; return q; };
    c.ncand = takei(1);
    c.ncon = takei(1); c.con_pair = takei(mk);
    k.ncrow = takei(1); k.con_efc = takei(mk);
    c.stat = (unsigned long long*)(((uintptr_t)ip + 7) & ~(uintptr_t)7);
    c.items = (int4*)(((uintptr_t)(c.stat + nstat) + 15) & ~(uintptr_t)15);
    s->fields[BS_F_NCON] = {c.ncon, 1, true};
    if (!s->use_tpc) {
      const int rc2 = ensure_serial_contact(s);
      if (rc2) { bs_sim_destroy(s); return rc2; }
    }
    if (s->use_tpc && !s->narrow_ws && !s->narrow_blocks && !s->cm.has_prim) {
      const size_t gbytes = (size_t)E * s->cm.ngeom * 12 * sizeof(double);
      he = hipMalloc(&s->gframe, gbytes);
      if (he != hipSuccess) {
    (void)hipGetLastError();  // a failed allocation leaves its error behind: the next call must not trip over it
        s->gframe = nullptr;
        bs_sim_destroy(s);
        return fail(BS_ERR_NOMEM, "bs_sim_create: geom-frame records (%zu bytes per environment): hipMalloc of %zu bytes for %lld environments failed: %s",
                    gbytes / (size_t)E, gbytes, (long long)n_envs, hipGetErrorString(he));
      }
      s->gframe_bytes = gbytes;
      HIPCHK_S(hipMemset(s->gframe, 0, gbytes));
      c.gframe = (double*)s->gframe;
      // candidates as 64-byte records in the memory of the five [row][E] candidate arrays (the same size: 7 doubles + 2 ints per slot)
      static_assert(sizeof(CandRec) == 64, "CandRec is one cache line");
      c.crec = (CandRec*)c.cand_dist;
    }
  } else {
    s->fields[BS_F_NCON] = {d.nefc, 0, true};  // zero rows: no contacts in this sim
  }
  for (auto& ev : s->ev) ev = nullptr;
  s->events = true;
  for (auto& ev : s->ev) HIPCHK_S(hipEventCreate(&ev));
  HIPCHK_S(hipEventCreateWithFlags(&s->mask_ev, hipEventDisableTiming));
  HIPCHK_S(hipHostMalloc((void**)&s->mask_pinned, (size_t)E, hipHostMallocDefault));
  {
    // ranges of a contact batch run on separate streams; every range is a whole number of 64-env
    // blocks and non-empty (the count is lowered until that holds)
    const char* sp = getenv("BS_SPLIT");
    int np = sp ? atoi(sp) : (n_envs >= (tl_part_of_mixed ? 8192 : 1024) ? 3 : 1);  // measured at the end of round 2: 3 ranges win from 1024 envs (4.12 vs 3.95 M env-steps/s; 14.0 vs 13.0 M at 4096; 24.6 vs 19.2 M at 8192; 34.2 vs 33.7 M for 2 ranges at 65 536), 4 lose
    np = np < 1 ? 1 : (np > 4 ? 4 : np);
    if (!s->use_tpc) np = 1;
    while (np > 1 && (n_envs < 128L * np || split_chunk(n_envs, np) * (np - 1) >= n_envs)) np--;
    s->split = np;
    if (np >= 3 && !getenv("BS_NL_GRID")) s->nl_grid = 1408;  // three narrowphase launches share the machine (round 3, after the support tables: 768 -> 47.6, 1024 -> 48.0, 1280 -> 48.8, 1536 -> 48.7, 2048 -> 48.4 M env-steps/s at 65 536 envs)
    if (s->split > 1) {
      HIPCHK_S(hipEventCreateWithFlags(&s->fork_ev, hipEventDisableTiming));
      for (int q = 0; q < s->split - 1; q++) {
        HIPCHK_S(hipStreamCreateWithFlags(&s->aux_stream[q], hipStreamNonBlocking));
        HIPCHK_S(hipEventCreateWithFlags(&s->join_ev[q], hipEventDisableTiming));
      }
    }
  }
  bs_launch_reset(grid_for(s), s->stream, s->dm, s->dd, (const uint8_t*)nullptr);
  HIPCHK_S(hipGetLastError());
  HIPCHK_S(hipStreamSynchronize(s->stream));
#undef HIPCHK_S
  *out = s;
  return BS_OK;
}

// which kernel family bs_step runs for this sim, and why if it is the slow generic one
extern "C" int bs_sim_kernel_path(const bs_sim* s, char* buf, int cap) {
  if (!s || !buf || cap < 1) return fail(BS_ERR_INVALID, "bs_sim_kernel_path: null argument");
  if (!s->parts.empty()) {
    // mixed-model sim: the top-level handle has no model of its own; fast only if every part is, the
    // description lists the parts ("part k: ...")
    std::string all;
    int fast = 1;
    for (size_t k = 0; k < s->parts.size(); k++) {
      char one[256];
      const int rc = bs_sim_kernel_path(s->parts[k], one, (int)sizeof one);
      if (rc < 0) return rc;
      if (rc == 0) fast = 0;
      all += (k ? "; part " : "part ") + std::to_string(k) + ": " + one;
    }
    snprintf(buf, (size_t)cap, "%s", all.c_str());
    return fast;
  }
  const bool tp = s->contacts ? s->use_tpc : s->use_tp;
  const char* kv = getenv("BS_KERNEL");
  std::string why;
  if (!tp) {
    if (kv && strcmp(kv, "serial") == 0) why = "BS_KERNEL=serial";
    else if (!s->model->tpv[1].tp_ok) why = s->model->tpv[1].tp_why;  // the wide build's reason: it admits everything the 32-lane one does
    else if (s->contacts && s->cm.maxcon > 64) why = "more than 64 contacts per environment requested";
    else why = "not selected";
  }
  std::string fastname;
  {
    const char* sol = s->dm.solver == 0 ? "_pgs" : s->dm.solver == 1 ? "_cg" : "";
    const bool chain = s->tp.chain && s->dm.solver == 2;
    const char* w = s->wide ? "_w" : "";  // the wide build: 64 lanes per environment
    if (s->contacts)
      fastname = std::string("tree-parallel (k_tp_col") + w + ", k_c_narrow_list, k_tpc" + (chain ? "_chain" : sol) + w + ")";
    else
      fastname = std::string("tree-parallel (k_tp") + (chain && s->dm.integrator == 0 ? (s->tp.quad_top >= 0 ? "_chain4" : "_chain") : (std::string(s->dm.integrator == 1 ? "_rk4" : "") + sol).c_str()) + w + ")";
  }
  snprintf(buf, (size_t)cap, "%s%s", tp ? fastname.c_str()
                                       : (s->contacts ? "serial (k_c_pos, k_c_broad, k_c_narrow_flat, k_c_dyn): " : "serial (k_step): "),
           why.c_str());
  return tp ? 1 : 0;
}

// device memory held by the sim, per arena (a mixed-model sim sums its parts)
extern "C" int bs_sim_footprint(const bs_sim* s, int64_t out[5]) {
  if (!s || !out) return fail(BS_ERR_INVALID, "bs_sim_footprint: null argument");
  for (int q = 0; q < 5; q++) out[q] = 0;
  if (!s->parts.empty()) {
    for (const bs_sim* c : s->parts) {
      int64_t t[5];
      const int rc = bs_sim_footprint(c, t);
      if (rc) return rc;
      for (int q = 0; q < 5; q++) out[q] += t[q];
    }
    return BS_OK;
  }
  out[0] = (int64_t)s->arena_bytes; out[1] = (int64_t)s->carena_bytes; out[2] = (int64_t)s->gframe_bytes;
  out[3] = (int64_t)s->sarena_bytes; out[4] = (int64_t)s->E;
  return BS_OK;
}

extern "C" int bs_sim_destroy(bs_sim* s) {
  if (!s) return BS_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (bs_sim* c : s->parts) {
    c->stream = c->own_stream;
    bs_sim_destroy(c);
  }
  if (s->events)
    for (auto& ev : s->ev)
      if (ev) (void)hipEventDestroy(ev);
  for (int q = 0; q < 3; q++) {
    if (s->aux_stream[q]) {
      (void)hipStreamSynchronize(s->aux_stream[q]);
      (void)hipStreamDestroy(s->aux_stream[q]);
    }
    if (s->join_ev[q]) (void)hipEventDestroy(s->join_ev[q]);
  }
  if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
  if (s->mask_ev) (void)hipEventDestroy(s->mask_ev);
  if (s->mask_pinned) (void)hipHostFree(s->mask_pinned);
  if (s->sarena) (void)hipFree(s->sarena);
  if (s->gframe) (void)hipFree(s->gframe);
  if (s->arena) (void)hipFree(s->arena);
  if (s->carena) (void)hipFree(s->carena);
  if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
  delete s;
  return BS_OK;
}

extern "C" int bs_set_stream(bs_sim* s, void* hip_stream, int use_own_stream) {
  if (!s) return fail(BS_ERR_INVALID, "bs_set_stream: null sim");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->stream = use_own_stream ? s->own_stream : (hipStream_t)hip_stream;
  for (bs_sim* c : s->parts) c->stream = s->stream;
  return BS_OK;
}

extern "C" int bs_sim_num_parts(const bs_sim* s) { return s ? (int)s->parts.size() : 0; }

extern "C" int bs_sim_part(bs_sim* s, int part, bs_sim** out, int64_t* env_start, int64_t* n_envs) {
  if (!s || !out) return fail(BS_ERR_INVALID, "bs_sim_part: null argument");
  if (part < 0 || part >= (int)s->parts.size())
    return fail(BS_ERR_INVALID, "bs_sim_part: part %d out of range [0,%d)", part, (int)s->parts.size());
  *out = s->parts[part];
  if (env_start) *env_start = s->part_start[part];
  if (n_envs) *n_envs = s->parts[part]->n_envs;
  return BS_OK;
}

extern "C" int bs_step(bs_sim* s, int n_steps) {
  if (!s) return fail(BS_ERR_INVALID, "bs_step: null sim");
  if (n_steps < 0) return fail(BS_ERR_INVALID, "bs_step: n_steps must be >= 0 (got %d)", n_steps);
  if (n_steps == 0) return BS_OK;
  HIPCHK(hipSetDevice(s->device));
  if (!s->parts.empty()) {
    if (s->split > 1) {
      HIPCHK(hipEventRecord(s->fork_ev, s->stream));
      for (size_t k = 0; k < s->parts.size(); k++) {
        bs_sim* c = s->parts[k];
        // contact steps are sequences of short launches with gaps between them; a contact-free part is one long
        // launch that fills the machine by itself (measured: PGS 29.0 M serial, 24.7 M concurrent)
        const bool aside = (k == 1 || k == 2) && c->contacts;
        hipStream_t st = aside ? c->own_stream : s->stream;
        if (aside) HIPCHK(hipStreamWaitEvent(st, s->fork_ev, 0));
        c->stream = st;
        launch_step(c, n_steps);
        c->stream = s->stream;
        if (aside) HIPCHK(hipEventRecord(s->join_ev[k - 1], st));
      }
      for (size_t k = 1; k < s->parts.size() && k <= 2; k++)
        if (s->parts[k]->contacts) HIPCHK(hipStreamWaitEvent(s->stream, s->join_ev[k - 1], 0));
    } else {
      for (bs_sim* c : s->parts) launch_step(c, n_steps);
    }
    HIPCHK(hipGetLastError());
    return BS_OK;
  }
  launch_step(s, n_steps);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

extern "C" int bs_forward(bs_sim* s) {
  if (!s) return fail(BS_ERR_INVALID, "bs_forward: null sim");
  HIPCHK(hipSetDevice(s->device));
  if (!s->parts.empty()) {
    for (bs_sim* c : s->parts) {
      const int rc = bs_forward(c);
      if (rc) return rc;
    }
    return BS_OK;
  }
  if (s->contacts && s->use_tpc) {
    // every intermediate written by the tree-parallel kernels too (k_tpc_full / _cg, round 2; _pgs, round 3): no
    // serial pipeline, none of its 43 KB per environment of solver buffers
    launch_tpc_pass(s, s->dd, s->stream, 1, 2, true);
  } else if (s->contacts) {
    const int rc = ensure_serial_contact(s);  // ineligible models: serial pipeline
    if (rc) return rc;
    launch_contact_pass(s, 0, 0);
  } else if (s->use_tp) {
    auto* launch = s->dm.solver == 0 ? bs_launch_tp_full_pgs : s->dm.solver == 1 ? bs_launch_tp_full_cg : bs_launch_tp_full;
    if (s->wide) launch = s->dm.solver == 0 ? bs_launch_tp_full_pgs_w : s->dm.solver == 1 ? bs_launch_tp_full_cg_w : bs_launch_tp_full_w;
    launch(grid_tp(s), s->dm.solver == 0 ? (size_t)(64 / s->lpe) * s->dm.nbody * tp_minv_stride(s->dm.nbody) * sizeof(double) : 0, s->stream,
           s->dm, s->tp, s->dd, 1, 1, 2);
  } else
    bs_launch_forward(grid_for(s), s->stream, s->dm, s->dd);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

extern "C" int bs_reset(bs_sim* s, const uint8_t* env_mask) {
  if (!s) return fail(BS_ERR_INVALID, "bs_reset: null sim");
  HIPCHK(hipSetDevice(s->device));
  if (!s->parts.empty()) {
    for (size_t k = 0; k < s->parts.size(); k++) {
      const int rc = bs_reset(s->parts[k], env_mask ? env_mask + s->part_start[k] : nullptr);
      if (rc) return rc;
    }
    return BS_OK;
  }
  const uint8_t* dmask = nullptr;
  if (env_mask) {
    // staged through the sim's pinned buffer: nothing is allocated and the host does not wait for
    // the device (only for the previous upload, if it has not left the staging buffer yet)
    if (s->mask_pending) HIPCHK(hipEventSynchronize(s->mask_ev));
    memcpy(s->mask_pinned, env_mask, (size_t)s->n_envs);
    HIPCHK(hipMemcpyAsync(s->mask_dev, s->mask_pinned, (size_t)s->n_envs, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipEventRecord(s->mask_ev, s->stream));
    s->mask_pending = true;
    dmask = s->mask_dev;
  }
  bs_launch_reset(grid_for(s), s->stream, s->dm, s->dd, dmask);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

// mj_resetData for the environments whose byte in `dev_mask` (device memory, n_envs bytes of the
// sim's GPU) is non-zero: one kernel on the sim's stream, no host involvement.
extern "C" int bs_reset_device_mask(bs_sim* s, const uint8_t* dev_mask) {
  if (!s || !dev_mask) return fail(BS_ERR_INVALID, "bs_reset_device_mask: null argument");
  HIPCHK(hipSetDevice(s->device));
  if (!s->parts.empty()) {
    for (size_t k = 0; k < s->parts.size(); k++) {
      const int rc = bs_reset_device_mask(s->parts[k], dev_mask + s->part_start[k]);
      if (rc) return rc;
    }
    return BS_OK;
  }
  bs_launch_reset(grid_for(s), s->stream, s->dm, s->dd, dev_mask);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

// ---- allocation-free masked reset (include/batchsim.h: bs_reset_noalloc).  The device copy of the
// mask lives in a file-scope __device__ array instead of memory the library allocates; each GPU holds
// its own copy of the array at its own address, looked up once per device.
static __device__ __align__(256) uint8_t scratch[BS_NOALLOC_MAX_ENVS];
static uint8_t* g_scratch_ptr[BS_NOALLOC_MAX_DEVICES];  // device address of `scratch` per device, 0 = not looked up yet
// one event per device, created together with the address lookup: recorded after the reset kernels that READ the
// array, waited for by the next call's copy INTO it — so two sims on different streams of one GPU, called back
// to back from one host thread, no longer overwrite a mask the first sim's kernel has not read yet
static hipEvent_t g_scratch_ev[BS_NOALLOC_MAX_DEVICES];
static bool g_scratch_ev_recorded[BS_NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mu;
static int scratch_on_device(int device, uint8_t** out) {
  if (device < 0 || device >= BS_NOALLOC_MAX_DEVICES)
    return fail(BS_ERR_UNSUPPORTED, "bs_reset_noalloc: device %d beyond the %d devices it keeps a scratch address for", device,
                BS_NOALLOC_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    HIPCHK(hipSetDevice(device));  // the symbol resolves to the current device's copy
    uint8_t* ptr = nullptr;
    HIPCHK(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(BS_ERR_HIP, "bs_reset_noalloc: hipGetSymbolAddress returned a null address");
    HIPCHK(hipEventCreateWithFlags(&g_scratch_ev[device], hipEventDisableTiming));
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return BS_OK;
}

extern "C" int bs_reset_noalloc(bs_sim* s, const uint8_t* env_mask) {
  if (!s || !env_mask) return fail(BS_ERR_INVALID, "bs_reset_noalloc: null argument");
  if (s->n_envs > BS_NOALLOC_MAX_ENVS)
    return fail(BS_ERR_UNSUPPORTED, "bs_reset_noalloc: %ld environments exceed the static scratch of %d (use bs_reset)",
                s->n_envs, BS_NOALLOC_MAX_ENVS);
  uint8_t* ptr = nullptr;
  const int rc = scratch_on_device(s->device, &ptr);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  // pinned (or registered) host memory: the copy is asynchronous; pageable memory: wait until the
  // copy has left the caller's buffer before returning
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, env_mask) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();
  // stream capture: the graph's own edges order the captured nodes, and an event recorded outside the capture may
  // not be waited for inside it — the cross-call event is skipped there (replays that overlap are the caller's to serialise)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing && !pinned)
    return fail(BS_ERR_INVALID, "bs_reset_noalloc: under stream capture the mask must be pinned host memory (a pageable one needs a stream synchronisation)");
  std::lock_guard<std::mutex> lock(g_scratch_mu);  // wait + copy + launch + record of one call are not interleaved with another host thread's
  if (!capturing && g_scratch_ev_recorded[s->device]) HIPCHK(hipStreamWaitEvent(s->stream, g_scratch_ev[s->device], 0));
  // every byte the kernels read is overwritten here, so nothing of the previous call's mask survives
  HIPCHK(hipMemcpyAsync(ptr, env_mask, (size_t)s->n_envs, hipMemcpyHostToDevice, s->stream));
  if (!pinned) HIPCHK(hipStreamSynchronize(s->stream));
  if (!s->parts.empty()) {
    for (size_t k = 0; k < s->parts.size(); k++) {
      bs_sim* c = s->parts[k];
      bs_launch_reset(grid_for(c), s->stream, c->dm, c->dd, ptr + s->part_start[k]);
    }
  } else {
    bs_launch_reset(grid_for(s), s->stream, s->dm, s->dd, ptr);
  }
  HIPCHK(hipGetLastError());
  if (!capturing) {
    HIPCHK(hipEventRecord(g_scratch_ev[s->device], s->stream));
    g_scratch_ev_recorded[s->device] = true;
  }
  return BS_OK;
}

static int refresh_derived(bs_sim* s) {
  bs_launch_frames(grid_for(s), s->stream, s->dm, s->dd,
                     (double*)s->fields[BS_F_XMAT].ptr, (double*)s->fields[BS_F_GEOM_XPOS].ptr,
                     (double*)s->fields[BS_F_GEOM_XMAT].ptr);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

extern "C" int bs_get_ptr(bs_sim* s, int field_id, void** dev_ptr, int64_t shape[2], int64_t stride[2],
                          int* is_int32) {
  if (!s || !dev_ptr || !shape || !stride) return fail(BS_ERR_INVALID, "bs_get_ptr: null argument");
  if (field_id < 0 || field_id >= BS_F_COUNT) return fail(BS_ERR_INVALID, "bs_get_ptr: unknown field %d", field_id);
  NO_PARTS(s, "bs_get_ptr");
  HIPCHK(hipSetDevice(s->device));
  if (field_id == BS_F_XMAT || field_id == BS_F_GEOM_XPOS || field_id == BS_F_GEOM_XMAT) {
    int rc = refresh_derived(s);
    if (rc) return rc;
  }
  const FieldInfo& f = s->fields[field_id];
  *dev_ptr = f.ptr;
  shape[0] = f.rows; shape[1] = s->n_envs;
  stride[0] = s->E; stride[1] = 1;
  if (is_int32) *is_int32 = f.is_int ? 1 : 0;
  return BS_OK;
}

extern "C" int bs_copy_field(bs_sim* s, int field_id, void* host, int direction) {
  if (!s || !host) return fail(BS_ERR_INVALID, "bs_copy_field: null argument");
  if (field_id < 0 || field_id >= BS_F_COUNT) return fail(BS_ERR_INVALID, "bs_copy_field: unknown field %d", field_id);
  if (direction != 0 && direction != 1) return fail(BS_ERR_INVALID, "bs_copy_field: direction must be 0 or 1");
  NO_PARTS(s, "bs_copy_field");
  HIPCHK(hipSetDevice(s->device));
  const bool derived = field_id == BS_F_XMAT || field_id == BS_F_GEOM_XPOS || field_id == BS_F_GEOM_XMAT;
  if (derived) {
    if (direction == 1) return fail(BS_ERR_INVALID, "bs_copy_field: field %d is read-only", field_id);
    int rc = refresh_derived(s);
    if (rc) return rc;
  }
  const FieldInfo& f = s->fields[field_id];
  if (f.rows == 0) return BS_OK;
  const size_t esz = f.is_int ? sizeof(int) : sizeof(double);
  HIPCHK(hipStreamSynchronize(s->stream));
  if (direction == 0)
    HIPCHK(hipMemcpy2D(host, (size_t)s->n_envs * esz, f.ptr, (size_t)s->E * esz, (size_t)s->n_envs * esz,
                       (size_t)f.rows, hipMemcpyDeviceToHost));
  else
    HIPCHK(hipMemcpy2D(f.ptr, (size_t)s->E * esz, host, (size_t)s->n_envs * esz, (size_t)s->n_envs * esz,
                       (size_t)f.rows, hipMemcpyHostToDevice));
  return BS_OK;
}

extern "C" int bs_get_contacts(bs_sim* s, int64_t env, int32_t* geom1, int32_t* geom2, double* dist, int cap,
                               int* n) {
  if (!s || !n) return fail(BS_ERR_INVALID, "bs_get_contacts: null argument");
  if (env < 0 || env >= s->n_envs) return fail(BS_ERR_INVALID, "bs_get_contacts: env %lld out of range", (long long)env);
  for (size_t k = 0; k < s->parts.size(); k++)
    if (env >= s->part_start[k] && env < s->part_start[k] + s->parts[k]->n_envs)
      return bs_get_contacts(s->parts[k], env - s->part_start[k], geom1, geom2, dist, cap, n);
  *n = 0;
  if (!s->contacts) return BS_OK;
  if (cap < 0 || (cap > 0 && (!geom1 || !geom2 || !dist))) return fail(BS_ERR_INVALID, "bs_get_contacts: null output");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  int nc = 0;
  HIPCHK(hipMemcpy(&nc, s->cd.ncon + env, sizeof(int), hipMemcpyDeviceToHost));
  if (nc > s->cm.maxcon) nc = s->cm.maxcon;
  *n = nc;
  const int m = nc < cap ? nc : cap;
  if (m <= 0) return BS_OK;
  std::vector<int> pr(m);
  HIPCHK(hipMemcpy2D(pr.data(), sizeof(int), s->cd.con_pair + env, (size_t)s->E * sizeof(int), sizeof(int), (size_t)m,
                     hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(dist, sizeof(double), s->cd.con_dist + env, (size_t)s->E * sizeof(double), sizeof(double),
                     (size_t)m, hipMemcpyDeviceToHost));
  for (int i = 0; i < m; i++) {
    geom1[i] = s->model->d.pair_geom1[pr[i]];
    geom2[i] = s->model->d.pair_geom2[pr[i]];
  }
  return BS_OK;
}

// sim.data.contact[i].pos / frame of one environment (rows of mjContact beyond bs_get_contacts)
extern "C" int bs_get_contact_geometry(bs_sim* s, int64_t env, double* pos, double* frame, int cap, int* n) {
  if (!s || !n) return fail(BS_ERR_INVALID, "bs_get_contact_geometry: null argument");
  if (env < 0 || env >= s->n_envs) return fail(BS_ERR_INVALID, "bs_get_contact_geometry: env %lld out of range", (long long)env);
  for (size_t k = 0; k < s->parts.size(); k++)
    if (env >= s->part_start[k] && env < s->part_start[k] + s->parts[k]->n_envs)
      return bs_get_contact_geometry(s->parts[k], env - s->part_start[k], pos, frame, cap, n);
  *n = 0;
  if (!s->contacts) return BS_OK;
  if (cap < 0 || (cap > 0 && (!pos || !frame))) return fail(BS_ERR_INVALID, "bs_get_contact_geometry: null output");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  int nc = 0;
  HIPCHK(hipMemcpy(&nc, s->cd.ncon + env, sizeof(int), hipMemcpyDeviceToHost));
  if (nc > s->cm.maxcon) nc = s->cm.maxcon;
  *n = nc;
  const int m = nc < cap ? nc : cap;
  if (m <= 0) return BS_OK;
  HIPCHK(hipMemcpy2D(pos, sizeof(double), s->cd.con_pos + env, (size_t)s->E * sizeof(double), sizeof(double),
                     (size_t)3 * m, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy2D(frame, sizeof(double), s->cd.con_frame + env, (size_t)s->E * sizeof(double), sizeof(double),
                     (size_t)9 * m, hipMemcpyDeviceToHost));
  return BS_OK;
}

// ---- MjSimState: time | qpos | qvel | qacc | qacc_warmstart, everything a later step depends on
// besides the inputs (ctrl, qfrc_applied), as one [rows][ld] device buffer of the caller
static const int kStateFields[5] = {BS_F_TIME, BS_F_QPOS, BS_F_QVEL, BS_F_QACC, BS_F_QACC_WARMSTART};
extern "C" int bs_state_rows(const bs_sim* s, int* rows) {
  if (!s || !rows) return fail(BS_ERR_INVALID, "bs_state_rows: null argument");
  NO_PARTS(s, "bs_state_rows");
  int r = 0;
  for (int f : kStateFields) r += s->fields[f].rows;
  *rows = r;
  return BS_OK;
}
static int state_copy(bs_sim* s, double* buf, int64_t ld, bool get) {
  if (!s || !buf) return fail(BS_ERR_INVALID, "bs_get_state / bs_set_state: null argument");
  NO_PARTS(s, "bs_get_state / bs_set_state");
  if (ld < s->n_envs) return fail(BS_ERR_INVALID, "bs_get_state / bs_set_state: ld %lld < n_envs %ld", (long long)ld, s->n_envs);
  HIPCHK(hipSetDevice(s->device));
  size_t row = 0;
  for (int f : kStateFields) {
    const FieldInfo& fi = s->fields[f];
    double* b = buf + row * (size_t)ld;
    if (get)
      HIPCHK(hipMemcpy2DAsync(b, (size_t)ld * 8, fi.ptr, (size_t)s->E * 8, (size_t)s->n_envs * 8, (size_t)fi.rows,
                              hipMemcpyDeviceToDevice, s->stream));
    else
      HIPCHK(hipMemcpy2DAsync(fi.ptr, (size_t)s->E * 8, b, (size_t)ld * 8, (size_t)s->n_envs * 8, (size_t)fi.rows,
                              hipMemcpyDeviceToDevice, s->stream));
    row += (size_t)fi.rows;
  }
  return BS_OK;
}
extern "C" int bs_get_state(bs_sim* s, double* dev_buf, int64_t ld) { return state_copy(s, dev_buf, ld, true); }
extern "C" int bs_set_state(bs_sim* s, const double* dev_buf, int64_t ld) {
  return state_copy(s, const_cast<double*>(dev_buf), ld, false);
}

extern "C" int bs_get_stream(bs_sim* s, void** hip_stream) {
  if (!s || !hip_stream) return fail(BS_ERR_INVALID, "bs_get_stream: null argument");
  *hip_stream = (void*)s->stream;
  return BS_OK;
}

extern "C" int bs_collision_stats(bs_sim* s, uint64_t out[3], int reset) {
  if (!s || !out) return fail(BS_ERR_INVALID, "bs_collision_stats: null argument");
  out[0] = out[1] = out[2] = 0;
  for (bs_sim* c : s->parts) {
    uint64_t t[3];
    const int rc = bs_collision_stats(c, t, reset);
    if (rc) return rc;
    for (int q = 0; q < 3; q++) out[q] += t[q];
  }
  if (!s->contacts) return BS_OK;
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  HIPCHK(hipMemcpy(out, s->cd.stat, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  {
    uint64_t part[BS_NSUB];  // the tree-parallel kernels count contacts on BS_NSUB addresses
    HIPCHK(hipMemcpy(part, s->cd.stat + BS_STAT_CON, sizeof part, hipMemcpyDeviceToHost));
    for (int q = 0; q < BS_NSUB; q++) out[2] += part[q];
    if (reset) HIPCHK(hipMemset(s->cd.stat + BS_STAT_CON, 0, sizeof part));
  }
#ifdef NP_PROFILE
  bs_np_profile_dump();
#endif
#ifdef NP_TIMING
  {
    uint64_t t[24];
    HIPCHK(hipMemcpy(t, s->cd.stat + 8, sizeof t, hipMemcpyDeviceToHost));
    fprintf(stderr, "np_timing");
    for (int q = 0; q < 24; q++) fprintf(stderr, " %llu", (unsigned long long)t[q]);
    fprintf(stderr, "\n");
    HIPCHK(hipMemset(s->cd.stat + 8, 0, sizeof t));
  }
#endif
#ifdef TPC_PROFILE
  {
    uint64_t t[24];
    HIPCHK(hipMemcpy(t, s->cd.stat + 8, sizeof t, hipMemcpyDeviceToHost));
    fprintf(stderr, "k_tpc section ticks:");
    for (int q = 0; q < 24; q++) fprintf(stderr, " [%d] %llu", q, (unsigned long long)t[q]);
    fprintf(stderr, "\n");
    if (reset) HIPCHK(hipMemset(s->cd.stat + 8, 0, sizeof t));
  }
#endif
  if (reset) HIPCHK(hipMemset(s->cd.stat, 0, 3 * sizeof(uint64_t)));
  return BS_OK;
}

extern "C" int bs_sync(bs_sim* s) {
  if (!s) return fail(BS_ERR_INVALID, "bs_sync: null sim");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  return BS_OK;
}

extern "C" int bs_stage_times(bs_sim* s, double* ms, int n) {
  if (!s || !ms || n < 5) return fail(BS_ERR_INVALID, "bs_stage_times: need room for 5 stage times");
  NO_PARTS(s, "bs_stage_times");
  HIPCHK(hipSetDevice(s->device));
  if (s->contacts) {
    if (s->dm.integrator == 1)
      return fail(BS_ERR_UNSUPPORTED, "bs_stage_times: the staged step is an Euler step; not available for RK4 with contacts");
    const int rc = ensure_serial_contact(s);
    if (rc) return rc;
    const dim3 g = dim3((unsigned)((s->n_envs + BS_BLOCK - 1) / BS_BLOCK));
    HIPCHK(hipEventRecord(s->ev[0], s->stream));
    bs_launch_c_pos(g, s->stream, s->dm, s->dd, s->cd, 1);
    HIPCHK(hipEventRecord(s->ev[1], s->stream));
    bs_launch_c_broad(dim3((unsigned)s->n_envs), s->stream, s->cm, s->dd, s->cd);
    launch_narrow(s, s->dd, s->stream);
    HIPCHK(hipEventRecord(s->ev[2], s->stream));
    bs_launch_c_dyn(g, s->hs_bytes, s->stream, s->dm, s->cm, s->dd, s->cd, s->kd, 1,
                       s->hs_bytes ? 1 : 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[3], s->stream));
    HIPCHK(hipEventSynchronize(s->ev[3]));
    float t[3];
    for (int k = 0; k < 3; k++) HIPCHK(hipEventElapsedTime(&t[k], s->ev[k], s->ev[k + 1]));
    ms[0] = t[0]; ms[1] = t[1]; ms[2] = 0.0; ms[3] = t[2]; ms[4] = 0.0;  // velocity+constraint+integrate fused in [3]
    return BS_OK;
  }
  const int stages[4] = {0, 2, 3, 4};
  HIPCHK(hipEventRecord(s->ev[0], s->stream));
  for (int k = 0; k < 4; k++) {
    bs_launch_stage(grid_for(s), s->stream, s->dm, s->dd, stages[k]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev[k + 1], s->stream));
  }
  HIPCHK(hipEventSynchronize(s->ev[4]));
  float t[4];
  for (int k = 0; k < 4; k++) HIPCHK(hipEventElapsedTime(&t[k], s->ev[k], s->ev[k + 1]));
  ms[0] = t[0]; ms[1] = 0.0; ms[2] = t[1]; ms[3] = t[2]; ms[4] = t[3];
  return BS_OK;
}


extern "C" int bs_pack_obs(bs_sim* s, float* out_dev, int64_t row_stride) {
  if (!s || !out_dev) return fail(BS_ERR_INVALID, "bs_pack_obs: null argument");
  NO_PARTS(s, "bs_pack_obs");
  const int nq = s->fields[BS_F_QPOS].rows;
  if (row_stride < nq + s->dm.nv + 1)
    return fail(BS_ERR_INVALID, "bs_pack_obs: row_stride %lld < nq + nv + 1 = %d", (long long)row_stride,
                nq + s->dm.nv + 1);
  HIPCHK(hipSetDevice(s->device));
  bs_launch_pack_obs(dim3((unsigned)((s->n_envs + 255) / 256)), s->stream, s->dm, s->dd, out_dev,
                     (long)row_stride, nq);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

extern "C" int bs_env_pre(bs_sim* s, const double* action_dev, int64_t ld) {
  if (!s || !action_dev) return fail(BS_ERR_INVALID, "bs_env_pre: null argument");
  NO_PARTS(s, "bs_env_pre");
  if (ld < s->dm.nu) return fail(BS_ERR_INVALID, "bs_env_pre: ld %lld < nu = %d", (long long)ld, s->dm.nu);
  HIPCHK(hipSetDevice(s->device));
  bs_launch_env_pre(dim3((unsigned)((s->n_envs + 255) / 256)), s->stream, s->dm, s->dd, action_dev, (long)ld);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

extern "C" int bs_env_post(bs_sim* s, const double* action_dev, int64_t ld, int32_t* t_dev, int episode_steps, float* obs_dev,
                           float* term_obs_dev, int64_t row_stride, float* reward_dev, uint8_t* flags_dev) {
  if (!s || !action_dev || !t_dev || !obs_dev || !term_obs_dev || !reward_dev || !flags_dev)
    return fail(BS_ERR_INVALID, "bs_env_post: null argument");
  NO_PARTS(s, "bs_env_post");
  const int nq = s->fields[BS_F_QPOS].rows;
  if (row_stride < nq + s->dm.nv)
    return fail(BS_ERR_INVALID, "bs_env_post: row_stride %lld < nq + nv = %d", (long long)row_stride, nq + s->dm.nv);
  if (episode_steps < 1) return fail(BS_ERR_INVALID, "bs_env_post: episode_steps must be >= 1");
  if (ld < s->dm.nu) return fail(BS_ERR_INVALID, "bs_env_post: ld %lld < nu = %d", (long long)ld, s->dm.nu);
  HIPCHK(hipSetDevice(s->device));
  bs_launch_env_post(dim3((unsigned)((s->n_envs + 255) / 256)), s->stream, s->dm, s->dd, action_dev, (long)ld, t_dev,
                     episode_steps, obs_dev, term_obs_dev, (long)row_stride, reward_dev, flags_dev, nq);
  HIPCHK(hipGetLastError());
  return BS_OK;
}

extern "C" int bs_timed_steps(bs_sim* s, int n_steps, double* elapsed_ms) {
  if (!s || !elapsed_ms) return fail(BS_ERR_INVALID, "bs_timed_steps: null argument");
  if (n_steps < 1) return fail(BS_ERR_INVALID, "bs_timed_steps: n_steps must be >= 1");
  NO_PARTS(s, "bs_timed_steps");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipEventRecord(s->ev[5], s->stream));
  for (int k = 0; k < n_steps; k++) launch_step(s, 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(s->ev[6], s->stream));
  HIPCHK(hipEventSynchronize(s->ev[6]));
  float t = 0;
  HIPCHK(hipEventElapsedTime(&t, s->ev[5], s->ev[6]));
  *elapsed_ms = t;
  return BS_OK;
}
