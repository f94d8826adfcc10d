// k_tp_kernels.hip — the tree-parallel kernels (physics_tp.h), ONE instantiation of tp_run per
// object file: the Makefile compiles this source once per kernel with -DTPK_<name>, so the
// instantiations (each a few thousand lines of straight-line fp64 code) build in parallel.
// Block = one 64-lane wavefront = two environments; __launch_bounds__(64, 2) = two waves per SIMD.
// the contact kernel of CHAIN models forms the sweeps' receive factors where they are used (physics_tp.h::tp_recv_factor):
// it is short of registers, the contact-free one is not and keeps them hoisted (167 vs 171 M env-steps/s at 4096 envs)
#ifdef TPK_tpc_chain
#define TP_RECV_OPAQUE 1
#endif
#include "kernels.h"
// kernel and launcher names: k_<name> / bs_launch_<name>, with the suffix _w in the wide build (-DTP_LPE=64)
#if TP_LPE == 64
#define KN(name) k_##name##_w
#define LN(name) bs_launch_##name##_w
#else
#define KN(name) k_##name
#define LN(name) bs_launch_##name
#endif
#ifndef TP_OCC
#define TP_OCC 2
#endif

#define TP_KERNEL(name, CONTACTS_, FULL_, RK4_, CG_, PGS_, PARK_, MINV_) TP_KERNEL_C(name, CONTACTS_, FULL_, RK4_, CG_, PGS_, PARK_, MINV_, false, false)
#define TP_KERNEL_C(name, CONTACTS_, FULL_, RK4_, CG_, PGS_, PARK_, MINV_, CHAIN_, QUAD_)                                   \
  __global__ __launch_bounds__(64, TP_OCC) void KN(name)(DevModel m, TpModel t, DevData d, int nsteps, int mode, int dump) { \
    __shared__ int tbl[TPL_TOTAL * 64];                                                                               \
    __shared__ double park[PARK_ ? (TP_MAXD + 1) * 64 : 1];                                                           \
    extern __shared__ double minv[]; /* PGS: 2 x nbody x tp_minv_stride(nbody) */                                     \
    const TpShared sh = {tbl, nullptr, nullptr, nullptr, PARK_ ? park : nullptr, MINV_ ? minv : nullptr};             \
    ColModel cm;                                                                                                      \
    ColData cd;                                                                                                       \
    tp_run<CONTACTS_, FULL_, RK4_, CG_, PGS_, CHAIN_, QUAD_>(m, t, d, nsteps, mode, dump, sh, cm, cd);                \
  }                                                                                                                   \
  void LN(name)(BS_TP_ARGS) { hipLaunchKernelGGL(KN(name), g, dim3(64), lds, st, m, t, d, nsteps, mode, dump); }

// both instantiations in one kernel, chosen by TpModel::chain (uniform): for the variants that are not worth an
// object file of their own per instantiation
#define TP_KERNEL_D(name, CONTACTS_, FULL_, RK4_, CG_, PGS_, PARK_, MINV_)                                           \
  __global__ __launch_bounds__(64, TP_OCC) void KN(name)(DevModel m, TpModel t, DevData d, int nsteps, int mode, int dump) { \
    __shared__ int tbl[TPL_TOTAL * 64];                                                                               \
    __shared__ double park[PARK_ ? (TP_MAXD + 1) * 64 : 1];                                                           \
    extern __shared__ double minv[];                                                                                  \
    const TpShared sh = {tbl, nullptr, nullptr, nullptr, PARK_ ? park : nullptr, MINV_ ? minv : nullptr};             \
    ColModel cm;                                                                                                      \
    ColData cd;                                                                                                       \
    if (TP_LPE != 64 && t.chain) tp_run<CONTACTS_, FULL_, RK4_, CG_, PGS_, TP_LPE != 64>(m, t, d, nsteps, mode, dump, sh, cm, cd); \
    else tp_run<CONTACTS_, FULL_, RK4_, CG_, PGS_, false>(m, t, d, nsteps, mode, dump, sh, cm, cd);                   \
  }                                                                                                                   \
  void LN(name)(BS_TP_ARGS) { hipLaunchKernelGGL(KN(name), g, dim3(64), lds, st, m, t, d, nsteps, mode, dump); }
#define TPC_KERNEL_D(name, CG_, PGS_)                                                                                  \
  __global__ __launch_bounds__(64, TPC_OCC) void KN(name)(DevModel m, TpModel t, DevData d, ColModel cm, ColData cd,  \
                                                          int mode, int dump) {                                      \
    __shared__ int tbl[TPL_TOTAL * 64];                                                                               \
    extern __shared__ double tpc_dyn[];                                                                               \
    const int ntri = m.nbody * (m.nbody + 1) / 2;                                                                     \
    const TpShared sh = {tbl, tpc_dyn, tpc_dyn + TP_ENVS * ntri, tpc_dyn + TP_ENVS * ntri + 64 * TPC_SLOT, nullptr, nullptr}; \
    if (TP_LPE != 64 && t.chain) tp_run<true, false, false, CG_, PGS_, TP_LPE != 64>(m, t, d, 1, mode, dump, sh, cm, cd); \
    else tp_run<true, false, false, CG_, PGS_, false>(m, t, d, 1, mode, dump, sh, cm, cd);                            \
  }                                                                                                                   \
  void LN(name)(BS_TPC_ARGS) { hipLaunchKernelGGL(KN(name), g, dim3(64), lds, st, m, t, d, cm, cd, mode, dump); }

// contacts: candidates from k_tp_col + the narrowphase kernel, one step per launch.  Dynamic LDS:
// packed Hessians for nbody rows (2 envs), then contact slots, then the U staging rows
#define TPC_KERNEL(name, CG_, PGS_) TPC_KERNEL_F(name, false, CG_, PGS_)
#define TPC_KERNEL_F(name, FULL_, CG_, PGS_) TPC_KERNEL_FC(name, FULL_, CG_, PGS_, false, false)
#define TPC_KERNEL_FC(name, FULL_, CG_, PGS_, CHAIN_, QUAD_)                                                                  \
  __global__ __launch_bounds__(64, TPC_OCC) void KN(name)(DevModel m, TpModel t, DevData d, ColModel cm, ColData cd,  \
                                                          int mode, int dump) {                                      \
    __shared__ int tbl[TPL_TOTAL * 64];                                                                               \
    extern __shared__ double tpc_dyn[];                                                                               \
    const int ntri = m.nbody * (m.nbody + 1) / 2;                                                                     \
    const TpShared sh = {tbl, tpc_dyn, tpc_dyn + TP_ENVS * ntri, tpc_dyn + TP_ENVS * ntri + 64 * TPC_SLOT, nullptr, nullptr}; \
    tp_run<true, FULL_, false, CG_, PGS_, CHAIN_, QUAD_>(m, t, d, 1, mode, dump, sh, cm, cd);                         \
  }                                                                                                                   \
  void LN(name)(BS_TPC_ARGS) { hipLaunchKernelGGL(KN(name), g, dim3(64), lds, st, m, t, d, cm, cd, mode, dump); }

#if defined(TPK_tp)
TP_KERNEL(tp, false, false, false, false, false, true, false)
#elif defined(TPK_tp_chain)
// Euler + Newton, contact-free, for models whose dofs below the first branching form single-child chains on
// consecutive lanes (TpModel::chain): the sweeps' packets of those levels move by DPP shifts
TP_KERNEL_C(tp_chain, false, false, false, false, false, true, false, true, false)
#elif defined(TPK_tp_chain4)
// ... whose chains are aligned groups of four lanes off one body (TpModel::quad_top, the exported hands): the chain sums
// by DPP quad permutes as well
TP_KERNEL_C(tp_chain4, false, false, false, false, false, true, false, true, true)
#elif defined(TPK_tp_cg)
TP_KERNEL_D(tp_cg, false, false, false, true, false, true, false)
#elif defined(TPK_tp_pgs)
TP_KERNEL_D(tp_pgs, false, false, false, false, true, false, true)
#elif defined(TPK_tp_rk4)
TP_KERNEL_D(tp_rk4, false, false, true, false, false, false, false)
#elif defined(TPK_tp_rk4_cg)
TP_KERNEL_D(tp_rk4_cg, false, false, true, true, false, false, false)
#elif defined(TPK_tp_rk4_pgs)
TP_KERNEL_D(tp_rk4_pgs, false, false, true, false, true, false, true)
#elif defined(TPK_tp_full)
TP_KERNEL(tp_full, false, true, false, false, false, false, false)
#elif defined(TPK_tp_full_cg)
TP_KERNEL(tp_full_cg, false, true, false, true, false, false, false)
#elif defined(TPK_tp_full_pgs)
TP_KERNEL(tp_full_pgs, false, true, false, false, true, false, true)
#elif defined(TPK_tp_col)
// kinematics + geom frames + broadphase, tree-parallel layout (physics_tp.h::tp_collide_prep)
#ifdef TPC_CUTS  // developer build: the kernel returns at section BS_COLCUT
static int g_colcut = -1;
void bs_set_colcut(int c) { g_colcut = c; }
__global__ __launch_bounds__(64) void KN(tp_col)(DevModel m, TpModel t, DevData d, ColModel cm, ColData cd, int colcut) {
  extern __shared__ double tpg_dyn[];
  tp_collide_prep(m, t, d, cm, cd, tpg_dyn, colcut);
}
void LN(tp_col)(dim3 g, size_t lds, hipStream_t st, const DevModel& m, const TpModel& t, const DevData& d,
                      const ColModel& cm, const ColData& cd) {
  hipLaunchKernelGGL(KN(tp_col), g, dim3(64), lds, st, m, t, d, cm, cd, g_colcut);
}
#else
__global__ __launch_bounds__(64) void KN(tp_col)(DevModel m, TpModel t, DevData d, ColModel cm, ColData cd) {
  extern __shared__ double tpg_dyn[];  // geom frames of the two environments
  tp_collide_prep(m, t, d, cm, cd, tpg_dyn);
}
void LN(tp_col)(dim3 g, size_t lds, hipStream_t st, const DevModel& m, const TpModel& t, const DevData& d,
                      const ColModel& cm, const ColData& cd) {
  hipLaunchKernelGGL(KN(tp_col), g, dim3(64), lds, st, m, t, d, cm, cd);
}
#endif
#elif defined(TPK_tpc)
TPC_KERNEL(tpc, false, false)
#elif defined(TPK_tpc_chain)
// Newton with contacts, chain models (see k_tp_chain).  No chain4 instantiation: the quad chain sums were measured in
// this kernel too and cost more than they saved (51.1 against 51.5 M env-steps/s; scratch 144 -> 192 B per lane)
TPC_KERNEL_FC(tpc_chain, false, false, false, true, false)
#elif defined(TPK_tpc_cg)
TPC_KERNEL_D(tpc_cg, true, false)
#elif defined(TPK_tpc_full)
// bs_forward with contacts: every intermediate written (the FULL dump of k_tp_full + the contact list)
TPC_KERNEL_F(tpc_full, true, false, false)
#elif defined(TPK_tpc_full_cg)
TPC_KERNEL_F(tpc_full_cg, true, true, false)
#elif defined(TPK_tpc_pgs) || defined(TPK_tpc_full_pgs)
// contacts, PGS: no Hessian; dynamic LDS = contact slots (wider: aref, forces, B per contact), the U staging rows,
// then M^-1 of the two environments.  k_tpc_full_pgs = bs_forward (every intermediate written), so that no eligible
// configuration needs the serial pipeline and its 43 KB per environment of solver buffers
#define TPC_PGS_KERNEL(name, FULL_)                                                                                   \
  __global__ __launch_bounds__(64, TPC_OCC) void KN(name)(DevModel m, TpModel t, DevData d, ColModel cm, ColData cd,  \
                                                          int mode, int dump) {                                      \
    __shared__ int tbl[TPL_TOTAL * 64];                                                                               \
    extern __shared__ double tpc_dyn[];                                                                               \
    double* cg = tpc_dyn;                                                                                             \
    double* U = cg + 64 * TPC_SLOT_PGS;                                                                               \
    double* minv = U + 64 * 3;                                                                                        \
    const TpShared sh = {tbl, nullptr, cg, U, nullptr, minv};                                                         \
    tp_run<true, FULL_, false, false, true>(m, t, d, 1, mode, dump, sh, cm, cd);                                      \
  }                                                                                                                   \
  void LN(name)(BS_TPC_ARGS) { hipLaunchKernelGGL(KN(name), g, dim3(64), lds, st, m, t, d, cm, cd, mode, dump); }
#ifdef TPK_tpc_pgs
TPC_PGS_KERNEL(tpc_pgs, false)
#else
TPC_PGS_KERNEL(tpc_full_pgs, true)
#endif
#else
#error "compile with -DTPK_<kernel name> (see the Makefile)"
#endif
