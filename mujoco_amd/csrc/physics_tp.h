// physics_tp.h — tree-parallel step kernel for gfx950: ONE LANE PER BODY (and its joint/dof),
// 32 lanes per environment, two environments per 64-wide wavefront.
//
// Every per-environment intermediate (poses, spatial inertias, the sparse mass-matrix row of a
// dof, its L^T D L factor, constraint rows, solver vectors) lives in the registers of the lane
// that owns the body/dof; lanes exchange parent/child/ancestor data with ds_bpermute, and
// environment-wide sums use DPP row operations + one ds_swizzle, so a step touches HBM only for
// the state it loads and the outputs it stores.  Tree recursions run level by level (depth <= 7
// for the exported hands) instead of body by body.
//
// Eligibility (checked on the host, otherwise the serial kernel in physics_dev.h is used):
// nbody <= 31 (lane 31 stays empty: the all-zero lane), at most one hinge/slide joint per body, dof depth <= TP_MAXD, every joint
// equality couples a dof (joint1) with one of its ancestors (joint2), a dof owns / is referenced
// by at most one equality, at most one actuator per joint, contacts disabled.
//
// Stage semantics and reference citations are those of physics_dev.h.
#pragma once
#include "physics_collide.h"
#include <type_traits>

// TP_LPE = lanes per environment.  32 (the default build): two environments per wavefront, up to 31 bodies and 32
// contacts per environment.  64 (the WIDE build, -DTP_LPE=64, kernels k_*_w): one environment per wavefront, up to 63
// bodies and 64 contacts — the same code with 64-bit lane masks, one more stage in the environment-wide sums and
// two-word ancestor masks, for models and contact counts the 32-lane kernels are not eligible for.
#ifndef TP_LPE
#define TP_LPE 32
#endif
#define TP_ENVS (64 / TP_LPE)  // environments per wavefront
#define TP_MAXLVL 66           // level tables: body depth <= 64
#define TP_MAXD 6   // max entries in a dof's mass-matrix row (itself + 5 ancestor dofs)
#define TP_MAXC 8   // max children per body / dof
#define TP_NPAIR ((TP_MAXD - 1) * TP_MAXD / 2)
// CHAIN instantiations unroll the child rounds of the branching level (the dofs hanging off the root dof: a hand's
// fingers) to this many: models with more children there run the generic kernels.  Unrolled to TP_MAXC = 8 a
// five-fingered hand spent three of eight rounds of every sweep fetching zeros from the zero lane.
#define TP_CHAIN_MAXC 5
// rows of the Newton Hessian that the CHAIN contact kernel factorises in registers (0: always the LDS version; models
// with more bodies take the LDS version at run time).  24 = the exported hands; measured at 65 536 envs: LDS version
// 40.7 M, 32 rows 41.9 M, 24 rows 42.9 M env-steps/s (the column lives in 2 x TPC_NREG registers during the factorisation)
#ifndef TPC_NREG
#define TPC_NREG 24
#endif

#ifndef TP_HELPER_ATTR
#define TP_HELPER_ATTR __forceinline__
#endif

// per-lane integer table, [field][32]
enum {
  TPI_PARENT = 0, TPI_DEPTH, TPI_JNT, TPI_DD, TPI_ROOT, TPI_ACT, TPI_EQ_OWN, TPI_EQ_OTHER, TPI_EQ_POS,
  TPI_EQ_SRC, TPI_ANC,                       // TPI_ANC + a, a = 0..TP_MAXD-1 (a=0 unused)
  TPI_BCHILD = TPI_ANC + TP_MAXD,            // + c
  TPI_DCHILD = TPI_BCHILD + TP_MAXC,         // + c
  TPI_COUNT = TPI_DCHILD + TP_MAXC,
  TPI_FRIC_ROW = TPI_COUNT, TPI_LIM_ROW,
  TPI_ANCMASK,  // bit k set: body lane k carries a dof on the path from this body to the root (itself included)
  TPI_ANCMASK_HI,  // lanes 32..63 (wide build)
  // derived rows (host, build_tp): what the kernels' prologue would otherwise form from the rows above with a handful
  // of instructions per entry — environment-relative ds_bpermute BYTE offsets with the fallbacks resolved (the kernel
  // adds its half's first lane), and the LDS child table's words (children 1.. as lane bytes, four per word)
  TPI_SA_ANC,                                // + a: the a-th ancestor dof's lane (own lane if none)
  TPI_SA_EQO = TPI_SA_ANC + TP_MAXD,         // lane of the equality's other dof (own lane if none)
  TPI_SA_EQS,                                // lane of the equality that reads this dof (own lane if none)
  TPI_SA_DC0, TPI_SA_BC0,                    // first dof / body child (the zero lane if none)
  TPI_CW,                                    // + r, r < 2 * TPL_CROWS: body-child words, then dof-child words
  TPI_TOTAL = TPI_CW + 2 * ((TP_MAXC - 1 + 3) / 4)
};

// fields kept in LDS (indexed inside loops or by another lane's id); the rest is read once
// from the table in global memory into registers
// LDS table rows (64 ints each): children 1.. of a body / dof as LANE BYTES, four per int (child 0 is held in a
// register; "none" is stored as the environment's zero lane, so a reader needs no test), then the ancestor masks
enum { TPL_CROWS = (TP_MAXC - 1 + 3) / 4, TPL_BCHILD = 0, TPL_DCHILD = TPL_CROWS, TPL_ANCMASK = 2 * TPL_CROWS, TPL_ANCMASK_HI = 2 * TPL_CROWS + 1,
       TPL_TOTAL = 2 * TPL_CROWS + (TP_LPE == 64 ? 2 : 1) };

// per-lane constants of the lane's constraint rows, [field][32]: everything in mj_makeImpedance /
// mj_referenceConstraint that does not depend on the state (K, B of the rows' solref; the whole
// frictionloss row), evaluated once on the host with the functions the kernels use
enum {
  TPD_IW = 0,                                      // dof_invweight0 of the lane's dof
  TPD_FRIC_D, TPD_FRIC_RF, TPD_FRIC_FL, TPD_FRIC_B,  // D = 1/R (0: no row), R * frictionloss, frictionloss, B
  TPD_LIM_K, TPD_LIM_B,                            // K < 0: joint not limited
  TPD_EQ_K, TPD_EQ_B, TPD_EQ_DIAG,                 // of the equality this lane owns
  // passive force and actuator of the lane's joint: limits that are off are +-1e300, so the clamps are unconditional
  TPD_STIFF, TPD_SPRING, TPD_ACT_GEAR, TPD_ACT_GAIN, TPD_ACT_B0, TPD_ACT_B1, TPD_ACT_B2,
  TPD_ACT_CLO, TPD_ACT_CHI, TPD_ACT_FLO, TPD_ACT_FHI,
  TPD_TOTAL
};
// ... and every model constant a lane reads per step, as ONE record per lane, [32][TPK_N] (lane-major):
// a single per-lane base address plus immediate offsets.  (Separate arrays cost one hoisted 64-bit address
// each; the compiler kept 33 of them in scratch and reloaded them every step.)
enum {
  TPK_BPOS = 0, TPK_BQUAT = 3, TPK_JPOS = 7, TPK_JAX = 10, TPK_IPOS = 13, TPK_IQUAT = 16, TPK_INERTIA = 20,
  TPK_ARMATURE = 23,
  TPK_D0 = 24,                          // + TPD_* field
  TPK_JMARGIN = TPK_D0 + TPD_TOTAL, TPK_JRANGE, TPK_JSOLIMP = TPK_JRANGE + 2,
  TPK_EQDATA = TPK_JSOLIMP + 5, TPK_EQSOLIMP = TPK_EQDATA + 5, TPK_EQQPOS0 = TPK_EQSOLIMP + 5,  // of the equality the lane owns; qpos0 of its other joint
  // joint-local pose in closed form (host-built): hinge  lp = bpos + (1-cos)*P1 - sin*P2,  lq = cos(h)*bquat + sin(h)*Q1
  // with P1 = R_b (jpos - k (k.jpos)), P2 = R_b (k x jpos), Q1 = bquat * (0, k);  slide  lp = bpos + dq*P2, P2 = R_b k
  TPK_P1 = TPK_EQQPOS0 + 1, TPK_P2 = TPK_P1 + 3, TPK_Q1 = TPK_P2 + 3,
  TPK_IB = TPK_Q1 + 4,                    // body inertia about its centre of mass in the BODY frame: 00 11 22 01 02 12
  TPK_N = (TPK_IB + 7) & ~1
};

struct TpModel {
  const int* lane;  // [TPI_TOTAL][32]
  const double* lane_d;  // [TPD_TOTAL][32]
  const double* lane_k;  // [32][TPK_N]
  const unsigned long long* tri_pk;  // [2][32]: row index (5 bits each) of the packed lower-triangle entries q = l + 32 t, t = 0..11 | 12..
  const unsigned char* tri_row;      // wide build: row index of every packed entry q, one byte each
  int maxdepth, maxdd, nroots;
  int jump_rounds;  // ceil(log2(maxdepth)): pointer-jumping rounds that reach the world from every body
  int fold_minlvl;  // leaves-to-root folds (crb, rne forces) stop here: no body above this depth carries a dof
  int quad_top;     // >= 0: lane of the one body every chain hangs off, (quad_top + 1) % 4 == 0, every later body sits in an aligned group of four lanes that is a chain off that body (tp_chain_sum_quad); -1 otherwise
  int first_dof;    // lowest lane with a dof: the rows below it are identity rows of the Newton Hessian (k_tpc skips their panels and unknowns)
  int palm_scalar;  // 1: the fold at fold_minlvl goes into ONE body whose dof has no ancestor dof (a palm): only two scalars are needed of it
  // PGS sweep order (mj_solPGS visits the rows in efc order: equality, frictionloss, limits): owner lanes
  // of the equality rows in equality order; the other rows follow the dof order, which is the lane order
  const int* pgs_order;
  int npgs;
  int roots[4];
  int lvl_bchild[TP_MAXLVL];   // max #children over bodies at depth L-1, indexed by child depth L
  int lvl_next[TP_MAXLVL];     // 1: every body at depth L is the lane after its parent (chains in depth-first order): the fold is a DPP shift
  int lvl_parent[TP_MAXLVL];   // >= 0: every body at depth L (two or more) is a child of this one body: its fold is a reduction over the lanes
  int lvl_dchild[TP_MAXD + 2]; // max #dof-children over dofs with dd == D-1, indexed by D
  int chain;                   // 1: every dof with dd >= 3 is on the lane after its parent dof's (the CHAIN kernels apply)
};

// sincos out of line: inlined into the step loop, its polynomial coefficients are hoisted out of the loop as
// register pairs, spilled, and reloaded from scratch one dependent round trip at a time inside every call
// (results by value: through pointers they went out by a store to and a load from scratch on every call)
struct TpSinCos { double sn, cs; };
__device__ __attribute__((noinline)) TpSinCos tp_sincos(double x) {
  TpSinCos r;
  sincos(x, &r.sn, &r.cs);
  return r;
}
// 1 / sqrt(x) for x in the normal range: v_rsq_f64 refined by three Goldschmidt rounds (the last one for h only),
// 12 VALU instructions against ~40 for an IEEE sqrt followed by an IEEE division; within an ulp or two, not
// correctly rounded.  For the pivots of k_tpc's dense Cholesky (24 of them per Newton iteration).
__device__ __forceinline__ double tp_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return h + h;
}
// 1 / x for finite nonzero x in the normal range: v_rcp_f64 and two Newton steps, 5 VALU instructions against ~25
// for the IEEE division sequence (scale, refine, fmas, fixup); within an ulp, not correctly rounded.  The
// tree-parallel kernels divide by pivots, impedances and curvatures that are bounded away from zero.
__device__ __forceinline__ double tp_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
// ---- cross-lane primitives.  `sa` arguments are ds_bpermute byte addresses (lane * 4).
__device__ __forceinline__ double bperm(double v, int sa) {
  const int lo = __builtin_amdgcn_ds_bpermute(sa, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(sa, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// A double moves as two v_mov_b32_dpp (a 64-bit DPP move only encodes row_newbcast on gfx950: llvm-mc, "DP ALU dpp
// only supports row_newbcast").  bound_ctrl:1 on purpose: a lane whose source is out of range or disabled reads 0 —
// the same value as `old = 0` with bound_ctrl:0 — but with bound_ctrl:0 the compiler has to load that 0 into the
// destination with a v_mov_b32 in front of EVERY DPP move (round 3: that was one VALU instruction in eight of k_tp_chain).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// v(lane) + v(lane ^ 16) inside each 32-lane half with gfx950's v_permlane16_swap (a VALU instruction:
// no trip through the LDS pipe as ds_swizzle takes).  The swap of a register with itself leaves row r0 =
// rows {0, 0, 2, 2} and r1 = rows {1, 1, 3, 3} of the input, so r0 + r1 has the same operands in the same
// order in both rows of a half: the bitwise-same sum in every lane.
__device__ __forceinline__ double sum16(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// lane masks of one environment: 32 bits, or 64 in the wide build
#if TP_LPE == 64
typedef unsigned long long tp_mask;
__device__ __forceinline__ int tp_ctz(tp_mask m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int tp_popc(tp_mask m) { return __popcll(m); }
// the lanes that any environment of the wavefront has set (uniform loop bounds)
__device__ __forceinline__ tp_mask tp_merged(unsigned long long ballot) { return ballot; }
#else
typedef unsigned tp_mask;
__device__ __forceinline__ int tp_ctz(tp_mask m) { return __builtin_ctz(m); }
__device__ __forceinline__ int tp_popc(tp_mask m) { return __popc(m); }
__device__ __forceinline__ tp_mask tp_merged(unsigned long long ballot) { return (unsigned)(ballot | (ballot >> 32)); }
#endif
__device__ __forceinline__ tp_mask tp_bit(int l) { return (tp_mask)1 << l; }
// the environment's part of a wavefront ballot (bit l = lane base + l)
__device__ __forceinline__ tp_mask tp_env_bits(unsigned long long ballot, int base) { return (tp_mask)(ballot >> base); }
#if TP_LPE == 64
// v(lane) + v(lane ^ 32): v_permlane32_swap of a register with itself leaves r0 = {low half, low half} and
// r1 = {high half, high half}, so r0 + r1 has the same operands in the same order in every lane
__device__ __forceinline__ double sum32(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
#endif
// all-reduce over the lanes of an environment; every lane gets the bitwise-same sum
__device__ __forceinline__ double env_sum(double v) {
  // the argument must be rounded before the first exchange: if the compiler fused a product in the
  // caller with the first add (fma(x_i, y_i, round(x_j y_j)) in lane i, the mirror image in lane j),
  // the two lanes of a pair would get sums that differ in the last bit, and every "environment-
  // uniform" branch downstream (line search, stopping rule) could split the lanes of an environment
  asm volatile("" : "+v"(v));
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
#if TP_LPE == 64
  return sum32(sum16(v));
#else
  return sum16(v);
#endif
}
__device__ __forceinline__ int env_sum_i(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
#if TP_LPE == 64
  const unsigned h = r[0] + r[1];
  const auto q = __builtin_amdgcn_permlane32_swap(h, h, false, false);
  return (int)(q[0] + q[1]);
#else
  return (int)(r[0] + r[1]);
#endif
}
__device__ __forceinline__ bool env_any(bool p, int base) {
  return tp_env_bits(__ballot(p), base) != 0;
}
// Root-to-body recursions by pointer jumping instead of one pass per tree level: every lane
// starts with its own term and the address of its parent's lane; a round adds (composes) what the
// jump target has gathered so far and then jumps as far again, so ceil(log2(depth)) rounds replace
// `depth` levels and no lane sits out a round.  The world lane and the lanes beyond nbody hold
// zeros / the identity and point at themselves.
template <int N>
__device__ __forceinline__ void tp_chain_sum(double* v, int par_sa, int rounds, bool isb) {
  int jp = par_sa;
  for (int r = 0; r < rounds; r++) {
    double w[N];
#pragma unroll
    for (int k = 0; k < N; k++) w[k] = bperm(v[k], jp);
    const int jn = __builtin_amdgcn_ds_bpermute(jp, jp);
    if (isb)
#pragma unroll
      for (int k = 0; k < N; k++) v[k] += w[k];
    jp = jn;
  }
}
// The same sums for a hand (TpModel::quad_top: every body after the palm sits in an aligned group of four lanes that is a
// chain off the palm, and nothing above the palm carries a dof, so every term above it is zero): the parent of a chain's
// first body is the palm, whose value never changes (it only ever receives zeros) and is fetched ONCE; inside a group the
// parent is one lane down and the grandparent two — DPP quad permutes at VALU rate, no LDS-pipe instruction.  The three
// rounds add exactly what the three pointer-jumping rounds add, in their order (masks are 0 / 1 factors: x + 1 * y and
// x + 0 * y are x + y and x), so the kernel that uses it (k_tp_chain4 = tp_run<..., CHAIN, QUAD>) stays bit-identical to the generic one.
template <int N>
__device__ __forceinline__ void tp_chain_sum_quad(double* v, int top_sa, int l, int top, int nbody) {
  const int ql = l & 3;
  const bool in = l > top && l < nbody;
  const double m1 = in && ql != 0 ? 1.0 : 0.0, p1 = in && ql == 0 ? 1.0 : 0.0;
  const double m2 = in && ql >= 2 ? 1.0 : 0.0, p2 = in && ql == 1 ? 1.0 : 0.0, p3 = in && ql == 3 ? 1.0 : 0.0;
  double pal[N];
#pragma unroll
  for (int k = 0; k < N; k++) pal[k] = bperm(v[k], top_sa);
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = __builtin_fma(p1, pal[k], __builtin_fma(m1, dpp_mov<0x90>(v[k]), v[k]));  // quad_perm [0,0,1,2]: the parent
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = __builtin_fma(p2, pal[k], __builtin_fma(m2, dpp_mov<0x44>(v[k]), v[k]));  // quad_perm [0,1,0,1]: two up, its sum so far
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = __builtin_fma(p3, pal[k], v[k]);                                           // four up: the palm
}
// world pose of every body from the poses relative to the parent (lp, lq).  mj_kinematics normalises every body's
// quaternion after its composition; products of unit quaternions stay unit to rounding, so ONE normalisation after the
// last round gives the same result to a few ulp (3 x ~20 instructions less), and the parent's rotation is applied in
// quaternion form (v + 2 w (u x v) + 2 u x (u x v): 18 instructions, 31 through the matrix)
__device__ __forceinline__ V3 tp_qrot(const Q4& q, const V3& v) {
  const V3 c = {q.y * v.z - q.z * v.y, q.z * v.x - q.x * v.z, q.x * v.y - q.y * v.x};
  const V3 t = {c.x + c.x, c.y + c.y, c.z + c.z};
  return {(v.x + q.w * t.x) + q.y * t.z - q.z * t.y, (v.y + q.w * t.y) + q.z * t.x - q.x * t.z,
          (v.z + q.w * t.z) + q.x * t.y - q.y * t.x};
}
__device__ __forceinline__ void tp_chain_pose(V3& pos, Q4& quat, const V3& lp, const Q4& lq, int par_sa, int world_sa,
                                              int rounds, bool isb) {
  pos = {0, 0, 0};
  quat = {1, 0, 0, 0};
  if (isb) { pos = lp; quat = lq; }
  int jp = par_sa;
  for (int r = 0; r < rounds; r++) {
    const V3 pp = {bperm(pos.x, jp), bperm(pos.y, jp), bperm(pos.z, jp)};
    const Q4 pq = {bperm(quat.w, jp), bperm(quat.x, jp), bperm(quat.y, jp), bperm(quat.z, jp)};
    const int jn = __builtin_amdgcn_ds_bpermute(jp, jp);
    if (isb && jp != world_sa) {
      const V3 tt = tp_qrot(pq, pos);
      pos = {pp.x + tt.x, pp.y + tt.y, pp.z + tt.z};
      quat = qmul(pq, quat);
    }
    jp = jn;
  }
  const double inv = tp_rsqrt(quat.w * quat.w + quat.x * quat.x + quat.y * quat.y + quat.z * quat.z);
  quat.w *= inv; quat.x *= inv; quat.y *= inv; quat.z *= inv;
}
__device__ __forceinline__ constexpr int pair_idx(int a, int b) {  // 1 <= a <= b <= TP_MAXD-1
  return (a - 1) * (TP_MAXD - 1) - (a - 1) * (a - 2) / 2 + (b - a);
}

typedef const int __attribute__((address_space(3))) * tp_lds_ptr;  // lane table in LDS
// Which pair of environments a workgroup steps.  Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share
// one), each XCD with its own L2, and the state arrays are [field][E]: a 128-byte line holds one field of 16
// environments = 8 consecutive pairs.  With pair = blockIdx the eight blocks that share a line sit on eight XCDs
// and every L2 fetches (and partially writes) every line; dealing each XCD a contiguous run of pairs keeps a line
// in one L2.  The grid is a multiple of 8 blocks (batchsim.hip::grid_tp_n); pairs past the batch are idle.
__device__ __forceinline__ int tp_block() { return (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); }
typedef double __attribute__((address_space(3))) * tp_lds_d;
typedef int __attribute__((address_space(3))) * tp_lds_i;

// ---- contacts in the tree-parallel layout (k_tpc): contact c of an environment is owned by lane
// c (so at most 32 contacts); its geometry lives in an LDS slot, its rows in the lane's registers.
// contact slot in LDS: CG_MASK holds the ancestor masks of the contact's two bodies (two 32-bit words in one double;
// two 64-bit words in the wide build, which shifts what follows by one double)
__host__ __device__ constexpr int tpc_slot(int lpe) { return lpe == 64 ? 19 : 17; }       // doubles per contact slot (odd stride)
__host__ __device__ constexpr int tpc_slot_pgs(int lpe) { return lpe == 64 ? 33 : 31; }   // ... of k_tpc_pgs (odd stride)
#define TPC_SLOT tpc_slot(TP_LPE)
#define TPC_SLOT_PGS tpc_slot_pgs(TP_LPE)
#define TPC_HTRI (32 * 33 / 2)                   // 32-lane build: packed lower-triangular Hessian, one per env
#define TPC_NTRI ((TPC_HTRI + 31) / 32)
enum { CG_POS = 0, CG_FRAME = 3, CG_DIST = 12, CG_PAIR = 13, CG_D = 14, CG_MASK = 15, CG_MU = 15 + (TP_LPE == 64 ? 2 : 1),  // PAIR: pair id | active rows (Newton) / row count (PGS)
       CG_AREF = CG_MU + 1, CG_F = CG_AREF + 4, CG_B = CG_F + 4 };  // PGS only: aref and force of the 4 rows, B = U M^-1 U^T (nn, n1, n2, 11, 12, 22)
struct TpShared { int* tbl; double* H; double* cg; double* U; double* park; double* minv; };  // park: [TP_MAXD + 1][64] or null
// per-environment M^-1 in LDS (PGS): nbody rows of (nbody | 1) doubles (odd stride: conflict-free columns)
__host__ __device__ inline int tp_minv_stride(int nbody) { return nbody | 1; }
// the two environments of a wavefront diverge, so LDS hand-offs between lanes of one environment
// are ordered with a fence + scheduling barrier instead of s_barrier (LDS is in-order per wave)
// Every block of these kernels is ONE wavefront and the hardware executes a wavefront's LDS
// instructions in order, so nothing has to be waited for: the fence (wavefront scope: no s_waitcnt,
// in particular none for outstanding scratch traffic) and the scheduling barrier only keep the
// compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void tp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
typedef double __attribute__((address_space(3))) * tp_lds_d_;
typedef int __attribute__((address_space(3))) * tp_lds_i_;
// ancestor masks of a contact's two bodies in its slot
__device__ __forceinline__ tp_mask tp_slot_mask(tp_lds_d_ sl, int which) {
#if TP_LPE == 64
  return ((unsigned long long __attribute__((address_space(3)))*)(sl + CG_MASK))[which];
#else
  return (tp_mask)((tp_lds_i_)(sl + CG_MASK))[which];
#endif
}
__device__ __forceinline__ void tp_slot_mask_set(tp_lds_d_ sl, int which, tp_mask m) {
#if TP_LPE == 64
  ((unsigned long long __attribute__((address_space(3)))*)(sl + CG_MASK))[which] = m;
#else
  ((tp_lds_i_)(sl + CG_MASK))[which] = (int)m;
#endif
}
// ancestor mask of body lane `lane` (wavefront lane index) from the LDS table
__device__ __forceinline__ tp_mask tp_anc_mask(const int* tbl, int lane) {
#if TP_LPE == 64
  return (unsigned long long)(unsigned)tbl[TPL_ANCMASK * 64 + lane] | ((unsigned long long)(unsigned)tbl[TPL_ANCMASK_HI * 64 + lane] << 32);
#else
  return (tp_mask)tbl[TPL_ANCMASK * 64 + lane];
#endif
}
struct TpRow { double r[TP_MAXD]; };
struct TpFac { TpRow R; double dinv; };

// what the dof-tree helpers need about this lane (passed by value, lives in registers)
struct TpCtx {
  int self_sa;           // own bpermute address
  int zero_sa;           // address of a lane beyond nbody: everything it holds is zero, so a lane that has
                         // nothing to receive reads from there and adds unconditionally (no per-value select)
  int anc_sa[TP_MAXD];   // address of the a-th ancestor dof's lane (own address if none)
  int dd;                // entries in this dof's row (0 = no dof on this lane)
  int maxdd;
  int nchild[TP_MAXD + 2];  // lvl_dchild
  int top_sa;            // address of the lane's (dd - 1)-th ancestor dof: the one above its chain (CHAIN kernels)
  int dchild0;           // first entry of the lane's dof-child list (the LDS table's row 0): most levels have one child,
                         // so the sweeps' usual case needs no table read in front of the permutes
  int dchild_w;          // dof-children 1..4 as lane bytes (row TPL_DCHILD of the LDS table), in a register for the CHAIN
                         // kernels' branching level: no table read in front of its fetches
  tp_lds_ptr tbl;        // LDS table
};
// 1.0 on the lane that receives at level D of a CHAIN sweep (its dof count is D - 1 and its child dof sits on the next
// lane), 0.0 elsewhere.  TP_RECV_OPAQUE (the contact kernel): formed from an opaque copy of the dof count where it is
// used — hoisted out of the solver loop the factors of the four shifted levels were eight more registers live through
// the whole kernel (scratch 184 -> 152 B per lane)
__device__ __forceinline__ double tp_recv_factor(const TpCtx& c, int D) {
  int dd = c.dd;
#ifdef TP_RECV_OPAQUE
  asm volatile("" : "+v"(dd));
#endif
  return dd == D - 1 && c.dchild0 == c.self_sa + 4 ? 1.0 : 0.0;
}
// ds_bpermute address of child ch >= 1 of this lane (row0 = TPL_BCHILD or TPL_DCHILD)
__device__ __forceinline__ int tp_child(tp_lds_ptr tbl, int row0, int ch) {
  const int w = tbl[(row0 + ((ch - 1) >> 2)) * 64 + threadIdx.x];
  return ((w >> (((ch - 1) & 3) * 8)) & 0xff) << 2;
}

static_assert(TP_CHAIN_MAXC <= 5, "tp_child_w holds children 1..4");
__device__ __forceinline__ int tp_child_w(int w, int ch) { return ((w >> ((ch - 1) * 8)) & 0xff) << 2; }

// what a parent dof receives from its child in a leaves-to-root sweep: a ds_bpermute per value, or — in the CHAIN
// instantiation, on the levels below the first branching (TpModel::chain: every dof of level D >= 3 sits on the lane
// after its parent dof's, single-child chains numbered depth-first) — a DPP wavefront shift, which keeps the
// sweeps' packets off the CU's LDS pipe.  Compile-time per level: a run-time choice doubled the sweeps' code and
// register pressure (scratch 96 -> 280 B) and lost more than the shifts won.
// A shifted value arrives on EVERY lane (whatever the next lane holds: a finite packet, see below); the lanes it is
// not meant for take it with the factor rm = 0, the receiving lane with rm = 1 — tp_add / tp_sub: one FMA (exact:
// v * 1 + acc rounds once, like acc + v) where a select of v cost two v_cndmask in front of the add.
__device__ __forceinline__ double tp_up(double x, int sa, bool shift) {
  if (shift) return dpp_mov<0x130>(x);  // wave_shl:1: lane l reads lane l + 1 (shift: constant once the level loop is unrolled)
  return bperm(x, sa);
}
__device__ __forceinline__ double tp_add(double acc, double v, double rm, bool shift) { return shift ? __builtin_fma(v, rm, acc) : acc + v; }
__device__ __forceinline__ double tp_sub(double acc, double v, double rm, bool shift) { return shift ? __builtin_fma(-v, rm, acc) : acc - v; }
// The packets a level sends (T, S, U below) carry NO sender-side select: a receiver reads either its child — a
// lane of the level that sends — or the environment's zero lane, whose rows, accumulators and right-hand side are
// exact zeros and so are the packets it forms; what the other lanes compute at a level that is not theirs is never
// read (round 3: the selects were 5 % of k_tp_chain's VALU instructions).
// sparse L^T D L of the distributed rows (mj_factorM layout): R[0] diagonal, R[a] the entry with
// the a-th ancestor dof.  Level by level from the leaves: a finished dof sends the outer-product
// packet T(a,b) = L(k,a) M(k,b) + (packets forwarded from below) to its parent, which applies the
// (1,b) entries to its own row and forwards the rest one position up.
template <bool CHAIN = false>
__device__ TP_HELPER_ATTR TpFac tp_factor(TpCtx c, TpRow Rin) {
  double* R = Rin.r;
  double A[TP_NPAIR];
#pragma unroll
  for (int k = 0; k < TP_NPAIR; k++) A[k] = 0;
  double dinv = 1.0;
#pragma unroll
  for (int D = TP_MAXD; D >= 1; D--) {
    if (D > c.maxdd) continue;
    const bool is = c.dd == D;
    double T[TP_NPAIR];
    const double r0 = is ? R[0] : 1.0;
    const double ir0 = tp_rcp(r0);
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++) {
      if (a >= D) continue;
      const double tmp = R[a] * ir0;
#pragma unroll
      for (int b = a; b < TP_MAXD; b++) {
        if (b >= D) continue;
        T[pair_idx(a, b)] = tmp * R[b] + A[pair_idx(a, b)];
      }
    }
    if (is) dinv = ir0;
    if (D == 1) break;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
#pragma unroll
        for (int b = a; b < TP_MAXD; b++) {
          if (b >= D) continue;
          const double v = tp_up(T[pair_idx(a, b)], sa, CHAIN && D >= 3);
          if (a == 1) R[b - 1] = tp_sub(R[b - 1], v, rm, CHAIN && D >= 3);
          else A[pair_idx(a - 1, b - 1)] = tp_add(A[pair_idx(a - 1, b - 1)], v, rm, CHAIN && D >= 3);
        }
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
  // L(k, a) = M(k, a) / d(k): the row is final from the lane's own level on (nothing is received after it), so it is
  // scaled once here with the pivot's reciprocal of that level — the same product the level formed as tmp
  TpFac out;
  out.R.r[0] = R[0];
#pragma unroll
  for (int a = 1; a < TP_MAXD; a++) out.R.r[a] = R[a] * dinv;
  out.dinv = dinv;
  return out;
}

// root-to-leaves half of a solve: x(k) -= sum over the ancestors a of L(k, a) x(a), level by level from the root.
// Every lane forms the update of every level (what it fetches at a level that is not its own is finite and dropped)
// and the lanes of the level keep it: one select per level instead of one per term, same order of rounding.
template <bool CHAIN>
__device__ __forceinline__ double tp_back(const TpCtx& c, const double* R, double x) {
#pragma unroll
  for (int D = 2; D <= TP_MAXD; D++) {
    if (D > c.maxdd) continue;
    double xs = x, xn = x;  // CHAIN: xs = x of the lane a places below (the ancestors of a chain dof, up to the first branching)
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++) {
      if (a >= D) continue;
      double xa;
      if (CHAIN && a <= D - 2) { xs = dpp_mov<0x138>(xs); xa = xs; }  // wave_shr:1
      else xa = bperm(x, CHAIN ? c.top_sa : c.anc_sa[a]);  // CHAIN: a == D - 1, the dof above the chain
      xn -= R[a] * xa;
    }
    if (c.dd == D) x = xn;
  }
  return x;
}

// tp_factor followed by tp_solve with the same matrix, in one leaves-to-root pass: when the rows of
// level D are final, their part of the forward substitution is sent up in the same exchange
template <bool CHAIN = false>
__device__ TP_HELPER_ATTR double tp_factor_solve(TpCtx c, TpRow Rin, double x) {
  double* R = Rin.r;
  double A[TP_NPAIR], W[TP_MAXD];
#pragma unroll
  for (int k = 0; k < TP_NPAIR; k++) A[k] = 0;
#pragma unroll
  for (int a = 0; a < TP_MAXD; a++) W[a] = 0;
  double dinv = 1.0;
#pragma unroll
  for (int D = TP_MAXD; D >= 1; D--) {
    if (D > c.maxdd) continue;
    const bool is = c.dd == D;
    double T[TP_NPAIR], S[TP_MAXD];
    const double r0 = is ? R[0] : 1.0;
    const double ir0 = tp_rcp(r0);
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++) {
      if (a >= D) continue;
      const double tmp = R[a] * ir0;
#pragma unroll
      for (int b = a; b < TP_MAXD; b++) {
        if (b >= D) continue;
        T[pair_idx(a, b)] = tmp * R[b] + A[pair_idx(a, b)];
      }
      S[a] = tmp * x + W[a];
    }
    if (is) dinv = ir0;
    if (D == 1) break;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
        const double sv = tp_up(S[a], sa, CHAIN && D >= 3);
        if (a == 1) x = tp_sub(x, sv, rm, CHAIN && D >= 3);
        else W[a - 1] = tp_add(W[a - 1], sv, rm, CHAIN && D >= 3);
#pragma unroll
        for (int b = a; b < TP_MAXD; b++) {
          if (b >= D) continue;
          const double v = tp_up(T[pair_idx(a, b)], sa, CHAIN && D >= 3);
          if (a == 1) R[b - 1] = tp_sub(R[b - 1], v, rm, CHAIN && D >= 3);
          else A[pair_idx(a - 1, b - 1)] = tp_add(A[pair_idx(a - 1, b - 1)], v, rm, CHAIN && D >= 3);
        }
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
#pragma unroll
  for (int a = 1; a < TP_MAXD; a++) R[a] *= dinv;  // as in tp_factor: the row scaled once, after its level
  x *= dinv;
  return tp_back<CHAIN>(c, R, x);
}

// two factorisations with the same sparsity at once (M and M + h * damping): the level loop and its
// cross-lane round trips are shared, so the second factor costs issue slots, not latency
template <bool CHAIN = false>
__device__ TP_HELPER_ATTR void tp_factor2(TpCtx c, TpRow R0in, TpRow R1in, TpFac& out0, TpFac& out1) {
  double* R[2] = {R0in.r, R1in.r};
  double A[2][TP_NPAIR];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int k = 0; k < TP_NPAIR; k++) A[n][k] = 0;
  double dinv[2] = {1.0, 1.0};
#pragma unroll
  for (int D = TP_MAXD; D >= 1; D--) {
    if (D > c.maxdd) continue;
    const bool is = c.dd == D;
    double T[2][TP_NPAIR];
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const double r0 = is ? R[n][0] : 1.0;
      const double ir0 = tp_rcp(r0);
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
        const double tmp = R[n][a] * ir0;
#pragma unroll
        for (int b = a; b < TP_MAXD; b++) {
          if (b >= D) continue;
          T[n][pair_idx(a, b)] = tmp * R[n][b] + A[n][pair_idx(a, b)];
        }
      }
      if (is) dinv[n] = ir0;
    }
    if (D == 1) break;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
#pragma unroll
        for (int b = a; b < TP_MAXD; b++) {
          if (b >= D) continue;
#pragma unroll
          for (int n = 0; n < 2; n++) {
            const double v = tp_up(T[n][pair_idx(a, b)], sa, CHAIN && D >= 3);
            if (a == 1) R[n][b - 1] = tp_sub(R[n][b - 1], v, rm, CHAIN && D >= 3);
            else A[n][pair_idx(a - 1, b - 1)] = tp_add(A[n][pair_idx(a - 1, b - 1)], v, rm, CHAIN && D >= 3);
          }
        }
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
  out0.R.r[0] = R[0][0]; out1.R.r[0] = R[1][0];
#pragma unroll
  for (int a = 1; a < TP_MAXD; a++) { out0.R.r[a] = R[0][a] * dinv[0]; out1.R.r[a] = R[1][a] * dinv[1]; }
  out0.dinv = dinv[0]; out1.dinv = dinv[1];
}

// x <- (L^T D L)^-1 x (mj_solveM): leaves-to-root sweep, diagonal scale, root-to-leaves sweep
template <bool CHAIN = false>
__device__ TP_HELPER_ATTR double tp_solve(TpCtx c, TpRow Rin, double dinv, double x) {
  const double* R = Rin.r;
  double W[TP_MAXD];
#pragma unroll
  for (int a = 0; a < TP_MAXD; a++) W[a] = 0;
#pragma unroll
  for (int D = TP_MAXD; D >= 2; D--) {
    if (D > c.maxdd) continue;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    double T[TP_MAXD];
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++)
      if (a < D) T[a] = R[a] * x + W[a];
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
        const double v = tp_up(T[a], sa, CHAIN && D >= 3);
        if (a == 1) x = tp_sub(x, v, rm, CHAIN && D >= 3);
        else W[a - 1] = tp_add(W[a - 1], v, rm, CHAIN && D >= 3);
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
  x *= dinv;
  return tp_back<CHAIN>(c, R, x);
}

// tp_solve(R, dinv, x) and tp_mul_m(M, v) in one pass: the leaves-to-root sweeps of the two have the
// same level loop, so the product rides along in the solve's cross-lane round trips
template <bool CHAIN = false>
__device__ TP_HELPER_ATTR double tp_solve_mulm(TpCtx c, TpRow Rin, double dinv, double x, TpRow Min, double v, double& mv) {
  const double* R = Rin.r;
  const double* Mr = Min.r;
  double res = c.dd > 0 ? Mr[0] * v : 0.0;
  double vs = v;
  const double vtop = CHAIN ? bperm(v, c.top_sa) : 0.0;
#pragma unroll
  for (int a = 1; a < TP_MAXD; a++) {
    if (a >= c.maxdd) continue;
    double va;
    if (CHAIN) {  // the a-th ancestor is the lane a places below, or (a == dd - 1) the dof above the chains
      vs = dpp_mov<0x138>(vs);
      va = a == c.dd - 1 ? vtop : vs;
    } else {
      va = bperm(v, c.anc_sa[a]);
    }
    if (a < c.dd) res += Mr[a] * va;
  }
  double W[TP_MAXD], V[TP_MAXD];
#pragma unroll
  for (int a = 0; a < TP_MAXD; a++) { W[a] = 0; V[a] = 0; }
#pragma unroll
  for (int D = TP_MAXD; D >= 2; D--) {
    if (D > c.maxdd) continue;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    double T[TP_MAXD], U[TP_MAXD];
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++)
      if (a < D) { T[a] = R[a] * x + W[a]; U[a] = Mr[a] * v + V[a]; }
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
        const double t = tp_up(T[a], sa, CHAIN && D >= 3), u = tp_up(U[a], sa, CHAIN && D >= 3);
        if (a == 1) { x = tp_sub(x, t, rm, CHAIN && D >= 3); res = tp_add(res, u, rm, CHAIN && D >= 3); }
        else { W[a - 1] = tp_add(W[a - 1], t, rm, CHAIN && D >= 3); V[a - 1] = tp_add(V[a - 1], u, rm, CHAIN && D >= 3); }
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
  mv = res;
  x *= dinv;
  return tp_back<CHAIN>(c, R, x);
}

// N right-hand sides at once: the level sweeps and their cross-lane round trips are shared
template <int N, bool CHAIN = false>
__device__ TP_HELPER_ATTR void tp_solve_n(TpCtx c, TpRow Rin, double dinv, double (&x)[N]) {
  const double* R = Rin.r;
  double W[N][TP_MAXD];
#pragma unroll
  for (int n = 0; n < N; n++)
#pragma unroll
    for (int a = 0; a < TP_MAXD; a++) W[n][a] = 0;
#pragma unroll
  for (int D = TP_MAXD; D >= 2; D--) {
    if (D > c.maxdd) continue;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    double T[N][TP_MAXD];
#pragma unroll
    for (int n = 0; n < N; n++)
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++)
        if (a < D) T[n][a] = R[a] * x[n] + W[n][a];
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
#pragma unroll
        for (int n = 0; n < N; n++) {
          const double v = tp_up(T[n][a], sa, CHAIN && D >= 3);
          if (a == 1) x[n] = tp_sub(x[n], v, rm, CHAIN && D >= 3);
          else W[n][a - 1] = tp_add(W[n][a - 1], v, rm, CHAIN && D >= 3);
        }
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
#pragma unroll
  for (int n = 0; n < N; n++) x[n] *= dinv;
#pragma unroll
  for (int D = 2; D <= TP_MAXD; D++) {
    if (D > c.maxdd) continue;
    double xn[N];
#pragma unroll
    for (int n = 0; n < N; n++) xn[n] = x[n];
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++) {
      if (a >= D) continue;
#pragma unroll
      for (int n = 0; n < N; n++) xn[n] -= R[a] * bperm(x[n], c.anc_sa[a]);
    }
    if (c.dd == D)
#pragma unroll
      for (int n = 0; n < N; n++) x[n] = xn[n];
  }
}

// res = M v with the distributed sparse rows (mj_mulM)
template <bool CHAIN = false>
__device__ TP_HELPER_ATTR double tp_mul_m(TpCtx c, TpRow Min, double v) {
  const double* Mr = Min.r;
  double res = c.dd > 0 ? Mr[0] * v : 0.0;
  double vs = v;
  const double vtop = CHAIN ? bperm(v, c.top_sa) : 0.0;
#pragma unroll
  for (int a = 1; a < TP_MAXD; a++) {
    if (a >= c.maxdd) continue;
    double va;
    if (CHAIN) {  // the a-th ancestor is the lane a places below, or (a == dd - 1) the dof above the chains
      vs = dpp_mov<0x138>(vs);
      va = a == c.dd - 1 ? vtop : vs;
    } else {
      va = bperm(v, c.anc_sa[a]);
    }
    if (a < c.dd) res += Mr[a] * va;
  }
  double W[TP_MAXD];
#pragma unroll
  for (int a = 0; a < TP_MAXD; a++) W[a] = 0;
#pragma unroll
  for (int D = TP_MAXD; D >= 2; D--) {
    if (D > c.maxdd) continue;
    const bool par = c.dd == D - 1;
    const double rm = tp_recv_factor(c, D);  // CHAIN: 1 on the lane whose child dof is on the next lane
    double T[TP_MAXD];
#pragma unroll
    for (int a = 1; a < TP_MAXD; a++)
      if (a < D) T[a] = Mr[a] * v + W[a];
    auto round = [&](int ch) {
      const int src = ch == 0 ? c.dchild0 : (CHAIN && D == 2 ? tp_child_w(c.dchild_w, ch) : tp_child(c.tbl, TPL_DCHILD, ch));
      const int sa = par ? src : c.zero_sa;
#pragma unroll
      for (int a = 1; a < TP_MAXD; a++) {
        if (a >= D) continue;
        const double x = tp_up(T[a], sa, CHAIN && D >= 3);
        if (a == 1) res = tp_add(res, x, rm, CHAIN && D >= 3);
        else W[a - 1] = tp_add(W[a - 1], x, rm, CHAIN && D >= 3);
      }
    };
    if (CHAIN && D == 2) {  // the branching level: every child's fetch issued before the first accumulation (table
                            // entries beyond the children name the zero lane), same order of accumulation
#pragma unroll
      for (int ch = 0; ch < TP_CHAIN_MAXC; ch++) round(ch);
    } else {
      for (int ch = 0; ch < (CHAIN && D >= 3 ? 1 : c.nchild[D]); ch++) round(ch);
    }
  }
  return res;
}

// per-lane constraint rows (fixed layout: friction row, two limit rows, one owned equality row)
struct TpRows {
  double Df, rf, aref_f, fl;     // dof friction loss
  double Dl0, Dl1, aref_l0, aref_l1;  // lower / upper joint limit (D = 0 when inactive)
  double De, aref_e, ce;         // owned equality: J = e_own - ce * e_other
  double De_o, ce_o;             // equality in which this dof is the 'other' one (from its owner)
  bool own;                      // owns an equality row
  double cD;                     // contact owned by this lane: D of its (up to 4) pyramid rows, 0 = none
};

struct TpSol {
  double qacc, Ma, grad, search, Mv;
  double jf, jl0, jl1, je;      // residuals J*qacc - aref
  double vf, vl0, vl1, ve;      // J*search
  double cj[4], cv[4];          // contact rows (k_tpc only): residuals and J*search
};

// impedance with the common power == 2 case evaluated without pow()
__host__ __device__ __forceinline__ double tp_impedance(const double* solimp, double pos, double margin) {
  const double dmin = fmin(fmax(solimp[0], BS_MINIMP), BS_MAXIMP), dmax = fmin(fmax(solimp[1], BS_MINIMP), BS_MAXIMP);
  const double width = fmax(solimp[2], BS_MINVAL), mid = fmin(fmax(solimp[3], BS_MINIMP), BS_MAXIMP);
  const double power = fmax(solimp[4], 1.0);
#ifdef __HIP_DEVICE_COMPILE__
  const double x = fabs(pos - margin) * tp_rcp(width);
#else
  const double x = fabs(pos - margin) / width;
#endif
  if (dmin == dmax || x >= 1.0) return (x >= 1.0) ? dmax : 0.5 * (dmin + dmax);
  if (x <= 0) return dmin;
  double y;
  if (power == 1.0) y = x;
#ifdef __HIP_DEVICE_COMPILE__
  else if (power == 2.0) y = (x <= mid) ? x * x * tp_rcp(mid) : 1 - (1 - x) * (1 - x) * tp_rcp(1 - mid);
#else
  else if (power == 2.0) y = (x <= mid) ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
#endif
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return dmin + y * (dmax - dmin);
}

// cost of this lane's rows at its residuals; optional forces f = {friction, lower, upper, equality}
template <bool FORCE>
__device__ __forceinline__ double tp_rows_cost(const TpRows& r, double jf, double jl0, double jl1, double je,
                                               double* f) {
  double cost = 0, fe = 0, ff = 0, f0 = 0, f1 = 0;
  if (r.own) { cost += 0.5 * r.De * je * je; fe = -r.De * je; }
  if (r.Df > 0) {
    if (jf <= -r.rf) { ff = r.fl; cost += -r.fl * (0.5 * r.rf + jf); }
    else if (jf >= r.rf) { ff = -r.fl; cost += -r.fl * (0.5 * r.rf - jf); }
    else { ff = -r.Df * jf; cost += 0.5 * r.Df * jf * jf; }
  }
  if (jl0 < 0) { f0 = -r.Dl0 * jl0; cost += 0.5 * r.Dl0 * jl0 * jl0; }
  if (jl1 < 0) { f1 = -r.Dl1 * jl1; cost += 0.5 * r.Dl1 * jl1 * jl1; }
  if (FORCE) { f[0] = ff; f[1] = f0; f[2] = f1; f[3] = fe; }
  return cost;
}

// cost (and forces) of the lane's contact rows: f = -D * min(jar, 0)
template <bool FORCE>
__device__ __forceinline__ double tp_con_cost(double D, const double* cj, double* cf) {
  double cost = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const double x = cj[q];
    if (FORCE) cf[q] = x < 0 ? -D * x : 0.0;
    if (x < 0) cost += 0.5 * D * x * x;
  }
  return cost;
}

struct TpLs { double alpha, cost, d0, d1; };

template <bool CON>
__device__ TP_HELPER_ATTR TpLs tp_ls_eval(TpRows r, TpSol s, double alpha, double g0, double g1, double g2) {
  double q0 = 0, q1 = 0, q2 = 0;
  // J * search of the lane's rows made opaque per evaluation: otherwise the curvature terms 0.5 D v^2 (they do not
  // depend on alpha) are hoisted out of the inlined evaluations, spilled, and reloaded from scratch inside each one —
  // a dependent round trip to memory for two multiplications
  asm volatile("" : "+v"(s.vf), "+v"(s.ve), "+v"(s.vl0), "+v"(s.vl1));
  if constexpr (CON) asm volatile("" : "+v"(s.cv[0]), "+v"(s.cv[1]), "+v"(s.cv[2]), "+v"(s.cv[3]));
  if constexpr (CON) {
#pragma unroll
    // one-sided rows without selects on q0 / q1: min(x, 0) contributes the same products when the row is
    // active and exact zeros when it is not
    for (int q = 0; q < 4; q++) {
      const double x = s.cj[q] + alpha * s.cv[q];
      const double xm = fmin(x, 0.0);
      q0 += 0.5 * r.cD * xm * xm; q1 += r.cD * xm * s.cv[q];
      if (x < 0) q2 += 0.5 * r.cD * s.cv[q] * s.cv[q];
    }
  }
  if (r.own) {
    const double x = s.je + alpha * s.ve;
    q0 += 0.5 * r.De * x * x; q1 += r.De * x * s.ve; q2 += 0.5 * r.De * s.ve * s.ve;
  }
  if (r.Df > 0) {
    const double x = s.jf + alpha * s.vf;
    if (x <= -r.rf) { q0 += -r.fl * (0.5 * r.rf + x); q1 += -r.fl * s.vf; }
    else if (x >= r.rf) { q0 += -r.fl * (0.5 * r.rf - x); q1 += r.fl * s.vf; }
    else { q0 += 0.5 * r.Df * x * x; q1 += r.Df * x * s.vf; q2 += 0.5 * r.Df * s.vf * s.vf; }
  }
  if (__ballot(r.Dl0 != 0 || r.Dl1 != 0)) {  // wave-uniform: most steps have no joint inside a limit's margin
    {
      const double x = s.jl0 + alpha * s.vl0, xm = fmin(x, 0.0);
      q0 += 0.5 * r.Dl0 * xm * xm; q1 += r.Dl0 * xm * s.vl0;
      if (x < 0) q2 += 0.5 * r.Dl0 * s.vl0 * s.vl0;
    }
    {
      const double x = s.jl1 + alpha * s.vl1, xm = fmin(x, 0.0);
      q0 += 0.5 * r.Dl1 * xm * xm; q1 += r.Dl1 * xm * s.vl1;
      if (x < 0) q2 += 0.5 * r.Dl1 * s.vl1 * s.vl1;
    }
  }
  q0 = env_sum(q0); q1 = env_sum(q1); q2 = env_sum(q2);
  TpLs p;
  p.alpha = alpha;
  p.cost = g0 + q0 + alpha * g1 + alpha * alpha * g2;
  p.d0 = g1 + q1 + 2 * alpha * g2;
  p.d1 = 2 * (g2 + q2);
  if (p.d1 <= 0) p.d1 = BS_MINVAL;
  return p;
}

// exact line search (same shape as physics_dev.h::line_search); all values are env-uniform
template <bool CON>
__device__ __forceinline__ double tp_line_search(const DevModel& m, const TpRows& r, const TpSol& s,
                                                 double gauss, double scale, double qfrc_smooth) {
  const double snorm = sqrt(env_sum(s.search * s.search));
  if (snorm < BS_MINVAL) return 0;
  const double gtol = m.tolerance * m.ls_tolerance * snorm / scale;
  const double g1 = env_sum(s.search * (s.Ma - qfrc_smooth));
  const double g2 = 0.5 * env_sum(s.search * s.Mv);
  int it = 0;
  const int maxit = m.ls_iterations;
  const TpLs p0 = tp_ls_eval<CON>(r, s, 0.0, gauss, g1, g2);
  TpLs p1 = tp_ls_eval<CON>(r, s, p0.alpha - p0.d0 * tp_rcp(p0.d1), gauss, g1, g2);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.d0) < gtol) return p1.alpha;
  const double dir = p1.d0 < 0 ? 1.0 : -1.0;
  bool p2update = false;
  TpLs p2 = p1;
  while (p1.d0 * dir <= -gtol && it < maxit) {
    p2 = p1; p2update = true;
    p1 = tp_ls_eval<CON>(r, s, p1.alpha - p1.d0 * tp_rcp(p1.d1), gauss, g1, g2);
    it++;
    if (fabs(p1.d0) < gtol) return p1.alpha;
  }
  if (it >= maxit || !p2update) return p1.alpha;
  TpLs p2next = p1;
  TpLs p1next = tp_ls_eval<CON>(r, s, p1.alpha - p1.d0 * tp_rcp(p1.d1), gauss, g1, g2);
  while (it < maxit) {
    const TpLs pmid = tp_ls_eval<CON>(r, s, 0.5 * (p1.alpha + p2.alpha), gauss, g1, g2);
    it++;
    bool b1 = false, b2 = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const TpLs c = k == 0 ? p1next : (k == 1 ? p2next : pmid);
      if (fabs(c.d0) < gtol) return c.alpha;
      if (c.d0 * dir < 0) {
        if (c.alpha * dir > p2.alpha * dir && c.alpha * dir < p1.alpha * dir) { p2 = c; b2 = true; }
      } else {
        if (c.alpha * dir < p1.alpha * dir && c.alpha * dir > p2.alpha * dir) { p1 = c; b1 = true; }
      }
    }
    if (!b1 && !b2) break;
    if (b1) p1next = tp_ls_eval<CON>(r, s, p1.alpha - p1.d0 * tp_rcp(p1.d1), gauss, g1, g2);
    if (b2) p2next = tp_ls_eval<CON>(r, s, p2.alpha - p2.d0 * tp_rcp(p2.d1), gauss, g1, g2);
  }
  const TpLs best = p1.cost <= p2.cost ? p1 : p2;
  return best.cost < p0.cost ? best.alpha : 0.0;
}

// ------------------------------------------------------------------ k_tp_col
// Collision preparation in the tree-parallel layout: kinematics (lane = body, same arithmetic as
// tp_run's A1), geom frames (lane = geom) and the broadphase (lane = candidate pair, bounding
// spheres + oriented-box filter, ordered ballot compaction = col_broadphase), two environments
// per wavefront.  Replaces k_c_pos + k_c_broad when k_tpc runs the dynamics; a state that fails
// mj_checkPos / mj_checkVel is treated as reset (k_tpc performs the reset itself).
#define TPG_SLOT 13  // doubles per geom frame in LDS: position, matrix (odd stride)
#ifdef TPC_CUTS
#define COL_T(i) { if (colcut == i) return; }
#else
#define COL_T(i)
#endif
__device__ __forceinline__ void tp_collide_prep(const DevModel& m, const TpModel& t, const DevData& d,
                                                const ColModel& cm, const ColData& cd, double* gfr, int colcut = -1) {
  COL_T(0)
  const int tid = threadIdx.x;
  const int l = tid & (TP_LPE - 1), base = tid & ~(TP_LPE - 1);
  long env = d.env0 + (long)tp_block() * TP_ENVS + (tid / TP_LPE);
  const bool env_ok = env < d.n_envs;
  if (!env_ok) env = d.n_envs - 1;
  const long E = d.E, e = env;
  const int self_sa = tid * 4;
  const int pv = t.lane[TPI_PARENT * TP_LPE + l];
  const int par_sa = pv >= 0 ? (base + pv) * 4 : self_sa;
  const int jnt = t.lane[TPI_JNT * TP_LPE + l];
  const bool has_dof = jnt >= 0, isb = l < m.nbody && l > 0;
  const int j = has_dof ? jnt : 0;
  // (the state is loaded on every lane, row 0 where there is no dof, and selected afterwards: guarded loads were
  // emitted one behind the other, each with its own wait; the pose constants come from the lane's one record)
  const double qpos_l = AT(d.qpos, j), qvel_l = AT(d.qvel, j);
  const double* const lk = t.lane_k + l * TPK_N;
  const V3 bpos = ld3(lk + TPK_BPOS);
  const Q4 bquat = ld4(lk + TPK_BQUAT);
  const int jtype = m.jnt_type[j];
  const double qpos0 = m.qpos0[j];
  double qpos = has_dof ? qpos_l : 0.0;
  const double qvel = has_dof ? qvel_l : 0.0;
  if (env_any(has_dof && (bad_num(qpos) || bad_num(qvel)), base)) qpos = qpos0;
  const double dq = qpos - qpos0;
  Q4 lq = bquat;
  V3 lp = bpos;
  if (has_dof) {  // the closed form of tp_run's joint-local pose
    const V3 p2 = ld3(lk + TPK_P2);
    if (jtype == BS_JNT_SLIDE) {
      lp = {bpos.x + p2.x * dq, bpos.y + p2.y * dq, bpos.z + p2.z * dq};
    } else {
      double sn, cs;
      sincos(0.5 * dq, &sn, &cs);
      const V3 p1 = ld3(lk + TPK_P1);
      const Q4 q1 = ld4(lk + TPK_Q1);
      const double omc = 2.0 * (sn * sn), s1 = 2.0 * (sn * cs);
      lq = {cs * bquat.w + sn * q1.w, cs * bquat.x + sn * q1.x, cs * bquat.y + sn * q1.y, cs * bquat.z + sn * q1.z};
      lp = {(bpos.x + omc * p1.x) - s1 * p2.x, (bpos.y + omc * p1.y) - s1 * p2.y, (bpos.z + omc * p1.z) - s1 * p2.z};
    }
  }
  V3 pos;
  Q4 quat;
  COL_T(1)
  tp_chain_pose(pos, quat, lp, lq, par_sa, base * 4, t.jump_rounds, isb);
  COL_T(2)
  // geom frames: lane = geom (chunks of 32), body pose fetched from the body's lane
  const tp_lds_d G = (tp_lds_d)gfr + (tid / TP_LPE) * (cm.ngeom * TPG_SLOT);
  for (int g0 = 0; g0 < cm.ngeom; g0 += TP_LPE) {
    const int g = g0 + l;
    const bool live = g < cm.ngeom;
    const int gb = live ? m.geom_bodyid[g] : 0;
    const int sa = (base + gb) * 4;
    const V3 bp = {bperm(pos.x, sa), bperm(pos.y, sa), bperm(pos.z, sa)};
    const Q4 bq = {bperm(quat.w, sa), bperm(quat.x, sa), bperm(quat.y, sa), bperm(quat.z, sa)};
    if (live) {
      const V3 p = mrot(q2m(bq), ld3(m.geom_pos + 3 * g));
      const M3 R = q2m(qmul(bq, ld4(m.geom_quat + 4 * g)));
      const double px = bp.x + p.x, py = bp.y + p.y, pz = bp.z + p.z;
      const tp_lds_d s = G + g * TPG_SLOT;
      s[0] = px; s[1] = py; s[2] = pz;
#pragma unroll
      for (int k = 0; k < 9; k++) s[3 + k] = R.a[k];
      if (env_ok && cd.gframe) {  // one record per geom for the persistent narrowphase (the lanes' records are adjacent)
        double* fr = cd.gframe + ((size_t)e * cm.ngeom + g) * 12;
        fr[0] = px; fr[1] = py; fr[2] = pz;
#pragma unroll
        for (int k = 0; k < 9; k++) fr[3 + k] = R.a[k];
      } else if (env_ok) {
        AT(cd.geom_xpos, 3 * g) = px; AT(cd.geom_xpos, 3 * g + 1) = py; AT(cd.geom_xpos, 3 * g + 2) = pz;
#pragma unroll
        for (int k = 0; k < 9; k++) AT(cd.geom_xmat, 9 * g + k) = R.a[k];
      }
    }
  }
  tp_wave_sync();
  COL_T(3)
  // broadphase over the candidate pairs, in pair order.  Pass 1: bounding spheres (plane: signed distance) for
  // every pair, survivors compacted IN ORDER into an LDS list; pass 2: the oriented-box filter over that list only
  // (a wavefront runs the fifteen-axis test when any of its lanes needs it: over all pairs that was every
  // trip, over the survivors it is one trip or two), then the ordered reservation of candidate slots.
  const tp_lds_i L = (tp_lds_i)((tp_lds_d)gfr + TP_ENVS * cm.ngeom * TPG_SLOT) + (tid / TP_LPE) * cm.npair;
  // per candidate slot: pair | sub-contact << 24 and geom1 | geom2 << 16, for the work-list items below
  const tp_lds_i PL = (tp_lds_i)((tp_lds_d)gfr + TP_ENVS * cm.ngeom * TPG_SLOT) + TP_ENVS * cm.npair + (tid / TP_LPE) * 2 * cm.maxcand;
  const tp_mask lower = tp_bit(l) - 1;
  int nlist = 0;
  for (int p0 = 0; p0 < cm.npair; p0 += TP_LPE) {
    const int p = p0 + l;
    bool keep = false, obb = false;
    if (p < cm.npair) {
      const int4 rec = *(const int4*)(cm.pair_bp + 4 * p);
      const int g1 = rec.x & 0xfff, g2 = (rec.x >> 12) & 0xfff, kind = (rec.x >> 24) & 3;
      const double thresh = __hiloint2double(rec.w, rec.z);
      const tp_lds_d s1 = G + g1 * TPG_SLOT, s2 = G + g2 * TPG_SLOT;
      const V3 x1 = {s1[0], s1[1], s1[2]}, x2 = {s2[0], s2[1], s2[2]};
      keep = true;
      if (kind == 1) {
        const V3 df = sub(x1, x2);
        keep = !(dot(df, df) > thresh);  // thresh = (margin + r1 + r2)^2: a conservative filter needs no square root (~30 instructions per pair trip)
        obb = keep && (rec.x >> 26 & 1);
      } else if (kind == 2) {
        const V3 n = {s1[3 + 2], s1[3 + 5], s1[3 + 8]};
        keep = !(dot(sub(x2, x1), n) > thresh);
      }
    }
    const tp_mask mk = tp_env_bits(__ballot(keep), base);
    if (keep) L[nlist + tp_popc(mk & lower)] = p | (obb ? 1 << 30 : 0);
    nlist += tp_popc(mk);
  }
  tp_wave_sync();
  int count = 0;
  for (int i0 = 0; i0 < nlist; i0 += TP_LPE) {
    const int i = i0 + l;
    bool keep = i < nlist;
    int p = 0;
    if (keep) {
      const int v = L[i];
      p = v & ~(1 << 30);
      if (v >> 30) {
        const int g1 = cm.pair_geom1[p], g2 = cm.pair_geom2[p];
        const double margin = fmax(cm.geom_margin[g1], cm.geom_margin[g2]);
        const tp_lds_d s1 = G + g1 * TPG_SLOT, s2 = G + g2 * TPG_SLOT;
        const V3 x1 = {s1[0], s1[1], s1[2]}, x2 = {s2[0], s2[1], s2[2]};
        double A[9], B[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { A[k] = s1[3 + k]; B[k] = s2[3 + k]; }
        keep = !obb_separated_m(cm.geom_obb + 6 * g1, cm.geom_obb + 6 * g2, A, B, x1, x2, margin);
      }
    }
    // a pair reserves pair_nsub consecutive slots (1, 2 or 8; 0 for geom types no routine handles):
    // ordered compaction with a prefix sum of the counts from three ballots
    const int cnt = keep ? cm.pair_nsub[p] : 0;
    const tp_mask m1 = tp_env_bits(__ballot(cnt & 1), base), m2 = tp_env_bits(__ballot(cnt & 2), base),
                  m8 = tp_env_bits(__ballot(cnt & 8), base);
    const int at = count + tp_popc(m1 & lower) + 2 * tp_popc(m2 & lower) + 8 * tp_popc(m8 & lower);
    if (env_ok && cnt) {
      const int gg = cm.pair_bp[4 * p];
      const int g12 = (gg & 0xfff) | ((gg >> 12 & 0xfff) << 16);
      for (int q = 0; q < cnt; q++)
        if (at + q < cm.maxcand) {
          if (cd.crec) cd.crec[(size_t)e * cm.maxcand + at + q].pair = p | (q << 24);
          else AT(cd.cand_pair, at + q) = p | (q << 24);
          PL[2 * (at + q)] = p | (q << 24); PL[2 * (at + q) + 1] = g12;
        }
    }
    count += tp_popc(m1) + 2 * tp_popc(m2) + 8 * tp_popc(m8);
  }
  if (l == 0 && env_ok) {
    if (count > cm.maxcand) AT(d.warning, 0) |= 8;
    AT(cd.ncand, 0) = count > cm.maxcand ? cm.maxcand : count;
  }
  COL_T(4)
  // work list for the narrowphase: the environment reserves `count` consecutive items of its sub-list
  if (count > cm.maxcand) count = cm.maxcand;
  const int sub = blockIdx.x & (BS_NSUB - 1);
  int off = 0;
  if (l == 0 && env_ok && count > 0)
    off = (int)atomicAdd(cd.stat + BS_STAT_ITEMS + d.range * BS_NSUB + sub, (unsigned long long)count);
  off = __builtin_amdgcn_ds_bpermute(base * 4, off);
  tp_wave_sync();
  if (env_ok) {
    // a sub-list holds every candidate of its environments, so the reservation fits whenever the counters
    // were cleared by the k_tpc pass of the previous step; the test keeps a missed clear (a failed launch)
    // from writing past the list — the narrowphase clamps the lengths the same way
    const long subcap = col_items_subcap(d.n_envs - d.env0, cm.maxcand);
    if (off + count <= subcap) {
      // an item names everything the narrowphase's set-up loads depend on — environment | slot, pair, both geoms — so
      // its pair record and its two geom frames are fetched together (one round trip after the item's, not three)
      int4* items = cd.items + col_items_base(d.env0, d.range, cm.maxcand) + (long)sub * subcap + off;
      const int tag = (int)(env - d.env0) << 8;
      for (int k = l; k < count; k += TP_LPE) items[k] = make_int4(tag | k, PL[2 * k], PL[2 * k + 1], 0);
    } else if (l == 0) {
      AT(d.warning, 0) |= 8;
    }
  }
}

// developer profiling (-DTPC_PROFILE): wave-clock ticks per section of k_tpc, summed into
// ColData::stat[8 + section] by lane 0 of every wavefront
// -DTPC_CUTS: the kernel returns at section `cut` (bits 8.. of `dump`, BS_CUT in the environment): the time of
// a section is the difference of two truncated runs, without timers inside the kernel
#if defined(TPC_PROFILE)
#define TPC_T(i) { const unsigned long long now_ = clock64(); tacc[i] += now_ - tlast; tlast = now_; }
#define TPC_TL(i) TPC_T(i)
#define TPC_C(i)
#elif defined(TPC_CUTS)
#define TPC_CLR { if constexpr (CON) { if (blockIdx.x == 0 && cd.stat) cd.stat[BS_STAT_ITEMS + d.range * BS_NSUB + threadIdx.x] = 0; } }
#define TPC_T(i) { if (cut == i) { TPC_CLR return; } }
#define TPC_TL(i) { if (cut == i) return 0.0; }
#define TPC_C(i) TPC_T(i)
#else
#define TPC_T(i)
#define TPC_TL(i)
#define TPC_C(i)
#endif

// ------------------------------------------------------------------ kernel body
// mode 0: nsteps x mj_step; mode 1: one mj_forward.  dump 0: state only, 1: + poses and
// generalized forces of the last forward pass, 2: every intermediate the parity tests read.
// CON: one step per launch with the contact candidates of k_c_broad / k_c_narrow (physics_collide.h):
// contact assembly, pyramid rows owned by lanes, dense Hessian per environment in LDS.
// QUAD (with CHAIN): the chain sums of hand-shaped models by DPP quad permutes (tp_chain_sum_quad).
// CG: mj_solCG instead of mj_solNewton (same primal loop, line search and stopping rule; the search
// direction is Polak-Ribiere in the M^-1 inner product, physics_dev.h::cg_gradient) — no Hessian.
// PGS: mj_solPGS (dual, projected Gauss-Seidel): M^-1 is formed column by
// column in LDS, the sweep visits the rows in efc order, the owner lane of a row computes its new
// force and broadcasts the change, every dof lane adds change * (M^-1 J^T)_i to its acceleration.
template <bool CON, bool FULL, bool RK4, bool CG = false, bool PGS = false, bool CHAIN = false, bool QUAD = false>
__device__ __forceinline__ void tp_run(const DevModel& m, const TpModel& t, const DevData& d, int nsteps,
                                       int mode, int dump, TpShared sh, const ColModel& cm, const ColData& cd) {
  int* tbl = sh.tbl;
  const int tid = threadIdx.x;
  constexpr int SL = (CON && PGS) ? TPC_SLOT_PGS : TPC_SLOT;  // contact slot stride
#ifdef TPC_CUTS
  const int cut = (dump >> 8) - 1;  // -1: the whole kernel
  dump &= 255;
#endif
#ifdef TPC_PROFILE
  unsigned long long tacc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  const int l = tid & (TP_LPE - 1), base = tid & ~(TP_LPE - 1);
  // Prologue (round 3): every load of the lane table, of the lane's model constants and of the state is issued
  // before the first wait.  Written the obvious way (`has_dof ? AT(d.qpos, j) : 0.0` per field, the LDS table built
  // first) the compiler emitted one guarded load per field, each behind its own s_waitcnt vmcnt(0) and a reload of
  // the kernel arguments: a dozen dependent memory round trips per wavefront with two wavefronts per SIMD to hide
  // them.  Indices are clamped instead (j = 0, act = 0 on lanes without one) and the values selected afterwards.
  // lane table: lane references become ds_bpermute byte addresses; child lists and ancestor
  // masks go to LDS, everything else straight into registers
  auto lane_ref = [&](int f) { const int v = t.lane[f * TP_LPE + l]; return v >= 0 ? (base + v) * 4 : v; };
  long env = d.env0 + (long)tp_block() * TP_ENVS + (tid / TP_LPE);
  const bool env_ok = env < d.n_envs;
  if (!env_ok) env = d.n_envs - 1;
  const long E = d.E, e = env;
  const int self_sa = tid * 4;
  const int par_sa = lane_ref(TPI_PARENT);
  const int depth = t.lane[TPI_DEPTH * TP_LPE + l], jnt = t.lane[TPI_JNT * TP_LPE + l];
  const int root = t.lane[TPI_ROOT * TP_LPE + l], act = t.lane[TPI_ACT * TP_LPE + l];
  const int eq_own = t.lane[TPI_EQ_OWN * TP_LPE + l], eq_pos = t.lane[TPI_EQ_POS * TP_LPE + l];
  const int eq_src_raw = t.lane[TPI_EQ_SRC * TP_LPE + l];
  const int dd_raw = t.lane[TPI_DD * TP_LPE + l];
  const int base4 = base * 4;
  const int eq_other_sa = base4 + t.lane[TPI_SA_EQO * TP_LPE + l];
  const int eq_src_sa = base4 + t.lane[TPI_SA_EQS * TP_LPE + l];
  const bool has_dof = jnt >= 0, is_src = eq_src_raw >= 0;
  const bool isb = l < m.nbody && l > 0;
  const int b = l < m.nbody ? l : 0, j = has_dof ? jnt : 0, ja = act >= 0 ? act : 0;
  // state and the lane's model constants (clamped indices: every lane loads)
  const double qpos_l = AT(d.qpos, j), qvel_l = AT(d.qvel, j), qfa_l = AT(d.qfrc_applied, j), qacc_l = AT(d.qacc, j);
  const double qws_l = AT(d.qacc_warmstart, j), ctrl_l = AT(d.ctrl, ja);
  const int jtype = m.jnt_type[j];
  const double mass_l = m.body_mass[b], qpos0 = m.qpos0[j], damping_l = m.dof_damping[j];
  double time = AT(d.time, 0);
  int warning = AT(d.warning, 0), solver_iter = AT(d.solver_iter, 0), nefc = AT(d.nefc, 0);
  int ncand_pre = 0;  // candidate count of the environment: requested here, needed at the contact assembly
  if constexpr (CON) ncand_pre = AT(cd.ncand, 0);
  TpCtx c;
  c.self_sa = self_sa; c.zero_sa = (base + TP_LPE - 1) * 4; c.dd = has_dof ? dd_raw : 0; c.maxdd = t.maxdd; c.tbl = (tp_lds_ptr)tbl;
  c.dchild0 = base4 + t.lane[TPI_SA_DC0 * TP_LPE + l];
  const int bchild0 = base4 + t.lane[TPI_SA_BC0 * TP_LPE + l];
#pragma unroll
  for (int a = 0; a < TP_MAXD; a++) c.anc_sa[a] = base4 + t.lane[(TPI_SA_ANC + a) * TP_LPE + l];
  c.top_sa = self_sa;
#pragma unroll
  for (int a = 1; a < TP_MAXD; a++)
    if (a == c.dd - 1) c.top_sa = c.anc_sa[a];
#pragma unroll
  for (int D = 0; D < TP_MAXD + 2; D++) c.nchild[D] = t.lvl_dchild[D];
  {
    // child words from the host (lane bytes relative to the environment's first lane; every byte gets the half's offset)
    const int bb = base * 0x01010101;
#pragma unroll
    for (int r = 0; r < 2 * TPL_CROWS; r++) tbl[r * 64 + tid] = t.lane[(TPI_CW + r) * TP_LPE + l] + bb;
  }
  tbl[TPL_ANCMASK * 64 + tid] = t.lane[TPI_ANCMASK * TP_LPE + l];
#if TP_LPE == 64
  tbl[TPL_ANCMASK_HI * 64 + tid] = t.lane[TPI_ANCMASK_HI * TP_LPE + l];
#endif
  __syncthreads();
  c.dchild_w = tbl[TPL_DCHILD * 64 + tid];  // (the CHAIN kernels keep it for the whole launch)

  // per-lane model constants
  const double* const lk = t.lane_k + l * TPK_N;  // the lane's record of per-step constants
  const double mass = isb ? mass_l : 0.0;
  const double damping = has_dof ? damping_l : 0.0;
  const double h = m.timestep;
  const double scale = tp_rcp(m.meaninertia * (m.nv > 1 ? m.nv : 1));

  // state
  double qpos = has_dof ? qpos_l : 0.0, qvel = has_dof ? qvel_l : 0.0;
  double qfrc_applied = has_dof ? qfa_l : 0.0, qacc = has_dof ? qacc_l : 0.0;
  double qacc_ws = has_dof ? qws_l : 0.0, ctrl = act >= 0 ? ctrl_l : 0.0;
  bool was_reset = false;
  auto reset_state = [&]() {
    qpos = has_dof ? qpos0 : 0.0; qvel = 0; ctrl = 0; qfrc_applied = 0; qacc = 0; qacc_ws = 0; time = 0;
    solver_iter = 0; nefc = 0; was_reset = true;
  };

  TPC_T(0)
  const int nstage = (RK4 && mode == 0) ? 4 : 1;  // RK4: the host picks the instantiation from opt.integrator
  for (int step = 0; step < nsteps; step++) {
    if (mode == 0) {  // A14 mj_checkPos / mj_checkVel
      if (env_any(has_dof && bad_num(qpos), base)) { reset_state(); warning |= 1; }
      if (env_any(has_dof && bad_num(qvel), base)) { reset_state(); warning |= 2; }
    }
    TPC_T(18)
    double rk_q0 = qpos, rk_v0 = qvel, rk_t0 = time, rk_dq = 0, rk_dv = 0;
    bool redone = false;
    for (int stage = 0; stage < nstage; stage++) {
      // ================= forward pass (single inlined site) =================
      // ---- A1 kinematics: joint-local transform first (no parent needed), then one
      // quaternion/vector compose per tree level
      // pose constants of the lane's body and joint are re-read every step (cache hits) rather
      // than held in 26 registers across the whole kernel
      const V3 bpos = ld3(lk + TPK_BPOS), jpos = ld3(lk + TPK_JPOS), jax = ld3(lk + TPK_JAX);
      const Q4 bquat = ld4(lk + TPK_BQUAT);
      const double dq = qpos - qpos0;
      Q4 lq = bquat;
      V3 lp = bpos;
      if (has_dof) {
        // closed form of bquat * (cos h, k sin h) and bpos + R_b (jpos - R(k, dq) jpos), h = dq / 2 (TPK_P1 .. TPK_Q1)
        const V3 p2 = ld3(lk + TPK_P2);
        if (jtype == BS_JNT_SLIDE) {
          lp = {bpos.x + p2.x * dq, bpos.y + p2.y * dq, bpos.z + p2.z * dq};
        } else {
          const TpSinCos sc = tp_sincos(0.5 * dq);
          const double sn = sc.sn, cs = sc.cs;
          const V3 p1 = ld3(lk + TPK_P1);
          const Q4 q1 = ld4(lk + TPK_Q1);
          const double omc = 2.0 * (sn * sn), s1 = 2.0 * (sn * cs);  // 1 - cos dq, sin dq
          lq = {cs * bquat.w + sn * q1.w, cs * bquat.x + sn * q1.x, cs * bquat.y + sn * q1.y, cs * bquat.z + sn * q1.z};
          lp = {(bpos.x + omc * p1.x) - s1 * p2.x, (bpos.y + omc * p1.y) - s1 * p2.y, (bpos.z + omc * p1.z) - s1 * p2.z};
        }
      }
      TPC_T(19)
      V3 pos;
      Q4 quat;
      tp_chain_pose(pos, quat, lp, lq, par_sa, base * 4, t.jump_rounds, isb);
      if (!FULL && stage == 0 && step == nsteps - 1 && env_ok && dump >= 1 && l < m.nbody) {
        // body poses are outputs of the step: stored here so they need not stay live
        AT(d.xpos, 3 * b) = pos.x; AT(d.xpos, 3 * b + 1) = pos.y; AT(d.xpos, 3 * b + 2) = pos.z;
        AT(d.xquat, 4 * b) = quat.w; AT(d.xquat, 4 * b + 1) = quat.x; AT(d.xquat, 4 * b + 2) = quat.y;
        AT(d.xquat, 4 * b + 3) = quat.z;
      }
      const M3 xmat = q2m(quat);
      V3 axis = mrot(xmat, jax), anchor;
      {
        const V3 a = mrot(xmat, jpos);
        anchor = {pos.x + a.x, pos.y + a.y, pos.z + a.z};
        if (jtype == BS_JNT_SLIDE) { anchor.x -= axis.x * dq; anchor.y -= axis.y * dq; anchor.z -= axis.z * dq; }
      }
      TPC_T(12)
      // ---- A2 comPos
      V3 xipos;
      {
        const V3 ip = mrot(xmat, ld3(lk + TPK_IPOS));
        xipos = {pos.x + ip.x, pos.y + ip.y, pos.z + ip.z};
      }
      V3 com = {0, 0, 0};
      if constexpr (!FULL) {
        // Reference point of the tree's spatial quantities.  mj_comPos takes the subtree centre of mass of the body's
        // root; the mass matrix, the bias forces and the contact Jacobians do not depend on the point (only cinert /
        // cdof / cvel do, and those are outputs of the full-dump kernels alone, which keep the centre of mass): the
        // step kernels take the root body's own inertial position — one fetch, where the centre of mass is three
        // environment-wide sums in front of everything that follows.  Results differ by rounding.
        const int ra = isb ? (base + root) * 4 : self_sa;
        const double rx = bperm(xipos.x, ra), ry = bperm(xipos.y, ra), rz = bperm(xipos.z, ra);
        if (isb) com = {rx, ry, rz};
      } else
      for (int r = 0; r < t.nroots; r++) {
        const int rb = t.roots[r];
        const bool mine = isb && root == rb;
        const double sx = env_sum(mine ? mass * xipos.x : 0.0), sy = env_sum(mine ? mass * xipos.y : 0.0),
                     sz = env_sum(mine ? mass * xipos.z : 0.0);
        const double sm = m.body_subtreemass[rb];
        const int ra = (base + rb) * 4;
        const double rx = bperm(xipos.x, ra), ry = bperm(xipos.y, ra), rz = bperm(xipos.z, ra);
        if (mine) {
          if (sm < BS_MINVAL) com = {rx, ry, rz};
          else { const double ism = tp_rcp(sm); com = {sx * ism, sy * ism, sz * ism}; }
        }
      }
      double cinert[10], cdof[6];
      {
        // xmat Ib xmat^T with Ib = R_i diag(I) R_i^T built on the host (TPK_IB: 00 11 22 01 02 12): 45 instructions where
        // composing the inertial frame's quaternion, its matrix and the diagonal product took 65
        const M3& R = xmat;
        const double* I = lk + TPK_IB;
        const double off[3] = {xipos.x - com.x, xipos.y - com.y, xipos.z - com.z};
        double tt[9];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
          const double a = R.a[3 * rr], bb = R.a[3 * rr + 1], cc = R.a[3 * rr + 2];
          tt[3 * rr] = a * I[0] + bb * I[3] + cc * I[4];
          tt[3 * rr + 1] = a * I[3] + bb * I[1] + cc * I[5];
          tt[3 * rr + 2] = a * I[4] + bb * I[5] + cc * I[2];
        }
        cinert[0] = tt[0] * R.a[0] + tt[1] * R.a[1] + tt[2] * R.a[2];
        cinert[1] = tt[3] * R.a[3] + tt[4] * R.a[4] + tt[5] * R.a[5];
        cinert[2] = tt[6] * R.a[6] + tt[7] * R.a[7] + tt[8] * R.a[8];
        cinert[3] = tt[0] * R.a[3] + tt[1] * R.a[4] + tt[2] * R.a[5];
        cinert[4] = tt[0] * R.a[6] + tt[1] * R.a[7] + tt[2] * R.a[8];
        cinert[5] = tt[3] * R.a[6] + tt[4] * R.a[7] + tt[5] * R.a[8];
        cinert[0] += mass * (off[1] * off[1] + off[2] * off[2]);
        cinert[1] += mass * (off[0] * off[0] + off[2] * off[2]);
        cinert[2] += mass * (off[0] * off[0] + off[1] * off[1]);
        cinert[3] -= mass * off[0] * off[1];
        cinert[4] -= mass * off[0] * off[2];
        cinert[5] -= mass * off[1] * off[2];
        cinert[6] = mass * off[0]; cinert[7] = mass * off[1]; cinert[8] = mass * off[2]; cinert[9] = mass;
        if (!isb)
#pragma unroll
          for (int k = 0; k < 10; k++) cinert[k] = 0;
      }
      if (has_dof && jtype == BS_JNT_SLIDE) {
        cdof[0] = cdof[1] = cdof[2] = 0;
        cdof[3] = axis.x; cdof[4] = axis.y; cdof[5] = axis.z;
      } else if (has_dof) {
        const V3 cc = cross(axis, V3{com.x - anchor.x, com.y - anchor.y, com.z - anchor.z});
        cdof[0] = axis.x; cdof[1] = axis.y; cdof[2] = axis.z;
        cdof[3] = cc.x; cdof[4] = cc.y; cdof[5] = cc.z;
      } else {
#pragma unroll
        for (int k = 0; k < 6; k++) cdof[k] = 0;
      }
      TPC_T(13)
      // ---- A8 comVel + rne body forces first, so that their leaves-to-root fold shares the level loop
      // (and its cross-lane round trips) with the crb fold of A3
      double cvel[6], qfrc_bias, f[6];
      {
        // cvel = sum of cdof * qvel over the body's chain; cacc = -gravity + sum over the chain of
        // (parent's cvel x cdof) * qvel: two chain sums and one parent fetch instead of one pass per level
        double acc[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cvel[k] = isb ? cdof[k] * qvel : 0.0;
        if constexpr (CHAIN && QUAD) tp_chain_sum_quad<6>(cvel, (base + t.quad_top) * 4, l, t.quad_top, m.nbody);
        else tp_chain_sum<6>(cvel, par_sa, t.jump_rounds, isb);
        {
          // (the body's own cvel stands in for its parent's: they differ by cdof * qvel, and the motion cross product of
          // cdof with itself is zero — one cross-lane round trip less in the chain between the two sums)
          double cdd[6];
          cross_motion(cdd, cvel, cdof);
#pragma unroll
          for (int k = 0; k < 6; k++) acc[k] = isb ? cdd[k] * qvel : 0.0;
        }
        if constexpr (CHAIN && QUAD) tp_chain_sum_quad<6>(acc, (base + t.quad_top) * 4, l, t.quad_top, m.nbody);
        else tp_chain_sum<6>(acc, par_sa, t.jump_rounds, isb);
        if (isb && !(m.disableflags & BS_DSBL_GRAVITY)) {
          acc[3] -= m.gravity[0]; acc[4] -= m.gravity[1]; acc[5] -= m.gravity[2];
        }
        double t0[6], t1[6], t2[6];
        mul_inert_vec(t0, cinert, acc);
        mul_inert_vec(t1, cinert, cvel);
        cross_force(t2, cvel, t1);
#pragma unroll
        for (int k = 0; k < 6; k++) f[k] = isb ? t0[k] + t2[k] : 0.0;
      }
      // ---- A3 crb (children folded into parents, deepest level first) + mass-matrix rows
      TpRow Mr;
      {
        double crb[10];
        double palm_m = 0, palm_b = 0;
        bool palm_top = false;
#pragma unroll
        for (int k = 0; k < 10; k++) crb[k] = cinert[k];
        for (int lvl = t.maxdepth; lvl >= (FULL ? 2 : t.fold_minlvl); lvl--) {
          const bool par = isb && depth == lvl - 1;
          if (t.lvl_parent[lvl] >= 0) {
            // all bodies of this level hang off one body (a palm and its fingers): their sum is a reduction over the
            // environment's lanes (DPP, VALU rate) instead of one ds_bpermute round per child — the contact-free
            // kernels are bound by the CU's LDS pipe in this section (the contact kernels neither gain nor lose).
            const bool kid = isb && depth == lvl, top = l == t.lvl_parent[lvl];
            if (!FULL && t.palm_scalar && lvl == t.fold_minlvl) {
              // Round 3: the last fold of a hand goes into the palm, whose composite inertia and force are used for
              // exactly two numbers — the palm dof's diagonal entry cdof . (I_c cdof) and its bias force cdof . f — and
              // both are LINEAR in what is folded: every child projects its own composite on the palm's motion axis and
              // two scalars are summed over the environment, where the fold summed sixteen values (a fifteenth of the
              // contact-free kernel's instructions, and sixteen dependent reductions).  Summation order differs from the
              // fold's by rounding; every instantiation shares this code.
              const int top_sa = (base + t.lvl_parent[lvl]) * 4;
              double cp[6], tb[6];
#pragma unroll
              for (int k = 0; k < 6; k++) cp[k] = bperm(cdof[k], top_sa);
              // (spelled in fused multiply-adds of a fixed order: left to the compiler's contraction the two
              // instantiations that are compared bit for bit — chain and generic — rounded these sums differently)
              {
                const double* I = crb;
                const double* v = cp;
                tb[0] = __builtin_fma(I[7], v[5], __builtin_fma(-I[8], v[4], __builtin_fma(I[4], v[2], __builtin_fma(I[3], v[1], I[0] * v[0]))));
                tb[1] = __builtin_fma(-I[6], v[5], __builtin_fma(I[8], v[3], __builtin_fma(I[5], v[2], __builtin_fma(I[1], v[1], I[3] * v[0]))));
                tb[2] = __builtin_fma(I[6], v[4], __builtin_fma(-I[7], v[3], __builtin_fma(I[2], v[2], __builtin_fma(I[5], v[1], I[4] * v[0]))));
                tb[3] = __builtin_fma(I[9], v[3], __builtin_fma(-I[7], v[2], I[8] * v[1]));
                tb[4] = __builtin_fma(I[9], v[4], __builtin_fma(-I[8], v[0], I[6] * v[2]));
                tb[5] = __builtin_fma(I[9], v[5], __builtin_fma(-I[6], v[1], I[7] * v[0]));
              }
              double s1 = cp[0] * tb[0], s2 = cp[0] * f[0];
#pragma unroll
              for (int k = 1; k < 6; k++) { s1 = __builtin_fma(cp[k], tb[k], s1); s2 = __builtin_fma(cp[k], f[k], s2); }
              const bool sel = kid || top;
              palm_m = env_sum(sel ? s1 : 0.0);
              palm_b = env_sum(sel ? s2 : 0.0);
              palm_top = top;
              continue;
            }
#pragma unroll
            for (int k = 0; k < 10; k++) { const double sk = env_sum(kid ? crb[k] : 0.0); if (top) crb[k] += sk; }
#pragma unroll
            for (int k = 0; k < 6; k++) { const double sk = env_sum(kid ? f[k] : 0.0); if (top) f[k] += sk; }
            continue;
          }
          if (t.lvl_next[lvl]) {
            // every body of this level sits on the lane after its parent's (single-child chains numbered depth-first):
            // the parent takes its child's values with a wavefront shift (DPP wave_shl:1), no LDS-pipe instruction
            // (taken through the lane's 0 / 1 factor, tp_add: the other lanes' values are finite)
            const double rm = par && bchild0 == self_sa + 4 ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 10; k++) crb[k] = tp_add(crb[k], dpp_mov<0x130>(crb[k]), rm, true);
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] = tp_add(f[k], dpp_mov<0x130>(f[k]), rm, true);
            continue;
          }
          for (int ch = 0; ch < t.lvl_bchild[lvl]; ch++) {
            const int src = ch == 0 ? bchild0 : tp_child(c.tbl, TPL_BCHILD, ch);
            const int sa = par ? src : c.zero_sa;
#pragma unroll
            for (int k = 0; k < 10; k++) crb[k] += bperm(crb[k], sa);
#pragma unroll
            for (int k = 0; k < 6; k++) f[k] += bperm(f[k], sa);
          }
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += cdof[k] * f[k];
        qfrc_bias = palm_top ? palm_b : s;
        double buf[6];
        mul_inert_vec(buf, crb, cdof);
        double s0 = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s0 += cdof[k] * buf[k];
        Mr.r[0] = has_dof ? lk[TPK_ARMATURE] + (palm_top ? palm_m : s0) : 1.0;
        double cds[6], cdtop[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { cds[k] = cdof[k]; cdtop[k] = CHAIN ? bperm(cdof[k], c.top_sa) : 0.0; }
        double sjtop = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) sjtop += cdtop[k] * buf[k];
#pragma unroll
        for (int a = 1; a < TP_MAXD; a++) {
          Mr.r[a] = 0;
          if (a >= t.maxdd) continue;
          double sj = 0;
          if constexpr (CHAIN) {
            // the a-th ancestor's motion axis: a lanes below (shifted once more per a), or the top dof's (fetched once;
            // its product sjtop is formed once and selected per a)
#pragma unroll
            for (int k = 0; k < 6; k++) {
              cds[k] = dpp_mov<0x138>(cds[k]);
              sj += cds[k] * buf[k];
            }
            if (a == c.dd - 1) sj = sjtop;
          } else {
#pragma unroll
            for (int k = 0; k < 6; k++) sj += bperm(cdof[k], c.anc_sa[a]) * buf[k];
          }
          if (a < c.dd) Mr.r[a] = sj;
        }
      }
      TPC_T(14)
      // ---- A4 factor M
      // with Euler's implicit damping the step needs the factors of M and of M + h * damping: both
      // here in one pass when there is LDS to park the second one until the integrator
      const bool euler_early = !RK4 && !FULL && sh.park && m.damped && mode == 0;
      TpFac LM;
      if (euler_early) {
        TpRow He = Mr;
        He.r[0] += h * damping;
        TpFac LE;
        tp_factor2<CHAIN>(c, Mr, He, LM, LE);
#pragma unroll
        for (int a = 0; a < TP_MAXD; a++) sh.park[a * 64 + tid] = LE.R.r[a];
        sh.park[TP_MAXD * 64 + tid] = LE.dinv;
      } else {
        LM = tp_factor<CHAIN>(c, Mr);
      }
      TPC_T(15)
      TPC_T(16)
      // ---- passive + A9 actuation + A10 smooth acceleration
      double qfrc_passive = 0, act_force = 0, qfrc_actuator = 0;
      {
        // joint and actuator constants from the per-lane table: one round of loads, no dependent branches
        const double* pc = lk + TPK_D0;
        if (has_dof && !(m.disableflags & BS_DSBL_PASSIVE))
          qfrc_passive = -pc[TPD_STIFF] * (qpos - pc[TPD_SPRING]) - damping * qvel;
        if (act >= 0 && !(m.disableflags & BS_DSBL_ACTUATION)) {
          const double gear = pc[TPD_ACT_GEAR];
          const double cc = fmin(fmax(ctrl, pc[TPD_ACT_CLO]), pc[TPD_ACT_CHI]);
          double f = pc[TPD_ACT_GAIN] * cc + pc[TPD_ACT_B0] + pc[TPD_ACT_B1] * (gear * qpos) +
                     pc[TPD_ACT_B2] * (gear * qvel);
          f = fmin(fmax(f, pc[TPD_ACT_FLO]), pc[TPD_ACT_FHI]);
          act_force = f;
          qfrc_actuator = gear * f;
        }
      }
      const double qfrc_smooth = has_dof ? qfrc_passive - qfrc_bias + qfrc_applied + qfrc_actuator : 0.0;
      // M * qacc_warmstart (needed by the primal solvers' warm start) rides along in the solve's sweeps
      double Ma_ws = 0;
      const bool warm = !PGS && !(m.disableflags & BS_DSBL_WARMSTART) && !(m.disableflags & BS_DSBL_CONSTRAINT);
      const double qacc_smooth = warm ? tp_solve_mulm<CHAIN>(c, LM.R, LM.dinv, qfrc_smooth, Mr, qacc_ws, Ma_ws)
                                      : tp_solve<CHAIN>(c, LM.R, LM.dinv, qfrc_smooth);
      const bool dump_now = stage == 0 && step == nsteps - 1 && env_ok && dump >= 1;
      if (dump_now) {
        // outputs of the step's mj_forward, stored as soon as they exist so they are not kept
        // alive through the solver (a mj_checkAcc redo simply stores them again)
        if (has_dof) {
          AT(d.qfrc_bias, j) = qfrc_bias; AT(d.qfrc_passive, j) = qfrc_passive;
          AT(d.qfrc_actuator, j) = qfrc_actuator; AT(d.qfrc_smooth, j) = qfrc_smooth;
          AT(d.qacc_smooth, j) = qacc_smooth;
        }
        if (act >= 0) AT(d.actuator_force, act) = act_force;
      }

      TPC_T(1)
      // ---- contacts (k_tpc): ordered hit compaction (same arithmetic as col_assemble) and the
      // pyramid rows of the contact this lane owns (same arithmetic as make_contact_rows)
      const bool con_on = !(m.disableflags & BS_DSBL_CONSTRAINT);
      double caref[4] = {0, 0, 0, 0}, cmu = 0, cDrow = 0;
      int cnr = 0, b1_sa = self_sa, b2_sa = self_sa, ncon_env = 0;
      V3 off1 = {0, 0, 0}, off2 = {0, 0, 0};
      tp_lds_d slot0 = nullptr, myslot = nullptr, Hs = nullptr, Us = nullptr;
      // u3 = (normal, tangent1, tangent2) components of J_contact * x for the lane's contact:
      // x is pushed down the body tree like a velocity (sum of cdof * x over the chain), then the
      // owner lane reads the spatial vectors of its two bodies and projects their difference
      auto con_jx = [&](double x, double* u3) {
        if constexpr (CON) {
          double cv[6];
#pragma unroll
          for (int k = 0; k < 6; k++) cv[k] = isb ? cdof[k] * x : 0.0;
          if constexpr (CHAIN && QUAD) tp_chain_sum_quad<6>(cv, (base + t.quad_top) * 4, l, t.quad_top, m.nbody);
        else tp_chain_sum<6>(cv, par_sa, t.jump_rounds, isb);
          double v1[6], v2[6];
#pragma unroll
          for (int k = 0; k < 6; k++) { v1[k] = bperm(cv[k], b1_sa); v2[k] = bperm(cv[k], b2_sa); }
          u3[0] = u3[1] = u3[2] = 0;
          if (cnr > 0) {
            const V3 w1 = cross(V3{v1[0], v1[1], v1[2]}, off1), w2 = cross(V3{v2[0], v2[1], v2[2]}, off2);
            const V3 rv = {(v2[3] + w2.x) - (v1[3] + w1.x), (v2[4] + w2.y) - (v1[4] + w1.y),
                           (v2[5] + w2.z) - (v1[5] + w1.z)};
            const tp_lds_d f = myslot + CG_FRAME;
            u3[0] = f[0] * rv.x + f[1] * rv.y + f[2] * rv.z;
            u3[1] = f[3] * rv.x + f[4] * rv.y + f[5] * rv.z;
            u3[2] = f[6] * rv.x + f[7] * rv.y + f[8] * rv.z;
          }
        }
      };
      // the same for two vectors at once: one trip down the tree carries both
      auto con_jx2 = [&](double x0, double x1, double* u3a, double* u3b) {
        if constexpr (CON) {
          double cv[12];
#pragma unroll
          for (int k = 0; k < 6; k++) { cv[k] = isb ? cdof[k] * x0 : 0.0; cv[6 + k] = isb ? cdof[k] * x1 : 0.0; }
          if constexpr (CHAIN && QUAD) tp_chain_sum_quad<12>(cv, (base + t.quad_top) * 4, l, t.quad_top, m.nbody);
        else tp_chain_sum<12>(cv, par_sa, t.jump_rounds, isb);
#pragma unroll
          for (int h2 = 0; h2 < 2; h2++) {
            double v1[6], v2[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { v1[k] = bperm(cv[6 * h2 + k], b1_sa); v2[k] = bperm(cv[6 * h2 + k], b2_sa); }
            double* u3 = h2 ? u3b : u3a;
            u3[0] = u3[1] = u3[2] = 0;
            if (cnr > 0) {
              const V3 w1 = cross(V3{v1[0], v1[1], v1[2]}, off1), w2 = cross(V3{v2[0], v2[1], v2[2]}, off2);
              const V3 rv = {(v2[3] + w2.x) - (v1[3] + w1.x), (v2[4] + w2.y) - (v1[4] + w1.y),
                             (v2[5] + w2.z) - (v1[5] + w1.z)};
              const tp_lds_d f = myslot + CG_FRAME;
              u3[0] = f[0] * rv.x + f[1] * rv.y + f[2] * rv.z;
              u3[1] = f[3] * rv.x + f[4] * rv.y + f[5] * rv.z;
              u3[2] = f[6] * rv.x + f[7] * rv.y + f[8] * rv.z;
            }
          }
        }
      };
      // SUB: subtract aref (residuals J a - aref) or not (J x).  A compile-time flag on purpose: written as
      // `- sub * caref` with sub = 0.0 the products cannot be folded away (0 * x is not 0 for every x), so aref of
      // the fixed rows and of the contact stayed live through the whole Newton loop — eight doubles per lane, reloaded
      // from scratch in every iteration to be multiplied by zero (round 3)
      auto con_rows = [&](const double* u3, auto SUB, double* out) {
        constexpr bool sub = decltype(SUB)::value;
        out[0] = cnr > 0 ? (sub ? (u3[0] + cmu * u3[1]) - caref[0] : (u3[0] + cmu * u3[1])) : 0.0;
        out[1] = cnr > 1 ? (sub ? (u3[0] - cmu * u3[1]) - caref[1] : (u3[0] - cmu * u3[1])) : 0.0;
        out[2] = cnr > 1 ? (sub ? (u3[0] + cmu * u3[2]) - caref[2] : (u3[0] + cmu * u3[2])) : 0.0;
        out[3] = cnr > 1 ? (sub ? (u3[0] - cmu * u3[2]) - caref[3] : (u3[0] - cmu * u3[2])) : 0.0;
      };
      constexpr std::true_type WITH_AREF{};
      constexpr std::false_type NO_AREF{};
      if constexpr (CON) {
        const int half = tid / TP_LPE;
        slot0 = (tp_lds_d)sh.cg + half * (TP_LPE * SL);
        myslot = slot0 + l * SL;
        Hs = (tp_lds_d)sh.H + half * (m.nbody * (m.nbody + 1) / 2);
        Us = (tp_lds_d)sh.U + half * (TP_LPE * 3);
        const int ncand = ncand_pre;
        int count = 0;
        for (int s0 = 0; s0 < cm.maxcand; s0 += TP_LPE) {
          if (!__ballot(s0 < ncand)) break;
          const int sidx = s0 + l;
          const CandRec* cr = cd.crec ? cd.crec + (size_t)e * cm.maxcand + (sidx < ncand ? sidx : 0) : nullptr;
          // the whole 64-byte record in one round of loads (what a slot without a hit holds besides its flag is stale
          // and unused): flag first, then the fields of the hits, was two dependent round trips
          int rec_pair = 0, rec_hit = 0;
          double rec_d[7] = {0, 0, 0, 0, 0, 0, 0};
          if (cr) {
            rec_pair = cr->pair; rec_hit = cr->hit; rec_d[0] = cr->dist;
#pragma unroll
            for (int k = 0; k < 3; k++) { rec_d[1 + k] = cr->pos[k]; rec_d[4 + k] = cr->nrm[k]; }
          }
          const bool hit = sidx < ncand && (cr ? rec_hit : AT(cd.cand_hit, sidx)) != 0;
          const tp_mask bits = tp_env_bits(__ballot(hit), base);
          const int n = count + tp_popc(bits & (tp_bit(l) - 1));
          if (hit && n < cm.maxcon) {
            const int pr = (cr ? rec_pair : AT(cd.cand_pair, sidx)) & BS_PAIR_MASK;
            const double dist = cr ? rec_d[0] : AT(cd.cand_dist, sidx);
            const V3 p = cr ? V3{rec_d[1], rec_d[2], rec_d[3]}
                            : V3{AT(cd.cand_pos, 3 * sidx), AT(cd.cand_pos, 3 * sidx + 1), AT(cd.cand_pos, 3 * sidx + 2)};
            const V3 x = normalized(cr ? V3{rec_d[4], rec_d[5], rec_d[6]}
                                       : V3{AT(cd.cand_nrm, 3 * sidx), AT(cd.cand_nrm, 3 * sidx + 1), AT(cd.cand_nrm, 3 * sidx + 2)});
            V3 y = {0, 0, 0};
            if (x.y < 0.5 && x.y > -0.5) y.y = 1;
            else y.z = 1;
            const double tt = dot(x, y);
            y = normalized(V3{y.x - tt * x.x, y.y - tt * x.y, y.z - tt * x.z});
            const V3 z = cross(x, y);
            const double fr[9] = {x.x, x.y, x.z, y.x, y.y, y.z, z.x, z.y, z.z};
            const tp_lds_d sl = slot0 + n * SL;
            sl[CG_POS] = p.x; sl[CG_POS + 1] = p.y; sl[CG_POS + 2] = p.z;
#pragma unroll
            for (int k = 0; k < 9; k++) sl[CG_FRAME + k] = fr[k];
            sl[CG_DIST] = dist;
            ((tp_lds_i)(sl + CG_PAIR))[0] = pr;
            if (env_ok && dump >= 1) {  // the contact list is an output: written by the last pass of a launch sequence only
              AT(cd.con_pair, n) = pr; AT(cd.con_dist, n) = dist;
              AT(cd.con_pos, 3 * n) = p.x; AT(cd.con_pos, 3 * n + 1) = p.y; AT(cd.con_pos, 3 * n + 2) = p.z;
#pragma unroll
              for (int k = 0; k < 9; k++) AT(cd.con_frame, 9 * n + k) = fr[k];
            }
          }
          count += tp_popc(bits);
        }
        if (count > cm.maxcon) { count = cm.maxcon; warning |= 8; }
        ncon_env = count;
        if (env_ok && l == 0) {
          AT(cd.ncon, 0) = count;
          if (cd.stat && count) atomicAdd(cd.stat + BS_STAT_CON + (blockIdx.x & (BS_NSUB - 1)), (unsigned long long)count);
        }
        tp_wave_sync();
        double cB = 0, ckimp = 0;
        V3 cpos = {0, 0, 0};
        if (l < count && con_on) {
          const int pr = ((tp_lds_i)(myslot + CG_PAIR))[0];
          const int bb = cm.pair_body[2 * pr], b2 = cm.pair_body[2 * pr + 1];
          const int b1 = bb & 0xffff, dim = bb >> 16;
          const double* pc = cm.pair_con + (size_t)BS_PAIR_CON * pr;  // the pair's state-independent row constants
          b1_sa = (base + b1) * 4; b2_sa = (base + b2) * 4;
          tp_slot_mask_set(myslot, 0, tp_anc_mask(tbl, base + b1));
          tp_slot_mask_set(myslot, 1, tp_anc_mask(tbl, base + b2));
          cpos = {myslot[CG_POS], myslot[CG_POS + 1], myslot[CG_POS + 2]};
          const double margin = pc[0];
          const double dist = myslot[CG_DIST];
          if (dist < margin) {
            const double mu = pc[1], K = pc[2], B = pc[3];
            const double imp = tp_impedance(pc + 4, dist, margin);
            double R = fmax(BS_MINVAL, (1 - imp) * pc[9] * tp_rcp(imp));
            if (dim > 1) R = fmax(BS_MINVAL, pc[10] * R);
            cnr = dim == 1 ? 1 : 4; cmu = dim == 1 ? 0.0 : mu; cDrow = tp_rcp(R);
            myslot[CG_D] = cDrow; myslot[CG_MU] = cmu;
            cB = B; ckimp = K * imp * (dist - margin);
          }
        }
        {  // converged code: every lane takes part in the permutes
          const V3 c1 = {bperm(com.x, b1_sa), bperm(com.y, b1_sa), bperm(com.z, b1_sa)};
          const V3 c2 = {bperm(com.x, b2_sa), bperm(com.y, b2_sa), bperm(com.z, b2_sa)};
          off1 = {cpos.x - c1.x, cpos.y - c1.y, cpos.z - c1.z};
          off2 = {cpos.x - c2.x, cpos.y - c2.y, cpos.z - c2.z};
        }
        double u3[3], vel[4];
        con_jx(qvel, u3);
        con_rows(u3, NO_AREF, vel);
#pragma unroll
        for (int q = 0; q < 4; q++) caref[q] = q < cnr ? -cB * vel[q] - ckimp : 0.0;
        if constexpr (PGS) {
          if (l < ncon_env) {
            ((tp_lds_i)(myslot + CG_PAIR))[1] = cnr;  // rows of this contact (0: beyond its margin)
#pragma unroll
            for (int q = 0; q < 4; q++) { myslot[CG_AREF + q] = caref[q]; myslot[CG_F + q] = 0.0; }
          }
        }
        tp_wave_sync();
      }

      TPC_T(2)
      // ---- A7 constraint rows
      TpRows r;
      r.Df = r.rf = r.aref_f = r.fl = 0; r.Dl0 = r.Dl1 = r.aref_l0 = r.aref_l1 = 0;
      r.De = r.aref_e = r.ce = 0; r.De_o = r.ce_o = 0; r.own = false; r.cD = cDrow;
      int nact = 0;
      const double* rc = lk + TPK_D0;  // state-independent parts of the rows (host, build_tp)
      if (con_on && has_dof) {
        const double iw = rc[TPD_IW], Df = rc[TPD_FRIC_D], K = rc[TPD_LIM_K];
        if (Df > 0) {
          r.Df = Df; r.fl = rc[TPD_FRIC_FL]; r.rf = rc[TPD_FRIC_RF];
          r.aref_f = -rc[TPD_FRIC_B] * qvel;
          nact++;
        }
        if (K >= 0) {
          const double margin = lk[TPK_JMARGIN], B = rc[TPD_LIM_B];
          const double d0 = qpos - lk[TPK_JRANGE], d1 = lk[TPK_JRANGE + 1] - qpos;
          if (d0 < margin) {
            const double imp = tp_impedance(lk + TPK_JSOLIMP, d0, margin);
            r.Dl0 = tp_rcp(fmax(BS_MINVAL, (1 - imp) * iw * tp_rcp(imp)));
            r.aref_l0 = -B * qvel - K * imp * (d0 - margin);
            nact++;
          }
          if (d1 < margin) {
            const double imp = tp_impedance(lk + TPK_JSOLIMP, d1, margin);
            r.Dl1 = tp_rcp(fmax(BS_MINVAL, (1 - imp) * iw * tp_rcp(imp)));
            r.aref_l1 = B * qvel - K * imp * (d1 - margin);
            nact++;
          }
        }
      }
      {
        // equality rows live on joint1's lane and need qpos/qvel of joint2 (an ancestor dof)
        const double qo = bperm(qpos, eq_other_sa), vo = bperm(qvel, eq_other_sa);
        if (con_on && eq_own >= 0) {
          const double* cf = lk + TPK_EQDATA;
          const double x = qo - lk[TPK_EQQPOS0];
          const double pos_e = (qpos - qpos0) - (cf[0] + x * (cf[1] + x * (cf[2] + x * (cf[3] + x * cf[4]))));
          const double deriv = cf[1] + x * (2 * cf[2] + x * (3 * cf[3] + x * 4 * cf[4]));
          const double diag = rc[TPD_EQ_DIAG], K = rc[TPD_EQ_K], B = rc[TPD_EQ_B];
          const double imp = tp_impedance(lk + TPK_EQSOLIMP, pos_e, 0.0);
          r.own = true; r.ce = deriv; r.De = tp_rcp(fmax(BS_MINVAL, (1 - imp) * diag * tp_rcp(imp)));
          r.aref_e = -B * (qvel - deriv * vo) - K * imp * pos_e;
          nact++;
        }
        const double De_s = bperm(r.De, eq_src_sa), ce_s = bperm(r.ce, eq_src_sa);
        if (is_src) { r.De_o = De_s; r.ce_o = ce_s; }
      }
      nefc = env_sum_i(nact + cnr);

      // ---- A11 Newton on the dual-free primal (mj_solNewton shape)
      double qfrc_constraint = 0, frc[4] = {0, 0, 0, 0};
      // J_contact^T f: contacts with a force, in contact order; the owner lane broadcasts the world
      // force, every dof lane on one of the two body chains projects it on its own motion axis
      auto con_jtf = [&](const double* cf) -> double {
        double qc = 0;
        if constexpr (CON) {
          V3 Fw = {0, 0, 0};
          bool hasf = false;
          if (cnr > 0) {
            const double fn = (cf[0] + cf[1]) + (cf[2] + cf[3]);
            const double f1 = cmu * (cf[0] - cf[1]), f2 = cmu * (cf[2] - cf[3]);
            const tp_lds_d f = myslot + CG_FRAME;
            Fw = {fn * f[0] + f1 * f[3] + f2 * f[6], fn * f[1] + f1 * f[4] + f2 * f[7], fn * f[2] + f1 * f[5] + f2 * f[8]};
            hasf = fn != 0;
          }
          const unsigned long long fb = __ballot(hasf);
          const tp_mask mine = tp_env_bits(fb, base);
          tp_mask merged = tp_merged(fb);
          while (merged) {
            const int ci = tp_ctz(merged);
            merged &= merged - 1;
            const int sa = (base + ci) * 4;
            const V3 F = {bperm(Fw.x, sa), bperm(Fw.y, sa), bperm(Fw.z, sa)};
            if ((mine >> ci) & 1) {
              const tp_lds_d sl = slot0 + ci * SL;
              const tp_mask m1 = tp_slot_mask(sl, 0), m2 = tp_slot_mask(sl, 1);
              const int sgn = (int)((m2 >> l) & 1) - (int)((m1 >> l) & 1);
              if (sgn != 0) {
                const V3 off = {sl[CG_POS] - com.x, sl[CG_POS + 1] - com.y, sl[CG_POS + 2] - com.z};
                const V3 tau = cross(off, F);
                const double pr = (cdof[0] * tau.x + cdof[1] * tau.y + cdof[2] * tau.z) +
                                  (cdof[3] * F.x + cdof[4] * F.y + cdof[5] * F.z);
                qc += sgn > 0 ? pr : -pr;
              }
            }
          }
        }
        return qc;
      };
      // dense Newton direction: H = (tree-sparse part Hr) + sum over contacts U C U^T with
      // U = J rows of (normal, tangent1, tangent2) and C the 3x3 block of the active pyramid rows;
      // packed lower triangle in LDS, row = body lane; factorised from the last lane upwards
      // (H = L^T L, the elimination order of mj_factorM, so the tree part creates no fill-in)
      auto con_newton_dir = [&](const TpRow& Hr, const double* cj, double g) -> double {
        double xres = 0;
        if constexpr (CON) {
          const int n = m.nbody, ntri = n * (n + 1) / 2;
          for (int q = l; q < ntri; q += TP_LPE) Hs[q] = 0;
          bool hasact = false;
          if (cnr > 0) {
            const double a0 = cj[0] < 0 ? 1.0 : 0.0, a1 = cj[1] < 0 ? 1.0 : 0.0, a2 = cj[2] < 0 ? 1.0 : 0.0,
                         a3 = cj[3] < 0 ? 1.0 : 0.0;
            ((tp_lds_i)(myslot + CG_PAIR))[1] = (cj[0] < 0 ? 1 : 0) | (cj[1] < 0 ? 2 : 0) | (cj[2] < 0 ? 4 : 0) | (cj[3] < 0 ? 8 : 0);
            hasact = (a0 + a1) + (a2 + a3) > 0;
          }
          const unsigned long long ab = __ballot(hasact);
          tp_wave_sync();
          const tp_lds_d Hrow = Hs + l * (l + 1) / 2;
          if (l < n) {
            Hrow[l] = Hr.r[0];
            // the ancestor lanes are recomputed here in every Newton iteration: left to itself the compiler hoists the
            // five LDS addresses out of the solver loop, spills them, and reloads them from scratch in each iteration
            int self_sa_it = c.self_sa, top_sa_it = c.top_sa, dd_it = c.dd;
            asm volatile("" : "+v"(self_sa_it), "+v"(top_sa_it), "+v"(dd_it));
#pragma unroll
            for (int a = 1; a < TP_MAXD; a++)
              if (a < dd_it) {
                // lane of the a-th ancestor dof; CHAIN models: a lanes below, or the dof above the chain — computed, so
                // that the six ancestor addresses need not stay live (they were reloaded from scratch here every iteration)
                const int asa = CHAIN ? (a == dd_it - 1 ? top_sa_it : self_sa_it - 4 * a) : c.anc_sa[a];
                Hrow[(asa >> 2) & (TP_LPE - 1)] = Hr.r[a];
              }
          }
          tp_wave_sync();
          const tp_mask mine = tp_env_bits(ab, base);
          tp_mask merged = tp_merged(ab);
          while (merged) {
            const int ci = tp_ctz(merged);
            merged &= merged - 1;
            const tp_lds_d sl = slot0 + ci * SL;
            double u0 = 0, u1 = 0, u2 = 0;
            tp_mask chain = 0;
            if ((mine >> ci) & 1) {
              const tp_mask m1 = tp_slot_mask(sl, 0), m2 = tp_slot_mask(sl, 1);
              const int sgn = (int)((m2 >> l) & 1) - (int)((m1 >> l) & 1);
              if (sgn != 0) {
                chain = (m1 ^ m2) & ((tp_bit(l) << 1) - 1);
                const V3 off = {sl[CG_POS] - com.x, sl[CG_POS + 1] - com.y, sl[CG_POS + 2] - com.z};
                const V3 w = cross(V3{cdof[0], cdof[1], cdof[2]}, off);
                const V3 v = {cdof[3] + w.x, cdof[4] + w.y, cdof[5] + w.z};
                const tp_lds_d f = sl + CG_FRAME;
                const double sg = sgn > 0 ? 1.0 : -1.0;
                u0 = sg * (f[0] * v.x + f[1] * v.y + f[2] * v.z);
                u1 = sg * (f[3] * v.x + f[4] * v.y + f[5] * v.z);
                u2 = sg * (f[6] * v.x + f[7] * v.y + f[8] * v.z);
              }
            }
            Us[3 * l] = u0; Us[3 * l + 1] = u1; Us[3 * l + 2] = u2;
            tp_wave_sync();
            if (chain) {
              // 3x3 block of the active pyramid rows: rows are n +- mu t1, n +- mu t2, all with weight D
              const int ab = ((tp_lds_i)(sl + CG_PAIR))[1];
              const double a0 = ab & 1 ? 1.0 : 0.0, a1 = ab & 2 ? 1.0 : 0.0, a2 = ab & 4 ? 1.0 : 0.0, a3 = ab & 8 ? 1.0 : 0.0;
              const double Dc = sl[CG_D], muc = sl[CG_MU], Dm = Dc * muc;
              const double Cnn = Dc * ((a0 + a1) + (a2 + a3)), Cn1 = Dm * (a0 - a1), Cn2 = Dm * (a2 - a3);
              const double C11 = Dm * muc * (a0 + a1), C22 = Dm * muc * (a2 + a3);
              const double v0 = Cnn * u0 + Cn1 * u1 + Cn2 * u2;
              const double v1 = Cn1 * u0 + C11 * u1;
              const double v2 = Cn2 * u0 + C22 * u2;
              while (chain) {  // two entries per trip: their eight LDS reads are in flight together
                const int k0 = tp_ctz(chain);
                chain &= chain - 1;
                const bool two = chain != 0;
                const int k1 = two ? tp_ctz(chain) : k0;
                chain &= chain - 1;
                const double a0 = Us[3 * k0], a1 = Us[3 * k0 + 1], a2 = Us[3 * k0 + 2], h0 = Hrow[k0];
                const double b0 = Us[3 * k1], b1 = Us[3 * k1 + 1], b2 = Us[3 * k1 + 2], h1 = Hrow[k1];
                Hrow[k0] = h0 + (v0 * a0 + v1 * a1 + v2 * a2);
                if (two) Hrow[k1] = h1 + (v0 * b0 + v1 * b1 + v2 * b2);
              }
            }
            tp_wave_sync();
          }
          TPC_TL(6)
          // factor H = L^T L from the last lane upwards, TWO columns per step (one pair of LDS hand-offs per
          // panel): for the panel (j1, j0 = j1 - 1) every lane forms the 2x2 block's factor itself from three
          // broadcast reads, lane l < j0 scales its two entries, and the rank-2 trailing update
          // H(i,k) -= L(j1,i) L(j1,k) + L(j0,i) L(j0,k) runs over the packed entries q < tri(j0), dealt
          // round-robin to the 32 lanes (entry q = l + 32 t).  The diagonal keeps 1 / L(j,j).
          // The forward substitution L^T y = g rides along: g is one more row of the trailing update, every
          // lane keeps its own entry in a register and publishes it in Gs for the panel that needs it.
          // row index of packed entry q = l + 32 t, 5 bits each (a table of ints would be spilled)
          // (host table: computed here it cost ~450 VALU instructions per launch, float sqrt + fix-ups per chunk)
#if TP_LPE == 64
          const unsigned char* const trow = t.tri_row;  // wide build: a byte per packed entry (33 chunks of 6 bits do not fit two words)
#else
          const unsigned long long pk0 = t.tri_pk[l], pk1 = t.tri_pk[TP_LPE + l];  // chunks 0..11 and 12..
#endif
          const tp_lds_d Gs = Us;  // the U staging rows are free again
          double greg = g, y = 0;
#if TP_LPE == 32 && TPC_NREG > 0
          // Round 3: the factorisation in REGISTERS (CHAIN kernels, n <= TPC_NREG).  Lane k takes COLUMN k of the packed
          // triangle, h[i] = H(i,k) for i >= k (what it reads for i < k is another row's entry: carried along, never used).
          // Panel (j1, j0): the 2x2 block and the two right-hand sides come from lanes j1 / j0 by ds_bpermute, every
          // lane k < j0 scales ITS OWN two entries (a = L(j1,k), b = L(j0,k): registers h[j1], h[j0], static indices),
          // publishes the pair in LDS, and the trailing update of its column is one broadcast 16-byte read and two FMAs
          // per row, straight-line: h[i] = (h[i] - a_i a) - b_i b — the same operations in the same order as the packed
          // round-robin update below, which spent ~65 instructions per trip on row indices, five LDS addresses per
          // entry and predicates for four FMAs.  Panels are paired statically from row TPC_NREG - 1 down (a two-column
          // panel is bitwise two single-column eliminations, so the pairing does not change the result); a phantom row
          // n (n odd) is an identity row: zeros in every column, one on its own lane, a zero right-hand side.  L goes back to the packed triangle for the back substitution.
          const bool regchol = CHAIN && n <= TPC_NREG;
          if (regchol) {
            typedef double tp_d2 __attribute__((ext_vector_type(2)));
            typedef tp_d2 __attribute__((address_space(3)))* tp_lds_d2;
            // opaque copies of the lane number and of the half's first permute address: as loop invariants the
            // compiler hoists every panel's lane masks and permute addresses out of the Newton loop and spills them
            int lo = l, base4 = base * 4;
            asm volatile("" : "+v"(lo), "+v"(base4));
            double h[TPC_NREG];
            const tp_lds_d Hcol = Hs + l;
#pragma unroll
            for (int i = 0; i < TPC_NREG; i++) h[i] = i < n ? Hcol[i * (i + 1) / 2] : (lo == i ? 1.0 : 0.0);  // rows >= n: identity
            const tp_lds_d2 AB = (tp_lds_d2)Us;  // (a, b) of every lane: one 16-byte entry each (Us is 16-byte aligned)
            static_assert(TPC_NREG % 2 == 0, "rows are paired (j1 odd, j1 - 1) down to (1, 0)");
#pragma unroll
            for (int j1 = TPC_NREG - 1; j1 >= 1; j1 -= 2) {
              const int j0 = j1 - 1;
              if (j0 >= n || j1 < t.first_dof) continue;  // wave-uniform; rows without a dof (identity rows, zero right-hand sides) are left as they are
              const int sa1 = base4 + 4 * j1, sa0 = base4 + 4 * j0;
              double d1 = bperm(h[j1], sa1);
              const double x10 = bperm(h[j1], sa0), d0r = bperm(h[j0], sa0);
              const double g1 = bperm(greg, sa1), g0 = bperm(greg, sa0);
              if (d1 < BS_MINVAL) d1 = BS_MINVAL;
              const double inv1 = tp_rsqrt(d1);
              const double y1 = g1 * inv1;
              const double l10 = x10 * inv1;
              double d0 = d0r - l10 * l10;
              if (d0 < BS_MINVAL) d0 = BS_MINVAL;
              const double inv0 = tp_rsqrt(d0);
              const double y0 = (g0 - l10 * y1) * inv0;
              const double a = h[j1] * inv1, b = (h[j0] - l10 * a) * inv0;
              greg -= a * y1;
              greg -= b * y0;
              h[j1] = lo < j0 ? a : (lo == j0 ? l10 : inv1);
              h[j0] = lo < j0 ? b : inv0;
              if (lo == j0) y = y0;
              if (lo == j1) y = y1;
              AB[lo] = tp_d2{a, b};
              tp_wave_sync();
              // four rows per round of broadcast reads (six / eight / twelve rows, or the next round's reads issued before
              // this round's FMAs, measured the same or slower: the column already holds 48 registers)
              {
                tp_d2 cur[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                  if (q < j0) cur[q] = AB[q];
#pragma unroll
                for (int i0 = 0; i0 < j0; i0 += 4) {
#pragma unroll
                  for (int q = 0; q < 4; q++)
                    if (i0 + q < j0) h[i0 + q] = (h[i0 + q] - cur[q].x * a) - cur[q].y * b;
#pragma unroll
                  for (int q = 0; q < 4; q++)
                    if (i0 + 4 + q < j0) cur[q] = AB[i0 + 4 + q];
                }
              }
              tp_wave_sync();
            }
            if (lo < n) {
#pragma unroll
              for (int i = 0; i < TPC_NREG; i++)
                if (i < n) Hcol[i * (i + 1) / 2] = h[i];  // ascending rows: what a lane k > i writes here lands in a later row and is overwritten
            }
            tp_wave_sync();
          } else {
#else
          {
#endif
          Gs[l] = greg;
          tp_wave_sync();
          for (int j1 = n - 1; j1 >= t.first_dof; j1 -= 2) {  // (panels of rows without a dof: identity rows, nothing to do)
            const int t1 = j1 * (j1 + 1) / 2, j0 = j1 - 1;
            const tp_lds_d H1 = Hs + t1;
            double d1 = H1[j1];
            if (d1 < BS_MINVAL) d1 = BS_MINVAL;
            const double inv1 = tp_rsqrt(d1);
            const double y1 = Gs[j1] * inv1;
            if (j0 < 0) {  // n odd: the last column stands alone and has nothing to update
              if (l == 0) { H1[0] = inv1; y = y1; }
              tp_wave_sync();
              break;
            }
            const int tj = t1 - j1;  // tri(j0)
            const tp_lds_d H0 = Hs + tj;
            const double l10 = H1[j0] * inv1;
            double d0 = H0[j0] - l10 * l10;
            if (d0 < BS_MINVAL) d0 = BS_MINVAL;
            const double inv0 = tp_rsqrt(d0);
            const double y0 = (Gs[j0] - l10 * y1) * inv0;
            if (l < j0) {
              const double a = H1[l] * inv1, b = (H0[l] - l10 * a) * inv0;
              H1[l] = a; H0[l] = b;
              greg -= a * y1;
              greg -= b * y0;
              Gs[l] = greg;
            }
            if (l == j0) { H1[j0] = l10; H0[j0] = inv0; y = y0; }
            if (l == j1) { H1[j1] = inv1; y = y1; }
            tp_wave_sync();
            // two chunks per trip, every LDS read of a trip issued before the first use (lanes beyond the
            // triangle read entry 0 and do not write).  A rolled loop on purpose: unrolled, the compiler hoists
            // the sixteen unpacked row indices out of the panel loop and spills them — reloads in here
#pragma unroll 1
            for (int tt = 0; tt * TP_LPE < tj; tt += 2) {
#if TP_LPE != 64
              const unsigned w = (unsigned)(tt < 12 ? pk0 >> (5 * tt) : pk1 >> (5 * (tt - 12)));
#endif
              double h[2], ai[2], ak[2], bi[2], bk[2];
              int q[2];
#pragma unroll
              for (int u = 0; u < 2; u++) {
                q[u] = l + (tt + u) * TP_LPE;
                const bool on = q[u] < tj;
#if TP_LPE == 64
                const int i = on ? (int)trow[q[u]] : 0;
#else
                const int i = on ? (int)((w >> (5 * u)) & 31u) : 0;
#endif
                const int k = on ? q[u] - i * (i + 1) / 2 : 0;
                h[u] = Hs[on ? q[u] : 0];
                ai[u] = H1[i]; ak[u] = H1[k]; bi[u] = H0[i]; bk[u] = H0[k];
              }
#pragma unroll
              for (int u = 0; u < 2; u++)
                if (q[u] < tj) Hs[q[u]] = (h[u] - ai[u] * ak[u]) - bi[u] * bk[u];
            }
            tp_wave_sync();
          }
          }
          TPC_TL(7)
          // L x = y (ascending), one unknown per step: every lane forms its candidate (y - acc) / L(l,l) from what it
          // has accumulated so far, lane k's is final at step k and goes to all lanes by ds_bpermute, the lanes below
          // add L(l,k) x_k (one read of the lane's own packed row).  Nine instructions per unknown; the version of
          // round 2 resolved four unknowns per cross-lane round trip redundantly on every lane from broadcast reads of
          // L (676 instructions for the hand's 24 rows against ~230, and a longer dependent chain in practice).
          // What a lane adds after its own step is never used (rows end at the diagonal: it reads past its row).
          {
            double acc = 0;
            const tp_lds_d Lrow = l < n ? Hrow : Hs;  // lanes beyond the bodies read row 0 (their result is unused)
            const double invl = l < n ? Hrow[l] : 0.0;
#pragma unroll 4
            for (int k = t.first_dof; k < n; k++) {  // the unknowns of the identity rows below the first dof are zero
              const double xl = (y - acc) * invl;
              const double xk = bperm(xl, (base + k) * 4);
              if (l == k) xres = xl;
              acc += Lrow[k] * xk;
            }
          }
          TPC_TL(8)
        }
        return xres;
      };
      TPC_T(3)
      if (nefc == 0) {  // env-uniform
        qacc = qacc_smooth; qacc_ws = qacc_smooth; solver_iter = 0;
      } else if constexpr (PGS) {
        // ---- A11 mj_solPGS (same arithmetic as physics_contact.h::stage_pgs, contact-free)
        const int mstride = tp_minv_stride(m.nbody);
        const tp_lds_d Minv = (tp_lds_d)sh.minv + (size_t)(tid / TP_LPE) * m.nbody * mstride;
        const tp_lds_d Mrow = Minv + (l < m.nbody ? l : 0) * mstride;  // lanes beyond the bodies never write
        const tp_mask dofs = tp_env_bits(__ballot(has_dof), base);
        for (tp_mask left = dofs; left;) {  // M^-1 e_j for every dof (lane j), three columns per sweep
          int col[3];
          double x[3];
#pragma unroll
          for (int n = 0; n < 3; n++) {
            col[n] = left ? tp_ctz(left) : -1;
            if (left) left &= left - 1;
            x[n] = l == col[n] ? 1.0 : 0.0;
          }
          tp_solve_n<3, CHAIN>(c, LM.R, LM.dinv, x);
#pragma unroll
          for (int n = 0; n < 3; n++)
            if (col[n] >= 0 && l < m.nbody) Mrow[col[n]] = x[n];
        }
        tp_wave_sync();
        TPC_T(20)
        const int ko = (eq_other_sa >> 2) & (TP_LPE - 1);  // lane of the equality's other dof (own lane if none)
        const tp_lds_d Orow = Minv + ko * mstride;
        // (M^-1 J_r^T)_l for the row kinds, J_r (M^-1 J_r^T) + 1/D on the owner lane
        const double mjj = has_dof ? Mrow[l] : 0.0;
        const double ARf = mjj + (r.Df > 0 ? tp_rcp(r.Df) : 0.0), AR0 = mjj + (r.Dl0 > 0 ? tp_rcp(r.Dl0) : 0.0),
                     AR1 = mjj + (r.Dl1 > 0 ? tp_rcp(r.Dl1) : 0.0);
        const double ARe = r.own ? (Mrow[l] - r.ce * Mrow[ko]) - r.ce * (Orow[l] - r.ce * Orow[ko]) + tp_rcp(r.De) : 1.0;
        // ---- contacts (k_tpc_pgs): U_c = Jacobian rows (normal, tangent1, tangent2) of contact c, one
        // entry per dof lane, recomputed from the lane's motion axis where needed; w_q = M^-1 U_c[q]^T
        // through the M^-1 rows in LDS; the 3x3 block B_c = U_c M^-1 U_c^T is formed once per step and
        // kept in the contact's slot (it couples the four pyramid rows of the contact with each other)
        const int half_ = tid / TP_LPE;
        const unsigned long long conb = __ballot(CON && l < ncon_env);
        const int ncon_max = CON ? (TP_LPE == 64 ? (int)__popcll(conb) : (int)max(__popcll(conb & 0xffffffffull), __popcll(conb >> 32))) : 0;
        auto con_u = [&](tp_lds_d sl, bool act, double& u0, double& u1, double& u2, tp_mask& chain) {
          u0 = u1 = u2 = 0; chain = 0;
          if constexpr (CON) {
            if (act) {
              const tp_mask m1 = tp_slot_mask(sl, 0), m2 = tp_slot_mask(sl, 1);
              chain = m1 ^ m2;
              const int sgn = (int)((m2 >> l) & 1) - (int)((m1 >> l) & 1);
              if (sgn != 0) {
                const V3 off = {sl[CG_POS] - com.x, sl[CG_POS + 1] - com.y, sl[CG_POS + 2] - com.z};
                const V3 w = cross(V3{cdof[0], cdof[1], cdof[2]}, off);
                const V3 v = {cdof[3] + w.x, cdof[4] + w.y, cdof[5] + w.z};
                const tp_lds_d f = sl + CG_FRAME;
                const double sg = sgn > 0 ? 1.0 : -1.0;
                u0 = sg * (f[0] * v.x + f[1] * v.y + f[2] * v.z);
                u1 = sg * (f[3] * v.x + f[4] * v.y + f[5] * v.z);
                u2 = sg * (f[6] * v.x + f[7] * v.y + f[8] * v.z);
              }
            }
          }
        };
        // w_q[l] = sum_k M^-1[l][k] U[q][k] over the dofs k of the two chains: U staged in LDS, M^-1 row of this lane
        auto con_w = [&](double u0, double u1, double u2, tp_mask chain, double& w0, double& w1, double& w2) {
          w0 = w1 = w2 = 0;
          if constexpr (CON) {
            Us[3 * l] = u0; Us[3 * l + 1] = u1; Us[3 * l + 2] = u2;
            tp_wave_sync();
            if (l < m.nbody)
              for (tp_mask ch = chain; ch; ch &= ch - 1) {
                const int k = tp_ctz(ch);
                const double mk = Mrow[k];
                w0 += mk * Us[3 * k]; w1 += mk * Us[3 * k + 1]; w2 += mk * Us[3 * k + 2];
              }
            tp_wave_sync();
          }
        };
        if constexpr (CON) {
          for (int ci = 0; ci < ncon_max; ci++) {
            const tp_lds_d sl = slot0 + ci * SL;
            const bool act = ci < ncon_env && ((tp_lds_i)(sl + CG_PAIR))[1] > 0;
            double u0, u1, u2, w0, w1, w2;
            tp_mask chain;
            con_u(sl, act, u0, u1, u2, chain);
            con_w(u0, u1, u2, chain, w0, w1, w2);
            const double bnn = env_sum(u0 * w0), bn1 = env_sum(u0 * w1), bn2 = env_sum(u0 * w2), b11 = env_sum(u1 * w1),
                         b12 = env_sum(u1 * w2), b22 = env_sum(u2 * w2);
            if (act && l == 0) {
              sl[CG_B] = bnn; sl[CG_B + 1] = bn1; sl[CG_B + 2] = bn2; sl[CG_B + 3] = b11; sl[CG_B + 4] = b12; sl[CG_B + 5] = b22;
            }
          }
          tp_wave_sync();
        }
        (void)half_;
        double fF = 0, fL0 = 0, fL1 = 0, fE = 0, acc = qacc_smooth;
        double cfo[4] = {0, 0, 0, 0};  // forces of the contact this lane owns
        auto jtf = [&]() -> double {  // J^T f on this lane
          const double fo = bperm(fE, eq_src_sa);
          double q = fF + fL0 - fL1 + fE;
          if (is_src) q -= r.ce_o * fo;
          return has_dof ? q : 0.0;
        };
        if (!(m.disableflags & BS_DSBL_WARMSTART)) {
          // forces the previous acceleration implies; kept if their dual cost is <= 0
          const double ao = bperm(qacc_ws, eq_other_sa);
          const double jf = qacc_ws - r.aref_f, jl0 = qacc_ws - r.aref_l0, jl1 = -qacc_ws - r.aref_l1;
          const double je = r.own ? (qacc_ws - r.ce * ao) - r.aref_e : 0.0;
          double f4[4];
          tp_rows_cost<true>(r, jf, jl0, jl1, je, f4);
          fF = r.Df > 0 ? f4[0] : 0.0; fL0 = r.Dl0 > 0 ? f4[1] : 0.0; fL1 = r.Dl1 > 0 ? f4[2] : 0.0; fE = r.own ? f4[3] : 0.0;
          if constexpr (CON) {  // contact rows: f = -D min(J qacc_ws - aref, 0)
            double u3[3], cjt[4];
            con_jx(qacc_ws, u3);
            con_rows(u3, WITH_AREF, cjt);
            tp_con_cost<true>(r.cD, cjt, cfo);
          }
          double jt = jtf();
          if constexpr (CON) { jt += con_jtf(cfo); if (!has_dof) jt = 0; }
          const double w = tp_solve<CHAIN>(c, LM.R, LM.dinv, jt);  // M^-1 J^T f
          const double wo = bperm(w, eq_other_sa), so = bperm(qacc_smooth, eq_other_sa);
          double cl = 0;
          if constexpr (CON) {
            double ua[3], ub[3], jw[4], js[4];
            con_jx2(w, qacc_smooth, ua, ub);
            con_rows(ua, NO_AREF, jw);   // J_r w
            con_rows(ub, WITH_AREF, js);   // J_r qacc_smooth - aref_r
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (q < cnr) cl += cfo[q] * (0.5 * (jw[q] + cfo[q] / r.cD) + js[q]);
          }
          if (r.Df > 0) cl += fF * (0.5 * (w + fF / r.Df) + (qacc_smooth - r.aref_f));
          if (r.Dl0 > 0) cl += fL0 * (0.5 * (w + fL0 / r.Dl0) + (qacc_smooth - r.aref_l0));
          if (r.Dl1 > 0) cl += fL1 * (0.5 * (-w + fL1 / r.Dl1) + (-qacc_smooth - r.aref_l1));
          if (r.own) cl += fE * (0.5 * ((w - r.ce * wo) + fE / r.De) + ((qacc_smooth - r.ce * so) - r.aref_e));
          if (env_sum(cl) > 0) { fF = fL0 = fL1 = fE = 0; cfo[0] = cfo[1] = cfo[2] = cfo[3] = 0; }
          else acc = qacc_smooth + w;
          if constexpr (CON) {
            if (l < ncon_env)
#pragma unroll
              for (int q = 0; q < 4; q++) myslot[CG_F + q] = cfo[q];
            tp_wave_sync();
          }
        }
        const tp_mask actF = tp_env_bits(__ballot(r.Df > 0), base), act0 = tp_env_bits(__ballot(r.Dl0 > 0), base),
                      act1 = tp_env_bits(__ballot(r.Dl1 > 0), base), actE = tp_env_bits(__ballot(r.own), base);
        int iter = 0;
        TPC_T(21)
        // reciprocals of D and of the diagonal of A + R once per step (the sweep is a latency chain)
        const double iDf = r.Df > 0 ? tp_rcp(r.Df) : 0.0, iD0 = r.Dl0 > 0 ? tp_rcp(r.Dl0) : 0.0, iD1 = r.Dl1 > 0 ? tp_rcp(r.Dl1) : 0.0,
                     iDe = r.own ? tp_rcp(r.De) : 0.0;
        const double iARf = tp_rcp(ARf), iAR0 = tp_rcp(AR0), iAR1 = tp_rcp(AR1), iARe = tp_rcp(ARe);
        // one row: the owner lane ol projects its force, everybody applies the change.  kind is
        // uniform over the wavefront, ol over the environment.
        auto row = [&](int kind, int ol, double& improvement) {
          const double ao = kind == 0 ? bperm(acc, eq_other_sa) : 0.0;
          double delta = 0;
          if (l == ol) {
            double iD, aref, old, ARii, iAR, ja;
            if (kind == 0) { iD = iDe; aref = r.aref_e; old = fE; ARii = ARe; iAR = iARe; ja = acc - r.ce * ao; }
            else if (kind == 1) { iD = iDf; aref = r.aref_f; old = fF; ARii = ARf; iAR = iARf; ja = acc; }
            else if (kind == 2) { iD = iD0; aref = r.aref_l0; old = fL0; ARii = AR0; iAR = iAR0; ja = acc; }
            else { iD = iD1; aref = r.aref_l1; old = fL1; ARii = AR1; iAR = iAR1; ja = -acc; }
            const double res = ja - aref + old * iD;
            double f = old - res * iAR;
            if (kind == 1) f = f < -r.fl ? -r.fl : (f > r.fl ? r.fl : f);
            else if (kind >= 2) { if (f < 0) f = 0; }
            delta = f - old;
            double change = 0.5 * delta * delta * ARii + delta * res;
            if (change > 1e-10) { f = old; delta = 0; change = 0; }
            if (kind == 0) fE = f; else if (kind == 1) fF = f; else if (kind == 2) fL0 = f; else fL1 = f;
            improvement -= change;
          }
          const int osa = (base + ol) * 4;
          const double dl = bperm(delta, osa);
          // (M^-1 J_r^T)_l: column ol of M^-1 (sign for the upper limit, minus ce * the other dof's column for an equality)
          double wl = Mrow[ol];
          if (kind == 3) wl = -wl;
          if (kind == 0) {
            const int ok = __builtin_amdgcn_ds_bpermute(osa, ko);
            const double ce = bperm(r.ce, osa);
            wl -= ce * Mrow[ok];
          }
          if (has_dof) acc += dl * wl;
        };
        const tp_mask actL = act0 | act1;
        while (iter < m.iterations) {
          double improvement = 0;
          // efc order: equalities (host schedule: their owners are in equality order), frictionloss rows
          // and limits in dof order = lane order, lower before upper limit of a joint
          for (int q = 0; q < t.npgs; q++) {
            const int ol = t.pgs_order[q];
            if ((actE >> ol) & 1) row(0, ol, improvement);
          }
          for (tp_mask mk = actF; mk; mk &= mk - 1) row(1, tp_ctz(mk), improvement);
          for (tp_mask mk = actL; mk; mk &= mk - 1) {
            const int ol = tp_ctz(mk);
            if ((act0 >> ol) & 1) row(2, ol, improvement);
            if ((act1 >> ol) & 1) row(3, ol, improvement);
          }
          if constexpr (CON) {
            // contact rows, in contact order: every lane of the environment updates the (up to) four
            // rows of contact ci with the same arithmetic (J_r acc from three environment-wide sums,
            // the rows' coupling through B_c), then adds the force change times M^-1 U_c^T to its acceleration
            for (int ci = 0; ci < ncon_max; ci++) {
              const tp_lds_d sl = slot0 + ci * SL;
              const int nr = ci < ncon_env ? ((tp_lds_i)(sl + CG_PAIR))[1] : 0;
              const bool act = nr > 0;
              double u0, u1, u2, w0, w1, w2;
              tp_mask chain;
              con_u(sl, act, u0, u1, u2, chain);
              double jn = env_sum(u0 * acc), j1 = env_sum(u1 * acc), j2 = env_sum(u2 * acc);
              con_w(u0, u1, u2, chain, w0, w1, w2);
              if (act) {
                const double Dc = sl[CG_D], muc = sl[CG_MU], iDc = tp_rcp(Dc);
                const double bnn = sl[CG_B], bn1 = sl[CG_B + 1], bn2 = sl[CG_B + 2], b11 = sl[CG_B + 3], b12 = sl[CG_B + 4],
                             b22 = sl[CG_B + 5];
                double dn = 0, d1 = 0, d2 = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                  if (q >= nr) continue;
                  // row q: J = n + s1 t1 + s2 t2 with (s1, s2) = (mu, 0), (-mu, 0), (0, mu), (0, -mu); condim 1: J = n
                  const double s1 = nr == 1 ? 0.0 : (q == 0 ? muc : (q == 1 ? -muc : 0.0));
                  const double s2 = nr == 1 ? 0.0 : (q == 2 ? muc : (q == 3 ? -muc : 0.0));
                  const double gn = bnn + s1 * bn1 + s2 * bn2, g1 = bn1 + s1 * b11 + s2 * b12, g2 = bn2 + s1 * b12 + s2 * b22;
                  const double ARii = gn + s1 * g1 + s2 * g2 + iDc;
                  const double old = sl[CG_F + q];
                  const double res = (jn + s1 * j1 + s2 * j2) - sl[CG_AREF + q] + old * iDc;
                  double f = old - res * tp_rcp(ARii);
                  if (f < 0) f = 0;
                  double delta = f - old, change = 0.5 * delta * delta * ARii + delta * res;
                  if (change > 1e-10) { f = old; delta = 0; change = 0; }
                  if (l == 0) { sl[CG_F + q] = f; improvement -= change; }
                  jn += delta * gn; j1 += delta * g1; j2 += delta * g2;
                  dn += delta; d1 += delta * s1; d2 += delta * s2;
                }
                if (has_dof) acc += dn * w0 + d1 * w1 + d2 * w2;
              }
              tp_wave_sync();
            }
          }
          iter++;
          if (env_sum(improvement) * scale < m.tolerance) break;
        }
        TPC_T(22)
        solver_iter = iter;
        frc[0] = fF; frc[1] = fL0; frc[2] = fL1; frc[3] = fE;
        qfrc_constraint = jtf();
        if constexpr (CON) {
          if (l < ncon_env)
#pragma unroll
            for (int q = 0; q < 4; q++) cfo[q] = myslot[CG_F + q];
          qfrc_constraint += con_jtf(cfo);
          if (!has_dof) qfrc_constraint = 0;
        }
        qacc = qacc_smooth + tp_solve<CHAIN>(c, LM.R, LM.dinv, qfrc_constraint);
        if (!has_dof) qacc = 0;
        qacc_ws = qacc;
      } else {
        TpSol s;
#pragma unroll
        for (int q = 0; q < 4; q++) { s.cj[q] = 0; s.cv[q] = 0; }
        auto residual_fixed = [&](double a, auto SUB, double& jf, double& jl0, double& jl1, double& je) {
          constexpr bool sub = decltype(SUB)::value;
          const double ao = bperm(a, eq_other_sa);
          if constexpr (sub) {
            jf = a - r.aref_f; jl0 = a - r.aref_l0; jl1 = -a - r.aref_l1;
            je = r.own ? (a - r.ce * ao) - r.aref_e : 0.0;
          } else {
            jf = a; jl0 = a; jl1 = -a;
            je = r.own ? (a - r.ce * ao) : 0.0;
          }
        };
        auto residual = [&](double a, auto SUB, double& jf, double& jl0, double& jl1, double& je, double* cjr) {
          residual_fixed(a, SUB, jf, jl0, jl1, je);
          if constexpr (CON) {
            double u3[3];
            con_jx(a, u3);
            con_rows(u3, SUB, cjr);
          }
        };
        // warm start: previous acceleration only if cheaper than the unconstrained one
        // (M qacc_smooth == qfrc_smooth, so its Gauss term vanishes and Ma is known)
        // warm start: the residuals of the previous acceleration and of the unconstrained one in ONE pass
        // (with contacts the two vectors ride down the body tree together), the cheaper candidate's
        // residuals are kept — no third evaluation
        bool use_smooth = true;
        {
          double wf, wl0, wl1, we, wc[4] = {0, 0, 0, 0};
          const bool ws_on = !(m.disableflags & BS_DSBL_WARMSTART);
          if (CON && ws_on) {
            residual_fixed(qacc_smooth, WITH_AREF, s.jf, s.jl0, s.jl1, s.je);
            residual_fixed(qacc_ws, WITH_AREF, wf, wl0, wl1, we);
            double ua[3], ub[3];
            con_jx2(qacc_smooth, qacc_ws, ua, ub);
            con_rows(ua, WITH_AREF, s.cj);
            con_rows(ub, WITH_AREF, wc);
          } else {
            residual(qacc_smooth, WITH_AREF, s.jf, s.jl0, s.jl1, s.je, s.cj);
            if (ws_on) residual(qacc_ws, WITH_AREF, wf, wl0, wl1, we, wc);
          }
          if (ws_on) {
            double cl = tp_rows_cost<false>(r, wf, wl0, wl1, we, nullptr) +
                        0.5 * (Ma_ws - qfrc_smooth) * (qacc_ws - qacc_smooth);
            if constexpr (CON) cl += tp_con_cost<false>(r.cD, wc, nullptr);
            const double cw = env_sum(cl);
            cl = tp_rows_cost<false>(r, s.jf, s.jl0, s.jl1, s.je, nullptr);
            if constexpr (CON) cl += tp_con_cost<false>(r.cD, s.cj, nullptr);
            const double cs = env_sum(cl);
            use_smooth = cw > cs;
            if (!use_smooth) {
              s.jf = wf; s.jl0 = wl0; s.jl1 = wl1; s.je = we;
              if constexpr (CON)
#pragma unroll
                for (int q = 0; q < 4; q++) s.cj[q] = wc[q];
            }
          }
        }
        s.qacc = has_dof ? (use_smooth ? qacc_smooth : qacc_ws) : 0.0;
        s.Ma = use_smooth ? qfrc_smooth : Ma_ws;
        double gauss = 0, cost = 0, oldcost = 0;
        double cg_grad = 0, cg_Mgrad = 0;  // CG: gradient and M^-1 gradient of the previous iteration
        int iter = 0;
        bool first = true;
        TPC_T(4)
        while (true) {
          // constraint update: forces, cost, qfrc_constraint = J^T f, gradient
          double cf[4] = {0, 0, 0, 0};
          double cr = tp_rows_cost<true>(r, s.jf, s.jl0, s.jl1, s.je, frc);
          if constexpr (CON) cr += tp_con_cost<true>(r.cD, s.cj, cf);
          gauss = env_sum(0.5 * (s.Ma - qfrc_smooth) * (s.qacc - qacc_smooth));
          cost = gauss + env_sum(cr);
          {
            const double fo = bperm(frc[3], eq_src_sa);
            qfrc_constraint = frc[0] + frc[1] - frc[2] + frc[3];
            if (is_src) qfrc_constraint -= r.ce_o * fo;
            if constexpr (CON) qfrc_constraint += con_jtf(cf);
            if (!has_dof) qfrc_constraint = 0;
          }
          s.grad = s.Ma - qfrc_smooth - qfrc_constraint;
          if (!first) {
            iter++;
            const double improvement = scale * (oldcost - cost);
            const double gradient = scale * sqrt(env_sum(s.grad * s.grad));
            if (improvement < m.tolerance || gradient < m.tolerance) break;
          }
          first = false;
          if (iter >= m.iterations) break;
          TPC_T(5)
          if constexpr (CG) {
            const double Mg = tp_solve<CHAIN>(c, LM.R, LM.dinv, s.grad);  // every lane takes part in the sweeps
            double beta = 0;
            if (iter > 0) {
              const double gg_old = env_sum(cg_grad * cg_Mgrad);
              const double num = env_sum(s.grad * (Mg - cg_Mgrad));
              beta = num / (gg_old > BS_MINVAL ? gg_old : BS_MINVAL);
              if (beta < 0) beta = 0;
            }
            s.search = has_dof ? -Mg + (iter > 0 ? beta * s.search : 0.0) : 0.0;
            cg_grad = s.grad; cg_Mgrad = Mg;
          } else {
          // H = M + J^T D_active J: without contacts it keeps the dof-tree sparsity
          TpRow Hr = Mr;
          if (has_dof) {
            double add = 0;
            if (r.Df > 0 && s.jf > -r.rf && s.jf < r.rf) add += r.Df;
            if (s.jl0 < 0) add += r.Dl0;
            if (s.jl1 < 0) add += r.Dl1;
            if (r.own) {
              add += r.De;
#pragma unroll
              for (int a = 1; a < TP_MAXD; a++)
                if (a == eq_pos) Hr.r[a] -= r.De * r.ce;
            }
            if (is_src) add += r.De_o * r.ce_o * r.ce_o;
            Hr.r[0] += add;
          }
          if constexpr (CON) {
            const double x = con_newton_dir(Hr, s.cj, has_dof ? s.grad : 0.0);
#ifdef TPC_CUTS
            if (cut >= 6 && cut <= 8) { TPC_CLR return; }
#endif
            s.search = has_dof ? -x : 0.0;
          } else {
            const double x = tp_factor_solve<CHAIN>(c, Hr, s.grad);
            s.search = has_dof ? -x : 0.0;
          }
          }
          TPC_T(9)
          s.Mv = tp_mul_m<CHAIN>(c, Mr, s.search);
          TPC_C(23)
          residual(s.search, NO_AREF, s.vf, s.vl0, s.vl1, s.ve, s.cv);
          TPC_C(24)
          const double alpha = tp_line_search<CON>(m, r, s, gauss, scale, qfrc_smooth);
          if (alpha == 0) break;
          s.qacc += alpha * s.search; s.Ma += alpha * s.Mv;
          s.jf += alpha * s.vf; s.jl0 += alpha * s.vl0; s.jl1 += alpha * s.vl1; s.je += alpha * s.ve;
          if constexpr (CON)
#pragma unroll
            for (int q = 0; q < 4; q++) s.cj[q] += alpha * s.cv[q];
          oldcost = cost;
          TPC_T(10)
        }
        TPC_C(25)
        solver_iter = iter;
        qacc = s.qacc;
        qacc_ws = s.qacc;
      }
      // ================= end of forward pass =================

      if (mode == 0 && stage == 0 && !redone && env_any(has_dof && bad_num(qacc), base)) {  // mj_checkAcc
        reset_state(); warning |= 4;
        if constexpr (CON) break;  // contacts belong to the old pose: the next step recomputes everything
        rk_q0 = qpos; rk_v0 = qvel; rk_t0 = time;
        redone = true; stage = -1;
        continue;
      }
      if (stage == 0 && step == nsteps - 1 && env_ok && dump >= 1) {
        if (has_dof) AT(d.qfrc_constraint, j) = qfrc_constraint;
        if (FULL && l < m.nbody) {
          AT(d.xpos, 3 * b) = pos.x; AT(d.xpos, 3 * b + 1) = pos.y; AT(d.xpos, 3 * b + 2) = pos.z;
          AT(d.xquat, 4 * b) = quat.w; AT(d.xquat, 4 * b + 1) = quat.x; AT(d.xquat, 4 * b + 2) = quat.y;
          AT(d.xquat, 4 * b + 3) = quat.z;
        }
        if (FULL && dump >= 2) {
          if (l < m.nbody) {
            AT(d.xipos, 3 * b) = xipos.x; AT(d.xipos, 3 * b + 1) = xipos.y; AT(d.xipos, 3 * b + 2) = xipos.z;
            if (root == b && b > 0) {
              AT(d.rootcom, 3 * b) = com.x; AT(d.rootcom, 3 * b + 1) = com.y; AT(d.rootcom, 3 * b + 2) = com.z;
            }
#pragma unroll
            for (int k = 0; k < 10; k++) AT(d.cinert, 10 * b + k) = cinert[k];
#pragma unroll
            for (int k = 0; k < 6; k++) AT(d.cvel, 6 * b + k) = cvel[k];
          }
          if (has_dof) {
            AT(d.xanchor, 3 * j) = anchor.x; AT(d.xanchor, 3 * j + 1) = anchor.y; AT(d.xanchor, 3 * j + 2) = anchor.z;
            AT(d.xaxis, 3 * j) = axis.x; AT(d.xaxis, 3 * j + 1) = axis.y; AT(d.xaxis, 3 * j + 2) = axis.z;
#pragma unroll
            for (int k = 0; k < 6; k++) AT(d.cdof, 6 * j + k) = cdof[k];
            const int adr = m.dof_Madr[j];
#pragma unroll
            for (int a = 0; a < TP_MAXD; a++)
              if (a < c.dd) { AT(d.qM, adr + a) = Mr.r[a]; AT(d.qLD, adr + a) = LM.R.r[a]; }
            AT(d.qLDiagInv, j) = LM.dinv;
            const int fr = t.lane[TPI_FRIC_ROW * TP_LPE + l], lr = t.lane[TPI_LIM_ROW * TP_LPE + l];
            if (fr >= 0) { AT(d.efc_force, fr) = frc[0]; AT(d.efc_D, fr) = r.Df; AT(d.efc_aref, fr) = r.aref_f; }
            if (lr >= 0) {
              AT(d.efc_force, lr) = frc[1]; AT(d.efc_force, lr + 1) = frc[2];
              AT(d.efc_D, lr) = r.Dl0; AT(d.efc_D, lr + 1) = r.Dl1;
              AT(d.efc_aref, lr) = r.aref_l0; AT(d.efc_aref, lr + 1) = r.aref_l1;
            }
            if (eq_own >= 0) { AT(d.efc_force, eq_own) = frc[3]; AT(d.efc_D, eq_own) = r.De; AT(d.efc_aref, eq_own) = r.aref_e; }
          }
        }
      }
      TPC_T(17)
      if (mode == 1) break;
      if (!RK4) {
        // ---- A12 Euler with implicit joint damping
        double acc = qacc;
        if (m.damped) {
          const double rhs = has_dof ? qfrc_smooth + qfrc_constraint : 0.0;
          if (euler_early) {  // own lane's values: no synchronisation needed
            TpFac LE;
#pragma unroll
            for (int a = 0; a < TP_MAXD; a++) LE.R.r[a] = sh.park[a * 64 + tid];
            LE.dinv = sh.park[TP_MAXD * 64 + tid];
            acc = tp_solve<CHAIN>(c, LE.R, LE.dinv, rhs);
          } else {
            // no early factor (k_tpc has no LDS to park it in): the forward substitution rides in the
            // factorisation's leaves-to-root exchanges, as in Newton's direction
            TpRow Hr = Mr;
            Hr.r[0] += h * damping;
            acc = tp_factor_solve<CHAIN>(c, Hr, rhs);
          }
        }
        qvel += h * acc;
        qpos += h * qvel;
        time += h;
      } else {
        // ---- A13 RK4, classical tableau: A = diag-below (1/2, 1/2, 1), B = (1/6, 1/3, 1/3, 1/6)
        const double Bw = (stage == 0 || stage == 3) ? 1.0 / 6.0 : 1.0 / 3.0;
        rk_dq += Bw * qvel; rk_dv += Bw * qacc;
        if (stage < 3) {
          const double A = stage == 2 ? 1.0 : 0.5;
          const double kq = qvel, kv = qacc;
          qpos = rk_q0 + h * (A * kq); qvel = rk_v0 + h * (A * kv); time = rk_t0 + A * h;
        } else {
          qvel = rk_v0 + h * rk_dv; qpos = rk_q0 + h * rk_dq; time = rk_t0 + h;
        }
      }
    }
  }
  TPC_C(26)
#ifdef TPC_PROFILE
  TPC_T(11)
  if constexpr (CON) {
    if (tid == 0 && cd.stat)
      for (int q = 0; q < 24; q++) atomicAdd(cd.stat + 8 + q, tacc[q]);
  } else if (blockIdx.x == 7 && tid == 0) {
    for (int q = 0; q < 24; q++) printf("sec %d %llu\n", q, tacc[q]);
  }
#endif
  if constexpr (CON) {
    // the narrowphase has consumed the work list of this range: clear its append counters for the
    // next pass (k_tp_col of the next pass starts after this kernel ends)
    if (blockIdx.x == 0 && cd.stat) cd.stat[BS_STAT_ITEMS + d.range * BS_NSUB + tid] = 0;
  }
  if (!env_ok) return;
  if (has_dof) {
    // (an opaque copy of the row: the prologue's load addresses are formed again here instead of staying live — or in
    // scratch — through the whole kernel)
    int jo = j;
    asm volatile("" : "+v"(jo));
    AT(d.qpos, jo) = qpos; AT(d.qvel, jo) = qvel; AT(d.qacc, jo) = qacc; AT(d.qacc_warmstart, jo) = qacc_ws;
    if (was_reset) AT(d.qfrc_applied, jo) = qfrc_applied;
  }
  if (was_reset && act >= 0) AT(d.ctrl, act) = ctrl;
  if (l == 0) {
    AT(d.time, 0) = time; AT(d.warning, 0) = warning; AT(d.solver_iter, 0) = solver_iter; AT(d.nefc, 0) = nefc;
  }
}
