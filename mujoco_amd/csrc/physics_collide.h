// physics_collide.h — collision stage on the device.
//
// A5 broadphase: one wavefront per environment tests the model's compile-time candidate pair
// list (reference output/DexterousDynamos.xml:138-159 excludes, contype/conaffinity of
// src/Mujoco_XML.py:70 already applied) with bounding spheres and writes an ORDERED candidate
// list per environment using wavefront ballots + popcount prefix sums (deterministic order).
// A6 narrowphase: one lane per (environment, candidate): Minkowski Portal Refinement on the
// convex hulls (each inflated by margin/2), support by hill climbing over the hull vertex graph;
// plane-hull = deepest vertex.  Hull data (~270 KB fp64 for the 12 distinct meshes) stays in L2.
// Contact assembly: one lane per environment compacts the hits in candidate order.
//
// [EXT] restates MuJoCo 2.1.0's mjc_Convex (libccd MPR) / mjc_PlaneConvex from published
// descriptions; parity is against oracle/oracle_collide.c (PARITY UNPINNED vs MuJoCo).
#pragma once
#include "physics_dev.h"

#define BS_CCD_EPS 2.220446049250313e-16
enum { BS_GEOM_PLANE = 0, BS_GEOM_HFIELD = 1, BS_GEOM_SPHERE = 2, BS_GEOM_CAPSULE = 3, BS_GEOM_ELLIPSOID = 4,
       BS_GEOM_CYLINDER = 5, BS_GEOM_BOX = 6, BS_GEOM_MESH = 7 };
// a candidate slot names one potential contact: pair id in the low 24 bits, sub-contact in the high 8
// (plane-capsule: 2 cap centres, plane-box: 8 corners of which at most 4 make a contact, capsule-capsule: 2 for
// parallel axes, every other pair: 1)
#define BS_PAIR_MASK 0xFFFFFF
__host__ __device__ inline int pair_nsub_types(int t1, int t2) {
  if (t1 == BS_GEOM_PLANE && t2 == BS_GEOM_CAPSULE) return 2;
  if (t1 == BS_GEOM_PLANE && t2 == BS_GEOM_BOX) return 8;
  if (t1 == BS_GEOM_CAPSULE && t2 == BS_GEOM_CAPSULE) return 2;
  return 1;
}
__host__ __device__ inline bool pair_supported_types(int t1, int t2) {  // t1 <= t2 (model.py::_build_pairs)
  return t1 != BS_GEOM_HFIELD && t2 != BS_GEOM_HFIELD && t2 != BS_GEOM_PLANE;
}

#define BS_PAIR_CON 12
struct ColModel {
  int npair, ngeom, maxcand, maxcon, mpr_iterations;
  double mpr_tolerance;
  const int *pair_geom1, *pair_geom2, *geom_type, *geom_dataid, *geom_condim, *geom_bodyid;
  const double *geom_rbound, *geom_margin, *geom_gap, *geom_friction, *geom_solref, *geom_solimp,
      *geom_solmix, *hull_vert;
  const int *mesh_vertadr, *hull_adj_start, *hull_adj;
  const int *hull_start;  // [nmesh][BS_NSTART] support vertex for a cube-map of directions (climb warm start)
  // hill-climbing data packed for few, wide loads (divergent 8-byte loads are what bounds the
  // narrowphase): per adjacency entry {x, y, z, vertex id} of the neighbour, per vertex
  // {x, y, z, -} and its adjacency range {first, end}
  // record.w packs the named vertex: id (bits 0-19), degree (20-31), first adjacency entry (32-63)
  const double4 *hull_nbr4;  // [nhulladj]
  const double4 *hull_start4; // [nmesh][BS_NSTART] record of hull_start's vertex (work-stealing narrowphase)
  const double4 *hull_vert4; // [nhullvert]
  // support candidates of the persistent narrowphase: per bin the vertices that are the support vertex for SOME
  // direction of the bin, BS_CAND_K records {x, y, z, -} (first record's w: count | index of the rest in hull_more4 << 32)
  const double4 *hull_cand4;  // [nmesh][BS_NSTART][BS_CAND_K]
  const double4 *hull_more4;  // candidates BS_CAND_K.. of the bins that have more
  const int2 *hull_range;    // [nhullvert]
  const double *geom_obb; // [ngeom][6] box around the hull in the geom frame: centre, half extents (< 0: none)
  const double *geom_size; // [ngeom][3] mjModel.geom_size (primitives)
  const int *pair_nsub;    // [npair] sub-contacts a pair can produce (0: pair of unsupported geom types)
  // state-independent part of a pair's contact rows (host, model_image; k_tpc reads one record instead of three
  // dependent levels of per-geom loads): {margin - gap, mu, K, B, mixed solimp[5], diagApprox, 2 mu_e^2 (dim > 1), -}
  const double *pair_con;  // [npair][BS_PAIR_CON]
  const int *pair_body;    // [npair][2]: body of geom1 | condim << 16, body of geom2
  // what the persistent narrowphase needs about a pair in one 32-byte record (instead of pair -> geoms ->
  // mesh tables, three dependent levels of loads per item): geom1 | plane << 16, geom2, first vertex record and
  // start-table row of hull 1 and of hull 2; the pair's margin = max of the geoms' follows as a double
  const int *pair_bp;      // [npair][4]: g1 | g2 << 12 | kind << 24 (1 spheres, 2 plane) | obb << 26, 0, threshold of the bounding test (one double; spheres: squared)
  const int *pair_np;      // [npair][8]: g1 | plane << 16, g2, vbase1, sbase1, vbase2, sbase2, margin (2 ints = one double)
  int nl_min_share;        // least number of work-list items a wavefront of k_c_narrow_list takes
  int has_prim;            // some pair involves a primitive geom: the narrowphase kernels' PRIM instantiation runs
};

// cube-map direction bins: 6 faces x BS_SG x BS_SG
#ifndef BS_SG
// Round 3, with the per-bin support candidates (hull_cand4): finer bins mean fewer bins that straddle the normal cones of
// several vertices, i.e. fewer candidates to compare per support call and fewer calls that continue into the overflow
// list.  Contacts at 65 536 envs, (bins per face edge, records held in line): (16, 4) 38.3, (32, 4) 42.4, (64, 4) 44.9,
// (96, 4) 46.1, (128, 4) 46.8, (96, 2) 47.0, (128, 3) 47.7, (128, 2) 47.9, (192, 2) 48.2, (256, 2) 48.2 M env-steps/s.
// (128, 2): 6.3 MB of candidate records per hull, 151 MB per device for the hand's 24 hulls, 1.5 s to build.
// (Round 2, hill climbing from a start vertex per bin: 16 -> 28.5, 32 -> 29.3, 48 -> 29.6, 64 -> 29.8 M.)
#define BS_SG 128
#endif
#define BS_NSTART (6 * BS_SG * BS_SG)
#ifndef BS_CAND_K
#define BS_CAND_K 2      // candidate records per bin held in line (bins with more continue through hull_more4)
#endif
#define BS_CAND_MAX (1 << 20)  // no cap in practice (a flat cap's normal sits in one bin with the cap's whole rim)
__host__ __device__ inline int dir_bin(double x, double y, double z) {
  const double ax = x < 0 ? -x : x, ay = y < 0 ? -y : y, az = z < 0 ? -z : z;
  int face;
  double u, v, w;
  if (ax >= ay && ax >= az) { face = x < 0 ? 1 : 0; w = ax; u = y; v = z; }
  else if (ay >= az) { face = y < 0 ? 3 : 2; w = ay; u = x; v = z; }
  else { face = z < 0 ? 5 : 4; w = az; u = x; v = y; }
  if (!(w > 0)) return 0;
#ifdef __HIP_DEVICE_COMPILE__
  // any bin is a valid start for the climb, so the device divides by reciprocal (one v_rcp_f64 + two Newton steps
  // for both quotients instead of two IEEE division sequences per support call and hull)
  double iw = __builtin_amdgcn_rcp(w);
  iw = fma(fma(-w, iw, 1.0), iw, iw);
  iw = fma(fma(-w, iw, 1.0), iw, iw);
  int iu = (int)((u * iw * 0.5 + 0.5) * BS_SG), iv = (int)((v * iw * 0.5 + 0.5) * BS_SG);
#else
  int iu = (int)((u / w * 0.5 + 0.5) * BS_SG), iv = (int)((v / w * 0.5 + 0.5) * BS_SG);
#endif
  iu = iu < 0 ? 0 : (iu >= BS_SG ? BS_SG - 1 : iu);
  iv = iv < 0 ? 0 : (iv >= BS_SG ? BS_SG - 1 : iv);
  return (face * BS_SG + iu) * BS_SG + iv;
}
// centre direction of a bin (host: table construction)
inline void bin_dir(int bin, double* d) {
  const int iv = bin % BS_SG, iu = (bin / BS_SG) % BS_SG, face = bin / (BS_SG * BS_SG);
  const double u = ((iu + 0.5) / BS_SG - 0.5) * 2, v = ((iv + 0.5) / BS_SG - 0.5) * 2, s = (face & 1) ? -1.0 : 1.0;
  if (face < 2) { d[0] = s; d[1] = u; d[2] = v; }
  else if (face < 4) { d[0] = u; d[1] = s; d[2] = v; }
  else { d[0] = u; d[1] = v; d[2] = s; }
}

// a candidate as ONE 64-byte record, [E][maxcand] (tree-parallel path with the persistent narrowphase): the
// broadphase writes `pair`, the narrowphase the rest, k_tpc's compaction reads it — one cache line per candidate
// where the [row][E] arrays (the layout of the one-environment-per-lane kernels) touch nine
struct CandRec { int pair, hit; double dist, pos[3], nrm[3]; };
struct ColData {
  // geom frames (the sim's derived buffers)
  double *geom_xpos, *geom_xmat;
  // ... and, for the persistent narrowphase of the tree-parallel path, the same frames as one record per
  // environment and geom, [E][ngeom][12] = position | matrix (k_tp_col writes it INSTEAD of the arrays above):
  // an item reads two contiguous 96-byte records where the [row][E] arrays cost 24 cache lines
  double *gframe;
  // candidates [maxcand][E]
  int *ncand, *cand_pair, *cand_hit;
  CandRec *crec;  // non-null: candidates live here instead of cand_pair / cand_hit / cand_dist / cand_pos / cand_nrm
  // narrowphase work list of the tree-parallel path: {(env - env0) << 8 | candidate slot, pair | sub-contact << 24,
  // geom1 | geom2 << 16, 0} per item (16 bytes: what the narrowphase's set-up loads depend on, in one load).  The
  // broadphase blocks of a range append to BS_NSUB sub-lists (block index modulo BS_NSUB, so the
  // append counters stat[BS_STAT_ITEMS + range * BS_NSUB + sub] are not one hot address); sub-list
  // s of a range starts at col_items_base() + s * col_items_subcap()
  int4 *items;
  double *cand_dist, *cand_pos, *cand_nrm;
  // contacts [maxcon][E]
  int *ncon, *con_pair;
  double *con_dist, *con_pos, *con_frame;
  // statistics for the work model: narrowphase pairs and MPR support iterations
  unsigned long long *stat;  // [0] narrow pairs, [1] mpr iterations, [2] contacts
#define BS_STAT_ITEMS 32
#define BS_NSUB 64
#define BS_STAT_CON (BS_STAT_ITEMS + 4 * BS_NSUB)  // contact counter of the tree-parallel kernels, spread over BS_NSUB addresses
};
// capacity of one sub-list: the environments of ceil(blocks / BS_NSUB) broadphase blocks (2 each)
__host__ __device__ inline long col_items_subcap(long nenv, int maxcand) {
  return (((nenv + 1) / 2 + BS_NSUB - 1) / BS_NSUB) * 2 * (long)maxcand;
}
// first entry of a range's sub-lists: ranges are contiguous in env0, each may round up by 2 * BS_NSUB envs
__host__ __device__ inline long col_items_base(long env0, int range, int maxcand) {
  return (env0 + (long)range * 2 * BS_NSUB) * (long)maxcand;
}

__host__ __device__ __forceinline__ bool c_is_zero(double x) { return fabs(x) < BS_CCD_EPS; }
__host__ __device__ __forceinline__ bool c_is_eq(double a, double b) {
  const double ab = fabs(a - b);
  if (ab < BS_CCD_EPS) return true;
  return ab < BS_CCD_EPS * fmax(fabs(a), fabs(b));
}
__host__ __device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ __forceinline__ V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ __forceinline__ V3 sub3(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ __forceinline__ V3 neg(const V3& a) { return {-a.x, -a.y, -a.z}; }
// a / |a|: v_rsq_f64 refined by Goldschmidt rounds and three multiplications (12 + 3 VALU instructions; the IEEE
// sqrt and three IEEE divisions were ~100, and MPR normalises a direction or two per portal step).  Within an ulp
// or two of the IEEE quotient; squared lengths outside the normal range take the IEEE path.
__host__ __device__ __forceinline__ V3 normalized(const V3& a) {
  const double d = dot(a, a);
  if (d > 1e-280 && d < 1e280) {
#ifdef __HIP_DEVICE_COMPILE__
    const double y = __builtin_amdgcn_rsq(d);
#else
    const double y = 1.0 / sqrt(d);  // host compile (tests/host_mpr: the two forms of mpr_step against each other on the CPU)
#endif
    double g = d * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    h = fma(h, r, h);
    const double inv = h + h;
    return {a.x * inv, a.y * inv, a.z * inv};
  }
  const double n = sqrt(d);
  if (n > 0) return {a.x / n, a.y / n, a.z / n};
  return a;
}
__host__ __device__ __forceinline__ bool v_eq(const V3& a, const V3& b) {
  return c_is_eq(a.x, b.x) && c_is_eq(a.y, b.y) && c_is_eq(a.z, b.z);
}

__device__ __forceinline__ int rec_id(double w) { return __double2loint(w) & 0xFFFFF; }
__device__ __forceinline__ int rec_deg(double w) { return (__double2loint(w) >> 20) & 0xFFF; }
__device__ __forceinline__ int rec_k0(double w) { return __double2hiint(w); }

// converged pairs of the persistent narrowphase that run mpr_finish together (measured at 65 536 envs)
#ifndef NP_FIN_BATCH
#define NP_FIN_BATCH 6
#endif
#ifndef MD_BATCH
#define MD_BATCH 8  // neighbour records per trip of md_support's climb loop
#endif
struct CObj {
  int type;        // BS_GEOM_* (PRIM instantiations only)
  V3 size;         // geom_size (primitives)
  // vert4: this mesh's vertices, nbr4: global adjacency records, start4: this mesh's start table;
  // every record is {x, y, z, id | degree << 20 | first adjacency entry << 32}, so a climb step is
  // one dependent load (the neighbours of the vertex just chosen), not two
  const double4 *vert4, *nbr4, *start4;
  const double4 *vert_all, *start_all;  // the whole tables (uniform) and this mesh's first record in them:
  int vbase, sbase;                     // md_support addresses records with 32-bit offsets from a uniform base
  const double4 *cand_all, *more_all;   // support candidates per bin (ColModel::hull_cand4 / hull_more4), row sbase + bin
  int cache;
  V3 xpos;
  M3 R;
  double margin;
};
struct Spt { V3 v, v1, v2; };  // support point of the Minkowski difference as the support calls return it
// ... and as the MPR state keeps it: v2 = v1 - v is rebuilt where the contact position needs it (one rounding of
// a coordinate-sized number away from the stored v2: 1e-17 m), a third less LDS per lane
struct SptS { V3 v, v1; };

#ifdef NP_PROFILE
__device__ unsigned long long np_hist_climb[32], np_hist_sup[64];
#endif
__device__ inline V3 hull_support(CObj& o, const V3& dir) {
#ifdef NP_PROFILE
  int trips = 0;
#endif
  const M3& R = o.R;
  const V3 d = {R.a[0] * dir.x + R.a[3] * dir.y + R.a[6] * dir.z, R.a[1] * dir.x + R.a[4] * dir.y + R.a[7] * dir.z,
                R.a[2] * dir.x + R.a[5] * dir.y + R.a[8] * dir.z};
  int prev = -1;
  const double4 vc = o.vert4[o.cache], vt = o.start4[dir_bin(d.x, d.y, d.z)];
  V3 best = {vc.x, vc.y, vc.z};
  double vmax = best.x * d.x + best.y * d.y + best.z * d.z, rec = vc.w;
  {  // better of (previous support vertex, table vertex for this direction bin) as the start
    const double val = vt.x * d.x + vt.y * d.y + vt.z * d.z;
    if (val > vmax) { vmax = val; rec = vt.w; best = {vt.x, vt.y, vt.z}; }
  }
  int imax = rec_id(rec);
  // steepest ascent over the hull's vertex adjacency; four neighbour records are loaded at a
  // time so the loads overlap
  while (imax != prev) {
#ifdef NP_PROFILE
    trips++;
#endif
    prev = imax;
    const int k0 = rec_k0(rec), k1 = k0 + rec_deg(rec);
    for (int k = k0; k < k1; k += 4) {
      double4 p[4];
#pragma unroll
      for (int q = 0; q < 4; q++) p[q] = o.nbr4[k + q < k1 ? k + q : k1 - 1];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double val = p[q].x * d.x + p[q].y * d.y + p[q].z * d.z;
        if (val > vmax) { vmax = val; rec = p[q].w; imax = rec_id(rec); best = {p[q].x, p[q].y, p[q].z}; }
      }
    }
  }
  o.cache = imax;
#ifdef NP_PROFILE
  atomicAdd(&np_hist_climb[trips < 31 ? trips : 31], 1ull);
#endif
  const V3 w = mrot(R, best);
  return {o.xpos.x + w.x + o.margin * dir.x, o.xpos.y + w.y + o.margin * dir.y, o.xpos.z + w.z + o.margin * dir.z};
}
// support point of the Minkowski difference: the two hill climbs of hull_support (same start
// rule, same neighbour order, same comparisons) run in ONE loop — a lane climbs hull 1, then
// hull 2, one batch of four neighbour records per iteration — so a wavefront pays for the longest
// sum of the two climbs among its lanes, not for the longest climb on each hull
// only_b (per lane; plane items of the persistent narrowphase): hull 1 is not climbed — the lane enters
// the loop in its second phase — and the result's v2 is what hull_support(b, -dir) returns
// record `idx` of a table with a 32-bit byte offset: base (uniform) + zero-extended offset is one address
// operand of the load, where 64-bit index arithmetic costs four VALU instructions per record
__device__ __forceinline__ double4 ld_rec32(const double4* base, int idx) {
  return *(const double4*)((const char*)base + ((unsigned)idx << 5));
}
#ifdef NP_TIMING
// developer build: wave-level counters of the persistent narrowphase (k_narrow.hip adds them to ColData::stat[8..])
#define NPT(i, v) { np_t[i] += (unsigned long long)(v); }
#else
#define NPT(i, v)
#endif
__device__ inline Spt md_support(CObj& a, CObj& b, const V3& dir, bool only_b = false
#ifdef NP_TIMING
                                 , unsigned long long* np_t = nullptr
#endif
) {
  const V3 da = {a.R.a[0] * dir.x + a.R.a[3] * dir.y + a.R.a[6] * dir.z, a.R.a[1] * dir.x + a.R.a[4] * dir.y + a.R.a[7] * dir.z,
                 a.R.a[2] * dir.x + a.R.a[5] * dir.y + a.R.a[8] * dir.z};
  const V3 nd = neg(dir);
  const V3 db = {b.R.a[0] * nd.x + b.R.a[3] * nd.y + b.R.a[6] * nd.z, b.R.a[1] * nd.x + b.R.a[4] * nd.y + b.R.a[7] * nd.z,
                 b.R.a[2] * nd.x + b.R.a[5] * nd.y + b.R.a[8] * nd.z};
  // The support vertex of a hull for direction d is the best of the candidates of d's bin (ColModel::hull_cand4: every
  // vertex that is the support for some direction of the bin) — one round of loads for both hulls where the hill climb
  // took three dependent ones (start records, then the neighbours of each hull's start vertex to confirm it).  Bins
  // with more than BS_CAND_K candidates (2 % of the hand's) continue through their list, both hulls in the same trips.
  const int ra0 = (a.sbase + dir_bin(da.x, da.y, da.z)) * BS_CAND_K, rb0 = (b.sbase + dir_bin(db.x, db.y, db.z)) * BS_CAND_K;
  double4 pa[BS_CAND_K], pb[BS_CAND_K];
#pragma unroll
  for (int q = 0; q < BS_CAND_K; q++) { pa[q] = ld_rec32(a.cand_all, ra0 + q); pb[q] = ld_rec32(b.cand_all, rb0 + q); }
  V3 besta = {pa[0].x, pa[0].y, pa[0].z}, bestb = {pb[0].x, pb[0].y, pb[0].z};
  double vmaxa = besta.x * da.x + besta.y * da.y + besta.z * da.z, vmaxb = bestb.x * db.x + bestb.y * db.y + bestb.z * db.z;
#pragma unroll
  for (int q = 1; q < BS_CAND_K; q++) {
    const double va = pa[q].x * da.x + pa[q].y * da.y + pa[q].z * da.z;
    if (va > vmaxa) { vmaxa = va; besta = {pa[q].x, pa[q].y, pa[q].z}; }
    const double vb = pb[q].x * db.x + pb[q].y * db.y + pb[q].z * db.z;
    if (vb > vmaxb) { vmaxb = vb; bestb = {pb[q].x, pb[q].y, pb[q].z}; }
  }
  int na = only_b ? 0 : __double2loint(pa[0].w) - BS_CAND_K, nb = __double2loint(pb[0].w) - BS_CAND_K;  // candidates left
  int ka = __double2hiint(pa[0].w), kb = __double2hiint(pb[0].w);
  NPT(6, 1) NPT(7, __popcll(__ballot(1)))
  while (na > 0 || nb > 0) {
    NPT(6, 1) NPT(7, __popcll(__ballot(1)))
    // (a hull that is done re-reads its last records: the comparisons cannot change its vertex)
    const int la = na > 0 ? na : 0, lb = nb > 0 ? nb : 0;
#pragma unroll
    for (int q = 0; q < BS_CAND_K; q++) {
      pa[q] = ld_rec32(a.more_all, ka + (q < la ? q : 0));
      pb[q] = ld_rec32(b.more_all, kb + (q < lb ? q : 0));
    }
#pragma unroll
    for (int q = 0; q < BS_CAND_K; q++) {
      const double va = pa[q].x * da.x + pa[q].y * da.y + pa[q].z * da.z;
      if (q < la && va > vmaxa) { vmaxa = va; besta = {pa[q].x, pa[q].y, pa[q].z}; }
      const double vb = pb[q].x * db.x + pb[q].y * db.y + pb[q].z * db.z;
      if (q < lb && vb > vmaxb) { vmaxb = vb; bestb = {pb[q].x, pb[q].y, pb[q].z}; }
    }
    if (na > 0) { ka += BS_CAND_K; na -= BS_CAND_K; }
    if (nb > 0) { kb += BS_CAND_K; nb -= BS_CAND_K; }
  }
  Spt s;
  {
    const V3 w = mrot(a.R, besta);
    s.v1 = {a.xpos.x + w.x + a.margin * dir.x, a.xpos.y + w.y + a.margin * dir.y, a.xpos.z + w.z + a.margin * dir.z};
  }
  {
    const V3 w = mrot(b.R, bestb);
    s.v2 = {b.xpos.x + w.x + b.margin * nd.x, b.xpos.y + w.y + b.margin * nd.y, b.xpos.z + w.z + b.margin * nd.z};
  }
  s.v = sub(s.v1, s.v2);
  return s;
}
__host__ __device__ __forceinline__ V3 portal_dir(const SptS* P) {
  return normalized(cross(sub(P[2].v, P[1].v), sub(P[3].v, P[1].v)));
}
__host__ __device__ __forceinline__ bool reach_tol(const SptS* P, const SptS& v4, const V3& dir, double tol) {
  const double dv4 = dot(v4.v, dir);
  const double m = fmin(fmin(dv4 - dot(P[1].v, dir), dv4 - dot(P[2].v, dir)), dv4 - dot(P[3].v, dir));
  return c_is_eq(m, tol) || m < tol;
}
__host__ __device__ __forceinline__ void expand_portal(SptS* P, const V3& p0, const SptS& v4) {
  const V3 v4v0 = cross(v4.v, p0);
  if (dot(P[1].v, v4v0) > 0) {
    if (dot(P[2].v, v4v0) > 0) P[1] = v4;
    else P[3] = v4;
  } else {
    if (dot(P[3].v, v4v0) > 0) P[2] = v4;
    else P[1] = v4;
  }
}
__host__ __device__ inline double origin_tri_dist2(const V3& a, const V3& b, const V3& c, V3& w) {
  const V3 ab = sub(b, a), ac = sub(c, a), ap = neg(a);
  const double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) { w = a; return dot(w, w); }
  const V3 bp = neg(b);
  const double d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) { w = b; return dot(w, w); }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    const double v = d1 / (d1 - d3);
    w = {a.x + v * ab.x, a.y + v * ab.y, a.z + v * ab.z};
    return dot(w, w);
  }
  const V3 cp = neg(c);
  const double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) { w = c; return dot(w, w); }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    const double v = d2 / (d2 - d6);
    w = {a.x + v * ac.x, a.y + v * ac.y, a.z + v * ac.z};
    return dot(w, w);
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    w = {b.x + v * (c.x - b.x), b.y + v * (c.y - b.y), b.z + v * (c.z - b.z)};
    return dot(w, w);
  }
  const double den = 1.0 / (va + vb + vc), v = vb * den, u = vc * den;
  w = {a.x + ab.x * v + ac.x * u, a.y + ab.y * v + ac.y * u, a.z + ab.z * v + ac.z * u};
  return dot(w, w);
}

// Minkowski Portal Refinement, one transition per support point.  The textbook phases (discover
// portal, refine until the portal encloses the origin ray, push the portal to the surface) are the
// values of `state`; the caller owns the loop around the support call, so the lanes of a
// wavefront stay converged on the expensive part whatever phase each one is in.
__host__ __device__ __forceinline__ void swap3(V3& a, V3& b) {
  const double x = a.x, y = a.y, z = a.z;
  a.x = b.x; a.y = b.y; a.z = b.z;
  b.x = x; b.y = y; b.z = z;
}
__host__ __device__ __forceinline__ void swap_spt(SptS& a, SptS& b) { swap3(a.v, b.v); swap3(a.v1, b.v1); }
struct MprState {
  V3 p0;     // portal point 0 = centre difference (its two support points are the geom centres)
  SptS P[3]; // portal points 1..3
  V3 d;           // direction of the next support query
  int pk;         // state (bits 0-2) | guard (3-10) | it (11-): one word, so a lane's slot is 200 bytes
};
__host__ __device__ __forceinline__ void mpr_begin(MprState& M, const V3& c1, const V3& c2) {
  const V3 origin = {0, 0, 0};
  M.p0 = sub(c1, c2);
  if (v_eq(M.p0, origin)) M.p0 = {BS_CCD_EPS * 10, 0, 0};
  M.d = normalized(neg(M.p0));
  M.pk = 0;
}
// Last part of the refinement: penetration depth, direction and contact position from the final portal.  A function of
// its own so that the persistent narrowphase can run it for a BATCH of lanes: inside mpr_step it executed on almost every
// trip of the support loop for the two or three lanes whose pair had just converged (~350 instructions at 3 of 64 lanes:
// the largest single loss of the kernel, round 3).
__host__ __device__ __forceinline__ void mpr_finish(const MprState& M, const V3& c1, const V3& c2, double& depth, V3& dir, V3& pos) {
  const SptS* P = M.P - 1;  // P[1..3]
  const V3& p0 = M.p0;
  const V3& d = M.d;
  const V3 origin = {0, 0, 0};
  V3 w;
  depth = sqrt(origin_tri_dist2(P[1].v, P[2].v, P[3].v, w));
  dir = c_is_zero(depth) ? origin : normalized(w);
  // position from barycentric coordinates of the origin in the portal tetrahedron
  double bb[4];
  bb[0] = dot(cross(P[1].v, P[2].v), P[3].v);
  bb[1] = dot(cross(P[3].v, P[2].v), p0);
  bb[2] = dot(cross(p0, P[1].v), P[3].v);
  bb[3] = dot(cross(P[2].v, P[1].v), p0);
  double sum = bb[0] + bb[1] + bb[2] + bb[3];
  if (c_is_zero(sum) || sum < 0) {
    bb[0] = 0;
    bb[1] = dot(cross(P[2].v, P[3].v), d);
    bb[2] = dot(cross(P[3].v, P[1].v), d);
    bb[3] = dot(cross(P[1].v, P[2].v), d);
    sum = bb[1] + bb[2] + bb[3];
  }
  const double inv = 1.0 / sum;
  V3 p1 = {0 + bb[0] * c1.x, 0 + bb[0] * c1.y, 0 + bb[0] * c1.z}, p2 = {0 + bb[0] * c2.x, 0 + bb[0] * c2.y, 0 + bb[0] * c2.z};
#pragma unroll
  for (int i = 1; i < 4; i++) {
    p1.x += bb[i] * P[i].v1.x; p1.y += bb[i] * P[i].v1.y; p1.z += bb[i] * P[i].v1.z;
    p2.x += bb[i] * (P[i].v1.x - P[i].v.x); p2.y += bb[i] * (P[i].v1.y - P[i].v.y); p2.z += bb[i] * (P[i].v1.z - P[i].v.z);
  }
  pos = {0.5 * (p1.x * inv + p2.x * inv), 0.5 * (p1.y * inv + p2.y * inv), 0.5 * (p1.z * inv + p2.z * inv)};
}
// consumes the support point `s` of direction M.d; returns 0 = query M.d again, 1 = the inflated
// hulls intersect (depth, dir = unit normal obj1 -> obj2, pos filled), 2 = they do not
// Two forms of the step, same arithmetic per lane: the default is ONE instruction stream for all portal phases (round 3);
// -DMPR_BRANCHY keeps the form of rounds 1-2 (one branch per phase).  tests/test_mpr_forms.py compiles both for the host
// (tests/host_mpr/) and checks that they agree bit for bit, call by call; on the GPU every contact test passes with
// either (89 of 89), the unified form is 1.7 % faster on the contact step (49.6 against 48.8 M env-steps/s at 65 536 envs)
// and needs 158 VGPRs where the branchy one needs 182.
#ifndef MPR_BRANCHY
// DEFER: 3 = the portal has converged, the caller owes the pair a call of mpr_finish
// Round 3: ONE instruction stream for all phases.  The phases used to be four branches — store the support point,
// test it, form the next direction — that a wavefront with lanes in every phase (the usual case in the persistent
// narrowphase) executed one after the other, each with its own LDS round trips and its own normalisation chain; the
// kernel is latency-bound (~24 000 cycles per trip of the support loop for ~1 000 instructions), so the four serial
// chains were the cost.  Every lane now reads the portal once, evaluates the tests of its phase next to the other
// phases' (unused results are discarded), stores the support point into the slot its phase selects, and forms the next
// direction with ONE cross product and ONE normalisation whose operands are selected by phase:
//   state 0:             d = p0 x P1                  = (p0 - 0) x (P1 - 0)
//   state 1, 2 (again):  d = (P1 - p0) x (P2 - p0)
//   state 2 (done), 3, 4: d = (P2 - P1) x (P3 - P1)   (portal_dir)
// Per lane the arithmetic is the branches' (x - 0 is x, bit for bit), so the contact sets do not move.
template <bool DEFER = false>
__host__ __device__ __forceinline__ int mpr_step(MprState& M, const Spt& sfull, const V3& c1, const V3& c2, double tol, int maxit,
                                        double& depth, V3& dir, V3& pos, int& iters) {
  // (portal points 1..3 are M.P[0..2]; no `M.P - 1` here: with a run-time index into a state held in private memory — the
  // work-stealing kernel — the compiler put the NEGATIVE base offset into the scratch instruction's unsigned SGPR operand,
  // an out-of-range address: the GPU fault of this form's first run)
  const V3 sv = sfull.v, sv1 = sfull.v1;
  const V3 p0 = M.p0, d = M.d;
  V3 P1v = M.P[0].v, P2v = M.P[1].v, P3v = M.P[2].v;  // (phases 0 / 1: what the slots still hold of the lane's last pair — unused)
  int state = M.pk & 7, guard = (M.pk >> 3) & 255, it = M.pk >> 11;
  const V3 origin = {0, 0, 0};
  // ---- the support point against the query direction
  const double dt = dot(sv, d);
  const bool refine = state >= 3;
  if (refine) iters++;
  const double m3 = fmin(fmin(dt - dot(P1v, d), dt - dot(P2v, d)), dt - dot(P3v, d));
  const bool rt = c_is_eq(m3, tol) || m3 < tol;  // reach_tol(P, s, d, tol)
  if (!refine) {
    if (c_is_zero(dt) || dt < 0) return 2;
  } else if (state == 3) {
    if (!(c_is_zero(dt) || dt > 0) || rt) return 2;
  } else if (rt || it > maxit) {
    // the portal has reached the surface: depth, normal and position (mpr_finish).  DEFER: the caller runs
    // mpr_finish itself, for several lanes at once — M is complete (the support point of this call is not used)
    if (DEFER) return 3;
    mpr_finish(M, c1, c2, depth, dir, pos);
    return 1;
  }
  // ---- which portal point the support point replaces
  //   state 2 (discovery): point 2 or 1 while the portal does not enclose the origin ray, else point 3
  const double x1 = dot(cross(P1v, sv), p0), x2 = dot(cross(sv, P2v), p0);
  const bool c1s = x1 < 0 && !c_is_zero(x1), c2s = x2 < 0 && !c_is_zero(x2);
  const bool cont = state == 2 && (c1s || c2s);
  //   state 3 / 4: expand_portal
  const V3 v4v0 = cross(sv, p0);
  const bool t1 = dot(P1v, v4v0) > 0, t2 = dot(P2v, v4v0) > 0, t3 = dot(P3v, v4v0) > 0;
  const int tgt34 = t1 ? (t2 ? 1 : 3) : (t3 ? 2 : 1);
  const int tgt2 = c1s ? 2 : (c2s ? 1 : 3);
  const int tgt = state == 0 ? 1 : (state == 1 ? 2 : (state == 2 ? tgt2 : tgt34));
  M.P[tgt - 1] = SptS{sv, sv1};
  if (tgt == 1) P1v = sv;
  if (tgt == 2) P2v = sv;
  if (tgt == 3) P3v = sv;
  if (refine) it++;
  if (cont && ++guard > 100) return 2;
  if (state == 3 && it > maxit + 1) return 2;
  // ---- next query direction
  const bool clsX = state == 0, clsY = state == 1 || cont;
  const V3 A = clsX ? p0 : (clsY ? P1v : P2v), B = clsX ? P1v : (clsY ? P2v : P3v), C = clsX ? origin : (clsY ? p0 : P1v);
  const V3 raw = cross(sub(A, C), sub(B, C));
  if (clsX && c_is_zero(dot(raw, raw))) {
    pos = {0.5 * (sfull.v1.x + sfull.v2.x), 0.5 * (sfull.v1.y + sfull.v2.y), 0.5 * (sfull.v1.z + sfull.v2.z)};
    if (v_eq(sv, origin)) { depth = 0; dir = origin; return 1; }  // touching
    depth = sqrt(dot(sv, sv));
    dir = normalized(sv);
    return 1;  // origin on the segment v0-v1
  }
  V3 dn = normalized(raw);
  if (state == 1) {
    if (dot(dn, p0) > 0) {
      swap_spt(M.P[0], M.P[1]);
      dn = neg(dn);
    }
    guard = 1;
    state = 2;
  } else if (state == 0) {
    state = 1;
  } else if (!cont && state != 4) {
    // discovery finished (state 2) or refinement (state 3): does the portal enclose the origin ray?
    if (state == 2) it = 0;
    const double x = dot(dn, P1v);
    if (c_is_zero(x) || x > 0) { it = 0; state = 4; }
    else state = 3;
  }
  M.d = dn;
  M.pk = state | (guard << 3) | (it << 11);
  return 0;
}
#else
// DEFER: 3 = the portal has converged, the caller owes the pair a call of mpr_finish
template <bool DEFER = false>
__host__ __device__ __forceinline__ int mpr_step(MprState& M, const Spt& sfull, const V3& c1, const V3& c2, double tol, int maxit,
                                        double& depth, V3& dir, V3& pos, int& iters) {
  SptS* P = M.P - 1;  // P[1..3]
  const SptS s = {sfull.v, sfull.v1};
  const V3& p0 = M.p0;
  V3& d = M.d;
  int state = M.pk & 7, guard = (M.pk >> 3) & 255, it = M.pk >> 11;
  const V3 origin = {0, 0, 0};
  bool check_refine = false;
  if (state == 0) {
    P[1] = s;
    const double dt = dot(P[1].v, d);
    if (c_is_zero(dt) || dt < 0) return 2;
    d = cross(p0, P[1].v);
    if (c_is_zero(dot(d, d))) {
      pos = {0.5 * (sfull.v1.x + sfull.v2.x), 0.5 * (sfull.v1.y + sfull.v2.y), 0.5 * (sfull.v1.z + sfull.v2.z)};
      if (v_eq(P[1].v, origin)) { depth = 0; dir = origin; return 1; }  // touching
      depth = sqrt(dot(P[1].v, P[1].v));
      dir = normalized(P[1].v);
      return 1;  // origin on the segment v0-v1
    }
    d = normalized(d);
    state = 1;
  } else if (state == 1) {
    P[2] = s;
    const double dt = dot(P[2].v, d);
    if (c_is_zero(dt) || dt < 0) return 2;
    d = normalized(cross(sub(P[1].v, p0), sub(P[2].v, p0)));
    if (dot(d, p0) > 0) {
      swap_spt(P[1], P[2]);
      d = neg(d);
    }
    guard = 1;
    state = 2;
  } else if (state == 2) {  // portal discovery
    P[3] = s;
    const double dt = dot(P[3].v, d);
    if (c_is_zero(dt) || dt < 0) return 2;
    bool cont = false;
    double x = dot(cross(P[1].v, P[3].v), p0);
    if (x < 0 && !c_is_zero(x)) { P[2] = P[3]; cont = true; }
    if (!cont) {
      x = dot(cross(P[3].v, P[2].v), p0);
      if (x < 0 && !c_is_zero(x)) { P[1] = P[3]; cont = true; }
    }
    if (cont) {
      d = normalized(cross(sub(P[1].v, p0), sub(P[2].v, p0)));
      if (++guard > 100) return 2;
    } else {
      it = 0;
      check_refine = true;
    }
  } else {
    // refinement (state 3) and penetration (state 4: push the portal to the surface) share their arithmetic — the
    // tolerance test, the portal expansion, the new portal direction — so the lanes of a wavefront that are in
    // either state run ONE copy of it together instead of two copies one after the other
    iters++;
    const bool rt = reach_tol(P, s, d, tol);
    if (state == 3) {
      const double y = dot(s.v, d);
      if (!(c_is_zero(y) || y > 0) || rt) return 2;
    } else if (rt || it > maxit) {
      // the portal has reached the surface: depth, normal and position (mpr_finish).  DEFER: the caller runs
      // mpr_finish itself, for several lanes at once — M is complete (the support point of this call is not used)
      if (DEFER) return 3;
      mpr_finish(M, c1, c2, depth, dir, pos);
      return 1;
    }
    expand_portal(P, p0, s);
    it++;
    if (state == 3) {
      if (it > maxit + 1) return 2;
      d = portal_dir(P);
      const double x = dot(d, P[1].v);
      if (c_is_zero(x) || x > 0) { it = 0; state = 4; }  // portal encloses the origin ray
    } else {
      d = portal_dir(P);
    }
  }

  if (check_refine) {
    if (it > maxit + 1) return 2;
    d = portal_dir(P);
    const double x = dot(d, P[1].v);
    if (c_is_zero(x) || x > 0) { it = 0; state = 4; }  // portal encloses the origin ray
    else state = 3;
  }
  M.pk = state | (guard << 3) | (it << 11);
  return 0;
}
#endif
// returns true when the inflated hulls intersect; fills depth, dir (unit, obj1 -> obj2), pos
// `Mp`: where the state lives.  The portal update copies structs through selected pointers, so the
// state is a memory object whatever one does; the kernels hand in a slot of LDS (one per lane,
// 81-dword stride: conflict-free), which costs a fraction of the scratch-memory round trips.
__device__ inline bool mpr_penetration(CObj& a, CObj& b, double tol, int maxit, double& depth, V3& dir, V3& pos,
                                       int& iters, MprState& M) {
  mpr_begin(M, a.xpos, b.xpos);
#ifdef NP_PROFILE
  int nsup = 0;
#endif
  for (;;) {
    const Spt s = md_support(a, b, M.d);
    const int r = mpr_step(M, s, a.xpos, b.xpos, tol, maxit, depth, dir, pos, iters);
#ifdef NP_PROFILE
    nsup++;
    if (r) atomicAdd(&np_hist_sup[nsup < 63 ? nsup : 63], 1ull);
#endif
    if (r) return r == 1;
  }
}

// ---- primitive geoms (north_star "plus plane/primitive pairs"; same closed forms and support
// functions as oracle/oracle_collide.c).  Ties (a direction component of exactly 0) take the positive side.
__device__ inline V3 geom_support_any(CObj& o, const V3& dir) {
  if (o.type == BS_GEOM_MESH) return hull_support(o, dir);
  const M3& R = o.R;
  const V3 d = {R.a[0] * dir.x + R.a[3] * dir.y + R.a[6] * dir.z, R.a[1] * dir.x + R.a[4] * dir.y + R.a[7] * dir.z,
                R.a[2] * dir.x + R.a[5] * dir.y + R.a[8] * dir.z};
  V3 v = {0, 0, 0};
  double infl = o.margin;
  switch (o.type) {
    case BS_GEOM_SPHERE: infl += o.size.x; break;
    case BS_GEOM_CAPSULE: v.z = d.z >= 0 ? o.size.y : -o.size.y; infl += o.size.x; break;
    case BS_GEOM_CYLINDER: {
      const double l = sqrt(d.x * d.x + d.y * d.y);
      if (l > 0) { v.x = o.size.x * d.x / l; v.y = o.size.x * d.y / l; }
      v.z = d.z >= 0 ? o.size.y : -o.size.y;
      break;
    }
    case BS_GEOM_BOX:
      v = {d.x >= 0 ? o.size.x : -o.size.x, d.y >= 0 ? o.size.y : -o.size.y, d.z >= 0 ? o.size.z : -o.size.z};
      break;
    case BS_GEOM_ELLIPSOID: {
      const double s0 = o.size.x * d.x, s1 = o.size.y * d.y, s2 = o.size.z * d.z, l = sqrt(s0 * s0 + s1 * s1 + s2 * s2);
      if (l > 0) v = {o.size.x * s0 / l, o.size.y * s1 / l, o.size.z * s2 / l};
      break;
    }
    default: break;
  }
  const V3 w = mrot(R, v);
  return {o.xpos.x + w.x + infl * dir.x, o.xpos.y + w.y + infl * dir.y, o.xpos.z + w.z + infl * dir.z};
}
// plane (point pp, normal n) against a point c of radius r: plane-capsule caps (r = radius), plane-box corners (r = 0)
__device__ inline bool plane_point(const V3& pp, const V3& n, const V3& c, double r, double margin, double& dist, V3& pos) {
  const double ds = dot(sub(c, pp), n) - r;
  if (ds > margin) return false;
  dist = ds;
  const double k = r + 0.5 * ds;
  pos = {c.x - k * n.x, c.y - k * n.y, c.z - k * n.z};
  return true;
}
__device__ inline bool sphere_sphere(const V3& c1, double r1, const V3& c2, double r2, double margin, double& dist, V3& pos,
                                     V3& nrm) {
  const V3 df = sub(c2, c1);
  const double len = sqrt(dot(df, df));
  if (len - r1 - r2 > margin) return false;
  if (len < 1e-15) nrm = {1, 0, 0};
  else nrm = {df.x / len, df.y / len, df.z / len};
  dist = len - r1 - r2;
  const double k = r1 + 0.5 * dist;
  pos = {c1.x + k * nrm.x, c1.y + k * nrm.y, c1.z + k * nrm.z};
  return true;
}
__device__ inline V3 seg_closest_point(const V3& c, const V3& axis, double h, const V3& p) {
  double t = dot(sub(p, c), axis);
  t = t > h ? h : (t < -h ? -h : t);
  return {c.x + t * axis.x, c.y + t * axis.y, c.z + t * axis.z};
}
__device__ inline void seg_seg_closest(const V3& c1, const V3& a1, double h1, const V3& c2, const V3& a2, double h2, V3& p1,
                                       V3& p2) {
  const V3 w = sub(c1, c2);
  const double b = dot(a1, a2), d1 = dot(a1, w), d2 = dot(a2, w), den = 1 - b * b;
  double t1, t2;
  if (den > 1e-12) {
    t1 = (b * d2 - d1) / den;
    t1 = t1 > h1 ? h1 : (t1 < -h1 ? -h1 : t1);
  } else t1 = 0;  // parallel axes: start from the middle of segment 1
  t2 = b * t1 + d2;
  if (t2 > h2 || t2 < -h2) {
    t2 = t2 > h2 ? h2 : -h2;
    t1 = b * t2 - d1;
    t1 = t1 > h1 ? h1 : (t1 < -h1 ? -h1 : t1);
  }
  p1 = {c1.x + t1 * a1.x, c1.y + t1 * a1.y, c1.z + t1 * a1.z};
  p2 = {c2.x + t2 * a2.x, c2.y + t2 * a2.y, c2.z + t2 * a2.z};
}

// capsule - capsule, sub-contact `sub` (same closed form as oracle_collide.c::capsule_capsule): one contact for axes in
// general position; for (near-)parallel axes up to two — the ends of each segment against the closest point of the
// other, in the order +end / -end of capsule 1, then of capsule 2, until two have been found
__device__ inline bool capsule_capsule(const V3& c1, const V3& a1, double h1, double r1, const V3& c2, const V3& a2, double h2,
                                       double r2, double margin, int sub, double& dist, V3& pos, V3& nrm) {
  const V3 w = sub3(c1, c2);
  const double b = dot(a1, a2), d1 = dot(a1, w), d2 = dot(a2, w), den = 1 - b * b;
  if (den > 1e-12) {
    if (sub != 0) return false;
    V3 p1, p2;
    seg_seg_closest(c1, a1, h1, c2, a2, h2, p1, p2);
    return sphere_sphere(p1, r1, p2, r2, margin, dist, pos, nrm);
  }
  int found = 0;
  for (int k = 0; k < 4 && found < 2; k++) {
    double t1, t2;
    if (k < 2) {
      t1 = k == 0 ? h1 : -h1;
      t2 = d2 + b * t1;
      t2 = t2 > h2 ? h2 : (t2 < -h2 ? -h2 : t2);
    } else {
      t2 = k == 2 ? h2 : -h2;
      t1 = b * t2 - d1;
      t1 = t1 > h1 ? h1 : (t1 < -h1 ? -h1 : t1);
    }
    const V3 p1 = {c1.x + t1 * a1.x, c1.y + t1 * a1.y, c1.z + t1 * a1.z}, p2 = {c2.x + t2 * a2.x, c2.y + t2 * a2.y, c2.z + t2 * a2.z};
    if (!sphere_sphere(p1, r1, p2, r2, margin, dist, pos, nrm)) continue;
    if (found == sub) return true;
    found++;
  }
  return false;
}

// ---------------------------------------------------------------- kernels' bodies
// geom frames for one environment lane (same arithmetic as k_frames, geoms only)
__device__ inline void col_geom_frames(const DevModel& m, const DevData& d, const ColData& c, long e) {
  const long E = d.E;
  for (int g = 0; g < m.ngeom; g++) {
    const int b = m.geom_bodyid[g];
    const Q4 q = {AT(d.xquat, 4 * b), AT(d.xquat, 4 * b + 1), AT(d.xquat, 4 * b + 2), AT(d.xquat, 4 * b + 3)};
    const V3 p = mrot(q2m(q), ld3(m.geom_pos + 3 * g));
    AT(c.geom_xpos, 3 * g) = AT(d.xpos, 3 * b) + p.x;
    AT(c.geom_xpos, 3 * g + 1) = AT(d.xpos, 3 * b + 1) + p.y;
    AT(c.geom_xpos, 3 * g + 2) = AT(d.xpos, 3 * b + 2) + p.z;
    const M3 G = q2m(qmul(q, ld4(m.geom_quat + 4 * g)));
    for (int k = 0; k < 9; k++) AT(c.geom_xmat, 9 * g + k) = G.a[k];
  }
}

// conservative oriented-box filter (15-axis separating-axis test) between the hull boxes of two
// mesh geoms, each inflated by margin/2: true only if the inflated hulls cannot touch.  It removes
// narrowphase work only; the contact set is the narrowphase's either way.
__device__ inline bool obb_separated_m(const double* oa, const double* ob, const double* A, const double* B,
                                       const V3& x1, const V3& x2, double margin);
__device__ inline bool obb_separated(const ColModel& cm, const DevData& d, const ColData& c, long e, int g1, int g2,
                                     const V3& x1, const V3& x2, double margin) {
  const long E = d.E;
  double A[9], B[9];
  for (int k = 0; k < 9; k++) { A[k] = AT(c.geom_xmat, 9 * g1 + k); B[k] = AT(c.geom_xmat, 9 * g2 + k); }
  return obb_separated_m(cm.geom_obb + 6 * g1, cm.geom_obb + 6 * g2, A, B, x1, x2, margin);
}
// oa/ob: box centre + half extents in the geom frame; A/B: geom rotation matrices (row-major)
__device__ inline bool obb_separated_m(const double* oa, const double* ob, const double* A, const double* B,
                                       const V3& x1, const V3& x2, double margin) {
  const double infl = 0.5 * margin + 1e-9;
  const double ha[3] = {oa[3] + infl, oa[4] + infl, oa[5] + infl}, hb[3] = {ob[3] + infl, ob[4] + infl, ob[5] + infl};
  // world centres and their difference in A's frame (columns of the geom matrices are the box axes)
  const V3 ca = {x1.x + A[0] * oa[0] + A[1] * oa[1] + A[2] * oa[2], x1.y + A[3] * oa[0] + A[4] * oa[1] + A[5] * oa[2],
                 x1.z + A[6] * oa[0] + A[7] * oa[1] + A[8] * oa[2]};
  const V3 cb = {x2.x + B[0] * ob[0] + B[1] * ob[1] + B[2] * ob[2], x2.y + B[3] * ob[0] + B[4] * ob[1] + B[5] * ob[2],
                 x2.z + B[6] * ob[0] + B[7] * ob[1] + B[8] * ob[2]};
  const V3 tw = sub(cb, ca);
  double R[3][3], Q[3][3], t[3];
  for (int i = 0; i < 3; i++) {
    t[i] = tw.x * A[i] + tw.y * A[3 + i] + tw.z * A[6 + i];
    for (int j = 0; j < 3; j++) {
      R[i][j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
      Q[i][j] = fabs(R[i][j]) + 1e-9;
    }
  }
  for (int i = 0; i < 3; i++)
    if (fabs(t[i]) > ha[i] + hb[0] * Q[i][0] + hb[1] * Q[i][1] + hb[2] * Q[i][2]) return true;
  for (int j = 0; j < 3; j++)
    if (fabs(t[0] * R[0][j] + t[1] * R[1][j] + t[2] * R[2][j]) > hb[j] + ha[0] * Q[0][j] + ha[1] * Q[1][j] + ha[2] * Q[2][j])
      return true;
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const double ra = ha[i1] * Q[i2][j] + ha[i2] * Q[i1][j], rb = hb[j1] * Q[i][j2] + hb[j2] * Q[i][j1];
      if (fabs(t[i2] * R[i1][j] - t[i1] * R[i2][j]) > ra + rb) return true;
    }
  }
  return false;
}

// A5: one 64-lane wavefront per environment; ordered compaction with ballot + popcount
__device__ inline void col_broadphase(const ColModel& cm, const DevData& d, const ColData& c, long e, int lane) {
  const long E = d.E;
  int count = 0;
  for (int p0 = 0; p0 < cm.npair; p0 += 64) {
    const int p = p0 + lane;
    bool keep = false;
    if (p < cm.npair) {
      const int g1 = cm.pair_geom1[p], g2 = cm.pair_geom2[p];
      const double margin = fmax(cm.geom_margin[g1], cm.geom_margin[g2]);
      const double r1 = cm.geom_rbound[g1], r2 = cm.geom_rbound[g2];
      const V3 x1 = {AT(c.geom_xpos, 3 * g1), AT(c.geom_xpos, 3 * g1 + 1), AT(c.geom_xpos, 3 * g1 + 2)};
      const V3 x2 = {AT(c.geom_xpos, 3 * g2), AT(c.geom_xpos, 3 * g2 + 1), AT(c.geom_xpos, 3 * g2 + 2)};
      keep = true;
      if (r1 > 0 && r2 > 0) {
        const V3 df = sub(x1, x2);
        keep = !(sqrt(dot(df, df)) > margin + r1 + r2);
        if (keep && cm.geom_obb[6 * g1 + 3] >= 0 && cm.geom_obb[6 * g2 + 3] >= 0)
          keep = !obb_separated(cm, d, c, e, g1, g2, x1, x2, margin);
      } else if (cm.geom_type[g1] == BS_GEOM_PLANE && r2 > 0) {
        const V3 n = {AT(c.geom_xmat, 9 * g1 + 2), AT(c.geom_xmat, 9 * g1 + 5), AT(c.geom_xmat, 9 * g1 + 8)};
        keep = !(dot(sub(x2, x1), n) > margin + r2);
      }
    }
    // a pair reserves pair_nsub consecutive slots (0 for geom types no routine handles): ordered
    // compaction with a prefix sum of the counts (1, 2 or 8) from three ballots
    const int cnt = keep ? cm.pair_nsub[p] : 0;
    const unsigned long long lower = (1ull << lane) - 1ull;
    const unsigned long long m1 = __ballot(cnt & 1), m2 = __ballot(cnt & 2), m8 = __ballot(cnt & 8);
    const int pos = count + __popcll(m1 & lower) + 2 * __popcll(m2 & lower) + 8 * __popcll(m8 & lower);
    for (int q = 0; q < cnt; q++)
      if (pos + q < cm.maxcand) AT(c.cand_pair, pos + q) = p | (q << 24);
    count += __popcll(m1) + 2 * __popcll(m2) + 8 * __popcll(m8);
  }
  if (lane == 0) {
    if (count > cm.maxcand) { count = cm.maxcand; AT(d.warning, 0) |= 8; }
    AT(c.ncand, 0) = count;
  }
}

// A6: one lane per (environment, candidate slot)
// `acc` (optional): per-lane work counters {pairs, MPR support iterations}; the kernel reduces them
// over the wavefront before touching ColData::stat (one atomic per wavefront, not per pair)
template <bool PRIM = false>
__device__ inline void col_narrowphase(const ColModel& cm, const DevData& d, const ColData& c, long e, int slot,
                                       unsigned* acc, MprState& M) {
  const long E = d.E;
  if (slot >= AT(c.ncand, 0)) return;
  const int ps = AT(c.cand_pair, slot);
  const int p = ps & BS_PAIR_MASK;
  const int g1 = cm.pair_geom1[p], g2 = cm.pair_geom2[p];
  const double margin = fmax(cm.geom_margin[g1], cm.geom_margin[g2]);
  bool hit = false;
  double dist = 0;
  V3 pos = {0, 0, 0}, nrm = {0, 0, 0};
  auto load_obj = [&](CObj& o, int g) {
    o.type = BS_GEOM_MESH; o.size = {0, 0, 0};
    if constexpr (PRIM) { o.type = cm.geom_type[g]; o.size = ld3(cm.geom_size + 3 * g); }
    const int mesh = (!PRIM || o.type == BS_GEOM_MESH) ? cm.geom_dataid[g] : 0;
    o.vert4 = cm.hull_vert4 + cm.mesh_vertadr[mesh];
    o.nbr4 = cm.hull_nbr4; o.cache = 0;
    o.start4 = cm.hull_start4 + (size_t)mesh * BS_NSTART;
    o.vert_all = cm.hull_vert4; o.start_all = cm.hull_start4; o.vbase = cm.mesh_vertadr[mesh]; o.sbase = mesh * BS_NSTART;
    o.cand_all = cm.hull_cand4; o.more_all = cm.hull_more4;
    o.xpos = {AT(c.geom_xpos, 3 * g), AT(c.geom_xpos, 3 * g + 1), AT(c.geom_xpos, 3 * g + 2)};
    for (int k = 0; k < 9; k++) o.R.a[k] = AT(c.geom_xmat, 9 * g + k);
    o.margin = 0;
  };
  CObj b;
  load_obj(b, g2);
  const int t1 = cm.geom_type[g1];
  bool done = false;
  if constexpr (PRIM) {
    const int t2 = b.type, sub = ps >> 24;
    if (t1 == BS_GEOM_PLANE && (t2 == BS_GEOM_CAPSULE || t2 == BS_GEOM_BOX)) {
      const V3 n = {AT(c.geom_xmat, 9 * g1 + 2), AT(c.geom_xmat, 9 * g1 + 5), AT(c.geom_xmat, 9 * g1 + 8)};
      const V3 pp = {AT(c.geom_xpos, 3 * g1), AT(c.geom_xpos, 3 * g1 + 1), AT(c.geom_xpos, 3 * g1 + 2)};
      if (t2 == BS_GEOM_CAPSULE) {  // cap centre +axis, then -axis
        const V3 w = mrot(b.R, V3{0, 0, sub == 0 ? b.size.y : -b.size.y});
        hit = plane_point(pp, n, V3{b.xpos.x + w.x, b.xpos.y + w.y, b.xpos.z + w.z}, b.size.x, margin, dist, pos);
      } else {
        // corner `sub`, bit 0 / 1 / 2 set = positive x / y / z side (mjc_PlaneBox's order [EXT]): a contact when the
        // corner is on the plane's side of the box centre and below the margin, and fewer than four before it were
        int before = 0;
        bool mine = false;
        V3 cs = {0, 0, 0};
        for (int j = 0; j <= sub; j++) {
          const V3 o = mrot(b.R, V3{j & 1 ? b.size.x : -b.size.x, j & 2 ? b.size.y : -b.size.y, j & 4 ? b.size.z : -b.size.z});
          const V3 cj = {b.xpos.x + o.x, b.xpos.y + o.y, b.xpos.z + o.z};
          const bool ok = !(dot(sub3(cj, pp), n) > margin) && !(dot(n, o) > 0);
          if (j < sub) before += ok;
          else { mine = ok; cs = cj; }
        }
        hit = mine && before < 4 && plane_point(pp, n, cs, 0.0, margin, dist, pos);
      }
      nrm = n;
      done = true;
    } else if (t1 == BS_GEOM_SPHERE || (t1 == BS_GEOM_CAPSULE && t2 == BS_GEOM_CAPSULE)) {
      const V3 x1 = {AT(c.geom_xpos, 3 * g1), AT(c.geom_xpos, 3 * g1 + 1), AT(c.geom_xpos, 3 * g1 + 2)};
      const V3 s1 = ld3(cm.geom_size + 3 * g1);
      const V3 a2 = {b.R.a[2], b.R.a[5], b.R.a[8]};
      if (t1 == BS_GEOM_SPHERE && t2 == BS_GEOM_SPHERE) {
        hit = sphere_sphere(x1, s1.x, b.xpos, b.size.x, margin, dist, pos, nrm);
        done = true;
      } else if (t1 == BS_GEOM_SPHERE && t2 == BS_GEOM_CAPSULE) {
        hit = sphere_sphere(x1, s1.x, seg_closest_point(b.xpos, a2, b.size.y, x1), b.size.x, margin, dist, pos, nrm);
        done = true;
      } else if (t1 == BS_GEOM_CAPSULE) {
        const V3 a1 = {AT(c.geom_xmat, 9 * g1 + 2), AT(c.geom_xmat, 9 * g1 + 5), AT(c.geom_xmat, 9 * g1 + 8)};
        hit = capsule_capsule(x1, a1, s1.y, s1.x, b.xpos, a2, b.size.y, b.size.x, margin, sub, dist, pos, nrm);
        done = true;
      }
    }
  }
  if (done) {
  } else if (t1 == BS_GEOM_PLANE) {
    b.margin = 0;
    const V3 n = {AT(c.geom_xmat, 9 * g1 + 2), AT(c.geom_xmat, 9 * g1 + 5), AT(c.geom_xmat, 9 * g1 + 8)};
    const V3 pp = {AT(c.geom_xpos, 3 * g1), AT(c.geom_xpos, 3 * g1 + 1), AT(c.geom_xpos, 3 * g1 + 2)};
    V3 v;
    if constexpr (PRIM) v = geom_support_any(b, neg(n));
    else v = hull_support(b, neg(n));
    const double ds = dot(sub(v, pp), n);
    if (!(ds > margin)) {
      hit = true; dist = ds; nrm = n;
      pos = {v.x - 0.5 * ds * n.x, v.y - 0.5 * ds * n.y, v.z - 0.5 * ds * n.z};
    }
  } else {
    CObj a;
    load_obj(a, g1);
    a.margin = 0.5 * margin; b.margin = 0.5 * margin;
    double depth;
    int iters = 0;
    bool pen;
    if (PRIM && !(a.type == BS_GEOM_MESH && b.type == BS_GEOM_MESH)) {
      // the same portal refinement around the primitives' support functions
      mpr_begin(M, a.xpos, b.xpos);
      for (;;) {
        Spt s;
        s.v1 = geom_support_any(a, M.d);
        s.v2 = geom_support_any(b, neg(M.d));
        s.v = sub(s.v1, s.v2);
        const int r = mpr_step(M, s, a.xpos, b.xpos, cm.mpr_tolerance, cm.mpr_iterations, depth, nrm, pos, iters);
        if (r) { pen = r == 1; break; }
      }
    } else {
      pen = mpr_penetration(a, b, cm.mpr_tolerance, cm.mpr_iterations, depth, nrm, pos, iters, M);
    }
    if (pen && !(nrm.x == 0 && nrm.y == 0 && nrm.z == 0)) { hit = true; dist = margin - depth; }
    if (acc) { acc[0] += 1; acc[1] += (unsigned)iters; }
    else if (c.stat) { atomicAdd(c.stat, 1ull); atomicAdd(c.stat + 1, (unsigned long long)iters); }
  }
  AT(c.cand_hit, slot) = hit ? 1 : 0;
  if (hit) {
    AT(c.cand_dist, slot) = dist;
    AT(c.cand_pos, 3 * slot) = pos.x; AT(c.cand_pos, 3 * slot + 1) = pos.y; AT(c.cand_pos, 3 * slot + 2) = pos.z;
    AT(c.cand_nrm, 3 * slot) = nrm.x; AT(c.cand_nrm, 3 * slot + 1) = nrm.y; AT(c.cand_nrm, 3 * slot + 2) = nrm.z;
  }
}

// A6, persistent-lane form (the default of the tree-parallel path): a wavefront owns the items
// [first, last) of the work list; every lane runs ONE loop around the support call and, when its item
// is finished, takes the next item of the wavefront's range inside that loop (a wave-uniform cursor
// advanced by ballot / popcount: no atomics), so a lane whose pair separates after one support call
// does not idle until the slowest pair of the wavefront has refined its portal.  A plane item is one
// support call of the hull along -normal (its partner is a dummy hull whose climb ends at once).
// Same arithmetic per item as col_narrowphase<false>: bitwise the same candidates.
__device__ inline void col_narrow_persistent(const ColModel& cm, const DevData& d, const ColData& c, unsigned first,
                                             unsigned last, const unsigned* pre, const int4* items, long subcap,
                                             unsigned* acc, MprState& M) {
  const long E = d.E;
  const int lane = threadIdx.x & 63;
  const unsigned long long lower = (1ull << lane) - 1ull;
  unsigned cursor = first;  // wave-uniform
  int sublist = 0;  // sub-list of the wavefront's first item (uniform), then of the lane's last item
  {
    int hi = BS_NSUB;
    while (hi - sublist > 1) {
      const int mid = (sublist + hi) >> 1;
      if (pre[mid] <= first) sublist = mid;
      else hi = mid;
    }
  }
  bool busy = false, plane = false, pend = false;
  long e = 0;
  int slot = 0, iters = 0;
  double margin = 0;
  V3 pn = {0, 0, 0}, pp = {0, 0, 0};
  CObj a, b;
  a.type = b.type = BS_GEOM_MESH; a.size = b.size = {0, 0, 0};
  a.vert4 = b.vert4 = cm.hull_vert4; a.nbr4 = b.nbr4 = cm.hull_nbr4; a.start4 = b.start4 = cm.hull_start4;
  a.vert_all = b.vert_all = cm.hull_vert4; a.start_all = b.start_all = cm.hull_start4;
  a.cand_all = b.cand_all = cm.hull_cand4; a.more_all = b.more_all = cm.hull_more4;
  a.vbase = b.vbase = 0; a.sbase = b.sbase = 0;
  a.cache = b.cache = 0; a.margin = b.margin = 0;
  a.xpos = b.xpos = {0, 0, 0};
  for (int k = 0; k < 9; k++) { a.R.a[k] = 0; b.R.a[k] = 0; }
#ifdef NP_TIMING
  unsigned long long np_t[24];
  for (int k = 0; k < 24; k++) np_t[k] = 0;
  unsigned long long np_last = clock64();
#define NPT_CLK(i) { const unsigned long long now_ = clock64(); np_t[i] += now_ - np_last; np_last = now_; }
#else
#define NPT_CLK(i)
#endif
  for (;;) {
    // ---- lanes without an item take the next ones of the wavefront's range
    const unsigned long long need = __ballot(!busy);
    NPT(3, 1)
    if (need) {
      NPT(4, __popcll(need))
      const unsigned i = cursor + (unsigned)__popcll(need & lower);
      cursor += (unsigned)__popcll(need);
      if (!busy && i < last) {
        // sub-list of item i: largest s with pre[s] <= i.  Items are handed out in increasing order, so a lane's
        // sub-list never decreases and the search resumes where its last one ended (a step or none, not six)
        while (pre[sublist + 1] <= i) sublist++;  // i < last <= pre[BS_NSUB]
        const int lo = sublist;
        const int4 it = items[lo * subcap + (i - pre[lo])];
        e = d.env0 + (it.x >> 8); slot = it.x & 255;
        const int p = it.y & BS_PAIR_MASK;
        const int4 r0 = *(const int4*)(cm.pair_np + 8 * p), r1 = *(const int4*)(cm.pair_np + 8 * p + 4);
        const int g1 = it.z & 0xffff, g2 = it.z >> 16 & 0xffff;
        // frames: one 96-byte record per geom (c.gframe) or the [row][E] arrays; both are fetched whole before anything
        // that depends on the pair record (plane or hull pair) is decided, so they travel with it
        double F1[12], F2[12];
        if (c.gframe) {
          const double* f1 = c.gframe + ((size_t)e * cm.ngeom + g1) * 12;
          const double* f2 = c.gframe + ((size_t)e * cm.ngeom + g2) * 12;
#pragma unroll
          for (int k = 0; k < 12; k++) { F1[k] = f1[k]; F2[k] = f2[k]; }
        } else {
#pragma unroll
          for (int k = 0; k < 3; k++) { F1[k] = AT(c.geom_xpos, 3 * g1 + k); F2[k] = AT(c.geom_xpos, 3 * g2 + k); }
#pragma unroll
          for (int k = 0; k < 9; k++) { F1[3 + k] = AT(c.geom_xmat, 9 * g1 + k); F2[3 + k] = AT(c.geom_xmat, 9 * g2 + k); }
        }
        plane = (r0.x >> 16) != 0;
        margin = __hiloint2double(r1.w, r1.z);
        b.vbase = r1.x; b.sbase = r1.y;
        b.cache = 0;
        b.xpos = {F2[0], F2[1], F2[2]};
#pragma unroll
        for (int k = 0; k < 9; k++) b.R.a[k] = F2[3 + k];
        iters = 0;
        if (plane) {
          pn = {F1[3 + 2], F1[3 + 5], F1[3 + 8]};
          pp = {F1[0], F1[1], F1[2]};
          b.margin = 0;
          // the partner is not climbed (md_support's only_b): it points at the same hull so its start loads are valid
          a.vbase = b.vbase; a.sbase = b.sbase; a.cache = 0; a.margin = 0;
          M.d = pn;  // the hull is queried along -d
        } else {
          a.vbase = r0.z; a.sbase = r0.w;
          a.cache = 0;
          a.xpos = {F1[0], F1[1], F1[2]};
#pragma unroll
          for (int k = 0; k < 9; k++) a.R.a[k] = F1[3 + k];
          a.margin = 0.5 * margin; b.margin = 0.5 * margin;
          mpr_begin(M, a.xpos, b.xpos);
        }
        busy = true;
      }
    }
    NPT_CLK(0)
    if (!__ballot(busy)) break;
    NPT(5, __popcll(__ballot(busy)))
#ifdef NP_TIMING
    {
      const int st = plane ? 5 : (M.pk & 7);
      for (int q = 0; q < 6; q++) { const unsigned long long bq = __ballot(busy && st == q); NPT(8 + q, bq != 0) NPT(14 + q, __popcll(bq)) }
    }
#endif
    // the candidate record of a finished item
    auto put_record = [&](bool hit, double dist, const V3& pos, const V3& nrm) {
      if (c.crec) {
        CandRec* cr = c.crec + (size_t)e * cm.maxcand + slot;
        cr->hit = hit ? 1 : 0;
        if (hit) {
          cr->dist = dist;
          cr->pos[0] = pos.x; cr->pos[1] = pos.y; cr->pos[2] = pos.z;
          cr->nrm[0] = nrm.x; cr->nrm[1] = nrm.y; cr->nrm[2] = nrm.z;
        }
      } else {
        AT(c.cand_hit, slot) = hit ? 1 : 0;
        if (hit) {
          AT(c.cand_dist, slot) = dist;
          AT(c.cand_pos, 3 * slot) = pos.x; AT(c.cand_pos, 3 * slot + 1) = pos.y; AT(c.cand_pos, 3 * slot + 2) = pos.z;
          AT(c.cand_nrm, 3 * slot) = nrm.x; AT(c.cand_nrm, 3 * slot + 1) = nrm.y; AT(c.cand_nrm, 3 * slot + 2) = nrm.z;
        }
      }
    };
    if (busy && !pend) {
#ifdef NP_TIMING
      const Spt s = md_support(a, b, M.d, plane, np_t);
#else
      const Spt s = md_support(a, b, M.d, plane);
#endif
      NPT_CLK(1)
      bool fin, hit = false;
      double dist = 0;
      V3 pos = {0, 0, 0}, nrm = {0, 0, 0};
      if (plane) {
        fin = true;
        const double ds = dot(sub(s.v2, pp), pn);
        if (!(ds > margin)) {
          hit = true; dist = ds; nrm = pn;
          pos = {s.v2.x - 0.5 * ds * pn.x, s.v2.y - 0.5 * ds * pn.y, s.v2.z - 0.5 * ds * pn.z};
        }
      } else {
        double depth = 0;
        const int r = mpr_step<true>(M, s, a.xpos, b.xpos, cm.mpr_tolerance, cm.mpr_iterations, depth, nrm, pos, iters);
        fin = r == 1 || r == 2;
        pend = r == 3;  // converged: depth, normal and position are computed below, for several lanes at once
        if (r == 1 && !(nrm.x == 0 && nrm.y == 0 && nrm.z == 0)) { hit = true; dist = margin - depth; }
        if (fin) { acc[0] += 1; acc[1] += (unsigned)iters; }
      }
      if (fin) {
        put_record(hit, dist, pos, nrm);
        busy = false;
      }
    }
    // ---- converged pairs wait (their lanes idle) until NP_FIN_BATCH of them can run mpr_finish together, or nothing else is left to do
    {
      const unsigned long long pb = __ballot(pend);
      if (pb && (__popcll(pb) >= NP_FIN_BATCH || !__ballot(busy && !pend))) {
        NPT(20, 1) NPT(21, __popcll(pb))
        if (pend) {
          double depth = 0;
          V3 pos = {0, 0, 0}, nrm = {0, 0, 0};
          mpr_finish(M, a.xpos, b.xpos, depth, nrm, pos);
          const bool hit = !(nrm.x == 0 && nrm.y == 0 && nrm.z == 0);
          acc[0] += 1; acc[1] += (unsigned)iters;
          put_record(hit, margin - depth, pos, nrm);
          busy = false; pend = false;
        }
      }
    }
    NPT_CLK(2)
  }
#ifdef NP_TIMING
  if (lane == 0 && c.stat)
    for (int k = 0; k < 24; k++) atomicAdd(c.stat + 8 + k, np_t[k]);
#endif
}

// A6, work-stealing form: the lanes of a block pull (environment, candidate slot) items from a
// shared counter and run one loop in which every iteration is ONE batch of four 32-byte record
// loads (the start-table record or four neighbours of the current vertex) for whichever hull the
// lane is climbing; MPR transitions and item set-up run in batches when enough lanes wait for
// them.  Same arithmetic per item as col_narrowphase (climbs start at the table vertex).
#ifndef NW_BATCH
#define NW_BATCH 64  // lanes that must wait before a set-up / MPR batch runs while others still climb
#endif
__device__ inline void col_narrow_ws(const ColModel& cm, const DevData& d, const ColData& c, long e0, const int* pre,
                                     int nenv, int* next, unsigned* acc) {
  const long E = d.E;
  const int total = pre[nenv];
  int sub = 0;  // 0 needs an item, 1 climbing, 3 both support points ready, 4 out of items
  long e = 0;
  int slot = 0, amesh = 0, bmesh = 0, which = 0, iters = 0;
  bool plane = false;
  double margin = 0, infl = 0;
  V3 ax = {0, 0, 0}, bx = {0, 0, 0}, pn = {0, 0, 0};
  M3 aR, bR;
  for (int q = 0; q < 9; q++) { aR.a[q] = 0; bR.a[q] = 0; }
  MprState M;
  mpr_begin(M, ax, bx);
  // climb
  V3 dloc = {0, 0, 0}, best = {0, 0, 0}, v1 = {0, 0, 0}, v2 = {0, 0, 0};
  double vmax = 0;
  int imax = 0, prev = 0, k = 0, kend = 0, bk0 = 0, bdeg = 0;
  const double4* ptr = cm.hull_nbr4;
  auto begin_climb = [&](int w) {
    which = w;
    const M3& R = w ? bR : aR;
    const V3 dw = w ? neg(M.d) : M.d;
    dloc = {R.a[0] * dw.x + R.a[3] * dw.y + R.a[6] * dw.z, R.a[1] * dw.x + R.a[4] * dw.y + R.a[7] * dw.z,
            R.a[2] * dw.x + R.a[5] * dw.y + R.a[8] * dw.z};
    ptr = cm.hull_start4 + (size_t)(w ? bmesh : amesh) * BS_NSTART + dir_bin(dloc.x, dloc.y, dloc.z);
    k = 0; kend = 1; vmax = -1e300; imax = -1; prev = -2;
  };
  for (;;) {
    // ---- item set-up, when enough lanes are idle (or nothing else can run)
    {
      const unsigned long long need = __ballot(sub == 0), busy = __ballot(sub == 1);
      if (need && (__popcll(need) >= NW_BATCH || busy == 0) && sub == 0) {
        const int item = atomicAdd(next, 1);
        if (item >= total) sub = 4;
        else {
          int lo = 0, hi = nenv;  // largest env index with pre[env] <= item
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= item) lo = mid;
            else hi = mid;
          }
          e = e0 + lo; slot = item - pre[lo];
          const int p = AT(c.cand_pair, slot) & BS_PAIR_MASK;
          const int g1 = cm.pair_geom1[p], g2 = cm.pair_geom2[p];
          margin = fmax(cm.geom_margin[g1], cm.geom_margin[g2]);
          bmesh = cm.geom_dataid[g2];
          bx = {AT(c.geom_xpos, 3 * g2), AT(c.geom_xpos, 3 * g2 + 1), AT(c.geom_xpos, 3 * g2 + 2)};
          for (int q = 0; q < 9; q++) bR.a[q] = AT(c.geom_xmat, 9 * g2 + q);
          ax = {AT(c.geom_xpos, 3 * g1), AT(c.geom_xpos, 3 * g1 + 1), AT(c.geom_xpos, 3 * g1 + 2)};
          plane = cm.geom_type[g1] == BS_GEOM_PLANE;
          iters = 0;
          if (plane) {
            pn = {AT(c.geom_xmat, 9 * g1 + 2), AT(c.geom_xmat, 9 * g1 + 5), AT(c.geom_xmat, 9 * g1 + 8)};
            infl = 0;
            M.d = pn;  // hull b is queried along -d
            begin_climb(1);
          } else {
            amesh = cm.geom_dataid[g1];
            for (int q = 0; q < 9; q++) aR.a[q] = AT(c.geom_xmat, 9 * g1 + q);
            infl = 0.5 * margin;
            mpr_begin(M, ax, bx);
            begin_climb(0);
          }
          sub = 1;
        }
      }
    }
    if (__ballot(sub != 4) == 0) break;
    // ---- one batch of records for every climbing lane
    if (sub == 1) {
      double4 p[4];
#pragma unroll
      for (int q = 0; q < 4; q++) p[q] = ptr[k + q < kend ? k + q : kend - 1];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double val = p[q].x * dloc.x + p[q].y * dloc.y + p[q].z * dloc.z;
        if (val > vmax) {
          vmax = val; best = {p[q].x, p[q].y, p[q].z};
          imax = rec_id(p[q].w); bk0 = rec_k0(p[q].w); bdeg = rec_deg(p[q].w);
        }
      }
      k += 4;
      if (k >= kend) {
        if (imax != prev && bdeg > 0) {  // moved: scan the new vertex's neighbours
          prev = imax; ptr = cm.hull_nbr4; k = bk0; kend = bk0 + bdeg;
        } else {  // local (= global) maximum: world-frame support point of the inflated hull
          const M3& R = which ? bR : aR;
          const V3& xp = which ? bx : ax;
          const V3 dw = which ? neg(M.d) : M.d;
          const V3 w = mrot(R, best);
          const V3 sp = {xp.x + w.x + infl * dw.x, xp.y + w.y + infl * dw.y, xp.z + w.z + infl * dw.z};
          if (which == 0) { v1 = sp; begin_climb(1); }
          else { v2 = sp; sub = 3; }
        }
      }
    }
    // ---- MPR transitions, batched
    {
      const unsigned long long ready = __ballot(sub == 3), busy = __ballot(sub == 1);
      if (ready && (__popcll(ready) >= NW_BATCH || busy == 0) && sub == 3) {
        bool fin = false, hit = false;
        double dist = 0;
        V3 pos = {0, 0, 0}, nrm = {0, 0, 0};
        if (plane) {
          const double ds = dot(sub3(v2, ax), pn);
          fin = true;
          if (!(ds > margin)) {
            hit = true; dist = ds; nrm = pn;
            pos = {v2.x - 0.5 * ds * pn.x, v2.y - 0.5 * ds * pn.y, v2.z - 0.5 * ds * pn.z};
          }
        } else {
          Spt s;
          s.v1 = v1; s.v2 = v2; s.v = sub3(v1, v2);
          double depth = 0;
          const int r = mpr_step(M, s, ax, bx, cm.mpr_tolerance, cm.mpr_iterations, depth, nrm, pos, iters);
          if (r) {
            fin = true;
            if (r == 1 && !(nrm.x == 0 && nrm.y == 0 && nrm.z == 0)) { hit = true; dist = margin - depth; }
            acc[0] += 1; acc[1] += (unsigned)iters;
          } else {
            begin_climb(0);
            sub = 1;
          }
        }
        if (fin) {
          AT(c.cand_hit, slot) = hit ? 1 : 0;
          if (hit) {
            AT(c.cand_dist, slot) = dist;
            AT(c.cand_pos, 3 * slot) = pos.x; AT(c.cand_pos, 3 * slot + 1) = pos.y; AT(c.cand_pos, 3 * slot + 2) = pos.z;
            AT(c.cand_nrm, 3 * slot) = nrm.x; AT(c.cand_nrm, 3 * slot + 1) = nrm.y; AT(c.cand_nrm, 3 * slot + 2) = nrm.z;
          }
          sub = 0;
        }
      }
    }
  }
}

// contact assembly: hits in candidate order -> contact list with completed tangent frames
__device__ inline void col_assemble(const ColModel& cm, const DevData& d, const ColData& c, long e) {
  const long E = d.E;
  const int nc = AT(c.ncand, 0);
  int n = 0;
  for (int s = 0; s < nc; s++) {
    if (!AT(c.cand_hit, s)) continue;
    if (n >= cm.maxcon) { AT(d.warning, 0) |= 8; break; }
    AT(c.con_pair, n) = AT(c.cand_pair, s) & BS_PAIR_MASK;
    AT(c.con_dist, n) = AT(c.cand_dist, s);
    for (int k = 0; k < 3; k++) AT(c.con_pos, 3 * n + k) = AT(c.cand_pos, 3 * s + k);
    V3 x = normalized(V3{AT(c.cand_nrm, 3 * s), AT(c.cand_nrm, 3 * s + 1), AT(c.cand_nrm, 3 * s + 2)});
    V3 y = {0, 0, 0};
    if (x.y < 0.5 && x.y > -0.5) y.y = 1;
    else y.z = 1;
    const double t = dot(x, y);
    y = normalized(V3{y.x - t * x.x, y.y - t * x.y, y.z - t * x.z});
    const V3 z = cross(x, y);
    AT(c.con_frame, 9 * n) = x.x; AT(c.con_frame, 9 * n + 1) = x.y; AT(c.con_frame, 9 * n + 2) = x.z;
    AT(c.con_frame, 9 * n + 3) = y.x; AT(c.con_frame, 9 * n + 4) = y.y; AT(c.con_frame, 9 * n + 5) = y.z;
    AT(c.con_frame, 9 * n + 6) = z.x; AT(c.con_frame, 9 * n + 7) = z.y; AT(c.con_frame, 9 * n + 8) = z.z;
    n++;
  }
  AT(c.ncon, 0) = n;
  if (c.stat && n) atomicAdd(c.stat + 2, (unsigned long long)n);
}
