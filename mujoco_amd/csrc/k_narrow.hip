// k_narrow.hip — narrowphase kernels (A6; physics_collide.h::col_narrowphase / col_narrow_ws) and
// their launchers.
#include "kernels.h"

// narrowphase over the flattened (environment, candidate slot) items of NB_ENVS environments:
// an exclusive scan of the candidate counts in LDS, then every lane takes items round-robin, so
// no lane idles on environments with few candidates and no wavefront is launched for empty slots
template <bool PRIM>
__global__ __launch_bounds__(NB_THREADS) void k_c_narrow_flat(ColModel cm, DevData d, ColData c) {
  __shared__ int pre[NB_ENVS + 1];
  __shared__ MprState mpr[NB_THREADS];  // MPR state of every lane (see mpr_penetration)
  const long e0 = d.env0 + (long)blockIdx.x * NB_ENVS;
  const int tid = threadIdx.x;
  if (tid < NB_ENVS) {  // first wavefront: inclusive scan of the candidate counts with DPP-free shuffles
    const long e = e0 + tid, E = d.E;
    int v = e < d.n_envs ? AT(c.ncand, 0) : 0;
    for (int off = 1; off < NB_ENVS; off <<= 1) {
      const int u = __shfl_up(v, off, 64);
      if (tid >= off) v += u;
    }
    pre[tid + 1] = v;
    if (tid == 0) pre[0] = 0;
  }
  __syncthreads();
  const int total = pre[NB_ENVS];
  unsigned acc[2] = {0, 0};
  for (int item = tid; item < total; item += NB_THREADS) {
    int lo = 0, hi = NB_ENVS;  // largest env index with pre[env] <= item
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= item) lo = mid;
      else hi = mid;
    }
    col_narrowphase<PRIM>(cm, d, c, e0 + lo, item - pre[lo], acc, mpr[tid]);
  }
  if (c.stat) {  // work counters: wavefront reduction, then one atomic each
    for (int off = 32; off > 0; off >>= 1) {
      acc[0] += __shfl_down(acc[0], off, 64);
      acc[1] += __shfl_down(acc[1], off, 64);
    }
    if ((tid & 63) == 0 && acc[0]) {
      atomicAdd(c.stat, (unsigned long long)acc[0]);
      atomicAdd(c.stat + 1, (unsigned long long)acc[1]);
    }
  }
}
__global__ __launch_bounds__(NW_THREADS) void k_c_narrow_ws(ColModel cm, DevData d, ColData c) {
  __shared__ int pre[NB_ENVS + 1];
  __shared__ int next;
  const long e0 = d.env0 + (long)blockIdx.x * NB_ENVS;
  const int tid = threadIdx.x;
  if (tid < NB_ENVS) {
    const long e = e0 + tid, E = d.E;
    int v = e < d.n_envs ? AT(c.ncand, 0) : 0;
    for (int off = 1; off < NB_ENVS; off <<= 1) {
      const int u = __shfl_up(v, off, 64);
      if (tid >= off) v += u;
    }
    pre[tid + 1] = v;
    if (tid == 0) { pre[0] = 0; next = 0; }
  }
  __syncthreads();
  unsigned acc[2] = {0, 0};
  col_narrow_ws(cm, d, c, e0, pre, NB_ENVS, &next, acc);
  if (c.stat) {
    for (int off = 32; off > 0; off >>= 1) {
      acc[0] += __shfl_down(acc[0], off, 64);
      acc[1] += __shfl_down(acc[1], off, 64);
    }
    if ((tid & 63) == 0 && acc[0]) {
      atomicAdd(c.stat, (unsigned long long)acc[0]);
      atomicAdd(c.stat + 1, (unsigned long long)acc[1]);
    }
  }
}
// narrowphase over the work list k_tp_col wrote: one lane per item whatever environment it belongs
// to, so every wavefront but the last is full
template <bool PRIM>
__global__ __launch_bounds__(NL_THREADS, NL_OCC) void k_c_narrow_list(ColModel cm, DevData d, ColData c) {
  __shared__ unsigned pre[BS_NSUB + 1];
  __shared__ MprState mpr[NL_THREADS];
  const int tid = threadIdx.x;
#ifdef NL_PAD
  __shared__ int padd[NL_PAD];
  if (c.stat == nullptr && d.n_envs < 0) padd[tid] = 1;
#endif
  const long subcap = col_items_subcap(d.n_envs - d.env0, cm.maxcand);
  if (tid < BS_NSUB) {  // first wavefront: inclusive scan of the sub-list lengths (never beyond a sub-list's capacity)
    unsigned v = (unsigned)c.stat[BS_STAT_ITEMS + d.range * BS_NSUB + tid];
    if ((long)v > subcap) v = (unsigned)subcap;
    for (int off = 1; off < BS_NSUB; off <<= 1) {
      const unsigned u = __shfl_up(v, off, 64);
      if (tid >= off) v += u;
    }
    pre[tid + 1] = v;
    if (tid == 0) pre[0] = 0;
  }
  __syncthreads();
  const unsigned total = pre[BS_NSUB];
  const int4* items = c.items + col_items_base(d.env0, d.range, cm.maxcand);
  unsigned acc[2] = {0, 0};
  if constexpr (!PRIM) {
    // every wavefront owns an equal share of the list and its lanes take items from it as they finish
    const unsigned nw = gridDim.x * (NL_THREADS / 64), w = blockIdx.x * (NL_THREADS / 64) + tid / 64;
    unsigned share = (total + nw - 1) / nw;
    if (share < (unsigned)cm.nl_min_share) share = (unsigned)cm.nl_min_share;  // a few items per lane, or the lanes idle in the range's tail
    const unsigned first = w * share < total ? w * share : total, last = first + share < total ? first + share : total;
    col_narrow_persistent(cm, d, c, first, last, pre, items, subcap, acc, mpr[tid]);
  } else {
    for (unsigned i = blockIdx.x * NL_THREADS + tid; i < total; i += gridDim.x * NL_THREADS) {
      int lo = 0, hi = BS_NSUB;  // sub-list of item i: largest s with pre[s] <= i
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= i) lo = mid;
        else hi = mid;
      }
      const int it = items[lo * subcap + (i - pre[lo])].x;
      col_narrowphase<PRIM>(cm, d, c, d.env0 + (it >> 8), it & 255, acc, mpr[tid]);
    }
  }
  if (c.stat) {
    for (int off = 32; off > 0; off >>= 1) {
      acc[0] += __shfl_down(acc[0], off, 64);
      acc[1] += __shfl_down(acc[1], off, 64);
    }
    if ((tid & 63) == 0 && acc[0]) {
      atomicAdd(c.stat, (unsigned long long)acc[0]);
      atomicAdd(c.stat + 1, (unsigned long long)acc[1]);
    }
  }
}

// prim: the model has pairs with primitive geoms (the PRIM instantiation carries their closed forms and
// support functions; hull / plane-only models run the lean one)
void bs_launch_narrow_flat(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c, int prim) {
  if (prim) hipLaunchKernelGGL(k_c_narrow_flat<true>, g, dim3(NB_THREADS), 0, st, cm, d, c);
  else hipLaunchKernelGGL(k_c_narrow_flat<false>, g, dim3(NB_THREADS), 0, st, cm, d, c);
}
void bs_launch_narrow_ws(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c) {
  hipLaunchKernelGGL(k_c_narrow_ws, g, dim3(NW_THREADS), 0, st, cm, d, c);
}
void bs_launch_narrow_list(dim3 g, hipStream_t st, const ColModel& cm, const DevData& d, const ColData& c, int prim) {
  if (prim) hipLaunchKernelGGL(k_c_narrow_list<true>, g, dim3(NL_THREADS), 0, st, cm, d, c);
  else hipLaunchKernelGGL(k_c_narrow_list<false>, g, dim3(NL_THREADS), 0, st, cm, d, c);
}
#ifdef NP_PROFILE
void bs_np_profile_dump() {
  unsigned long long hc[32], hs[64];
  (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(np_hist_climb), sizeof hc);
  (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(np_hist_sup), sizeof hs);
  fprintf(stderr, "climb trips per support:");
  for (int q = 0; q < 32; q++) fprintf(stderr, " %llu", hc[q]);
  fprintf(stderr, "\nsupport pairs per item:");
  for (int q = 0; q < 64; q++) fprintf(stderr, " %llu", hs[q]);
  fprintf(stderr, "\n");
}
#endif
