// Test infrastructure (CPU): the product's MPR step functions (mujoco_amd/csrc/physics_collide.h: mpr_begin, mpr_step,
// mpr_finish — __host__ __device__) compiled for the HOST, around a brute-force support function, so that the two forms
// of mpr_step (default: one instruction stream; -DMPR_BRANCHY: one branch per portal phase) can be compared with each
// other, call by call, without a GPU.  Built twice by tests/test_mpr_forms.py (hipcc --cuda-host-only).
#include "../../mujoco_amd/csrc/physics_collide.h"
#include <cstring>

namespace {
struct Hull { const double* v; int n; V3 x; M3 R; double margin; };
// support of a posed hull along `dir` (world): the arithmetic of hull_support's last lines around an exhaustive search
V3 support(const Hull& h, const V3& dir) {
  const M3& R = h.R;
  const V3 d = {R.a[0] * dir.x + R.a[3] * dir.y + R.a[6] * dir.z, R.a[1] * dir.x + R.a[4] * dir.y + R.a[7] * dir.z,
                R.a[2] * dir.x + R.a[5] * dir.y + R.a[8] * dir.z};
  int best = 0;
  double vmax = h.v[0] * d.x + h.v[1] * d.y + h.v[2] * d.z;
  for (int i = 1; i < h.n; i++) {
    const double val = h.v[3 * i] * d.x + h.v[3 * i + 1] * d.y + h.v[3 * i + 2] * d.z;
    if (val > vmax) { vmax = val; best = i; }
  }
  const V3 w = mrot(R, V3{h.v[3 * best], h.v[3 * best + 1], h.v[3 * best + 2]});
  return {h.x.x + w.x + h.margin * dir.x, h.x.y + w.y + h.margin * dir.y, h.x.z + w.z + h.margin * dir.z};
}
unsigned long long mix(unsigned long long h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}
}  // namespace

// One pair.  pose = {xa[3], Ra[9], xb[3], Rb[9]}.  out = {depth, dir[3], pos[3]}.  Returns mpr_step's final code
// (1 hit, 2 miss); *calls = support calls, *iters = refinement iterations, *trace = a hash of every call's return
// code, packed state word and next direction (bit patterns): equal traces = the same path, step by step.
extern "C" int mpr_host_pair(const double* va, int na, const double* vb, int nb, const double* pose, double margin,
                             double tol, int maxit, double* out, int* calls, int* iters, unsigned long long* trace) {
  Hull a = {va, na, {pose[0], pose[1], pose[2]}, {}, 0.5 * margin}, b = {vb, nb, {pose[12], pose[13], pose[14]}, {}, 0.5 * margin};
  for (int k = 0; k < 9; k++) { a.R.a[k] = pose[3 + k]; b.R.a[k] = pose[15 + k]; }
  MprState M;
  memset(&M, 0, sizeof M);
  mpr_begin(M, a.x, b.x);
  double depth = 0;
  V3 dir = {0, 0, 0}, pos = {0, 0, 0};
  int it = 0, n = 0, r = 0;
  unsigned long long h = 1469598103934665603ull;
  for (;;) {
    Spt s;
    s.v1 = support(a, M.d);
    s.v2 = support(b, neg(M.d));
    s.v = sub(s.v1, s.v2);
    r = mpr_step<false>(M, s, a.x, b.x, tol, maxit, depth, dir, pos, it);
    n++;
    h = mix(h, &r, sizeof r);
    if (!r) { h = mix(h, &M.pk, sizeof M.pk); h = mix(h, &M.d, sizeof M.d); }
    if (r || n > 10000) break;
  }
  out[0] = depth; out[1] = dir.x; out[2] = dir.y; out[3] = dir.z; out[4] = pos.x; out[5] = pos.y; out[6] = pos.z;
  *calls = n; *iters = it; *trace = h;
  return r;
}
