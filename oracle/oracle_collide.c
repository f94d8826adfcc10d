/* oracle_collide.c — collision stage of the CPU restatement.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h).
 *
 * Restates, from published descriptions [EXT], what MuJoCo 2.1.0 does for the exporter's geoms
 * (type="mesh", margin 0.0005, contype=conaffinity=1: reference src/Mujoco_XML.py:70; excludes
 * src/Mujoco_XML.py:218-219,267 / output/DexterousDynamos.xml:138-159):
 *   - candidate pairs were filtered at compile time (mujoco_amd/model.py::_build_pairs);
 *   - broadphase: bounding spheres about the geom centre, inflated by the pair margin;
 *   - mesh-mesh narrowphase (mjc_Convex): Minkowski Portal Refinement (libccd's
 *     ccdMPRPenetration, XenoCollide) on the convex hulls, each hull inflated by margin/2,
 *     support by steepest-ascent hill climbing over the hull's vertex graph; at most ONE contact
 *     per pair; dist = margin - depth, position = MPR's barycentric estimate;
 *   - plane-mesh (the build's own ground plane): deepest hull vertex;
 *   - primitive geoms (north_star "plus plane/primitive pairs"; the dialect's geom default is mesh,
 *     src/Mujoco_XML.py:70, primitives only appear when a user edits it): closed forms for
 *     plane-sphere / plane-capsule (2 contacts) / plane-box (mjc_PlaneBox's rule [EXT]: corners on the plane's side of
 *     the box centre and below the margin, the first four in corner order),
 *     sphere-sphere, sphere-capsule, capsule-capsule (closest points of the segments; parallel axes: up to two,
 *     at the ends of the overlap, as mjc_CapsuleCapsule [EXT]);
 *     every other convex pair (box, cylinder, ellipsoid, mesh in any combination, sphere-box,
 *     capsule-box) goes through the same MPR with the primitives' support functions — MuJoCo's
 *     native box-box / capsule-box routines return more contacts per pair [EXT], this returns one;
 *   - contact parameters: friction/condim = max over the geoms, solref/solimp mixed by solmix,
 *     margin/gap = max; tangent frame completed as mju_makeFrame does.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

enum { GEOM_PLANE = 0, GEOM_HFIELD = 1, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_ELLIPSOID = 4, GEOM_CYLINDER = 5,
       GEOM_BOX = 6, GEOM_MESH = 7 };
enum { DSBL_CONTACT = 1 << 4 };

#define CCD_EPS DBL_EPSILON
static int is_zero(double x) { return fabs(x) < CCD_EPS; }
static int is_eq(double a, double b) {
  double ab = fabs(a - b);
  if (ab < CCD_EPS) return 1;
  double aa = fabs(a), bb = fabs(b);
  return ab < CCD_EPS * (bb > aa ? bb : aa);
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double *r, const double *a, const double *b) {
  double t[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  memcpy(r, t, sizeof t);
}
static void sub3(double *r, const double *a, const double *b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static double norm3(const double *a) { return sqrt(dot3(a, a)); }
static void normalize3(double *a) {
  double n = norm3(a);
  if (n > 0) { a[0] /= n; a[1] /= n; a[2] /= n; }
}
static int vec_eq(const double *a, const double *b) { return is_eq(a[0], b[0]) && is_eq(a[1], b[1]) && is_eq(a[2], b[2]); }

/* one convex object: a mesh geom's hull in world pose, inflated by `margin` */
typedef struct {
  int type;                     /* GEOM_* */
  const double *size;           /* geom_size (primitives) */
  const double *vert;           /* hull vertices of the mesh (local) */
  const int32_t *adj_start, *adj; /* CSR adjacency: adj_start is absolute per global vertex */
  int nvert, vertadr;
  const double *xpos, *xmat;
  double margin;
  int cache;                    /* last support vertex (hill-climb warm start) */
} cobj;

typedef struct { double v[3], v1[3], v2[3]; } spt; /* Minkowski-difference support point */

/* support of the inflated hull in world direction dir (unit length) */
static void hull_support(cobj *o, const double *dir, double *res) {
  const double *R = o->xmat;
  double d[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2],
                 R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  int imax = o->cache, prev = -1;
  double vmax = dot3(o->vert + 3 * imax, d);
  while (imax != prev) {
    prev = imax;
    int g = o->vertadr + prev;
    for (int k = o->adj_start[g]; k < o->adj_start[g + 1]; k++) {
      int nb = o->adj[k];
      double val = dot3(o->vert + 3 * nb, d);
      if (val > vmax) { vmax = val; imax = nb; }
    }
  }
  o->cache = imax;
  const double *v = o->vert + 3 * imax;
  for (int r = 0; r < 3; r++)
    res[r] = o->xpos[r] + R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2] + o->margin * dir[r];
}

/* support of any convex geom (inflated by o->margin) in world direction dir (unit length).  Ties
 * (a direction component of exactly 0) take the positive side, on the device as well. */
static void geom_support(cobj *o, const double *dir, double *res) {
  if (o->type == GEOM_MESH) { hull_support(o, dir, res); return; }
  const double *R = o->xmat, *sz = o->size;
  double d[3] = {R[0] * dir[0] + R[3] * dir[1] + R[6] * dir[2], R[1] * dir[0] + R[4] * dir[1] + R[7] * dir[2],
                 R[2] * dir[0] + R[5] * dir[1] + R[8] * dir[2]};
  double v[3] = {0, 0, 0}, infl = o->margin;
  switch (o->type) {
    case GEOM_SPHERE: infl += sz[0]; break;
    case GEOM_CAPSULE: v[2] = d[2] >= 0 ? sz[1] : -sz[1]; infl += sz[0]; break;
    case GEOM_CYLINDER: {
      double l = sqrt(d[0] * d[0] + d[1] * d[1]);
      if (l > 0) { v[0] = sz[0] * d[0] / l; v[1] = sz[0] * d[1] / l; }
      v[2] = d[2] >= 0 ? sz[1] : -sz[1];
      break;
    }
    case GEOM_BOX:
      for (int k = 0; k < 3; k++) v[k] = d[k] >= 0 ? sz[k] : -sz[k];
      break;
    case GEOM_ELLIPSOID: {
      double s0 = sz[0] * d[0], s1 = sz[1] * d[1], s2 = sz[2] * d[2], l = sqrt(s0 * s0 + s1 * s1 + s2 * s2);
      if (l > 0) { v[0] = sz[0] * s0 / l; v[1] = sz[1] * s1 / l; v[2] = sz[2] * s2 / l; }
      break;
    }
    default: break;
  }
  for (int r = 0; r < 3; r++)
    res[r] = o->xpos[r] + R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2] + infl * dir[r];
}

static void md_support(cobj *a, cobj *b, const double *dir, spt *s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  geom_support(a, dir, s->v1);
  geom_support(b, nd, s->v2);
  sub3(s->v, s->v1, s->v2);
}

/* ---- Minkowski Portal Refinement (XenoCollide), following libccd's control flow */
typedef struct { spt p[4]; int size; } portal_t;

static void portal_dir(const portal_t *P, double *dir) {
  double a[3], b[3];
  sub3(a, P->p[2].v, P->p[1].v);
  sub3(b, P->p[3].v, P->p[1].v);
  cross3(dir, a, b);
  normalize3(dir);
}
static int reach_tolerance(const portal_t *P, const spt *v4, const double *dir, double tol) {
  double dv4 = dot3(v4->v, dir);
  double d1 = dv4 - dot3(P->p[1].v, dir), d2 = dv4 - dot3(P->p[2].v, dir), d3 = dv4 - dot3(P->p[3].v, dir);
  double m = d1 < d2 ? d1 : d2;
  m = m < d3 ? m : d3;
  return is_eq(m, tol) || m < tol;
}
static void expand_portal(portal_t *P, const spt *v4) {
  double v4v0[3];
  cross3(v4v0, v4->v, P->p[0].v);
  if (dot3(P->p[1].v, v4v0) > 0) {
    if (dot3(P->p[2].v, v4v0) > 0) P->p[1] = *v4;
    else P->p[3] = *v4;
  } else {
    if (dot3(P->p[3].v, v4v0) > 0) P->p[2] = *v4;
    else P->p[1] = *v4;
  }
}
static int discover_portal(cobj *a, cobj *b, portal_t *P) {
  double dir[3], va[3], vb[3], d;
  memcpy(P->p[0].v1, a->xpos, 24);
  memcpy(P->p[0].v2, b->xpos, 24);
  sub3(P->p[0].v, P->p[0].v1, P->p[0].v2);
  P->size = 1;
  const double origin[3] = {0, 0, 0};
  if (vec_eq(P->p[0].v, origin)) { P->p[0].v[0] = CCD_EPS * 10; P->p[0].v[1] = P->p[0].v[2] = 0; }
  dir[0] = -P->p[0].v[0]; dir[1] = -P->p[0].v[1]; dir[2] = -P->p[0].v[2];
  normalize3(dir);
  md_support(a, b, dir, &P->p[1]);
  P->size = 2;
  d = dot3(P->p[1].v, dir);
  if (is_zero(d) || d < 0) return -1;
  cross3(dir, P->p[0].v, P->p[1].v);
  if (is_zero(dot3(dir, dir))) return vec_eq(P->p[1].v, origin) ? 1 : 2;
  normalize3(dir);
  md_support(a, b, dir, &P->p[2]);
  d = dot3(P->p[2].v, dir);
  if (is_zero(d) || d < 0) return -1;
  P->size = 3;
  sub3(va, P->p[1].v, P->p[0].v);
  sub3(vb, P->p[2].v, P->p[0].v);
  cross3(dir, va, vb);
  normalize3(dir);
  if (dot3(dir, P->p[0].v) > 0) {
    spt t = P->p[1]; P->p[1] = P->p[2]; P->p[2] = t;
    dir[0] = -dir[0]; dir[1] = -dir[1]; dir[2] = -dir[2];
  }
  int guard = 0;
  while (P->size < 4) {
    if (++guard > 100) return -1;
    md_support(a, b, dir, &P->p[3]);
    d = dot3(P->p[3].v, dir);
    if (is_zero(d) || d < 0) return -1;
    int cont = 0;
    cross3(va, P->p[1].v, P->p[3].v);
    d = dot3(va, P->p[0].v);
    if (d < 0 && !is_zero(d)) { P->p[2] = P->p[3]; cont = 1; }
    if (!cont) {
      cross3(va, P->p[3].v, P->p[2].v);
      d = dot3(va, P->p[0].v);
      if (d < 0 && !is_zero(d)) { P->p[1] = P->p[3]; cont = 1; }
    }
    if (cont) {
      sub3(va, P->p[1].v, P->p[0].v);
      sub3(vb, P->p[2].v, P->p[0].v);
      cross3(dir, va, vb);
      normalize3(dir);
    } else P->size = 4;
  }
  return 0;
}
static int refine_portal(cobj *a, cobj *b, portal_t *P, double tol, int maxit, int64_t *iters) {
  double dir[3];
  spt v4;
  for (int it = 0; it <= maxit + 1; it++) {
    portal_dir(P, dir);
    double d = dot3(dir, P->p[1].v);
    if (is_zero(d) || d > 0) return 0; /* portal encapsulates the origin */
    md_support(a, b, dir, &v4);
    (*iters)++;
    d = dot3(v4.v, dir);
    if (!(is_zero(d) || d > 0) || reach_tolerance(P, &v4, dir, tol)) return -1;
    expand_portal(P, &v4);
  }
  return -1;
}
/* squared distance from the origin to triangle (a,b,c) and the closest point */
static double origin_tri_dist2(const double *a, const double *b, const double *c, double *w) {
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(ab, b, a);
  sub3(ac, c, a);
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { memcpy(w, a, 24); return dot3(w, w); }
  double bp[3] = {-b[0], -b[1], -b[2]};
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { memcpy(w, b, 24); return dot3(w, w); }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    double v = d1 / (d1 - d3);
    for (int k = 0; k < 3; k++) w[k] = a[k] + v * ab[k];
    return dot3(w, w);
  }
  double cp[3] = {-c[0], -c[1], -c[2]};
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { memcpy(w, c, 24); return dot3(w, w); }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    double v = d2 / (d2 - d6);
    for (int k = 0; k < 3; k++) w[k] = a[k] + v * ac[k];
    return dot3(w, w);
  }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    for (int k = 0; k < 3; k++) w[k] = b[k] + v * (c[k] - b[k]);
    return dot3(w, w);
  }
  double den = 1.0 / (va + vb + vc), v = vb * den, u = vc * den;
  for (int k = 0; k < 3; k++) w[k] = a[k] + ab[k] * v + ac[k] * u;
  return dot3(w, w);
}
static void find_pos(const portal_t *P, double *pos) {
  double dir[3], t[3], bb[4], sum;
  portal_dir(P, dir);
  cross3(t, P->p[1].v, P->p[2].v); bb[0] = dot3(t, P->p[3].v);
  cross3(t, P->p[3].v, P->p[2].v); bb[1] = dot3(t, P->p[0].v);
  cross3(t, P->p[0].v, P->p[1].v); bb[2] = dot3(t, P->p[3].v);
  cross3(t, P->p[2].v, P->p[1].v); bb[3] = dot3(t, P->p[0].v);
  sum = bb[0] + bb[1] + bb[2] + bb[3];
  if (is_zero(sum) || sum < 0) {
    bb[0] = 0;
    cross3(t, P->p[2].v, P->p[3].v); bb[1] = dot3(t, dir);
    cross3(t, P->p[3].v, P->p[1].v); bb[2] = dot3(t, dir);
    cross3(t, P->p[1].v, P->p[2].v); bb[3] = dot3(t, dir);
    sum = bb[1] + bb[2] + bb[3];
  }
  double inv = 1.0 / sum;
  for (int k = 0; k < 3; k++) {
    double p1 = 0, p2 = 0;
    for (int i = 0; i < 4; i++) { p1 += bb[i] * P->p[i].v1[k]; p2 += bb[i] * P->p[i].v2[k]; }
    pos[k] = 0.5 * (p1 * inv + p2 * inv);
  }
}
/* returns 0 when the inflated hulls intersect: depth, unit direction (obj1 -> obj2), position */
static int mpr_penetration(cobj *a, cobj *b, double tol, int maxit, double *depth, double *dir, double *pos,
                           int64_t *iters) {
  portal_t P;
  int res = discover_portal(a, b, &P);
  if (res < 0) return -1;
  if (res == 1) { /* touching contact: no direction */
    *depth = 0; dir[0] = dir[1] = dir[2] = 0;
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (P.p[1].v1[k] + P.p[1].v2[k]);
    return 0;
  }
  if (res == 2) { /* origin on the segment v0-v1 */
    *depth = norm3(P.p[1].v);
    memcpy(dir, P.p[1].v, 24);
    normalize3(dir);
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (P.p[1].v1[k] + P.p[1].v2[k]);
    return 0;
  }
  if (refine_portal(a, b, &P, tol, maxit, iters) < 0) return -1;
  int it = 0;
  for (;;) {
    double d[3];
    spt v4;
    portal_dir(&P, d);
    md_support(a, b, d, &v4);
    (*iters)++;
    if (reach_tolerance(&P, &v4, d, tol) || it > maxit) {
      double w[3];
      *depth = sqrt(origin_tri_dist2(P.p[1].v, P.p[2].v, P.p[3].v, w));
      if (is_zero(*depth)) dir[0] = dir[1] = dir[2] = 0;
      else { memcpy(dir, w, 24); normalize3(dir); }
      find_pos(&P, pos);
      return 0;
    }
    expand_portal(&P, &v4);
    it++;
  }
}

static void make_cobj(const orc_model *m, const orc_data *d, int g, double margin, cobj *o) {
  memset(o, 0, sizeof *o);
  o->type = m->geom_type[g];
  o->size = m->geom_size + 3 * g;
  if (o->type == GEOM_MESH) {
    int mesh = m->geom_dataid[g];
    o->vertadr = m->mesh_vertadr[mesh];
    o->nvert = m->mesh_vertnum[mesh];
    o->vert = m->hull_vert + 3 * (size_t)o->vertadr;
    o->adj_start = m->hull_adj_start;
    o->adj = m->hull_adj;
  }
  o->xpos = d->geom_xpos + 3 * g;
  o->xmat = d->geom_xmat + 9 * g;
  o->margin = margin;
  o->cache = 0;
}

/* mju_makeFrame: complete a right-handed frame from its (unit) first axis */
static void make_frame(double *f) {
  double *x = f, *y = f + 3, *z = f + 6;
  normalize3(x);
  y[0] = y[1] = y[2] = 0;
  if (x[1] < 0.5 && x[1] > -0.5) y[1] = 1;
  else y[2] = 1;
  double t = dot3(x, y);
  for (int k = 0; k < 3; k++) y[k] -= t * x[k];
  normalize3(y);
  cross3(z, x, y);
}

/* mesh-mesh: 1 contact or none.  `dist`, `pos`, `normal` as MuJoCo's mjContact fields */
static int convex_pair(const orc_model *m, const orc_data *d, int g1, int g2, double margin, double *dist,
                       double *pos, double *normal, int64_t *iters) {
  cobj a, b;
  make_cobj(m, d, g1, 0.5 * margin, &a);
  make_cobj(m, d, g2, 0.5 * margin, &b);
  double depth, dir[3];
  if (mpr_penetration(&a, &b, m->mpr_tolerance, m->mpr_iterations, &depth, dir, pos, iters) != 0) return 0;
  if (dir[0] == 0 && dir[1] == 0 && dir[2] == 0) return 0;
  *dist = margin - depth;
  memcpy(normal, dir, 24);
  return 1;
}
/* plane (g1) - convex (g2): the deepest point of the geom (its support along -normal), contact at
 * the midpoint.  For a sphere this is mjc_PlaneSphere's closed form. */
static int plane_pair(const orc_model *m, const orc_data *d, int g1, int g2, double margin, double *dist,
                      double *pos, double *normal) {
  cobj b;
  make_cobj(m, d, g2, 0.0, &b);
  const double *pm = d->geom_xmat + 9 * g1, *pp = d->geom_xpos + 3 * g1;
  double n[3] = {pm[2], pm[5], pm[8]}, nd[3] = {-pm[2], -pm[5], -pm[8]}, v[3], df[3];
  geom_support(&b, nd, v);
  sub3(df, v, pp);
  double ds = dot3(df, n);
  if (ds > margin) return 0;
  *dist = ds;
  for (int k = 0; k < 3; k++) pos[k] = v[k] - 0.5 * ds * n[k];
  memcpy(normal, n, 24);
  return 1;
}
/* plane (g1) - point of g2 with radius r: the shared closed form of plane-capsule (cap centres,
 * r = capsule radius) and plane-box (corners, r = 0) */
static int plane_point(const orc_data *d, int g1, const double *c, double r, double margin, double *dist, double *pos,
                       double *normal) {
  const double *pm = d->geom_xmat + 9 * g1, *pp = d->geom_xpos + 3 * g1;
  double n[3] = {pm[2], pm[5], pm[8]}, df[3];
  sub3(df, c, pp);
  double ds = dot3(df, n) - r;
  if (ds > margin) return 0;
  *dist = ds;
  for (int k = 0; k < 3; k++) pos[k] = c[k] - (r + 0.5 * ds) * n[k];
  memcpy(normal, n, 24);
  return 1;
}
/* sphere (c1, r1) - sphere (c2, r2): mjc_SphereSphere's closed form; normal from 1 to 2 */
static int sphere_sphere(const double *c1, double r1, const double *c2, double r2, double margin, double *dist, double *pos,
                         double *normal) {
  double df[3];
  sub3(df, c2, c1);
  double len = norm3(df);
  if (len - r1 - r2 > margin) return 0;
  if (len < 1e-15) { normal[0] = 1; normal[1] = normal[2] = 0; }
  else for (int k = 0; k < 3; k++) normal[k] = df[k] / len;
  *dist = len - r1 - r2;
  for (int k = 0; k < 3; k++) pos[k] = c1[k] + (r1 + 0.5 * *dist) * normal[k];
  return 1;
}
/* point of the segment c +- h * axis closest to p */
static void seg_closest_point(const double *c, const double *axis, double h, const double *p, double *out) {
  double df[3];
  sub3(df, p, c);
  double t = dot3(df, axis);
  t = t > h ? h : (t < -h ? -h : t);
  for (int k = 0; k < 3; k++) out[k] = c[k] + t * axis[k];
}
/* closest points of two segments c1 +- h1 a1, c2 +- h2 a2 (unit axes) */
static void seg_seg_closest(const double *c1, const double *a1, double h1, const double *c2, const double *a2, double h2,
                            double *p1, double *p2) {
  double w[3];
  sub3(w, c1, c2);
  double b = dot3(a1, a2), d1 = dot3(a1, w), d2 = dot3(a2, w), den = 1 - b * b, t1, t2;
  if (den > 1e-12) {
    t1 = (b * d2 - d1) / den;
    t1 = t1 > h1 ? h1 : (t1 < -h1 ? -h1 : t1);
  } else t1 = 0; /* parallel axes: start from the middle of segment 1 */
  t2 = b * t1 + d2;
  if (t2 > h2 || t2 < -h2) {
    t2 = t2 > h2 ? h2 : -h2;
    t1 = b * t2 - d1;
    t1 = t1 > h1 ? h1 : (t1 < -h1 ? -h1 : t1);
  }
  for (int k = 0; k < 3; k++) { p1[k] = c1[k] + t1 * a1[k]; p2[k] = c2[k] + t2 * a2[k]; }
}

/* capsule - capsule, sub-contact `sub` (mjc_CapsuleCapsule's shape [EXT]): axes in general position give ONE contact
 * between the closest points of the two segments; (near-)parallel axes give up to TWO — the end points of each
 * segment against the closest point of the other one, tested in the order +end / -end of capsule 1, then of capsule 2,
 * until two have been found — so that a capsule resting on a parallel one is supported at both ends of the overlap */
static int capsule_capsule(const double *c1, const double *a1, double h1, double r1, const double *c2, const double *a2,
                           double h2, double r2, double margin, int sub, double *dist, double *pos, double *normal) {
  double w[3];
  sub3(w, c1, c2);
  double b = dot3(a1, a2), d1 = dot3(a1, w), d2 = dot3(a2, w), den = 1 - b * b;
  if (den > 1e-12) {
    if (sub != 0) return 0;
    double p1[3], p2[3];
    seg_seg_closest(c1, a1, h1, c2, a2, h2, p1, p2);
    return sphere_sphere(p1, r1, p2, r2, margin, dist, pos, normal);
  }
  int found = 0;
  for (int k = 0; k < 4 && found < 2; k++) {
    double t1, t2, p1[3], p2[3];
    if (k < 2) { /* an end of segment 1 and the point of segment 2 closest to it */
      t1 = k == 0 ? h1 : -h1;
      t2 = d2 + b * t1;
      t2 = t2 > h2 ? h2 : (t2 < -h2 ? -h2 : t2);
    } else { /* an end of segment 2 and the point of segment 1 closest to it */
      t2 = k == 2 ? h2 : -h2;
      t1 = b * t2 - d1;
      t1 = t1 > h1 ? h1 : (t1 < -h1 ? -h1 : t1);
    }
    for (int q = 0; q < 3; q++) { p1[q] = c1[q] + t1 * a1[q]; p2[q] = c2[q] + t2 * a2[q]; }
    if (!sphere_sphere(p1, r1, p2, r2, margin, dist, pos, normal)) continue;
    if (found == sub) return 1;
    found++;
  }
  return 0;
}

/* contacts a candidate pair can produce (sub-contacts are numbered 0 .. n-1) */
int orc_pair_nsub(const orc_model *m, int g1, int g2) {
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE) return 2;
  if (t1 == GEOM_PLANE && t2 == GEOM_BOX) return 8;
  if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) return 2;
  return 1;
}
/* sub-contact `sub` of the pair (g1, g2): 1 when it exists.  Counts MPR work in iters / npair. */
static int narrow_sub(const orc_model *m, const orc_data *d, int g1, int g2, int sub, double margin, double *dist,
                      double *pos, double *normal, int64_t *iters, int64_t *npair) {
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const double *x1 = d->geom_xpos + 3 * g1, *x2 = d->geom_xpos + 3 * g2;
  const double *R1 = d->geom_xmat + 9 * g1, *R2 = d->geom_xmat + 9 * g2;
  const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
  if (t1 == GEOM_HFIELD || t2 == GEOM_HFIELD || t2 == GEOM_PLANE) return 0;
  if (t1 == GEOM_PLANE) {
    if (t2 == GEOM_CAPSULE) { /* cap centre +axis first, then -axis */
      double sg = sub == 0 ? s2[1] : -s2[1], c[3] = {x2[0] + sg * R2[2], x2[1] + sg * R2[5], x2[2] + sg * R2[8]};
      return plane_point(d, g1, c, s2[0], margin, dist, pos, normal);
    }
    if (t2 == GEOM_BOX) {
      /* corner `sub`: bit 0 / 1 / 2 set = positive x / y / z side (mjc_PlaneBox's corner order [EXT]).  A corner
       * makes a contact when it is on the plane's side of the box centre (ldist <= 0) and below the margin, and
       * fewer than four corners before it did: MuJoCo stops at four */
      const double *pm = d->geom_xmat + 9 * g1, *pp = d->geom_xpos + 3 * g1;
      double n[3] = {pm[2], pm[5], pm[8]};
      int before = 0, mine = 0;
      double cs[3] = {0, 0, 0};
      for (int j = 0; j <= sub; j++) {
        double v[3] = {j & 1 ? s2[0] : -s2[0], j & 2 ? s2[1] : -s2[1], j & 4 ? s2[2] : -s2[2]}, o[3], c[3], df[3];
        for (int r = 0; r < 3; r++) { o[r] = R2[3 * r] * v[0] + R2[3 * r + 1] * v[1] + R2[3 * r + 2] * v[2]; c[r] = x2[r] + o[r]; }
        sub3(df, c, pp);
        int ok = !(dot3(df, n) > margin) && !(dot3(n, o) > 0);
        if (j < sub) before += ok;
        else { mine = ok; memcpy(cs, c, 24); }
      }
      if (!mine || before >= 4) return 0;
      return plane_point(d, g1, cs, 0.0, margin, dist, pos, normal);
    }
    return plane_pair(m, d, g1, g2, margin, dist, pos, normal);
  }
  if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) return sphere_sphere(x1, s1[0], x2, s2[0], margin, dist, pos, normal);
  if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) {
    double ax[3] = {R2[2], R2[5], R2[8]}, p[3];
    seg_closest_point(x2, ax, s2[1], x1, p);
    return sphere_sphere(x1, s1[0], p, s2[0], margin, dist, pos, normal);
  }
  if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) {
    double a1[3] = {R1[2], R1[5], R1[8]}, a2[3] = {R2[2], R2[5], R2[8]};
    return capsule_capsule(x1, a1, s1[1], s1[0], x2, a2, s2[1], s2[0], margin, sub, dist, pos, normal);
  }
  (*npair)++;
  return convex_pair(m, d, g1, g2, margin, dist, pos, normal, iters);
}

int orc_narrow_pair(const orc_model *m, const orc_data *d, int g1, int g2, double margin, double *dist,
                    double *pos, double *normal) {
  int64_t it = 0, np = 0;
  return narrow_sub(m, d, g1, g2, 0, margin, dist, pos, normal, &it, &np);
}

void orc_collision(const orc_model *m, orc_data *d) {
  d->ncon = 0;
  if (m->disableflags & DSBL_CONTACT) return;
  int full = 0;
  for (int p = 0; p < m->npair && !full; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
    int t1 = m->geom_type[g1];
    /* broadphase: bounding spheres (a plane has rbound 0: sphere-plane test instead) */
    if (m->geom_rbound[g1] > 0 && m->geom_rbound[g2] > 0) {
      double df[3];
      sub3(df, d->geom_xpos + 3 * g1, d->geom_xpos + 3 * g2);
      if (norm3(df) > margin + m->geom_rbound[g1] + m->geom_rbound[g2]) continue;
    } else if (t1 == GEOM_PLANE && m->geom_rbound[g2] > 0) {
      const double *pm = d->geom_xmat + 9 * g1;
      double n[3] = {pm[2], pm[5], pm[8]}, df[3];
      sub3(df, d->geom_xpos + 3 * g2, d->geom_xpos + 3 * g1);
      if (dot3(df, n) > margin + m->geom_rbound[g2]) continue;
    }
    const int nsub = orc_pair_nsub(m, g1, g2);
    for (int sub = 0; sub < nsub; sub++) {
    double dist, pos[3], nrm[3];
    if (!narrow_sub(m, d, g1, g2, sub, margin, &dist, pos, nrm, &d->stat_mpr_iter, &d->stat_npair_narrow)) continue;
    if (d->ncon >= ORC_MAXCON) { d->warning |= 8; full = 1; break; }
    orc_contact *c = d->contact + d->ncon++;
    memset(c, 0, sizeof *c);
    c->dist = dist;
    memcpy(c->pos, pos, 24);
    memcpy(c->frame, nrm, 24);
    make_frame(c->frame);
    c->includemargin = margin - gap;
    c->geom1 = g1; c->geom2 = g2;
    c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    const double *f1 = m->geom_friction + 3 * g1, *f2 = m->geom_friction + 3 * g2;
    double fr[3];
    for (int k = 0; k < 3; k++) fr[k] = fmax(f1[k], f2[k]);
    c->friction[0] = c->friction[1] = fr[0]; c->friction[2] = fr[1]; c->friction[3] = c->friction[4] = fr[2];
    double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
    if (s1 >= 1e-15 && s2 >= 1e-15) mix = s1 / (s1 + s2);
    else if (s1 < 1e-15 && s2 < 1e-15) mix = 0.5;
    else mix = s1 < 1e-15 ? 0.0 : 1.0;
    for (int k = 0; k < 2; k++) c->solref[k] = mix * m->geom_solref[2 * g1 + k] + (1 - mix) * m->geom_solref[2 * g2 + k];
    for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
    c->efc_address = -1;
    }
  }
}
