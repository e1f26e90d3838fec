// gfx950 device code of the path tracer, templated on the arithmetic type R.
//
//   R = double : parity mode.  Expressions keep the reference's operand order and the
//                translation unit is compiled with -ffp-contract=off, so closest hits are
//                bit-identical to the CPU renderer.
//   R = float  : production mode (contraction allowed).
//
// Reference lines are cited per routine; `ref:` paths are relative to the reference root.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "pt_types.hpp"

namespace rtx {

#define RTX_DEV __device__ __forceinline__

template <class R> struct V3 { R x, y, z; };

template <class R> RTX_DEV V3<R> mk3(const R* p) { return {p[0], p[1], p[2]}; }
template <class R> RTX_DEV V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> RTX_DEV V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> RTX_DEV V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class R> RTX_DEV V3<R> operator*(V3<R> a, R t) { return {a.x * t, a.y * t, a.z * t}; }
template <class R> RTX_DEV V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
// ref: core/vec3.c:37-63
template <class R> RTX_DEV R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> RTX_DEV V3<R> cross(V3<R> a, V3<R> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> RTX_DEV R len2(V3<R> v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
RTX_DEV float rsqrt_(float x) { return __frsqrt_rn(x); }
RTX_DEV float sqrt_(float x) { return __fsqrt_rn(x); }
RTX_DEV double sqrt_(double x) { return sqrt(x); }
RTX_DEV float abs_(float x) { return fabsf(x); }
RTX_DEV double abs_(double x) { return fabs(x); }
RTX_DEV float min_(float a, float b) { return fminf(a, b); }
RTX_DEV double min_(double a, double b) { return fmin(a, b); }
RTX_DEV float max_(float a, float b) { return fmaxf(a, b); }
RTX_DEV double max_(double a, double b) { return fmax(a, b); }
template <class R> RTX_DEV R len(V3<R> v) { return sqrt_(len2(v)); }
template <class R> RTX_DEV V3<R> normalize(V3<R> v) {
  R l = len(v);
  if (l > R(0)) return v * (R(1) / l);
  return v;
}
// ref: core/vec3.c:65-83
template <class R> RTX_DEV V3<R> reflect(V3<R> v, V3<R> n) { return v - n * (R(2) * dot(v, n)); }
template <class R> RTX_DEV V3<R> refract(V3<R> uv, V3<R> n, R eta) {
  R ct = min_(dot(-uv, n), R(1));
  V3<R> perp = (uv + n * ct) * eta;
  R par = -sqrt_(abs_(R(1) - len2(perp)));
  return perp + n * par;
}
template <class R> RTX_DEV V3<R> lerp3(V3<R> a, V3<R> b, R t) { return a * (R(1) - t) + b * t; }

template <class R> struct Consts;
template <> struct Consts<float> { static constexpr float pi = 3.14159265358979323846f; };
template <> struct Consts<double> { static constexpr double pi = 3.14159265358979323846; };

RTX_DEV void sincos_(float x, float* s, float* c) { __sincosf(x, s, c); }
RTX_DEV void sincos_(double x, double* s, double* c) { sincos(x, s, c); }
// sin/cos of 2*pi*u for u in [0,1): the hardware sine takes its argument in revolutions.
RTX_DEV void sincos_turns(float u, float* s, float* c) {
  *s = __builtin_amdgcn_sinf(u);
  *c = __builtin_amdgcn_cosf(u);
}
RTX_DEV void sincos_turns(double u, double* s, double* c) { sincos(2.0 * Consts<double>::pi * u, s, c); }

template <class R> struct Ray { V3<R> o, d; };
template <class R> RTX_DEV V3<R> at(const Ray<R>& r, R t) { return r.o + r.d * t; }  // ref: core/ray.c:12

// The winner of a closest-hit query.  Point/normal are derived from it afterwards, once,
// instead of after every successful primitive test.
template <class R> struct Cand {
  R t;
  int type, idx;
  // hit_record::u/v survive across primitive tests in the reference (scene.hpp:575,629): a
  // plane or box hit reports the u,v of the last sphere/triangle/quad/disk test that
  // succeeded before it.  uv_kind: 0 none (0,0), 1 values in (ua, ub), 2 sphere `uv_idx` at t = ua.
  R ua, ub;
  int uv_kind, uv_idx;
  int tied;   // the walk decided an exact tie in t without the reference's leaf order (tie_goes_to)
};

template <class R> struct Hit {
  V3<R> p, n;
  R t, u, v;
  int front, mat, type, idx;
  int tied;   // Cand::tied of the query that found it
};

// ------------------------------------------------------------------------------------------
// primitive tests: return acceptance and the accepted parameter only
// ------------------------------------------------------------------------------------------

// ref: engine/bvh.hpp:35-64 (AABB::hit).  The ternaries keep the old bound when the slab
// value is NaN (0 * inf), exactly as the reference's do.
template <class R> RTX_DEV bool slab_hit(const R* lo, const R* hi, const Ray<R>& r, R t_min, R t_max) {
  R inv = R(1) / r.d.x;
  R t0 = (lo[0] - r.o.x) * inv, t1 = (hi[0] - r.o.x) * inv;
  if (inv < R(0)) { R s = t0; t0 = t1; t1 = s; }
  t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
  if (t_max <= t_min) return false;
  inv = R(1) / r.d.y;
  t0 = (lo[1] - r.o.y) * inv; t1 = (hi[1] - r.o.y) * inv;
  if (inv < R(0)) { R s = t0; t0 = t1; t1 = s; }
  t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
  if (t_max <= t_min) return false;
  inv = R(1) / r.d.z;
  t0 = (lo[2] - r.o.z) * inv; t1 = (hi[2] - r.o.z) * inv;
  if (inv < R(0)) { R s = t0; t0 = t1; t1 = s; }
  t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
  if (t_max <= t_min) return false;
  return true;
}

// ref: engine/sphere.hpp:71-92
template <class R> RTX_DEV bool sphere_t(const DSphere<R>& s, const Ray<R>& r, R t_min, R t_max, R& t_out) {
  V3<R> oc = r.o - mk3(s.c);
  R a = len2(r.d);
  R half_b = dot(oc, r.d);
  R c = len2(oc) - s.radius * s.radius;
  R disc = half_b * half_b - a * c;
  if (disc < R(0)) return false;
  R sq = sqrt_(disc);
  R root = (-half_b - sq) / a;
  if (root < t_min || root > t_max) {
    root = (-half_b + sq) / a;
    if (root < t_min || root > t_max) return false;
  }
  t_out = root;
  return true;
}

// ref: engine/primitives.hpp:35-48
template <class R> RTX_DEV bool plane_t(const DPlane<R>& pl, const Ray<R>& r, R t_min, R t_max, R& t_out) {
  V3<R> n = mk3(pl.n);
  R denom = dot(n, r.d);
  if (abs_(denom) < R(1e-8)) return false;
  R t = dot(mk3(pl.p) - r.o, n) / denom;
  if (t < t_min || t > t_max) return false;
  t_out = t;
  return true;
}

// ref: engine/primitives.hpp:78-110 (Moller-Trumbore, two-sided)
template <class R> RTX_DEV bool tri_t(const DTri<R>& tr, const Ray<R>& r, R t_min, R t_max, R& t_out, R& u_out, R& v_out) {
  V3<R> v0 = mk3(tr.v0);
  V3<R> e1 = mk3(tr.v1) - v0, e2 = mk3(tr.v2) - v0;
  V3<R> h = cross(r.d, e2);
  R a = dot(e1, h);
  if (abs_(a) < R(1e-8)) return false;
  R f = R(1) / a;
  V3<R> s = r.o - v0;
  R u = f * dot(s, h);
  if (u < R(0) || u > R(1)) return false;
  V3<R> q = cross(s, e1);
  R v = f * dot(r.d, q);
  if (v < R(0) || u + v > R(1)) return false;
  R t = f * dot(e2, q);
  if (t < t_min || t > t_max) return false;
  t_out = t; u_out = u; v_out = v;
  return true;
}

// ref: engine/primitives.hpp:153-181.  The reported parameter is the slab entry, or the
// caller's t_min when the origin is inside the box.
template <class R> RTX_DEV bool box_t(const DBox<R>& b, const Ray<R>& r, R t_min, R t_max, R& t_out) {
  R tmin = t_min, tmax = t_max;
  const R o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    R inv = R(1) / d[ax];
    R t0 = (b.lo[ax] - o[ax]) * inv, t1 = (b.hi[ax] - o[ax]) * inv;
    if (inv < R(0)) { R s = t0; t0 = t1; t1 = s; }
    tmin = t0 > tmin ? t0 : tmin; tmax = t1 < tmax ? t1 : tmax;
    if (tmax <= tmin) return false;
  }
  t_out = tmin;
  return true;
}

// ref: engine/primitives.hpp:274-300
template <class R> RTX_DEV bool quad_t(const DQuad<R>& q, const Ray<R>& r, R t_min, R t_max, R& t_out, R& u_out, R& v_out) {
  V3<R> n = mk3(q.n);
  R denom = dot(n, r.d);
  if (abs_(denom) < R(1e-8)) return false;
  V3<R> corner = mk3(q.corner);
  R t = dot(corner - r.o, n) / denom;
  if (t < t_min || t > t_max) return false;
  V3<R> d = at(r, t) - corner;
  V3<R> eu = mk3(q.eu), ev = mk3(q.ev);
  R up = dot(d, eu) / len2(eu), vp = dot(d, ev) / len2(ev);
  if (up < R(0) || up > R(1) || vp < R(0) || vp > R(1)) return false;
  t_out = t; u_out = up; v_out = vp;
  return true;
}

// ref: engine/primitives.hpp:396-430
template <class R> RTX_DEV bool disk_t(const DDisk<R>& dl, const Ray<R>& r, R t_min, R t_max, R& t_out, R& u_out, R& v_out) {
  V3<R> n = mk3(dl.n);
  R denom = dot(n, r.d);
  if (abs_(denom) < R(1e-8)) return false;
  V3<R> c = mk3(dl.c);
  R t = dot(c - r.o, n) / denom;
  if (t < t_min || t > t_max) return false;
  V3<R> d = at(r, t) - c;
  R d2 = len2(d);
  if (d2 > dl.radius * dl.radius) return false;
  R dist = sqrt_(d2);
  u_out = (dist > R(1e-8)) ? (dot(d, mk3(dl.tan)) / dist + R(1)) * R(0.5) : R(0.5);
  v_out = (dist > R(1e-8)) ? (dot(d, mk3(dl.bit)) / dist + R(1)) * R(0.5) : R(0.5);
  t_out = t;
  return true;
}

template <class R> RTX_DEV void cand_init(Cand<R>& c, R t_max) {
  c.t = t_max; c.type = -1; c.idx = -1; c.ua = R(0); c.ub = R(0); c.uv_kind = 0; c.uv_idx = -1; c.tied = 0;
}

// One bounded primitive (sphere / triangle / box) against the current candidate.
template <class R> RTX_DEV void test_bounded(const DScene<R>& sc, int type, int idx, const Ray<R>& r, R t_min, Cand<R>& c) {
  R t, u, v;
  if (type == PRIM_SPHERE) {
    if (sphere_t(sc.spheres[idx], r, t_min, c.t, t)) { c.t = t; c.type = PRIM_SPHERE; c.idx = idx; c.uv_kind = 2; c.ua = t; c.uv_idx = idx; }
  } else if (type == PRIM_TRIANGLE) {
    if (tri_t(sc.tris[idx], r, t_min, c.t, t, u, v)) { c.t = t; c.type = PRIM_TRIANGLE; c.idx = idx; c.uv_kind = 1; c.ua = u; c.ub = v; }
  } else if (type == PRIM_BOX) {
    if (box_t(sc.boxes[idx], r, t_min, c.t, t)) { c.t = t; c.type = PRIM_BOX; c.idx = idx; }
  }
}

// Planes, quad lights, disk lights: the linear tail of Scene::hit (ref: engine/scene.hpp:640-671).
template <class R> RTX_DEV void test_unbounded(const DScene<R>& sc, const Ray<R>& r, R t_min, Cand<R>& c) {
  R t, u, v;
  for (int i = 0; i < sc.n_planes; ++i)
    if (plane_t(sc.planes[i], r, t_min, c.t, t)) { c.t = t; c.type = PRIM_PLANE; c.idx = i; }
  for (int i = 0; i < sc.n_quads; ++i)
    if (quad_t(sc.quads[i], r, t_min, c.t, t, u, v)) { c.t = t; c.type = PRIM_QUAD; c.idx = i; c.uv_kind = 1; c.ua = u; c.ub = v; }
  for (int i = 0; i < sc.n_disks; ++i)
    if (disk_t(sc.disks[i], r, t_min, c.t, t, u, v)) { c.t = t; c.type = PRIM_DISK; c.idx = i; c.uv_kind = 1; c.ua = u; c.ub = v; }
}

template <class R> RTX_DEV void face_normal(Hit<R>& h, const Ray<R>& r, V3<R> outward) {  // ref: core/hit.c:21-24
  h.front = dot(r.d, outward) < R(0);
  h.n = h.front ? outward : -outward;
}

// ref: engine/sphere.hpp:26-32 (get_sphere_uv)
template <class R> RTX_DEV void sphere_uv(V3<R> on, R& u, R& v) {
  R theta = acos(-on.y);
  R phi = atan2(-on.z, on.x) + Consts<R>::pi;
  u = phi / (R(2) * Consts<R>::pi);
  v = theta / Consts<R>::pi;
}

// Point, normal, face and material of the winner (the writes the reference's hit routines
// make on success: sphere.hpp:94-98, primitives.hpp:50-53,112-116,180-195,302-305,419-422,
// scene.hpp:657,669).  WANT_UV also resolves hit_record::u/v.
template <class R, bool WANT_UV> RTX_DEV void finalize_hit(const DScene<R>& sc, const Ray<R>& r, const Cand<R>& c, Hit<R>& h) {
  h.t = c.t; h.type = c.type; h.idx = c.idx; h.tied = 0;
  h.p = at(r, c.t);
  h.u = R(0); h.v = R(0);
  switch (c.type) {
    case PRIM_SPHERE: {
      const DSphere<R>& s = sc.spheres[c.idx];
      face_normal(h, r, (h.p - mk3(s.c)) * (R(1) / s.radius));
      h.mat = s.mat;
      break;
    }
    case PRIM_PLANE: {
      const DPlane<R>& pl = sc.planes[c.idx];
      face_normal(h, r, mk3(pl.n));
      h.mat = pl.mat;
      break;
    }
    case PRIM_TRIANGLE: {
      const DTri<R>& tr = sc.tris[c.idx];
      V3<R> v0 = mk3(tr.v0);
      face_normal(h, r, normalize(cross(mk3(tr.v1) - v0, mk3(tr.v2) - v0)));
      h.mat = tr.mat;
      break;
    }
    case PRIM_BOX: {
      const DBox<R>& b = sc.boxes[c.idx];
      const R eps = R(1e-4);
      V3<R> on;
      if (abs_(h.p.x - b.lo[0]) < eps) on = {R(-1), R(0), R(0)};
      else if (abs_(h.p.x - b.hi[0]) < eps) on = {R(1), R(0), R(0)};
      else if (abs_(h.p.y - b.lo[1]) < eps) on = {R(0), R(-1), R(0)};
      else if (abs_(h.p.y - b.hi[1]) < eps) on = {R(0), R(1), R(0)};
      else if (abs_(h.p.z - b.lo[2]) < eps) on = {R(0), R(0), R(-1)};
      else on = {R(0), R(0), R(1)};
      face_normal(h, r, on);
      h.mat = b.mat;
      break;
    }
    case PRIM_QUAD: face_normal(h, r, mk3(sc.quads[c.idx].n)); h.mat = -c.idx - 1; break;
    default: face_normal(h, r, mk3(sc.disks[c.idx].n)); h.mat = -1000 - c.idx; break;
  }
  if (WANT_UV) {
    if (c.uv_kind == 1) { h.u = c.ua; h.v = c.ub; }
    else if (c.uv_kind == 2) {
      const DSphere<R>& s = sc.spheres[c.uv_idx];
      V3<R> on = (at(r, c.ua) - mk3(s.c)) * (R(1) / s.radius);
      sphere_uv(on, h.u, h.v);
    }
  }
}

// ------------------------------------------------------------------------------------------
// closest hit / any hit through the production BVH
// ------------------------------------------------------------------------------------------
// Per-lane traversal stack in LDS, laid out [level][lane] so a wave's accesses to one level
// are one conflict-free row.
struct LaneStack {
  int* base;
  static constexpr int stride = kBlock;   // word i of this lane's stack is base[i * stride]: every kernel with stacks runs kBlock-lane workgroups (a constant keeps a multiply out of every push and pop)
  RTX_DEV int& at(int level) const { return base[level * stride]; }
  RTX_DEV void push(int& sp, int v) const { base[sp * stride] = v; ++sp; }
  RTX_DEV int pop(int& sp) const { --sp; return base[sp * stride]; }
  RTX_DEV void put(int i, int v) const { base[i * stride] = v; }
  RTX_DEV int get(int i) const { return base[i * stride]; }
};
// Capacity: the host sizes the LDS part of every launch for the scene's tree.  The double-precision kernels walk the reference's
// binary tree, one parked child per level: kStackDepth (48) words cover a median-split tree of any size the ABI accepts (2^30
// primitives are 30 levels).  The single-precision kernels walk the eight-wide tree with walk_stack_slots(levels) words; a scene
// whose tree is deeper than kMaxWalkLevels is refused at upload.

template <class R> RTX_DEV R slab_entry(const DNodeRef& nd, const Ray<R>& r, V3<R> inv, R t_min, R t_max, bool& ok) {
  R t0 = (R(nd.lo[0]) - r.o.x) * inv.x, t1 = (R(nd.hi[0]) - r.o.x) * inv.x;
  R lo = min_(t0, t1), hi = max_(t0, t1);
  t0 = (R(nd.lo[1]) - r.o.y) * inv.y; t1 = (R(nd.hi[1]) - r.o.y) * inv.y;
  lo = max_(lo, min_(t0, t1)); hi = min_(hi, max_(t0, t1));
  t0 = (R(nd.lo[2]) - r.o.z) * inv.z; t1 = (R(nd.hi[2]) - r.o.z) * inv.z;
  lo = max_(lo, min_(t0, t1)); hi = min_(hi, max_(t0, t1));
  lo = max_(lo, t_min); hi = min_(hi, t_max);
  ok = lo < hi;   // strict: a box that is flat along an axis is never entered, as in the reference (bvh.hpp:43)
  return lo;
}

template <class R, bool ANY> RTX_DEV bool walk_bounded(const DScene<R>& sc, const Ray<R>& r, R t_min, Cand<R>& c, LaneStack st) {
  if (sc.n_prims <= 0) return false;
  if (sc.n_prims <= kLinearMaxPrims) {
    for (int i = 0; i < sc.n_prims; ++i) {
      DPrimRef p = sc.prims[i];
      test_bounded(sc, p.type, p.index, r, t_min, c);
      if (ANY && c.type >= 0) return true;
    }
    return c.type >= 0;
  }
  V3<R> inv = {R(1) / r.d.x, R(1) / r.d.y, R(1) / r.d.z};
  int sp = 0;
  int node = 0;
  for (;;) {
    const DNodeRef nd = sc.nodes[node];
    bool ok;
    slab_entry(nd, r, inv, t_min, c.t, ok);
    bool descend = false;
    if (ok) {
      if (nd.count > 0) {
        for (int i = 0; i < nd.count; ++i) {
          DPrimRef p = sc.prims[nd.first + i];
          test_bounded(sc, p.type, p.index, r, t_min, c);
        }
        if (ANY && c.type >= 0) return true;
      } else {
        // visit the nearer child first; the other goes on the stack
        bool ok_l, ok_r;
        const DNodeRef l = sc.nodes[node + 1];
        const DNodeRef rr = sc.nodes[nd.right];
        R el = slab_entry(l, r, inv, t_min, c.t, ok_l);
        R er = slab_entry(rr, r, inv, t_min, c.t, ok_r);
        if (ok_l && ok_r) {
          int near = el <= er ? node + 1 : nd.right, far = el <= er ? nd.right : node + 1;
          st.push(sp, far);
          node = near; descend = true;
        } else if (ok_l) { node = node + 1; descend = true; }
        else if (ok_r) { node = nd.right; descend = true; }
      }
    }
    if (!descend) {
      if (sp == 0) break;
      node = st.pop(sp);
    }
  }
  return c.type >= 0;
}

// ------------------------------------------------------------------------------------------
// Single-precision walk over the eight-wide tree (DNode8 / DLeaf, pt_types.hpp), cut into resumable steps so that the
// lanes of a persistent kernel can pick up a new ray when theirs is finished.
//   walk8_inner  fetches ONE node record (64 bytes), tests its eight quantised boxes, continues with the nearest
//                child entered and parks the others, nearest first, as ONE stack entry;
//   walk8_leaf   fetches ONE leaf record and tests its primitive (a triangle in place; a sphere or a box through its
//                table).
// w.node is a node record's unit (>= 0), a leaf code ~unit (< 0) or kWalkDone.  A stack entry is the group of siblings still to
// visit under one node, two words: the node's inner_base (where its children's records start), and  up to seven nibbles (8 | slot, next
// first; a nibble of 0 ends the group) | n_inner << 28  (any-hit walks, which take the children in slot order: one bit per child entered
// | n_inner << 28);  a child's code follows from its slot.  One entry per level at most, so levels + 1 entries always
// suffice.  A parked child is not re-tested against the interval the hits found meanwhile have shortened (its children /
// primitives are).  A box is entered on a <= b: the tree's own grouping may make a box flat along an axis, which must not hide
// its primitives.  On an exact tie in t the primitive later in the reference's leaf order wins, whatever the visiting order
// (DScene::rank; best_p = the current winner's primitive number).
// ------------------------------------------------------------------------------------------
struct Walk8 { V3<float> inv; int node, sp, best_p, tied; };

RTX_DEV void walk8_begin(Walk8& w, const Ray<float>& r) {
  // the quantised boxes multiply 1/d by box units: keep it finite (a direction component below 1e-18 moves the ray by less
  // than 1e-9 over the longest ray, so the slab of that axis is constant along it either way)
  w.inv = {fminf(fmaxf(1.0f / r.d.x, -1e18f), 1e18f), fminf(fmaxf(1.0f / r.d.y, -1e18f), 1e18f), fminf(fmaxf(1.0f / r.d.z, -1e18f), 1e18f)};
  w.node = 0; w.sp = 0; w.best_p = -1; w.tied = 0;
}
RTX_DEV bool walk8_at_inner(const Walk8& w) { return w.node >= 0 && w.node != kWalkDone; }
RTX_DEV bool walk8_at_leaf(const Walk8& w) { return w.node < 0; }

RTX_DEV uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
RTX_DEV uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }

// Next parked child, or kWalkDone.  SORTED: the entry holds nibbles, nearest child first; otherwise one bit per child entered,
// taken lowest slot first.
template <bool SORTED> RTX_DEV int walk8_pop(Walk8& w, LaneStack st) {
  if (w.sp == 0) return kWalkDone;
  const int e = 2 * (w.sp - 1);
  const int ib = st.get(e);
  const uint32_t pk = (uint32_t)st.get(e + 1);
  const int ni = (int)(pk >> 28);
  int slot;
  if constexpr (SORTED) {
    slot = (int)(pk & 7u);
    const uint32_t rest = (pk & 0x0fffffffu) >> 4;
    if (!(rest & 8u)) --w.sp;                        // the next nibble names no child: the group is used up
    else st.put(e + 1, (int)(rest | (pk & 0xf0000000u)));
  } else {
    slot = __builtin_ctz(pk);                        // (an entry is kept only while one of its eight low bits is set)
    const uint32_t rest = pk & (pk - 1u);
    if (!(rest & 0xffu)) --w.sp;
    else st.put(e + 1, (int)rest);
  }
  return slot < ni ? ib + slot : ~(ib + slot);   // child k's record is unit inner_base + k; the leaf records follow the inner children
}

// SORTED = false (any-hit queries): the children entered are visited in slot order, the sorting network is left out.
template <bool SORTED = true> RTX_DEV void walk8_inner(const DScene<float>& sc, const Ray<float>& r, float t_min, float t_best, Walk8& w, LaneStack st) {
  const uint4* rec = reinterpret_cast<const uint4*>(sc.nodes8 + w.node);
  const uint4 h0 = rec[0], h1 = rec[1], h2 = rec[2], h3 = rec[3];
  const uint32_t exps = h0.z;
  // the node's origin: a point of the scene's 2^21-step grid
  const float ox = fmaf((float)(h0.x & 0x1fffffu), sc.wt_step[0], sc.wt_anchor[0]);
  const float oy = fmaf((float)((h0.x >> 21) | ((h0.y & 0x3ffu) << 11)), sc.wt_step[1], sc.wt_anchor[1]);
  const float oz = fmaf((float)((h0.y >> 10) & 0x1fffffu), sc.wt_step[2], sc.wt_anchor[2]);
  // slab distance of a quantised plane q:  (org + q * unit - o) / d  =  q * (unit / d) + (org - o) / d
  const float ax = __uint_as_float((exps & 0xffu) << 23) * w.inv.x, ay = __uint_as_float(((exps >> 8) & 0xffu) << 23) * w.inv.y,
              az = __uint_as_float(((exps >> 16) & 0xffu) << 23) * w.inv.z;
  const float bx = (ox - r.o.x) * w.inv.x, by = (oy - r.o.y) * w.inv.y, bz = (oz - r.o.z) * w.inv.z;
  // The ray enters a slab through its lower plane when it travels up the axis, through its upper plane otherwise: which of
  // the two plane sets is "near" is chosen once per node and axis (the planes of a child are ordered, lower <= upper, so this is
  // min / max of the two distances without computing them per child).  An unused slot holds planes (255, 0): its far distance
  // lies before its near one on every axis, no ray enters it.
  const bool ux = w.inv.x >= 0.f, uy = w.inv.y >= 0.f, uz = w.inv.z >= 0.f;
  const uint32_t nx[2] = {ux ? h1.x : h2.z, ux ? h1.y : h2.w}, fx[2] = {ux ? h2.z : h1.x, ux ? h2.w : h1.y};
  const uint32_t ny[2] = {uy ? h1.z : h3.x, uy ? h1.w : h3.y}, fy[2] = {uy ? h3.x : h1.z, uy ? h3.y : h1.w};
  const uint32_t nz[2] = {uz ? h2.x : h3.z, uz ? h2.y : h3.w}, fz[2] = {uz ? h3.z : h2.x, uz ? h3.w : h2.y};
  const int n_inner = (int)((exps >> 24) & 15u);
  const int inner_base = (int)h0.w;   // child k's record is unit inner_base + k (k >= n_inner: a leaf record)
  uint32_t key[8];
  uint32_t entered = 0u;                       // !SORTED: bit k = child k entered
  constexpr uint32_t kNoChild = 0xfffffff0u;   // sorts behind every child entered; its low nibble says "no child"
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int sh = 8 * (k & 3);
    const float a = fmaxf(fmaxf(fmaf((float)((nx[k >> 2] >> sh) & 0xffu), ax, bx), fmaf((float)((ny[k >> 2] >> sh) & 0xffu), ay, by)),
                          fmaxf(fmaf((float)((nz[k >> 2] >> sh) & 0xffu), az, bz), t_min));
    const float b = fminf(fminf(fmaf((float)((fx[k >> 2] >> sh) & 0xffu), ax, bx), fmaf((float)((fy[k >> 2] >> sh) & 0xffu), ay, by)),
                          fminf(fmaf((float)((fz[k >> 2] >> sh) & 0xffu), az, bz), t_best));
    const bool ok = a <= b;
    // sortable key: the entry distance (a positive float, whose bit pattern orders as an unsigned integer) with the nibble
    // 8 | slot in its four lowest bits
    if constexpr (SORTED) key[k] = ok ? ((__float_as_uint(a) & ~15u) | (8u | (uint32_t)k)) : kNoChild;
    else entered |= ok ? (1u << k) : 0u;
  }
  // An unused slot (planes 255 and 0) is entered by no ray as long as 255 units make a difference to the slab distance.  They stop
  // doing so when the node is so small and so far from the ray's origin that 255 * ax is lost in the rounding of bx (on every axis,
  // for the slot to be entered; testing one is enough to rule it out): only then the slots beyond the node's children are masked
  // one by one.
  if (255.0f * fabsf(ax) <= fabsf(bx) * 2.4e-7f) {
    const int last = (int)(exps >> 28);   // n_child - 1
#pragma unroll
    for (int k = 2; k < 8; ++k)           // a node has at least two children
      if (k > last) key[k] = kNoChild;
    entered &= (2u << last) - 1u;
  }
  if constexpr (!SORTED) {
    // slot order: the lowest slot entered is next, the others are parked as a bit mask
    const uint32_t rest = entered & (entered - 1u);
    if (rest) {
      const int e = 2 * w.sp;
      st.put(e, inner_base); st.put(e + 1, (int)(rest | ((uint32_t)n_inner << 28)));
      ++w.sp;
    }
    if (entered) { const int k = __builtin_ctz(entered); w.node = k < n_inner ? inner_base + k : ~(inner_base + k); }
    else w.node = walk8_pop<false>(w, st);
    return;
  }
  // sorting network on the keys, ascending (19 compare-exchanges of one min and one max each)
#define RTX_CX(i, j) { const uint32_t lo_ = umin_(key[i], key[j]); key[j] = umax_(key[i], key[j]); key[i] = lo_; }
  RTX_CX(0, 1) RTX_CX(2, 3) RTX_CX(4, 5) RTX_CX(6, 7)
  RTX_CX(0, 2) RTX_CX(1, 3) RTX_CX(4, 6) RTX_CX(5, 7)
  RTX_CX(1, 2) RTX_CX(5, 6) RTX_CX(0, 4) RTX_CX(3, 7)
  RTX_CX(1, 5) RTX_CX(2, 6)
  RTX_CX(1, 4) RTX_CX(3, 6)
  RTX_CX(2, 4) RTX_CX(3, 5)
  RTX_CX(3, 4)
#undef RTX_CX
  if (key[1] & 8u) {   // park the siblings of the nearest child as one entry (a key that names no child leaves a nibble of 0)
    // eight funnel shifts by one nibble: the low nibbles of key[1..7], then n_inner, pushed in at the top; key[1]'s ends at the bottom
    uint32_t slots = 0u;
#pragma unroll
    for (int i = 1; i < 8; ++i) slots = __builtin_amdgcn_alignbit(key[i], slots, 4);
    slots = __builtin_amdgcn_alignbit((uint32_t)n_inner, slots, 4);
    const int e = 2 * w.sp;
    st.put(e, inner_base); st.put(e + 1, (int)slots);
    ++w.sp;
  }
  if (key[0] & 8u) { const int k = (int)(key[0] & 7u); w.node = k < n_inner ? inner_base + k : ~(inner_base + k); }
  else w.node = walk8_pop<true>(w, st);
}

// Möller-Trumbore on a leaf record's triangle (ref: engine/primitives.hpp:78-110): v = the first vertex and the two edges from it,
// v1 - v0 and v2 - v0, subtracted when the record was written (the same single-precision differences the test would take); NaN
// entries fail the first comparison.
RTX_DEV bool leaf_tri_t(const float* v, const Ray<float>& r, float t_min, float t_max, float& t, float& u, float& vv) {
  const V3<float> v0 = {v[0], v[1], v[2]};
  const V3<float> e1 = {v[3], v[4], v[5]}, e2 = {v[6], v[7], v[8]};
  const V3<float> h = cross(r.d, e2);
  const float a = dot(e1, h);
  const float f = 1.0f / a;
  const V3<float> s = r.o - v0;
  u = f * dot(s, h);
  const V3<float> q = cross(s, e1);
  vv = f * dot(r.d, q);
  t = f * dot(e2, q);
  return fabsf(a) >= 1e-8f && u >= 0.f && u <= 1.f && vv >= 0.f && u + vv <= 1.f && t >= t_min && t <= t_max;
}

// Exact tie in t between primitive `p` and the current winner `best` (primitive numbers): does p win?  The later one in the
// reference's leaf order does (Scene::hit accepts t == closest_so_far, ref: engine/scene.hpp:602-631).  While that order is
// not on the device yet (the host is still building the reference-order tree) the higher primitive number wins and the tie is
// noted in the walk's state: the caller lists the work item the ray belongs to, and the host renders those items again once the
// order is known (about one closest-hit query in ten million ties on a soup of small triangles).
RTX_DEV bool tie_goes_to(const DScene<float>& sc, int p, int best, Walk8& w) {
  if (sc.rank) return sc.rank[p] > sc.rank[best];
  w.tied = 1;
  return p > best;
}
// Notes work item `item` as one in which such a tie occurred.
RTX_DEV void note_tied_item(const DRenderParams& P, DCounters* counters, uint32_t item) {
  const unsigned long long k = atomicAdd(&counters->ties, 1ull);
  if (P.tie_list && k < (unsigned long long)P.tie_cap) P.tie_list[k] = item;
}

// Tests the primitive of the leaf record with code `leaf` (a ~unit) against the ray; updates the candidate and the tie state.
RTX_DEV void walk8_leaf_test(const DScene<float>& sc, const Ray<float>& r, float t_min, Cand<float>& c, Walk8& w, int leaf) {
  const uint4* rec = reinterpret_cast<const uint4*>(sc.nodes8 + (~leaf));
  const uint4 q0 = rec[0], q1 = rec[1];
  const uint2 q2 = *reinterpret_cast<const uint2*>(rec + 2);
  const float v[9] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w), __uint_as_float(q1.x),
                      __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x)};
  const int ref = (int)q2.y;
  float t, u, vv;
  // (the range check on ref costs two instructions and keeps a record that is no leaf record from ever being taken for a triangle)
  bool ok = leaf_tri_t(v, r, t_min, c.t, t, u, vv) && (uint32_t)(ref - sc.n_spheres) < (uint32_t)sc.n_tris;
  if (ok && t == c.t && w.best_p >= 0) ok = tie_goes_to(sc, ref, w.best_p, w);   // exact tie: the later one in the reference's order wins
  if (ok) { c.t = t; c.type = PRIM_TRIANGLE; c.idx = ref - sc.n_spheres; c.uv_kind = 1; c.ua = u; c.ub = vv; w.best_p = ref; }
  const int p = -2 - ref;
  const bool sphere = (uint32_t)p < (uint32_t)sc.n_spheres, box = (uint32_t)(p - sc.n_spheres - sc.n_tris) < (uint32_t)sc.n_boxes;
  if (ref < -1 && (sphere || box)) {   // rare in a tree of triangles
    const int type = sphere ? PRIM_SPHERE : PRIM_BOX;
    const int idx = sphere ? p : p - sc.n_spheres - sc.n_tris;
    Cand<float> trial = c;
    test_bounded(sc, type, idx, r, t_min, trial);
    const bool changed = trial.type == type && trial.idx == idx && (trial.t != c.t || c.type != type || c.idx != idx);
    if (changed && !(trial.t == c.t && w.best_p >= 0 && !tie_goes_to(sc, p, w.best_p, w))) { c = trial; w.best_p = p; }
  }
}

template <bool ANY> RTX_DEV void walk8_leaf(const DScene<float>& sc, const Ray<float>& r, float t_min, Cand<float>& c, Walk8& w, LaneStack st) {
  walk8_leaf_test(sc, r, t_min, c, w, w.node);
  if (ANY && c.type >= 0) { w.node = kWalkDone; return; }
  w.node = walk8_pop<!ANY>(w, st);
}

template <bool ANY> RTX_DEV bool walk8(const DScene<float>& sc, const Ray<float>& r, float t_min, Cand<float>& c, LaneStack st) {
  Walk8 w;
  walk8_begin(w, r);
#ifdef RTX_WALK_STATS
  int walk_steps = 0;   // diagnostic build: steps per ray come back in hit.u
#endif
  while (w.node != kWalkDone) {
#ifdef RTX_WALK_STATS
    ++walk_steps;
#endif
    if (w.node >= 0) walk8_inner<!ANY>(sc, r, t_min, c.t, w, st);
    else { walk8_leaf<ANY>(sc, r, t_min, c, w, st); if (ANY && c.type >= 0) return true; }
  }
#ifdef RTX_WALK_STATS
  c.ua = (float)walk_steps; c.uv_kind = 1;
#endif
  c.tied = w.tied;
  return c.type >= 0;
}

template <class R, bool ANY> RTX_DEV bool walk_tree(const DScene<R>& sc, const Ray<R>& r, R t_min, Cand<R>& c, LaneStack st) {
  if constexpr (std::is_same<R, float>::value) {
    if (sc.n_prims > kLinearMaxPrims) return walk8<ANY>(sc, r, t_min, c, st);
  }
  return walk_bounded<R, ANY>(sc, r, t_min, c, st);
}

template <class R, bool WANT_UV> RTX_DEV bool closest_hit(const DScene<R>& sc, const Ray<R>& r, R t_min, R t_max, Hit<R>& h, LaneStack st) {
  Cand<R> c;
  cand_init(c, t_max);
  walk_tree<R, false>(sc, r, t_min, c, st);
  test_unbounded(sc, r, t_min, c);
  if (c.type < 0) return false;
  finalize_hit<R, WANT_UV>(sc, r, c, h);
  h.tied = c.tied;
  return true;
}

// Visibility: "does Scene::hit return true on [t_min, t_max]" (ref: engine/renderer.cpp:679).
template <class R> RTX_DEV bool occluded_unbounded(const DScene<R>& sc, const Ray<R>& r, R t_min, R t_max) {
  R t, u, v;
  bool any = false;
  for (int i = 0; i < sc.n_planes; ++i) any |= plane_t(sc.planes[i], r, t_min, t_max, t);
  for (int i = 0; i < sc.n_quads; ++i) any |= quad_t(sc.quads[i], r, t_min, t_max, t, u, v);
  for (int i = 0; i < sc.n_disks; ++i) any |= disk_t(sc.disks[i], r, t_min, t_max, t, u, v);
  return any;
}
template <class R> RTX_DEV bool occluded(const DScene<R>& sc, const Ray<R>& r, R t_min, R t_max, LaneStack st) {
  Cand<R> c;
  cand_init(c, t_max);
  if (walk_tree<R, true>(sc, r, t_min, c, st)) return true;
  return occluded_unbounded(sc, r, t_min, t_max);
}

// ------------------------------------------------------------------------------------------
// counter-based random numbers: Philox-4x32-10, key = (pixel, seed), counter = (sample, block).
// Block 0 holds the pixel jitter (top 24 bits of words 0 and 1).  Path vertex b draws ONE block,
// 1+b, cut into 16-bit fields:  word0 = light u | light v,  word1 = scatter u1 | scatter u2,
// word2 = light select | roulette,  word3 = fresnel | spare.  A field k maps to (k + 0.5) / 65536,
// exactly representable in float and double, so both precisions see equal numbers.  (32-bit
// integer multiplies are quarter-rate on CDNA: one block per vertex instead of two is worth
// about a tenth of the whole kernel.)
// ------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
RTX_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1) {
  uint32_t c2 = 0, c3 = 0;
#ifndef RTX_PHILOX_ROUNDS
#define RTX_PHILOX_ROUNDS 10
#endif
#pragma unroll
  for (int i = 0; i < RTX_PHILOX_ROUNDS; ++i) {
    uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}
// A second, cheaper block generator for the single-precision production path: the pcg4d hash (Jarzynski & Olano,
// "Hash Functions for GPU Rendering", 2020) of (pixel, sample, block, seed), run for three mix rounds with an xor-shift
// after each (both halves of every word are used).  16 low multiplies instead of Philox's 20 low + 20 high ones; on the
// VALU-bound Cornell kernel the generator was ~15 % of the time.  The published two rounds are not enough here: over 64
// seeds the image error (relMSE against the reference's converged films) had twice Philox's spread and outliers; with
// three rounds mean and spread equal Philox's on all three test scenes (scripts/rng_quality.py).
#ifndef RTX_PCG_ROUNDS
#define RTX_PCG_ROUNDS 3
#endif
RTX_DEV U4 pcg4d(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
#pragma unroll
  for (int i = 0; i < RTX_PCG_ROUNDS; ++i) {
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
  }
  return {x, y, z, w};
}
enum { RNG_PHILOX = 0, RNG_PCG4D = 1 };
struct PathKey;
RTX_DEV U4 rng_block(uint32_t rng, uint32_t pixel, uint32_t seed, uint32_t sample, uint32_t block) {
  if (rng == RNG_PCG4D) return pcg4d(pixel, sample, block, seed);
  return philox4x32_10(sample, block, pixel, seed);
}

template <class R> RTX_DEV R u24(uint32_t x) { return R(x >> 8) * R(1.0 / 16777216.0); }
template <class R> RTX_DEV R u16hi(uint32_t x) { return (R(x >> 16) + R(0.5)) * R(1.0 / 65536.0); }
template <class R> RTX_DEV R u16lo(uint32_t x) { return (R(x & 0xffffu) + R(0.5)) * R(1.0 / 65536.0); }

// Uniform direction on the unit sphere from two uniforms (same law as the reference's
// normalised rejection sample, ref: engine/renderer.cpp:45-56).
template <class R> RTX_DEV V3<R> unit_vector(R u1, R u2) {
  R z = R(1) - R(2) * u1;
  R rr = sqrt_(max_(R(0), R(1) - z * z));
  R s, c;
  sincos_turns(u2, &s, &c);
  return {rr * c, rr * s, z};
}

// ------------------------------------------------------------------------------------------
// environment map (ref: engine/texture.hpp:487-549 lookup, :555-627 sampling, :634-687 pdf)
// ------------------------------------------------------------------------------------------
template <class R> RTX_DEV V3<R> env_rotate(const DEnv<R>& e, V3<R> d, R sign) {
  if (e.rotation != R(0)) {
    R s, c;
    sincos_(sign * e.rotation, &s, &c);
    R nx = d.x * c - d.z * s, nz = d.x * s + d.z * c;
    d.x = nx; d.z = nz;
  }
  return d;
}
RTX_DEV float floor_(float x) { return floorf(x); }
RTX_DEV double floor_(double x) { return floor(x); }
RTX_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
template <class R> RTX_DEV R clamp_(R v, R lo, R hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <class R> RTX_DEV V3<R> env_lookup(const DEnv<R>& e, V3<R> direction) {
  if (!e.valid) return {R(0), R(0), R(0)};
  V3<R> d = env_rotate(e, normalize(direction), R(1));
  R phi = atan2(d.z, d.x);
  R theta = asin(clamp_(d.y, R(-1), R(1)));
  R u = R(0.5) + phi / (R(2) * Consts<R>::pi);
  R v = R(0.5) + theta / Consts<R>::pi;
  R fx = u * R(e.w) - R(0.5);
  R fy = (R(1) - v) * R(e.h) - R(0.5);
  int x0 = (int)floor_(fx), y0 = (int)floor_(fy);
  int x1 = x0 + 1, y1 = y0 + 1;
  R tx = fx - R(x0), ty = fy - R(y0);
  x0 = ((x0 % e.w) + e.w) % e.w; x1 = ((x1 % e.w) + e.w) % e.w;
  y0 = clampi(y0, 0, e.h - 1); y1 = clampi(y1, 0, e.h - 1);
  auto px = [&](int x, int y) -> V3<R> {
    const float* t = e.texel + ((size_t)y * e.w + x) * 3;
    return {R(t[0]) * e.intensity, R(t[1]) * e.intensity, R(t[2]) * e.intensity};
  };
  V3<R> c0 = lerp3(px(x0, y0), px(x1, y0), tx);
  V3<R> c1 = lerp3(px(x0, y1), px(x1, y1), tx);
  return lerp3(c0, c1, ty);
}

template <class R> RTX_DEV R env_pdf(const DEnv<R>& e, V3<R> direction) {
  if (!e.importance) return R(0);
  V3<R> d = env_rotate(e, normalize(direction), R(1));
  R phi = atan2(d.z, d.x);
  R theta = acos(clamp_(d.y, R(-1), R(1)));
  R u = R(0.5) + phi / (R(2) * Consts<R>::pi);
  R v = theta / Consts<R>::pi;
  int x = clampi((int)(u * R(e.w)), 0, e.w - 1);
  int y = clampi((int)(v * R(e.h)), 0, e.h - 1);
  R mp = (e.marginal[y + 1] - e.marginal[y]) * R(e.h);
  size_t off = (size_t)y * (e.w + 1);
  R cp = (e.conditional[off + x + 1] - e.conditional[off + x]) * R(e.w);
  R st = sin(theta);
  if (st < R(1e-10)) return R(0);
  return mp * cp / (R(2) * Consts<R>::pi * Consts<R>::pi * st);
}

// index of the first element greater than x, minus one (std::upper_bound - 1), clamped
template <class R> RTX_DEV int cdf_bin(const R* cdf, int n_plus_1, R x) {
  int lo = 0, hi = n_plus_1;  // first index with cdf[i] > x
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cdf[mid] > x) hi = mid; else lo = mid + 1;
  }
  return clampi(lo - 1, 0, n_plus_1 - 2);
}

// The same index, the bisection started between two entries of the function's guide (DEnv): x lies in [k / kEnvGuide,
// (k + 1) / kEnvGuide), so the first element greater than x is neither before the first one greater than k / kEnvGuide nor after
// the first one greater than (k + 1) / kEnvGuide.
template <class R> RTX_DEV int cdf_bin_guided(const R* cdf, int n_plus_1, R x, const int32_t* guide) {
  const int k = clampi((int)(x * R(kEnvGuide)), 0, kEnvGuide - 1);
  int lo = guide[k], hi = guide[k + 1];
  if (!(x >= R(0) && x < R(1))) { lo = 0; hi = n_plus_1; }   // (numbers outside [0, 1) take the whole row, as cdf_bin does)
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cdf[mid] > x) hi = mid; else lo = mid + 1;
  }
  return clampi(lo - 1, 0, n_plus_1 - 2);
}

template <class R> RTX_DEV bool env_sample(const DEnv<R>& e, R u1, R u2, V3<R>& dir, V3<R>& Le, R& pdf) {
  if (!e.valid) return false;
  int y = cdf_bin_guided(e.marginal, e.h + 1, u1, e.marginal_guide);
  R mb = e.marginal[y], ma = e.marginal[y + 1];
  R row_t = (ma > mb) ? (u1 - mb) / (ma - mb) : R(0);
  const R* cond = e.conditional + (size_t)y * (e.w + 1);
  int x = cdf_bin_guided(cond, e.w + 1, u2, e.conditional_guide + (size_t)y * (kEnvGuide + 1));
  R lo = cond[x], hi = cond[x + 1];
  R col_t = (hi > lo) ? (u2 - lo) / (hi - lo) : R(0);
  R u = (R(x) + col_t + R(0.5)) / R(e.w);   // half-texel shift relative to env_pdf's binning, as in the reference
  R v = (R(y) + row_t + R(0.5)) / R(e.h);
  R phi = (u - R(0.5)) * R(2) * Consts<R>::pi;
  R theta = v * Consts<R>::pi;
  R st, ct, sp, cp;
  sincos_(theta, &st, &ct);
  sincos_(phi, &sp, &cp);
  V3<R> d = {st * cp, ct, st * sp};
  d = env_rotate(e, d, R(-1));
  dir = d;
  Le = env_lookup(e, d);
  pdf = env_pdf(e, d);
  return pdf > R(0);
}

// ------------------------------------------------------------------------------------------
// materials (ref: engine/material.hpp:71-76, :115-117, :308-312; engine/texture.hpp:268-310)
// ------------------------------------------------------------------------------------------
template <class R> RTX_DEV bool is_emissive(const DMaterial<R>& m) {
  return m.emission[0] > R(0) || m.emission[1] > R(0) || m.emission[2] > R(0);
}

template <class R> RTX_DEV V3<R> albedo_at(const DScene<R>& sc, const DMaterial<R>& m, const Hit<R>& h) {
  if (!m.has_texture) return mk3(m.albedo);
  if (m.tex_type == TEX_CHECKER) {
    R inv = R(1) / m.scale;
    int x = (int)floor_(h.p.x * inv), y = (int)floor_(h.p.y * inv), z = (int)floor_(h.p.z * inv);
    return ((x + y + z) % 2) == 0 ? mk3(m.c1) : mk3(m.c2);
  }
  if (m.tex_type == TEX_IMAGE) {
    if (m.tex_image < 0 || m.tex_image >= sc.n_images) return {R(1), R(0), R(1)};
    DImage im = sc.images[m.tex_image];
    if (im.w == 0 || im.h == 0) return {R(1), R(0), R(1)};
    R u = clamp_(h.u, R(0), R(1));
    R v = R(1) - clamp_(h.v, R(0), R(1));
    int i = (int)(u * R(im.w)), j = (int)(v * R(im.h));
    i = i < im.w - 1 ? i : im.w - 1; j = j < im.h - 1 ? j : im.h - 1;
    const uint8_t* t = im.rgb + ((size_t)j * im.w + i) * 3;
    return {R(t[0]) / R(255), R(t[1]) / R(255), R(t[2]) / R(255)};
  }
  return mk3(m.c1);
}

// hit_record::tangent of the winning primitive, as its hit routine writes it (ref: engine/sphere.hpp:39-50,102;
// engine/primitives.hpp:55-57 plane, :119 triangle, :197-204 box, :310 quad light, :425 disk light).
template <class R> RTX_DEV V3<R> hit_tangent(const DScene<R>& sc, const Hit<R>& h) {
  switch (h.type) {
    case PRIM_SPHERE: {
      const DSphere<R>& s = sc.spheres[h.idx];
      const V3<R> on = (h.p - mk3(s.c)) * (R(1) / s.radius);
      if (abs_(on.y) > R(0.999)) return {R(1), R(0), R(0)};
      return normalize(V3<R>{-on.z, R(0), on.x});
    }
    case PRIM_PLANE: {
      const V3<R> n = mk3(sc.planes[h.idx].n);
      const V3<R> up = abs_(n.y) < R(0.999) ? V3<R>{R(0), R(1), R(0)} : V3<R>{R(1), R(0), R(0)};
      return normalize(cross(up, n));
    }
    case PRIM_TRIANGLE: {
      const DTri<R>& tr = sc.tris[h.idx];
      return normalize(mk3(tr.v1) - mk3(tr.v0));
    }
    case PRIM_BOX: return abs_(h.n.x) > R(0.5) ? V3<R>{R(0), R(0), R(1)} : V3<R>{R(1), R(0), R(0)};
    case PRIM_QUAD: return normalize(mk3(sc.quads[h.idx].eu));
    default: return mk3(sc.disks[h.idx].tan);
  }
}

// NormalMap::sample (ref: engine/texture.hpp:83-109) and apply_normal_map (ref: engine/renderer.cpp:72-94):
// replaces the shading normal of `h` by the tangent-space texel carried into the frame (tangent, N x T, N).
template <class R> RTX_DEV void apply_normal_map(const DScene<R>& sc, const DMaterial<R>& m, Hit<R>& h) {
  if (m.nmap_image < 0 || m.nmap_image >= sc.n_images) return;
  const DImage im = sc.images[m.nmap_image];
  const R u = clamp_(h.u, R(0), R(1));
  const R v = R(1) - clamp_(h.v, R(0), R(1));
  int i = (int)(u * R(im.w)), j = (int)(v * R(im.h));
  i = i < im.w - 1 ? i : im.w - 1; j = j < im.h - 1 ? j : im.h - 1;
  const uint8_t* t = im.rgb + ((size_t)j * im.w + i) * 3;
  R nx = (R(t[0]) / R(255)) * R(2) - R(1);
  R ny = (R(t[1]) / R(255)) * R(2) - R(1);
  R nz = (R(t[2]) / R(255)) * R(2) - R(1);
  if (m.nmap_strength < R(1)) {
    nx = nx * m.nmap_strength;
    ny = ny * m.nmap_strength;
    nz = nz * m.nmap_strength + (R(1) - m.nmap_strength);
  }
  const V3<R> tn = normalize(V3<R>{nx, ny, nz});
  const V3<R> N = h.n;
  V3<R> T = hit_tangent(sc, h);
  T = normalize(T - N * dot(N, T));
  const V3<R> B = cross(N, T);
  h.n = normalize(V3<R>{T.x * tn.x + B.x * tn.y + N.x * tn.z, T.y * tn.x + B.y * tn.y + N.y * tn.z, T.z * tn.x + B.z * tn.y + N.z * tn.z});
}

template <class R> RTX_DEV R schlick(R cosine, R ref_idx) {
  R r0 = (R(1) - ref_idx) / (R(1) + ref_idx);
  r0 = r0 * r0;
  R x = R(1) - cosine;
  R x2 = x * x;
  return r0 + (R(1) - r0) * (x2 * x2 * x);
}

template <class R> RTX_DEV R power_heuristic(R a, R b) { R a2 = a * a, b2 = b * b; return a2 / (a2 + b2); }  // ref: engine/renderer.cpp:574-578

// ------------------------------------------------------------------------------------------
// light sampling (ref: engine/scene.hpp:238-388, :407-463, :470-488, :686-700)
// ------------------------------------------------------------------------------------------
template <class R> struct LightPick { V3<R> pos, n, Le; R pdf, dist; bool valid; };

template <class R> RTX_DEV LightPick<R> sample_light(const DScene<R>& sc, const DMaterial<R>* mats, V3<R> from, R rv, R lu, R lv) {
  LightPick<R> out;
  out.valid = false; out.pdf = R(0); out.dist = R(0);
  out.pos = out.n = out.Le = {R(0), R(0), R(0)};
  const int np = sc.n_points, ne = sc.n_emissive, nq = sc.n_quads, nd = sc.n_disks;
  const int total = np + ne + nq + nd;
  const bool has_env = sc.env.importance != 0;
  if (total == 0 && !has_env) return out;
  bool pick_env = false;
  if (has_env && total > 0) {
    pick_env = rv < R(0.5);
    rv = pick_env ? (rv * R(2)) : ((rv - R(0.5)) * R(2));
  } else if (has_env) {
    pick_env = true;
  }
  if (pick_env) {
    V3<R> d, Le; R pdf;
    if (env_sample(sc.env, lu, lv, d, Le, pdf)) {
      out.pos = from + d * R(1e6);
      out.n = -d; out.Le = Le; out.pdf = pdf; out.dist = R(1e6); out.valid = true;
      if (total > 0) out.pdf *= R(0.5);
    }
    return out;
  }
  int idx = (int)(rv * R(total));
  if (idx >= total) idx = total - 1;
  if (idx < np) {
    const DPoint<R>& pl = sc.points[idx];
    out.pos = mk3(pl.pos); out.n = {R(0), R(-1), R(0)}; out.Le = mk3(pl.I);
    out.pdf = R(1);
  } else if (idx < np + ne) {
    const DSphere<R>& sp = sc.spheres[sc.emissive[idx - np]];
    V3<R> local = unit_vector(lu, lv);  // z = 1-2u, phi = 2 pi v: scene.hpp:254-258
    out.pos = mk3(sp.c) + local * sp.radius;
    out.n = local;
    out.Le = mk3(mats[sp.mat].emission);
    R area = R(4) * Consts<R>::pi * sp.radius * sp.radius;
    out.pdf = R(1) / area;
  } else if (idx < np + ne + nq) {
    const DQuad<R>& q = sc.quads[idx - np - ne];
    out.pos = mk3(q.corner) + (mk3(q.eu) * lu + mk3(q.ev) * lv);
    out.n = mk3(q.n); out.Le = mk3(q.Le);
    out.pdf = R(1) / q.area;
  } else {
    const DDisk<R>& dl = sc.disks[idx - np - ne - nq];   // ref: engine/primitives.hpp:350-384
    R sx = R(2) * lu - R(1), sy = R(2) * lv - R(1);
    V3<R> c = mk3(dl.c);
    if (sx == R(0) && sy == R(0)) {
      out.pos = c;
    } else {
      R rs, theta;
      if (abs_(sx) > abs_(sy)) { rs = sx; theta = (Consts<R>::pi / R(4)) * (sy / sx); }
      else { rs = sy; theta = (Consts<R>::pi / R(2)) - (Consts<R>::pi / R(4)) * (sx / sy); }
      R st, ct;
      sincos_(theta, &st, &ct);
      R ar = dl.radius * rs;
      out.pos = c + (mk3(dl.tan) * (ar * ct) + mk3(dl.bit) * (ar * st));
    }
    out.n = mk3(dl.n); out.Le = mk3(dl.Le);
    out.pdf = R(1) / dl.area;
  }
  out.dist = len(out.pos - from);
  out.valid = true;
  out.pdf /= R(total);
  if (has_env) out.pdf *= R(0.5);
  return out;
}

template <class R> RTX_DEV R light_pdf(const DScene<R>& sc, V3<R> lp) {
  const int total = sc.n_points + sc.n_emissive + sc.n_quads + sc.n_disks;
  const bool has_env = sc.env.importance != 0;
  if (total == 0 && !has_env) return R(0);
  R sel = R(1) / R(total);
  if (has_env) sel *= R(0.5);
  for (int i = 0; i < sc.n_emissive; ++i) {
    const DSphere<R>& sp = sc.spheres[sc.emissive[i]];
    R d2 = len2(lp - mk3(sp.c));
    R r2 = sp.radius * sp.radius;
    if (abs_(d2 - r2) < r2 * R(0.01)) { R area = R(4) * Consts<R>::pi * r2; return sel / area; }
  }
  for (int i = 0; i < sc.n_quads; ++i) {
    const DQuad<R>& q = sc.quads[i];
    V3<R> tp = lp - mk3(q.corner);
    R dp = abs_(dot(tp, mk3(q.n)));
    if (dp < R(0.01)) {
      R up = dot(tp, mk3(q.eu)) / len2(mk3(q.eu));
      R vp = dot(tp, mk3(q.ev)) / len2(mk3(q.ev));
      if (up >= R(-0.01) && up <= R(1.01) && vp >= R(-0.01) && vp <= R(1.01)) return sel / q.area;
    }
  }
  for (int i = 0; i < sc.n_disks; ++i) {
    const DDisk<R>& dl = sc.disks[i];
    V3<R> tp = lp - mk3(dl.c);
    R dp = abs_(dot(tp, mk3(dl.n)));
    R rad2 = len2(tp) - dp * dp;
    if (dp < R(0.01) && rad2 <= dl.radius * dl.radius * R(1.01)) return sel / dl.area;
  }
  return R(0);
}

template <class R> RTX_DEV R env_light_pdf(const DScene<R>& sc, V3<R> dir) {
  if (!sc.env.importance) return R(0);
  R pdf = env_pdf(sc.env, dir);
  if (sc.n_points + sc.n_emissive + sc.n_quads + sc.n_disks > 0) pdf *= R(0.5);
  return pdf;
}

template <class R> RTX_DEV V3<R> area_light_emission(const DScene<R>& sc, int id) {
  if (id >= -sc.n_quads && id < 0) return mk3(sc.quads[-id - 1].Le);
  if (id <= -1000) { int k = -id - 1000; if (k < sc.n_disks) return mk3(sc.disks[k].Le); }
  return {R(0), R(0), R(0)};
}

template <class R> RTX_DEV V3<R> background(const DScene<R>& sc, const DRenderParams& P, const Ray<R>& r) {  // ref: engine/renderer.cpp:978-994
  if (sc.env.valid) return env_lookup(sc.env, r.d);
  if (P.use_gradient) {
    V3<R> ud = normalize(r.d);
    R t = R(0.5) * (ud.y + R(1));
    V3<R> bot = {R(P.bg_bottom[0]), R(P.bg_bottom[1]), R(P.bg_bottom[2])};
    V3<R> top = {R(P.bg_top[0]), R(P.bg_top[1]), R(P.bg_top[2])};
    return lerp3(bot, top, t);
  }
  return {R(P.bg[0]), R(P.bg[1]), R(P.bg[2])};
}

// ------------------------------------------------------------------------------------------
// "flat" closest hit for small scenes (single precision only)
// ------------------------------------------------------------------------------------------
// Scenes whose bounded primitives are a handful of spheres (every Cornell / environment
// configuration) need no tree and no per-lane memory traffic: every primitive comes from a
// wave-uniform (scalar) load and is tested by every lane with predicated, branch-free code.
// The winner's shading data travel in the candidate itself (4 selects per accepted test), so
// there is no divergent "which primitive type won" switch afterwards.  Test order and tie
// rule are the reference's: bounded primitives in leaf order, then planes, quads, disks; a
// later primitive with equal t wins.
RTX_DEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }

// ref: engine/sphere.hpp:71-92 — near root if it lies in range, else far root
RTX_DEV bool flat_sphere_t(const DSphere<float>& s, const Ray<float>& r, float a, float inv_a, float t_min, float t_max, float& t) {
  V3<float> oc = r.o - mk3(s.c);
  float hb = dot(oc, r.d);
  float cc = len2(oc) - s.radius * s.radius;
  float disc = hb * hb - a * cc;
  float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f));
  float t0 = (-hb - sq) * inv_a, t1 = (-hb + sq) * inv_a;
  bool ok0 = t0 >= t_min && t0 <= t_max, ok1 = t1 >= t_min && t1 <= t_max;
  t = ok0 ? t0 : t1;
  return disc >= 0.f && (ok0 || ok1);
}
// ref: engine/primitives.hpp:35-48
RTX_DEV bool flat_plane_t(const float* p, const float* n, const Ray<float>& r, float t_min, float t_max, float& t) {
  float denom = n[0] * r.d.x + n[1] * r.d.y + n[2] * r.d.z;
  float num = (p[0] - r.o.x) * n[0] + (p[1] - r.o.y) * n[1] + (p[2] - r.o.z) * n[2];
  t = num * rcp_(denom);
  return fabsf(denom) >= 1e-8f && t >= t_min && t <= t_max;
}
// The same test for a plane whose constant pd = n . p was computed at upload: t = (pd - n . o) / (n . d).
RTX_DEV bool flat_plane_pd(const float* n, float pd, const Ray<float>& r, float t_min, float t_max, float& t) {
  float denom = n[0] * r.d.x + n[1] * r.d.y + n[2] * r.d.z;
  float num = pd - (n[0] * r.o.x + n[1] * r.o.y + n[2] * r.o.z);
  t = num * rcp_(denom);
  return fabsf(denom) >= 1e-8f && t >= t_min && t <= t_max;
}
// ref: engine/primitives.hpp:274-300
RTX_DEV bool flat_quad_t(const DQuad<float>& q, const Ray<float>& r, float t_min, float t_max, float& t) {
  bool ok = flat_plane_t(q.corner, q.n, r, t_min, t_max, t);
  V3<float> d = at(r, t) - mk3(q.corner);
  V3<float> eu = mk3(q.eu), ev = mk3(q.ev);
  float up = dot(d, eu) * rcp_(len2(eu)), vp = dot(d, ev) * rcp_(len2(ev));
  return ok && up >= 0.f && up <= 1.f && vp >= 0.f && vp <= 1.f;
}
// ref: engine/primitives.hpp:396-417
RTX_DEV bool flat_disk_t(const DDisk<float>& dl, const Ray<float>& r, float t_min, float t_max, float& t) {
  bool ok = flat_plane_t(dl.c, dl.n, r, t_min, t_max, t);
  V3<float> d = at(r, t) - mk3(dl.c);
  return ok && len2(d) <= dl.radius * dl.radius;
}

// Closest hit of a flat scene: every primitive is tested without branches; only the distance and the
// position of the winner in the test order are carried along (flat_closest_tk), its shading record is fetched at the end
// (flat_hit_from).
RTX_DEV void flat_closest_tk(const DScene<float>& sc, const Ray<float>& r, float t_min, float t_max, float& best, int& k) {
  best = t_max;
  k = -1;
  int n = 0;
  const float a = len2(r.d), inv_a = rcp_(a);
  float t;
  for (int i = 0; i < sc.n_prims; ++i, ++n) {   // all spheres in flat scenes, in reference leaf order
    const bool ok = flat_sphere_t(sc.spheres[sc.prims[i].index], r, a, inv_a, t_min, best, t);
    best = ok ? t : best; k = ok ? n : k;
  }
  for (int i = 0; i < sc.n_planes; ++i, ++n) {
    const bool ok = flat_plane_pd(sc.planes[i].n, sc.planes[i].pd, r, t_min, best, t);
    best = ok ? t : best; k = ok ? n : k;
  }
  for (int i = 0; i < sc.n_quads; ++i, ++n) {
    const bool ok = flat_quad_t(sc.quads[i], r, t_min, best, t);
    best = ok ? t : best; k = ok ? n : k;
  }
  for (int i = 0; i < sc.n_disks; ++i, ++n) {
    const bool ok = flat_disk_t(sc.disks[i], r, t_min, best, t);
    best = ok ? t : best; k = ok ? n : k;
  }
}
RTX_DEV bool flat_hit_from(const DScene<float>& sc, const Ray<float>& r, float best, int k, Hit<float>& h) {
  if (k < 0) return false;
  const DFlatRec rec = sc.flat[k];
  h.t = best; h.type = rec.type; h.idx = rec.idx; h.mat = rec.mat; h.u = 0.f; h.v = 0.f; h.tied = 0;
  h.p = at(r, best);
  V3<float> abc = {rec.a, rec.b, rec.c};
  V3<float> on = rec.w != 0.f ? (h.p - abc) * rec.w : abc;
  face_normal(h, r, on);
  return true;
}
RTX_DEV bool flat_closest(const DScene<float>& sc, const Ray<float>& r, float t_min, float t_max, Hit<float>& h) {
  float best;
  int k;
  flat_closest_tk(sc, r, t_min, t_max, best, k);
  return flat_hit_from(sc, r, best, k, h);
}

// UNIT: the caller's direction has unit length (the light-sample rays of the path tracer), so |d|^2 is taken as 1.
template <bool UNIT = false> RTX_DEV bool flat_occluded(const DScene<float>& sc, const Ray<float>& r, float t_min, float t_max) {
  const float a = UNIT ? 1.f : len2(r.d), inv_a = UNIT ? 1.f : rcp_(a);
  float t;
  bool any = false;
  for (int i = 0; i < sc.n_prims; ++i) any |= flat_sphere_t(sc.spheres[sc.prims[i].index], r, a, inv_a, t_min, t_max, t);
  for (int i = 0; i < sc.n_planes; ++i) any |= flat_plane_pd(sc.planes[i].n, sc.planes[i].pd, r, t_min, t_max, t);
  for (int i = 0; i < sc.n_quads; ++i) any |= flat_quad_t(sc.quads[i], r, t_min, t_max, t);
  for (int i = 0; i < sc.n_disks; ++i) any |= flat_disk_t(sc.disks[i], r, t_min, t_max, t);
  return any;
}

// Is any primitive of material `mat` (>= 0: area lights carry negative ids) hit in [t_min, t_max] at all?  When the answer is no,
// the closest hit of that ray - if it has one - belongs to another material, and a caller that only asks "is the closest hit of
// material mat" (the re-projection of manifold NEE) need not make the query.  The tests are flat_closest_tk's, against the whole
// interval instead of the shrinking one, so a primitive that could win there is never missed here.  A primitive is tested only if
// some lane of the wave asks for its material (its record is wave-uniform): one sphere test where the closest-hit query makes
// eight.
RTX_DEV bool flat_hits_material(const DScene<float>& sc, const Ray<float>& r, float t_min, float t_max, int mat) {
  const float a = len2(r.d), inv_a = rcp_(a);
  float t;
  bool any = false;
  int n = 0;
  for (int i = 0; i < sc.n_prims; ++i, ++n) {
    const bool mine = sc.flat[n].mat == mat;
    if (__ballot(mine)) any |= mine && flat_sphere_t(sc.spheres[sc.prims[i].index], r, a, inv_a, t_min, t_max, t);
  }
  for (int i = 0; i < sc.n_planes; ++i, ++n) {
    const bool mine = sc.flat[n].mat == mat;
    if (__ballot(mine)) any |= mine && flat_plane_pd(sc.planes[i].n, sc.planes[i].pd, r, t_min, t_max, t);
  }
  return any;
}

// How a kernel instance intersects: MODE_FLAT (float, small sphere-only scenes, no image
// textures), MODE_TREE (anything: production BVH / linear list, exact operand order in double).
template <class R, int MODE> RTX_DEV bool trace_closest(const DScene<R>& sc, const Ray<R>& r, R t_min, R t_max, Hit<R>& h, LaneStack st) {
  if constexpr (MODE == MODE_FLAT) return flat_closest(sc, r, t_min, t_max, h);
  else if constexpr (MODE == MODE_TREE_UV) return closest_hit<R, true>(sc, r, t_min, t_max, h, st);
  else return closest_hit<R, false>(sc, r, t_min, t_max, h, st);
}
template <class R, int MODE, bool UNIT = false> RTX_DEV bool trace_occluded(const DScene<R>& sc, const Ray<R>& r, R t_min, R t_max, LaneStack st) {
  if constexpr (MODE == MODE_FLAT) return flat_occluded<UNIT>(sc, r, t_min, t_max);
  else return occluded(sc, r, t_min, t_max, st);
}

// ------------------------------------------------------------------------------------------
// the integrator, one path vertex at a time
// ------------------------------------------------------------------------------------------
// Renderer::ray_color_path (ref: engine/renderer.cpp:580-841, use_mnee = false) issues three
// Scene::hit calls per diffuse bounce: the bounce's own closest hit, the shadow ray, and a
// "look-ahead" closest hit of the ray the NEXT iteration traces again (:797-835).  Here the
// look-ahead and the next iteration share one trace: `pending` marks that the vertex about to
// be found still owes the previous vertex its MIS-weighted emission.
enum : uint32_t { F_SPEC = 1u, F_DIFFUSE = 2u, F_PENDING = 4u, F_ALIVE = 8u };

template <class R> struct PathState {
  Ray<R> ray;
  V3<R> thr, acc;
  R bsdf_pdf;
  uint32_t bounce_flags;   // bounce << 4 | F_*
};

struct PathKey { uint32_t pixel, seed, sample, rng; };   // rng: RNG_PHILOX or RNG_PCG4D

template <class R> RTX_DEV void path_begin(PathState<R>& ps, const DCamera<R>& cam, const DRenderParams& P, int x, int y, PathKey key) {
  U4 j = rng_block(key.rng, key.pixel, key.seed, key.sample, 0u);
  R u = (R(x) + u24<R>(j.x)) / R(P.width);     // ref: engine/renderer.cpp:385-386
  R v = (R(y) + u24<R>(j.y)) / R(P.height);
  V3<R> o = mk3(cam.origin);                    // ref: engine/scene.hpp:99-108
  ps.ray.o = o;
  ps.ray.d = ((mk3(cam.llc) + mk3(cam.horizontal) * u) + mk3(cam.vertical) * v) - o;
  ps.thr = {R(1), R(1), R(1)};
  ps.acc = {R(0), R(0), R(0)};
  ps.bsdf_pdf = R(0);
  ps.bounce_flags = F_ALIVE;
}

#ifdef RTX_PHASE_STATS
// Diagnostic build only (never shipped).  Slots 0..13: (wave-trips, lanes) per STEP_* bit of the fused kernel's loop; from 16 on:
// RTX_SITE(k) counts, wherever it stands, the times a wave passed and the lanes that were active.
static __device__ unsigned long long g_phase_stats[64];
#define RTX_SITE(k)                                                                                                                       \
  {                                                                                                                                       \
    const unsigned long long m_ = __ballot(1);                                                                                            \
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) { atomicAdd(&g_phase_stats[2 * (k)], 1ull); atomicAdd(&g_phase_stats[2 * (k) + 1], (unsigned long long)__popcll(m_)); } \
  }
#else
#define RTX_SITE(k)
#endif
}  // namespace rtx
#include "pt_mnee.hpp"
namespace rtx {

// What one vertex did, for the statistics: bit 0 = a shadow ray was cast, bit 1 = the vertex was shaded.
enum : uint32_t { STEP_SHADOW = 1u, STEP_SHADED = 2u, STEP_TIED = 4u };   // TIED: the vertex's closest-hit query decided a tie without the reference's order
// Diagnostic build (-DRTX_PHASE_STATS, never shipped): which branches of the vertex a lane took, so that the fused kernel can count
// how often a WAVE pays for each of them (scripts/phase_table.py weights the static instruction counts with these frequencies).
enum : uint32_t { STEP_X_SETTLE = 8u, STEP_X_MISS = 16u, STEP_X_AREA = 32u, STEP_X_LAMBERT = 64u, STEP_X_METAL = 128u, STEP_X_GLASS = 256u, STEP_X_ROULETTE = 512u,
                  STEP_X_LIGHT_SAMPLE = 1024u, STEP_X_DEPTH = 2048u };
#ifdef RTX_PHASE_STATS
#define RTX_PHASE(bit) (did |= (bit))
#else
#define RTX_PHASE(bit) ((void)0)
#endif

// Shade the vertex a trace of ps.ray found (`hit`, `rec`): settle the previous vertex's pending
// MIS term, add emission, sample a light, scatter, Russian roulette.  The visibility query of
// the light sample is handed to `shadow(ray, t_max, contribution)`: the fused kernel traces it
// on the spot, the wavefront pipeline queues it.  `mats` is the material table (an LDS copy when
// the kernel staged one).
// MNEE = true adds manifold next event estimation at Lambertian vertices (ref: engine/renderer.cpp:716-732);
// it needs `shadow` to return whether the light sample was visible (the reference skips MNEE when an
// unoccluded sample is rejected, :691/:695) and `mnee(point, albedo, bounce)`, which evaluates it (pt_mnee.hpp) and adds to
// ps.acc.  It is called with ps.thr still the throughput ahead of this vertex.
template <class R, bool MNEE, bool NMAP, class ShadowFn, class MneeFn>
RTX_DEV uint32_t path_vertex_impl(PathState<R>& ps, const bool hit, Hit<R>& rec, const DScene<R>& sc, const DMaterial<R>* mats,
                                  const DRenderParams& P, PathKey key, ShadowFn&& shadow, MneeFn&& mnee) {
  const bool nee = P.use_nee != 0, mis = P.use_mis != 0;
  const bool env_is = sc.env.importance != 0;
  uint32_t did = 0;
  const int bounce = (int)(ps.bounce_flags >> 4);
  const bool was_spec = ps.bounce_flags & F_SPEC, was_diffuse = ps.bounce_flags & F_DIFFUSE, pending = ps.bounce_flags & F_PENDING;

  if (pending) {  // ref: engine/renderer.cpp:797-835, evaluated on the shared trace
    if (hit) {
      // A quad/disk light id is negative; the reference indexes its material array with it
      // (undefined behaviour).  Defined here as "not emissive": the arrival branch weights it.
      if (rec.mat >= 0) {
        const DMaterial<R>& nm = mats[rec.mat];
        if (is_emissive(nm)) {
          RTX_PHASE(STEP_X_SETTLE);
          R cl = abs_(dot(rec.n, -ps.ray.d));
          R lpa = light_pdf(sc, rec.p);
          if (lpa > R(0) && cl > R(0)) {
            R lps = lpa * rec.t * rec.t / cl;
            R w = power_heuristic(ps.bsdf_pdf, lps);
            ps.acc = ps.acc + (ps.thr * mk3(nm.emission)) * w;
          }
        }
      }
    } else if (env_is) {
      V3<R> ee = env_lookup(sc.env, ps.ray.d);
      R ep = env_light_pdf(sc, ps.ray.d);
      if (ep > R(0)) { R w = power_heuristic(ps.bsdf_pdf, ep); ps.acc = ps.acc + (ps.thr * ee) * w; }
    }
  }
  if (bounce >= P.max_depth) { RTX_PHASE(STEP_X_DEPTH); ps.bounce_flags = 0; return did; }

  if (!hit) {  // ref: engine/renderer.cpp:591-606
    RTX_PHASE(STEP_X_MISS);
    bool handled = nee && mis && was_diffuse && env_is;
    if (!handled) ps.acc = ps.acc + ps.thr * background(sc, P, ps.ray);
    ps.bounce_flags = 0;
    return did;
  }
  did |= STEP_SHADED;
  if (rec.mat < 0) {  // area light, ref: engine/renderer.cpp:609-630
    RTX_PHASE(STEP_X_AREA);
    V3<R> Le = area_light_emission(sc, rec.mat);
    if (bounce == 0 || was_spec || !nee) {
      ps.acc = ps.acc + ps.thr * Le;
    } else if (mis && was_diffuse && ps.bsdf_pdf > R(0)) {
      R cl = abs_(dot(rec.n, -normalize(ps.ray.d)));
      R lpa = light_pdf(sc, rec.p);
      if (lpa > R(0) && cl > R(0)) {
        R lps = lpa * rec.t * rec.t / cl;
        R w = power_heuristic(ps.bsdf_pdf, lps);
        ps.acc = ps.acc + (ps.thr * Le) * w;
      }
    }
    ps.bounce_flags = 0;
    return did;
  }
  const DMaterial<R>& m = mats[rec.mat];
  if constexpr (NMAP) apply_normal_map(sc, m, rec);
  V3<R> surf = albedo_at(sc, m, rec);
  if (bounce == 0 || was_spec || !nee) ps.acc = ps.acc + ps.thr * mk3(m.emission);

  V3<R> sdir = {R(0), R(0), R(0)};
  V3<R> atten = surf;
  uint32_t flags = F_ALIVE;
  R bsdf_pdf = R(0);
  const U4 rn = rng_block(key.rng, key.pixel, key.seed, key.sample, 1u + (uint32_t)bounce);

  if (m.type == MAT_LAMBERTIAN) {
    RTX_PHASE(STEP_X_LAMBERT);
    if (nee) {  // ref: engine/renderer.cpp:654-714
      RTX_PHASE(STEP_X_LIGHT_SAMPLE);
      bool skip_mnee = false;
      LightPick<R> ls = sample_light(sc, mats, rec.p, u16hi<R>(rn.z), u16hi<R>(rn.x), u16lo<R>(rn.x));
      if (ls.valid) {
        bool is_env = ls.dist > R(1e5);
        V3<R> ldir; R ldist;
        if (is_env) { ldir = normalize(ls.pos - rec.p); ldist = R(1e9); }
        else { V3<R> tl = ls.pos - rec.p; ldist = len(tl); ldir = tl * (R(1) / ldist); }
        R cs = dot(rec.n, ldir);
        if (cs > R(0)) {
          // The reference casts the shadow ray first and evaluates the pdfs only if it is
          // unoccluded; the contribution does not depend on that order.
          did |= STEP_SHADOW;
          R lps; bool ok = true;
          if (is_env) lps = ls.pdf;
          else {
            R cl = abs_(dot(ls.n, -ldir));
            ok = cl > R(0);
            lps = ok ? ls.pdf * ldist * ldist / cl : R(0);
          }
          V3<R> contrib = {R(0), R(0), R(0)};
          if (ok && lps > R(0)) {
            R bpl = cs / Consts<R>::pi;
            R w = mis ? power_heuristic(lps, bpl) : R(1);
            contrib = ((ls.Le * surf) * ps.thr) * (cs / (Consts<R>::pi * lps) * w);
          }
          Ray<R> sr = {rec.p, ldir};
          if constexpr (MNEE) {
            const bool visible = shadow(sr, is_env ? R(1e9) : (ldist - R(0.001)), contrib);
            if (visible && !(ok && lps > R(0))) skip_mnee = true;   // `goto skip_nee`, ref: renderer.cpp:691, :695
          } else {
            shadow(sr, is_env ? R(1e9) : (ldist - R(0.001)), contrib);
          }
        }
      }
      if constexpr (MNEE) {
        if (!skip_mnee) mnee(rec.p, surf, bounce);
      }
    }
    sdir = rec.n + unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y));   // ref: engine/renderer.cpp:737-743
    if (len2(sdir) < R(1e-8)) sdir = rec.n;
    sdir = normalize(sdir);
    bsdf_pdf = dot(rec.n, sdir) / Consts<R>::pi;
    flags |= F_DIFFUSE;
  } else if (m.type == MAT_METAL) {  // ref: engine/renderer.cpp:747-759 (below-surface directions continue)
    RTX_PHASE(STEP_X_METAL);
    bool spec = true;
    V3<R> refl = reflect(normalize(ps.ray.d), rec.n);
    if (m.fuzz > R(0)) { refl = refl + unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y)) * m.fuzz; spec = m.fuzz < R(0.1); }
    sdir = refl;
    if (spec) flags |= F_SPEC;
  } else {  // dielectric, ref: engine/renderer.cpp:761-778
    RTX_PHASE(STEP_X_GLASS);
    flags |= F_SPEC;
    atten = {R(1), R(1), R(1)};
    R ratio = rec.front ? (R(1) / m.ior) : m.ior;
    V3<R> ud = normalize(ps.ray.d);
    R ct = min_(dot(-ud, rec.n), R(1));
    R sn = sqrt_(R(1) - ct * ct);
    if (ratio * sn > R(1) || schlick(ct, ratio) > u16hi<R>(rn.w)) sdir = reflect(ud, rec.n);
    else sdir = refract(ud, rec.n, ratio);
  }
  ps.thr = ps.thr * atten;
  // Russian roulette, ref: engine/renderer.cpp:785-790.  Written without an early return: a
  // terminated path simply gets flags = 0 (its ray and throughput are never read again).  With
  // `return` here hipcc 7.2 -O2/-O3 emitted, in one instantiation, code that gave the SURVIVING
  // lanes the old ray.d.y/z (the copy meant for terminated lanes ran for every lane of the branch).
  bool killed = false;
#ifdef RTX_RR_EARLY_RETURN
  // The early-return form, compiled only into the toolchain-guard library (make guard; tests/test_gpu_toolchain_guard.py
  // renders with it and reports whether the compiler still gets it wrong).  Never part of librtx_hip.so.
  if (bounce > 3) {
    R p = max_(ps.thr.x, max_(ps.thr.y, ps.thr.z));
    if (p < R(0.01) || u16lo<R>(rn.z) > p) { ps.bounce_flags = 0; return did; }
    ps.thr = ps.thr * (R(1) / p);
  }
#else
  if (bounce > 3) {
    RTX_PHASE(STEP_X_ROULETTE);
    R p = max_(ps.thr.x, max_(ps.thr.y, ps.thr.z));
    killed = p < R(0.01) || u16lo<R>(rn.z) > p;
    R boost = killed ? R(1) : R(1) / p;
    ps.thr = ps.thr * boost;
  }
#endif
  ps.ray.o = rec.p;
  ps.ray.d = sdir;
  ps.bsdf_pdf = bsdf_pdf;
  const bool next_pending = nee && mis && !(flags & F_SPEC) && bsdf_pdf > R(0);
  if (next_pending) flags |= F_PENDING;
  if (bounce + 1 >= P.max_depth && !next_pending) flags = 0;
  if (killed) flags = 0;
  ps.bounce_flags = flags ? (((uint32_t)(bounce + 1) << 4) | flags) : 0u;
  return did;
}

template <class R, bool NMAP, class ShadowFn>
RTX_DEV uint32_t path_vertex(PathState<R>& ps, const bool hit, Hit<R>& rec, const DScene<R>& sc, const DMaterial<R>* mats,
                             const DRenderParams& P, PathKey key, ShadowFn&& shadow) {
  return path_vertex_impl<R, false, NMAP>(ps, hit, rec, sc, mats, P, key, shadow, [](V3<R>, V3<R>, int) {});
}

// Renderer::ray_color_whitted (ref: engine/renderer.cpp:483-569) unrolled into one vertex per call:
// the recursion  c * (direct + L(child))  becomes  acc += thr * c * direct;  thr *= c.  `occluded(ray, t_max)`
// answers Scene::is_shadowed (ref: engine/scene.hpp:554-564).  Returns STEP_* and adds the number of shadow
// queries to `n_shadow`.
template <class R, class OccFn>
RTX_DEV uint32_t whitted_vertex(PathState<R>& ps, const bool hit, Hit<R>& rec, const DScene<R>& sc, const DMaterial<R>* mats,
                                const DRenderParams& P, PathKey key, OccFn&& occluded, uint32_t& n_shadow) {
  const int bounce = (int)(ps.bounce_flags >> 4);
  if (bounce >= P.max_depth) { ps.bounce_flags = 0; return 0u; }   // depth <= 0: black
  // a quad / disk light has a negative material id; the reference indexes its material array with it
  // (undefined behaviour, renderer.cpp:493) — defined here as "no hit"
  if (!hit || rec.mat < 0) {
    ps.acc = ps.acc + ps.thr * background(sc, P, ps.ray);
    ps.bounce_flags = 0;
    return hit ? STEP_SHADED : 0u;
  }
  const DMaterial<R>& m = mats[rec.mat];
  apply_normal_map(sc, m, rec);   // ref: engine/renderer.cpp:496
  const V3<R> surf = albedo_at(sc, m, rec);
  const U4 rn = rng_block(key.rng, key.pixel, key.seed, key.sample, 1u + (uint32_t)bounce);
  V3<R> dir;
  if (m.type == MAT_LAMBERTIAN) {
    V3<R> direct = {R(0), R(0), R(0)};
    for (int i = 0; i < sc.n_points; ++i) {
      const V3<R> lpos = mk3(sc.points[i].pos);
      const V3<R> to_light = lpos - rec.p;   // unnormalised direction with t up to |to_light|, as the reference does
      const R distance = len(to_light);
      ++n_shadow;
      Ray<R> sr = {rec.p, to_light};
      if (!occluded(sr, distance)) {
        const R diffuse = max_(R(0), dot(rec.n, normalize(lpos - rec.p)));
        direct = direct + mk3(sc.points[i].I) * diffuse;
      }
    }
    ps.acc = ps.acc + ps.thr * (surf * direct);
    ps.thr = ps.thr * surf;
    dir = rec.n + unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y));
    if (len2(dir) < R(1e-8)) dir = rec.n;
  } else if (m.type == MAT_METAL) {
    dir = reflect(normalize(ps.ray.d), rec.n);
    if (!(dot(dir, rec.n) > R(0))) { ps.bounce_flags = 0; return STEP_SHADED; }
    ps.thr = ps.thr * surf;
  } else {
    R ratio = rec.front ? (R(1) / m.ior) : m.ior;
    V3<R> ud = normalize(ps.ray.d);
    R ct = min_(dot(-ud, rec.n), R(1));
    R sn = sqrt_(R(1) - ct * ct);
    if (ratio * sn > R(1) || schlick(ct, ratio) > u16hi<R>(rn.w)) dir = reflect(ud, rec.n);
    else dir = refract(ud, rec.n, ratio);
  }
  ps.ray.o = rec.p;
  ps.ray.d = dir;
  ps.bounce_flags = (bounce + 1 >= P.max_depth) ? 0u : ((((uint32_t)(bounce + 1)) << 4) | F_ALIVE);   // depth <= 0 returns black
  return STEP_SHADED;
}

}  // namespace rtx
#include "pt_photon.hpp"
namespace rtx {

// The closest-hit queries of manifold NEE (pt_mnee.hpp): Scene::hit through this kernel instance's structures, counted.
template <class R, int MODE> struct MneeTracer {
  const DScene<R>& sc;
  LaneStack st;
  uint32_t& queries;
  RTX_DEV bool operator()(const Ray<R>& r, R t_min, R t_max, Hit<R>& h) {
    ++queries;
    return trace_closest<R, MODE>(sc, r, t_min, t_max, h, st);
  }
  // A query whose only question is "does the closest hit have material mat": false = it certainly has not, and the query counts
  // as made; true = undecided, make it.  (Decided here for single-precision flat scenes only.)
  RTX_DEV bool may_hit_material(const Ray<R>& r, R t_min, R t_max, int mat) {
    if constexpr (MODE == MODE_FLAT && std::is_same<R, float>::value) {
      if (flat_hits_material(sc, r, t_min, t_max, mat)) return true;
      ++queries;
      return false;
    } else {
      return true;
    }
  }
};

// Fused form: trace, shade, and trace the shadow ray(s) in place.  `extra_traces` counts the closest-hit
// queries MNEE makes, `extra_shadow` the shadow queries of the Whitted integrator.
template <class R, int MODE, int INTEG = INTEG_PATH, bool NMAP = (MODE == MODE_TREE_UV)>
RTX_DEV uint32_t path_step(PathState<R>& ps, const DScene<R>& sc, const DMaterial<R>* mats, const DRenderParams& P, PathKey key, LaneStack st, uint32_t& extra_traces,
                           uint32_t& extra_shadow) {
  Hit<R> rec;
  const bool hit = trace_closest<R, MODE>(sc, ps.ray, R(0.001), R(1e9), rec, st);
  if constexpr (INTEG == INTEG_PATHPHOTON) {
    static_assert(INTEG != INTEG_PATHPHOTON, "PathPhoton vertices are stepped by the kernel itself: the density estimate is a wave-wide operation");
    return 0u;
  } else if constexpr (INTEG == INTEG_WHITTED) {
    return whitted_vertex(ps, hit, rec, sc, mats, P, key, [&](const Ray<R>& sr, R t_max) { return trace_occluded<R, MODE>(sc, sr, R(0.001), t_max, st); }, extra_shadow);
  } else {
    const uint32_t tied = (hit && rec.tied) ? STEP_TIED : 0u;
    return tied | path_vertex_impl<R, INTEG == INTEG_PATH_MNEE, NMAP>(
        ps, hit, rec, sc, mats, P, key,
        [&](const Ray<R>& sr, R t_max, V3<R> contrib) {
          const bool visible = !trace_occluded<R, MODE, true>(sc, sr, R(0.001), t_max, st);   // light-sample rays are normalised
          if (visible) ps.acc = ps.acc + contrib;
          return visible;
        },
        [&](V3<R> hp, V3<R> surf, int bounce) {   // MNEE::evaluate on the spot (ref: engine/renderer.cpp:716-732)
          MneeTracer<R, MODE> trace = {sc, st, extra_traces};
          V3<R> mc;
          if (mnee_evaluate(sc, mats, hp, key, bounce, mc, trace)) {
            V3<R> md = ((mc * surf) * ps.thr) * (R(1) / R(3.14159265358979323846));
            ps.acc = ps.acc + md;
          }
        });
  }
}

}  // namespace rtx
