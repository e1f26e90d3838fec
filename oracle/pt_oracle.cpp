// TEST INFRASTRUCTURE ONLY — see pt_oracle.h.
//
// CPU restatement, in double precision, of the reference's path-tracing hot path.  Each
// routine names the reference lines it follows.  Arithmetic is written in the reference's
// operand order and the file is built with -ffp-contract=off for baseline x86-64, so that
// (a) closest hits equal the reference's bit for bit and (b) with the mt19937 sampler and
// one thread, whole films equal the reference's bit for bit.  oracle/_ref/ref_harness
// (the real reference, built in place) generated the fixtures that pin both claims.
#include "pt_oracle.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>
#include <random>
#include <map>
#include <vector>

namespace {

constexpr double kPi = 3.14159265358979323846;

// ----------------------------------------------------------------------------------------
// vector arithmetic — core/vec3.c:9-91
// ----------------------------------------------------------------------------------------
struct V3 { double x, y, z; };
inline V3 mk(const double* p) { return {p[0], p[1], p[2]}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 scale(V3 v, double t) { return {v.x * t, v.y * t, v.z * t}; }
inline V3 neg(V3 v) { return {-v.x, -v.y, -v.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double len2(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline double len(V3 v) { return std::sqrt(len2(v)); }
inline V3 normalize(V3 v) { double l = len(v); if (l > 0.0) return scale(v, 1.0 / l); return v; }
inline V3 reflect(V3 v, V3 n) { return sub(v, scale(n, 2.0 * dot(v, n))); }
inline V3 refract(V3 uv, V3 n, double eta) {  // core/vec3.c:69-79
  double cos_theta = std::fmin(dot(neg(uv), n), 1.0);
  V3 perp = scale(add(uv, scale(n, cos_theta)), eta);
  double par = -std::sqrt(std::fabs(1.0 - len2(perp)));
  return add(perp, scale(n, par));
}
inline V3 lerp(V3 a, V3 b, double t) { return add(scale(a, 1.0 - t), scale(b, t)); }

struct Ray { V3 o, d; };
inline V3 at(const Ray& r, double t) { return add(r.o, scale(r.d, t)); }  // core/ray.c:12

// hit_record (core/hit.h:20-29) without the tangent (only normal maps read it), plus the
// identity of the primitive that produced it.
struct Hit {
  V3 p{0, 0, 0}, n{0, 0, 0};
  V3 tangent{0, 0, 0};   // hit_record::tangent: every hit routine writes it
  double t = 0, u = 0, v = 0;
  bool front = false;
  int mat = 0;
  int ptype = -1, pidx = -1;
};
inline void face_normal(Hit& h, const Ray& r, V3 outward) {  // core/hit.c:21-24
  h.front = dot(r.d, outward) < 0;
  h.n = h.front ? outward : neg(outward);
}

// ----------------------------------------------------------------------------------------
// scene storage
// ----------------------------------------------------------------------------------------
struct Box6 { V3 lo{1e30, 1e30, 1e30}, hi{-1e30, -1e30, -1e30}; };  // AABB, engine/bvh.hpp:25-27
inline void grow(Box6& b, V3 p) {  // engine/bvh.hpp:69-76
  b.lo.x = std::fmin(b.lo.x, p.x); b.lo.y = std::fmin(b.lo.y, p.y); b.lo.z = std::fmin(b.lo.z, p.z);
  b.hi.x = std::fmax(b.hi.x, p.x); b.hi.y = std::fmax(b.hi.y, p.y); b.hi.z = std::fmax(b.hi.z, p.z);
}
inline void grow(Box6& b, const Box6& o) { grow(b, o.lo); grow(b, o.hi); }
inline V3 centroid(const Box6& b) { return {(b.lo.x + b.hi.x) * 0.5, (b.lo.y + b.hi.y) * 0.5, (b.lo.z + b.hi.z) * 0.5}; }

struct PrimRef { Box6 box; int index = 0, type = 0; };
struct Node { Box6 box; int left = -1, right = -1, first = 0, count = 0; };

struct EnvMap {
  std::vector<float> texel;
  int w = 0, h = 0;
  double intensity = 1.0, rotation = 0.0;
  std::vector<double> marginal, conditional;
  double total = 0.0;
  bool valid() const { return !texel.empty() && w > 0 && h > 0; }
  bool importance() const { return valid() && !marginal.empty() && total > 0.0; }  // texture.hpp:692-694
};

}  // namespace

struct orc_scene {
  std::vector<rtx_sphere> spheres;
  std::vector<rtx_plane> planes;
  std::vector<rtx_triangle> tris;
  std::vector<rtx_box> boxes;
  std::vector<rtx_point_light> points;
  std::vector<rtx_quad_light> quads;
  std::vector<rtx_disk_light> disks;
  std::vector<rtx_material> mats;
  std::vector<std::vector<uint8_t>> img_data;
  std::vector<int> img_w, img_h;
  EnvMap env;
  bool has_env = false;
  std::vector<int> emissive_spheres;  // Scene::get_emissive_spheres, engine/scene.hpp:222-230
  std::vector<Node> nodes;
  std::vector<PrimRef> prims;
};

namespace {

inline bool emissive(const rtx_material& m) {  // material.hpp:115-117
  return m.emission[0] > 0.0 || m.emission[1] > 0.0 || m.emission[2] > 0.0;
}

// ----------------------------------------------------------------------------------------
// BVH build — engine/bvh.hpp:148-218 on the primitive list of engine/scene.hpp:496-533
// ----------------------------------------------------------------------------------------
int longest_axis(const Box6& b) {  // bvh.hpp:100-107
  double dx = b.hi.x - b.lo.x, dy = b.hi.y - b.lo.y, dz = b.hi.z - b.lo.z;
  if (dx > dy && dx > dz) return 0;
  if (dy > dz) return 1;
  return 2;
}

// Nodes are numbered in preorder as the reference's emplace_back sequence numbers them (bvh.hpp:167-218): a node, its whole
// left subtree, its right subtree.  The node count under a range depends on its length only, so every subtree's slot is
// known before it is built and disjoint ranges can be built by different threads without changing a single node.
struct NodeCount {
  std::map<int, int> memo;
  int operator()(int count) {
    if (count <= 2) return 1;
    auto it = memo.find(count);
    if (it != memo.end()) return it->second;
    const int l = count / 2;
    const int v = 1 + (*this)(l) + (*this)(count - l);
    memo[count] = v;
    return v;
  }
  int lookup(int count) const { return count <= 2 ? 1 : memo.at(count); }
};

void build_range(orc_scene& s, const NodeCount& nc, int me, int start, int end) {
  Box6 bounds;
  for (int i = start; i < end; ++i) grow(bounds, s.prims[i].box);
  s.nodes[me].box = bounds;
  int count = end - start;
  if (count <= 2) { s.nodes[me].first = start; s.nodes[me].count = count; return; }
  Box6 cb;
  for (int i = start; i < end; ++i) grow(cb, centroid(s.prims[i].box));
  int axis = longest_axis(cb);
  int mid = start + count / 2;
  // Same library routine as the reference on the same sequence and ordering relation, hence
  // the same permutation.
  std::nth_element(s.prims.begin() + start, s.prims.begin() + mid, s.prims.begin() + end,
                   [axis](const PrimRef& a, const PrimRef& b) {
                     V3 ca = centroid(a.box), cb2 = centroid(b.box);
                     if (axis == 0) return ca.x < cb2.x;
                     if (axis == 1) return ca.y < cb2.y;
                     return ca.z < cb2.z;
                   });
  const int l = me + 1, r = me + 1 + nc.lookup(mid - start);
  s.nodes[me].left = l;
  s.nodes[me].right = r;
  if (count > 65536) {
#pragma omp task shared(s, nc) firstprivate(l, start, mid)
    build_range(s, nc, l, start, mid);
    build_range(s, nc, r, mid, end);
#pragma omp taskwait
  } else {
    build_range(s, nc, l, start, mid);
    build_range(s, nc, r, mid, end);
  }
}

void build_bvh(orc_scene& s) {
  s.prims.clear(); s.nodes.clear();
  s.prims.reserve(s.spheres.size() + s.tris.size() + s.boxes.size());
  for (size_t i = 0; i < s.spheres.size(); ++i) {
    PrimRef p; p.type = RTX_PRIM_SPHERE; p.index = (int)i;
    V3 c = mk(s.spheres[i].center); double r = s.spheres[i].radius; V3 rad = {r, r, r};
    p.box.lo = sub(c, rad); p.box.hi = add(c, rad);
    s.prims.push_back(p);
  }
  for (size_t i = 0; i < s.tris.size(); ++i) {
    PrimRef p; p.type = RTX_PRIM_TRIANGLE; p.index = (int)i;
    grow(p.box, mk(s.tris[i].v0)); grow(p.box, mk(s.tris[i].v1)); grow(p.box, mk(s.tris[i].v2));
    s.prims.push_back(p);
  }
  for (size_t i = 0; i < s.boxes.size(); ++i) {
    PrimRef p; p.type = RTX_PRIM_BOX; p.index = (int)i;
    p.box.lo = mk(s.boxes[i].box_min); p.box.hi = mk(s.boxes[i].box_max);
    s.prims.push_back(p);
  }
  if (s.prims.empty()) return;
  NodeCount nc;
  s.nodes.resize((size_t)nc((int)s.prims.size()));
#pragma omp parallel
#pragma omp single
  build_range(s, nc, 0, 0, (int)s.prims.size());
}

// ----------------------------------------------------------------------------------------
// intersection routines
// ----------------------------------------------------------------------------------------
bool slab(const Box6& b, const Ray& r, double t_min, double t_max) {  // AABB::hit, bvh.hpp:35-64
  double inv = 1.0 / r.d.x;
  double t0 = (b.lo.x - r.o.x) * inv, t1 = (b.hi.x - r.o.x) * inv;
  if (inv < 0.0) std::swap(t0, t1);
  t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
  if (t_max <= t_min) return false;
  inv = 1.0 / r.d.y;
  t0 = (b.lo.y - r.o.y) * inv; t1 = (b.hi.y - r.o.y) * inv;
  if (inv < 0.0) std::swap(t0, t1);
  t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
  if (t_max <= t_min) return false;
  inv = 1.0 / r.d.z;
  t0 = (b.lo.z - r.o.z) * inv; t1 = (b.hi.z - r.o.z) * inv;
  if (inv < 0.0) std::swap(t0, t1);
  t_min = t0 > t_min ? t0 : t_min; t_max = t1 < t_max ? t1 : t_max;
  if (t_max <= t_min) return false;
  return true;
}

bool hit_sphere(const rtx_sphere& s, const Ray& r, double t_min, double t_max, Hit& h) {  // sphere.hpp:71-105
  V3 c = mk(s.center);
  V3 oc = sub(r.o, c);
  double a = len2(r.d);
  double half_b = dot(oc, r.d);
  double cc = len2(oc) - s.radius * s.radius;
  double disc = half_b * half_b - a * cc;
  if (disc < 0) return false;
  double sq = std::sqrt(disc);
  double root = (-half_b - sq) / a;
  if (root < t_min || root > t_max) {
    root = (-half_b + sq) / a;
    if (root < t_min || root > t_max) return false;
  }
  h.t = root;
  h.p = at(r, root);
  V3 on = scale(sub(h.p, c), 1.0 / s.radius);
  face_normal(h, r, on);
  h.mat = s.material_id;
  double theta = std::acos(-on.y);                    // get_sphere_uv, sphere.hpp:26-32
  double phi = std::atan2(-on.z, on.x) + kPi;
  h.u = phi / (2.0 * kPi);
  h.v = theta / kPi;
  if (std::fabs(on.y) > 0.999) h.tangent = {1.0, 0.0, 0.0};   // get_sphere_tangent, sphere.hpp:39-50
  else h.tangent = normalize(V3{-on.z, 0.0, on.x});
  return true;
}

bool hit_plane(const rtx_plane& pl, const Ray& r, double t_min, double t_max, Hit& h) {  // primitives.hpp:35-60
  V3 n = mk(pl.normal);
  double denom = dot(n, r.d);
  if (std::fabs(denom) < 1e-8) return false;
  double t = dot(sub(mk(pl.point), r.o), n) / denom;
  if (t < t_min || t > t_max) return false;
  h.t = t; h.p = at(r, t);
  face_normal(h, r, n);
  h.mat = pl.material_id;
  V3 up = (std::fabs(n.y) < 0.999) ? V3{0, 1, 0} : V3{1, 0, 0};   // primitives.hpp:55-57
  h.tangent = normalize(cross(up, n));
  return true;  // u, v deliberately left as they were (the reference does not write them)
}

bool hit_triangle(const rtx_triangle& tr, const Ray& r, double t_min, double t_max, Hit& h) {  // primitives.hpp:78-126
  V3 v0 = mk(tr.v0);
  V3 e1 = sub(mk(tr.v1), v0), e2 = sub(mk(tr.v2), v0);
  V3 hh = cross(r.d, e2);
  double a = dot(e1, hh);
  if (std::fabs(a) < 1e-8) return false;
  double f = 1.0 / a;
  V3 s = sub(r.o, v0);
  double u = f * dot(s, hh);
  if (u < 0.0 || u > 1.0) return false;
  V3 q = cross(s, e1);
  double v = f * dot(r.d, q);
  if (v < 0.0 || u + v > 1.0) return false;
  double t = f * dot(e2, q);
  if (t < t_min || t > t_max) return false;
  h.t = t; h.p = at(r, t);
  face_normal(h, r, normalize(cross(e1, e2)));
  h.mat = tr.material_id;
  h.u = u; h.v = v;
  h.tangent = normalize(e1);   // primitives.hpp:119
  return true;
}

bool hit_box(const rtx_box& b, const Ray& r, double t_min, double t_max, Hit& h) {  // primitives.hpp:153-207
  double tmin = t_min, tmax = t_max;
  const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
  for (int ax = 0; ax < 3; ++ax) {
    double inv = 1.0 / d[ax];
    double t0 = (b.box_min[ax] - o[ax]) * inv, t1 = (b.box_max[ax] - o[ax]) * inv;
    if (inv < 0.0) std::swap(t0, t1);
    tmin = t0 > tmin ? t0 : tmin; tmax = t1 < tmax ? t1 : tmax;
    if (tmax <= tmin) return false;
  }
  h.t = tmin; h.p = at(r, tmin);
  V3 on;
  const double eps = 1e-4;
  if (std::fabs(h.p.x - b.box_min[0]) < eps) on = {-1, 0, 0};
  else if (std::fabs(h.p.x - b.box_max[0]) < eps) on = {1, 0, 0};
  else if (std::fabs(h.p.y - b.box_min[1]) < eps) on = {0, -1, 0};
  else if (std::fabs(h.p.y - b.box_max[1]) < eps) on = {0, 1, 0};
  else if (std::fabs(h.p.z - b.box_min[2]) < eps) on = {0, 0, -1};
  else on = {0, 0, 1};
  face_normal(h, r, on);
  h.mat = b.material_id;
  if (std::fabs(on.x) > 0.5) h.tangent = {0, 0, 1};   // primitives.hpp:197-204
  else h.tangent = {1, 0, 0};
  return true;
}

bool hit_quad(const rtx_quad_light& q, const Ray& r, double t_min, double t_max, Hit& h) {  // primitives.hpp:274-313
  V3 n = mk(q.normal);
  double denom = dot(n, r.d);
  if (std::fabs(denom) < 1e-8) return false;
  V3 corner = mk(q.corner);
  double t = dot(sub(corner, r.o), n) / denom;
  if (t < t_min || t > t_max) return false;
  V3 p = at(r, t);
  V3 d = sub(p, corner);
  V3 eu = mk(q.edge_u), ev = mk(q.edge_v);
  double up = dot(d, eu) / len2(eu), vp = dot(d, ev) / len2(ev);
  if (up < 0.0 || up > 1.0 || vp < 0.0 || vp > 1.0) return false;
  h.t = t; h.p = p;
  face_normal(h, r, n);
  h.u = up; h.v = vp;
  h.tangent = normalize(eu);   // primitives.hpp:310
  return true;
}

bool hit_disk(const rtx_disk_light& dl, const Ray& r, double t_min, double t_max, Hit& h) {  // primitives.hpp:396-433
  V3 n = mk(dl.normal);
  double denom = dot(n, r.d);
  if (std::fabs(denom) < 1e-8) return false;
  V3 c = mk(dl.center);
  double t = dot(sub(c, r.o), n) / denom;
  if (t < t_min || t > t_max) return false;
  V3 p = at(r, t);
  V3 d = sub(p, c);
  double d2 = len2(d);
  if (d2 > dl.radius * dl.radius) return false;
  h.t = t; h.p = p;
  face_normal(h, r, n);
  double dist = std::sqrt(d2);
  h.u = (dist > 1e-8) ? (dot(d, mk(dl.tangent)) / dist + 1.0) * 0.5 : 0.5;
  h.v = (dist > 1e-8) ? (dot(d, mk(dl.bitangent)) / dist + 1.0) * 0.5 : 0.5;
  h.tangent = mk(dl.tangent);   // primitives.hpp:425
  return true;
}

// Scene::hit — engine/scene.hpp:574-674.  `tmp` plays the reference's temp_rec: it persists
// across primitive tests, so members a routine does not write (plane/box: u, v) keep the
// value an earlier successful test left there.
bool scene_hit(const orc_scene& s, const Ray& r, double t_min, double t_max, Hit& rec) {
  Hit tmp;
  bool any = false;
  double closest = t_max;
  if (!s.nodes.empty()) {
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0 && sp < 64) {
      int ni = stack[--sp];
      if (ni < 0 || ni >= (int)s.nodes.size()) continue;
      const Node& nd = s.nodes[ni];
      if (!slab(nd.box, r, t_min, closest)) continue;
      if (nd.count > 0) {
        for (int i = 0; i < nd.count; ++i) {
          const PrimRef& pr = s.prims[nd.first + i];
          bool did = false;
          switch (pr.type) {
            case RTX_PRIM_SPHERE: did = hit_sphere(s.spheres[pr.index], r, t_min, closest, tmp); break;
            case RTX_PRIM_TRIANGLE: did = hit_triangle(s.tris[pr.index], r, t_min, closest, tmp); break;
            case RTX_PRIM_BOX: did = hit_box(s.boxes[pr.index], r, t_min, closest, tmp); break;
          }
          if (did) { any = true; closest = tmp.t; tmp.ptype = pr.type; tmp.pidx = pr.index; rec = tmp; }
        }
      } else {
        if (nd.right >= 0 && sp < 63) stack[sp++] = nd.right;
        if (nd.left >= 0 && sp < 63) stack[sp++] = nd.left;
      }
    }
  }
  for (size_t i = 0; i < s.planes.size(); ++i)
    if (hit_plane(s.planes[i], r, t_min, closest, tmp)) {
      any = true; closest = tmp.t; tmp.ptype = RTX_PRIM_PLANE; tmp.pidx = (int)i; rec = tmp;
    }
  for (size_t i = 0; i < s.quads.size(); ++i)
    if (hit_quad(s.quads[i], r, t_min, closest, tmp)) {
      any = true; closest = tmp.t; tmp.ptype = RTX_PRIM_QUAD_LIGHT; tmp.pidx = (int)i;
      tmp.mat = -1;  // QuadLight::material_id default (primitives.hpp:228), written by QuadLight::hit
      rec = tmp; rec.mat = -(int)i - 1;  // scene.hpp:657
    }
  for (size_t i = 0; i < s.disks.size(); ++i)
    if (hit_disk(s.disks[i], r, t_min, closest, tmp)) {
      any = true; closest = tmp.t; tmp.ptype = RTX_PRIM_DISK_LIGHT; tmp.pidx = (int)i;
      tmp.mat = -1;
      rec = tmp; rec.mat = -1000 - (int)i;  // scene.hpp:669
    }
  return any;
}

// ----------------------------------------------------------------------------------------
// environment map — engine/texture.hpp:403-473 (CDFs), :487-549 (lookup), :634-687 (pdf)
// ----------------------------------------------------------------------------------------
void build_env_cdf(EnvMap& e) {
  if (!e.valid()) return;
  e.marginal.assign(e.h + 1, 0.0);
  e.conditional.assign((size_t)e.h * (e.w + 1), 0.0);
  std::vector<double> rows(e.h);
  for (int y = 0; y < e.h; ++y) {
    double theta = kPi * (y + 0.5) / e.h;
    double st = std::sin(theta);
    double sum = 0.0;
    size_t off = (size_t)y * (e.w + 1);
    e.conditional[off] = 0.0;
    for (int x = 0; x < e.w; ++x) {
      size_t idx = ((size_t)y * e.w + x) * 3;
      float r = e.texel[idx], g = e.texel[idx + 1], b = e.texel[idx + 2];
      double lum = 0.2126 * r + 0.7152 * g + 0.0722 * b;
      sum += lum * st;
      e.conditional[off + x + 1] = sum;
    }
    if (sum > 0.0) for (int x = 1; x <= e.w; ++x) e.conditional[off + x] /= sum;
    else for (int x = 1; x <= e.w; ++x) e.conditional[off + x] = (double)x / e.w;
    rows[y] = sum;
  }
  e.marginal[0] = 0.0;
  for (int y = 0; y < e.h; ++y) e.marginal[y + 1] = e.marginal[y] + rows[y];
  e.total = e.marginal[e.h];
  if (e.total > 0.0) for (int y = 1; y <= e.h; ++y) e.marginal[y] /= e.total;
  else for (int y = 1; y <= e.h; ++y) e.marginal[y] = (double)y / e.h;
}

V3 env_lookup(const EnvMap& e, V3 direction) {
  if (!e.valid()) return {0, 0, 0};
  V3 d = normalize(direction);
  if (e.rotation != 0.0) {
    double cr = std::cos(e.rotation), sr = std::sin(e.rotation);
    double nx = d.x * cr - d.z * sr, nz = d.x * sr + d.z * cr;
    d.x = nx; d.z = nz;
  }
  double phi = std::atan2(d.z, d.x);
  double theta = std::asin(std::clamp(d.y, -1.0, 1.0));
  double u = 0.5 + phi / (2.0 * kPi);
  double v = 0.5 + theta / kPi;
  double fx = u * e.w - 0.5;
  double fy = (1.0 - v) * e.h - 0.5;
  int x0 = (int)std::floor(fx), y0 = (int)std::floor(fy);
  int x1 = x0 + 1, y1 = y0 + 1;
  double tx = fx - x0, ty = fy - y0;
  x0 = ((x0 % e.w) + e.w) % e.w; x1 = ((x1 % e.w) + e.w) % e.w;
  y0 = std::clamp(y0, 0, e.h - 1); y1 = std::clamp(y1, 0, e.h - 1);
  auto px = [&](int x, int y) -> V3 {
    size_t idx = ((size_t)y * e.w + x) * 3;
    return {e.texel[idx] * e.intensity, e.texel[idx + 1] * e.intensity, e.texel[idx + 2] * e.intensity};
  };
  V3 c0 = lerp(px(x0, y0), px(x1, y0), tx);
  V3 c1 = lerp(px(x0, y1), px(x1, y1), tx);
  return lerp(c0, c1, ty);
}

double env_pdf(const EnvMap& e, V3 direction) {
  if (!e.valid() || e.marginal.empty() || e.conditional.empty() || e.total <= 0.0) return 0.0;
  V3 d = normalize(direction);
  if (e.rotation != 0.0) {
    double cr = std::cos(e.rotation), sr = std::sin(e.rotation);
    double nx = d.x * cr - d.z * sr, nz = d.x * sr + d.z * cr;
    d.x = nx; d.z = nz;
  }
  double phi = std::atan2(d.z, d.x);
  double theta = std::acos(std::clamp(d.y, -1.0, 1.0));
  double u = 0.5 + phi / (2.0 * kPi);
  double v = theta / kPi;
  int x = (int)(u * e.w), y = (int)(v * e.h);
  x = std::clamp(x, 0, e.w - 1); y = std::clamp(y, 0, e.h - 1);
  double mp = (e.marginal[y + 1] - e.marginal[y]) * e.h;
  size_t off = (size_t)y * (e.w + 1);
  double cp = (e.conditional[off + x + 1] - e.conditional[off + x]) * e.w;
  double pdf_uv = mp * cp;
  double st = std::sin(theta);
  if (st < 1e-10) return 0.0;
  return pdf_uv / (2.0 * kPi * kPi * st);
}

struct EnvPick { V3 dir, emission; double pdf; bool valid; };
EnvPick env_sample_direction(const EnvMap& e, double u1, double u2) {  // texture.hpp:555-627
  EnvPick out{}; out.valid = false;
  if (!e.valid() || e.marginal.empty() || e.conditional.empty()) return out;
  int y = (int)(std::upper_bound(e.marginal.begin(), e.marginal.end(), u1) - e.marginal.begin()) - 1;
  y = std::clamp(y, 0, e.h - 1);
  double mb = e.marginal[y], ma = e.marginal[y + 1];
  double row_t = (ma > mb) ? (u1 - mb) / (ma - mb) : 0.0;
  size_t off = (size_t)y * (e.w + 1);
  auto cb = e.conditional.begin() + off;
  int x = (int)(std::upper_bound(cb, cb + e.w + 1, u2) - cb) - 1;
  x = std::clamp(x, 0, e.w - 1);
  double lo = e.conditional[off + x], hi = e.conditional[off + x + 1];
  double col_t = (hi > lo) ? (u2 - lo) / (hi - lo) : 0.0;
  double u = (x + col_t + 0.5) / e.w;
  double v = (y + row_t + 0.5) / e.h;
  double phi = (u - 0.5) * 2.0 * kPi;
  double theta = v * kPi;
  double st = std::sin(theta), ct = std::cos(theta);
  V3 dir = {st * std::cos(phi), ct, st * std::sin(phi)};
  if (e.rotation != 0.0) {
    double cr = std::cos(-e.rotation), sr = std::sin(-e.rotation);
    double nx = dir.x * cr - dir.z * sr, nz = dir.x * sr + dir.z * cr;
    dir.x = nx; dir.z = nz;
  }
  out.dir = dir;
  out.emission = env_lookup(e, dir);
  out.pdf = env_pdf(e, dir);
  out.valid = out.pdf > 0.0;
  return out;
}

// ----------------------------------------------------------------------------------------
// materials / textures — material.hpp:71-76, :308-312; texture.hpp:268-310
// ----------------------------------------------------------------------------------------
V3 albedo_at(const orc_scene& s, const rtx_material& m, V3 p, double u, double v) {
  if (!m.has_texture) return mk(m.albedo);
  switch (m.tex_type) {
    case RTX_TEX_SOLID: return mk(m.tex_color1);
    case RTX_TEX_CHECKER: {
      double inv = 1.0 / m.tex_scale;
      int x = (int)std::floor(p.x * inv), y = (int)std::floor(p.y * inv), z = (int)std::floor(p.z * inv);
      return ((x + y + z) % 2) == 0 ? mk(m.tex_color1) : mk(m.tex_color2);
    }
    case RTX_TEX_IMAGE: {
      if (m.tex_image < 0 || m.tex_image >= (int)s.img_data.size()) return {1.0, 0.0, 1.0};
      int w = s.img_w[m.tex_image], h = s.img_h[m.tex_image];
      if (w == 0 || h == 0) return {1.0, 0.0, 1.0};
      u = std::clamp(u, 0.0, 1.0);
      v = 1.0 - std::clamp(v, 0.0, 1.0);
      int i = (int)(u * w), j = (int)(v * h);
      i = std::min(i, w - 1); j = std::min(j, h - 1);
      const uint8_t* t = &s.img_data[m.tex_image][((size_t)j * w + i) * 3];
      return {t[0] / 255.0, t[1] / 255.0, t[2] / 255.0};
    }
  }
  return mk(m.tex_color1);
}

// NormalMap::sample (texture.hpp:83-109) followed by apply_normal_map (renderer.cpp:72-94).
void apply_normal_map(const orc_scene& s, const rtx_material& m, Hit& rec) {
  if (!m.has_normal_map) return;
  V3 tn{0.0, 0.0, 1.0};
  if (m.normal_image >= 0 && m.normal_image < (int)s.img_data.size() && s.img_w[m.normal_image] > 0 && s.img_h[m.normal_image] > 0) {
    int w = s.img_w[m.normal_image], h = s.img_h[m.normal_image];
    double u = std::clamp(rec.u, 0.0, 1.0);
    double v = 1.0 - std::clamp(rec.v, 0.0, 1.0);
    int i = std::min((int)(u * w), w - 1), j = std::min((int)(v * h), h - 1);
    const uint8_t* t = &s.img_data[m.normal_image][((size_t)j * w + i) * 3];
    double nx = (t[0] / 255.0) * 2.0 - 1.0, ny = (t[1] / 255.0) * 2.0 - 1.0, nz = (t[2] / 255.0) * 2.0 - 1.0;
    if (m.normal_strength < 1.0) {
      nx = nx * m.normal_strength;
      ny = ny * m.normal_strength;
      nz = nz * m.normal_strength + (1.0 - m.normal_strength);
    }
    tn = normalize(V3{nx, ny, nz});
  }
  V3 N = rec.n, T = rec.tangent;
  T = normalize(sub(T, scale(N, dot(N, T))));
  V3 B = cross(N, T);
  rec.n = normalize(V3{T.x * tn.x + B.x * tn.y + N.x * tn.z, T.y * tn.x + B.y * tn.y + N.y * tn.z, T.z * tn.x + B.z * tn.y + N.z * tn.z});
}

inline double schlick(double cosine, double ref_idx) {
  double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
  r0 = r0 * r0;
  return r0 + (1.0 - r0) * std::pow((1.0 - cosine), 5.0);
}

// ----------------------------------------------------------------------------------------
// random numbers
// ----------------------------------------------------------------------------------------
inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
  uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  uint32_t n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  uint32_t k[2] = {key[0], key[1]};
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
inline double u24(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }

// Sampler A: the reference's generator, drawn in the reference's order
// (engine/renderer.cpp:33-56).  std::uniform_real_distribution is the same library type
// the reference uses, so equal seeds give equal doubles.
struct MtSampler {
  std::mt19937 gen;
  std::uniform_real_distribution<double> dist{0.0, 1.0};
  double next() { return dist(gen); }
  double range(double lo, double hi) { return lo + (hi - lo) * next(); }
  void begin_sample(uint32_t, uint32_t, uint32_t) {}
  void begin_bounce(int) {}
  double jitter_x() { return next(); }
  double jitter_y() { return next(); }
  double light_select() { return next(); }
  double light_u() { return next(); }
  double light_v() { return next(); }
  double fresnel() { return next(); }
  double roulette() { return next(); }
  double scatter_u1() { return next(); }
  double scatter_u2() { return next(); }
  // the two uniforms MNEE::evaluate draws for its j-th light, in order (mnee.hpp:645, :678, :718-720)
  void mnee_pair(int, double& a, double& b) { a = next(); b = next(); }
  V3 unit_vector() {  // random_in_unit_sphere + normalize, renderer.cpp:45-56
    for (;;) {
      double a = range(-1, 1), b = range(-1, 1), c = range(-1, 1);
      V3 p = {a, b, c};
      if (len2(p) < 1.0) return normalize(p);
    }
  }
};

// Sampler B: counter-based, identical to the GPU's.  key = (pixel index, seed);
// counter = (sample index, block, 0, 0).  Block 0 holds the pixel jitter (top 24 bits of words
// 0 and 1, times 2^-24).  Path vertex b draws one block, 1+b, cut into 16-bit fields:
// word0 = light u | light v, word1 = scatter u1 | scatter u2, word2 = light select | roulette,
// word3 = fresnel | spare; a field k maps to (k + 0.5) / 65536.
inline void pcg4d(uint32_t x, uint32_t y, uint32_t z, uint32_t w, uint32_t out[4]) {   // the GPU's pcg4d (pt_device.hpp), word for word
  x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u; z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
  for (int i = 0; i < 3; ++i) {
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
  }
  out[0] = x; out[1] = y; out[2] = z; out[3] = w;
}

struct CounterSampler {
  bool pcg = false;   // fill blocks with pcg4d(pixel, sample, block, seed) instead of Philox
  void block(uint32_t blk, uint32_t o[4]) {
    if (pcg) { pcg4d(key[0], sample, blk, key[1], o); return; }
    uint32_t c[4] = {sample, blk, 0, 0};
    philox4x32_10(c, key, o);
  }
  uint32_t key[2]{0, 0}, sample = 0;
  uint32_t w[4];
  int bounce = -1;
  bool have = false;
  static double hi(uint32_t x) { return ((double)(x >> 16) + 0.5) * (1.0 / 65536.0); }
  static double lo(uint32_t x) { return ((double)(x & 0xffffu) + 0.5) * (1.0 / 65536.0); }
  void begin_sample(uint32_t pixel, uint32_t seed, uint32_t s) { key[0] = pixel; key[1] = seed; sample = s; bounce = -1; have = false; }
  void begin_bounce(int bn) { bounce = bn; have = false; }
  uint32_t* W() {
    if (!have) { block(bounce < 0 ? 0u : 1u + (uint32_t)bounce, w); have = true; }
    return w;
  }
  double jitter_x() { return u24(W()[0]); }
  double jitter_y() { return u24(W()[1]); }
  double light_u() { return hi(W()[0]); }
  double light_v() { return lo(W()[0]); }
  double light_select() { return hi(W()[2]); }
  double roulette() { return lo(W()[2]); }
  double fresnel() { return hi(W()[3]); }
  double scatter_u1() { return hi(W()[1]); }
  double scatter_u2() { return lo(W()[1]); }
  // MNEE light j of this vertex: extra block (1 + bounce) | ((1 + j / 4) << 16), word j % 4, high | low field
  void mnee_pair(int j, double& a, double& b) {
    uint32_t o[4];
    block((1u + (uint32_t)bounce) | ((1u + (uint32_t)j / 4u) << 16), o);
    a = hi(o[j & 3]); b = lo(o[j & 3]);
  }
  V3 unit_vector() {  // uniform on the sphere: same law as the normalised rejection sample
    double z = 1.0 - 2.0 * hi(W()[1]);
    double r = std::sqrt(std::fmax(0.0, 1.0 - z * z));
    double phi = 2.0 * kPi * lo(W()[1]);
    return {r * std::cos(phi), r * std::sin(phi), z};
  }
};

// ----------------------------------------------------------------------------------------
// light sampling — engine/scene.hpp:238-388 (sample), :407-463 (pdf), :470-488 (env pdf)
// ----------------------------------------------------------------------------------------
struct LightPick { V3 pos, n, Le; double pdf = 0, dist = 0; bool valid = false; };

inline V3 disk_sample_point(const rtx_disk_light& dl, double u, double v) {  // DiskLight::sample_point, primitives.hpp:350-384
  double sx = 2.0 * u - 1.0, sy = 2.0 * v - 1.0;
  V3 c = mk(dl.center);
  if (sx == 0 && sy == 0) return c;
  double rs, theta;
  if (std::fabs(sx) > std::fabs(sy)) { rs = sx; theta = (kPi / 4.0) * (sy / sx); }
  else { rs = sy; theta = (kPi / 2.0) - (kPi / 4.0) * (sx / sy); }
  double ct = std::cos(theta), st = std::sin(theta);
  double ar = dl.radius * rs;
  V3 offv = add(scale(mk(dl.tangent), ar * ct), scale(mk(dl.bitangent), ar * st));
  return add(c, offv);
}

template <class S>
LightPick sample_light(const orc_scene& s, V3 from, S& rng) {
  LightPick out;
  size_t np = s.points.size(), ne = s.emissive_spheres.size(), nq = s.quads.size(), nd = s.disks.size();
  size_t total = np + ne + nq + nd;
  bool has_env = s.has_env && s.env.importance();
  if (total == 0 && !has_env) return out;
  double rv = rng.light_select();
  bool pick_env = false;
  if (has_env && total > 0) {
    pick_env = (rv < 0.5);
    rv = pick_env ? (rv * 2.0) : ((rv - 0.5) * 2.0);
  } else if (has_env) {
    pick_env = true;
  }
  if (pick_env) {
    double u1 = rng.light_u(), u2 = rng.light_v();
    EnvPick e = env_sample_direction(s.env, u1, u2);
    if (e.valid) {
      out.pos = add(from, scale(e.dir, 1e6));
      out.n = neg(e.dir);
      out.Le = e.emission;
      out.pdf = e.pdf;
      out.dist = 1e6;
      out.valid = true;
      if (total > 0) out.pdf *= 0.5;
    }
    return out;
  }
  size_t idx = (size_t)(rv * total);
  if (idx >= total) idx = total - 1;
  size_t off = 0;
  if (idx < np) {
    const rtx_point_light& pl = s.points[idx];
    out.pos = mk(pl.position); out.n = {0, -1, 0}; out.Le = mk(pl.intensity);
    out.pdf = 1.0; out.dist = len(sub(out.pos, from)); out.valid = true;
  }
  off += np;
  if (!out.valid && idx < off + ne) {  // sample_sphere_light, scene.hpp:238-273
    const rtx_sphere& sp = s.spheres[s.emissive_spheres[idx - off]];
    double z = 1.0 - 2.0 * rng.light_u();
    double r = std::sqrt(std::fmax(0.0, 1.0 - z * z));
    double phi = 2.0 * kPi * rng.light_v();
    V3 local = {r * std::cos(phi), r * std::sin(phi), z};
    out.pos = add(mk(sp.center), scale(local, sp.radius));
    out.n = local;
    out.Le = mk(s.mats[sp.material_id].emission);
    double area = 4.0 * kPi * sp.radius * sp.radius;
    out.pdf = 1.0 / area;
    out.dist = len(sub(out.pos, from));
    out.valid = true;
  }
  off += ne;
  if (!out.valid && idx < off + nq) {
    const rtx_quad_light& q = s.quads[idx - off];
    double u = rng.light_u(), v = rng.light_v();
    out.pos = add(mk(q.corner), add(scale(mk(q.edge_u), u), scale(mk(q.edge_v), v)));
    out.n = mk(q.normal); out.Le = mk(q.emission);
    out.pdf = 1.0 / q.area;
    out.dist = len(sub(out.pos, from));
    out.valid = true;
  }
  off += nq;
  if (!out.valid && idx < off + nd) {  // DiskLight::sample_point, primitives.hpp:350-384
    const rtx_disk_light& dl = s.disks[idx - off];
    double u = rng.light_u(), v = rng.light_v();
    out.pos = disk_sample_point(dl, u, v);
    out.n = mk(dl.normal); out.Le = mk(dl.emission);
    out.pdf = 1.0 / dl.area;
    out.dist = len(sub(out.pos, from));
    out.valid = true;
  }
  if (out.valid) {
    out.pdf /= (double)total;
    if (has_env) out.pdf *= 0.5;
  }
  return out;
}

double light_pdf(const orc_scene& s, V3 lp) {
  size_t total = s.points.size() + s.emissive_spheres.size() + s.quads.size() + s.disks.size();
  bool has_env = s.has_env && s.env.importance();
  if (total == 0 && !has_env) return 0.0;
  double sel = 1.0 / (double)total;
  if (has_env) sel *= 0.5;
  for (int si : s.emissive_spheres) {
    const rtx_sphere& sp = s.spheres[si];
    double d2 = len2(sub(lp, mk(sp.center)));
    double r2 = sp.radius * sp.radius;
    if (std::abs(d2 - r2) < r2 * 0.01) { double area = 4.0 * kPi * r2; return sel / area; }
  }
  for (const rtx_quad_light& q : s.quads) {
    V3 tp = sub(lp, mk(q.corner));
    double dp = std::abs(dot(tp, mk(q.normal)));
    if (dp < 0.01) {
      double up = dot(tp, mk(q.edge_u)) / len2(mk(q.edge_u));
      double vp = dot(tp, mk(q.edge_v)) / len2(mk(q.edge_v));
      if (up >= -0.01 && up <= 1.01 && vp >= -0.01 && vp <= 1.01) return sel / q.area;
    }
  }
  for (const rtx_disk_light& dl : s.disks) {
    V3 tp = sub(lp, mk(dl.center));
    double dp = std::abs(dot(tp, mk(dl.normal)));
    double rad2 = len2(tp) - dp * dp;
    if (dp < 0.01 && rad2 <= dl.radius * dl.radius * 1.01) return sel / dl.area;
  }
  return 0.0;
}

double env_light_pdf(const orc_scene& s, V3 dir) {
  if (!s.has_env || !s.env.importance()) return 0.0;
  size_t total = s.points.size() + s.emissive_spheres.size() + s.quads.size() + s.disks.size();
  double pdf = env_pdf(s.env, dir);
  if (total > 0) pdf *= 0.5;
  return pdf;
}

V3 area_light_emission(const orc_scene& s, int id) {  // scene.hpp:686-700
  if (id >= -(int)s.quads.size() && id < 0) return mk(s.quads[(size_t)(-id - 1)].emission);
  if (id <= -1000) { size_t k = (size_t)(-id - 1000); if (k < s.disks.size()) return mk(s.disks[k].emission); }
  return {0, 0, 0};
}

inline double power_heuristic(double a, double b) { double a2 = a * a, b2 = b * b; return a2 / (a2 + b2); }

struct Tally { uint64_t closest = 0, shadow = 0, segments = 0; };

V3 background(const orc_scene& s, const rtx_settings& st, const Ray& r) {  // renderer.cpp:978-994
  if (s.has_env && s.env.valid()) return env_lookup(s.env, r.d);
  if (st.use_background_gradient) {
    V3 ud = normalize(r.d);
    double t = 0.5 * (ud.y + 1.0);
    return lerp(mk(st.background_bottom), mk(st.background_top), t);
  }
  return mk(st.background_color);
}

// Renderer::ray_color_path — engine/renderer.cpp:580-841.
// ----------------------------------------------------------------------------------------
// Manifold next event estimation — engine/spectral/mnee.hpp (called at engine/renderer.cpp:716-732)
// ----------------------------------------------------------------------------------------
struct MVertex { V3 p, n; double eta; int mat; };   // ManifoldVertex, mnee.hpp:32-39 (members the solver reads)
constexpr int kMneeMaxIter = 20, kMneeMaxDepth = 4;  // mnee.hpp:59-60
constexpr double kMneeEps = 1e-6, kMneeStepEps = 1e-10;

inline void mnee_frame(V3 n, V3& t, V3& b) {  // mnee.hpp:299-304
  V3 up = (std::fabs(n.y) < 0.9) ? V3{0, 1, 0} : V3{1, 0, 0};
  t = normalize(cross(up, n));
  b = cross(n, t);
}
inline void mnee_constraint(V3 prev, V3 cur, V3 next, double eta, V3 t, V3 b, double& cu, double& cv) {  // mnee.hpp:328-347
  V3 wi = normalize(sub(prev, cur)), wo = normalize(sub(next, cur));
  double wi_u = dot(wi, t), wi_v = dot(wi, b), wo_u = dot(wo, t), wo_v = dot(wo, b);
  cu = eta * wi_u + wo_u; cv = eta * wi_v + wo_v;
}
inline bool mnee_reproject(const orc_scene& s, V3& pos, V3& n, int expected, Tally& tl) {  // mnee.hpp:401-437
  V3 dirs[2] = {n, neg(n)};
  for (int d = 0; d < 2; ++d) {
    Hit rec; tl.closest++;
    if (scene_hit(s, Ray{pos, dirs[d]}, 0.0, 0.1, rec)) {
      if (rec.mat == expected) { pos = rec.p; n = rec.n; return true; }
    }
  }
  V3 off = add(pos, scale(n, 0.05));
  Hit rec; tl.closest++;
  if (scene_hit(s, Ray{off, neg(n)}, 0.0, 0.2, rec)) {
    if (rec.mat == expected) { pos = rec.p; n = rec.n; return true; }
  }
  return false;
}
// build_path, mnee.hpp:538-617: throughput of the chain, then a visibility walk along it
inline bool mnee_build_path(const orc_scene& s, V3 sp, V3 lp, const MVertex* ch, int n, double& throughput, Tally& tl) {
  throughput = 1.0;
  for (int i = 0; i < n; ++i) {
    V3 prev = (i == 0) ? sp : ch[i - 1].p, next = (i == n - 1) ? lp : ch[i + 1].p, cur = ch[i].p;
    V3 wi = normalize(sub(prev, cur)), wo = normalize(sub(next, cur));
    double eta = ch[i].eta;
    double cos_i = std::fabs(dot(wi, ch[i].n)), cos_t = std::fabs(dot(wo, ch[i].n));
    double r0 = (1.0 - eta) / (1.0 + eta);
    r0 = r0 * r0;
    double fr = r0 + (1.0 - r0) * std::pow(1.0 - cos_i, 5.0);
    double ft = 1.0 - fr;
    double geometry = (eta * eta * cos_t) / cos_i;
    throughput *= ft * geometry;
    if (throughput < 1e-10) return false;
  }
  V3 current = sp;
  for (int i = 0; i <= n; ++i) {
    V3 target = (i == n) ? lp : ch[i].p;
    V3 dir = normalize(sub(target, current));
    double dist = len(sub(target, current));
    Hit rec; tl.closest++;
    if (scene_hit(s, Ray{current, dir}, 0.001, dist - 0.001, rec)) {
      if (i < n) {
        double hd = len(sub(rec.p, ch[i].p));
        if (hd > 0.01) return false;
      } else return false;
    }
    if (i < n) current = ch[i].p;
  }
  return true;
}
// solve_manifold, mnee.hpp:442-533: Gauss-Seidel Newton on the per-vertex refraction constraints
inline bool mnee_solve(const orc_scene& s, V3 sp, V3 lp, MVertex* ch, int n, double& throughput, Tally& tl) {
  V3 tg[kMneeMaxDepth], bt[kMneeMaxDepth];
  for (int i = 0; i < n; ++i) mnee_frame(ch[i].n, tg[i], bt[i]);
  for (int iter = 0; iter < kMneeMaxIter; ++iter) {
    double max_c = 0.0, cu[kMneeMaxDepth], cv[kMneeMaxDepth];
    for (int i = 0; i < n; ++i) {
      V3 prev = (i == 0) ? sp : ch[i - 1].p, next = (i == n - 1) ? lp : ch[i + 1].p;
      mnee_constraint(prev, ch[i].p, next, ch[i].eta, tg[i], bt[i], cu[i], cv[i]);
      double mag = std::sqrt(cu[i] * cu[i] + cv[i] * cv[i]);
      max_c = std::fmax(max_c, mag);
    }
    if (max_c < kMneeEps) return mnee_build_path(s, sp, lp, ch, n, throughput, tl);
    for (int i = 0; i < n; ++i) {
      V3 prev = (i == 0) ? sp : ch[i - 1].p, next = (i == n - 1) ? lp : ch[i + 1].p;
      const double h = 1e-5;   // compute_jacobian_block, mnee.hpp:354-382
      double c0u, c0v, cuu, cuv, cvu, cvv;
      mnee_constraint(prev, ch[i].p, next, ch[i].eta, tg[i], bt[i], c0u, c0v);
      mnee_constraint(prev, add(ch[i].p, scale(tg[i], h)), next, ch[i].eta, tg[i], bt[i], cuu, cuv);
      mnee_constraint(prev, add(ch[i].p, scale(bt[i], h)), next, ch[i].eta, tg[i], bt[i], cvu, cvv);
      double J00 = (cuu - c0u) / h, J01 = (cvu - c0u) / h, J10 = (cuv - c0v) / h, J11 = (cvv - c0v) / h;
      double b0 = -cu[i], b1 = -cv[i], st0, st1;
      double det = J00 * J11 - J01 * J10;   // solve_2x2, mnee.hpp:387-396
      if (std::fabs(det) < kMneeStepEps) { st0 = -0.5 * cu[i]; st1 = -0.5 * cv[i]; }
      else { double inv = 1.0 / det; st0 = inv * (J11 * b0 - J01 * b1); st1 = inv * (-J10 * b0 + J00 * b1); }
      double mag = std::sqrt(st0 * st0 + st1 * st1);
      if (mag > 0.1) { double sc = 0.1 / mag; st0 *= sc; st1 *= sc; }
      V3 delta = add(scale(tg[i], st0), scale(bt[i], st1));
      ch[i].p = add(ch[i].p, delta);
      mnee_reproject(s, ch[i].p, ch[i].n, ch[i].mat, tl);
      mnee_frame(ch[i].n, tg[i], bt[i]);
    }
  }
  return false;
}
// connect_to_light_straight, mnee.hpp:94-143
inline bool mnee_straight(const orc_scene& s, V3 sp, V3 lp, double& throughput, Tally& tl) {
  V3 to_light = sub(lp, sp);
  double dist = len(to_light);
  V3 dir = scale(to_light, 1.0 / dist);
  MVertex ch[kMneeMaxDepth];
  int n = 0;
  Ray probe{sp, dir};
  V3 cur = sp;
  for (int i = 0; i < kMneeMaxDepth; ++i) {
    Hit rec;
    double max_t = len(sub(lp, cur)) - kMneeEps;
    tl.closest++;
    if (!scene_hit(s, probe, 0.001, max_t, rec)) break;
    // a quad / disk light in the way has a negative id (the reference indexes its material array with it:
    // undefined behaviour); defined here as "not a dielectric"
    if (rec.mat < 0 || s.mats[rec.mat].type != RTX_MAT_DIELECTRIC) return false;
    const rtx_material& m = s.mats[rec.mat];
    ch[n].p = rec.p; ch[n].n = rec.n;
    ch[n].eta = rec.front ? (1.0 / m.refraction_index) : m.refraction_index;
    ch[n].mat = rec.mat;
    ++n;
    cur = rec.p;
    probe = Ray{add(cur, scale(dir, 0.001)), dir};
  }
  if (n == 0) return false;
  return mnee_solve(s, sp, lp, ch, n, throughput, tl);
}
// try_sphere_seed_path, mnee.hpp:194-295
inline bool mnee_sphere_seed(const orc_scene& s, const rtx_sphere& sph, const rtx_material& m, V3 sp, V3 lp, double& throughput, Tally& tl) {
  V3 to_light = normalize(sub(lp, sp)), to_shade = normalize(sub(sp, lp));
  V3 c = mk(sph.center);
  V3 pts[2], nrm[2];
  const V3 origins[2] = {sp, lp}, dirs[2] = {to_light, to_shade};
  for (int k = 0; k < 2; ++k) {
    Hit rec;
    if (!hit_sphere(sph, Ray{origins[k], dirs[k]}, 0.001, 1e9, rec)) {
      V3 to_center = sub(c, origins[k]);
      double proj = dot(to_center, dirs[k]);
      if (proj < 0) return false;
      V3 closest = add(origins[k], scale(dirs[k], proj));
      V3 to_closest = sub(c, closest);
      double da = len(to_closest);
      if (da >= sph.radius) {
        V3 d2 = normalize(sub(closest, c));
        pts[k] = add(c, scale(d2, sph.radius));
      } else {
        double off = std::sqrt(sph.radius * sph.radius - da * da);
        pts[k] = add(origins[k], scale(dirs[k], proj - off));
      }
      nrm[k] = normalize(sub(pts[k], c));
    } else { pts[k] = rec.p; nrm[k] = rec.n; }
  }
  MVertex ch[kMneeMaxDepth];
  int mat_id = sph.material_id;
  ch[0] = {pts[0], nrm[0], 1.0 / m.refraction_index, mat_id};
  ch[1] = {pts[1], neg(nrm[1]), m.refraction_index, mat_id};
  return mnee_solve(s, sp, lp, ch, 2, throughput, tl);
}
// connect_to_light, mnee.hpp:73-88 (+ connect_to_light_seeded, :152-189)
inline bool mnee_connect(const orc_scene& s, V3 sp, V3 lp, double& throughput, Tally& tl) {
  if (mnee_straight(s, sp, lp, throughput, tl)) return true;
  for (const rtx_sphere& sph : s.spheres) {
    const rtx_material& m = s.mats[sph.material_id];
    if (m.type != RTX_MAT_DIELECTRIC) continue;
    V3 to_sphere = sub(mk(sph.center), sp), to_light = sub(lp, sp);
    double ds = len(to_sphere), dl = len(to_light);
    if (ds > dl + sph.radius) continue;
    if (ds < sph.radius) continue;
    V3 a = scale(to_sphere, 1.0 / ds), b = scale(to_light, 1.0 / dl);
    if (dot(a, b) < 0.0) continue;
    if (mnee_sphere_seed(s, sph, m, sp, lp, throughput, tl)) return true;
  }
  return false;
}
// MNEE::evaluate, mnee.hpp:632-763.  Returns whether any light contributed.
template <class S>
bool mnee_evaluate(const orc_scene& s, V3 hp, S& rng, V3& contribution, Tally& tl) {
  contribution = {0, 0, 0};
  bool any = false;
  int j = 0;
  for (const rtx_quad_light& q : s.quads) {
    double u, v; rng.mnee_pair(j++, u, v);
    V3 lp = add(mk(q.corner), add(scale(mk(q.edge_u), u), scale(mk(q.edge_v), v)));   // QuadLight::sample_point, primitives.hpp:258-262
    V3 ln = mk(q.normal);
    double w;
    if (mnee_connect(s, hp, lp, w, tl)) {
      V3 to_light = sub(lp, hp);
      double d2 = len2(to_light);
      V3 wil = normalize(neg(to_light));
      double cl = std::fabs(dot(ln, wil));
      double lpdf = d2 / (q.area * cl);
      contribution = add(contribution, scale(mk(q.emission), w * cl / (d2 * lpdf)));
      any = true;
    }
  }
  for (const rtx_disk_light& dl : s.disks) {
    double u, v; rng.mnee_pair(j++, u, v);
    V3 lp = disk_sample_point(dl, u, v);
    V3 ln = mk(dl.normal);
    double w;
    if (mnee_connect(s, hp, lp, w, tl)) {
      V3 to_light = sub(lp, hp);
      double d2 = len2(to_light);
      V3 wil = normalize(neg(to_light));
      double cl = std::fabs(dot(ln, wil));
      double lpdf = d2 / (dl.area * cl);
      contribution = add(contribution, scale(mk(dl.emission), w * cl / (d2 * lpdf)));
      any = true;
    }
  }
  for (int si : s.emissive_spheres) {   // spheres in index order whose material is emissive
    const rtx_sphere& sph = s.spheres[si];
    V3 to_sphere = sub(mk(sph.center), hp);
    double dc = len(to_sphere);
    V3 w3 = scale(to_sphere, 1.0 / dc);
    double ctm = std::sqrt(1.0 - (sph.radius * sph.radius) / (dc * dc));
    if (ctm < 0.0) ctm = 0.0;
    double r1, r2; rng.mnee_pair(j++, r1, r2);
    double ct = 1.0 + r1 * (ctm - 1.0);
    double sn = std::sqrt(1.0 - ct * ct);
    double phi = 2.0 * 3.14159265358979323846 * r2;
    V3 uv = (std::fabs(w3.x) > 0.9) ? V3{0, 1, 0} : V3{1, 0, 0};
    uv = normalize(cross(w3, uv));
    V3 vv = cross(w3, uv);
    V3 sd = add(scale(uv, sn * std::cos(phi)), add(scale(vv, sn * std::sin(phi)), scale(w3, ct)));
    double t = dc * ct - std::sqrt(sph.radius * sph.radius - dc * dc * (1.0 - ct * ct));
    V3 lp = add(hp, scale(sd, t));
    V3 ln = normalize(sub(lp, mk(sph.center)));
    double w;
    if (mnee_connect(s, hp, lp, w, tl)) {
      double solid = 2.0 * 3.14159265358979323846 * (1.0 - ctm);
      double pdf = 1.0 / solid;
      double cal = std::fabs(dot(ln, neg(sd)));
      contribution = add(contribution, scale(mk(s.mats[sph.material_id].emission), w * cal / pdf));
      any = true;
    }
  }
  return any;
}

template <class S>
V3 path_radiance(const orc_scene& s, const rtx_settings& st, Ray r, S& rng, Tally& tl) {
  V3 thr = {1.0, 1.0, 1.0}, acc = {0.0, 0.0, 0.0};
  Ray cur = r;
  bool spec = false, was_diffuse = false;
  double bsdf_pdf = 0.0;
  const bool nee = st.use_nee != 0, mis = st.use_mis != 0;
  const bool env_is = s.has_env && s.env.importance();
  for (int bounce = 0; bounce < st.max_depth; ++bounce) {
    rng.begin_bounce(bounce);
    Hit rec;
    tl.closest++;
    if (!scene_hit(s, cur, 0.001, 1e9, rec)) {
      V3 bg = background(s, st, cur);
      bool handled = nee && mis && was_diffuse && env_is;
      if (!handled) acc = add(acc, mul(thr, bg));
      break;
    }
    tl.segments++;
    if (rec.mat < 0) {  // area light, renderer.cpp:609-630
      V3 Le = area_light_emission(s, rec.mat);
      if (bounce == 0 || spec || !nee) {
        acc = add(acc, mul(thr, Le));
      } else if (mis && was_diffuse && bsdf_pdf > 0) {
        double hd = rec.t;
        double cl = std::abs(dot(rec.n, neg(normalize(cur.d))));
        double lpa = light_pdf(s, rec.p);
        if (lpa > 0 && cl > 0) {
          double lps = lpa * hd * hd / cl;
          double w = power_heuristic(bsdf_pdf, lps);
          acc = add(acc, scale(mul(thr, Le), w));
        }
      }
      break;
    }
    const rtx_material& m = s.mats[rec.mat];
    apply_normal_map(s, m, rec);   // renderer.cpp:635
    V3 surf = albedo_at(s, m, rec.p, rec.u, rec.v);
    if (bounce == 0 || spec || !nee) acc = add(acc, mul(thr, mk(m.emission)));
    V3 sdir{0, 0, 0};
    V3 atten = surf;
    spec = false; was_diffuse = false; bsdf_pdf = 0.0;
    switch (m.type) {
      case RTX_MAT_LAMBERTIAN: {
        if (nee) {
          bool skip_mnee = false;
          LightPick ls = sample_light(s, rec.p, rng);
          if (ls.valid) {
            bool is_env = ls.dist > 1e5;
            V3 ldir; double ldist;
            if (is_env) { ldir = normalize(sub(ls.pos, rec.p)); ldist = 1e9; }
            else { V3 tl2 = sub(ls.pos, rec.p); ldist = len(tl2); ldir = scale(tl2, 1.0 / ldist); }
            double cs = dot(rec.n, ldir);
            if (cs > 0) {
              Hit sh;
              tl.shadow++;
              bool occluded = scene_hit(s, Ray{rec.p, ldir}, 0.001, is_env ? 1e9 : (ldist - 0.001), sh);
              if (!occluded) {
                double lps; bool ok = true;
                if (is_env) lps = ls.pdf;
                else {
                  double cl = std::abs(dot(ls.n, neg(ldir)));
                  if (cl <= 0) ok = false;
                  lps = ok ? ls.pdf * ldist * ldist / cl : 0.0;
                }
                if (ok && lps > 0) {
                  double bpl = cs / kPi;
                  double w = 1.0;
                  if (mis) w = power_heuristic(lps, bpl);
                  V3 direct = scale(mul(mul(ls.Le, surf), thr), cs / (kPi * lps) * w);
                  acc = add(acc, direct);
                } else skip_mnee = true;   // `goto skip_nee` jumps over the MNEE block too (renderer.cpp:691, :695)
              }
            }
          }
          if (st.use_mnee && !skip_mnee) {  // renderer.cpp:716-732
            V3 mc;
            if (mnee_evaluate(s, rec.p, rng, mc, tl)) {
              V3 md = mul(mul(mc, surf), thr);
              md = scale(md, 1.0 / 3.14159265358979323846);
              acc = add(acc, md);
            }
          }
        }
        sdir = add(rec.n, rng.unit_vector());
        if (len2(sdir) < 1e-8) sdir = rec.n;
        sdir = normalize(sdir);
        bsdf_pdf = dot(rec.n, sdir) / kPi;
        was_diffuse = true;
        break;
      }
      case RTX_MAT_METAL: {
        spec = true;
        V3 refl = reflect(normalize(cur.d), rec.n);
        if (m.fuzz > 0.0) { refl = add(refl, scale(rng.unit_vector(), m.fuzz)); spec = (m.fuzz < 0.1); }
        sdir = refl;  // a direction below the surface is NOT terminated (renderer.cpp:755-758)
        break;
      }
      case RTX_MAT_DIELECTRIC: {
        spec = true;
        atten = {1.0, 1.0, 1.0};
        double ratio = rec.front ? (1.0 / m.refraction_index) : m.refraction_index;
        V3 ud = normalize(cur.d);
        double ct = std::fmin(dot(neg(ud), rec.n), 1.0);
        double sn = std::sqrt(1.0 - ct * ct);
        if (ratio * sn > 1.0 || schlick(ct, ratio) > rng.fresnel()) sdir = reflect(ud, rec.n);
        else sdir = refract(ud, rec.n, ratio);
        break;
      }
    }
    thr = mul(thr, atten);
    if (bounce > 3) {  // renderer.cpp:785-790
      double p = std::fmax(thr.x, std::fmax(thr.y, thr.z));
      if (p < 0.01) break;
      if (rng.roulette() > p) break;
      thr = scale(thr, 1.0 / p);
    }
    Ray next{rec.p, sdir};
    if (nee && mis && !spec && bsdf_pdf > 0) {  // look-ahead, renderer.cpp:797-835
      Hit nr;
      tl.closest++;
      if (scene_hit(s, next, 0.001, 1e9, nr)) {
        // A negative id (quad/disk light) indexes before the material array in the reference
        // (undefined behaviour); defined here as "not emissive": the arrival branch above
        // already weights those hits.
        if (nr.mat >= 0 && emissive(s.mats[nr.mat])) {
          double hd = nr.t;
          double cl = std::abs(dot(nr.n, neg(sdir)));
          double lpa = light_pdf(s, nr.p);
          if (lpa > 0 && cl > 0) {
            double lps = lpa * hd * hd / cl;
            double w = power_heuristic(bsdf_pdf, lps);
            acc = add(acc, scale(mul(thr, mk(s.mats[nr.mat].emission)), w));
          }
        }
      } else if (env_is) {
        V3 ee = env_lookup(s.env, sdir);
        double ep = env_light_pdf(s, sdir);
        if (ep > 0) { double w = power_heuristic(bsdf_pdf, ep); acc = add(acc, scale(mul(thr, ee), w)); }
      }
    }
    cur = next;
  }
  return acc;
}

// Renderer::ray_color_whitted — engine/renderer.cpp:483-569 (recursive, as the reference; `level` counts
// the recursion depth for the counter-based sampler)
template <class S>
V3 whitted_radiance(const orc_scene& s, const rtx_settings& st, Ray r, int depth, int level, S& rng, Tally& tl) {
  if (depth <= 0) return {0, 0, 0};
  rng.begin_bounce(level);
  Hit rec;
  tl.closest++;
  if (scene_hit(s, r, 0.001, 1e9, rec)) {
    tl.segments++;
    if (rec.mat >= 0) {   // a quad / disk light hit indexes materials[negative] in the reference (undefined); defined as a miss
      const rtx_material& m = s.mats[rec.mat];
      apply_normal_map(s, m, rec);   // renderer.cpp:496
      V3 surf = albedo_at(s, m, rec.p, rec.u, rec.v);
      switch (m.type) {
        case RTX_MAT_LAMBERTIAN: {
          V3 direct = {0, 0, 0};
          for (const rtx_point_light& pl : s.points) {
            V3 to_light = sub(mk(pl.position), rec.p);   // Scene::is_shadowed, scene.hpp:554-564: unnormalised direction, t up to |to_light|
            double distance = len(to_light);
            Hit sh;
            tl.shadow++;
            if (!scene_hit(s, Ray{rec.p, to_light}, 0.001, distance, sh)) {
              V3 ldir = normalize(sub(mk(pl.position), rec.p));
              double diffuse = std::fmax(0.0, dot(rec.n, ldir));
              direct = add(direct, scale(mk(pl.intensity), diffuse));
            }
          }
          V3 sdir = add(rec.n, rng.unit_vector());
          if (len2(sdir) < 1e-8) sdir = rec.n;
          V3 indirect = whitted_radiance(s, st, Ray{rec.p, sdir}, depth - 1, level + 1, rng, tl);
          return mul(surf, add(direct, indirect));
        }
        case RTX_MAT_METAL: {
          V3 refl = reflect(normalize(r.d), rec.n);
          if (dot(refl, rec.n) > 0) return mul(surf, whitted_radiance(s, st, Ray{rec.p, refl}, depth - 1, level + 1, rng, tl));
          return {0, 0, 0};
        }
        case RTX_MAT_DIELECTRIC: {
          double ratio = rec.front ? (1.0 / m.refraction_index) : m.refraction_index;
          V3 ud = normalize(r.d);
          double ct = std::fmin(dot(neg(ud), rec.n), 1.0);
          double sn = std::sqrt(1.0 - ct * ct);
          V3 dir;
          if (ratio * sn > 1.0 || schlick(ct, ratio) > rng.fresnel()) dir = reflect(ud, rec.n);
          else dir = refract(ud, rec.n, ratio);
          return mul(V3{1.0, 1.0, 1.0}, whitted_radiance(s, st, Ray{rec.p, dir}, depth - 1, level + 1, rng, tl));
        }
      }
    }
  }
  return background(s, st, r);
}


// ----------------------------------------------------------------------------------------
// BDPTIntegrator — engine/bdpt.cpp.  Sampler B numbers the blocks of the light subpath from 0x8001 (its origin) on;
// a vertex of either subpath draws from the fields of its own block: word0 = cosine-hemisphere r1 | r2 of the light's
// emission direction, word1 = scatter r1 | r2 (or the unit vector, or the point on a quad / disk light),
// word2 = light choice | roulette, word3 = fresnel.
// ----------------------------------------------------------------------------------------
constexpr double kBdptEps = 1e-6;          // bdpt.cpp:14
constexpr double kInvPi = 0.31830988618379067;   // material.hpp:24

struct BVertex {                            // PathVertex, bdpt.hpp:54-110 (the members the integrator reads)
  int type = 2;                             // 0 camera, 1 light, 2 surface
  V3 p{0, 0, 0}, n{0, 1, 0};
  int mat_type = RTX_MAT_LAMBERTIAN;
  V3 albedo{1, 1, 1};
  double roughness = 0.0, ior = 1.5;
  V3 thr{1, 1, 1};
  double pdf_fwd = 0.0;
  V3 Le{0, 0, 0};
  V3 wi{0, 0, 0}, wo{0, 0, 0};
  bool connectible() const {               // bdpt.hpp:88-101
    if (type == 1 || type == 0) return true;
    return !(mat_type == RTX_MAT_METAL && roughness < 0.001) && mat_type != RTX_MAT_DIELECTRIC;
  }
  bool is_light() const { return type == 1 || len2(Le) > 0.001; }   // bdpt.hpp:106-109
};

inline V3 bdpt_cosine_hemisphere(V3 normal, double r1, double r2) {   // bdpt.cpp:39-65
  double phi = 2.0 * kPi * r1;
  double cos_theta = std::sqrt(r2), sin_theta = std::sqrt(1.0 - r2);
  double x = std::cos(phi) * sin_theta, y = std::sin(phi) * sin_theta, z = cos_theta;
  V3 w = normal;
  V3 a = (std::fabs(w.x) > 0.9) ? V3{0, 1, 0} : V3{1, 0, 0};
  V3 u = normalize(cross(a, w));
  V3 v = cross(w, u);
  return normalize(V3{u.x * x + v.x * y + w.x * z, u.y * x + v.y * y + w.y * z, u.z * x + v.z * y + w.z * z});
}

inline V3 bdpt_eval_bsdf(const BVertex& v, V3 wi, V3 wo) {   // bdpt.cpp:489-527
  double cos_o = dot(v.n, wo);
  if (v.mat_type == RTX_MAT_LAMBERTIAN) {
    if (cos_o <= 0) return {0, 0, 0};
    return scale(v.albedo, kInvPi);
  }
  if (v.mat_type == RTX_MAT_METAL) {
    if (cos_o <= 0) return {0, 0, 0};
    V3 reflected = reflect(neg(wi), v.n);
    double cos_r = dot(reflected, wo);
    if (v.roughness < 0.001) return {0, 0, 0};
    double n = 2.0 / (v.roughness * v.roughness) - 2.0;
    if (cos_r > 0) { double D = (n + 1.0) * kInvPi * 0.5 * std::pow(cos_r, n); return scale(v.albedo, D); }
    return {0, 0, 0};
  }
  return {0, 0, 0};
}

inline double bdpt_bsdf_pdf(const BVertex& v, V3 wi, V3 wo) {   // bdpt.cpp:529-560
  double cos_o = dot(v.n, wo);
  if (v.mat_type == RTX_MAT_LAMBERTIAN) { if (cos_o <= 0) return 0; return cos_o * kInvPi; }
  if (v.mat_type == RTX_MAT_METAL) {
    if (cos_o <= 0) return 0;
    if (v.roughness < 0.001) return 0;
    V3 reflected = reflect(neg(wi), v.n);
    double cos_r = dot(reflected, wo);
    if (cos_r > 0) { double n = 2.0 / (v.roughness * v.roughness) - 2.0; return (n + 1.0) * kInvPi * 0.5 * std::pow(cos_r, n); }
    return 0;
  }
  return 0;
}

template <class S> bool bdpt_sample_bsdf(const BVertex& v, V3 wi, V3& wo, double& pdf, V3& f, S& rng) {   // bdpt.cpp:562-631
  if (v.mat_type == RTX_MAT_LAMBERTIAN) {
    double r1 = rng.scatter_u1(), r2 = rng.scatter_u2();
    wo = bdpt_cosine_hemisphere(v.n, r1, r2);
    pdf = dot(v.n, wo) * kInvPi;
    f = scale(v.albedo, kInvPi);
    return true;
  }
  if (v.mat_type == RTX_MAT_METAL) {
    V3 reflected = reflect(neg(wi), v.n);
    if (v.roughness < 0.001) { wo = reflected; pdf = 1.0; f = v.albedo; return true; }
    V3 jit = scale(rng.unit_vector(), v.roughness);
    wo = normalize(add(reflected, jit));
    if (dot(wo, v.n) <= 0) return false;
    double cos_r = dot(reflected, wo);
    double n = 2.0 / (v.roughness * v.roughness) - 2.0;
    pdf = (n + 1.0) * kInvPi * 0.5 * std::pow(std::max(0.0, cos_r), n);
    f = v.albedo;
    return true;
  }
  if (v.mat_type == RTX_MAT_DIELECTRIC) {
    double cos_i = dot(wi, v.n);
    double eta = (cos_i > 0) ? (1.0 / v.ior) : v.ior;
    V3 n = (cos_i > 0) ? v.n : neg(v.n);
    cos_i = std::fabs(cos_i);
    double sin2 = eta * eta * (1.0 - cos_i * cos_i);
    double r0 = (1.0 - v.ior) / (1.0 + v.ior);
    r0 = r0 * r0;
    double fresnel = r0 + (1.0 - r0) * std::pow(1.0 - cos_i, 5.0);
    if (sin2 > 1.0 || rng.fresnel() < fresnel) wo = reflect(neg(wi), n);   // the uniform is drawn only when refraction is possible
    else wo = refract(neg(wi), n, eta);
    pdf = 1.0; f = {1, 1, 1};
    return true;
  }
  return false;
}

// the loop shared by trace_eye_path (bdpt.cpp:145-217) and trace_light_path (:248-306); eye = area-light hits end the
// path WITH a light vertex, surface vertices carry the material's emission
template <class S> void bdpt_walk(const orc_scene& s, const rtx_settings& st, Ray ray, V3 thr, bool eye, int level0, std::vector<BVertex>& path, S& rng, Tally& tl) {
  for (int bounce = 0; bounce < st.max_depth; ++bounce) {
    rng.begin_bounce(level0 + bounce);
    Hit rec;
    tl.closest++;
    if (!scene_hit(s, ray, 0.001, 1e9, rec)) break;
    if (rec.mat < 0) {
      if (eye) {
        tl.segments++;
        BVertex lv;
        lv.type = 1; lv.p = rec.p; lv.n = rec.n; lv.Le = area_light_emission(s, rec.mat); lv.thr = thr;
        lv.wi = neg(normalize(ray.d));
        path.push_back(lv);
      }
      break;
    }
    tl.segments++;
    const rtx_material& m = s.mats[rec.mat];
    BVertex v;
    v.type = 2; v.p = rec.p; v.n = rec.n; v.mat_type = m.type;
    v.albedo = albedo_at(s, m, rec.p, rec.u, rec.v);
    if (eye) v.Le = mk(m.emission);
    v.roughness = m.fuzz; v.ior = m.refraction_index;
    v.thr = thr;
    v.wi = neg(normalize(ray.d));
    path.push_back(v);
    V3 wo, f;
    double pdf;
    if (!bdpt_sample_bsdf(v, v.wi, wo, pdf, f, rng)) break;
    if (pdf < kBdptEps) break;
    double cos_theta = std::fabs(dot(v.n, wo));
    thr = mul(thr, scale(f, cos_theta / pdf));
    if (bounce > 3) {
      double q = std::max(0.05, 1.0 - std::max({thr.x, thr.y, thr.z}));
      if (rng.roulette() < q) break;
      thr = scale(thr, 1.0 / (1.0 - q));
    }
    path.back().wo = wo;
    path.back().pdf_fwd = pdf;
    ray = Ray{rec.p, wo};
  }
}

template <class S> bool bdpt_light_origin(const orc_scene& s, BVertex& v, V3& dir, double& pdf_pos, double& pdf_dir, S& rng) {   // bdpt.cpp:311-413
  size_t n_point = s.points.size(), n_sph = 0;
  for (const rtx_sphere& sp : s.spheres) if (len2(mk(s.mats[sp.material_id].emission)) > 0.001) n_sph++;
  size_t n_quad = s.quads.size(), n_disk = s.disks.size();
  size_t total = n_point + n_sph + n_quad + n_disk;
  if (total == 0) return false;
  double choice_pdf = 1.0 / total;
  size_t idx = (size_t)(rng.light_select() * total);
  idx = std::min(idx, total - 1);
  v.type = 1;
  if (idx < n_point) {
    const rtx_point_light& l = s.points[idx];
    v.p = mk(l.position); v.n = {0, 1, 0}; v.Le = mk(l.intensity);
    pdf_pos = choice_pdf;
    dir = rng.unit_vector();
    pdf_dir = 1.0 / (4.0 * kPi);
    return true;
  }
  idx -= n_point;
  if (idx < n_sph) {
    size_t k = 0;
    for (const rtx_sphere& sp : s.spheres) {
      const rtx_material& m = s.mats[sp.material_id];
      if (len2(mk(m.emission)) > 0.001) {
        if (k == idx) {
          V3 d = rng.unit_vector();
          v.p = add(mk(sp.center), scale(d, sp.radius)); v.n = d; v.Le = mk(m.emission);
          double area = 4.0 * kPi * sp.radius * sp.radius;
          pdf_pos = choice_pdf / area;
          double r1 = rng.light_u(), r2 = rng.light_v();
          dir = bdpt_cosine_hemisphere(d, r1, r2);
          pdf_dir = dot(d, dir) * kInvPi;
          return true;
        }
        k++;
      }
    }
  }
  idx -= n_sph;
  if (idx < n_quad) {
    const rtx_quad_light& q = s.quads[idx];
    double u = rng.scatter_u1(), w = rng.scatter_u2();
    v.p = add(mk(q.corner), add(scale(mk(q.edge_u), u), scale(mk(q.edge_v), w)));   // QuadLight::sample_point, primitives.hpp:258-262
    v.n = mk(q.normal); v.Le = mk(q.emission);
    pdf_pos = choice_pdf * (1.0 / q.area);
    double r1 = rng.light_u(), r2 = rng.light_v();
    dir = bdpt_cosine_hemisphere(v.n, r1, r2);
    pdf_dir = dot(v.n, dir) * kInvPi;
    return true;
  }
  idx -= n_quad;
  if (idx < n_disk) {
    const rtx_disk_light& dl = s.disks[idx];
    double u = rng.scatter_u1(), w = rng.scatter_u2();
    v.p = disk_sample_point(dl, u, w);
    v.n = mk(dl.normal); v.Le = mk(dl.emission);
    pdf_pos = choice_pdf * (1.0 / dl.area);
    double r1 = rng.light_u(), r2 = rng.light_v();
    dir = bdpt_cosine_hemisphere(v.n, r1, r2);
    pdf_dir = dot(v.n, dir) * kInvPi;
    return true;
  }
  return false;
}

inline double bdpt_mis_weight(const std::vector<BVertex>& eye, const std::vector<BVertex>& light, int s_, int t) {   // bdpt.cpp:633-711
  if (s_ == 0 && t == 1) return 1.0;
  const BVertex* ev = (t > 0 && t <= (int)eye.size()) ? &eye[t - 1] : nullptr;
  const BVertex* lv = (s_ > 0 && s_ <= (int)light.size()) ? &light[s_ - 1] : nullptr;
  double sum_ri = 1.0;
  double ri = 1.0;
  if (t > 1 && ev && ev->connectible()) {
    double pf = ev->pdf_fwd, pr = bdpt_bsdf_pdf(*ev, ev->wo, ev->wi);
    if (pf > kBdptEps && pr > kBdptEps) { ri *= pr / pf; double ri2 = ri * ri; sum_ri += ri2; }
  }
  ri = 1.0;
  if (s_ > 0 && lv && lv->connectible()) {
    double pf = lv->pdf_fwd, pr = bdpt_bsdf_pdf(*lv, lv->wo, lv->wi);
    if (pf > kBdptEps && pr > kBdptEps) { ri *= pr / pf; double ri2 = ri * ri; sum_ri += ri2; }
  }
  if (s_ == 0 && ev && ev->is_light()) sum_ri += 1.0;
  if (s_ == 1 && lv) sum_ri += 0.5;
  return 1.0 / sum_ri;
}

inline V3 bdpt_connect(const orc_scene& s, const std::vector<BVertex>& eye, const std::vector<BVertex>& light, int s_, int t, Tally& tl) {   // bdpt.cpp:415-487
  V3 L = {0, 0, 0};
  if (t == 0) return L;
  if (s_ == 0) {
    const BVertex& ev = eye[t - 1];
    if (ev.is_light()) L = mul(ev.thr, ev.Le);
    return L;
  }
  if (t == 1) return L;
  const BVertex& ev = eye[t - 1];
  const BVertex& lv = light[s_ - 1];
  {  // visible(), bdpt.cpp:713-723
    V3 d = sub(lv.p, ev.p);
    double dist = len(d);
    if (!(dist < kBdptEps)) {
      Hit rec;
      tl.shadow++;
      if (scene_hit(s, Ray{ev.p, scale(d, 1.0 / dist)}, 0.001, dist - 0.001, rec)) return L;
    }
  }
  V3 d = sub(lv.p, ev.p);
  double dist_sq = len2(d);
  double dist = std::sqrt(dist_sq);
  V3 wo_eye = scale(d, 1.0 / dist);
  V3 wi_light = neg(wo_eye);
  V3 f_eye = bdpt_eval_bsdf(ev, ev.wi, wo_eye);
  if (len2(f_eye) < kBdptEps) return L;
  V3 f_light;
  if (s_ == 1) f_light = scale(lv.Le, std::fabs(dot(lv.n, wi_light)));
  else f_light = bdpt_eval_bsdf(lv, lv.wi, wi_light);
  if (len2(f_light) < kBdptEps) return L;
  double G = 0.0;   // geometry_term, bdpt.cpp:725-738
  {
    V3 dd = sub(lv.p, ev.p);
    double d2 = len2(dd);
    if (!(d2 < kBdptEps)) {
      double dl = std::sqrt(d2);
      V3 dir = scale(dd, 1.0 / dl);
      G = std::fabs(dot(ev.n, dir)) * std::fabs(dot(lv.n, neg(dir))) / d2;
    }
  }
  V3 eye_c = mul(ev.thr, f_eye);
  V3 light_c = (s_ == 1) ? f_light : mul(lv.thr, f_light);
  return scale(mul(eye_c, light_c), G);
}

template <class S>
V3 bdpt_radiance(const orc_scene& s, const rtx_settings& st, Ray r, S& rng, Tally& tl) {   // BDPTIntegrator::Li, bdpt.cpp:71-128
  std::vector<BVertex> eye, light;
  {  // trace_eye_path, bdpt.cpp:130-220
    BVertex cam;
    cam.type = 0; cam.p = r.o; cam.thr = {1, 1, 1}; cam.pdf_fwd = 1.0;
    eye.push_back(cam);
    bdpt_walk(s, st, r, V3{1, 1, 1}, true, 0, eye, rng, tl);
  }
  {  // trace_light_path, bdpt.cpp:222-309
    rng.begin_bounce(0x8000);
    BVertex lv;
    V3 dir;
    double pdf_pos, pdf_dir;
    if (bdpt_light_origin(s, lv, dir, pdf_pos, pdf_dir, rng)) {
      lv.pdf_fwd = pdf_pos;
      light.push_back(lv);
      if (!(pdf_dir < kBdptEps)) {
        double cos_theta = std::fabs(dot(lv.n, dir));
        V3 thr = scale(lv.Le, cos_theta / (pdf_pos * pdf_dir));
        bdpt_walk(s, st, Ray{lv.p, dir}, thr, false, 0x8001, light, rng, tl);
      }
    }
  }
  const int num_eye = (int)eye.size(), num_light = (int)light.size();
  V3 L = {0, 0, 0};
  for (int t = 1; t <= num_eye; ++t)
    for (int s_ = 0; s_ <= num_light; ++s_) {
      if (s_ + t < 2) continue;
      if (t > 1 && !eye[t - 1].connectible()) continue;
      if (s_ > 0 && !light[s_ - 1].connectible()) continue;
      V3 Lp = bdpt_connect(s, eye, light, s_, t, tl);
      double w = st.use_mis ? bdpt_mis_weight(eye, light, s_, t) : 1.0;
      V3 wl = scale(Lp, w);
      if (st.clamp_max > 0.0) {
        double lum = 0.2126 * wl.x + 0.7152 * wl.y + 0.0722 * wl.z;
        if (lum > st.clamp_max) wl = scale(wl, st.clamp_max / lum);
      }
      L = add(L, wl);
    }
  return L;
}


// ----------------------------------------------------------------------------------------
// PathPhoton mode — engine/renderer.cpp:344-372 (photon pass), :844-976 (ray_color_path_with_caustics),
// engine/photon_integrator.hpp:93-293 (caustic photons), engine/photon_map.hpp (kd-tree, k-nearest estimate).
// The reference draws the photon pass from two generators: PhotonIntegrator::rng_ (seed 42) and, for the aim point
// of every photon, a thread_local std::mt19937 seeded with the hash of the calling thread's id
// (photon_integrator.hpp:457-459), which differs from process to process; g_photon_aim_seed carries the value a
// particular reference run used (the harness prints it), so that run can be reproduced bit for bit.
// ----------------------------------------------------------------------------------------
unsigned long long g_photon_aim_seed = 0;
struct PhotonMapO;
const PhotonMapO* g_photon_map = nullptr;   // the map of the render in progress (one render at a time)

struct OPhoton { V3 p, dir, power; uint8_t axis = 0; };

struct PhotonMapO {   // photon_map.hpp:70-317, with the same standard-library algorithms
  std::vector<OPhoton> ph;
  static double ax(const V3& p, int a) { return a == 0 ? p.x : a == 1 ? p.y : p.z; }
  void build_tree(size_t start, size_t end, int depth) {
    if (end - start <= 1) return;
    int axis = depth % 3;
    size_t mid = start + (end - start) / 2;
    std::nth_element(ph.begin() + start, ph.begin() + mid, ph.begin() + end,
                     [axis](const OPhoton& a, const OPhoton& b) { return ax(a.p, axis) < ax(b.p, axis); });
    ph[mid].axis = (uint8_t)axis;
    build_tree(start, mid, depth + 1);
    build_tree(mid + 1, end, depth + 1);
  }
  void build() { if (!ph.empty()) build_tree(0, ph.size(), 0); }
  struct Res { const OPhoton* p; double d2; bool operator<(const Res& o) const { return d2 < o.d2; } };
  void search(size_t start, size_t end, V3 pos, int k, double max_d2, std::priority_queue<Res>& heap) const {
    if (start >= end) return;
    size_t mid = start + (end - start) / 2;
    const OPhoton& q = ph[mid];
    double d2 = len2(sub(q.p, pos));
    double rad = max_d2;
    if ((int)heap.size() >= k) rad = std::min(rad, heap.top().d2);
    if (d2 < rad) {
      Res r{&q, d2};
      if ((int)heap.size() < k) heap.push(r);
      else if (d2 < heap.top().d2) { heap.pop(); heap.push(r); }
    }
    int axis = q.axis;
    double delta = ax(pos, axis) - ax(q.p, axis);
    double delta2 = delta * delta;
    if (delta < 0) {
      search(start, mid, pos, k, max_d2, heap);
      if ((int)heap.size() < k || delta2 < heap.top().d2) search(mid + 1, end, pos, k, max_d2, heap);
    } else {
      search(mid + 1, end, pos, k, max_d2, heap);
      if ((int)heap.size() < k || delta2 < heap.top().d2) search(start, mid, pos, k, max_d2, heap);
    }
  }
  V3 estimate(V3 pos, V3 normal, int k, double max_radius) const {   // estimate_irradiance, photon_map.hpp:150-186
    double max_d2 = max_radius * max_radius;
    if (ph.empty()) return {0, 0, 0};
    std::priority_queue<Res> heap;
    search(0, ph.size(), pos, k, max_d2, heap);
    std::vector<Res> near;
    near.reserve(heap.size());
    while (!heap.empty()) { near.push_back(heap.top()); heap.pop(); }
    std::reverse(near.begin(), near.end());
    if (near.empty()) return {0, 0, 0};
    double r2 = near.back().d2;
    if (r2 < 1e-10) r2 = 1e-10;
    V3 flux = {0, 0, 0};
    for (const Res& r : near) {
      double cos_theta = -dot(r.p->dir, normal);
      if (cos_theta <= 0) continue;
      double w = 1.0 - std::sqrt(r.d2 / r2);
      flux = add(flux, scale(r.p->power, w));
    }
    double area = kPi * r2;
    return scale(flux, 3.0 / area);
  }
};

// the draws of the photon pass, in the reference's order
struct PhotonMt {
  std::mt19937 rng{42};          // PhotonIntegrator::rng_, photon_integrator.hpp:46
  std::mt19937 aim;              // sample_sphere()'s thread_local generator
  std::uniform_real_distribution<double> dist{0.0, 1.0};
  void begin(uint64_t) {}
  double pick() { return dist(rng); }
  double light_u() { return dist(rng); }
  double light_v() { return dist(rng); }
  double aim_u() { return dist(aim); }
  double aim_v() { return dist(aim); }
  double roulette(int) { return dist(rng); }
  double fresnel(int) { return dist(rng); }
  void lambert(int) { (void)dist(rng); }   // drawn and, in the caustic pass, not used (photon_integrator.hpp:284-286)
};
// counter-based: photon i is "pixel" i, sample 0xfffffffe; block 0 = pick | -, light u | v, aim u | v; block 1 + depth = roulette, fresnel
struct PhotonCounter {
  CounterSampler cs;
  uint32_t seed = 0;
  uint32_t b0[4], bd[4];
  int have_d = -1;
  void begin(uint64_t i) { cs.begin_sample((uint32_t)i, seed, 0xfffffffeu); cs.block(0u, b0); have_d = -1; }
  uint32_t* D(int depth) { if (have_d != depth) { cs.block(1u + (uint32_t)depth, bd); have_d = depth; } return bd; }
  double pick() { return CounterSampler::hi(b0[2]); }
  double light_u() { return CounterSampler::hi(b0[0]); }
  double light_v() { return CounterSampler::lo(b0[0]); }
  double aim_u() { return CounterSampler::hi(b0[1]); }
  double aim_v() { return CounterSampler::lo(b0[1]); }
  double roulette(int d) { return CounterSampler::lo(D(d)[2]); }
  double fresnel(int d) { return CounterSampler::hi(D(d)[3]); }
  void lambert(int) {}
};

template <class PS>
void trace_photon(const orc_scene& s, Ray r, V3 power, int depth, bool hit_specular, PS& ps, PhotonMapO& map, Tally& tl) {   // photon_integrator.hpp:202-293, caustic_only = true
  for (;;) {   // the reference recurses; every branch is a tail call
    if (depth >= 10) return;
    double max_component = std::max({power.x, power.y, power.z});
    if (depth > 3) {
      double cp = std::min(0.9, max_component);
      if (ps.roulette(depth) > cp) return;
      power = scale(power, 1.0 / cp);
    }
    Hit rec;
    tl.closest++;
    if (!scene_hit(s, r, 0.001, 1e30, rec)) return;
    if (rec.mat < 0) return;   // a quad / disk light: the reference indexes materials[negative] (undefined); defined as "the photon is lost"
    const rtx_material& m = s.mats[rec.mat];
    if (m.type == RTX_MAT_DIELECTRIC) {
      V3 outward; double ni_over_nt, cosine;
      if (dot(r.d, rec.n) > 0) { outward = neg(rec.n); ni_over_nt = m.refraction_index; cosine = m.refraction_index * dot(r.d, rec.n); }
      else { outward = rec.n; ni_over_nt = 1.0 / m.refraction_index; cosine = -dot(r.d, rec.n); }
      V3 refracted{0, 0, 0};
      double reflect_prob = 1.0;
      {  // refract_vec, :477-489
        V3 uv = normalize(r.d);
        double dt = dot(uv, outward);
        double disc = 1.0 - ni_over_nt * ni_over_nt * (1.0 - dt * dt);
        if (disc > 0) {
          refracted = sub(scale(sub(uv, scale(outward, dt)), ni_over_nt), scale(outward, std::sqrt(disc)));
          double r0 = (1.0 - m.refraction_index) / (1.0 + m.refraction_index);   // schlick, :491-495
          r0 = r0 * r0;
          reflect_prob = r0 + (1.0 - r0) * std::pow(1.0 - cosine, 5.0);
        }
      }
      if (ps.fresnel(depth) < reflect_prob) {
        V3 refl = sub(r.d, scale(rec.n, 2.0 * dot(r.d, rec.n)));
        r = Ray{add(rec.p, scale(refl, 0.001)), refl};
      } else {
        power = mul(power, mk(m.albedo));
        r = Ray{add(rec.p, scale(refracted, 0.001)), refracted};
      }
      depth++; hit_specular = true;
      continue;
    }
    if (m.type == RTX_MAT_METAL) {
      V3 refl = sub(r.d, scale(rec.n, 2.0 * dot(r.d, rec.n)));
      power = mul(power, mk(m.albedo));
      r = Ray{add(rec.p, scale(refl, 0.001)), refl};
      depth++; hit_specular = true;
      continue;
    }
    if (m.type == RTX_MAT_LAMBERTIAN) {
      if (hit_specular) { OPhoton q; q.p = rec.p; q.dir = neg(r.d); q.power = power; map.ph.push_back(q); }
      ps.lambert(depth);
    }
    return;
  }
}

template <class PS>
void trace_caustic_photons(const orc_scene& s, size_t n_caustic, PS& ps, PhotonMapO& map, Tally& tl) {   // photon_integrator.hpp:93-200
  std::vector<size_t> lights;
  double total_power = 0.0;
  for (size_t i = 0; i < s.spheres.size(); ++i) {
    const rtx_material& m = s.mats[s.spheres[i].material_id];
    if (m.emission[0] > 0 || m.emission[1] > 0 || m.emission[2] > 0) { lights.push_back(i); total_power += (m.emission[0] + m.emission[1] + m.emission[2]) / 3.0; }
  }
  if (lights.empty()) return;
  std::vector<size_t> targets;
  for (size_t i = 0; i < s.spheres.size(); ++i) {
    const rtx_material& m = s.mats[s.spheres[i].material_id];
    if (m.type == RTX_MAT_DIELECTRIC || m.type == RTX_MAT_METAL || m.fuzz < 0.1) targets.push_back(i);
  }
  if (targets.empty()) return;
  uint64_t photon_index = 0;
  for (size_t li : lights) {
    const rtx_sphere& ls = s.spheres[li];
    const rtx_material& lm = s.mats[ls.material_id];
    double light_power = (lm.emission[0] + lm.emission[1] + lm.emission[2]) / 3.0;
    double fraction = light_power / total_power;
    size_t n = (size_t)(n_caustic * fraction);
    double area = 4.0 * kPi * ls.radius * ls.radius;
    V3 photon_power = scale(mk(lm.emission), area / (double)n);
    for (size_t i = 0; i < n; ++i, ++photon_index) {
      ps.begin(photon_index);
      size_t ti = targets[(size_t)(ps.pick() * targets.size()) % targets.size()];
      const rtx_sphere& tg = s.spheres[ti];
      double u = ps.light_u(), v = ps.light_v();
      double theta = 2.0 * kPi * u, phi = std::acos(2.0 * v - 1.0);
      V3 local = {std::sin(phi) * std::cos(theta), std::sin(phi) * std::sin(theta), std::cos(phi)};
      V3 origin = add(mk(ls.center), scale(local, ls.radius * 1.001));
      double au = ps.aim_u(), av = ps.aim_v();   // sample_sphere, :457-471
      double at = 2.0 * kPi * au, ap = std::acos(2.0 * av - 1.0);
      V3 sp = {std::sin(ap) * std::cos(at), std::sin(ap) * std::sin(at), std::cos(ap)};
      V3 target_point = add(mk(tg.center), scale(sp, tg.radius * 0.5));
      V3 to_target = sub(target_point, origin);
      V3 dir = normalize(to_target);
      double dist_sq = len2(to_target);
      double solid_angle = kPi * tg.radius * tg.radius / dist_sq;
      V3 adjusted = scale(photon_power, 4.0 * kPi / solid_angle);
      trace_photon(s, Ray{origin, dir}, adjusted, 0, true, ps, map, tl);
    }
  }
}

// Renderer::ray_color_path_with_caustics — engine/renderer.cpp:844-976
template <class S>
V3 pathphoton_radiance(const orc_scene& s, const rtx_settings& st, Ray r, const PhotonMapO& map, S& rng, Tally& tl) {
  V3 thr = {1.0, 1.0, 1.0}, acc = {0.0, 0.0, 0.0};
  Ray cur = r;
  bool spec = false;
  const bool nee = st.use_nee != 0;
  for (int bounce = 0; bounce < st.max_depth; ++bounce) {
    rng.begin_bounce(bounce);
    Hit rec;
    tl.closest++;
    if (!scene_hit(s, cur, 0.001, 1e9, rec)) { acc = add(acc, mul(thr, background(s, st, cur))); break; }
    tl.segments++;
    if (rec.mat < 0) {
      if (bounce == 0 || spec || !nee) acc = add(acc, mul(thr, area_light_emission(s, rec.mat)));
      break;
    }
    const rtx_material& m = s.mats[rec.mat];
    apply_normal_map(s, m, rec);
    V3 surf = albedo_at(s, m, rec.p, rec.u, rec.v);
    if (bounce == 0 || spec || !nee) acc = add(acc, mul(thr, mk(m.emission)));
    V3 sdir{0, 0, 0};
    V3 atten = surf;
    spec = false;
    if (m.type == RTX_MAT_LAMBERTIAN) {
      if (!map.ph.empty()) {
        V3 caustic = map.estimate(rec.p, rec.n, st.photon_gather_count, (double)st.caustic_gather_radius);
        acc = add(acc, mul(thr, mul(surf, caustic)));
      }
      if (nee) {
        LightPick ls = sample_light(s, rec.p, rng);
        if (ls.valid) {
          V3 to_light = sub(ls.pos, rec.p);
          double ld = len(to_light);
          V3 ldir = scale(to_light, 1.0 / ld);
          double cs = dot(rec.n, ldir);
          if (cs > 0) {
            Hit sh;
            bool is_env = ls.dist > 1e5;
            double sd = is_env ? 1e9 : ld - 0.001;
            tl.shadow++;
            if (!scene_hit(s, Ray{rec.p, ldir}, 0.001, sd, sh)) {
              double cl = std::abs(dot(ls.n, neg(ldir)));
              double falloff = is_env ? 1.0 : 1.0 / (ld * ld);
              V3 direct = scale(mul(surf, ls.Le), cs * falloff * cl / (ls.pdf * kPi));
              acc = add(acc, mul(thr, direct));
            }
          }
        }
      }
      sdir = add(rec.n, rng.unit_vector());
      if (len2(sdir) < 1e-8) sdir = rec.n;
      sdir = normalize(sdir);
    } else if (m.type == RTX_MAT_METAL) {
      spec = true;
      V3 refl = reflect(normalize(cur.d), rec.n);
      if (m.fuzz > 0.0) { refl = add(refl, scale(rng.unit_vector(), m.fuzz)); spec = (m.fuzz < 0.1); }
      sdir = refl;   // both `break`s at :937-938 leave the switch only
    } else if (m.type == RTX_MAT_DIELECTRIC) {
      spec = true;
      atten = {1.0, 1.0, 1.0};
      double ratio = rec.front ? (1.0 / m.refraction_index) : m.refraction_index;
      V3 ud = normalize(cur.d);
      double ct = std::fmin(dot(neg(ud), rec.n), 1.0);
      double sn = std::sqrt(1.0 - ct * ct);
      if (ratio * sn > 1.0 || schlick(ct, ratio) > rng.fresnel()) sdir = reflect(ud, rec.n);
      else sdir = refract(ud, rec.n, ratio);
    } else return acc;
    thr = mul(thr, atten);
    if (bounce > 3) {
      double p = std::fmax(thr.x, std::fmax(thr.y, thr.z));
      if (p < 0.01) break;
      if (rng.roulette() > p) break;
      thr = scale(thr, 1.0 / p);
    }
    cur = Ray{rec.p, sdir};
  }
  return acc;
}

inline Ray camera_ray(const rtx_camera& c, double s, double t) {  // scene.hpp:99-108
  V3 o = mk(c.origin);
  V3 d = sub(add(add(mk(c.lower_left_corner), scale(mk(c.horizontal), s)), scale(mk(c.vertical), t)), o);
  return {o, d};
}

template <class S>
void render_rows(const orc_scene& s, const rtx_camera& cam, const rtx_settings& st, S& rng, int y, double* film, Tally& tl) {
  for (int x = 0; x < st.width; ++x) {
    V3 px = {0, 0, 0};
    for (int k = 0; k < st.samples_per_pixel; ++k) {
      rng.begin_sample((uint32_t)(y * st.width + x), st.seed, (uint32_t)k);
      double u = ((double)x + rng.jitter_x()) / st.width;
      double v = ((double)y + rng.jitter_y()) / st.height;
      Ray r = camera_ray(cam, u, v);
      px = add(px, st.mode == RTX_MODE_WHITTED ? whitted_radiance(s, st, r, st.max_depth, 0, rng, tl) : st.mode == RTX_MODE_BDPT ? bdpt_radiance(s, st, r, rng, tl)
                   : st.mode == RTX_MODE_PATHPHOTON ? pathphoton_radiance(s, st, r, *g_photon_map, rng, tl) : path_radiance(s, st, r, rng, tl));
    }
    px = scale(px, 1.0 / st.samples_per_pixel);
    double* o = film + ((size_t)y * st.width + x) * 3;
    o[0] = px.x; o[1] = px.y; o[2] = px.z;
  }
}

}  // namespace

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
extern "C" {

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

int orc_create(const rtx_scene_desc* d, orc_scene** out) {
  if (!d || !out) return RTX_ERR_INVALID_ARGUMENT;
  orc_scene* s = new orc_scene();
  s->spheres.assign(d->spheres, d->spheres + d->n_spheres);
  s->planes.assign(d->planes, d->planes + d->n_planes);
  s->tris.assign(d->triangles, d->triangles + d->n_triangles);
  s->boxes.assign(d->boxes, d->boxes + d->n_boxes);
  s->points.assign(d->point_lights, d->point_lights + d->n_point_lights);
  s->quads.assign(d->quad_lights, d->quad_lights + d->n_quad_lights);
  s->disks.assign(d->disk_lights, d->disk_lights + d->n_disk_lights);
  s->mats.assign(d->materials, d->materials + d->n_materials);
  for (size_t i = 0; i < d->n_images; ++i) {
    const rtx_image& im = d->images[i];
    s->img_data.emplace_back(im.rgb, im.rgb + (size_t)im.width * im.height * 3);
    s->img_w.push_back(im.width); s->img_h.push_back(im.height);
  }
  if (d->environment && d->environment->rgb && d->environment->width > 0 && d->environment->height > 0) {
    const rtx_environment& e = *d->environment;
    s->env.texel.assign(e.rgb, e.rgb + (size_t)e.width * e.height * 3);
    s->env.w = e.width; s->env.h = e.height; s->env.intensity = e.intensity; s->env.rotation = e.rotation;
    build_env_cdf(s->env);
    s->has_env = true;
  }
  for (size_t i = 0; i < s->spheres.size(); ++i) {
    int m = s->spheres[i].material_id;
    if (m >= 0 && m < (int)s->mats.size() && emissive(s->mats[m])) s->emissive_spheres.push_back((int)i);
  }
  build_bvh(*s);
  *out = s;
  return RTX_OK;
}

void orc_destroy(orc_scene* s) { delete s; }

int orc_bvh_counts(const orc_scene* s, int64_t* n_nodes, int64_t* n_prims) {
  *n_nodes = (int64_t)s->nodes.size(); *n_prims = (int64_t)s->prims.size();
  return RTX_OK;
}

int orc_bvh_dump(const orc_scene* s, double* boxes6, int32_t* node_ints4, int32_t* prim_ints2) {
  for (size_t i = 0; i < s->nodes.size(); ++i) {
    const Node& n = s->nodes[i];
    double b[6] = {n.box.lo.x, n.box.lo.y, n.box.lo.z, n.box.hi.x, n.box.hi.y, n.box.hi.z};
    memcpy(boxes6 + i * 6, b, sizeof b);
    node_ints4[i * 4] = n.left; node_ints4[i * 4 + 1] = n.right; node_ints4[i * 4 + 2] = n.first; node_ints4[i * 4 + 3] = n.count;
  }
  for (size_t i = 0; i < s->prims.size(); ++i) { prim_ints2[i * 2] = s->prims[i].type; prim_ints2[i * 2 + 1] = s->prims[i].index; }
  return RTX_OK;
}

int orc_trace_closest(const orc_scene* s, const double* rays8, size_t n, rtx_hit* out, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
  for (long long i = 0; i < (long long)n; ++i) {
    const double* q = rays8 + (size_t)i * 8;
    Ray r{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
    Hit h;
    bool ok = scene_hit(*s, r, q[6], q[7], h);
    rtx_hit& o = out[i];
    memset(&o, 0, sizeof o);
    o.hit = ok; o.prim_type = ok ? h.ptype : -1; o.prim_index = ok ? h.pidx : -1;
    if (ok) {
      o.material_id = h.mat; o.front_face = h.front; o.t = h.t;
      o.point[0] = h.p.x; o.point[1] = h.p.y; o.point[2] = h.p.z;
      o.normal[0] = h.n.x; o.normal[1] = h.n.y; o.normal[2] = h.n.z;
      o.u = h.u; o.v = h.v;
    }
  }
  return RTX_OK;
}

int orc_render(const orc_scene* s, const rtx_camera* cam, const rtx_settings* st, int rng_mode, int threads,
               double* film, rtx_stats* stats) {
  if (!s || !cam || !st || !film) return RTX_ERR_INVALID_ARGUMENT;
  if (threads < 1) threads = 1;
  const int W = st->width, H = st->height;
  const int sc = st->shard_count > 0 ? st->shard_count : 1, si = st->shard_index;
  memset(film, 0, sizeof(double) * (size_t)W * H * 3);
  uint64_t closest = 0, shadow = 0, segments = 0, rows_done = 0;
  double t0 = omp_get_wtime();
  PhotonMapO photon_map;
  if (st->mode == RTX_MODE_PATHPHOTON) {   // engine/renderer.cpp:344-372: the photon pass comes first, on the calling thread
    Tally ptl;
    if (rng_mode == ORC_RNG_MT19937) { PhotonMt ps; ps.aim.seed(g_photon_aim_seed); trace_caustic_photons(*s, (size_t)st->caustic_photon_count, ps, photon_map, ptl); }
    else { PhotonCounter ps; ps.cs.pcg = rng_mode == ORC_RNG_COUNTER_PCG; ps.seed = st->seed; trace_caustic_photons(*s, (size_t)st->caustic_photon_count, ps, photon_map, ptl); }
    photon_map.build();
    closest += ptl.closest;
    g_photon_map = &photon_map;
  }
  if (rng_mode == ORC_RNG_MT19937) {
    // One generator per thread seeded seed + thread id, rows handed out in order; with one
    // thread this is the reference's traversal order exactly.
#pragma omp parallel num_threads(threads) reduction(+ : closest, shadow, segments, rows_done)
    {
      MtSampler rng;
      rng.gen.seed(st->seed + (unsigned)omp_get_thread_num());
      Tally tl;
#pragma omp for schedule(dynamic, 1)
      for (int y = 0; y < H; ++y) {
        if (y % sc != si) continue;
        render_rows(*s, *cam, *st, rng, y, film, tl);
        rows_done++;
      }
      closest += tl.closest; shadow += tl.shadow; segments += tl.segments;
    }
  } else {
#pragma omp parallel num_threads(threads) reduction(+ : closest, shadow, segments, rows_done)
    {
      CounterSampler rng;
      rng.pcg = rng_mode == ORC_RNG_COUNTER_PCG;
      Tally tl;
#pragma omp for schedule(dynamic, 1)
      for (int y = 0; y < H; ++y) {
        if (y % sc != si) continue;
        render_rows(*s, *cam, *st, rng, y, film, tl);
        rows_done++;
      }
      closest += tl.closest; shadow += tl.shadow; segments += tl.segments;
    }
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->samples = rows_done * (uint64_t)W * (uint64_t)st->samples_per_pixel;
    stats->rays_closest = closest; stats->rays_shadow = shadow; stats->segments = segments;
    stats->seconds_render = stats->seconds_total = omp_get_wtime() - t0;
  }
  return RTX_OK;
}

int orc_kat(const orc_scene* s, unsigned seed, int n, double* out, size_t cap, size_t* written) {
  MtSampler rng;
  rng.gen.seed(seed);
  std::vector<double> o;
  for (int i = 0; i < n; ++i) {
    double px = rng.range(-3, 3), py = rng.range(0, 5), pz = rng.range(-5, 2);
    V3 p = {px, py, pz};
    LightPick ls = sample_light(*s, p, rng);
    o.push_back(ls.valid ? 1.0 : 0.0);
    if (ls.valid) {
      double v[11] = {ls.pos.x, ls.pos.y, ls.pos.z, ls.n.x, ls.n.y, ls.n.z, ls.Le.x, ls.Le.y, ls.Le.z, ls.pdf, ls.dist};
      o.insert(o.end(), v, v + 11);
      o.push_back(light_pdf(*s, ls.pos));
    }
    double dx = rng.range(-1, 1), dy = rng.range(-1, 1), dz = rng.range(-1, 1);
    V3 dir = {dx, dy, dz};
    o.push_back(env_light_pdf(*s, dir));
    if (s->has_env) { V3 c = env_lookup(s->env, dir); o.push_back(c.x); o.push_back(c.y); o.push_back(c.z); }
    double uu = rng.next(), vv = rng.next();
    for (const rtx_material& m : s->mats) { V3 a = albedo_at(*s, m, p, uu, vv); o.push_back(a.x); o.push_back(a.y); o.push_back(a.z); }
    double cosv = rng.next(), ratio = (i & 1) ? 1.5 : 1.0 / 1.5;
    o.push_back(schlick(cosv, ratio));
    double nx = rng.range(-1, 1), ny = rng.range(-1, 1), nz = rng.range(-1, 1);
    V3 nrm = normalize(V3{nx, ny, nz});
    V3 ud = normalize(dir);
    V3 rf = refract(ud, nrm, ratio), rl = reflect(ud, nrm);
    double v6[6] = {rf.x, rf.y, rf.z, rl.x, rl.y, rl.z};
    o.insert(o.end(), v6, v6 + 6);
  }
  *written = o.size();
  if (o.size() > cap) return RTX_ERR_INVALID_ARGUMENT;
  memcpy(out, o.data(), o.size() * 8);
  return RTX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// post-process (engine/renderer.cpp:102-198 tone mapping operators, :232-258 quantisation)
// ------------------------------------------------------------------------------------------
namespace {
double pp_clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }            // std::clamp(v, 0.0, 1.0)
double pp_u2_partial(double x) {                                                          // renderer.cpp:131-139
  const double A = 0.15, B = 0.50, C = 0.10, D = 0.20, E = 0.02, F = 0.30;
  return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
double pp_tone(double x, int mapper) {                                                    // renderer.cpp:155-184
  switch (mapper) {
    case 1: return x / (1.0 + x);                                                         // :106-108
    case 2: { const double wp2 = 4.0 * 4.0; return x * (1.0 + x / wp2) / (1.0 + x); }      // :114-117
    case 3: { const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14;             // :123-130
              return pp_clamp01((x * (a * x + b)) / (x * (c * x + d) + e)); }
    case 4: { const double curr = pp_u2_partial(x * 2.0);                                  // :144-150
              const double white_scale = 1.0 / pp_u2_partial(11.2);
              return curr * white_scale; }
    default: return pp_clamp01(x);
  }
}
}  // namespace

extern "C" void orc_set_photon_aim_seed(unsigned long long seed) { g_photon_aim_seed = seed; }

extern "C" void orc_pcg4d(uint32_t pixel, uint32_t sample, uint32_t block, uint32_t seed, uint32_t out[4]) { pcg4d(pixel, sample, block, seed, out); }

extern "C" int orc_postprocess(const float* film, int width, int height, int tone_mapper, double exposure, uint8_t* rgb8) {
  if (!film || !rgb8 || width <= 0 || height <= 0) return RTX_ERR_INVALID_ARGUMENT;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
      for (int ch = 0; ch < 3; ++ch) {
        const double c = (double)film[((size_t)y * width + x) * 3 + ch] * exposure;   // :189-193
        const double m = pp_tone(c, tone_mapper);
        const double g = std::sqrt(pp_clamp01(m));                                      // :241-243
        const int iv = (int)(255.999 * g);                                              // :246-248
        rgb8[((size_t)(height - 1 - y) * width + x) * 3 + ch] = (uint8_t)iv;            // :251-255
      }
  return RTX_OK;
}
