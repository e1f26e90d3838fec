// Manifold next event estimation on the device: MNEE::evaluate and its solver
// (ref: engine/spectral/mnee.hpp:56-763, called from Renderer::ray_color_path at
// engine/renderer.cpp:716-732).  Every routine follows its reference counterpart statement by
// statement; closest-hit queries go through `trace`, a functor
//     bool trace(const Ray<R>&, R t_min, R t_max, Hit<R>&)
// with the semantics of Scene::hit.  Two type parameters: S, the type the manifold walk computes in, and R, the type of the
// scene and of the rays `trace` takes.  S is always double: the Newton solver's convergence threshold (1e-6 on unit
// vectors) and finite-difference step (1e-5) leave no room for single precision.  With R = double this is the reference's
// arithmetic (the parity mode).  With R = float the closest-hit queries - nearly all of the work - run in single precision
// on the production structures, and every point they return on a sphere or a plane is put back onto that surface in double
// (mnee_surface_point), so that the solver sees a consistent surface: a vertex that is off by single-precision rounding
// ALONG the surface is still a valid manifold vertex, one that is off ACROSS it is not.
#pragma once

#include "pt_device.hpp"

namespace rtx {

template <class R> struct MVertex { V3<R> p, n; R eta; int mat; };   // ManifoldVertex, mnee.hpp:32-39 (members the solver reads)
constexpr int kMneeMaxIter = 20, kMneeMaxDepth = 4;                  // mnee.hpp:59-60

// The chain of a connection: up to kMneeMaxDepth vertices, addressed with a run-time index through selects so that the
// vertices stay in registers (an array indexed at run time would live in scratch memory, which is where this solver spent
// its time).
template <class R> struct MChain {
  MVertex<R> v0, v1, v2, v3;
  RTX_DEV MVertex<R> get(int i) const { return i == 0 ? v0 : i == 1 ? v1 : i == 2 ? v2 : v3; }
  RTX_DEV V3<R> p(int i) const { return i == 0 ? v0.p : i == 1 ? v1.p : i == 2 ? v2.p : v3.p; }
  RTX_DEV void set(int i, const MVertex<R>& v) { if (i == 0) v0 = v; else if (i == 1) v1 = v; else if (i == 2) v2 = v; else v3 = v; }
};
// The same for chains of at most two vertices (every seed chain through a sphere, and the straight chain through one sphere):
// half the registers while the solver runs.
template <class R> struct MChain2 {
  MVertex<R> v0, v1;
  RTX_DEV MVertex<R> get(int i) const { return i == 0 ? v0 : v1; }
  RTX_DEV V3<R> p(int i) const { return i == 0 ? v0.p : v1.p; }
  RTX_DEV void set(int i, const MVertex<R>& v) { if (i == 0) v0 = v; else v1 = v; }
};

template <class A, class B> struct mnee_same { static constexpr bool value = false; };
template <class A> struct mnee_same<A, A> { static constexpr bool value = true; };
template <class T, class U> RTX_DEV V3<T> v3_to(V3<U> v) { return {T(v.x), T(v.y), T(v.z)}; }

// Closest hit of the S-typed ray (o, d) through the R-typed scene; on a hit, the point and the face-forwarded normal in S
// (rec.front is set to match).  With R = float the single-precision query only decides WHICH primitive is hit first; for a
// sphere or a plane the hit itself is then computed again in S from the S-typed ray, with the reference's root choice
// (engine/sphere.hpp:71-92, engine/primitives.hpp:35-48).  Without that, every reprojection would move its vertex by the
// rounding of a single-precision origin (4e-7 at these scenes' coordinates), which is the size of the solver's convergence
// threshold: three quarters of the connections the double-precision walk finds were lost that way.
template <class S, class R, class Trace> RTX_DEV bool mnee_trace(const DScene<R>& sc, V3<S> o, V3<S> d, S t_min, S t_max, Hit<R>& rec, V3<S>& p, V3<S>& n, Trace& trace) {
  Ray<R> r = {v3_to<R>(o), v3_to<R>(d)};
  if (!trace(r, R(t_min), R(t_max), rec)) return false;
  p = v3_to<S>(rec.p); n = v3_to<S>(rec.n);
  if constexpr (!mnee_same<S, R>::value) {
    if (rec.type == PRIM_SPHERE) {
      const DSphere<R>& sp = sc.spheres[rec.idx];
      const V3<S> c = {S(sp.c[0]), S(sp.c[1]), S(sp.c[2])};
      const S rad = S(sp.radius);
      const V3<S> oc = o - c;
      const S a = len2(d), hb = dot(oc, d), cc = len2(oc) - rad * rad, disc = hb * hb - a * cc;
      V3<S> out;
      if (disc >= S(0)) {
        // the root in range, near one first; when rounding leaves neither in range (a ray that starts on the sphere), the root
        // next to the single-precision distance
        const S sq = sqrt_(disc), t0 = (-hb - sq) / a, t1 = (-hb + sq) / a, tf = S(rec.t);
        const bool in0 = t0 >= t_min && t0 <= t_max, in1 = t1 >= t_min && t1 <= t_max;
        const S t = in0 ? t0 : in1 ? t1 : (abs_(t0 - tf) < abs_(t1 - tf) ? t0 : t1);
        p = o + d * t; out = (p - c) * (S(1) / rad);
      } else { out = normalize(p - c); p = c + out * rad; }   // a grazing hit the two precisions disagree on: the single-precision point, put onto the sphere
      rec.front = dot(d, out) < S(0);
      n = rec.front ? out : -out;
    } else if (rec.type == PRIM_PLANE) {
      const DPlane<R>& pl = sc.planes[rec.idx];
      const V3<S> pn = {S(pl.n[0]), S(pl.n[1]), S(pl.n[2])}, p0 = {S(pl.p[0]), S(pl.p[1]), S(pl.p[2])};
      const S denom = dot(pn, d);
      const S t = dot(p0 - o, pn) / denom;
      if (abs_(denom) >= S(1e-8)) p = o + d * t;
      else p = p - pn * (dot(p - p0, pn) / len2(pn));
      rec.front = denom < S(0);
      n = rec.front ? pn : -pn;
    }
  }
  return true;
}

// normalize() of the solver's inner loop.  FAST (the single-precision production mode, where the walk's doubles need not equal
// the reference's bit for bit): 1 / |v| from the hardware's reciprocal square root estimate and two Newton steps, full double
// precision for the lengths that occur here (1e-3 to 1e2: no rescaling), a third of the instructions of sqrt followed by a
// division.  Otherwise the reference's v / sqrt(v . v).
template <bool FAST, class R> RTX_DEV V3<R> mnee_normalize(V3<R> v) {
  if constexpr (FAST && std::is_same<R, double>::value) {
    const double l2 = len2(v);
    double y = __builtin_amdgcn_rsq(l2);
    y = fma(0.5 * y, fma(-l2 * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-l2 * y, y, 1.0), y);
    return l2 > 0.0 ? v * y : v;
  } else {
    return normalize(v);
  }
}

template <bool FAST = false, class R> RTX_DEV void mnee_frame(V3<R> n, V3<R>& t, V3<R>& b) {   // ref: mnee.hpp:299-304
  V3<R> up = (abs_(n.y) < R(0.9)) ? V3<R>{R(0), R(1), R(0)} : V3<R>{R(1), R(0), R(0)};
  t = mnee_normalize<FAST>(cross(up, n));
  b = cross(n, t);
}

template <bool FAST = false, class R> RTX_DEV void mnee_constraint(V3<R> prev, V3<R> cur, V3<R> next, R eta, V3<R> t, V3<R> b, R& cu, R& cv) {   // ref: mnee.hpp:328-347
  V3<R> wi = mnee_normalize<FAST>(prev - cur), wo = mnee_normalize<FAST>(next - cur);
  R wi_u = dot(wi, t), wi_v = dot(wi, b), wo_u = dot(wo, t), wo_v = dot(wo, b);
  cu = eta * wi_u + wo_u; cv = eta * wi_v + wo_v;
}

template <class S, class R, class Trace> RTX_DEV bool mnee_reproject(const DScene<R>& sc, V3<S>& pos, V3<S>& n, int expected, Trace& trace) {   // ref: mnee.hpp:401-437
  Hit<R> rec;
  V3<S> hp, hn;
  for (int d = 0; d < 2; ++d) {
    if (d == 0) { RTX_SITE(18) } else { RTX_SITE(19) }
    // The first probe leaves the surface it should find in nineteen of twenty cases (a Newton step moves the vertex along the
    // tangent plane, i.e. off a convex surface on its outer side): ask first whether it can meet that material at all.
    if (d == 0 && !trace.may_hit_material(Ray<R>{v3_to<R>(pos), v3_to<R>(n)}, R(0), R(0.1), expected)) continue;
    if (d == 0) { RTX_SITE(26) }
    if (mnee_trace(sc, pos, d == 0 ? n : -n, S(0), S(0.1), rec, hp, hn, trace)) {
      if (rec.mat == expected) { pos = hp; n = hn; return true; }
    }
  }
  RTX_SITE(20)
  if (mnee_trace(sc, pos + n * S(0.05), -n, S(0), S(0.2), rec, hp, hn, trace)) {
    if (rec.mat == expected) { pos = hp; n = hn; return true; }
  }
  return false;
}

// ref: build_path, mnee.hpp:538-617 — throughput of the chain, then a visibility walk along it
template <class R, class SR, class Trace, class Chain> RTX_DEV bool mnee_build_path(const DScene<SR>& sc, V3<R> sp, V3<R> lp, const Chain& ch, int n, R& throughput, Trace& trace) {
  throughput = R(1);
  for (int i = 0; i < n; ++i) {
    const MVertex<R> vi = ch.get(i);
    V3<R> prev = (i == 0) ? sp : ch.p(i - 1), next = (i == n - 1) ? lp : ch.p(i + 1), cur = vi.p;
    V3<R> wi = normalize(prev - cur), wo = normalize(next - cur);
    R eta = vi.eta;
    R cos_i = abs_(dot(wi, vi.n)), cos_t = abs_(dot(wo, vi.n));
    R r0 = (R(1) - eta) / (R(1) + eta);
    r0 = r0 * r0;
    R fr;
    if constexpr (mnee_same<R, SR>::value) fr = r0 + (R(1) - r0) * pow(R(1) - cos_i, R(5));   // std::pow, as the reference
    else { const R x = R(1) - cos_i, x2 = x * x; fr = r0 + (R(1) - r0) * (x2 * x2 * x); }      // (approximate pow() is NaN for a base rounded below zero)
    R ft = R(1) - fr;
    R geometry = (eta * eta * cos_t) / cos_i;
    throughput *= ft * geometry;
    // A chain of NaNs (the cone sample of an emissive sphere is NaN for a point that rounding put inside that sphere) fails in
    // the reference at the visibility walk below, whose plane test takes a NaN distance for a hit; here it fails at once.
    if constexpr (mnee_same<R, SR>::value) { if (throughput < R(1e-10)) return false; }
    else { if (!(throughput >= R(1e-10))) return false; }
  }
  V3<R> current = sp;
  for (int i = 0; i <= n; ++i) {
    V3<R> target = (i == n) ? lp : ch.p(i);
    V3<R> dir = normalize(target - current);
    R dist = len(target - current);
    Hit<SR> rec;
    V3<R> hp, hn;
    if (mnee_trace(sc, current, dir, R(0.001), dist - R(0.001), rec, hp, hn, trace)) {
      if (i < n) {
        R hd = len(hp - ch.p(i));
        if (hd > R(0.01)) return false;
      } else return false;
    }
    if (i < n) current = ch.p(i);
  }
  return true;
}

// ref: solve_manifold, mnee.hpp:442-533 — Gauss-Seidel Newton on the per-vertex refraction
// constraints with finite-difference 2x2 Jacobians (compute_jacobian_block :354-382, solve_2x2 :387-396)
template <class R, class SR, class Trace, class Chain> RTX_DEV bool mnee_solve(const DScene<SR>& sc, V3<R> sp, V3<R> lp, Chain& ch, int n, R& throughput, Trace& trace) {
  // (the reference keeps every vertex's tangent frame in an array and renews it after the vertex's re-projection; a frame is a
  // function of the vertex's normal alone, so it is computed where it is used: the same values, a dozen registers fewer per vertex)
  constexpr bool kFast = !mnee_same<R, SR>::value;
  for (int iter = 0; iter < kMneeMaxIter; ++iter) {
    RTX_SITE(16)
    R max_c = R(0);
    R c_u0 = R(0), c_u1 = R(0), c_u2 = R(0), c_u3 = R(0), c_v0 = R(0), c_v1 = R(0), c_v2 = R(0), c_v3 = R(0);   // cu[i], cv[i]
    for (int i = 0; i < n; ++i) {
      const MVertex<R> vi = ch.get(i);
      V3<R> prev = (i == 0) ? sp : ch.p(i - 1), next = (i == n - 1) ? lp : ch.p(i + 1);
      V3<R> tg, bt;
      mnee_frame<kFast>(vi.n, tg, bt);
      R cu, cv;
      mnee_constraint<kFast>(prev, vi.p, next, vi.eta, tg, bt, cu, cv);
      if (i == 0) { c_u0 = cu; c_v0 = cv; } else if (i == 1) { c_u1 = cu; c_v1 = cv; } else if (i == 2) { c_u2 = cu; c_v2 = cv; } else { c_u3 = cu; c_v3 = cv; }
      R mag = sqrt_(cu * cu + cv * cv);
      max_c = max_(max_c, mag);
    }
    if (max_c < R(1e-6)) { RTX_SITE(21) return mnee_build_path(sc, sp, lp, ch, n, throughput, trace); }
    for (int i = 0; i < n; ++i) {
      RTX_SITE(17)
      MVertex<R> vi = ch.get(i);
      V3<R> prev = (i == 0) ? sp : ch.p(i - 1), next = (i == n - 1) ? lp : ch.p(i + 1);
      V3<R> tg, bt;
      mnee_frame<kFast>(vi.n, tg, bt);
      const R cu_i = i == 0 ? c_u0 : i == 1 ? c_u1 : i == 2 ? c_u2 : c_u3, cv_i = i == 0 ? c_v0 : i == 1 ? c_v1 : i == 2 ? c_v2 : c_v3;
      const R h = R(1e-5);
      R c0u, c0v, cuu, cuv, cvu, cvv;
      // (the first vertex's neighbours have not moved since the residuals above were taken: the same operands, the same values)
      if (i == 0) { c0u = c_u0; c0v = c_v0; }
      else mnee_constraint<kFast>(prev, vi.p, next, vi.eta, tg, bt, c0u, c0v);
      mnee_constraint<kFast>(prev, vi.p + tg * h, next, vi.eta, tg, bt, cuu, cuv);
      mnee_constraint<kFast>(prev, vi.p + bt * h, next, vi.eta, tg, bt, cvu, cvv);
      R J00 = (cuu - c0u) / h, J01 = (cvu - c0u) / h, J10 = (cuv - c0v) / h, J11 = (cvv - c0v) / h;
      R b0 = -cu_i, b1 = -cv_i, st0, st1;
      R det = J00 * J11 - J01 * J10;
      if (abs_(det) < R(1e-10)) { st0 = R(-0.5) * cu_i; st1 = R(-0.5) * cv_i; }
      else { R inv = R(1) / det; st0 = inv * (J11 * b0 - J01 * b1); st1 = inv * (-J10 * b0 + J00 * b1); }
      R mag = sqrt_(st0 * st0 + st1 * st1);
      if (mag > R(0.1)) { R sc2 = R(0.1) / mag; st0 *= sc2; st1 *= sc2; }
      vi.p = vi.p + (tg * st0 + bt * st1);
      mnee_reproject(sc, vi.p, vi.n, vi.mat, trace);
      ch.set(i, vi);
    }
  }
  return false;
}

// ref: connect_to_light_straight, mnee.hpp:94-143 — the chain of dielectric crossings on the straight line to the light.
// Returns its length; 0 = no chain (nothing in the way, or something that is not a dielectric).
template <class R, class SR, class Trace> RTX_DEV int mnee_straight_chain(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<R> sp, V3<R> lp, MChain<R>& ch, Trace& trace) {
  V3<R> to_light = lp - sp;
  R dist = len(to_light);
  V3<R> dir = to_light * (R(1) / dist);
  int n = 0;
  V3<R> probe_o = sp;
  V3<R> cur = sp;
  for (int i = 0; i < kMneeMaxDepth; ++i) {
    Hit<SR> rec;
    V3<R> hp, hn;
    R max_t = len(lp - cur) - R(1e-6);
    RTX_SITE(25)
    if (!mnee_trace(sc, probe_o, dir, R(0.001), max_t, rec, hp, hn, trace)) break;
    // a quad / disk light in the way has a negative id (the reference indexes its material
    // array with it: undefined behaviour); defined here as "not a dielectric"
    if (rec.mat < 0 || mats[rec.mat].type != MAT_DIELECTRIC) return 0;
    MVertex<R> v;
    v.p = hp; v.n = hn;
    v.eta = rec.front ? (R(1) / R(mats[rec.mat].ior)) : R(mats[rec.mat].ior);
    v.mat = rec.mat;
    ch.set(n, v);
    ++n;
    cur = hp;
    probe_o = cur + dir * R(0.001);
  }
  return n;
}

// Sphere::hit on one sphere: point and face-forwarded normal (ref: engine/sphere.hpp:71-105)
template <class R, class SR> RTX_DEV bool mnee_sphere_hit(const DSphere<SR>& s0, const Ray<R>& r, V3<R>& p, V3<R>& n) {
  DSphere<R> s;
  s.c[0] = R(s0.c[0]); s.c[1] = R(s0.c[1]); s.c[2] = R(s0.c[2]); s.radius = R(s0.radius); s.mat = s0.mat; s.pad = 0;
  R t;
  if (!sphere_t(s, r, R(0.001), R(1e9), t)) return false;
  p = at(r, t);
  V3<R> outward = (p - mk3(s.c)) * (R(1) / s.radius);
  n = dot(r.d, outward) < R(0) ? outward : -outward;
  return true;
}

// ref: try_sphere_seed_path, mnee.hpp:194-295 — the two-vertex seed chain through one dielectric sphere.  Returns 2, or 0.
template <class R, class SR> RTX_DEV int mnee_sphere_seed_chain(const DSphere<SR>& sph0, const DMaterial<SR>& m0, V3<R> sp, V3<R> lp, MChain<R>& ch) {
  struct { R radius; int mat; } sph = {R(sph0.radius), sph0.mat};
  struct { R ior; } m = {R(m0.ior)};
  V3<R> c = {R(sph0.c[0]), R(sph0.c[1]), R(sph0.c[2])};
  V3<R> pts[2], nrm[2];
  for (int k = 0; k < 2; ++k) {
    V3<R> origin = k == 0 ? sp : lp;
    V3<R> dir = k == 0 ? normalize(lp - sp) : normalize(sp - lp);
    Ray<R> rr = {origin, dir};
    if (!mnee_sphere_hit(sph0, rr, pts[k], nrm[k])) {
      V3<R> to_center = c - origin;
      R proj = dot(to_center, dir);
      if (proj < R(0)) return 0;
      V3<R> closest = origin + dir * proj;
      R da = len(c - closest);
      if (da >= sph.radius) pts[k] = c + normalize(closest - c) * sph.radius;
      else { R off = sqrt_(sph.radius * sph.radius - da * da); pts[k] = origin + dir * (proj - off); }
      nrm[k] = normalize(pts[k] - c);
    }
  }
  ch.v0.p = pts[0]; ch.v0.n = nrm[0]; ch.v0.eta = R(1) / m.ior; ch.v0.mat = sph.mat;
  ch.v1.p = pts[1]; ch.v1.n = -nrm[1]; ch.v1.eta = m.ior; ch.v1.mat = sph.mat;
  return 2;
}

// ref: connect_to_light, mnee.hpp:73-88, with connect_to_light_seeded, :152-189: the straight chain first, then a seed chain
// through every dielectric sphere that lies towards the light; the first chain the solver brings to convergence wins.  (One
// loop over the attempts, so that the solver and its closest-hit queries are instantiated once.)
// The chain attempt i starts from: i = -1 the straight chain, i >= 0 the seed chain through sphere i.  Returns its length, 0 = the
// attempt does not exist (nothing dielectric in the way / the sphere is no dielectric or does not lie towards the light).
template <class R, class SR, class Trace> RTX_DEV int mnee_attempt_chain(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<R> sp, V3<R> lp, int i, MChain<R>& ch, Trace& trace) {
  if (i < 0) return mnee_straight_chain(sc, mats, sp, lp, ch, trace);
  const DSphere<SR>& sph = sc.spheres[i];
  const DMaterial<SR>& m = mats[sph.mat];
  if (m.type != MAT_DIELECTRIC) return 0;
  V3<R> to_sphere = V3<R>{R(sph.c[0]), R(sph.c[1]), R(sph.c[2])} - sp, to_light = lp - sp;
  R ds = len(to_sphere), dl = len(to_light);
  if (ds > dl + R(sph.radius)) return 0;
  if (ds < R(sph.radius)) return 0;
  if (dot(to_sphere * (R(1) / ds), to_light * (R(1) / dl)) < R(0)) return 0;
  return mnee_sphere_seed_chain(sph, m, sp, lp, ch);
}

template <class R, class SR, class Trace> RTX_DEV bool mnee_connect(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<R> sp, V3<R> lp, R& throughput, Trace& trace) {
  for (int i = -1; i < sc.n_spheres; ++i) {
    MChain<R> ch{};
    const int n = mnee_attempt_chain(sc, mats, sp, lp, i, ch, trace);
    if (n == 0) continue;
    RTX_SITE(22)
    bool ok;
    if (n <= 2) {   // the common case, with half the chain's registers
      MChain2<R> c2 = {ch.v0, ch.v1};
      ok = mnee_solve(sc, sp, lp, c2, n, throughput, trace);
    } else {
      ok = mnee_solve(sc, sp, lp, ch, n, throughput, trace);
    }
    if (ok) { RTX_SITE(24) return true; }
  }
  return false;
}

// The two uniforms of the j-th light of a vertex: extra Philox block (1 + bounce) | ((1 + j / 4) << 16),
// word j % 4, high | low 16-bit field.
template <class R> RTX_DEV void mnee_pair(PathKey key, int bounce, int j, R& a, R& b) {
  const U4 o = rng_block(key.rng, key.pixel, key.seed, key.sample, (1u + (uint32_t)bounce) | ((1u + (uint32_t)j / 4u) << 16));
  const uint32_t w = (j & 3) == 0 ? o.x : (j & 3) == 1 ? o.y : (j & 3) == 2 ? o.z : o.w;
  a = u16hi<R>(w); b = u16lo<R>(w);
}

// One light's sample point as MNEE::evaluate draws it (ref: mnee.hpp:632-763): the lights in the reference's order - quads,
// disks, then the spheres (in index order) whose material is emissive -, light j with the j-th pair of uniforms.
template <class R> struct MneeLight { V3<R> lp, ln, Le, sd; R area, ctm; bool sphere; };

template <class SR> RTX_DEV int mnee_light_count(const DScene<SR>& sc) { return sc.n_quads + sc.n_disks + sc.n_emissive; }

template <class SR> RTX_DEV MneeLight<double> mnee_light(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<double> hp, PathKey key, int bounce, int j) {
  using R = double;   // the type the walk computes in; SR is the scene's
  auto mk3 = [](const SR* a) { return V3<R>{R(a[0]), R(a[1]), R(a[2])}; };
  const R two_pi = R(2) * R(3.14159265358979323846);
  const int nq = sc.n_quads, nd = sc.n_disks;
  MneeLight<R> L;
  R u, v;
  mnee_pair(key, bounce, j, u, v);
  L.sd = {R(0), R(0), R(0)};
  L.area = R(0); L.ctm = R(0);
  L.sphere = j >= nq + nd;
  if (j < nq) {
    const DQuad<SR>& q = sc.quads[j];
    L.lp = mk3(q.corner) + (mk3(q.eu) * u + mk3(q.ev) * v);   // ref: QuadLight::sample_point, primitives.hpp:258-262
    L.ln = mk3(q.n); L.Le = mk3(q.Le); L.area = R(q.area);
  } else if (!L.sphere) {
    const DDisk<SR>& dl = sc.disks[j - nq];
    L.lp = mk3(dl.c);   // ref: DiskLight::sample_point, primitives.hpp:350-384
    R sx = R(2) * u - R(1), sy = R(2) * v - R(1);
    if (!(sx == R(0) && sy == R(0))) {
      R rs, theta;
      if (abs_(sx) > abs_(sy)) { rs = sx; theta = (Consts<R>::pi / R(4)) * (sy / sx); }
      else { rs = sy; theta = (Consts<R>::pi / R(2)) - (Consts<R>::pi / R(4)) * (sx / sy); }
      R st, ct;
      sincos_(theta, &st, &ct);
      R ar = R(dl.radius) * rs;
      L.lp = L.lp + (mk3(dl.tan) * (ar * ct) + mk3(dl.bit) * (ar * st));
    }
    L.ln = mk3(dl.n); L.Le = mk3(dl.Le); L.area = R(dl.area);
  } else {
    const DSphere<SR>& sph = sc.spheres[sc.emissive[j - nq - nd]];
    V3<R> to_sphere = mk3(sph.c) - hp;
    R dc = len(to_sphere);
    V3<R> w3 = to_sphere * (R(1) / dc);
    const R radius = R(sph.radius);
    L.ctm = sqrt_(R(1) - (radius * radius) / (dc * dc));
    if (L.ctm < R(0)) L.ctm = R(0);
    const R r1 = u, r2 = v;
    R ct = R(1) + r1 * (L.ctm - R(1));
    R sn = sqrt_(R(1) - ct * ct);
    R phi = two_pi * r2;
    V3<R> uv = (abs_(w3.x) > R(0.9)) ? V3<R>{R(0), R(1), R(0)} : V3<R>{R(1), R(0), R(0)};
    uv = normalize(cross(w3, uv));
    V3<R> vv = cross(w3, uv);
    R sphi, cphi;
    sincos_(phi, &sphi, &cphi);
    L.sd = uv * (sn * cphi) + (vv * (sn * sphi) + w3 * ct);
    R t = dc * ct - sqrt_(radius * radius - dc * dc * (R(1) - ct * ct));
    L.lp = hp + L.sd * t;
    L.ln = normalize(L.lp - mk3(sph.c));
    L.Le = mk3(mats[sph.mat].emission);
  }
  return L;
}

// What a converged connection of throughput w to that sample contributes.
RTX_DEV V3<double> mnee_light_contribution(const MneeLight<double>& L, V3<double> hp, double w) {
  using R = double;
  const R two_pi = R(2) * R(3.14159265358979323846);
  if (L.sphere) {
    R solid = two_pi * (R(1) - L.ctm);
    R pdf = R(1) / solid;
    R cal = abs_(dot(L.ln, -L.sd));
    return L.Le * (w * cal / pdf);
  }
  V3<R> to_light = L.lp - hp;
  R d2 = len2(to_light);
  R cl = abs_(dot(L.ln, normalize(-to_light)));
  R lpdf = d2 / (L.area * cl);
  return L.Le * (w * cl / (d2 * lpdf));
}

// One light of MNEE::evaluate: sample, connect, weigh.
template <class SR, class Trace> RTX_DEV bool mnee_evaluate_light(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<double> hp, PathKey key, int bounce, int j, V3<double>& c, Trace& trace) {
  const MneeLight<double> L = mnee_light(sc, mats, hp, key, bounce, j);
  double w;
  if (!mnee_connect(sc, mats, hp, L.lp, w, trace)) return false;
  c = mnee_light_contribution(L, hp, w);
  return true;
}

// ref: MNEE::evaluate, mnee.hpp:632-763.  Returns whether any light contributed.
template <class SR, class Trace> RTX_DEV bool mnee_evaluate(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<SR> hp0, PathKey key, int bounce, V3<SR>& contribution_out, Trace& trace) {
  using R = double;
  const V3<R> hp = v3_to<R>(hp0);
  V3<R> contribution = {R(0), R(0), R(0)};
  bool any = false;
  const int total = mnee_light_count(sc);
  for (int j = 0; j < total; ++j) {   // one loop, so that mnee_connect is instantiated once
    V3<R> c;
    if (mnee_evaluate_light(sc, mats, hp, key, bounce, j, c, trace)) { contribution = contribution + c; any = true; }
  }
  contribution_out = v3_to<SR>(contribution);
  return any;
}

// ---- MNEE::evaluate cut at its solves, for a kernel that batches them (render_fused_mnee_batched_kernel) ----
// The attempts of a vertex in the order mnee_evaluate makes them: light j = 0 .. lights - 1, for each the straight chain (i = -1)
// and the seed chain through every sphere (i = 0 ..).  mnee_next_attempt moves (j, i) forward to the first attempt that has a
// chain (false: none left); mnee_run_attempt runs the solver on that attempt and, if it converges, returns the light's
// contribution - after which the caller goes on with (j + 1, -1), as connect_to_light returns at the first chain that converges,
// or with (j, i + 1) otherwise.  The chain is built twice (to see that it exists, and again for the solve; `quiet` answers the
// second build's queries without counting them), which costs a thousandth of a solve and keeps 60 registers free between the two.
template <class SR, class Trace> RTX_DEV bool mnee_next_attempt(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<double> hp, PathKey key, int bounce, int& j, int& i, Trace& trace) {
  const int total = mnee_light_count(sc);
  for (; j < total; ++j, i = -1) {
    const MneeLight<double> L = mnee_light(sc, mats, hp, key, bounce, j);
    for (; i < sc.n_spheres; ++i) {
      MChain<double> ch{};
      if (mnee_attempt_chain(sc, mats, hp, L.lp, i, ch, trace) > 0) return true;
    }
  }
  return false;
}
template <class SR, class Trace> RTX_DEV bool mnee_run_attempt(const DScene<SR>& sc, const DMaterial<SR>* mats, V3<double> hp, PathKey key, int bounce, int j, int i, V3<double>& c, Trace& trace,
                                                            Trace& quiet) {
  const MneeLight<double> L = mnee_light(sc, mats, hp, key, bounce, j);
  MChain<double> ch{};
  const int n = mnee_attempt_chain(sc, mats, hp, L.lp, i, ch, quiet);
  double w;
  if (n == 0) return false;
  bool ok;
  if (n <= 2) {   // the common case, with half the chain's registers (a wave runs the long form only for the lanes that need it)
    MChain2<double> c2 = {ch.v0, ch.v1};
    ok = mnee_solve(sc, hp, L.lp, c2, n, w, trace);
  } else {
    ok = mnee_solve(sc, hp, L.lp, ch, n, w, trace);
  }
  if (!ok) return false;
  c = mnee_light_contribution(L, hp, w);
  return true;
}

}  // namespace rtx
