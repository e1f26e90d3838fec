// Manifold next event estimation on the device: MNEE::evaluate and its solver
// (ref: engine/spectral/mnee.hpp:56-763, called from Renderer::ray_color_path at
// engine/renderer.cpp:716-732).  Every routine follows its reference counterpart statement by
// statement; closest-hit queries go through `trace`, a functor
//     bool trace(const Ray<R>&, R t_min, R t_max, Hit<R>&)
// with the semantics of Scene::hit.  Instantiated for double only: the Newton solver's
// convergence threshold (1e-6 on unit vectors) and finite-difference step (1e-5) leave no
// room for single precision.
#pragma once

#include "pt_device.hpp"

namespace rtx {

template <class R> struct MVertex { V3<R> p, n; R eta; int mat; };   // ManifoldVertex, mnee.hpp:32-39 (members the solver reads)
constexpr int kMneeMaxIter = 20, kMneeMaxDepth = 4;                  // mnee.hpp:59-60

template <class R> RTX_DEV void mnee_frame(V3<R> n, V3<R>& t, V3<R>& b) {   // ref: mnee.hpp:299-304
  V3<R> up = (abs_(n.y) < R(0.9)) ? V3<R>{R(0), R(1), R(0)} : V3<R>{R(1), R(0), R(0)};
  t = normalize(cross(up, n));
  b = cross(n, t);
}

template <class R> RTX_DEV void mnee_constraint(V3<R> prev, V3<R> cur, V3<R> next, R eta, V3<R> t, V3<R> b, R& cu, R& cv) {   // ref: mnee.hpp:328-347
  V3<R> wi = normalize(prev - cur), wo = normalize(next - cur);
  R wi_u = dot(wi, t), wi_v = dot(wi, b), wo_u = dot(wo, t), wo_v = dot(wo, b);
  cu = eta * wi_u + wo_u; cv = eta * wi_v + wo_v;
}

template <class R, class Trace> RTX_DEV bool mnee_reproject(V3<R>& pos, V3<R>& n, int expected, Trace& trace) {   // ref: mnee.hpp:401-437
  for (int d = 0; d < 2; ++d) {
    Hit<R> rec;
    Ray<R> sr = {pos, d == 0 ? n : -n};
    if (trace(sr, R(0), R(0.1), rec)) {
      if (rec.mat == expected) { pos = rec.p; n = rec.n; return true; }
    }
  }
  Hit<R> rec;
  Ray<R> down = {pos + n * R(0.05), -n};
  if (trace(down, R(0), R(0.2), rec)) {
    if (rec.mat == expected) { pos = rec.p; n = rec.n; return true; }
  }
  return false;
}

// ref: build_path, mnee.hpp:538-617 — throughput of the chain, then a visibility walk along it
template <class R, class Trace> RTX_DEV bool mnee_build_path(V3<R> sp, V3<R> lp, const MVertex<R>* ch, int n, R& throughput, Trace& trace) {
  throughput = R(1);
  for (int i = 0; i < n; ++i) {
    V3<R> prev = (i == 0) ? sp : ch[i - 1].p, next = (i == n - 1) ? lp : ch[i + 1].p, cur = ch[i].p;
    V3<R> wi = normalize(prev - cur), wo = normalize(next - cur);
    R eta = ch[i].eta;
    R cos_i = abs_(dot(wi, ch[i].n)), cos_t = abs_(dot(wo, ch[i].n));
    R r0 = (R(1) - eta) / (R(1) + eta);
    r0 = r0 * r0;
    R fr = r0 + (R(1) - r0) * pow(R(1) - cos_i, R(5));
    R ft = R(1) - fr;
    R geometry = (eta * eta * cos_t) / cos_i;
    throughput *= ft * geometry;
    if (throughput < R(1e-10)) return false;
  }
  V3<R> current = sp;
  for (int i = 0; i <= n; ++i) {
    V3<R> target = (i == n) ? lp : ch[i].p;
    V3<R> dir = normalize(target - current);
    R dist = len(target - current);
    Hit<R> rec;
    Ray<R> cr = {current, dir};
    if (trace(cr, R(0.001), dist - R(0.001), rec)) {
      if (i < n) {
        R hd = len(rec.p - ch[i].p);
        if (hd > R(0.01)) return false;
      } else return false;
    }
    if (i < n) current = ch[i].p;
  }
  return true;
}

// ref: solve_manifold, mnee.hpp:442-533 — Gauss-Seidel Newton on the per-vertex refraction
// constraints with finite-difference 2x2 Jacobians (compute_jacobian_block :354-382, solve_2x2 :387-396)
template <class R, class Trace> RTX_DEV bool mnee_solve(V3<R> sp, V3<R> lp, MVertex<R>* ch, int n, R& throughput, Trace& trace) {
  V3<R> tg[kMneeMaxDepth], bt[kMneeMaxDepth];
  for (int i = 0; i < n; ++i) mnee_frame(ch[i].n, tg[i], bt[i]);
  for (int iter = 0; iter < kMneeMaxIter; ++iter) {
    R max_c = R(0), cu[kMneeMaxDepth], cv[kMneeMaxDepth];
    for (int i = 0; i < n; ++i) {
      V3<R> prev = (i == 0) ? sp : ch[i - 1].p, next = (i == n - 1) ? lp : ch[i + 1].p;
      mnee_constraint(prev, ch[i].p, next, ch[i].eta, tg[i], bt[i], cu[i], cv[i]);
      R mag = sqrt_(cu[i] * cu[i] + cv[i] * cv[i]);
      max_c = max_(max_c, mag);
    }
    if (max_c < R(1e-6)) return mnee_build_path(sp, lp, ch, n, throughput, trace);
    for (int i = 0; i < n; ++i) {
      V3<R> prev = (i == 0) ? sp : ch[i - 1].p, next = (i == n - 1) ? lp : ch[i + 1].p;
      const R h = R(1e-5);
      R c0u, c0v, cuu, cuv, cvu, cvv;
      mnee_constraint(prev, ch[i].p, next, ch[i].eta, tg[i], bt[i], c0u, c0v);
      mnee_constraint(prev, ch[i].p + tg[i] * h, next, ch[i].eta, tg[i], bt[i], cuu, cuv);
      mnee_constraint(prev, ch[i].p + bt[i] * h, next, ch[i].eta, tg[i], bt[i], cvu, cvv);
      R J00 = (cuu - c0u) / h, J01 = (cvu - c0u) / h, J10 = (cuv - c0v) / h, J11 = (cvv - c0v) / h;
      R b0 = -cu[i], b1 = -cv[i], st0, st1;
      R det = J00 * J11 - J01 * J10;
      if (abs_(det) < R(1e-10)) { st0 = R(-0.5) * cu[i]; st1 = R(-0.5) * cv[i]; }
      else { R inv = R(1) / det; st0 = inv * (J11 * b0 - J01 * b1); st1 = inv * (-J10 * b0 + J00 * b1); }
      R mag = sqrt_(st0 * st0 + st1 * st1);
      if (mag > R(0.1)) { R sc = R(0.1) / mag; st0 *= sc; st1 *= sc; }
      ch[i].p = ch[i].p + (tg[i] * st0 + bt[i] * st1);
      mnee_reproject(ch[i].p, ch[i].n, ch[i].mat, trace);
      mnee_frame(ch[i].n, tg[i], bt[i]);
    }
  }
  return false;
}

// ref: connect_to_light_straight, mnee.hpp:94-143
template <class R, class Trace> RTX_DEV bool mnee_straight(const DMaterial<R>* mats, V3<R> sp, V3<R> lp, R& throughput, Trace& trace) {
  V3<R> to_light = lp - sp;
  R dist = len(to_light);
  V3<R> dir = to_light * (R(1) / dist);
  MVertex<R> ch[kMneeMaxDepth];
  int n = 0;
  Ray<R> probe = {sp, dir};
  V3<R> cur = sp;
  for (int i = 0; i < kMneeMaxDepth; ++i) {
    Hit<R> rec;
    R max_t = len(lp - cur) - R(1e-6);
    if (!trace(probe, R(0.001), max_t, rec)) break;
    // a quad / disk light in the way has a negative id (the reference indexes its material
    // array with it: undefined behaviour); defined here as "not a dielectric"
    if (rec.mat < 0 || mats[rec.mat].type != MAT_DIELECTRIC) return false;
    ch[n].p = rec.p; ch[n].n = rec.n;
    ch[n].eta = rec.front ? (R(1) / mats[rec.mat].ior) : mats[rec.mat].ior;
    ch[n].mat = rec.mat;
    ++n;
    cur = rec.p;
    probe = {cur + dir * R(0.001), dir};
  }
  if (n == 0) return false;
  return mnee_solve(sp, lp, ch, n, throughput, trace);
}

// Sphere::hit on one sphere: point and face-forwarded normal (ref: engine/sphere.hpp:71-105)
template <class R> RTX_DEV bool mnee_sphere_hit(const DSphere<R>& s, const Ray<R>& r, V3<R>& p, V3<R>& n) {
  R t;
  if (!sphere_t(s, r, R(0.001), R(1e9), t)) return false;
  p = at(r, t);
  V3<R> outward = (p - mk3(s.c)) * (R(1) / s.radius);
  n = dot(r.d, outward) < R(0) ? outward : -outward;
  return true;
}

// ref: try_sphere_seed_path, mnee.hpp:194-295
template <class R, class Trace> RTX_DEV bool mnee_sphere_seed(const DSphere<R>& sph, const DMaterial<R>& m, V3<R> sp, V3<R> lp, R& throughput, Trace& trace) {
  V3<R> c = mk3(sph.c);
  V3<R> pts[2], nrm[2];
  for (int k = 0; k < 2; ++k) {
    V3<R> origin = k == 0 ? sp : lp;
    V3<R> dir = k == 0 ? normalize(lp - sp) : normalize(sp - lp);
    Ray<R> rr = {origin, dir};
    if (!mnee_sphere_hit(sph, rr, pts[k], nrm[k])) {
      V3<R> to_center = c - origin;
      R proj = dot(to_center, dir);
      if (proj < R(0)) return false;
      V3<R> closest = origin + dir * proj;
      R da = len(c - closest);
      if (da >= sph.radius) pts[k] = c + normalize(closest - c) * sph.radius;
      else { R off = sqrt_(sph.radius * sph.radius - da * da); pts[k] = origin + dir * (proj - off); }
      nrm[k] = normalize(pts[k] - c);
    }
  }
  MVertex<R> ch[kMneeMaxDepth];
  ch[0].p = pts[0]; ch[0].n = nrm[0]; ch[0].eta = R(1) / m.ior; ch[0].mat = sph.mat;
  ch[1].p = pts[1]; ch[1].n = -nrm[1]; ch[1].eta = m.ior; ch[1].mat = sph.mat;
  return mnee_solve(sp, lp, ch, 2, throughput, trace);
}

// ref: connect_to_light, mnee.hpp:73-88, with connect_to_light_seeded, :152-189
template <class R, class Trace> RTX_DEV bool mnee_connect(const DScene<R>& sc, const DMaterial<R>* mats, V3<R> sp, V3<R> lp, R& throughput, Trace& trace) {
  if (mnee_straight(mats, sp, lp, throughput, trace)) return true;
  for (int i = 0; i < sc.n_spheres; ++i) {
    const DSphere<R>& sph = sc.spheres[i];
    const DMaterial<R>& m = mats[sph.mat];
    if (m.type != MAT_DIELECTRIC) continue;
    V3<R> to_sphere = mk3(sph.c) - sp, to_light = lp - sp;
    R ds = len(to_sphere), dl = len(to_light);
    if (ds > dl + sph.radius) continue;
    if (ds < sph.radius) continue;
    if (dot(to_sphere * (R(1) / ds), to_light * (R(1) / dl)) < R(0)) continue;
    if (mnee_sphere_seed(sph, m, sp, lp, throughput, trace)) return true;
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

// ref: MNEE::evaluate, mnee.hpp:632-763.  Returns whether any light contributed.
template <class R, class Trace> RTX_DEV bool mnee_evaluate(const DScene<R>& sc, const DMaterial<R>* mats, V3<R> hp, PathKey key, int bounce, V3<R>& contribution, Trace& trace) {
  contribution = {R(0), R(0), R(0)};
  bool any = false;
  int j = 0;
  const R two_pi = R(2) * R(3.14159265358979323846);
  for (int i = 0; i < sc.n_quads; ++i) {
    const DQuad<R>& q = sc.quads[i];
    R u, v;
    mnee_pair(key, bounce, j++, u, v);
    V3<R> lp = mk3(q.corner) + (mk3(q.eu) * u + mk3(q.ev) * v);   // ref: QuadLight::sample_point, primitives.hpp:258-262
    V3<R> ln = mk3(q.n);
    R w;
    if (mnee_connect(sc, mats, hp, lp, w, trace)) {
      V3<R> to_light = lp - hp;
      R d2 = len2(to_light);
      R cl = abs_(dot(ln, normalize(-to_light)));
      R lpdf = d2 / (q.area * cl);
      contribution = contribution + mk3(q.Le) * (w * cl / (d2 * lpdf));
      any = true;
    }
  }
  for (int i = 0; i < sc.n_disks; ++i) {
    const DDisk<R>& dl = sc.disks[i];
    R u, v;
    mnee_pair(key, bounce, j++, u, v);
    V3<R> lp = mk3(dl.c);   // ref: DiskLight::sample_point, primitives.hpp:350-384
    R sx = R(2) * u - R(1), sy = R(2) * v - R(1);
    if (!(sx == R(0) && sy == R(0))) {
      R rs, theta;
      if (abs_(sx) > abs_(sy)) { rs = sx; theta = (Consts<R>::pi / R(4)) * (sy / sx); }
      else { rs = sy; theta = (Consts<R>::pi / R(2)) - (Consts<R>::pi / R(4)) * (sx / sy); }
      R st, ct;
      sincos_(theta, &st, &ct);
      R ar = dl.radius * rs;
      lp = lp + (mk3(dl.tan) * (ar * ct) + mk3(dl.bit) * (ar * st));
    }
    V3<R> ln = mk3(dl.n);
    R w;
    if (mnee_connect(sc, mats, hp, lp, w, trace)) {
      V3<R> to_light = lp - hp;
      R d2 = len2(to_light);
      R cl = abs_(dot(ln, normalize(-to_light)));
      R lpdf = d2 / (dl.area * cl);
      contribution = contribution + mk3(dl.Le) * (w * cl / (d2 * lpdf));
      any = true;
    }
  }
  for (int e = 0; e < sc.n_emissive; ++e) {   // spheres in index order whose material is emissive
    const DSphere<R>& sph = sc.spheres[sc.emissive[e]];
    V3<R> to_sphere = mk3(sph.c) - hp;
    R dc = len(to_sphere);
    V3<R> w3 = to_sphere * (R(1) / dc);
    R ctm = sqrt_(R(1) - (sph.radius * sph.radius) / (dc * dc));
    if (ctm < R(0)) ctm = R(0);
    R r1, r2;
    mnee_pair(key, bounce, j++, r1, r2);
    R ct = R(1) + r1 * (ctm - R(1));
    R sn = sqrt_(R(1) - ct * ct);
    R phi = two_pi * r2;
    V3<R> uv = (abs_(w3.x) > R(0.9)) ? V3<R>{R(0), R(1), R(0)} : V3<R>{R(1), R(0), R(0)};
    uv = normalize(cross(w3, uv));
    V3<R> vv = cross(w3, uv);
    R sphi, cphi;
    sincos_(phi, &sphi, &cphi);
    V3<R> sd = uv * (sn * cphi) + (vv * (sn * sphi) + w3 * ct);
    R t = dc * ct - sqrt_(sph.radius * sph.radius - dc * dc * (R(1) - ct * ct));
    V3<R> lp = hp + sd * t;
    V3<R> ln = normalize(lp - mk3(sph.c));
    R w;
    if (mnee_connect(sc, mats, hp, lp, w, trace)) {
      R solid = two_pi * (R(1) - ctm);
      R pdf = R(1) / solid;
      R cal = abs_(dot(ln, -sd));
      contribution = contribution + mk3(mats[sph.mat].emission) * (w * cal / pdf);
      any = true;
    }
  }
  return any;
}

}  // namespace rtx
