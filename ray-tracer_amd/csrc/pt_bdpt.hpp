// Bidirectional path tracing (ref: engine/bdpt.cpp, BDPTIntegrator::Li :71-128 with max_eye_depth = max_light_depth
// = max_depth as engine/renderer.cpp:304-313 sets them).
//
// One lane renders one sample at a time.  The light subpath is traced first and only its CONNECTIBLE vertices are kept
// (the strategies the reference skips, :102-103, need no storage), in a per-lane slice of a global buffer laid out
// [field][vertex][lane] so that a wave's accesses to one field of one vertex are contiguous.  The eye subpath is not
// stored at all: a vertex is connected to every kept light vertex (strategies t = index + 1, s = 0 .. num_light) right
// after its continuation has been sampled, which is when the members the weights read (wo, pdf_fwd) are final.  The
// strategies with t = 1 contribute nothing in the reference (:436-442) and are not visited.  Strategy contributions are
// added in the reference's order (t outer, s inner).
//
// Random numbers: eye vertex b draws block 1 + b, the light origin block 0x8001, light vertex b block 0x8002 + b;
// fields of a block: word0 = r1 | r2 of the light's emission direction, word1 = scatter r1 | r2 (or the unit vector,
// or the point on a quad / disk light), word2 = light choice | roulette, word3 = fresnel.
#pragma once

#include "pt_device.hpp"

namespace rtx {

template <class R> struct BdptConst { static constexpr R inv_pi = R(0.31830988618379067); };   // INV_PI, ref: engine/material.hpp:24

// DiskLight::sample_point (ref: engine/primitives.hpp:350-384)
template <class R> RTX_DEV V3<R> disk_point(const DDisk<R>& dl, R u, R v) {
  const R sx = R(2) * u - R(1), sy = R(2) * v - R(1);
  const V3<R> c = mk3(dl.c);
  if (sx == R(0) && sy == R(0)) return c;
  R rs, theta;
  if (abs_(sx) > abs_(sy)) { rs = sx; theta = (Consts<R>::pi / R(4)) * (sy / sx); }
  else { rs = sy; theta = (Consts<R>::pi / R(2)) - (Consts<R>::pi / R(4)) * (sx / sy); }
  R st, ct;
  sincos_(theta, &st, &ct);
  const R ar = dl.radius * rs;
  return c + (mk3(dl.tan) * (ar * ct) + mk3(dl.bit) * (ar * st));
}

static_assert(kBdptStoreFields == 18, "per kept light vertex: p n thr albedo wi (3 each), roughness, light-side MIS term, flags");

template <class R> struct BdptVertex {   // PathVertex (ref: engine/bdpt.hpp:54-110), the members the integrator reads
  V3<R> p, n, thr, albedo, wi, wo, Le;
  R pdf_fwd, roughness, ior;
  int type;       // 0 camera, 1 light, 2 surface
  int mat_type;
};

template <class R> RTX_DEV bool bdpt_connectible(const BdptVertex<R>& v) {   // ref: engine/bdpt.hpp:88-101
  if (v.type != 2) return true;
  return !(v.mat_type == MAT_METAL && v.roughness < R(0.001)) && v.mat_type != MAT_DIELECTRIC;
}

template <class R> RTX_DEV V3<R> bdpt_cosine_hemisphere(V3<R> normal, R r1, R r2) {   // ref: engine/bdpt.cpp:39-65
  const R cos_theta = sqrt_(r2), sin_theta = sqrt_(R(1) - r2);
  R sp, cp;
  sincos_turns(r1, &sp, &cp);
  const R x = cp * sin_theta, y = sp * sin_theta, z = cos_theta;
  const V3<R> w = normal;
  const V3<R> a = abs_(w.x) > R(0.9) ? V3<R>{R(0), R(1), R(0)} : V3<R>{R(1), R(0), R(0)};
  const V3<R> u = normalize(cross(a, w));
  const V3<R> v = cross(w, u);
  return normalize(V3<R>{u.x * x + v.x * y + w.x * z, u.y * x + v.y * y + w.y * z, u.z * x + v.z * y + w.z * z});
}

// evaluate_bsdf (ref: engine/bdpt.cpp:489-527) and bsdf_pdf (:529-560) on (normal, material kind, albedo, roughness)
template <class R> RTX_DEV V3<R> bdpt_eval(V3<R> n, int mat_type, V3<R> albedo, R roughness, V3<R> wi, V3<R> wo) {
  const R cos_o = dot(n, wo);
  if (mat_type == MAT_LAMBERTIAN) {
    if (cos_o <= R(0)) return {R(0), R(0), R(0)};
    return albedo * BdptConst<R>::inv_pi;
  }
  if (mat_type == MAT_METAL) {
    if (cos_o <= R(0) || roughness < R(0.001)) return {R(0), R(0), R(0)};
    const V3<R> reflected = reflect(-wi, n);
    const R cos_r = dot(reflected, wo);
    const R e = R(2) / (roughness * roughness) - R(2);
    if (cos_r > R(0)) return albedo * ((e + R(1)) * BdptConst<R>::inv_pi * R(0.5) * pow(cos_r, e));
  }
  return {R(0), R(0), R(0)};
}
template <class R> RTX_DEV R bdpt_pdf(V3<R> n, int mat_type, R roughness, V3<R> wi, V3<R> wo) {
  const R cos_o = dot(n, wo);
  if (mat_type == MAT_LAMBERTIAN) return cos_o <= R(0) ? R(0) : cos_o * BdptConst<R>::inv_pi;
  if (mat_type == MAT_METAL) {
    if (cos_o <= R(0) || roughness < R(0.001)) return R(0);
    const V3<R> reflected = reflect(-wi, n);
    const R cos_r = dot(reflected, wo);
    if (cos_r > R(0)) { const R e = R(2) / (roughness * roughness) - R(2); return (e + R(1)) * BdptConst<R>::inv_pi * R(0.5) * pow(cos_r, e); }
  }
  return R(0);
}

// sample_bsdf (ref: engine/bdpt.cpp:562-631)
template <class R> RTX_DEV bool bdpt_sample_bsdf(const BdptVertex<R>& v, const U4& rn, V3<R>& wo, R& pdf, V3<R>& f) {
  if (v.mat_type == MAT_LAMBERTIAN) {
    wo = bdpt_cosine_hemisphere(v.n, u16hi<R>(rn.y), u16lo<R>(rn.y));
    pdf = dot(v.n, wo) * BdptConst<R>::inv_pi;
    f = v.albedo * BdptConst<R>::inv_pi;
    return true;
  }
  if (v.mat_type == MAT_METAL) {
    const V3<R> reflected = reflect(-v.wi, v.n);
    if (v.roughness < R(0.001)) { wo = reflected; pdf = R(1); f = v.albedo; return true; }
    wo = normalize(reflected + unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y)) * v.roughness);
    if (dot(wo, v.n) <= R(0)) return false;
    const R cos_r = dot(reflected, wo);
    const R e = R(2) / (v.roughness * v.roughness) - R(2);
    pdf = (e + R(1)) * BdptConst<R>::inv_pi * R(0.5) * pow(max_(R(0), cos_r), e);
    f = v.albedo;
    return true;
  }
  R cos_i = dot(v.wi, v.n);
  const R eta = cos_i > R(0) ? R(1) / v.ior : v.ior;
  const V3<R> n = cos_i > R(0) ? v.n : -v.n;
  cos_i = abs_(cos_i);
  const R sin2 = eta * eta * (R(1) - cos_i * cos_i);
  R r0 = (R(1) - v.ior) / (R(1) + v.ior);
  r0 = r0 * r0;
  const R x = R(1) - cos_i;
  const R fresnel = r0 + (R(1) - r0) * pow(x, R(5));
  if (sin2 > R(1) || u16hi<R>(rn.w) < fresnel) wo = reflect(-v.wi, n);
  else wo = refract(-v.wi, n, eta);
  pdf = R(1); f = {R(1), R(1), R(1)};
  return true;
}

// The kept light vertices of one lane.
template <class R> struct BdptStore {
  R* base;            // + lane
  size_t lanes, cap;  // stride between vertices = lanes, between fields = cap * lanes
  RTX_DEV R& at(int field, int k) const { return base[((size_t)field * cap + (size_t)k) * lanes]; }
  RTX_DEV void put3(int field, int k, V3<R> v) const { at(field, k) = v.x; at(field + 1, k) = v.y; at(field + 2, k) = v.z; }
  RTX_DEV V3<R> get3(int field, int k) const { return {at(field, k), at(field + 1, k), at(field + 2, k)}; }
};

struct BdptTally { uint32_t closest, shadow, segments; };

template <class R> struct BdptEps { static constexpr R v = R(1e-6); };   // BDPT_EPSILON, ref: engine/bdpt.cpp:14

// Shared by both subpaths (ref: engine/bdpt.cpp:145-217 and :248-306): make the surface vertex of a hit.
template <class R> RTX_DEV BdptVertex<R> bdpt_surface_vertex(const DScene<R>& sc, const DMaterial<R>* mats, const Hit<R>& rec, const Ray<R>& r, V3<R> thr, bool eye) {
  const DMaterial<R>& m = mats[rec.mat];
  BdptVertex<R> v;
  v.type = 2; v.p = rec.p; v.n = rec.n; v.mat_type = m.type;
  v.albedo = albedo_at(sc, m, rec);
  v.Le = eye ? mk3(m.emission) : V3<R>{R(0), R(0), R(0)};
  v.roughness = m.fuzz; v.ior = m.ior; v.thr = thr;
  v.wi = -normalize(r.d);
  v.wo = {R(0), R(0), R(0)}; v.pdf_fwd = R(0);
  return v;
}
// Continuation of a subpath at v (sample, throughput, roulette).  Returns false when the path ends here; v.wo and
// v.pdf_fwd are set only when it goes on, as in the reference (:211-213).
template <class R> RTX_DEV bool bdpt_advance(BdptVertex<R>& v, const U4& rn, int bounce, V3<R>& thr, Ray<R>& r) {
  const R eps = BdptEps<R>::v;
  V3<R> wo, f;
  R pdf;
  if (!bdpt_sample_bsdf(v, rn, wo, pdf, f)) return false;
  if (pdf < eps) return false;
  const R cos_theta = abs_(dot(v.n, wo));
  thr = thr * (f * (cos_theta / pdf));
  if (bounce > 3) {
    const R q = max_(R(0.05), R(1) - max_(thr.x, max_(thr.y, thr.z)));
    if (u16lo<R>(rn.z) < q) return false;
    thr = thr * (R(1) / (R(1) - q));
  }
  v.wo = wo; v.pdf_fwd = pdf;
  r = {v.p, wo};
  return true;
}
// (pdf_rev / pdf_fwd)^2 of a connection vertex, the term mis_weight adds for it (ref: engine/bdpt.cpp:671-694)
template <class R> RTX_DEV R bdpt_mis_term(const BdptVertex<R>& v) {
  const R eps = BdptEps<R>::v;
  const R pr = bdpt_pdf(v.n, v.mat_type, v.roughness, v.wo, v.wi);
  if (v.pdf_fwd > eps && pr > eps) { const R ri = pr / v.pdf_fwd; return ri * ri; }
  return R(0);
}
template <class R> RTX_DEV void bdpt_keep(const BdptStore<R>& store, int& n_light, const BdptVertex<R>& v, bool origin) {
  const int k = n_light++;
  store.put3(0, k, v.p); store.put3(3, k, v.n); store.put3(6, k, v.thr); store.put3(9, k, v.albedo); store.put3(12, k, v.wi);
  store.at(15, k) = v.roughness;
  store.at(16, k) = bdpt_mis_term(v);
  store.at(17, k) = R(v.mat_type | (origin ? 16 : 0));
}

// The origin of the light subpath (ref: engine/bdpt.cpp sample_light_origin :311-413): the light vertex, the emission direction and
// its two pdfs.  false: the scene has no light.
template <class R> RTX_DEV bool bdpt_light_origin(const DScene<R>& sc, const DMaterial<R>* mats, PathKey key, BdptVertex<R>& lv, V3<R>& dir, R& pdf_pos, R& pdf_dir) {
  const U4 rn = rng_block(key.rng, key.pixel, key.seed, key.sample, 0x8001u);
  int n_sph = 0;
  for (int i = 0; i < sc.n_spheres; ++i) { const DMaterial<R>& m = mats[sc.spheres[i].mat]; n_sph += len2(mk3(m.emission)) > R(0.001) ? 1 : 0; }
  const int total = sc.n_points + n_sph + sc.n_quads + sc.n_disks;
  if (total <= 0) return false;
  const R choice_pdf = R(1) / R(total);
  int idx = (int)(u16hi<R>(rn.z) * R(total));
  idx = idx < total - 1 ? idx : total - 1;
  lv.type = 1; lv.mat_type = MAT_LAMBERTIAN; lv.albedo = {R(1), R(1), R(1)}; lv.thr = {R(1), R(1), R(1)};
  lv.roughness = R(0); lv.ior = R(1.5); lv.wi = {R(0), R(0), R(0)}; lv.wo = {R(0), R(0), R(0)};
  if (idx < sc.n_points) {
    lv.p = mk3(sc.points[idx].pos); lv.n = {R(0), R(1), R(0)}; lv.Le = mk3(sc.points[idx].I);
    pdf_pos = choice_pdf;
    dir = unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y));
    pdf_dir = R(1) / (R(4) * Consts<R>::pi);
  } else if (idx < sc.n_points + n_sph) {
    int want = idx - sc.n_points, which = 0;
    for (int i = 0; i < sc.n_spheres; ++i) {
      const DMaterial<R>& m = mats[sc.spheres[i].mat];
      if (len2(mk3(m.emission)) > R(0.001)) { if (want == 0) { which = i; break; } --want; }
    }
    const DSphere<R>& sp = sc.spheres[which];
    const V3<R> d = unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y));
    lv.p = mk3(sp.c) + d * sp.radius; lv.n = d; lv.Le = mk3(mats[sp.mat].emission);
    pdf_pos = choice_pdf / (R(4) * Consts<R>::pi * sp.radius * sp.radius);
    dir = bdpt_cosine_hemisphere(d, u16hi<R>(rn.x), u16lo<R>(rn.x));
    pdf_dir = dot(d, dir) * BdptConst<R>::inv_pi;
  } else if (idx < sc.n_points + n_sph + sc.n_quads) {
    const DQuad<R>& q = sc.quads[idx - sc.n_points - n_sph];
    lv.p = mk3(q.corner) + (mk3(q.eu) * u16hi<R>(rn.y) + mk3(q.ev) * u16lo<R>(rn.y));   // ref: engine/primitives.hpp:258-262
    lv.n = mk3(q.n); lv.Le = mk3(q.Le);
    pdf_pos = choice_pdf * (R(1) / q.area);
    dir = bdpt_cosine_hemisphere(lv.n, u16hi<R>(rn.x), u16lo<R>(rn.x));
    pdf_dir = dot(lv.n, dir) * BdptConst<R>::inv_pi;
  } else {
    const DDisk<R>& dl = sc.disks[idx - sc.n_points - n_sph - sc.n_quads];
    lv.p = disk_point(dl, u16hi<R>(rn.y), u16lo<R>(rn.y));
    lv.n = mk3(dl.n); lv.Le = mk3(dl.Le);
    pdf_pos = choice_pdf * (R(1) / dl.area);
    dir = bdpt_cosine_hemisphere(lv.n, u16hi<R>(rn.x), u16lo<R>(rn.x));
    pdf_dir = dot(lv.n, dir) * BdptConst<R>::inv_pi;
  }
  lv.pdf_fwd = pdf_pos;
  return true;
}

// One strategy's contribution, weighted and clamped (ref: engine/bdpt.cpp:87-125)
template <class R> RTX_DEV void bdpt_add_strategy(V3<R>& L, const DRenderParams& P, V3<R> Lp, R sum_ri) {
  const R clamp_max = R(P.clamp_max);
  V3<R> wl = Lp * (P.use_mis ? R(1) / sum_ri : R(1));
  if (clamp_max > R(0)) {
    const R lum = R(0.2126) * wl.x + R(0.7152) * wl.y + R(0.0722) * wl.z;
    if (lum > clamp_max) wl = wl * (clamp_max / lum);
  }
  L = L + wl;
}

// The eye vertex of a round of connections: what connect_paths reads of it.
template <class R> struct BdptEye {
  V3<R> p, n, thr, albedo, wi;
  R roughness, eye_term;
  int mat_type;
};
template <class R> RTX_DEV BdptEye<R> bdpt_eye_of(const BdptVertex<R>& ev) {
  return {ev.p, ev.n, ev.thr, ev.albedo, ev.wi, ev.roughness, bdpt_mis_term(ev), ev.mat_type};
}
// s = 0: the eye subpath found the light by itself
template <class R> RTX_DEV void bdpt_strategy_s0(V3<R>& L, const DRenderParams& P, const BdptVertex<R>& ev, R eye_term) {
  const bool ev_light = ev.type == 1 || len2(ev.Le) > R(0.001);
  bdpt_add_strategy(L, P, ev_light ? ev.thr * ev.Le : V3<R>{R(0), R(0), R(0)}, R(1) + eye_term + (ev_light ? R(1) : R(0)));
}
// The connection of the eye vertex with kept light vertex k (ref: connect_paths, engine/bdpt.cpp:415-487).  `occluded(ray, t_max)`
// answers visible() (:713-723).
template <class R, class Occ> RTX_DEV void bdpt_connect_one(const BdptStore<R>& store, int k, const BdptEye<R>& ev, V3<R> Le0, V3<R>& L, const DRenderParams& P, BdptTally& tl, Occ&& occluded) {
  const R eps = BdptEps<R>::v;
  const V3<R> lp = store.get3(0, k);
  const V3<R> d = lp - ev.p;
  const R dist_sq = len2(d);
  const R dist = sqrt_(dist_sq);
  // The reference asks visible() first (:449) and then finds most connections empty (a vertex facing away); the
  // product is the same with the visibility ray last, cast only for connections that carry something.
  const int flags = (int)store.at(17, k);
  const bool origin = (flags & 16) != 0;
  const V3<R> wo_eye = d * (R(1) / dist), wi_light = -wo_eye;
  const V3<R> f_eye = bdpt_eval(ev.n, ev.mat_type, ev.albedo, ev.roughness, ev.wi, wo_eye);
  if (len2(f_eye) < eps) return;
  const V3<R> ln = store.get3(3, k);
  V3<R> f_light;
  if (origin) f_light = Le0 * abs_(dot(ln, wi_light));
  else f_light = bdpt_eval(ln, flags & 15, store.get3(9, k), store.at(15, k), store.get3(12, k), wi_light);
  if (len2(f_light) < eps) return;
  R G = R(0);   // geometry_term, :725-738
  if (!(dist_sq < eps)) G = abs_(dot(ev.n, wo_eye)) * abs_(dot(ln, -wo_eye)) / dist_sq;
  if (!(dist < eps)) {   // visible(), :713-723
    ++tl.shadow;
    const Ray<R> sr = {ev.p, wo_eye};
    if (occluded(sr, dist - R(0.001))) return;
  }
  const R sum_ri = R(1) + ev.eye_term + store.at(16, k) + (origin ? R(0.5) : R(0));
  const V3<R> light_c = origin ? f_light : store.get3(6, k) * f_light;
  bdpt_add_strategy(L, P, ((ev.thr * f_eye) * light_c) * G, sum_ri);
}

// The light vertex an eye subpath ends with when it meets an area light (ref: engine/bdpt.cpp:153-166)
template <class R> RTX_DEV BdptVertex<R> bdpt_area_light_vertex(const DScene<R>& sc, const Hit<R>& rec, const Ray<R>& r, V3<R> thr) {
  BdptVertex<R> lv;
  lv.type = 1; lv.p = rec.p; lv.n = rec.n; lv.mat_type = MAT_LAMBERTIAN; lv.albedo = {R(1), R(1), R(1)};
  lv.Le = area_light_emission(sc, rec.mat); lv.thr = thr; lv.roughness = R(0); lv.ior = R(1.5);
  lv.wi = -normalize(r.d); lv.wo = {R(0), R(0), R(0)}; lv.pdf_fwd = R(0);
  return lv;
}

// One sample: the radiance BDPTIntegrator::Li returns for the camera ray `ray`.
template <class R, int MODE>
RTX_DEV V3<R> bdpt_sample(const DScene<R>& sc, const DMaterial<R>* mats, const DRenderParams& P, PathKey key, Ray<R> ray, LaneStack st, const BdptStore<R>& store,
                          BdptTally& tl) {
  const R eps = BdptEps<R>::v;
  int n_light = 0;         // kept (connectible) light vertices
  V3<R> Le0 = {R(0), R(0), R(0)};

  // ---- light subpath (ref: engine/bdpt.cpp:222-309) ----
  {
    BdptVertex<R> lv;
    V3<R> dir;
    R pdf_pos, pdf_dir;
    if (bdpt_light_origin(sc, mats, key, lv, dir, pdf_pos, pdf_dir)) {
      Le0 = lv.Le;
      bdpt_keep(store, n_light, lv, true);
      if (!(pdf_dir < eps)) {
        V3<R> thr = lv.Le * (abs_(dot(lv.n, dir)) / (pdf_pos * pdf_dir));
        Ray<R> r = {lv.p, dir};
        for (int bounce = 0; bounce < P.max_depth; ++bounce) {
          Hit<R> rec;
          ++tl.closest;
          if (!trace_closest<R, MODE>(sc, r, R(0.001), R(1e9), rec, st)) break;
          if (rec.mat < 0) break;   // area lights end the light subpath without a vertex (:255-258)
          ++tl.segments;
          BdptVertex<R> v = bdpt_surface_vertex(sc, mats, rec, r, thr, false);
          const U4 rb = rng_block(key.rng, key.pixel, key.seed, key.sample, 0x8002u + (uint32_t)bounce);
          const bool go = bdpt_advance(v, rb, bounce, thr, r);
          if (bdpt_connectible(v)) bdpt_keep(store, n_light, v, false);
          if (!go) break;
        }
      }
    }
  }

  // ---- eye subpath (ref: engine/bdpt.cpp:130-220) with its connections (Li :87-125, connect_paths :415-487) ----
  V3<R> L = {R(0), R(0), R(0)};
  auto connect_all = [&](const BdptVertex<R>& v) {
    const BdptEye<R> ev = bdpt_eye_of(v);
    bdpt_strategy_s0(L, P, v, ev.eye_term);
    for (int k = 0; k < n_light; ++k)
      bdpt_connect_one(store, k, ev, Le0, L, P, tl, [&](const Ray<R>& sr, R t_max) { return trace_occluded<R, MODE>(sc, sr, R(0.001), t_max, st); });
  };
  {
    V3<R> thr = {R(1), R(1), R(1)};
    Ray<R> r = ray;
    for (int bounce = 0; bounce < P.max_depth; ++bounce) {
      Hit<R> rec;
      ++tl.closest;
      if (!trace_closest<R, MODE>(sc, r, R(0.001), R(1e9), rec, st)) break;
      ++tl.segments;
      if (rec.mat < 0) {   // an area light ends the eye subpath with a light vertex (:153-166)
        connect_all(bdpt_area_light_vertex(sc, rec, r, thr));
        break;
      }
      BdptVertex<R> v = bdpt_surface_vertex(sc, mats, rec, r, thr, true);
      const U4 rb = rng_block(key.rng, key.pixel, key.seed, key.sample, 1u + (uint32_t)bounce);
      const bool go = bdpt_advance(v, rb, bounce, thr, r);
      if (bdpt_connectible(v)) connect_all(v);
      if (!go) break;
    }
  }
  return L;
}

// One lane per (pixel, chunk) work item at a time, grid-stride; chunk sums go to `partial` as in the path tracer.
#ifndef RTX_BDPT_MIN_WAVES
#define RTX_BDPT_MIN_WAVES 6   // single precision; measured on caustics_test 800x800x128: 2 (free): 306 ms, 3: 232, 4: 236, 5: 263, 6: 218, 8: 274
#endif
template <class R, int MODE, bool LDS_MATS>
__global__ __launch_bounds__(kBlock, (sizeof(R) == 4 ? RTX_BDPT_MIN_WAVES : 1)) void render_bdpt_kernel(DScene<R> sc, DCamera<R> cam, DRenderParams P, R* __restrict__ partial, DCounters* __restrict__ counters,
                                                             int stack_first_int, R* __restrict__ light_store, unsigned long long store_cap) {
  const DMaterial<R>* mats = sc.mats;
  if constexpr (LDS_MATS) mats = stage_materials(sc);
  const uint32_t npix = (uint32_t)P.rows * (uint32_t)P.width;
  const uint32_t n_items = npix * (uint32_t)P.n_chunks;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st = lane_stack(stack_first_int);
  BdptStore<R> store = {light_store + lane, (size_t)stride, (size_t)store_cap};
  BdptTally tl = {0u, 0u, 0u};
  uint32_t n_samples = 0;
  for (uint32_t item = lane; item < n_items; item += stride) {
    const uint32_t chunk = item / npix;
    const uint32_t pix = item - chunk * npix;
    const uint32_t row_local = pix / (uint32_t)P.width;
    const uint32_t x = pix - row_local * (uint32_t)P.width;
    const uint32_t y = (uint32_t)P.shard_index + row_local * (uint32_t)P.shard_count;
    PathKey key = {y * (uint32_t)P.width + x, P.seed, 0u, (uint32_t)P.rng};
    const int s0 = (int)chunk * P.chunk_spp;
    const int s1 = s0 + P.chunk_spp < P.spp ? s0 + P.chunk_spp : P.spp;
    V3<R> sum = {R(0), R(0), R(0)};
    for (int s = s0; s < s1; ++s) {
      key.sample = (uint32_t)s;
      PathState<R> ps;
      path_begin(ps, cam, P, (int)x, (int)y, key);
      sum = sum + bdpt_sample<R, MODE>(sc, mats, P, key, ps.ray, st, store, tl);
      ++n_samples;
    }
    R* o = partial + (size_t)item * 3u;
    o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
  }
  uint32_t v[4] = {n_samples, tl.closest, tl.shadow, tl.segments};
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += (uint32_t)__shfl_xor((int)v[k], off, 64);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counters->samples, (unsigned long long)v[0]);
    atomicAdd(&counters->closest, (unsigned long long)v[1]);
    atomicAdd(&counters->shadow, (unsigned long long)v[2]);
    atomicAdd(&counters->segments, (unsigned long long)v[3]);
  }
}

}  // namespace rtx
