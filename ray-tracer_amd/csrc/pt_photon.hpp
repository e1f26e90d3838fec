// PathPhoton mode (ref: engine/renderer.cpp:344-372 photon pass, :844-976 ray_color_path_with_caustics;
// engine/photon_integrator.hpp:93-293 caustic photons; engine/photon_map.hpp:104-186, :222-305 kd-tree and estimate).
//
// photon_trace_kernel (pt_photon_kernel.hpp): one lane per emitted photon (the reference emits them one after the other from one generator;
// here photon i draws from counter blocks keyed (i, seed, sample 0xfffffffe): block 0 = light point u | v, aim point
// u | v, target choice; block 1 + depth = roulette, fresnel).  A photon leaves at most one record (the caustic pass stops
// at the first Lambertian surface), written to slot i, so the map is the same whatever the launch geometry.  The host
// compacts the records in photon order and builds the kd-tree as the reference does (median by std::nth_element,
// axis = depth % 3); photon_query (pt_photon_wave.hpp) searches it, one shading point at a time per wave.
#pragma once

#include "pt_device.hpp"
#include "pt_photon_wave.hpp"

namespace rtx {

// Renderer::ray_color_path_with_caustics (ref: engine/renderer.cpp:844-976), one vertex per call like path_vertex.
// `occluded(ray, t_max)` answers the shadow query; `gather(point, normal, weight)` takes the request for the caustic term
// weight * estimate(point, normal), which the caller adds to ps.acc (photon_gather_wave).
template <class R, class OccFn, class GatherFn>
RTX_DEV uint32_t photon_vertex(PathState<R>& ps, const bool hit, Hit<R>& rec, const DScene<R>& sc, const DMaterial<R>* mats, const DRenderParams& P, PathKey key,
                               OccFn&& occluded, GatherFn&& gather) {
  const bool nee = P.use_nee != 0;
  const int bounce = (int)(ps.bounce_flags >> 4);
  const bool was_spec = ps.bounce_flags & F_SPEC;
  if (bounce >= P.max_depth) { ps.bounce_flags = 0; return 0u; }
  if (!hit) { ps.acc = ps.acc + ps.thr * background(sc, P, ps.ray); ps.bounce_flags = 0; return 0u; }
  uint32_t did = STEP_SHADED;
  if (rec.mat < 0) {
    if (bounce == 0 || was_spec || !nee) ps.acc = ps.acc + ps.thr * area_light_emission(sc, rec.mat);
    ps.bounce_flags = 0;
    return did;
  }
  const DMaterial<R>& m = mats[rec.mat];
  apply_normal_map(sc, m, rec);
  const V3<R> surf = albedo_at(sc, m, rec);
  if (bounce == 0 || was_spec || !nee) ps.acc = ps.acc + ps.thr * mk3(m.emission);
  const U4 rn = rng_block(key.rng, key.pixel, key.seed, key.sample, 1u + (uint32_t)bounce);
  V3<R> sdir, atten = surf;
  uint32_t flags = F_ALIVE;
  if (m.type == MAT_LAMBERTIAN) {
    if (P.n_photons > 0) gather(rec.p, rec.n, ps.thr * surf);
    if (nee) {
      const LightPick<R> ls = sample_light(sc, mats, rec.p, u16hi<R>(rn.z), u16hi<R>(rn.x), u16lo<R>(rn.x));
      if (ls.valid) {
        const V3<R> to_light = ls.pos - rec.p;
        const R ld = len(to_light);
        const V3<R> ldir = to_light * (R(1) / ld);
        const R cs = dot(rec.n, ldir);
        if (cs > R(0)) {
          const bool is_env = ls.dist > R(1e5);
          did |= STEP_SHADOW;
          const Ray<R> sr = {rec.p, ldir};
          if (!occluded(sr, is_env ? R(1e9) : ld - R(0.001))) {
            const R cl = abs_(dot(ls.n, -ldir));
            const R falloff = is_env ? R(1) : R(1) / (ld * ld);
            ps.acc = ps.acc + ps.thr * ((surf * ls.Le) * (cs * falloff * cl / (ls.pdf * Consts<R>::pi)));
          }
        }
      }
    }
    sdir = rec.n + unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y));
    if (len2(sdir) < R(1e-8)) sdir = rec.n;
    sdir = normalize(sdir);
  } else if (m.type == MAT_METAL) {
    flags |= F_SPEC;
    V3<R> refl = reflect(normalize(ps.ray.d), rec.n);
    if (m.fuzz > R(0)) { refl = refl + unit_vector(u16hi<R>(rn.y), u16lo<R>(rn.y)) * m.fuzz; if (!(m.fuzz < R(0.1))) flags &= ~F_SPEC; }
    sdir = refl;   // a direction below the surface goes on: both `break`s at :937-938 leave the switch only
  } else {
    flags |= F_SPEC;
    atten = {R(1), R(1), R(1)};
    const R ratio = rec.front ? (R(1) / m.ior) : m.ior;
    const V3<R> ud = normalize(ps.ray.d);
    const R ct = min_(dot(-ud, rec.n), R(1));
    const R sn = sqrt_(R(1) - ct * ct);
    if (ratio * sn > R(1) || schlick(ct, ratio) > u16hi<R>(rn.w)) sdir = reflect(ud, rec.n);
    else sdir = refract(ud, rec.n, ratio);
  }
  ps.thr = ps.thr * atten;
  if (bounce > 3) {
    const R p = max_(ps.thr.x, max_(ps.thr.y, ps.thr.z));
    if (p < R(0.01) || u16lo<R>(rn.z) > p) { ps.bounce_flags = 0; return did; }
    ps.thr = ps.thr * (R(1) / p);
  }
  ps.ray.o = rec.p;
  ps.ray.d = sdir;
  ps.bounce_flags = (bounce + 1 >= P.max_depth) ? 0u : ((((uint32_t)(bounce + 1)) << 4) | flags);
  return did;
}

}  // namespace rtx
