// PathPhoton mode (ref: engine/renderer.cpp:344-372 photon pass, :844-976 ray_color_path_with_caustics;
// engine/photon_integrator.hpp:93-293 caustic photons; engine/photon_map.hpp:104-186, :222-305 kd-tree and estimate).
//
// photon_trace_kernel (pt_photon_kernel.hpp): one lane per emitted photon (the reference emits them one after the other from one generator;
// here photon i draws from counter blocks keyed (i, seed, sample 0xfffffffe): block 0 = light point u | v, aim point
// u | v, target choice; block 1 + depth = roulette, fresnel).  A photon leaves at most one record (the caustic pass stops
// at the first Lambertian surface), written to slot i, so the map is the same whatever the launch geometry.  The host
// compacts the records in photon order and builds the kd-tree as the reference does (median by std::nth_element,
// axis = depth % 3); photon_estimate walks it with the reference's visiting order and pruning rule, keeping the k
// nearest in a per-lane array.
#pragma once

#include "pt_device.hpp"

namespace rtx {

constexpr int kPhotonStack = 64;        // kd-tree walk: two frames per level

// PhotonMap::estimate_irradiance (ref: engine/photon_map.hpp:150-186) over find_nearest / search_tree (:117-140, :250-305).
template <class R> RTX_DEV V3<R> photon_estimate(const DRenderParams& P, V3<R> pos, V3<R> normal) {
  const DPhotonNode<R>* nodes = (const DPhotonNode<R>*)P.ph_nodes;
  const DPhotonData<R>* data = (const DPhotonData<R>*)P.ph_data;
  const int k = P.gather_k;
  const R radius = R(P.gather_radius);
  const R max_d2 = radius * radius;
  R best_d2[kPhotonMaxGather];
  int best_id[kPhotonMaxGather];
  int count = 0;
  constexpr int top = 0;    // the farthest kept photon is the root of the heap
  int f_start[kPhotonStack], f_end[kPhotonStack];
  R f_delta2[kPhotonStack];   // < 0: visit unconditionally; else the far side of a split, visited if it can still hold a closer photon
  int sp = 0;
  f_start[0] = 0; f_end[0] = P.n_photons; f_delta2[0] = R(-1); sp = 1;
  while (sp > 0) {
    --sp;
    const int start = f_start[sp], end = f_end[sp];
    const R cond = f_delta2[sp];
    if (start >= end) continue;
    // The reference enters the far side of a split whenever fewer than k photons are kept, however far away it is
    // (photon_map.hpp:295, :301), i.e. it walks the whole tree for every point with fewer than k photons in reach.  No
    // photon beyond the split plane can be nearer than the plane, so a far side at max_d2 or more cannot add anything:
    // skipping it gives the same k nearest.
    if (cond >= R(0) && (cond >= max_d2 || !(count < k || cond < best_d2[top]))) continue;
    const int mid = start + (end - start) / 2;
    const DPhotonNode<R> q = nodes[mid];
    const V3<R> diff = V3<R>{q.p[0], q.p[1], q.p[2]} - pos;
    const R d2 = len2(diff);
    R rad = max_d2;
    if (count >= k) rad = min_(rad, best_d2[top]);
    if (d2 < rad) {   // best_d2 / best_id form a binary max-heap on the squared distance, like the reference's priority_queue
      int j;
      if (count < k) {
        j = count++;
        while (j > 0) {   // sift up
          const int parent = (j - 1) >> 1;
          if (!(best_d2[parent] < d2)) break;
          best_d2[j] = best_d2[parent]; best_id[j] = best_id[parent];
          j = parent;
        }
      } else {          // d2 < best_d2[0] holds (rad is at most that): replace the farthest, sift down
        j = 0;
        for (;;) {
          int child = 2 * j + 1;
          if (child >= k) break;
          if (child + 1 < k && best_d2[child + 1] > best_d2[child]) ++child;
          if (!(best_d2[child] > d2)) break;
          best_d2[j] = best_d2[child]; best_id[j] = best_id[child];
          j = child;
        }
      }
      best_d2[j] = d2; best_id[j] = mid;
    }
    const R pa = q.axis == 0 ? pos.x : q.axis == 1 ? pos.y : pos.z;
    const R delta = pa - q.p[q.axis];
    const R delta2 = delta * delta;
    if (sp + 2 > kPhotonStack) continue;   // deeper than any tree of < 2^31 photons
    // near side first: it is pushed last
    if (delta < R(0)) {
      f_start[sp] = mid + 1; f_end[sp] = end; f_delta2[sp] = delta2; ++sp;
      f_start[sp] = start; f_end[sp] = mid; f_delta2[sp] = R(-1); ++sp;
    } else {
      f_start[sp] = start; f_end[sp] = mid; f_delta2[sp] = delta2; ++sp;
      f_start[sp] = mid + 1; f_end[sp] = end; f_delta2[sp] = R(-1); ++sp;
    }
  }
  if (count == 0) return {R(0), R(0), R(0)};
  R r2 = best_d2[0];
  for (int j = 1; j < count; ++j) r2 = max_(r2, best_d2[j]);
  if (r2 < R(1e-10)) r2 = R(1e-10);
  V3<R> flux = {R(0), R(0), R(0)};
  for (int j = 0; j < count; ++j) {
    const DPhotonData<R> d = data[best_id[j]];
    const R cos_theta = -(d.dir[0] * normal.x + d.dir[1] * normal.y + d.dir[2] * normal.z);
    if (cos_theta <= R(0)) continue;
    const R w = R(1) - sqrt_(best_d2[j] / r2);
    flux = flux + V3<R>{d.power[0], d.power[1], d.power[2]} * w;
  }
  return flux * (R(3) / (Consts<R>::pi * r2));
}

// Renderer::ray_color_path_with_caustics (ref: engine/renderer.cpp:844-976), one vertex per call like path_vertex.
// `occluded(ray, t_max)` answers the shadow query.
template <class R, class OccFn>
RTX_DEV uint32_t photon_vertex(PathState<R>& ps, const bool hit, Hit<R>& rec, const DScene<R>& sc, const DMaterial<R>* mats, const DRenderParams& P, PathKey key,
                               OccFn&& occluded) {
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
    if (P.n_photons > 0) ps.acc = ps.acc + ps.thr * (surf * photon_estimate<R>(P, rec.p, rec.n));
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
