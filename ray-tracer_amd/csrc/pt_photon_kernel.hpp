// The caustic photon pass of the PathPhoton mode as a kernel; see pt_photon.hpp for the scheme.
#pragma once

#include "pt_kernels.hpp"

namespace rtx {

// trace_photon with caustic_only = true (ref: engine/photon_integrator.hpp:202-293), iteratively
template <class R, int MODE>
__global__ __launch_bounds__(kBlock) void photon_trace_kernel(DScene<R> sc, DRenderParams P, const DPhotonLight* __restrict__ lights, int n_lights,
                                                              const int32_t* __restrict__ targets, int n_targets, uint32_t n_photons, R* __restrict__ out,
                                                              DCounters* __restrict__ counters, int stack_first_int) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st = lane_stack(stack_first_int);
  uint32_t n_closest = 0;
  if (i < n_photons) {
    const size_t n = n_photons;
    out[9 * n + i] = R(0);   // no record
    int li = 0;
    while (li + 1 < n_lights && i >= lights[li].first + lights[li].count) ++li;
    const DPhotonLight L = lights[li];
    const DSphere<R>& ls = sc.spheres[L.sphere];
    const U4 b0 = rng_block((uint32_t)P.rng, i, P.seed, 0xfffffffeu, 0u);
    // emit_caustic_photon_from_sphere, :164-200
    int ti = (int)(u16hi<R>(b0.z) * R(n_targets));
    ti = targets[(unsigned)ti % (unsigned)n_targets];
    const DSphere<R>& tg = sc.spheres[ti];
    const R two_pi = R(2) * Consts<R>::pi;
    R s_t, c_t, s_p, c_p;
    sincos_(two_pi * u16hi<R>(b0.x), &s_t, &c_t);
    R phi = acos(R(2) * u16lo<R>(b0.x) - R(1));
    sincos_(phi, &s_p, &c_p);
    const V3<R> local = {s_p * c_t, s_p * s_t, c_p};
    const V3<R> origin = mk3(ls.c) + local * (ls.radius * R(1.001));
    sincos_(two_pi * u16hi<R>(b0.y), &s_t, &c_t);   // sample_sphere, :457-471
    phi = acos(R(2) * u16lo<R>(b0.y) - R(1));
    sincos_(phi, &s_p, &c_p);
    const V3<R> aim = {s_p * c_t, s_p * s_t, c_p};
    const V3<R> target_point = mk3(tg.c) + aim * (tg.radius * R(0.5));
    const V3<R> to_target = target_point - origin;
    const R dist_sq = len2(to_target);
    const R solid_angle = Consts<R>::pi * tg.radius * tg.radius / dist_sq;
    V3<R> power = V3<R>{R(L.power[0]), R(L.power[1]), R(L.power[2])} * (R(4) * Consts<R>::pi / solid_angle);
    Ray<R> r = {origin, normalize(to_target)};
    for (int depth = 0; depth < 10; ++depth) {
      const U4 bd = rng_block((uint32_t)P.rng, i, P.seed, 0xfffffffeu, 1u + (uint32_t)depth);
      if (depth > 3) {
        const R cp = min_(R(0.9), max_(power.x, max_(power.y, power.z)));
        if (u16lo<R>(bd.z) > cp) break;
        power = power * (R(1) / cp);
      }
      Hit<R> rec;
      ++n_closest;
      if (!trace_closest<R, MODE>(sc, r, R(0.001), R(1e30), rec, st)) break;
      if (rec.mat < 0) break;   // a quad / disk light: the reference indexes materials[negative] (undefined behaviour); the photon is lost
      const DMaterial<R>& m = sc.mats[rec.mat];
      if (m.type == MAT_DIELECTRIC) {
        V3<R> outward; R ni_over_nt, cosine;
        const R dn = dot(r.d, rec.n);
        if (dn > R(0)) { outward = -rec.n; ni_over_nt = m.ior; cosine = m.ior * dn; }
        else { outward = rec.n; ni_over_nt = R(1) / m.ior; cosine = -dn; }
        const V3<R> uv = normalize(r.d);   // refract_vec, :477-489
        const R dt = dot(uv, outward);
        const R disc = R(1) - ni_over_nt * ni_over_nt * (R(1) - dt * dt);
        V3<R> refracted = {R(0), R(0), R(0)};
        R reflect_prob = R(1);
        if (disc > R(0)) {
          refracted = (uv - outward * dt) * ni_over_nt - outward * sqrt_(disc);
          R r0 = (R(1) - m.ior) / (R(1) + m.ior);   // schlick, :491-495
          r0 = r0 * r0;
          reflect_prob = r0 + (R(1) - r0) * pow(R(1) - cosine, R(5));
        }
        if (u16hi<R>(bd.w) < reflect_prob) {
          const V3<R> refl = r.d - rec.n * (R(2) * dot(r.d, rec.n));
          r = {rec.p + refl * R(0.001), refl};
        } else {
          power = power * mk3(m.albedo);
          r = {rec.p + refracted * R(0.001), refracted};
        }
        continue;
      }
      if (m.type == MAT_METAL) {
        const V3<R> refl = r.d - rec.n * (R(2) * dot(r.d, rec.n));
        power = power * mk3(m.albedo);
        r = {rec.p + refl * R(0.001), refl};
        continue;
      }
      // Lambertian: the photon is stored (hit_specular is true from the start, :199) and the caustic pass ends here
      out[0 * n + i] = rec.p.x; out[1 * n + i] = rec.p.y; out[2 * n + i] = rec.p.z;
      out[3 * n + i] = -r.d.x; out[4 * n + i] = -r.d.y; out[5 * n + i] = -r.d.z;
      out[6 * n + i] = power.x; out[7 * n + i] = power.y; out[8 * n + i] = power.z;
      out[9 * n + i] = R(1);
      break;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n_closest += (uint32_t)__shfl_xor((int)n_closest, off, 64);
  if ((threadIdx.x & 63) == 0 && n_closest) atomicAdd(&counters->closest, (unsigned long long)n_closest);
}

}  // namespace rtx
