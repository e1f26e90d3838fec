// Double-precision instantiations: the parity / validation kernels.
// Built with -ffp-contract=off so that no multiply-add is fused: the reference CPU build
// (baseline x86-64) rounds after every multiply and every add, and so must these.
#include "pt_kernels.hpp"
#include "pt_bdpt.hpp"
#include "pt_photon_kernel.hpp"
#include "pt_host.hpp"

namespace rtx {

// Scene::hit (ref: engine/scene.hpp:574-674) on the reference-order binary BVH: same node
// order (left before right; right pushed first), same 64-entry stack rules (:582-586,
// :634-635), same acceptance of ties (a later primitive with equal t wins), same slab
// test (engine/bvh.hpp:35-64).  One ray per lane; the per-lane stack lives in LDS.
__global__ __launch_bounds__(64) void trace_ref_kernel(DScene<double> sc, DSceneRef ref, const double* __restrict__ rays8,
                                                        unsigned long long n, DHitOut* __restrict__ out) {
  __shared__ int stack_mem[64 * 64];
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int* stack = stack_mem + threadIdx.x;  // [level][lane]
  const double* q = rays8 + i * 8ull;
  Ray<double> r = {{q[0], q[1], q[2]}, {q[3], q[4], q[5]}};
  const double t_min = q[6];
  Cand<double> c;
  cand_init(c, q[7]);
  if (ref.n_nodes > 0) {
    int sp = 0;
    stack[(sp++) * 64] = 0;
    while (sp > 0 && sp < 64) {
      int ni = stack[(--sp) * 64];
      if (ni < 0 || ni >= ref.n_nodes) continue;
      const DNodeRef nd = ref.nodes[ni];
      if (!slab_hit(nd.lo, nd.hi, r, t_min, c.t)) continue;
      if (nd.count > 0) {
        for (int k = 0; k < nd.count; ++k) {
          int pi = nd.first + k;
          if (pi < 0 || pi >= ref.n_prims) continue;
          DPrimRef p = ref.prims[pi];
          test_bounded(sc, p.type, p.index, r, t_min, c);
        }
      } else {
        if (nd.right >= 0 && sp < 63) stack[(sp++) * 64] = nd.right;
        if (nd.left >= 0 && sp < 63) stack[(sp++) * 64] = nd.left;
      }
    }
  }
  test_unbounded(sc, r, t_min, c);
  DHitOut o;
  const bool ok = c.type >= 0;
  Hit<double> h;
  h.p = {0, 0, 0}; h.n = {0, 0, 0}; h.t = 0; h.u = 0; h.v = 0; h.front = 0; h.mat = 0; h.type = -1; h.idx = -1;
  if (ok) finalize_hit<double, true>(sc, r, c, h);
  o.hit = ok; o.prim_type = h.type; o.prim_index = h.idx; o.material_id = h.mat; o.front_face = h.front; o.pad = 0;
  o.t = h.t; o.point[0] = h.p.x; o.point[1] = h.p.y; o.point[2] = h.p.z;
  o.normal[0] = h.n.x; o.normal[1] = h.n.y; o.normal[2] = h.n.z; o.u = h.u; o.v = h.v;
  out[i] = o;
}

void launch_trace_ref(const DScene<double>& sc, const DSceneRef& ref, const double* rays8, unsigned long long n, DHitOut* out, hipStream_t stream) {
  if (n == 0) return;
  unsigned grid = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(trace_ref_kernel, dim3(grid), dim3(64), 0, stream, sc, ref, rays8, n, out);
}

void launch_photon_trace(const DScene<double>& sc, const DRenderParams& P, const DPhotonLight* lights, int n_lights, const int32_t* targets, int n_targets,
                         uint32_t n_photons, double* out, DCounters* counters, size_t lds_bytes, hipStream_t stream) {
  const unsigned grid = (n_photons + kBlock - 1) / kBlock;
  hipLaunchKernelGGL((photon_trace_kernel<double, MODE_TREE_UV>), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, P, lights, n_lights, targets, n_targets, n_photons, out, counters, 0);
}

int launch_render_bdpt(const DScene<double>& sc, const DCamera<double>& cam, const DRenderParams& P, double* partial, DCounters* counters, const FusedLaunch& L,
                       double* light_store, unsigned long long store_cap, hipStream_t stream) {
  dim3 g(L.grid), b(kBlock);
  if (L.mode == MODE_FLAT) return -1;
  if (L.lds_mats) hipLaunchKernelGGL((render_bdpt_kernel<double, MODE_TREE_UV, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int, light_store, store_cap);
  else hipLaunchKernelGGL((render_bdpt_kernel<double, MODE_TREE_UV, false>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int, light_store, store_cap);
  return 0;
}

int launch_render_fused(const DScene<double>& sc, const DCamera<double>& cam, const DRenderParams& P, double* partial, DCounters* counters,
                        const FusedLaunch& L, hipStream_t stream) {
  dim3 g(L.grid), b(kBlock);
  if (L.integ == INTEG_PATH_MNEE && L.mnee_batched) {   // the solves of a wave batched (pt_kernels.hpp)
    if (L.mode == MODE_TREE && L.lds_mats)
      hipLaunchKernelGGL((render_fused_mnee_batched_kernel<double, MODE_TREE, true, 0>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (L.mode == MODE_TREE_UV && L.lds_mats)
      hipLaunchKernelGGL((render_fused_mnee_batched_kernel<double, MODE_TREE_UV, true, 0>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (!L.lds_mats && L.mode != MODE_FLAT)
      hipLaunchKernelGGL((render_fused_mnee_batched_kernel<double, MODE_TREE_UV, false, 0>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else
      return -1;
    return 0;
  }
  if (L.integ == INTEG_PATH_MNEE) {
    if (L.mode == MODE_TREE && L.lds_mats)
      hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE, true, INTEG_PATH_MNEE>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (L.mode == MODE_TREE_UV && L.lds_mats)
      hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, true, INTEG_PATH_MNEE>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (!L.lds_mats && L.mode != MODE_FLAT)
      hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, false, INTEG_PATH_MNEE>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else
      return -1;
    return 0;
  }
  if (L.integ == INTEG_PATHPHOTON) {   // as Whitted: the general (TREE_UV) instances only
    if (L.mode != MODE_TREE_UV) return -1;
    if (L.lds_mats) hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, true, INTEG_PATHPHOTON>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, false, INTEG_PATHPHOTON>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    return 0;
  }
  if (L.integ == INTEG_WHITTED) {   // the Whitted integrator exists for the general (TREE_UV) instances only
    if (L.mode != MODE_TREE_UV) return -1;
    if (L.lds_mats) hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, true, INTEG_WHITTED>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, false, INTEG_WHITTED>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    return 0;
  }
  if (L.mode == MODE_TREE && L.lds_mats)
    hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else if (L.mode == MODE_TREE_UV && L.lds_mats)
    hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else if (!L.lds_mats && L.mode != MODE_FLAT)
    hipLaunchKernelGGL((render_fused_kernel<double, MODE_TREE_UV, false>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else
    return -1;
  return 0;
}

void launch_film_reduce(const double* partial, float* film, unsigned long long npix, int n_chunks, int spp, hipStream_t stream) {
  unsigned grid = (unsigned)((npix * 3ull + 255) / 256);
  hipLaunchKernelGGL(film_reduce_kernel<double>, dim3(grid), dim3(256), 0, stream, partial, film, npix, n_chunks, spp);
}

int launch_trace_fast(const DScene<double>& sc, const double* rays8, unsigned long long n, DHitOut* out, int mode, size_t lds_bytes, hipStream_t stream, int n_cus) {
  if (n == 0) return 0;
  if (mode == MODE_FLAT) return -1;
  unsigned grid = trace_fast_grid(n, n_cus);
  hipLaunchKernelGGL((trace_fast_kernel<double, MODE_TREE_UV>), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, rays8, n, out);
  return 0;
}

// ------------------------------------------------------------------------------------------
// Post-process: exposure, tone map, clamp, sqrt gamma, 255.999 quantisation, vertical flip
// (ref: engine/renderer.cpp:102-198, :232-258).  One thread per channel value; double precision,
// no contraction (this file is built with -ffp-contract=off), so the bytes equal the reference's.
// ------------------------------------------------------------------------------------------
namespace {
RTX_DEV double pp_clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
RTX_DEV double pp_u2_partial(double x) {   // ref: renderer.cpp:131-139
  const double A = 0.15, B = 0.50, C = 0.10, D = 0.20, E = 0.02, F = 0.30;
  return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ film, unsigned char* __restrict__ out, int width, int height, int mapper,
                                                         double exposure) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long row = (unsigned long long)width * 3ull;
  if (i >= row * (unsigned long long)height) return;
  const unsigned long long y = i / row, within = i - y * row;
  const double x = (double)film[i] * exposure;   // ref: renderer.cpp:189-193
  double m;
  switch (mapper) {                              // ref: renderer.cpp:155-184
    case RTX_TONE_REINHARD: m = x / (1.0 + x); break;
    case RTX_TONE_REINHARD_EXTENDED: { const double wp2 = 4.0 * 4.0; m = x * (1.0 + x / wp2) / (1.0 + x); break; }
    case RTX_TONE_ACES: { const double a = 2.51, b = 0.03, c = 2.43, d = 0.59, e = 0.14; m = pp_clamp01((x * (a * x + b)) / (x * (c * x + d) + e)); break; }
    case RTX_TONE_UNCHARTED2: { const double curr = pp_u2_partial(x * 2.0); const double white_scale = 1.0 / pp_u2_partial(11.2); m = curr * white_scale; break; }
    default: m = pp_clamp01(x); break;
  }
  const double g = sqrt(pp_clamp01(m));          // ref: renderer.cpp:241-243
  const double q = 255.999 * g;                  // ref: renderer.cpp:246-248
  const int iv = q == q ? (int)q : 0;
  out[((unsigned long long)height - 1ull - y) * row + within] = (unsigned char)iv;   // ref: renderer.cpp:251-255
}
}  // namespace

void launch_postprocess(const float* film, unsigned char* out, int width, int height, int mapper, double exposure, hipStream_t stream) {
  const unsigned long long n = (unsigned long long)width * height * 3ull;
  hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, film, out, width, height, mapper, exposure);
}

}  // namespace rtx
