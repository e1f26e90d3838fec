// Single-precision instantiations: the production kernels.
#include <cstdio>

#include "pt_kernels.hpp"
#include "pt_bdpt.hpp"
#include "pt_photon_kernel.hpp"
#include "pt_host.hpp"

namespace rtx {

// Counterpart of the reference backend's smoke kernel (ref: engine/cuda/cuda_test.cu:14-19).
__global__ void selftest_kernel(float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)i * 0.001f;
}

void launch_selftest(float* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n);
}

__global__ void zero_words_kernel(uint32_t* p, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
void launch_zero_words(uint32_t* p, unsigned n, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(zero_words_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, n);
}

void launch_photon_trace(const DScene<float>& sc, const DRenderParams& P, const DPhotonLight* lights, int n_lights, const int32_t* targets, int n_targets,
                         uint32_t n_photons, float* out, DCounters* counters, size_t lds_bytes, hipStream_t stream) {
  const unsigned grid = (n_photons + kBlock - 1) / kBlock;
  hipLaunchKernelGGL((photon_trace_kernel<float, MODE_TREE_UV>), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, P, lights, n_lights, targets, n_targets, n_photons, out, counters, 0);
}

int launch_render_bdpt(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* partial, DCounters* counters, const FusedLaunch& L,
                       float* light_store, unsigned long long store_cap, hipStream_t stream) {
  dim3 g(L.grid), b(kBlock);
  if (L.mode == MODE_FLAT && L.lds_mats) { hipLaunchKernelGGL((render_bdpt_kernel<float, MODE_FLAT, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int, light_store, store_cap); return 0; }
  if (L.mode == MODE_FLAT) return -1;
  if (L.lds_mats) hipLaunchKernelGGL((render_bdpt_kernel<float, MODE_TREE_UV, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int, light_store, store_cap);
  else hipLaunchKernelGGL((render_bdpt_kernel<float, MODE_TREE_UV, false>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int, light_store, store_cap);
  return 0;
}

static int launch_render_fused_instance(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* partial, DCounters* counters,
                                        const FusedLaunch& L, hipStream_t stream) {
  dim3 g(L.grid), b(kBlock);
  if (L.integ == INTEG_PATHPHOTON) {   // the flat instance or the general (TREE_UV) ones
    if (L.mode == MODE_FLAT && L.lds_mats) {
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATHPHOTON>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
      return 0;
    }
    if (L.mode != MODE_TREE_UV) return -1;
    if (L.lds_mats) hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, true, INTEG_PATHPHOTON>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, false, INTEG_PATHPHOTON>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    return 0;
  }
  if (L.integ == INTEG_WHITTED) {   // the Whitted integrator exists for the general (TREE_UV) instances only
    if (L.mode != MODE_TREE_UV) return -1;
    if (L.lds_mats) hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, true, INTEG_WHITTED>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, false, INTEG_WHITTED>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    return 0;
  }
  if (L.integ == INTEG_PATH_MNEE && L.mnee_batched) {   // manifold NEE with the solves of a wave batched (pt_kernels.hpp)
    if (L.mode == MODE_FLAT && L.lds_mats && !sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && sc.n_quads == 0 && P.use_nee && P.use_mis && P.rng == 1 /* RNG_PCG4D */)
      hipLaunchKernelGGL((render_fused_mnee_batched_kernel<float, MODE_FLAT, true, 31>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (L.mode == MODE_FLAT && L.lds_mats) hipLaunchKernelGGL((render_fused_mnee_batched_kernel<float, MODE_FLAT, true, 0>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (L.mode == MODE_FLAT) return -1;
    else if (L.lds_mats) hipLaunchKernelGGL((render_fused_mnee_batched_kernel<float, MODE_TREE_UV, true, 0>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else hipLaunchKernelGGL((render_fused_mnee_batched_kernel<float, MODE_TREE_UV, false, 0>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    return 0;
  }
  if (L.integ == INTEG_PATH_MNEE) {   // manifold NEE: single-precision traces, the manifold walk itself in double (pt_mnee.hpp)
    if (L.mode == MODE_FLAT && L.lds_mats && !sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && sc.n_quads == 0 && P.use_nee && P.use_mis && P.rng == 1 /* RNG_PCG4D */)
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH_MNEE, 31>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);   // the lean instance (see below)
    else if (L.mode == MODE_FLAT && L.lds_mats) hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH_MNEE>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (L.mode == MODE_FLAT) return -1;
    else if (L.lds_mats) hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, true, INTEG_PATH_MNEE>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, false, INTEG_PATH_MNEE>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    return 0;
  }
  if (L.integ != INTEG_PATH) return -1;
  if (L.mode == MODE_FLAT && L.lds_mats)
  {
    // the flat instance comes in several sizes: the lean ones for scenes without environment map, point and disk lights,
    // with the settings that do not vary pinned (less code, fewer live scalars: the kernel keeps the scene in scalar registers
    // and spills what does not fit to vector lanes).  Measured, 800x800x128: the Cornell box with a quad light 9.88 -> 9.47 ms with
    // instance 27 instead of 3; the same pinning for the environment-map scene (30 instead of 6) was 4 % slower and is not used.
    if (L.regroup > 0 && !P.item_list && !sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && sc.n_quads == 0 && P.use_nee && P.use_mis && P.rng == 1 /* RNG_PCG4D */)
      hipLaunchKernelGGL((render_fused_regroup_kernel<31>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.regroup);   // experiment: material-coherent waves
    else if (!sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && sc.n_quads == 0 && P.use_nee && P.use_mis && P.rng == 1 /* RNG_PCG4D */)
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH, 31>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (!sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && sc.n_quads == 0)
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH, 7>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (!sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && sc.n_emissive == 0 && P.use_nee && P.use_mis && P.rng == 1 /* RNG_PCG4D */)   // quad lights only: the Cornell box with its ceiling panel
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH, 91>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (!sc.env.valid && sc.n_points == 0 && sc.n_disks == 0 && P.use_nee && P.use_mis && P.rng == 1 /* RNG_PCG4D */)
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH, 27>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (!sc.env.valid && sc.n_points == 0 && sc.n_disks == 0)
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH, 3>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else if (sc.n_points == 0 && sc.n_disks == 0 && sc.n_quads == 0)
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true, INTEG_PATH, 6>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
    else
      hipLaunchKernelGGL((render_fused_kernel<float, MODE_FLAT, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  }
  else if (L.mode == MODE_TREE && L.lds_mats)
    hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else if (L.mode == MODE_TREE_UV && L.lds_mats && sc.n_nmaps == 0)
    hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, true, INTEG_PATH, 32>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else if (L.mode == MODE_TREE_UV && L.lds_mats)
    hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, true>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else if (!L.lds_mats && L.mode != MODE_FLAT && sc.n_nmaps == 0)
    hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, false, INTEG_PATH, 32>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else if (!L.lds_mats && L.mode != MODE_FLAT)
    hipLaunchKernelGGL((render_fused_kernel<float, MODE_TREE_UV, false>), g, b, L.lds_bytes, stream, sc, cam, P, partial, counters, L.stack_first_int);
  else
    return -1;
  return 0;
}

int launch_render_fused(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* partial, DCounters* counters,
                        const FusedLaunch& L, hipStream_t stream) {
  const int rc = launch_render_fused_instance(sc, cam, P, partial, counters, L, stream);
#ifdef RTX_PHASE_STATS
  {
    unsigned long long h[64];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_stats), sizeof h);
    fprintf(stderr, "phase stats:");
    for (int b = 0; b < 32; ++b) if (h[2 * b]) fprintf(stderr, " %d:%llu/%llu", b, h[2 * b], h[2 * b + 1]);
    fprintf(stderr, "\n");
    for (auto& v : h) v = 0ull;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_stats), h, sizeof h);
  }
#endif
  return rc;
}

void launch_film_reduce(const float* partial, float* film, unsigned long long npix, int n_chunks, int spp, hipStream_t stream) {
  unsigned grid = (unsigned)((npix * 3ull + 255) / 256);
  hipLaunchKernelGGL(film_reduce_kernel<float>, dim3(grid), dim3(256), 0, stream, partial, film, npix, n_chunks, spp);
}

int launch_trace_fast(const DScene<float>& sc, const double* rays8, unsigned long long n, DHitOut* out, int mode, size_t lds_bytes, hipStream_t stream, int n_cus) {
  if (n == 0) return 0;
  unsigned grid = trace_fast_grid(n, n_cus);
  if (mode == MODE_FLAT)
    hipLaunchKernelGGL((trace_fast_kernel<float, MODE_FLAT>), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, rays8, n, out);
  else
    hipLaunchKernelGGL((trace_fast_kernel<float, MODE_TREE_UV>), dim3(grid), dim3(kBlock), lds_bytes, stream, sc, rays8, n, out);
  return 0;
}

}  // namespace rtx
