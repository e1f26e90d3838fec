// Kernel templates + their launchers.  Instantiated for double in kernels_f64.hip (built with
// -ffp-contract=off) and for float in kernels_f32.hip.
#pragma once

#include "pt_device.hpp"

namespace rtx {

extern __shared__ int lds_dyn[];

RTX_DEV LaneStack lane_stack() {
  LaneStack st;
  st.base = lds_dyn + threadIdx.x;
  st.stride = blockDim.x;
  return st;
}

RTX_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffull), off, 64);
    unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// Fused path tracer: every lane owns one work item = (pixel, sample chunk) at a time and
// advances its path one vertex per loop trip; a lane whose path ends starts its next sample
// (or fetches its next work item) in the same trip, so lanes stay busy until the item queue
// is empty.  Chunk sums go to `partial`; film_reduce_kernel adds them in chunk order, which
// makes the film independent of grid size, scheduling and sharding.
// ------------------------------------------------------------------------------------------
template <class R>
__global__ __launch_bounds__(kBlock) void render_fused_kernel(DScene<R> sc, DCamera<R> cam, DRenderParams P, R* __restrict__ partial,
                                                               DCounters* __restrict__ counters) {
  const unsigned long long npix = (unsigned long long)P.rows * (unsigned long long)P.width;
  const unsigned long long n_items = npix * (unsigned long long)P.n_chunks;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long next = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st = lane_stack();

  StepCounts cnt = {0, 0, 0};
  uint32_t n_samples = 0;
  PathState<R> ps;
  ps.alive = false;
  V3<R> sum = {R(0), R(0), R(0)};
  unsigned long long cur = 0;
  int s = 0, s_end = 0, x = 0, y = 0;
  PathKey key = {0u, P.seed, 0u};

  auto fetch = [&]() -> bool {
    if (next >= n_items) return false;
    cur = next;
    next += stride;
    unsigned long long pix = cur % npix;
    int chunk = (int)(cur / npix);
    int row_local = (int)(pix / (unsigned long long)P.width);
    x = (int)(pix - (unsigned long long)row_local * P.width);
    y = P.shard_index + row_local * P.shard_count;
    key.pixel = (uint32_t)(y * P.width + x);
    s = chunk * P.chunk_spp;
    s_end = s + P.chunk_spp < P.spp ? s + P.chunk_spp : P.spp;
    sum = {R(0), R(0), R(0)};
    return s < s_end;
  };

  bool active = fetch();
  if (active) { key.sample = (uint32_t)s++; path_begin(ps, cam, P, x, y, key); n_samples++; }
  while (active) {
    path_step(ps, sc, P, key, st, cnt);
    if (!ps.alive) {
      sum = sum + ps.acc;
      if (s == s_end) {
        R* o = partial + cur * 3ull;
        o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
        active = fetch();
        if (!active) break;
      }
      key.sample = (uint32_t)s++;
      path_begin(ps, cam, P, x, y, key);
      n_samples++;
    }
  }

  unsigned long long a = wave_sum(n_samples), b = wave_sum(cnt.closest), c = wave_sum(cnt.shadow), d = wave_sum(cnt.segments);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counters->samples, a);
    atomicAdd(&counters->closest, b);
    atomicAdd(&counters->shadow, c);
    atomicAdd(&counters->segments, d);
  }
}

// film[pixel] = (sum of the pixel's chunk sums, in chunk order) * (1 / spp)
// ref: engine/renderer.cpp:447 (pixel_color * (1.0 / samples_per_pixel))
template <class R>
__global__ void film_reduce_kernel(const R* __restrict__ partial, float* __restrict__ film, unsigned long long npix, int n_chunks, int spp) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * 3ull) return;
  R acc = R(0);
  for (int c = 0; c < n_chunks; ++c) acc = acc + partial[(unsigned long long)c * npix * 3ull + i];
  film[i] = (float)(acc * (R(1) / R(spp)));
}

// Closest hit of a ray batch through the production BVH (the structure the integrator walks).
template <class R>
__global__ __launch_bounds__(kBlock) void trace_fast_kernel(DScene<R> sc, const double* __restrict__ rays8, unsigned long long n, DHitOut* __restrict__ out) {
  unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* q = rays8 + i * 8ull;
  Ray<R> r = {{R(q[0]), R(q[1]), R(q[2])}, {R(q[3]), R(q[4]), R(q[5])}};
  Hit<R> h;
  bool ok = closest_hit<R, true>(sc, r, R(q[6]), R(q[7]), h, lane_stack());
  DHitOut o;
  o.hit = ok; o.prim_type = ok ? h.type : -1; o.prim_index = ok ? h.idx : -1;
  o.material_id = ok ? h.mat : 0; o.front_face = ok ? h.front : 0; o.pad = 0;
  o.t = ok ? (double)h.t : 0.0;
  o.point[0] = ok ? (double)h.p.x : 0.0; o.point[1] = ok ? (double)h.p.y : 0.0; o.point[2] = ok ? (double)h.p.z : 0.0;
  o.normal[0] = ok ? (double)h.n.x : 0.0; o.normal[1] = ok ? (double)h.n.y : 0.0; o.normal[2] = ok ? (double)h.n.z : 0.0;
  o.u = ok ? (double)h.u : 0.0; o.v = ok ? (double)h.v : 0.0;
  out[i] = o;
}

}  // namespace rtx
