// Wavefront (queue-based) pipeline, single precision.
//
// A pool of path slots lives in HBM as structure-of-arrays.  One iteration advances every live
// path by one vertex with three kernels:
//
//   wf_generate_extend  slots whose path ended add it to their chunk sum and start the next
//                       sample (or fetch the next (pixel, chunk) work item); every live slot's ray
//                       is traced to its closest hit; surviving slots are compacted into the next
//                       ray queue
//   wf_shade            the vertex is shaded (emission, MIS, light sample, scatter, roulette);
//                       light samples that need a visibility test are compacted into the
//                       shadow-ray queue
//   wf_connect          shadow rays are traced (any hit); unoccluded ones add their contribution
//
// Queues are compacted with wave-level ballot + prefix popcount and ONE atomic per wave.
// The traversal kernels carry only a ray and a candidate, so they run at 2-4x the occupancy
// of the fused kernel; that is what pays for the queue traffic on scenes that walk a tree.
// Per-slot arithmetic and sample order equal the fused kernel's, so films agree to rounding.
#include "pt_kernels.hpp"
#include "pt_host.hpp"

namespace rtx {

namespace {

struct WfArrays {
  float* ray[6];
  float* thr[3];
  float* acc[3];
  float* sum[3];
  float* bsdf_pdf;
  uint32_t *bflags, *xy, *pixel, *sample, *s_end, *cur, *next;
  float *hit_t, *hit_ua, *hit_ub;
  int32_t *hit_id, *hit_uv;          // type << 24 | idx (or -1);  uv_kind << 24 | uv_idx
  uint32_t* q[2];
  uint32_t* sh_slot;
  float* sh[7];                       // origin, direction, t_max
  float* sh_c[3];
  uint32_t n_slots;
};

struct WfControl { uint32_t n_q[2]; uint32_t n_shadow; uint32_t pad; };

constexpr int kWfArrays = 6 + 3 + 3 + 3 + 1 + 7 + 3 + 2 + 2 + 1 + 7 + 3;   // 4-byte arrays of n_slots entries

// Append to a queue from (possibly divergent) code: the active lanes of the wave take
// consecutive entries with one atomic.
RTX_DEV uint32_t wave_append(uint32_t* counter) {
  const unsigned long long mask = __ballot(1);
  const int lane = (int)(threadIdx.x & 63);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, leader, 64);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

RTX_DEV void wave_count(unsigned long long* counter, bool pred) {
  const unsigned long long m = __ballot(pred);
  if ((threadIdx.x & 63) == (unsigned)(__ffsll((long long)__ballot(1)) - 1) && m) atomicAdd(counter, (unsigned long long)__popcll(m));
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void wf_generate_extend(DScene<float> sc, DCamera<float> cam, DRenderParams P, WfArrays A, WfControl* ctl,
                                                             int first, int qin, int qout, float* __restrict__ partial, DCounters* counters) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = first ? A.n_slots : ctl->n_q[qin];
  if (i >= n) return;
  const uint32_t slot = first ? i : A.q[qin][i];
  uint32_t bf = first ? 0u : A.bflags[slot];
  Ray<float> ray;
  bool live = true, started = false;
  if (!(bf & F_ALIVE)) {
    const uint32_t npix = (uint32_t)P.rows * (uint32_t)P.width;
    const uint32_t n_items = npix * (uint32_t)P.n_chunks;
    uint32_t s = first ? 0u : A.sample[slot], s_end = first ? 0u : A.s_end[slot];
    V3<float> sum = {0.f, 0.f, 0.f};
    if (!first) {
      sum = {A.sum[0][slot] + A.acc[0][slot], A.sum[1][slot] + A.acc[1][slot], A.sum[2][slot] + A.acc[2][slot]};
    }
    uint32_t pixel, xy;
    if (s == s_end) {
      if (s_end != 0u) {
        float* o = partial + (size_t)A.cur[slot] * 3u;
        o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
      }
      const uint32_t item = first ? i : A.next[slot];
      if (item >= n_items) {
        live = false;
        A.s_end[slot] = 0u; A.sample[slot] = 0u;
      } else {
        const uint32_t chunk = item / npix;
        const uint32_t pix = item - chunk * npix;
        const uint32_t row_local = pix / (uint32_t)P.width;
        const uint32_t x = pix - row_local * (uint32_t)P.width;
        const uint32_t y = (uint32_t)P.shard_index + row_local * (uint32_t)P.shard_count;
        xy = (y << 16) | x;
        pixel = y * (uint32_t)P.width + x;
        s = chunk * (uint32_t)P.chunk_spp;
        s_end = s + (uint32_t)P.chunk_spp < (uint32_t)P.spp ? s + (uint32_t)P.chunk_spp : (uint32_t)P.spp;
        sum = {0.f, 0.f, 0.f};
        A.cur[slot] = item; A.next[slot] = item + A.n_slots; A.xy[slot] = xy; A.pixel[slot] = pixel; A.s_end[slot] = s_end;
      }
    } else {
      xy = A.xy[slot]; pixel = A.pixel[slot];
    }
    if (live) {
      PathState<float> ps;
      PathKey key = {pixel, P.seed, s};
      path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key);
      ray = ps.ray;
      A.sample[slot] = s + 1u;     // index of the NEXT sample; the current one is sample - 1
      A.sum[0][slot] = sum.x; A.sum[1][slot] = sum.y; A.sum[2][slot] = sum.z;
      A.ray[0][slot] = ray.o.x; A.ray[1][slot] = ray.o.y; A.ray[2][slot] = ray.o.z;
      A.ray[3][slot] = ray.d.x; A.ray[4][slot] = ray.d.y; A.ray[5][slot] = ray.d.z;
      A.thr[0][slot] = 1.f; A.thr[1][slot] = 1.f; A.thr[2][slot] = 1.f;
      A.acc[0][slot] = 0.f; A.acc[1][slot] = 0.f; A.acc[2][slot] = 0.f;
      A.bsdf_pdf[slot] = 0.f;
      A.bflags[slot] = F_ALIVE;
      started = true;
    }
  } else {
    ray.o = {A.ray[0][slot], A.ray[1][slot], A.ray[2][slot]};
    ray.d = {A.ray[3][slot], A.ray[4][slot], A.ray[5][slot]};
  }
  if (live) {
    int id = -1, uv = 0;
    float t = 0.f, ua = 0.f, ub = 0.f;
    if constexpr (MODE == MODE_FLAT) {
      Hit<float> h;   // flat scenes: the candidate is cheap to redo in the shade kernel; keep t + id
      if (flat_closest(sc, ray, 0.001f, 1e9f, h)) { id = (h.type << 24) | h.idx; t = h.t; }
    } else {
      Cand<float> c;
      cand_init(c, 1e9f);
      walk_tree<float, false>(sc, ray, 0.001f, c, lane_stack(0));
      test_unbounded(sc, ray, 0.001f, c);
      if (c.type >= 0) { id = (c.type << 24) | c.idx; t = c.t; ua = c.ua; ub = c.ub; uv = (c.uv_kind << 24) | (c.uv_idx & 0xffffff); }
    }
    A.hit_t[slot] = t; A.hit_id[slot] = id;
    if constexpr (MODE == MODE_TREE_UV) { A.hit_ua[slot] = ua; A.hit_ub[slot] = ub; A.hit_uv[slot] = uv; }
    A.q[qout][wave_append(&ctl->n_q[qout])] = slot;
  }
  wave_count(&counters->closest, live);
  wave_count(&counters->samples, started);
}

// Rebuild the Hit of a stored candidate (point, normal, face, material, u/v).
template <int MODE> RTX_DEV bool wf_load_hit(const DScene<float>& sc, const WfArrays& A, uint32_t slot, const Ray<float>& ray, Hit<float>& h) {
  const int id = A.hit_id[slot];
  if (id < 0) return false;
  Cand<float> c;
  c.t = A.hit_t[slot]; c.type = id >> 24; c.idx = id & 0xffffff; c.ua = 0.f; c.ub = 0.f; c.uv_kind = 0; c.uv_idx = -1;
  if constexpr (MODE == MODE_TREE_UV) {
    const int uv = A.hit_uv[slot];
    c.ua = A.hit_ua[slot]; c.ub = A.hit_ub[slot]; c.uv_kind = uv >> 24; c.uv_idx = uv & 0xffffff;
    finalize_hit<float, true>(sc, ray, c, h);
  } else {
    finalize_hit<float, false>(sc, ray, c, h);
  }
  return true;
}

template <int MODE, bool LDS_MATS>
__global__ __launch_bounds__(kBlock) void wf_shade(DScene<float> sc, DRenderParams P, WfArrays A, WfControl* ctl, int q, DCounters* counters) {
  const DMaterial<float>* mats = sc.mats;
  if constexpr (LDS_MATS) mats = stage_materials(sc);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ctl->n_q[q]) return;
  const uint32_t slot = A.q[q][i];
  PathState<float> ps;
  ps.ray.o = {A.ray[0][slot], A.ray[1][slot], A.ray[2][slot]};
  ps.ray.d = {A.ray[3][slot], A.ray[4][slot], A.ray[5][slot]};
  ps.thr = {A.thr[0][slot], A.thr[1][slot], A.thr[2][slot]};
  ps.acc = {A.acc[0][slot], A.acc[1][slot], A.acc[2][slot]};
  ps.bsdf_pdf = A.bsdf_pdf[slot];
  ps.bounce_flags = A.bflags[slot];
  PathKey key = {A.pixel[slot], P.seed, A.sample[slot] - 1u};
  Hit<float> rec;
  const bool hit = wf_load_hit<MODE>(sc, A, slot, ps.ray, rec);
  const uint32_t did = path_vertex(ps, hit, rec, sc, mats, P, key, [&](const Ray<float>& sr, float t_max, V3<float> contrib) {
    const uint32_t k = wave_append(&ctl->n_shadow);
    A.sh_slot[k] = slot;
    A.sh[0][k] = sr.o.x; A.sh[1][k] = sr.o.y; A.sh[2][k] = sr.o.z;
    A.sh[3][k] = sr.d.x; A.sh[4][k] = sr.d.y; A.sh[5][k] = sr.d.z; A.sh[6][k] = t_max;
    A.sh_c[0][k] = contrib.x; A.sh_c[1][k] = contrib.y; A.sh_c[2][k] = contrib.z;
  });
  A.ray[0][slot] = ps.ray.o.x; A.ray[1][slot] = ps.ray.o.y; A.ray[2][slot] = ps.ray.o.z;
  A.ray[3][slot] = ps.ray.d.x; A.ray[4][slot] = ps.ray.d.y; A.ray[5][slot] = ps.ray.d.z;
  A.thr[0][slot] = ps.thr.x; A.thr[1][slot] = ps.thr.y; A.thr[2][slot] = ps.thr.z;
  A.acc[0][slot] = ps.acc.x; A.acc[1][slot] = ps.acc.y; A.acc[2][slot] = ps.acc.z;
  A.bsdf_pdf[slot] = ps.bsdf_pdf;
  A.bflags[slot] = ps.bounce_flags;
  wave_count(&counters->segments, (did & STEP_SHADED) != 0);
  wave_count(&counters->shadow, (did & STEP_SHADOW) != 0);
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void wf_connect(DScene<float> sc, WfArrays A, const WfControl* ctl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ctl->n_shadow) return;
  Ray<float> sr;
  sr.o = {A.sh[0][i], A.sh[1][i], A.sh[2][i]};
  sr.d = {A.sh[3][i], A.sh[4][i], A.sh[5][i]};
  const float t_max = A.sh[6][i];
  if (!trace_occluded<float, MODE>(sc, sr, 0.001f, t_max, lane_stack(0))) {
    const uint32_t slot = A.sh_slot[i];   // at most one shadow ray per slot per iteration: no atomics needed
    A.acc[0][slot] += A.sh_c[0][i]; A.acc[1][slot] += A.sh_c[1][i]; A.acc[2][slot] += A.sh_c[2][i];
  }
}

template <int MODE> void launch_iteration_kernels(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, const WfArrays& A, WfControl* ctl,
                                                  float* partial, DCounters* counters, bool lds_mats, size_t mat_lds, size_t stack_lds, unsigned grid,
                                                  int step /*0 first generate, 1 shade, 2 connect, 3 generate*/, int qa, int qb, hipStream_t stream) {
  const dim3 g(grid), b(kBlock);
  const size_t walk_lds = MODE == MODE_FLAT ? 0 : stack_lds;
  if (step == 0 || step == 3)
    hipLaunchKernelGGL((wf_generate_extend<MODE>), g, b, walk_lds, stream, sc, cam, P, A, ctl, step == 0 ? 1 : 0, qa, qb, partial, counters);
  else if (step == 1) {
    if (lds_mats) hipLaunchKernelGGL((wf_shade<MODE, true>), g, b, mat_lds, stream, sc, P, A, ctl, qa, counters);
    else hipLaunchKernelGGL((wf_shade<MODE, false>), g, b, 0, stream, sc, P, A, ctl, qa, counters);
  } else
    hipLaunchKernelGGL((wf_connect<MODE>), g, b, walk_lds, stream, sc, A, ctl);
}

}  // namespace

int wavefront_render(WavefrontPool& pool, const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* d_film,
                     DCounters* counters, int n_cus, int mode, bool needs_stack, hipStream_t stream, uint32_t& launches, std::string& err) {
  const unsigned long long npix = (unsigned long long)P.rows * P.width;
  const unsigned long long items = npix * (unsigned long long)P.n_chunks;
  if (items >= 0x7fffffffull) { err = "film too large for one launch"; return RTX_ERR_INVALID_ARGUMENT; }
  // Pool size: enough slots to keep every CU at full occupancy several times over.
  const uint32_t n_slots = (uint32_t)std::min<unsigned long long>(items, (unsigned long long)n_cus * 2048ull * 8ull);
  if (!pool.state.ensure((size_t)kWfArrays * n_slots * 4) || !pool.control.ensure(sizeof(WfControl)) ||
      !pool.queues.ensure((size_t)items * 3 * sizeof(float))) { err = "pool allocation"; return RTX_ERR_OUT_OF_MEMORY; }
  WfArrays A;
  {
    uint32_t* base = (uint32_t*)pool.state.ptr;
    size_t k = 0;
    auto nextf = [&]() { return (float*)(base + (k++) * (size_t)n_slots); };
    auto nextu = [&]() { return (uint32_t*)(base + (k++) * (size_t)n_slots); };
    for (auto& p : A.ray) p = nextf();
    for (auto& p : A.thr) p = nextf();
    for (auto& p : A.acc) p = nextf();
    for (auto& p : A.sum) p = nextf();
    A.bsdf_pdf = nextf();
    A.bflags = nextu(); A.xy = nextu(); A.pixel = nextu(); A.sample = nextu(); A.s_end = nextu(); A.cur = nextu(); A.next = nextu();
    A.hit_t = nextf(); A.hit_ua = nextf(); A.hit_ub = nextf();
    A.hit_id = (int32_t*)nextu(); A.hit_uv = (int32_t*)nextu();
    A.q[0] = nextu(); A.q[1] = nextu();
    A.sh_slot = nextu();
    for (auto& p : A.sh) p = nextf();
    for (auto& p : A.sh_c) p = nextf();
    A.n_slots = n_slots;
    if (k != (size_t)kWfArrays) { err = "internal: array count"; return RTX_ERR_INVALID_ARGUMENT; }
  }
  WfControl* ctl = (WfControl*)pool.control.ptr;
  float* partial = (float*)pool.queues.ptr;
  const size_t mat_bytes = (size_t)sc.n_mats * sizeof(DMaterial<float>);
  const bool lds_mats = mat_bytes <= kMaxLdsMaterialBytes;
  const size_t mat_lds = (mat_bytes + 15) & ~(size_t)15;
  const size_t stack_lds = needs_stack ? (size_t)kStackDepth * kBlock * sizeof(int) : 0;
  if (!lds_mats) mode = MODE_TREE_UV;

  auto run = [&](int step, int qa, int qb, unsigned grid) {
    if (mode == MODE_FLAT) launch_iteration_kernels<MODE_FLAT>(sc, cam, P, A, ctl, partial, counters, lds_mats, mat_lds, stack_lds, grid, step, qa, qb, stream);
    else if (mode == MODE_TREE) launch_iteration_kernels<MODE_TREE>(sc, cam, P, A, ctl, partial, counters, lds_mats, mat_lds, stack_lds, grid, step, qa, qb, stream);
    else launch_iteration_kernels<MODE_TREE_UV>(sc, cam, P, A, ctl, partial, counters, lds_mats, mat_lds, stack_lds, grid, step, qa, qb, stream);
    ++launches;
  };
  if (hipMemsetAsync(ctl, 0, sizeof(WfControl), stream) != hipSuccess) { err = "memset"; return RTX_ERR_HIP; }
  unsigned live_bound = n_slots;   // upper bound of the ray queue length (it only shrinks)
  unsigned grid = (live_bound + kBlock - 1) / kBlock;
  run(0, 0, 0, grid);
  int cur = 0;
  for (unsigned iter = 0;; ++iter) {
    run(1, cur, 0, grid);
    run(2, 0, 0, grid);
    // clear the queue the next generate/extend fills and the shadow queue
    if (hipMemsetAsync(&ctl->n_q[1 - cur], 0, sizeof(uint32_t), stream) != hipSuccess ||
        hipMemsetAsync(&ctl->n_shadow, 0, sizeof(uint32_t), stream) != hipSuccess) { err = "memset"; return RTX_ERR_HIP; }
    run(3, cur, 1 - cur, grid);
    cur = 1 - cur;
    if ((iter & 7u) == 7u || live_bound < 65536u) {
      uint32_t n_live = 0;
      if (hipMemcpyAsync(&n_live, &ctl->n_q[cur], sizeof n_live, hipMemcpyDeviceToHost, stream) != hipSuccess ||
          hipStreamSynchronize(stream) != hipSuccess) { err = hipGetErrorString(hipGetLastError()); return RTX_ERR_HIP; }
      if (n_live == 0) break;
      live_bound = n_live;
      grid = (live_bound + kBlock - 1) / kBlock;
    }
    if (iter > 4000000u) { err = "iteration limit"; return RTX_ERR_HIP; }
  }
  launch_film_reduce(partial, d_film, npix, P.n_chunks, P.spp, stream);
  ++launches;
  return RTX_OK;
}

}  // namespace rtx
