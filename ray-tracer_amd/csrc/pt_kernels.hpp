// Kernel templates.  Instantiated for double in kernels_f64.hip (built with -ffp-contract=off)
// and for float in kernels_f32.hip.
#pragma once

#include "pt_device.hpp"

namespace rtx {

// Dynamic LDS of a workgroup: [material table (when staged)] [traversal stacks, [level][lane]].
extern __shared__ int lds_dyn[];

RTX_DEV LaneStack lane_stack(int first_int) {
  LaneStack st;
  st.base = lds_dyn + first_int + threadIdx.x;
  return st;
}

RTX_DEV uint32_t wave_max(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

template <class R> constexpr int mat_ints() { return (int)(sizeof(DMaterial<R>) / sizeof(int)); }

// Copy the material table into LDS (it is indexed per lane at every shaded vertex).
template <class R> RTX_DEV const DMaterial<R>* stage_materials(const DScene<R>& sc) {
  const int n = sc.n_mats * mat_ints<R>();
  const int* src = reinterpret_cast<const int*>(sc.mats);
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds_dyn[i] = src[i];
  __syncthreads();
  return reinterpret_cast<const DMaterial<R>*>(lds_dyn);
}

// ------------------------------------------------------------------------------------------
// Fused path tracer: every lane owns one work item = (pixel, sample chunk) at a time and
// advances its path one vertex per loop trip; a lane whose path ends starts its next sample
// (or fetches its next work item) in the same trip, so lanes stay busy until the item queue
// is empty.  Chunk sums go to `partial`; film_reduce_kernel adds them in chunk order, which
// makes the film independent of grid size, scheduling and sharding.
// ------------------------------------------------------------------------------------------
// The flat instance is capped at 128 VGPRs (4 waves per SIMD): measured 7 % faster than the
// compiler's free choice of 140 despite 32 bytes of scratch per lane.
#ifndef RTX_FLAT_MIN_WAVES
#define RTX_FLAT_MIN_WAVES 4
#endif
// The lean flat instances (see LEAN below) need fewer registers.  With a grid of exactly the resident workgroups they were best at 6
// waves per SIMD (3: 310.8 ms, 4: 309.7, 5: 298.9, 6: 292.3, 7: 296.1, 8: 330.4 on the 1080p Cornell box); with many short
// workgroups (rtx_api.cpp) more waves pay: 5: 167.9 ms, 6: 168.0, 7: 163.8, 8: 164.0 (800 x 800 x 128, two scenes: 15.2 / 14.6 /
// 14.7 / 14.7) when that grid was introduced; measured again on the kernels as they are now, 7: 166.0 ms, 8 (64 registers, ten of
// them spilled): 160.5, and 8.62 / 6.86 -> 8.38 / 6.77 ms for the two 800 x 800 scenes.
#ifndef RTX_LEAN_ENV_MIN_WAVES
#define RTX_LEAN_ENV_MIN_WAVES 6   // lean instance that keeps the environment map
#endif
#ifndef RTX_LEAN_MIN_WAVES
#define RTX_LEAN_MIN_WAVES 8
#endif
#ifndef RTX_TREE_MIN_WAVES
#define RTX_TREE_MIN_WAVES 5   // single-precision tree-walking path instances (earth_demo 4K x 256 spp: 2: 93.2 ms, 3: 92.9, 4: 76.2, 5: 69.5, 6: 77.0)
#endif
#ifndef RTX_F64_MIN_WAVES
#define RTX_F64_MIN_WAVES 1   // double-precision instances (other than manifold NEE): the compiler's choice, 247 registers, 2 waves (1080p Cornell box, 32 spp: 32.5 ms; capped for 2 / 3 / 4 waves: 32.9 / 32.9 / 30.3, but the Whitted instance 19.5 -> 19.6 / 21.5 / 25.8)
#endif
#ifndef RTX_WHITTED_MIN_WAVES
#define RTX_WHITTED_MIN_WAVES 6   // the single-precision Whitted instance (demo.json 2560 x 1440 x 256 spp: 3: 71.7 ms, 4: 64.8, 5: 65.6, 6: 55.3)
#endif
#ifndef RTX_PHOTON_MIN_WAVES
#define RTX_PHOTON_MIN_WAVES 3   // single-precision PathPhoton instance (measured: 4 waves per SIMD spill and are slower)
#endif
#ifndef RTX_PHOTON_FLAT_MIN_WAVES
#define RTX_PHOTON_FLAT_MIN_WAVES 5   // PathPhoton on a flat scene (1080p Cornell box, 16 spp, the reference's photon defaults: 3: 131.1 ms, 4: 105.8, 5: 98.7, 6: 184.0)
#endif
#ifndef RTX_MNEE_F32_MIN_WAVES
#define RTX_MNEE_F32_MIN_WAVES 4   // manifold NEE instances with single-precision traces (measured on the 1080p Cornell box, 16 spp: 2: 342 ms, 3: 301, 4: 288, 5: 421)
#endif
#ifndef RTX_MNEE_MIN_WAVES
#define RTX_MNEE_MIN_WAVES 3   // manifold NEE instance (double precision): measured 1: 330 ms, 2: 240, 3: 219, 4: 263 (cornell_box_quad 800x800x16)
#endif
// LEAN (bit mask) compiles scene features out of an instance by pinning the scene's counts to zero, so that every
// branch on them folds away: 1 = no environment map, 2 = no point and no disk lights, 4 = no quad lights,
// 8 = NEE and MIS on, 16 = the pcg4d generator, 32 = no material has a normal map (the tangent-frame code is left out),
// 64 = no emissive sphere.
template <class R, int MODE, bool LDS_MATS, int INTEG = INTEG_PATH, int LEAN = 0>
__global__ __launch_bounds__(kBlock, (INTEG == INTEG_PATH_MNEE && sizeof(R) == 4) ? RTX_MNEE_F32_MIN_WAVES : (MODE == MODE_FLAT ? (INTEG == INTEG_PATHPHOTON ? RTX_PHOTON_FLAT_MIN_WAVES : (LEAN & 1) ? RTX_LEAN_MIN_WAVES : (LEAN ? RTX_LEAN_ENV_MIN_WAVES : RTX_FLAT_MIN_WAVES)) : (INTEG == INTEG_PATH_MNEE ? RTX_MNEE_MIN_WAVES : (sizeof(R) == 4 ? (INTEG == INTEG_PATHPHOTON ? RTX_PHOTON_MIN_WAVES : INTEG == INTEG_WHITTED ? RTX_WHITTED_MIN_WAVES : RTX_TREE_MIN_WAVES) : RTX_F64_MIN_WAVES)))) void render_fused_kernel(DScene<R> sc, DCamera<R> cam, DRenderParams P, R* __restrict__ partial,
                                                               DCounters* __restrict__ counters, int stack_first_int) {
  if constexpr ((LEAN & 1) != 0) { sc.env.valid = 0; sc.env.importance = 0; }
  if constexpr ((LEAN & 2) != 0) { sc.n_points = 0; sc.n_disks = 0; }
  if constexpr ((LEAN & 4) != 0) { sc.n_quads = 0; }
  if constexpr ((LEAN & 8) != 0) { P.use_nee = 1; P.use_mis = 1; }
  if constexpr ((LEAN & 16) != 0) { P.rng = RNG_PCG4D; }
  if constexpr ((LEAN & 64) != 0) { sc.n_emissive = 0; }
  const DMaterial<R>* mats = sc.mats;
  if constexpr (LDS_MATS) mats = stage_materials(sc);
  const uint32_t npix = (uint32_t)P.rows * (uint32_t)P.width;
  const uint32_t n_items = P.item_list ? P.n_list : npix * (uint32_t)P.n_chunks;   // (a repair pass renders the listed items only)
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t next = blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st = lane_stack(stack_first_int);

  // Statistics: every trip adds the number of active lanes to wave-uniform counters.  A lane
  // that leaves the loop stops adding, so the lane that stays longest holds the wave's totals:
  // they are recovered with a max over the lanes at the end.
  uint32_t w_steps = 0, w_shadow = 0, w_shaded = 0, w_samples = 0;
  uint32_t extra_traces = 0, extra_shadow = 0;   // per lane: closest-hit queries made by MNEE, shadow queries of the Whitted integrator
  PathState<R> ps;
  ps.bounce_flags = 0;
  V3<R> sum = {R(0), R(0), R(0)};
  uint32_t cur = 0, xy = 0;
  int s = 0, s_end = 0;
  PathKey key = {0u, P.seed, 0u, (uint32_t)P.rng};

  auto fetch = [&]() -> bool {
    if (next >= n_items) return false;
    cur = P.item_list ? P.item_list[next] : next;
    next += stride;
    uint32_t chunk = cur / npix;
    uint32_t pix = cur - chunk * npix;
    uint32_t row_local = pix / (uint32_t)P.width;
    uint32_t x = pix - row_local * (uint32_t)P.width;
    uint32_t y = (uint32_t)P.shard_index + row_local * (uint32_t)P.shard_count;
    xy = (y << 16) | x;
    key.pixel = y * (uint32_t)P.width + x;
    s = (int)chunk * P.chunk_spp;
    s_end = s + P.chunk_spp < P.spp ? s + P.chunk_spp : P.spp;
    sum = {R(0), R(0), R(0)};
    return s < s_end;
  };

  bool active = fetch();
  if (active) { key.sample = (uint32_t)s++; path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key); }
  w_samples += (uint32_t)__popcll(__ballot(active));
  if constexpr (INTEG == INTEG_PATHPHOTON) {
    // The density estimate is computed by the whole wave (pt_photon_wave.hpp), so every lane stays in the loop until the
    // wave has no path left; a lane without a path skips the vertex and helps with the estimates of the others.
    __shared__ R cand_d2[kBlock / 64][kPhotonCand];
    __shared__ int cand_id[kBlock / 64][kPhotonCand];
    __shared__ __align__(16) R photon_table[photon_table_words<R>()];
    PhotonCand<R> pc;
    pc.d2 = cand_d2[threadIdx.x >> 6]; pc.id = cand_id[threadIdx.x >> 6];
    pc.cap = kPhotonCand; pc.d2_alt = nullptr; pc.id_alt = nullptr;
    if (P.ph_cand_cap > 0) {   // more photons per estimate than the LDS list holds: this wave's two lists in device memory
      const size_t cap = (size_t)P.ph_cand_cap, wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
      char* base = (char*)P.ph_cand + wave * cap * 2 * (sizeof(R) + sizeof(int));
      pc.d2 = (R*)base; pc.d2_alt = pc.d2 + cap;
      pc.id = (int*)(pc.d2_alt + cap); pc.id_alt = pc.id + cap;
      pc.cap = P.ph_cand_cap;
    }
    photon_stage_table<R>(P, photon_table, pc);
    __syncthreads();
    for (;;) {
      bool want = false;
      V3<R> g_pos = {R(0), R(0), R(0)}, g_n = g_pos, g_w = g_pos;
      uint32_t did = 0u;
      if (active) {
        Hit<R> rec;
        const bool hit = trace_closest<R, MODE>(sc, ps.ray, R(0.001), R(1e9), rec, st);
        did = photon_vertex(
            ps, hit, rec, sc, mats, P, key, [&](const Ray<R>& sr, R t_max) { return trace_occluded<R, MODE>(sc, sr, R(0.001), t_max, st); },
            [&](V3<R> pos, V3<R> n, V3<R> w) { want = true; g_pos = pos; g_n = n; g_w = w; });
      }
      w_steps += (uint32_t)__popcll(__ballot(active));
      w_shadow += (uint32_t)__popcll(__ballot((did & STEP_SHADOW) != 0));
      w_shaded += (uint32_t)__popcll(__ballot((did & STEP_SHADED) != 0));
      photon_gather_wave<R>(P, want, g_pos, g_n, g_w, ps.acc, pc);
      bool started = false;
      if (active && !(ps.bounce_flags & F_ALIVE)) {
        sum = sum + ps.acc;
        if (s == s_end) {
          R* o = partial + (size_t)cur * 3u;
          o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
          active = fetch();
        }
        if (active) {
          key.sample = (uint32_t)s++;
          path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key);
          started = true;
        }
      }
      w_samples += (uint32_t)__popcll(__ballot(started));
      if (!__any(active)) break;
    }
  } else {
    while (active) {
      w_steps += (uint32_t)__popcll(__ballot(1));
      const uint32_t did = path_step<R, MODE, INTEG, MODE == MODE_TREE_UV && (LEAN & 32) == 0>(ps, sc, mats, P, key, st, extra_traces, extra_shadow);
      if constexpr (MODE != MODE_FLAT && sizeof(R) == 4) { if (did & STEP_TIED) note_tied_item(P, counters, cur); }
      w_shadow += (uint32_t)__popcll(__ballot((did & STEP_SHADOW) != 0));
      w_shaded += (uint32_t)__popcll(__ballot((did & STEP_SHADED) != 0));
#ifdef RTX_PHASE_STATS
      {   // per phase: wave-trips in which some lane took it, and the lanes that did (slot 12: all trips; 13: paths that ended)
        const uint32_t bits = did | (1u << 12) | ((ps.bounce_flags & F_ALIVE) ? 0u : (1u << 13));
        for (int b = 0; b < 14; ++b) {
          const unsigned long long m = __ballot((bits >> b) & 1u);
          if ((threadIdx.x & 63) == 0 && m) { atomicAdd(&g_phase_stats[2 * b], 1ull); atomicAdd(&g_phase_stats[2 * b + 1], (unsigned long long)__popcll(m)); }
        }
      }
#endif
      bool started = false;
      if (!(ps.bounce_flags & F_ALIVE)) {
        sum = sum + ps.acc;
        if (s == s_end) {
          R* o = partial + (size_t)cur * 3u;
          o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
          active = fetch();
          if (!active) break;
        }
        key.sample = (uint32_t)s++;
        path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key);
        started = true;
      }
      w_samples += (uint32_t)__popcll(__ballot(started));   // outside the divergent branch: counters must be added in uniform flow
    }
  }
  w_samples = wave_max(w_samples); w_steps = wave_max(w_steps); w_shadow = wave_max(w_shadow); w_shaded = wave_max(w_shaded);
  if constexpr (INTEG != INTEG_PATH) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      extra_traces += (uint32_t)__shfl_xor((int)extra_traces, off, 64);
      extra_shadow += (uint32_t)__shfl_xor((int)extra_shadow, off, 64);
    }
    w_steps += extra_traces;
    w_shadow += extra_shadow;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counters->samples, (unsigned long long)w_samples);
    atomicAdd(&counters->closest, (unsigned long long)w_steps);
    atomicAdd(&counters->shadow, (unsigned long long)w_shadow);
    atomicAdd(&counters->segments, (unsigned long long)w_shaded);
  }
}

// ------------------------------------------------------------------------------------------
// Manifold NEE with the solves of a wave BATCHED (INTEG_PATH_MNEE; both precisions).
// A manifold solve costs a hundred path vertices (20 Newton iterations of double-precision algebra and two closest-hit queries
// per chain vertex), and only about half of the Lambertian vertices ask for one (the seed chain through a glass sphere is tried
// when the sphere lies towards the light): solved on the spot, as render_fused_kernel does, a solve runs with 30 of a wave's 64
// lanes.  Here a vertex only looks for its first solve (pt_mnee.hpp mnee_next_attempt: the light's sample, the straight chain, the
// seed conditions - cheap) and NOTES it (point, weight = albedo x throughput / pi, sample, bounce, light and attempt: a "task"),
// and its lane stops; the other lanes go on with their paths - cheap trips - until kMneeBatch lanes of the wave hold a task,
// and then the wave runs one solve for each of them at once; a task whose solve fails and which has another attempt left stays.  Nothing moves between lanes.  A task's contribution is a function of the task alone and goes to
// an accumulator of its own (sum_mnee), added to the chunk sum when the work item ends, so the film does not depend on when the
// wave happened to solve: launch- and shard-invariant like the other kernels'.  The solver also runs with little else alive in
// registers (on the spot, the whole vertex is).
// ------------------------------------------------------------------------------------------
#ifndef RTX_MNEE_BATCH
#define RTX_MNEE_BATCH 56   // lanes of 64 that must hold a task (1080p Cornell box, 16 spp: 32: 248 ms, 48: 211, 56: 206, 62: 212; solved on the spot: 270)
#endif
#ifndef RTX_MNEE_BATCHED_MIN_WAVES
#define RTX_MNEE_BATCHED_MIN_WAVES 3   // (the same frame: 3: 197.6 ms, 4: 205.4, 5: 282.5)
#endif
constexpr int kMneeBatch = RTX_MNEE_BATCH;
template <class R, int MODE, bool LDS_MATS, int LEAN>
__global__ __launch_bounds__(kBlock, (sizeof(R) == 4 ? RTX_MNEE_BATCHED_MIN_WAVES : RTX_MNEE_MIN_WAVES)) void render_fused_mnee_batched_kernel(DScene<R> sc, DCamera<R> cam, DRenderParams P, R* __restrict__ partial,
                                                                                                                                    DCounters* __restrict__ counters, int stack_first_int) {
  if constexpr ((LEAN & 1) != 0) { sc.env.valid = 0; sc.env.importance = 0; }
  if constexpr ((LEAN & 2) != 0) { sc.n_points = 0; sc.n_disks = 0; }
  if constexpr ((LEAN & 4) != 0) { sc.n_quads = 0; }
  if constexpr ((LEAN & 8) != 0) { P.use_nee = 1; P.use_mis = 1; }
  if constexpr ((LEAN & 16) != 0) { P.rng = RNG_PCG4D; }
  const DMaterial<R>* mats = sc.mats;
  if constexpr (LDS_MATS) mats = stage_materials(sc);
  const uint32_t npix = (uint32_t)P.rows * (uint32_t)P.width;
  const uint32_t n_items = P.item_list ? P.n_list : npix * (uint32_t)P.n_chunks;
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t next = blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st = lane_stack(stack_first_int);
  uint32_t w_steps = 0, w_shadow = 0, w_shaded = 0, w_samples = 0, extra_traces = 0;
  PathState<R> ps;
  ps.ray = {{R(0), R(0), R(0)}, {R(0), R(0), R(1)}}; ps.thr = {R(0), R(0), R(0)}; ps.acc = {R(0), R(0), R(0)}; ps.bsdf_pdf = R(0);
  ps.bounce_flags = 0;
  V3<R> sum = {R(0), R(0), R(0)}, sum_mnee = {R(0), R(0), R(0)};
  uint32_t cur = 0, xy = 0;
  int s = 0, s_end = 0;
  PathKey key = {0u, P.seed, 0u, (uint32_t)P.rng};
  // the lane's task
  bool pending = false, finishing = false;   // finishing: the work item's last path has ended, its chunk sum waits for the task
  V3<R> task_p = {R(0), R(0), R(0)}, task_w = {R(0), R(0), R(0)};
  uint32_t task_sample = 0u;
  int task_bounce = 0, task_j = 0, task_i = -1;   // the attempt to run: light j, chain i (pt_mnee.hpp)
  uint32_t quiet_queries = 0u;

  auto fetch = [&]() -> bool {
    if (next >= n_items) return false;
    cur = P.item_list ? P.item_list[next] : next;
    next += stride;
    uint32_t chunk = cur / npix;
    uint32_t pix = cur - chunk * npix;
    uint32_t row_local = pix / (uint32_t)P.width;
    uint32_t x = pix - row_local * (uint32_t)P.width;
    uint32_t y = (uint32_t)P.shard_index + row_local * (uint32_t)P.shard_count;
    xy = (y << 16) | x;
    key.pixel = y * (uint32_t)P.width + x;
    s = (int)chunk * P.chunk_spp;
    s_end = s + P.chunk_spp < P.spp ? s + P.chunk_spp : P.spp;
    sum = {R(0), R(0), R(0)}; sum_mnee = {R(0), R(0), R(0)};
    return s < s_end;
  };
  auto store_item = [&]() {
    R* o = partial + (size_t)cur * 3u;
    o[0] = sum.x + sum_mnee.x; o[1] = sum.y + sum_mnee.y; o[2] = sum.z + sum_mnee.z;
  };

  bool active = fetch();
  if (active) { key.sample = (uint32_t)s++; path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key); }
  w_samples += (uint32_t)__popcll(__ballot(active));
  for (;;) {
    // 1. solve, when enough lanes hold a task or no lane can do anything else
    const int n_pending = __popcll(__ballot(pending)), n_active = __popcll(__ballot(active));
    if (n_active == 0) break;
    const bool someone_can_go = __ballot(active && !pending) != 0ull;
    if (n_pending > 0 && (n_pending * 64 >= kMneeBatch * n_active || !someone_can_go)) {
      bool started = false;
      if (pending) {
        PathKey tk = key;
        tk.sample = task_sample;
        MneeTracer<R, MODE> trace = {sc, st, extra_traces}, quiet = {sc, st, quiet_queries};
        const V3<double> hp = v3_to<double>(task_p);
        V3<double> c;
        if (mnee_run_attempt(sc, mats, hp, tk, task_bounce, task_j, task_i, c, trace, quiet)) {   // MNEE::evaluate (ref: engine/renderer.cpp:716-732), one solve of it
          sum_mnee = sum_mnee + v3_to<R>(c) * task_w;
          ++task_j; task_i = -1;
        } else {
          ++task_i;
        }
        pending = mnee_next_attempt(sc, mats, hp, tk, task_bounce, task_j, task_i, trace);
        if (!pending && finishing) {
          finishing = false;
          store_item();
          active = fetch();
          if (active) { key.sample = (uint32_t)s++; path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key); started = true; }
        }
      }
      w_samples += (uint32_t)__popcll(__ballot(started));
      continue;
    }
    // 2. one vertex for every lane that has a path and no task
    const bool go = active && !pending;
    uint32_t did = 0u;
    bool started = false;
    if (go) {
      Hit<R> rec;
      const bool hit = trace_closest<R, MODE>(sc, ps.ray, R(0.001), R(1e9), rec, st);
      if constexpr (MODE != MODE_FLAT) { if (hit && rec.tied) note_tied_item(P, counters, cur); }
      did = path_vertex_impl<R, true, MODE == MODE_TREE_UV && (LEAN & 32) == 0>(
          ps, hit, rec, sc, mats, P, key,
          [&](const Ray<R>& sr, R t_max, V3<R> contrib) {
            const bool visible = !trace_occluded<R, MODE, true>(sc, sr, R(0.001), t_max, st);
            if (visible) ps.acc = ps.acc + contrib;
            return visible;
          },
          [&](V3<R> hp, V3<R> surf, int bounce) {
            task_p = hp; task_w = (surf * ps.thr) * (R(1) / R(3.14159265358979323846));
            task_sample = key.sample; task_bounce = bounce; task_j = 0; task_i = -1;
            MneeTracer<R, MODE> trace = {sc, st, extra_traces};
            pending = mnee_next_attempt(sc, mats, v3_to<double>(hp), key, bounce, task_j, task_i, trace);
          });
      if (!(ps.bounce_flags & F_ALIVE)) {
        sum = sum + ps.acc;
        bool more = true;
        if (s == s_end) {
          if (pending) { finishing = true; more = false; }
          else { store_item(); active = fetch(); more = active; }
        }
        if (more) { key.sample = (uint32_t)s++; path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key); started = true; }
      }
    }
    w_steps += (uint32_t)__popcll(__ballot(go));
    w_shadow += (uint32_t)__popcll(__ballot((did & STEP_SHADOW) != 0));
    w_shaded += (uint32_t)__popcll(__ballot((did & STEP_SHADED) != 0));
    w_samples += (uint32_t)__popcll(__ballot(started));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) extra_traces += (uint32_t)__shfl_xor((int)extra_traces, off, 64);
  if ((threadIdx.x & 63) == 0) {   // (every lane stays to the end: the counters are the same in all of them)
    atomicAdd(&counters->samples, (unsigned long long)w_samples);
    atomicAdd(&counters->closest, (unsigned long long)(w_steps + extra_traces));
    atomicAdd(&counters->shadow, (unsigned long long)w_shadow);
    atomicAdd(&counters->segments, (unsigned long long)w_shaded);
  }
}

// ------------------------------------------------------------------------------------------
// The same path tracer with MATERIAL-COHERENT waves (experiment, RTX_FUSED_REGROUP=1; flat scenes, plain path tracing).
// After the closest-hit query of a trip every lane knows what its vertex needs next: a Lambertian vertex (light sample, shadow
// query, cosine scatter), a specular one (mirror / glass), the end of its path (miss, light: the next sample starts), or
// nothing (no work left).  The 256 lanes of the workgroup then trade places through LDS - a stable partition by that class, the
// whole lane state moves: path, hit, work item - so that a wave holds one class where the counts allow it, and the shading code of
// the other classes is skipped by whole waves instead of being paid under a mask.  Which lane computes a path changes, what is
// computed does not: films equal render_fused_kernel's to the rounding of a separately compiled kernel.
// Measured on the 1080p Cornell box (profiles/r03_c2_regroup/): the exchange costs more than the skipped code saves; the
// plain kernel stays the one that is launched.
// ------------------------------------------------------------------------------------------
constexpr int kRegroupWords = 25;
template <int LEAN>
__global__ __launch_bounds__(kBlock, 6) void render_fused_regroup_kernel(DScene<float> sc, DCamera<float> cam, DRenderParams P, float* __restrict__ partial,
                                                                                         DCounters* __restrict__ counters, int regroup_min_minority) {
  using R = float;
  if constexpr ((LEAN & 1) != 0) { sc.env.valid = 0; sc.env.importance = 0; }
  if constexpr ((LEAN & 2) != 0) { sc.n_points = 0; sc.n_disks = 0; }
  if constexpr ((LEAN & 4) != 0) { sc.n_quads = 0; }
  if constexpr ((LEAN & 8) != 0) { P.use_nee = 1; P.use_mis = 1; }
  if constexpr ((LEAN & 16) != 0) { P.rng = RNG_PCG4D; }
  if constexpr ((LEAN & 64) != 0) { sc.n_emissive = 0; }
  const DMaterial<R>* mats = stage_materials(sc);
  __shared__ uint32_t xch[kRegroupWords][kBlock];
  __shared__ uint32_t cnt[kBlock / 64][4];
  const uint32_t npix = (uint32_t)P.rows * (uint32_t)P.width;
  const uint32_t n_items = npix * (uint32_t)P.n_chunks;
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t next = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  uint32_t w_steps = 0, w_shadow = 0, w_shaded = 0, w_samples = 0;
  PathState<R> ps;
  ps.ray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}}; ps.thr = {0.f, 0.f, 0.f}; ps.acc = {0.f, 0.f, 0.f}; ps.bsdf_pdf = 0.f;
  ps.bounce_flags = 0;
  V3<R> sum = {R(0), R(0), R(0)};
  uint32_t cur = 0, xy = 0, pixel = 0;
  int s = 0, s_end = 0;
  auto fetch = [&]() -> bool {
    if (next >= n_items) return false;
    cur = next;
    next += stride;
    uint32_t chunk = cur / npix;
    uint32_t pix = cur - chunk * npix;
    uint32_t row_local = pix / (uint32_t)P.width;
    uint32_t x = pix - row_local * (uint32_t)P.width;
    uint32_t y = (uint32_t)P.shard_index + row_local * (uint32_t)P.shard_count;
    xy = (y << 16) | x;
    pixel = y * (uint32_t)P.width + x;
    s = (int)chunk * P.chunk_spp;
    s_end = s + P.chunk_spp < P.spp ? s + P.chunk_spp : P.spp;
    sum = {R(0), R(0), R(0)};
    return s < s_end;
  };
  bool active = fetch();
  if (active) { PathKey key = {pixel, P.seed, (uint32_t)s++, (uint32_t)P.rng}; path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key); }
  w_samples += (uint32_t)__popcll(__ballot(active));
  for (;;) {
    // 1. closest hit: the same work for every lane with a path
    float t = 0.f;
    int k = -1;
    if (active) flat_closest_tk(sc, ps.ray, 0.001f, 1e9f, t, k);
    w_steps += (uint32_t)__popcll(__ballot(active));
    // 2. what the vertex needs: 0 Lambertian, 1 specular, 2 the path ends here (miss, light), 3 no path
    int cls = 3;
    if (active) {
      cls = 2;
      if (k >= 0) {
        const int mat = sc.flat[k].mat;
        if (mat >= 0) cls = mats[mat].type == MAT_LAMBERTIAN ? 0 : 1;
      }
    }
    // 3. stable partition of the workgroup's lanes by class
    const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
    if (lane == 0) { cnt[wave][0] = (uint32_t)__popcll(m0); cnt[wave][1] = (uint32_t)__popcll(m1); cnt[wave][2] = (uint32_t)__popcll(m2); cnt[wave][3] = (uint32_t)__popcll(m3); }
    __syncthreads();
    uint32_t tot[4] = {0u, 0u, 0u, 0u}, before = 0u;   // lanes of my class in earlier waves
    int mixed_waves = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) {
      const uint32_t c0 = cnt[w][0], c1 = cnt[w][1], c2 = cnt[w][2], c3 = cnt[w][3];
      tot[0] += c0; tot[1] += c1; tot[2] += c2; tot[3] += c3;
      if (w < wave) before += cls == 0 ? c0 : cls == 1 ? c1 : cls == 2 ? c2 : c3;
      const uint32_t major = c0 > c1 ? (c0 > c2 ? (c0 > c3 ? c0 : c3) : (c2 > c3 ? c2 : c3)) : (c1 > c2 ? (c1 > c3 ? c1 : c3) : (c2 > c3 ? c2 : c3));
      mixed_waves += (64u - major) >= (uint32_t)regroup_min_minority ? 1 : 0;
    }
    if (tot[3] == (uint32_t)kBlock) break;   // no lane of the workgroup has a path left (the same answer in every lane)
    if (mixed_waves > 0) {                   // (the same in every lane of the workgroup)
      const unsigned long long mine = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
      const uint32_t base = cls == 0 ? 0u : cls == 1 ? tot[0] : cls == 2 ? tot[0] + tot[1] : tot[0] + tot[1] + tot[2];
      const uint32_t dest = base + before + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
      const uint32_t out[kRegroupWords] = {__float_as_uint(ps.ray.o.x), __float_as_uint(ps.ray.o.y), __float_as_uint(ps.ray.o.z), __float_as_uint(ps.ray.d.x),
                                           __float_as_uint(ps.ray.d.y), __float_as_uint(ps.ray.d.z), __float_as_uint(ps.thr.x), __float_as_uint(ps.thr.y),
                                           __float_as_uint(ps.thr.z), __float_as_uint(ps.acc.x), __float_as_uint(ps.acc.y), __float_as_uint(ps.acc.z),
                                           __float_as_uint(ps.bsdf_pdf), ps.bounce_flags, __float_as_uint(t), (uint32_t)k, __float_as_uint(sum.x),
                                           __float_as_uint(sum.y), __float_as_uint(sum.z), cur, xy, (uint32_t)s, (uint32_t)s_end, next, (uint32_t)cls};
#pragma unroll
      for (int i = 0; i < kRegroupWords; ++i) xch[i][dest] = out[i];
      __syncthreads();
      uint32_t in[kRegroupWords];
#pragma unroll
      for (int i = 0; i < kRegroupWords; ++i) in[i] = xch[i][threadIdx.x];
      ps.ray.o = {__uint_as_float(in[0]), __uint_as_float(in[1]), __uint_as_float(in[2])};
      ps.ray.d = {__uint_as_float(in[3]), __uint_as_float(in[4]), __uint_as_float(in[5])};
      ps.thr = {__uint_as_float(in[6]), __uint_as_float(in[7]), __uint_as_float(in[8])};
      ps.acc = {__uint_as_float(in[9]), __uint_as_float(in[10]), __uint_as_float(in[11])};
      ps.bsdf_pdf = __uint_as_float(in[12]); ps.bounce_flags = in[13];
      t = __uint_as_float(in[14]); k = (int)in[15];
      sum = {__uint_as_float(in[16]), __uint_as_float(in[17]), __uint_as_float(in[18])};
      cur = in[19]; xy = in[20]; s = (int)in[21]; s_end = (int)in[22]; next = in[23]; cls = (int)in[24];
      pixel = (xy >> 16) * (uint32_t)P.width + (xy & 0xffffu);
      active = cls != 3;
    }
    __syncthreads();   // cnt and xch are free for the next trip
    // 4. shade, as path_step does after its trace
    uint32_t did = 0;
    if (active) {
      PathKey key = {pixel, P.seed, (uint32_t)(s - 1), (uint32_t)P.rng};
      Hit<R> rec;
      const bool hit = flat_hit_from(sc, ps.ray, t, k, rec);
      did = path_vertex<R, false>(ps, hit, rec, sc, mats, P, key, [&](const Ray<R>& sr, R t_max, V3<R> contrib) {
        if (!flat_occluded<true>(sc, sr, 0.001f, t_max)) ps.acc = ps.acc + contrib;
      });
    }
    w_shadow += (uint32_t)__popcll(__ballot((did & STEP_SHADOW) != 0));
    w_shaded += (uint32_t)__popcll(__ballot((did & STEP_SHADED) != 0));
    bool started = false;
    if (active && !(ps.bounce_flags & F_ALIVE)) {
      sum = sum + ps.acc;
      if (s == s_end) {
        R* o = partial + (size_t)cur * 3u;
        o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
        active = fetch();
      }
      if (active) {
        PathKey key = {pixel, P.seed, (uint32_t)s++, (uint32_t)P.rng};
        path_begin(ps, cam, P, (int)(xy & 0xffffu), (int)(xy >> 16), key);
        started = true;
      }
    }
    w_samples += (uint32_t)__popcll(__ballot(started));
  }
  if (lane == 0) {
    atomicAdd(&counters->samples, (unsigned long long)w_samples);
    atomicAdd(&counters->closest, (unsigned long long)w_steps);
    atomicAdd(&counters->shadow, (unsigned long long)w_shadow);
    atomicAdd(&counters->segments, (unsigned long long)w_shaded);
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

// Closest hit of a ray batch through the structure the integrator walks.
template <class R, int MODE>
__global__ __launch_bounds__(kBlock) void trace_fast_kernel(DScene<R> sc, const double* __restrict__ rays8, unsigned long long n, DHitOut* __restrict__ out) {
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
  const double* q = rays8 + i * 8ull;
  Ray<R> r = {{R(q[0]), R(q[1]), R(q[2])}, {R(q[3]), R(q[4]), R(q[5])}};
  Hit<R> h;
  bool ok = trace_closest<R, MODE>(sc, r, R(q[6]), R(q[7]), h, lane_stack(0));
  DHitOut o;
  o.hit = ok; o.prim_type = ok ? h.type : -1; o.prim_index = ok ? h.idx : -1;
  o.material_id = ok ? h.mat : 0; o.front_face = ok ? h.front : 0; o.pad = 0;
  o.t = ok ? (double)h.t : 0.0;
  o.point[0] = ok ? (double)h.p.x : 0.0; o.point[1] = ok ? (double)h.p.y : 0.0; o.point[2] = ok ? (double)h.p.z : 0.0;
  o.normal[0] = ok ? (double)h.n.x : 0.0; o.normal[1] = ok ? (double)h.n.y : 0.0; o.normal[2] = ok ? (double)h.n.z : 0.0;
  o.u = ok ? (double)h.u : 0.0; o.v = ok ? (double)h.v : 0.0;
  out[i] = o;
  }
}

}  // namespace rtx
