// Wavefront (queue-based) pipeline, single precision.
//
// A pool of path slots lives in HBM as structure-of-arrays.  wf_start gives every slot its first
// (pixel, chunk) work item and camera ray; after that one turn advances every live path by one
// vertex with three kernels:
//
//   wf_extend           every queued slot's ray is traced to its closest hit (scenes that walk a tree:
//                       to its closest bounded primitive; wf_shade tests the planes and lights)
//   wf_shade            the vertex is shaded (emission, MIS, light sample, scatter, roulette);
//                       light samples that need a visibility test are compacted into the
//                       shadow-ray queue; a slot whose path ended adds it to its chunk sum and
//                       starts its next sample (or takes its next work item) on the spot; slots
//                       with a ray to trace are compacted into the next ray queue
//   wf_connect          shadow rays are traced (any hit); unoccluded ones add their contribution
//                       to the chunk sum of their slot
//
// Queues are compacted with ballot + prefix popcount and ONE atomic per workgroup.
// The traversal kernels are persistent: a lane carries only a ray and a candidate and takes its
// next ray from the queue when its walk ends, so a wave is not held back by its longest walk; that
// is what pays for the queue traffic on scenes that walk a tree.
// Per-slot arithmetic and sample order equal the fused kernel's, so films agree to rounding.
#include <cstring>   // rocPRIM's headers use memset without including it

#include <rocprim/rocprim.hpp>

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
  int32_t *hit_id, *hit_uv;          // type << kIdBits | idx (or -1);  uv_kind << kIdBits | uv_idx
  uint32_t* q[2];
  uint32_t* sh_slot;
  float* sh[7];                       // origin, direction, t_max
  float* sh_c[3];
  // Ray sorting (scenes that walk a tree): keys and queue entries before and after the radix sort.  The traversal kernels take
  // their rays in the order of order_out, so that the lanes of a wave - and the waves in flight at any moment - start in the same
  // corner of the scene and fetch the same node and leaf records.
  uint32_t *key_in, *key_out, *order_in, *order_out;
  uint32_t n_slots;
};

// Queue lengths and the cursors the persistent kernels hand their entries out from.  No kernel runs between wf_connect and the
// wf_extend after it, so each has a cursor of its own; the counters rotate so that the one a turn fills was cleared a turn earlier
// (by wf_shade, after every kernel that read it has finished).
// A stored hit packs the primitive's type above its index within that type: scenes with 2^28 or more primitives of one type are
// refused by wavefront_render.
constexpr int kIdBits = 28;
constexpr int kIdMask = (1 << kIdBits) - 1;

// Every counter has a 128-byte line of its own: a queue length that one kernel's workgroups read must not share a line with a
// counter the same kernel's atomics hammer (with all seven in one line, wf_shade ran at one workgroup per 100 ns whatever the
// workgroup's size).
struct alignas(128) WfCounter { uint32_t v; uint32_t pad[31]; };
struct WfControl {
  WfCounter n_q_[3], n_shadow_[2], cursor_rays_, cursor_shadow_;
};
// One turn: wf_shade reads the ray queue A.q[q_in] of n_q[n_in] entries, fills A.q[q_out] / n_q[n_out] and clears n_q[n_zero];
// its shadow rays are counted in n_shadow[sh] (it clears the other one).
struct WfTurn { int q_in, q_out, n_in, n_out, n_zero, sh; };
// Statistics of the streaming kernels.  Every workgroup adds its counts to the bank its index selects: atomics on ONE address
// serialise at the L2 (a few nanoseconds each, four per workgroup and turn), atomics on different lines do not.  wf_stats_fold
// adds the banks up at the end of a render.
struct alignas(64) WfStatBank { unsigned long long segments, shadow, closest, samples, pad[4]; };
constexpr int kStatBanks = 64;
struct WfShared { WfControl ctl; WfStatBank bank[kStatBanks]; };

// A persistent traversal wave runs its leaf step when at least this many lanes have a leaf record pending (or none has a node).
// With one primitive per leaf record the step is short, and waiting for company does not pay: measured on the 10 M-triangle scene
// (closest-hit kernel, 64 spp): 1: 223.7 ms, 4: 228.0, 8: 228.5.
#ifndef RTX_LEAF_BATCH
#define RTX_LEAF_BATCH 1
#endif
constexpr int kLeafBatch = RTX_LEAF_BATCH;
#ifndef RTX_WALK_MIN_WAVES
#define RTX_WALK_MIN_WAVES 1   // waves per SIMD the traversal kernels are compiled for (caps their registers): left alone they take 73 VGPRs and run 6; capped at 72 (7 waves) or 64 (8) they are no faster (240 and 243 ms against 238 for the closest-hit walk of the 10 M-triangle frame)
#endif


// The host reads the length of the ray queue every eighth turn (the kernels of the turns in between are enqueued without waiting),
// every turn once the queue is short - and every turn while it is long: the streaming kernels' grids follow the last length read,
// and a turn of millions of rays hides the round trip, where workgroups launched for entries that no longer exist would not be free
// (10 M triangles, 1080p x 64 spp: 362.3 -> 360.3 ms).
constexpr uint32_t kSyncEveryTurnAbove = 4u << 20;
constexpr uint32_t kOverlapBelowSlots = 48u << 20;   // launches of fewer path slots run the two walks of a turn side by side (wavefront_render)
#ifndef RTX_POOL_MULT
#define RTX_POOL_MULT 256   // path slots per resident lane (180 B each: 24 GB at most on 256 CUs); 10 M triangles at 1080p x 64 spp (133 M work items of one sample): 16: 440 ms, 32: 409, 64: 389, 128: 374, 256 (every work item in flight): 364
#endif
constexpr int kWfArrays = 6 + 3 + 3 + 3 + 1 + 7 + 3 + 2 + 2 + 1 + 7 + 3 + 4;   // 4-byte arrays of n_slots entries

// Queue appends of the streaming kernels (wf_start, wf_shade): every thread of the workgroup reaches the call (they synchronise
// the workgroup), ONE returning atomic per workgroup and queue.  Returning atomics on one address serialise at about 11 ns each
// (scripts/atomic_rate.hip), which at one atomic per wave was most of these kernels' run time.  With every counter on a line of
// its own (WfControl) the workgroup size no longer matters (10 M-triangle frame: 1024 threads 398 ms, 512: 395, 256: 398; with
// all counters in one line: 410 / 490 / 613); it stays at the largest.
#ifndef RTX_STREAM_BLOCK
#define RTX_STREAM_BLOCK 1024
#endif
constexpr int kStreamBlock = RTX_STREAM_BLOCK;
struct BlockScratch { uint32_t cnt[kStreamBlock / 64]; uint32_t cnt_b[kStreamBlock / 64]; uint32_t base, base_b; uint32_t sum[4]; };

// Entry of the queue for each thread with `want` (0xffffffff for the others).
RTX_DEV uint32_t block_append(BlockScratch& bs, uint32_t* counter, bool want) {
  const unsigned long long mask = __ballot(want);
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  __syncthreads();   // a previous use of bs is over
  if (lane == 0) bs.cnt[wave] = (uint32_t)__popcll(mask);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) total += bs.cnt[k];
    bs.base = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t at = bs.base;
  for (int k = 0; k < wave; ++k) at += bs.cnt[k];
  at += (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  return want ? at : 0xffffffffu;
}

// Two queues at once (one wait for both atomics): entries for the threads with want_a / want_b, 0xffffffff for the others.
RTX_DEV void block_append2(BlockScratch& bs, uint32_t* counter_a, bool want_a, uint32_t& at_a, uint32_t* counter_b, bool want_b, uint32_t& at_b) {
  const unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6), waves = (int)(blockDim.x >> 6);
  __syncthreads();   // a previous use of bs is over
  if (lane == 0) { bs.cnt[wave] = (uint32_t)__popcll(ma); bs.cnt_b[wave] = (uint32_t)__popcll(mb); }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int k = 0; k < waves; ++k) total += bs.cnt[k];
    bs.base = total ? atomicAdd(counter_a, total) : 0u;
  } else if (threadIdx.x == 64) {
    uint32_t total = 0;
    for (int k = 0; k < waves; ++k) total += bs.cnt_b[k];
    bs.base_b = total ? atomicAdd(counter_b, total) : 0u;
  }
  __syncthreads();
  uint32_t a = bs.base, b = bs.base_b;
  for (int k = 0; k < wave; ++k) { a += bs.cnt[k]; b += bs.cnt_b[k]; }
  a += (uint32_t)__popcll(ma & ((1ull << lane) - 1ull));
  b += (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
  at_a = want_a ? a : 0xffffffffu;
  at_b = want_b ? b : 0xffffffffu;
}

// The workgroup's counts of threads with each of four flags go to its statistics bank: segments, shadow, closest, samples.
RTX_DEV void block_count4(BlockScratch& bs, WfStatBank* banks, bool segments, bool shadow, bool closest, bool samples) {
  const unsigned long long m[4] = {__ballot(segments), __ballot(shadow), __ballot(closest), __ballot(samples)};
  __syncthreads();
  if (threadIdx.x < 4) bs.sum[threadIdx.x] = 0u;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (m[k]) atomicAdd(&bs.sum[k], (uint32_t)__popcll(m[k]));
  }
  __syncthreads();
  if (threadIdx.x < 4 && bs.sum[threadIdx.x]) atomicAdd(&banks[blockIdx.x & (kStatBanks - 1)].segments + threadIdx.x, (unsigned long long)bs.sum[threadIdx.x]);
}

__global__ __launch_bounds__(64) void wf_stats_fold(const WfStatBank* __restrict__ banks, DCounters* counters) {
  unsigned long long v[4] = {banks[threadIdx.x].segments, banks[threadIdx.x].shadow, banks[threadIdx.x].closest, banks[threadIdx.x].samples};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (int off = 32; off > 0; off >>= 1) v[k] += (unsigned long long)__shfl_xor((long long)v[k], off, 64);
  if (threadIdx.x == 0) { counters->segments += v[0]; counters->shadow += v[1]; counters->closest += v[2]; counters->samples += v[3]; }
}

// (pixel, chunk) work item `item` of a render: where its pixel lies and which samples it covers.
struct WorkItem { uint32_t xy, pixel, s_begin, s_end, id; };   // id: the item's number in the frame (where its chunk sum goes)
RTX_DEV uint32_t work_items(const DRenderParams& P) { return P.item_list ? P.n_list : (uint32_t)P.rows * (uint32_t)P.width * (uint32_t)P.n_chunks; }
// The index-th work item of the launch: item `index` of the frame, or, in a repair pass, the index-th item of its list.
RTX_DEV WorkItem work_item(const DRenderParams& P, uint32_t index) {
  const uint32_t item = P.item_list ? P.item_list[index] : index;
  const uint32_t npix = (uint32_t)P.rows * (uint32_t)P.width;
  const uint32_t chunk = item / npix;
  const uint32_t pix = item - chunk * npix;
  const uint32_t row_local = pix / (uint32_t)P.width;
  const uint32_t x = pix - row_local * (uint32_t)P.width;
  const uint32_t y = (uint32_t)P.shard_index + row_local * (uint32_t)P.shard_count;
  const uint32_t s = chunk * (uint32_t)P.chunk_spp;
  return {(y << 16) | x, y * (uint32_t)P.width + x, s, s + (uint32_t)P.chunk_spp < (uint32_t)P.spp ? s + (uint32_t)P.chunk_spp : (uint32_t)P.spp, item};
}

// Slot i takes work item i (a render has at least as many items as slots) and starts its first sample; all slots go into ray
// queue 0.
__global__ __launch_bounds__(kStreamBlock) void wf_start(DCamera<float> cam, DRenderParams P, WfArrays A, WfControl* ctl, WfStatBank* banks) {
  const uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ BlockScratch bs;
  const bool live = sl < A.n_slots;
  if (live) {
    const WorkItem it = work_item(P, sl);
    A.cur[sl] = it.id; A.next[sl] = sl + A.n_slots; A.xy[sl] = it.xy; A.pixel[sl] = it.pixel; A.s_end[sl] = it.s_end;
    PathState<float> ps;
    PathKey key = {it.pixel, P.seed, it.s_begin, (uint32_t)P.rng};
    path_begin(ps, cam, P, (int)(it.xy & 0xffffu), (int)(it.xy >> 16), key);
    A.sample[sl] = it.s_begin + 1u;     // index of the NEXT sample; the current one is sample - 1
    A.sum[0][sl] = 0.f; A.sum[1][sl] = 0.f; A.sum[2][sl] = 0.f;
    A.ray[0][sl] = ps.ray.o.x; A.ray[1][sl] = ps.ray.o.y; A.ray[2][sl] = ps.ray.o.z;
    A.ray[3][sl] = ps.ray.d.x; A.ray[4][sl] = ps.ray.d.y; A.ray[5][sl] = ps.ray.d.z;
    A.thr[0][sl] = 1.f; A.thr[1][sl] = 1.f; A.thr[2][sl] = 1.f;
    A.acc[0][sl] = 0.f; A.acc[1][sl] = 0.f; A.acc[2][sl] = 0.f;
    A.bsdf_pdf[sl] = 0.f;
    A.bflags[sl] = F_ALIVE;
  }
  const uint32_t at = block_append(bs, &ctl->n_q_[0].v, live);
  if (live) A.q[0][at] = sl;
  block_count4(bs, banks, false, false, live, live);
}

// Sort key of a ray: the Morton code of its origin's cell in a 128^3 grid over the tree's bounding box (21 bits) above the
// octant of its direction (3 bits).  Entries past the queue's end get the largest key and stay at the end (the sort is stable).
RTX_DEV uint32_t spread7(uint32_t v) {   // 7 bits -> every third bit
  v &= 0x7fu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
RTX_DEV uint32_t ray_sort_key(const DScene<float>& sc, V3<float> o, V3<float> d) {
  const float gx = fminf(fmaxf((o.x - sc.bb_lo[0]) * sc.bb_scale[0], 0.f), 127.f), gy = fminf(fmaxf((o.y - sc.bb_lo[1]) * sc.bb_scale[1], 0.f), 127.f),
              gz = fminf(fmaxf((o.z - sc.bb_lo[2]) * sc.bb_scale[2], 0.f), 127.f);
  const uint32_t m = (spread7((uint32_t)gx) << 2) | (spread7((uint32_t)gy) << 1) | spread7((uint32_t)gz);
  const uint32_t oct = (d.x < 0.f ? 4u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 1u : 0u);
  return (m << 3) | oct;
}
constexpr unsigned kSortKeyBits = 24;

// shadow = 0: the ray queue q (entries = path slots); shadow = 1: the shadow-ray queue (entries = its own indices)
__global__ __launch_bounds__(256) void wf_sort_keys(DScene<float> sc, WfArrays A, const WfControl* __restrict__ ctl, int q /* ray queue array */, int nq /* its counter, or the shadow counter */, int shadow, uint32_t bound) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  const uint32_t n = shadow ? ctl->n_shadow_[nq].v : ctl->n_q_[nq].v;
  uint32_t key = 0xffffffffu, entry = 0u;
  if (i < n) {
    if (shadow) {
      entry = i;
      key = ray_sort_key(sc, V3<float>{A.sh[0][i], A.sh[1][i], A.sh[2][i]}, V3<float>{A.sh[3][i], A.sh[4][i], A.sh[5][i]});
    } else {
      entry = A.q[q][i];
      key = ray_sort_key(sc, V3<float>{A.ray[0][entry], A.ray[1][entry], A.ray[2][entry]}, V3<float>{A.ray[3][entry], A.ray[4][entry], A.ray[5][entry]});
    }
  }
  A.key_in[i] = key; A.order_in[i] = entry;
}

// Material queues (experiment, RTX_MATERIAL_SORT=1): the ray queue wf_shade is about to take, ordered by what its vertices need -
// 0 Lambertian, 1 metal, 2 dielectric, 3 no bounded hit (a plane, a light or nothing: wf_shade finds out) - so that the waves of the
// shade kernel run one material branch each.  The key of an entry comes from its stored hit (one fetch of the primitive's
// material index, one of the material's type); entries past the queue's end sort last.
constexpr unsigned kMaterialKeyBits = 3;
__global__ __launch_bounds__(256) void wf_material_keys(DScene<float> sc, WfArrays A, const WfControl* __restrict__ ctl, int q, int nq, uint32_t bound) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bound) return;
  const uint32_t n = ctl->n_q_[nq].v;
  uint32_t key = 7u, entry = 0u;
  if (i < n) {
    entry = A.q[q][i];
    const int id = A.hit_id[entry];
    key = 3u;
    if (id >= 0) {
      const int type = id >> kIdBits, idx = id & kIdMask;
      const int mat = type == PRIM_SPHERE ? sc.spheres[idx].mat : type == PRIM_TRIANGLE ? sc.tris[idx].mat : type == PRIM_BOX ? sc.boxes[idx].mat : -1;
      if (mat >= 0) key = (uint32_t)sc.mats[mat].type;
    }
  }
  A.key_in[i] = key; A.order_in[i] = entry;
}

// Both traversal kernels are persistent: a launch holds as many lanes as the chip keeps
// resident, every lane walks one ray at a time in resumable steps and takes its next queue
// entry (grid-stride) when its ray is finished, so a wave is not held back by its longest
// walk.  A wave leaves when none of its lanes has a ray and none can get one.
#ifdef RTX_WALK_STATS
__device__ unsigned long long g_walk_stats[8];   // diagnostic build only
#endif

// A wave takes its queue entries in chunks: one atomic on the shared cursor per kFeedChunk entries instead of one per refill.
// All fields are the same in every lane.
#ifndef RTX_FEED_CHUNK
#define RTX_FEED_CHUNK 256
#endif
constexpr uint32_t kFeedChunk = RTX_FEED_CHUNK;
struct WaveFeed { uint32_t next, end; bool dry; };

// Entry of the queue of n for each lane with `want` (0xffffffff once the queue is used up); every lane of the wave calls it.
RTX_DEV uint32_t feed_take(WaveFeed& f, uint32_t* cursor, bool want, uint32_t n) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return 0xffffffffu;
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t need = (uint32_t)__popcll(mask), rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  uint32_t got = 0xffffffffu;
  const uint32_t avail = f.end - f.next, served = avail < need ? avail : need;
  if (want && rank < served) got = f.next + rank;
  f.next += served;
  if (served < need && !f.dry) {
    uint32_t base = 0u;
    if (lane == 0) base = atomicAdd(cursor, kFeedChunk);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    f.dry = base >= n;
    f.next = f.dry ? n : base;
    f.end = f.dry ? n : (base + kFeedChunk < n ? base + kFeedChunk : n);
    const uint32_t avail2 = f.end - f.next, rest = need - served, served2 = avail2 < rest ? avail2 : rest;
    if (want && rank >= served && rank - served < served2) got = f.next + (rank - served);
    f.next += served2;
  }
  return got;
}

// How many lanes of a wave must be without a ray before it stops to hand out new ones (storing results, promoting the prefetched
// rays and prefetching again costs the whole wave some sixty instructions).
#ifndef RTX_REFILL_IDLE
#define RTX_REFILL_IDLE 24   // 10 M triangles, 16 spp: 1: 207 ms, 8: 163, 16: 151, 32: 152 (when the kernel was written); 1080p x 64 spp frame now: 8: 374.1 ms, 16: 362.2, 20: 360.1, 24: 358.7, 28: 360.0, 32: 364.1
#endif

template <int MODE>
__global__ __launch_bounds__(kBlock, RTX_WALK_MIN_WAVES) void wf_extend(DScene<float> sc, WfArrays A, WfControl* ctl, int nq /* the queue's counter */, int stack_levels, DCounters* counters,
                                                    const uint32_t* __restrict__ order /* the queue's entries in the order to take them */, DRenderParams P) {
  const uint32_t n = ctl->n_q_[nq].v;
  uint32_t n_nodes = 0, n_prims = 0;   // node records and leaf records this lane fetched
  WaveFeed feed = {0u, 0u, false};
  LaneStack st = lane_stack(0);
  const bool tree = MODE != MODE_FLAT && sc.n_prims > kLinearMaxPrims;
  // the ray being walked
  uint32_t slot = 0;
  Ray<float> ray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
  Cand<float> c;
  Walk8 w;
  cand_init(c, 1e9f);
  w.node = kWalkDone; w.sp = 0; w.best_p = -1; w.tied = 0; w.inv = {0.f, 0.f, 0.f};
  int state = 0;   // 0 no ray, 1 walking, 2 walk finished, result not yet stored
  // the next ray, loaded while the current one is walked, so that a lane changes rays without waiting on memory
  uint32_t nslot = 0;
  Ray<float> nray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
  bool have_next = false;
  for (;;) {
    const int walking = __popcll(__ballot(state == 1));
    if (64 - walking >= RTX_REFILL_IDLE || walking == 0) {
#ifdef RTX_WALK_STATS
      { const unsigned long long ms = __ballot(state == 2); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_walk_stats[5], 1ull); atomicAdd(&g_walk_stats[6], (unsigned long long)__popcll(ms)); } }
#endif
      if (state == 2) {
        int id = -1, uv = 0;
        float t = 0.f, ua = 0.f, ub = 0.f;
        if constexpr (MODE == MODE_FLAT) {
          Hit<float> h;   // flat scenes: the candidate is cheap to redo in the shade kernel; keep t + id
          if (flat_closest(sc, ray, 0.001f, 1e9f, h)) { id = (h.type << kIdBits) | h.idx; t = h.t; }
        } else {
          if (!tree) test_unbounded(sc, ray, 0.001f, c);   // (after a tree walk, wf_shade tests the planes and lights: every lane is busy there)
          if (c.type >= 0) { id = (c.type << kIdBits) | c.idx; t = c.t; ua = c.ua; ub = c.ub; uv = (c.uv_kind << kIdBits) | (c.uv_idx & kIdMask); }
        }
        A.hit_t[slot] = t; A.hit_id[slot] = id;
        if constexpr (MODE == MODE_TREE_UV) { A.hit_ua[slot] = ua; A.hit_ub[slot] = ub; A.hit_uv[slot] = uv; }
        if (tree && w.tied) { note_tied_item(P, counters, A.cur[slot]); w.tied = 0; }   // (decided without the reference's order)
        state = 0;
      }
      if (state == 0 && have_next) {   // take over the prefetched ray
        slot = nslot; ray = nray;
        cand_init(c, 1e9f);
        if (tree) walk8_begin(w, ray);
        state = 1; have_next = false;
      }
      const uint32_t e = feed_take(feed, &ctl->cursor_rays_.v, !have_next, n);
      if (e != 0xffffffffu) {
        nslot = order[e];
        nray.o = {A.ray[0][nslot], A.ray[1][nslot], A.ray[2][nslot]};
        nray.d = {A.ray[3][nslot], A.ray[4][nslot], A.ray[5][nslot]};
        have_next = true;
      }
      if (__ballot(state == 1 || have_next) == 0ull) break;
    }
    if constexpr (MODE == MODE_FLAT) {
      if (state == 1) state = 2;
    } else if (!tree) {
      if (state == 1) { walk_bounded<float, false>(sc, ray, 0.001f, c, st); state = 2; }
    } else {
      const bool inner = state == 1 && walk8_at_inner(w);
#ifdef RTX_WALK_STATS
      { const unsigned long long mi = __ballot(inner); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_walk_stats[0], 1ull); if (mi) { atomicAdd(&g_walk_stats[1], 1ull); atomicAdd(&g_walk_stats[2], (unsigned long long)__popcll(mi)); } } }
#endif
      if (inner) { walk8_inner(sc, ray, 0.001f, c.t, w, st); ++n_nodes; }
      // leaves wait until a batch of lanes has one, or until no lane has an inner node left
      const bool leaf = state == 1 && walk8_at_leaf(w);
      if (kLeafBatch <= 1 || __popcll(__ballot(leaf)) >= kLeafBatch || __ballot(state == 1 && walk8_at_inner(w)) == 0ull) {
#ifdef RTX_WALK_STATS
        { const unsigned long long ml = __ballot(leaf); if ((threadIdx.x & 63) == 0 && ml) { atomicAdd(&g_walk_stats[3], 1ull); atomicAdd(&g_walk_stats[4], (unsigned long long)__popcll(ml)); } }
#endif
        if (leaf) { walk8_leaf<false>(sc, ray, 0.001f, c, w, st); ++n_prims; }
      }
      if (state == 1 && w.node == kWalkDone) state = 2;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    n_nodes += (uint32_t)__shfl_xor((int)n_nodes, off, 64);
    n_prims += (uint32_t)__shfl_xor((int)n_prims, off, 64);
  }
  if ((threadIdx.x & 63) == 0 && (n_nodes | n_prims)) {
    atomicAdd(&counters->nodes_visited, (unsigned long long)n_nodes);
    atomicAdd(&counters->prims_tested, (unsigned long long)n_prims);
  }
}

// Rebuild the Hit of a stored candidate (point, normal, face, material, u/v).  After a tree walk the candidate is the nearest
// BOUNDED primitive only: the planes and lights (the linear tail of Scene::hit, engine/scene.hpp:640-672) are tested here, against
// the interval that candidate leaves.
template <int MODE> RTX_DEV bool wf_load_hit(const DScene<float>& sc, const WfArrays& A, uint32_t slot, const Ray<float>& ray, Hit<float>& h) {
  const int id = A.hit_id[slot];
  const bool tree = MODE != MODE_FLAT && sc.n_prims > kLinearMaxPrims;
  if (id < 0 && !tree) return false;
  Cand<float> c;
  cand_init(c, 1e9f);
  if (id >= 0) { c.t = A.hit_t[slot]; c.type = id >> kIdBits; c.idx = id & kIdMask; }
  if constexpr (MODE == MODE_TREE_UV) {
    if (id >= 0) {
      const int uv = A.hit_uv[slot];
      c.ua = A.hit_ua[slot]; c.ub = A.hit_ub[slot]; c.uv_kind = uv >> kIdBits; c.uv_idx = uv & kIdMask;
    }
  }
  if (tree) test_unbounded(sc, ray, 0.001f, c);
  if (c.type < 0) return false;
  if constexpr (MODE == MODE_TREE_UV) finalize_hit<float, true>(sc, ray, c, h);
  else finalize_hit<float, false>(sc, ray, c, h);
  return true;
}

// Where a visible light sample's contribution goes: the chunk sum of its slot, or - the path that cast it was the last of its
// work item, whose sum is already written out - that item's entry of the partial sums.
constexpr uint32_t kShadowToItem = 0x80000000u;

template <int MODE, bool LDS_MATS, bool NMAP = false>
__global__ __launch_bounds__(kStreamBlock) void wf_shade(DScene<float> sc, DCamera<float> cam, DRenderParams P, WfArrays A, WfControl* ctl, WfTurn turn,
                                                   float* __restrict__ partial, WfStatBank* banks, const uint32_t* __restrict__ order /* the queue's entries in the order to take them, or NULL = as queued */) {
  const DMaterial<float>* mats = sc.mats;
  if constexpr (LDS_MATS) mats = stage_materials(sc);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ BlockScratch bs;
  if (i == 0) {   // every kernel of the turn before has finished: its cursors and the counters it read are free
    ctl->cursor_rays_.v = 0u; ctl->cursor_shadow_.v = 0u;
    ctl->n_q_[turn.n_zero].v = 0u; ctl->n_shadow_[1 - turn.sh].v = 0u;
  }
  const bool valid = i < ctl->n_q_[turn.n_in].v;
  uint32_t did = 0;
  bool cast = false;           // a light sample needs its visibility ray
  Ray<float> sh_ray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
  float sh_tmax = 0.f;
  V3<float> sh_contrib = {0.f, 0.f, 0.f};
  uint32_t slot = 0, sh_target = 0;
  bool live = false, started = false;   // the slot has a ray to trace next; it began a new sample
  if (valid) {
  slot = order ? order[i] : A.q[turn.q_in][i];
  sh_target = slot;
  PathState<float> ps;
  ps.ray.o = {A.ray[0][slot], A.ray[1][slot], A.ray[2][slot]};
  ps.ray.d = {A.ray[3][slot], A.ray[4][slot], A.ray[5][slot]};
  ps.thr = {A.thr[0][slot], A.thr[1][slot], A.thr[2][slot]};
  ps.acc = {A.acc[0][slot], A.acc[1][slot], A.acc[2][slot]};
  ps.bsdf_pdf = A.bsdf_pdf[slot];
  ps.bounce_flags = A.bflags[slot];
  const uint32_t pixel = A.pixel[slot], s = A.sample[slot];
  PathKey key = {pixel, P.seed, s - 1u, (uint32_t)P.rng};
  Hit<float> rec;
  const bool hit = wf_load_hit<MODE>(sc, A, slot, ps.ray, rec);
  did = path_vertex<float, NMAP>(ps, hit, rec, sc, mats, P, key, [&](const Ray<float>& sr, float t_max, V3<float> contrib) {
    cast = true; sh_ray = sr; sh_tmax = t_max; sh_contrib = contrib;
  });
  // a light sample behind a plane or a light needs no walk (wf_connect tests the tree only)
  if (cast && MODE != MODE_FLAT && sc.n_prims > kLinearMaxPrims && occluded_unbounded(sc, sh_ray, 0.001f, sh_tmax)) cast = false;
  live = true;
  if (!(ps.bounce_flags & F_ALIVE)) {
    // The path ended: it joins the chunk sum, and the slot starts its next sample or takes its next (pixel, chunk) work item.
    const uint32_t s_end = A.s_end[slot];
    V3<float> sum = {A.sum[0][slot] + ps.acc.x, A.sum[1][slot] + ps.acc.y, A.sum[2][slot] + ps.acc.z};
    WorkItem it = {A.xy[slot], pixel, s, s_end, 0u};
    if (s == s_end) {
      const uint32_t done = A.cur[slot];
      float* o = partial + (size_t)done * 3u;
      o[0] = sum.x; o[1] = sum.y; o[2] = sum.z;
      sh_target = kShadowToItem | done;
      sum = {0.f, 0.f, 0.f};
      const uint32_t item = A.next[slot];
      if (item >= work_items(P)) {
        live = false;
      } else {
        it = work_item(P, item);
        A.cur[slot] = it.id; A.next[slot] = item + A.n_slots; A.xy[slot] = it.xy; A.pixel[slot] = it.pixel; A.s_end[slot] = it.s_end;
      }
    }
    if (live) {
      PathKey next_key = {it.pixel, P.seed, it.s_begin, (uint32_t)P.rng};
      path_begin(ps, cam, P, (int)(it.xy & 0xffffu), (int)(it.xy >> 16), next_key);
      A.sample[slot] = it.s_begin + 1u;     // index of the NEXT sample; the current one is sample - 1
      A.sum[0][slot] = sum.x; A.sum[1][slot] = sum.y; A.sum[2][slot] = sum.z;
      started = true;
    }
  }
  if (live) {
    A.ray[0][slot] = ps.ray.o.x; A.ray[1][slot] = ps.ray.o.y; A.ray[2][slot] = ps.ray.o.z;
    A.ray[3][slot] = ps.ray.d.x; A.ray[4][slot] = ps.ray.d.y; A.ray[5][slot] = ps.ray.d.z;
    A.thr[0][slot] = ps.thr.x; A.thr[1][slot] = ps.thr.y; A.thr[2][slot] = ps.thr.z;
    A.acc[0][slot] = ps.acc.x; A.acc[1][slot] = ps.acc.y; A.acc[2][slot] = ps.acc.z;
    A.bsdf_pdf[slot] = ps.bsdf_pdf;
    A.bflags[slot] = ps.bounce_flags;
  }
  }
  uint32_t k, at;
  block_append2(bs, &ctl->n_shadow_[turn.sh].v, cast, k, &ctl->n_q_[turn.n_out].v, live, at);
  if (cast) {
    A.sh_slot[k] = sh_target;
    A.sh[0][k] = sh_ray.o.x; A.sh[1][k] = sh_ray.o.y; A.sh[2][k] = sh_ray.o.z;
    A.sh[3][k] = sh_ray.d.x; A.sh[4][k] = sh_ray.d.y; A.sh[5][k] = sh_ray.d.z; A.sh[6][k] = sh_tmax;
    A.sh_c[0][k] = sh_contrib.x; A.sh_c[1][k] = sh_contrib.y; A.sh_c[2][k] = sh_contrib.z;
  }
  if (live) A.q[turn.q_out][at] = slot;
  block_count4(bs, banks, (did & STEP_SHADED) != 0, (did & STEP_SHADOW) != 0, live, started);
}

template <int MODE>
__global__ __launch_bounds__(kBlock, RTX_WALK_MIN_WAVES) void wf_connect(DScene<float> sc, WfArrays A, WfControl* ctl, int sh /* the shadow queue's counter */, float* __restrict__ partial, int stack_levels,
                                                     const uint32_t* __restrict__ order /* shadow-queue indices in the order to take them, or NULL = as queued */) {
  const uint32_t n = ctl->n_shadow_[sh].v;
  WaveFeed feed = {0u, 0u, false};
  LaneStack st = lane_stack(0);
  const bool tree = MODE != MODE_FLAT && sc.n_prims > kLinearMaxPrims;
  uint32_t i = 0;
  float t_max = 0.f;
  Ray<float> sr = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
  Cand<float> c;
  Walk8 w;
  cand_init(c, 1e9f);
  w.node = kWalkDone; w.sp = 0; w.best_p = -1; w.tied = 0; w.inv = {0.f, 0.f, 0.f};
  int state = 0;   // 0 no ray, 1 walking, 2 finished (occluded), 3 finished (visible)
  uint32_t ni = 0;   // the next shadow ray, loaded while the current one is walked
  float nt_max = 0.f;
  Ray<float> nsr = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}};
  bool have_next = false;
  for (;;) {
    const int walking = __popcll(__ballot(state == 1));
    if (64 - walking >= RTX_REFILL_IDLE || walking == 0) {
      if (state == 3) {
        const uint32_t to = A.sh_slot[i];   // at most one shadow ray per slot (and per finished work item) per turn: no atomics needed
        if (to & kShadowToItem) {
          float* o = partial + (size_t)(to & ~kShadowToItem) * 3u;
          o[0] += A.sh_c[0][i]; o[1] += A.sh_c[1][i]; o[2] += A.sh_c[2][i];
        } else {
          A.sum[0][to] += A.sh_c[0][i]; A.sum[1][to] += A.sh_c[1][i]; A.sum[2][to] += A.sh_c[2][i];
        }
      }
      if (state != 1) state = 0;
      if (state == 0 && have_next) {
        i = ni; sr = nsr; t_max = nt_max;
        cand_init(c, t_max);
        if (tree) walk8_begin(w, sr);
        state = 1; have_next = false;
      }
      const uint32_t e = feed_take(feed, &ctl->cursor_shadow_.v, !have_next, n);
      if (e != 0xffffffffu) {
        ni = order ? order[e] : e;
        nsr.o = {A.sh[0][ni], A.sh[1][ni], A.sh[2][ni]};
        nsr.d = {A.sh[3][ni], A.sh[4][ni], A.sh[5][ni]};
        nt_max = A.sh[6][ni];
        have_next = true;
      }
      if (__ballot(state == 1 || have_next) == 0ull) break;
    }
    if constexpr (MODE == MODE_FLAT) {
      if (state == 1) state = flat_occluded<true>(sc, sr, 0.001f, t_max) ? 2 : 3;   // queued light-sample rays are normalised
    } else if (!tree) {
      if (state == 1) state = occluded(sc, sr, 0.001f, t_max, st) ? 2 : 3;
    } else {
      const bool inner = state == 1 && walk8_at_inner(w);
      if (inner) walk8_inner<false>(sc, sr, 0.001f, c.t, w, st);   // any hit ends the query: no nearest-first order
      const bool leaf = state == 1 && walk8_at_leaf(w);
      if (kLeafBatch <= 1 || __popcll(__ballot(leaf)) >= kLeafBatch || __ballot(state == 1 && walk8_at_inner(w)) == 0ull) {
        if (leaf) walk8_leaf<true>(sc, sr, 0.001f, c, w, st);
      }
      if (state == 1 && w.node == kWalkDone) state = c.type >= 0 ? 2 : 3;   // (wf_shade has tested the planes and lights)
    }
  }
}

template <int MODE> void launch_turn_kernels(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, const WfArrays& A, WfControl* ctl,
                                             WfStatBank* banks, float* partial, DCounters* counters, bool lds_mats, size_t mat_lds, int stack_levels, unsigned grid,
                                             unsigned resident_blocks, const WfTurn* turn /* NULL: the start of a render */, hipStream_t stream, uint32_t& launches,
                                             WavefrontPool& pool) {
  const dim3 g((grid + (kStreamBlock / kBlock) - 1) / (kStreamBlock / kBlock)), bs(kStreamBlock), b(kBlock);   // `grid` counts kBlock-sized groups
  const dim3 gw(grid < resident_blocks ? grid : resident_blocks);   // persistent traversal kernels
  const size_t walk_lds = MODE == MODE_FLAT ? 0 : (size_t)stack_levels * kBlock * sizeof(int);
  const bool sorted = pool.sort_rays && MODE != MODE_FLAT && sc.n_nodes8 > 0;
  const uint32_t bound = grid * (unsigned)kBlock < A.n_slots ? grid * (unsigned)kBlock : A.n_slots;   // the queues are no longer than this
  auto sort_queue = [&](int q, int nq, int shadow) {
    hipLaunchKernelGGL(wf_sort_keys, dim3((bound + 255) / 256), dim3(256), 0, stream, sc, A, ctl, q, nq, shadow, bound);
    size_t bytes = pool.sort_tmp.cap;
    (void)rocprim::radix_sort_pairs(pool.sort_tmp.ptr, bytes, A.key_in, A.key_out, A.order_in, A.order_out, (size_t)bound, 0u, kSortKeyBits, stream);
    launches += 2;
  };
  int q_out = 0, n_out = 0;
  if (!turn) {
    hipLaunchKernelGGL(wf_start, g, bs, 0, stream, cam, P, A, ctl, banks);
    ++launches;
  } else {
    q_out = turn->q_out; n_out = turn->n_out;
    const uint32_t* shade_order = nullptr;
    if (pool.sort_materials) {   // experiment: material queues
      hipLaunchKernelGGL(wf_material_keys, dim3((bound + 255) / 256), dim3(256), 0, stream, sc, A, ctl, turn->q_in, turn->n_in, bound);
      size_t bytes = pool.sort_tmp.cap;
      (void)rocprim::radix_sort_pairs(pool.sort_tmp.ptr, bytes, A.key_in, A.key_out, A.order_in, A.order_out, (size_t)bound, 0u, kMaterialKeyBits, stream);
      launches += 2;
      shade_order = A.order_out;
    }
    if (MODE == MODE_TREE_UV && sc.n_nmaps > 0) {   // the instances that apply normal maps
      if (lds_mats) hipLaunchKernelGGL((wf_shade<MODE, true, MODE == MODE_TREE_UV>), g, bs, mat_lds, stream, sc, cam, P, A, ctl, *turn, partial, banks, shade_order);
      else hipLaunchKernelGGL((wf_shade<MODE, false, MODE == MODE_TREE_UV>), g, bs, 0, stream, sc, cam, P, A, ctl, *turn, partial, banks, shade_order);
    } else if (lds_mats) hipLaunchKernelGGL((wf_shade<MODE, true>), g, bs, mat_lds, stream, sc, cam, P, A, ctl, *turn, partial, banks, shade_order);
    else hipLaunchKernelGGL((wf_shade<MODE, false>), g, bs, 0, stream, sc, cam, P, A, ctl, *turn, partial, banks, shade_order);
    if (sorted) sort_queue(0, turn->sh, 1);
    if (pool.overlap) {   // the any-hit walk on the side stream, behind the shade kernel; the closest-hit walk below runs beside it
      (void)hipEventRecord(pool.ev_fork, stream);
      (void)hipStreamWaitEvent(pool.side, pool.ev_fork, 0);
      hipLaunchKernelGGL((wf_connect<MODE>), gw, b, walk_lds, pool.side, sc, A, ctl, turn->sh, partial, stack_levels, nullptr);
      (void)hipEventRecord(pool.ev_join, pool.side);
    } else {
      hipLaunchKernelGGL((wf_connect<MODE>), gw, b, walk_lds, stream, sc, A, ctl, turn->sh, partial, stack_levels, sorted ? A.order_out : nullptr);
    }
    launches += 2;
  }
  if (sorted) sort_queue(q_out, n_out, 0);
  if (!pool.mark(stream)) pool.timing_ok = false;   // an event could not be made or recorded: this render reports no traversal time
  hipLaunchKernelGGL((wf_extend<MODE>), gw, b, walk_lds, stream, sc, A, ctl, n_out, stack_levels, counters, sorted ? A.order_out : A.q[q_out], P);
  if (!pool.mark(stream)) pool.timing_ok = false;
  if (pool.overlap && turn) (void)hipStreamWaitEvent(stream, pool.ev_join, 0);   // the next kernel on `stream` needs both walks
  ++launches;
}

// Blocks of the traversal kernels one CU keeps resident (the smaller of the two kernels' figures).
template <int MODE> int walk_blocks_per_cu(int stack_levels) {
  const size_t walk_lds = MODE == MODE_FLAT ? 0 : (size_t)stack_levels * kBlock * sizeof(int);
  int a = 0, b = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, wf_extend<MODE>, kBlock, walk_lds) != hipSuccess) a = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, wf_connect<MODE>, kBlock, walk_lds) != hipSuccess) b = 0;
  const int nb = a < b ? a : b;
  return nb > 0 ? nb : 2;
}

}  // namespace

int wavefront_render(WavefrontPool& pool, const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* d_film,
                     DCounters* counters, int n_cus, int mode, int stack_levels, hipStream_t stream, uint32_t& launches, std::string& err) {
  const unsigned long long npix = (unsigned long long)P.rows * P.width;
  const unsigned long long all_items = npix * (unsigned long long)P.n_chunks;
  if (all_items >= 0x7fffffffull) { err = "film too large for one launch"; return RTX_ERR_INVALID_ARGUMENT; }
  // (a repair pass renders the listed work items only; the chunk sums of the others stay as the first pass left them)
  const unsigned long long items = P.item_list ? (unsigned long long)P.n_list : all_items;
  for (int count : {sc.n_spheres, sc.n_planes, sc.n_tris, sc.n_boxes, sc.n_quads, sc.n_disks})
    if (count > kIdMask) { err = "the queue pipeline indexes at most 2^28 primitives of one type"; return RTX_ERR_UNSUPPORTED; }
  // Pool size: enough slots to keep every work item in flight if memory allows (RTX_POOL_MULT slots per resident lane at most,
  // kWfArrays * 4 bytes each: 24 GB for a 1080p frame at 64 spp).  A device that cannot give that much - shared, or holding a large
  // scene - gets a pool of half the size, and so on down to 16 slots per lane (measured on the 10 M-triangle frame: 256: 364 ms,
  // 128: 374, 64: 389, 32: 409, 16: 440); the environment variable RTX_POOL_MULT caps the first try.
  if (!pool.control.ensure(sizeof(WfShared)) || !pool.queues.ensure((size_t)all_items * 3 * sizeof(float))) { err = "pool allocation (chunk sums)"; return RTX_ERR_OUT_OF_MEMORY; }
  unsigned long long mult = (unsigned long long)RTX_POOL_MULT;
  if (const char* e = getenv("RTX_POOL_MULT")) { const long v = atol(e); if (v >= 1 && (unsigned long long)v < mult) mult = (unsigned long long)v; }
  uint32_t n_slots = 0;
  for (;;) {
    n_slots = (uint32_t)std::min<unsigned long long>(items, (unsigned long long)n_cus * 2048ull * mult);
    if (pool.state.ensure((size_t)kWfArrays * n_slots * 4)) break;
    (void)hipGetLastError();   // the failed allocation's error code
    if (mult <= 16ull) { err = "pool allocation (" + std::to_string((size_t)kWfArrays * n_slots * 4) + " bytes of path slots, the smallest pool tried)"; return RTX_ERR_OUT_OF_MEMORY; }
    mult /= 2ull;
  }
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
    A.key_in = nextu(); A.key_out = nextu(); A.order_in = nextu(); A.order_out = nextu();
    A.n_slots = n_slots;
    if (k != (size_t)kWfArrays) { err = "internal: array count"; return RTX_ERR_INVALID_ARGUMENT; }
  }
  // Sorting the ray queues by origin cell and direction octant before each traversal launch is off by default: measured on a 10 M
  // triangle soup, the closest-hit kernel took the same time (it is bound by vector-ALU issue, not by the lines it fetches) and the
  // sorts added 68 ms to a 608 ms frame.  RTX_RAY_SORT=1 switches it on (experiments only; films do not depend on it).
  pool.sort_rays = sc.n_nodes8 > 0 && mode != MODE_FLAT && getenv("RTX_RAY_SORT") != nullptr;
  // Material queues (the shade kernel takes its entries ordered by material class) are off by default as well: measured on the
  // same frame, the keys and the sort cost several times what the shade kernel gains.  RTX_MATERIAL_SORT=1 switches them on
  // (experiments only; films do not depend on it).  The ray sort and the material sort share the key / order arrays: the shade
  // kernel's order is used up before the ray sort of the same turn writes them again.
  pool.sort_materials = getenv("RTX_MATERIAL_SORT") != nullptr;
  // The two tree walks of a turn - closest hit for the ray queue, any hit for the shadow queue - do not depend on each other.  In
  // a small launch the end of a persistent walk, when most waves have run out of rays, is a large share of its time: then the
  // any-hit walk goes to a stream of its own and the two run side by side, each filling the other's tail (10 M triangles, 64 spp,
  // every 8th row of a 1080p frame: 56.0 -> 52.3 ms; every 4th: 100.8 -> 97.8; every 2nd and the whole frame: no gain, 189.1 -> 190.1
  // and 374.1 -> 375.8).  Large launches keep them one after the other, so that each kernel's time can be measured on its own.
  // RTX_WALK_OVERLAP=0 / 1 forces one or the other (films do not depend on it).
  pool.overlap = n_slots < kOverlapBelowSlots;
  if (const char* e = getenv("RTX_WALK_OVERLAP")) pool.overlap = atoi(e) != 0;
  pool.overlap = pool.overlap && !pool.sort_rays && pool.ensure_side();
  if (pool.sort_rays || pool.sort_materials) {
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, A.key_in, A.key_out, A.order_in, A.order_out, (size_t)n_slots, 0u, kSortKeyBits, stream) != hipSuccess ||
        !pool.sort_tmp.ensure(bytes)) { err = "pool allocation (ray sort)"; return RTX_ERR_OUT_OF_MEMORY; }
  }
  WfControl* ctl = &((WfShared*)pool.control.ptr)->ctl;
  WfStatBank* banks = ((WfShared*)pool.control.ptr)->bank;
  float* partial = (float*)pool.queues.ptr;
  const size_t mat_bytes = (size_t)sc.n_mats * sizeof(DMaterial<float>);
  const bool lds_mats = mat_bytes <= kMaxLdsMaterialBytes;
  const size_t mat_lds = (mat_bytes + 15) & ~(size_t)15;
  if (!lds_mats) mode = MODE_TREE_UV;
  const int per_cu = mode == MODE_FLAT ? walk_blocks_per_cu<MODE_FLAT>(stack_levels) : mode == MODE_TREE ? walk_blocks_per_cu<MODE_TREE>(stack_levels)
                                                                                                       : walk_blocks_per_cu<MODE_TREE_UV>(stack_levels);
  const unsigned resident = (unsigned)n_cus * (unsigned)per_cu;

  auto run = [&](const WfTurn* turn, unsigned grid) {
    if (mode == MODE_FLAT) launch_turn_kernels<MODE_FLAT>(sc, cam, P, A, ctl, banks, partial, counters, lds_mats, mat_lds, stack_levels, grid, resident, turn, stream, launches, pool);
    else if (mode == MODE_TREE) launch_turn_kernels<MODE_TREE>(sc, cam, P, A, ctl, banks, partial, counters, lds_mats, mat_lds, stack_levels, grid, resident, turn, stream, launches, pool);
    else launch_turn_kernels<MODE_TREE_UV>(sc, cam, P, A, ctl, banks, partial, counters, lds_mats, mat_lds, stack_levels, grid, resident, turn, stream, launches, pool);
  };
  if (hipMemsetAsync(pool.control.ptr, 0, sizeof(WfShared), stream) != hipSuccess) { err = "memset"; return RTX_ERR_HIP; }
  pool.events_used = 0;
  pool.timing_ok = true;
  unsigned live_bound = n_slots;   // upper bound of the ray queue length (it only shrinks)
  unsigned grid = (live_bound + kBlock - 1) / kBlock;
  run(nullptr, grid);
  for (unsigned iter = 0;; ++iter) {
    const WfTurn turn = {(int)(iter & 1u), (int)((iter + 1u) & 1u), (int)(iter % 3u), (int)((iter + 1u) % 3u), (int)((iter + 2u) % 3u), (int)(iter & 1u)};
    run(&turn, grid);
    if ((iter & 7u) == 7u || live_bound < 65536u || live_bound > kSyncEveryTurnAbove) {
      uint32_t n_live = 0;
      if (hipMemcpyAsync(&n_live, &ctl->n_q_[turn.n_out].v, sizeof n_live, hipMemcpyDeviceToHost, stream) != hipSuccess ||
          hipStreamSynchronize(stream) != hipSuccess) { err = hipGetErrorString(hipGetLastError()); return RTX_ERR_HIP; }
      if (n_live == 0) break;
      live_bound = n_live;
      grid = (live_bound + kBlock - 1) / kBlock;
    }
    if (iter > 4000000u) { err = "iteration limit"; return RTX_ERR_HIP; }
  }
  hipLaunchKernelGGL(wf_stats_fold, dim3(1), dim3(64), 0, stream, banks, counters);
  launch_film_reduce(partial, d_film, npix, P.n_chunks, P.spp, stream);
  launches += 2;
#ifdef RTX_WALK_STATS
  {
    unsigned long long h[8];
    hipStreamSynchronize(stream);
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_walk_stats), sizeof h);
    fprintf(stderr, "walk stats (extend): wave trips %llu, inner phases %llu with %.1f lanes, leaf phases %llu with %.1f lanes, refills %llu storing %.1f results\n", h[0], h[1],
            (double)h[2] / (h[1] ? h[1] : 1), h[3], (double)h[4] / (h[3] ? h[3] : 1), h[5], (double)h[6] / (h[5] ? h[5] : 1));
  }
#endif
  return RTX_OK;
}

}  // namespace rtx
