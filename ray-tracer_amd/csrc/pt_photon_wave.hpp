// The PathPhoton density estimate, computed by a whole wave for one shading point at a time.
// (ref: engine/photon_map.hpp:150-186 estimate_irradiance over :117-140, :250-305 find_nearest / search_tree)
//
// A k-nearest search per lane keeps k (distance, photon) pairs per lane in scratch memory and indexes them with a
// different index in every lane; measured on the 1080p Cornell box that search was 98 % of the render.  Here the 64 lanes
// of a wave serve the wave's shading points one after the other:
//  * the kd-tree walk is the same in every lane (scalar state: the range, and two bit masks instead of a stack, see
//    photon_query) and stops at subtrees of DRenderParams::ph_bucket photons (by default 64^3: maps below that size are
//    not walked at all); a subtree is searched through bounding boxes of its runs of 64 photons, one box or one photon per
//    lane, read with coalesced loads (photon_scan);
//  * photons nearer than the current bound are appended to a candidate list (in LDS, kPhotonCand entries per wave, for k up to
//    kPhotonMaxGather; in device memory for larger k, see photon_select_list); when
//    the list runs full, and once at the end, the k nearest are selected from it by a bit-wise search for the k-th
//    smallest distance (the order of non-negative IEEE numbers is the order of their bit patterns), counted with ballots;
//    the k-th distance becomes the new bound;
//  * the kernel sum over the kept photons is split over the lanes and reduced in a fixed order.
// The kept set is the reference's (the k nearest within the gather radius; photons at exactly the k-th distance are kept
// in list order up to k), the order of the floating-point sum is not.  The result does not depend on which lane or
// wave asks, so films stay independent of the launch geometry.
#pragma once

#include "pt_device.hpp"

namespace rtx {

constexpr int kPhotonCand = 256;   // candidate list entries per wave: a multiple of 64, at least kPhotonMaxGather + 64
static_assert(kPhotonCand % 64 == 0 && kPhotonCand >= kPhotonMaxGather + 64, "candidate list too small");
constexpr int kPhotonCandPerLane = kPhotonCand / 64;

#ifndef RTX_PHOTON_UNROLL
#define RTX_PHOTON_UNROLL 2   // measured on the 1080p Cornell box, 50 000 photons: 8: 156 ms, 4: 147, 2: 144
#endif
constexpr int kPhotonUnroll = RTX_PHOTON_UNROLL;

constexpr int kPhotonTopLevels = 10;   // levels of the kd-tree copied into LDS (a walk step there costs an LDS read, not a trip to L2)
constexpr int kPhotonTop = 1 << kPhotonTopLevels;

// The workgroup's photon table in LDS holds one of two things, chosen by the size of the map (the same choice in every
// workgroup): for a map that is walked (more than ph_bucket photons) the first kPhotonTopLevels levels of the kd-tree,
// addressed like a binary heap (root 1, children 2h and 2h + 1); for a map that is not, its bounding boxes (group boxes, then
// run boxes) as far as kPhotonLdsBoxes entries hold them all.  Either way the serial steps of a search read LDS, not L2.
constexpr int kPhotonLdsBoxes = 896;
template <class R> constexpr int photon_table_words() {   // in units of R
  return (kPhotonLdsBoxes * (int)sizeof(DPhotonBox<R>) > kPhotonTop * (int)sizeof(DPhotonNode<R>) ? kPhotonLdsBoxes * (int)sizeof(DPhotonBox<R>)
                                                                                                      : kPhotonTop * (int)sizeof(DPhotonNode<R>)) / (int)sizeof(R);
}

// d2, id: this wave's candidate list in LDS.  top / lbox: the workgroup's photon table seen as tree top or as boxes;
// lds_groups, lds_runs: whether the group boxes (at lbox[g]) and the run boxes (at lbox[n_groups + r]) are in it.
template <class R> struct PhotonCand {
  R* d2; int* id;
  int cap;                 // entries the list holds: kPhotonCand in LDS, DRenderParams::ph_cand_cap in device memory
  R* d2_alt; int* id_alt;  // device-memory lists only: the second list a selection writes its survivors to
  const DPhotonNode<R>* top;
  const DPhotonBox<R>* lbox;
  bool lds_groups, lds_runs;
};

// Fills the photon table (all threads of the workgroup; the caller synchronises afterwards) and completes `pc`.
template <class R> RTX_DEV void photon_stage_table(const DRenderParams& P, R* table, PhotonCand<R>& pc) {
  pc.top = reinterpret_cast<const DPhotonNode<R>*>(table);
  pc.lbox = reinterpret_cast<const DPhotonBox<R>*>(table);
  pc.lds_groups = pc.lds_runs = false;
  if (P.n_photons <= 0) return;
  if (P.n_photons > P.ph_bucket) {
    DPhotonNode<R>* top = reinterpret_cast<DPhotonNode<R>*>(table);
    const DPhotonNode<R>* nodes = (const DPhotonNode<R>*)P.ph_nodes;
    for (int h = 1 + (int)threadIdx.x; h < kPhotonTop; h += (int)blockDim.x) {
      const int level = 31 - __clz(h);
      int s = 0, e = P.n_photons;
      for (int i = level - 1; i >= 0 && s < e; --i) {
        const int m = s + (e - s) / 2;
        if ((h >> i) & 1) s = m + 1; else e = m;
      }
      if (s < e) top[h] = nodes[s + (e - s) / 2];
    }
    return;
  }
  const int n_runs = (P.n_photons + 63) / 64, n_groups = (n_runs + 63) / 64;
  pc.lds_groups = n_groups <= kPhotonLdsBoxes;
  pc.lds_runs = n_groups + n_runs <= kPhotonLdsBoxes;
  DPhotonBox<R>* lbox = reinterpret_cast<DPhotonBox<R>*>(table);
  const DPhotonBox<R>* boxes = (const DPhotonBox<R>*)P.ph_boxes;   // run boxes, then group boxes
  if (pc.lds_groups)
    for (int g = (int)threadIdx.x; g < n_groups; g += (int)blockDim.x) lbox[g] = boxes[n_runs + g];
  if (pc.lds_runs)
    for (int r = (int)threadIdx.x; r < n_runs; r += (int)blockDim.x) lbox[n_groups + r] = boxes[r];
}

RTX_DEV uint32_t dist_key(float v) { return __float_as_uint(v); }
RTX_DEV unsigned long long dist_key(double v) { return (unsigned long long)__double_as_longlong(v); }
RTX_DEV float key_dist(uint32_t k) { return __uint_as_float(k); }
RTX_DEV double key_dist(unsigned long long k) { return __longlong_as_double((long long)k); }

RTX_DEV float lane_value(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
RTX_DEV double lane_value(double v, int src) {
  const long long b = __double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
RTX_DEV int lanes_below(unsigned long long mask) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }

// Squared distance from `pos` to a box, scaled down a little: never above the distance of a photon inside, rounding included.
template <class R> RTX_DEV R box_distance2(const DPhotonBox<R> bx, const V3<R> pos) {
  const R dx = max_(max_(bx.lo[0] - pos.x, pos.x - bx.hi[0]), R(0));
  const R dy = max_(max_(bx.lo[1] - pos.y, pos.y - bx.hi[1]), R(0));
  const R dz = max_(max_(bx.lo[2] - pos.z, pos.z - bx.hi[2]), R(0));
  return (dx * dx + dy * dy + dz * dz) * R(0.9999);
}

// The same selection for a list of any length in device memory (k above kPhotonMaxGather): the keys are read again from the list
// for every bit of the search instead of being held in registers, and the survivors go to the wave's second list, which then
// becomes the list (compacting in place would overwrite entries that are still to be read).
template <class R> RTX_DEV R photon_select_list(PhotonCand<R>& pc, int& n, const int k, const int lane) {
  using K = decltype(dist_key(R(0)));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the list entries other lanes appended are read below
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  K kth = 0;
  for (int bit = (int)sizeof(K) * 8 - 2; bit >= 0; --bit) {
    const K trial = kth | (K(1) << bit);
    int below = 0;
    for (int base = 0; base < n; base += 64) {
      const int slot = base + lane;
      const K key = slot < n ? dist_key(pc.d2[slot]) : ~K(0);
      below += (int)__popcll(__ballot(key < trial));
    }
    if (below < k) kth = trial;
  }
  int out = 0;
  for (int pass = 0; pass < 2; ++pass)   // the keys below the k-th, then the keys equal to it as far as k entries go, both in list order
    for (int base = 0; base < n; base += 64) {
      const int slot = base + lane;
      const K key = slot < n ? dist_key(pc.d2[slot]) : ~K(0);
      const bool t = slot < n && (pass == 0 ? key < kth : key == kth);
      const unsigned long long m = __ballot(t);
      if (t) { const int pos = out + lanes_below(m); if (pos < k) { pc.d2_alt[pos] = key_dist(key); pc.id_alt[pos] = pc.id[slot]; } }
      out += (int)__popcll(m);
    }
  R* td = pc.d2; pc.d2 = pc.d2_alt; pc.d2_alt = td;
  int* ti = pc.id; pc.id = pc.id_alt; pc.id_alt = ti;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  n = k;
  return key_dist(kth);
}

// Cuts the candidate list (n > k entries) down to its k nearest and returns the k-th smallest distance.
template <class R> RTX_DEV R photon_select(PhotonCand<R>& pc, int& n, const int k, const int lane) {
  if (pc.cap != kPhotonCand) return photon_select_list(pc, n, k, lane);
  using K = decltype(dist_key(R(0)));
  K key[kPhotonCandPerLane];
  int id[kPhotonCandPerLane];
  bool in[kPhotonCandPerLane];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < kPhotonCandPerLane; ++j) {
    const int slot = j * 64 + lane;
    in[j] = slot < n;
    key[j] = in[j] ? dist_key(pc.d2[slot]) : ~K(0);
    id[j] = in[j] ? pc.id[slot] : 0;
  }
  K kth = 0;   // the largest T with fewer than k keys below T is the k-th smallest key
  for (int bit = (int)sizeof(K) * 8 - 2; bit >= 0; --bit) {   // distances are not negative: the sign bit stays clear
    const K trial = kth | (K(1) << bit);
    int below = 0;
#pragma unroll
    for (int j = 0; j < kPhotonCandPerLane; ++j) below += (int)__popcll(__ballot(key[j] < trial));
    if (below < k) kth = trial;
  }
  __builtin_amdgcn_wave_barrier();
  int base = 0;
#pragma unroll
  for (int j = 0; j < kPhotonCandPerLane; ++j) {
    const bool t = in[j] && key[j] < kth;
    const unsigned long long m = __ballot(t);
    if (t) { const int pos = base + lanes_below(m); pc.d2[pos] = key_dist(key[j]); pc.id[pos] = id[j]; }
    base += (int)__popcll(m);
  }
#pragma unroll
  for (int j = 0; j < kPhotonCandPerLane; ++j) {
    const bool t = in[j] && key[j] == kth;
    const unsigned long long m = __ballot(t);
    if (t) { const int pos = base + lanes_below(m); if (pos < k) { pc.d2[pos] = key_dist(key[j]); pc.id[pos] = id[j]; } }
    base += (int)__popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  n = k;   // base >= k: at least one key equals kth and fewer than k are below it
  return key_dist(kth);
}

// Photons [start, end) of the tree's array (a subtree), searched without walking it: neighbours in the array are neighbours
// in space, so the host keeps the box of every run of 64 photons and of every group of 64 runs.  One lane tests one group
// box, then one run box of a group in reach, then one photon of a run in reach.  Adds to the candidate list (n entries) and
// tightens `bound`.
template <class R> RTX_DEV void photon_scan(const DRenderParams& P, const V3<R> pos, const int start, const int end, PhotonCand<R>& pc, int& n, R& bound, const int lane) {
  const DPhotonNode<R>* nodes = (const DPhotonNode<R>*)P.ph_nodes;
  const DPhotonBox<R>* boxes = (const DPhotonBox<R>*)P.ph_boxes;     // box r: photons [64 r, 64 r + 64)
  const int n_groups = ((P.n_photons + 63) / 64 + 63) / 64;
  const DPhotonBox<R>* groups = boxes + (P.n_photons + 63) / 64;      // box g: runs [64 g, 64 g + 64)
  const int k = P.gather_k < P.n_photons ? P.gather_k : P.n_photons;
  const int select_at = (pc.cap - 64 < k + 64) ? pc.cap - 64 : k + 64;   // list length that triggers a selection (> k)
  const int run0 = start >> 6, run1 = (end - 1) >> 6;
  for (int gb = run0 >> 6; gb <= run1 >> 6; gb += 64) {
    R group_d2 = bound;
    if (gb + lane <= run1 >> 6) group_d2 = box_distance2(pc.lds_groups ? pc.lbox[gb + lane] : groups[gb + lane], pos);
    unsigned long long todo = __ballot(group_d2 < bound);
    while (todo != 0ull) {
      const int gl = (int)__ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      if (!(lane_value(group_d2, gl) < bound)) continue;   // the bound may have tightened since the box was tested
      const int rb = (gb + gl) * 64;
      R run_d2 = bound;
      if (rb + lane >= run0 && rb + lane <= run1) run_d2 = box_distance2(pc.lds_runs ? pc.lbox[n_groups + rb + lane] : boxes[rb + lane], pos);
      unsigned long long runs = __ballot(run_d2 < bound);
      while (runs != 0ull) {   // kPhotonUnroll runs per trip: their loads are issued together
        int first[kPhotonUnroll];
        R reach[kPhotonUnroll];
        DPhotonNode<R> q[kPhotonUnroll];
#pragma unroll
        for (int j = 0; j < kPhotonUnroll; ++j) {
          first[j] = -1; reach[j] = R(0);
          if (runs != 0ull) {
            const int l = (int)__ffsll((long long)runs) - 1;
            runs &= runs - 1ull;
            first[j] = (rb + l) * 64;
            reach[j] = lane_value(run_d2, l);
          }
        }
#pragma unroll
        for (int j = 0; j < kPhotonUnroll; ++j) {
          const int i = first[j] + lane;
          q[j] = nodes[(first[j] >= 0 && i >= start && i < end) ? i : start];
        }
#pragma unroll
        for (int j = 0; j < kPhotonUnroll; ++j) {
          if (first[j] < 0 || !(reach[j] < bound)) continue;
          const int i = first[j] + lane;
          const R d2 = len2(V3<R>{q[j].p[0], q[j].p[1], q[j].p[2]} - pos);
          bool take = i >= start && i < end && d2 < bound;
          unsigned long long m = __ballot(take);
          if (m == 0ull) continue;
          if (n + (int)__popcll(m) > pc.cap) {   // n > cap - 64 >= k
            bound = photon_select(pc, n, k, lane);
            take = take && d2 < bound;
            m = __ballot(take);
          }
          if (take) { const int slot = n + lanes_below(m); pc.d2[slot] = d2; pc.id[slot] = i; }
          n += (int)__popcll(m);
          if (n > k && n >= select_at) bound = photon_select(pc, n, k, lane);
        }
      }
    }
  }
}

// One shading point (the same `pos`, `normal` in every lane).  Every lane of the wave must be here.
// Walk state: the implicit layout of the tree (the median of a range sits at its middle) lets a range be recomputed from
// the root by replaying the turns taken, so `turns` (bit l = the walk went right at level l) and `pending` (bit l = the
// far side at level l is still to be visited) replace a stack; the deepest pending level is taken next (near side first,
// far side on the way back up, like the reference's recursion).
// The reference enters the far side of a split whenever fewer than k photons are kept, however far away it is
// (photon_map.hpp:295, :301), i.e. it walks the whole tree for every point with fewer than k photons in reach.  No
// photon beyond the split plane is nearer than the plane, so a far side at the bound or beyond adds nothing; the bound
// only tightens during the walk, so such a far side is not marked pending at all.
template <class R> RTX_DEV V3<R> photon_query(const DRenderParams& P, const V3<R> pos, const V3<R> normal, PhotonCand<R> pc, const int lane) {
  const DPhotonNode<R>* nodes = (const DPhotonNode<R>*)P.ph_nodes;
  const DPhotonData<R>* data = (const DPhotonData<R>*)P.ph_data;
  const int k = P.gather_k < P.n_photons ? P.gather_k : P.n_photons;   // there are no more photons than that to keep
  const R radius = R(P.gather_radius);
  R bound = radius * radius;   // a photon is a candidate if it is nearer than this
  int n = 0;
  uint32_t turns = 0u, pending = 0u;
  int level = 0, start = 0, end = P.n_photons;
  uint32_t heap = 1u;   // heap-order number of the current node while level < kPhotonTopLevels
  for (;;) {
    if (start >= end) {   // this side is done: go back to the deepest far side still pending
      if (pending == 0u) break;
      level = 31 - __clz((int)pending);
      pending &= ~(1u << level);
      int s = 0, e = P.n_photons;
      heap = 1;
      for (int i = 0; i < level; ++i) {
        const int m = s + (e - s) / 2;
        const uint32_t right = (turns >> i) & 1u;
        if (right) s = m + 1; else e = m;
        heap = 2 * heap + right;
      }
      const int m = s + (e - s) / 2;
      const DPhotonNode<R> q = level < kPhotonTopLevels ? pc.top[heap] : nodes[m];
      const R pa = q.axis == 0 ? pos.x : q.axis == 1 ? pos.y : pos.z;
      const R delta = pa - q.p[q.axis];
      if (!(delta * delta < bound)) continue;   // start >= end still holds: the next pending level is taken
      if ((turns >> level) & 1u) { start = s; end = m; } else { start = m + 1; end = e; }
      turns ^= 1u << level;
      heap = 2u * heap + ((turns >> level) & 1u);
      ++level;
      continue;
    }
    if (end - start <= P.ph_bucket) {   // no more walking
      photon_scan<R>(P, pos, start, end, pc, n, bound, lane);
      start = end;
      continue;
    }
    const int mid = start + (end - start) / 2;
    const DPhotonNode<R> q = level < kPhotonTopLevels ? pc.top[heap] : nodes[mid];
    const R d2 = len2(V3<R>{q.p[0], q.p[1], q.p[2]} - pos);
    if (d2 < bound) {
      if (n == pc.cap) bound = photon_select(pc, n, k, lane);
      if (d2 < bound) {
        if (lane == 0) { pc.d2[n] = d2; pc.id[n] = mid; }
        ++n;
      }
    }
    const R pa = q.axis == 0 ? pos.x : q.axis == 1 ? pos.y : pos.z;
    const R delta = pa - q.p[q.axis];
    if (level < 31 && delta * delta < bound) pending |= 1u << level;   // level 31 cannot exist with fewer than 2^31 photons
    if (delta < R(0)) { turns &= ~(1u << level); end = mid; heap = 2 * heap; } else { turns |= 1u << level; start = mid + 1; heap = 2 * heap + 1; }
    ++level;
  }
  if (n == 0) return {R(0), R(0), R(0)};
  if (n > k) photon_select(pc, n, k, lane);
  if (pc.cap != kPhotonCand) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
  __builtin_amdgcn_wave_barrier();
  R r2 = R(0);
  for (int slot = lane; slot < n; slot += 64) r2 = max_(r2, pc.d2[slot]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) r2 = max_(r2, __shfl_xor(r2, off, 64));
  if (r2 < R(1e-10)) r2 = R(1e-10);
  V3<R> flux = {R(0), R(0), R(0)};
  for (int slot = lane; slot < n; slot += 64) {
    const DPhotonData<R> d = data[pc.id[slot]];
    const R cos_theta = -(d.dir[0] * normal.x + d.dir[1] * normal.y + d.dir[2] * normal.z);
    if (cos_theta <= R(0)) continue;
    const R w = R(1) - sqrt_(pc.d2[slot] / r2);
    flux = flux + V3<R>{d.power[0], d.power[1], d.power[2]} * w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    flux.x = flux.x + __shfl_xor(flux.x, off, 64);
    flux.y = flux.y + __shfl_xor(flux.y, off, 64);
    flux.z = flux.z + __shfl_xor(flux.z, off, 64);
  }
  __builtin_amdgcn_wave_barrier();
  return flux * (R(3) / (Consts<R>::pi * r2));
}

// Every lane of the wave calls this in uniform flow; lanes with `want` get acc += weight * estimate(pos, normal).
template <class R> RTX_DEV void photon_gather_wave(const DRenderParams& P, const bool want, const V3<R> pos, const V3<R> normal, const V3<R> weight, V3<R>& acc,
                                                   PhotonCand<R> pc) {
  const int lane = (int)(threadIdx.x & 63u);
  unsigned long long todo = __ballot(want);
  while (todo != 0ull) {
    const int src = (int)__ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    const V3<R> qp = {lane_value(pos.x, src), lane_value(pos.y, src), lane_value(pos.z, src)};
    const V3<R> qn = {lane_value(normal.x, src), lane_value(normal.y, src), lane_value(normal.z, src)};
    const V3<R> est = photon_query<R>(P, qp, qn, pc, lane);
    if (lane == src) acc = acc + weight * est;
  }
}

}  // namespace rtx
