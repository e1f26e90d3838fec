// The PathPhoton density estimate, computed by a whole wave for one shading point at a time.
// (ref: engine/photon_map.hpp:150-186 estimate_irradiance over :117-140, :250-305 find_nearest / search_tree)
//
// A k-nearest search per lane keeps k (distance, photon) pairs per lane in scratch memory and indexes them with a
// different index in every lane; measured on the 1080p Cornell box that search was 98 % of the render.  Here the 64 lanes
// of a wave serve the wave's shading points one after the other:
//  * the kd-tree walk is the same in every lane (scalar state: the range, and two bit masks instead of a stack, see
//    photon_query); a range of at most 64 photons is tested in one step, one photon per lane, read with one coalesced load;
//  * photons nearer than the current bound are appended to a candidate list in LDS (kPhotonCand entries per wave); when
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

#ifndef RTX_PHOTON_BUCKET
#define RTX_PHOTON_BUCKET 1024   // measured on the 1080p Cornell box, 50 000 photons: 64: 896 ms, 256: 645, 512: 529, 1024: 470, 2048: 494
#endif
constexpr int kPhotonBucket = RTX_PHOTON_BUCKET;   // a subtree of at most this many photons is tested without walking it

#ifndef RTX_PHOTON_UNROLL
#define RTX_PHOTON_UNROLL 8
#endif
constexpr int kPhotonUnroll = RTX_PHOTON_UNROLL;

constexpr int kPhotonTopLevels = 10;   // levels of the kd-tree copied into LDS (a walk step there costs an LDS read, not a trip to L2)
constexpr int kPhotonTop = 1 << kPhotonTopLevels;

// d2, id: this wave's candidate list in LDS.  top: the workgroup's copy of the first kPhotonTopLevels levels of the tree,
// addressed like a binary heap (root 1, children 2h and 2h + 1).
template <class R> struct PhotonCand { R* d2; int* id; const DPhotonNode<R>* top; };

// Fills `top` (all threads of the workgroup; the caller synchronises afterwards).
template <class R> RTX_DEV void photon_stage_top(const DRenderParams& P, DPhotonNode<R>* top) {
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

// Cuts the candidate list (n > k entries) down to its k nearest and returns the k-th smallest distance.
template <class R> RTX_DEV R photon_select(PhotonCand<R> pc, int& n, const int k, const int lane) {
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
  const int k = P.gather_k;
  const R radius = R(P.gather_radius);
  R bound = radius * radius;   // a photon is a candidate if it is nearer than this
  const int select_at = (kPhotonCand - 64 < k + 64) ? kPhotonCand - 64 : k + 64;   // list length that triggers a selection (> k)
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
    if (end - start <= kPhotonBucket) {   // the whole subtree at once, 64 photons per step, one per lane
      for (int first = start; first < end; first += 64 * kPhotonUnroll) {   // the loads of kPhotonUnroll steps are issued together
        DPhotonNode<R> q[kPhotonUnroll];
#pragma unroll
        for (int j = 0; j < kPhotonUnroll; ++j) { const int i = first + 64 * j + lane; q[j] = nodes[i < end ? i : first]; }
#pragma unroll
        for (int j = 0; j < kPhotonUnroll; ++j) {
          const int i = first + 64 * j + lane;
          const R d2 = len2(V3<R>{q[j].p[0], q[j].p[1], q[j].p[2]} - pos);
          bool take = i < end && d2 < bound;
          unsigned long long m = __ballot(take);
          if (m != 0ull) {
            if (n + (int)__popcll(m) > kPhotonCand) {   // n > kPhotonCand - 64 >= k
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
      start = end;
      continue;
    }
    const int mid = start + (end - start) / 2;
    const DPhotonNode<R> q = level < kPhotonTopLevels ? pc.top[heap] : nodes[mid];
    const R d2 = len2(V3<R>{q.p[0], q.p[1], q.p[2]} - pos);
    if (d2 < bound) {
      if (n == kPhotonCand) bound = photon_select(pc, n, k, lane);
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
