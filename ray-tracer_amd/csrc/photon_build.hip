// The caustic photon map of PathPhoton, built on the device from the photon pass's records: the reference's kd-tree
// (ref: engine/photon_map.hpp:104-108 build, :222-245 build_tree - the median of a range along axis depth % 3 sits at the middle
// of the range, the two halves are built the same way) and the bounding boxes the wave-wide search uses (pt_photon_wave.hpp).
//
// PhotonMap::build_tree moves the median of every range to its middle with std::nth_element.  Which photon ends up where is
// decided by the medians alone when no two coordinates are equal: position mid of [lo, hi) holds the median of the photons the
// range holds, and which photons those are follows from the medians of the ranges above.  So the tree can be built level by
// level without knowing how nth_element permutes: at depth d every position knows its range (it follows from the position), and
// ONE stable sort of all photons by (range, coordinate along axis d % 3) puts every range's median at its middle and the smaller /
// larger photons on its two sides; positions that were a middle at a shallower depth are ranges of their own and stay.
// floor(log2 n) such sorts build the tree.  Coordinates enter the sort keys as their ranks along each axis (three sorts ahead of
// the levels), so a key is two 32-bit halves whatever the precision, and equal coordinates keep their emission order (the
// reference's choice among equal coordinates is whatever nth_element leaves; any choice is a valid kd-tree and the k nearest
// photons of a point do not depend on it).
#include <hip/hip_runtime.h>

#include <cstring>   // (rocPRIM's texture iterator uses memset without including it)
#include <string>

#include <rocprim/rocprim.hpp>

#include "pt_host.hpp"

namespace rtx {
namespace {

struct Temp {
  void* p = nullptr;
  ~Temp() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
  template <class T> T* as() const { return (T*)p; }
};

// The range position p belongs to at depth `depth` of the build: [lo, hi).  `fixed`: p was the middle of a range at a shallower
// depth (`at` = that depth, the node's split axis is at % 3), or is alone in its range (at = -1: a leaf, axis 0 as the reference's
// default-constructed photon has).
struct Range { uint32_t lo, hi; bool fixed; int at; };
__device__ Range range_of(uint32_t p, uint32_t n, int depth) {
  Range r = {0u, n, false, -1};
  for (int d = 0; d < depth; ++d) {
    if (r.hi - r.lo <= 1u) break;
    const uint32_t mid = r.lo + (r.hi - r.lo) / 2u;
    if (p < mid) r.hi = mid;
    else if (p > mid) r.lo = mid + 1u;
    else { r.fixed = true; r.at = d; r.lo = p; r.hi = p + 1u; return r; }
  }
  if (r.hi - r.lo <= 1u) r.fixed = true;
  return r;
}

template <class R> __global__ void valid_flags_kernel(const R* raw, uint32_t n, uint8_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = raw[(size_t)9 * n + i] != R(0) ? 1 : 0;
}

// order-preserving 64-bit image of a coordinate
__device__ uint64_t sortable(double v) {
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
template <class R> __global__ void coord_keys_kernel(const R* raw, uint32_t n, const uint32_t* src, uint32_t m, int axis, uint64_t* keys, uint32_t* slots) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < m) { keys[e] = sortable((double)raw[(size_t)axis * n + src[e]]); slots[e] = e; }
}
__global__ void ranks_kernel(const uint32_t* sorted_slots, uint32_t m, uint32_t* rank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) rank[sorted_slots[i]] = i;
}
__global__ void identity_kernel(uint32_t* a, uint32_t m) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) a[i] = i;
}
__global__ void level_keys_kernel(const uint32_t* at_pos, const uint32_t* rank, uint32_t m, int depth, uint64_t* keys) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const Range r = range_of(p, m, depth);
  keys[p] = ((uint64_t)r.lo << 32) | (uint64_t)(r.fixed ? 0u : rank[at_pos[p]]);
}
template <class R> __global__ void write_map_kernel(const R* raw, uint32_t n, const uint32_t* src, const uint32_t* at_pos, uint32_t m, int levels, DPhotonNode<R>* nodes, DPhotonData<R>* data) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const Range r = range_of(p, m, levels + 1);
  const size_t s = src[at_pos[p]];
  DPhotonNode<R> nd;
  DPhotonData<R> dt;
  for (int a = 0; a < 3; ++a) { nd.p[a] = raw[(size_t)a * n + s]; dt.dir[a] = raw[(size_t)(3 + a) * n + s]; dt.power[a] = raw[(size_t)(6 + a) * n + s]; }
  nd.axis = r.at >= 0 ? r.at % 3 : 0;
  nodes[p] = nd; data[p] = dt;
}
// one wave per box: the box of a run of 64 photons (first = 0), or of a group of 64 run boxes
template <class R> __global__ void boxes_kernel(const DPhotonNode<R>* nodes, uint32_t m, DPhotonBox<R>* boxes, uint32_t n_runs, uint32_t n_groups, int groups) {
  const uint32_t b = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= (groups ? n_groups : n_runs)) return;
  const uint32_t i = b * 64u + lane, count = groups ? n_runs : m;
  const bool have = i < count;
  R lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    // lane 0 always has an element (b * 64 < count)
    const uint32_t j = have ? i : b * 64u;
    lo[a] = groups ? boxes[j].lo[a] : nodes[j].p[a];
    hi[a] = groups ? boxes[j].hi[a] : nodes[j].p[a];
    for (int off = 32; off > 0; off >>= 1) {
      const R ol = __shfl_xor(lo[a], off, 64), oh = __shfl_xor(hi[a], off, 64);
      lo[a] = ol < lo[a] ? ol : lo[a]; hi[a] = oh > hi[a] ? oh : hi[a];
    }
  }
  if (lane == 0) {
    DPhotonBox<R> bx;
    for (int a = 0; a < 3; ++a) { bx.lo[a] = lo[a]; bx.hi[a] = hi[a]; }
    boxes[(groups ? n_runs : 0u) + b] = bx;
  }
}

}  // namespace

#define PHOTON_TRY(expr)                                                                  \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { err = std::string(#expr ": ") + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP; } \
  } while (0)

template <class R>
int build_photon_map(const R* raw, uint32_t n, DeviceBuffer& nodes, DeviceBuffer& data, DeviceBuffer& boxes, int32_t& n_stored, hipStream_t stream, std::string& err) {
  n_stored = 0;
  if (n == 0) return RTX_OK;
  const unsigned grid_n = (n + 255u) / 256u;
  Temp flags, src, count, sel_tmp;
  if (!flags.alloc(n) || !src.alloc((size_t)n * 4) || !count.alloc(sizeof(uint32_t))) { err = "photon map: temporary buffers"; return RTX_ERR_OUT_OF_MEMORY; }
  // 1. the photons that were stored, in emission order (the reference's push_back sequence)
  hipLaunchKernelGGL(valid_flags_kernel<R>, dim3(grid_n), dim3(256), 0, stream, raw, n, flags.as<uint8_t>());
  size_t sel_bytes = 0;
  PHOTON_TRY(rocprim::select(nullptr, sel_bytes, rocprim::counting_iterator<uint32_t>(0u), flags.as<uint8_t>(), src.as<uint32_t>(), count.as<uint32_t>(), (size_t)n, stream));
  if (!sel_tmp.alloc(sel_bytes)) { err = "photon map: selection buffer"; return RTX_ERR_OUT_OF_MEMORY; }
  PHOTON_TRY(rocprim::select(sel_tmp.p, sel_bytes, rocprim::counting_iterator<uint32_t>(0u), flags.as<uint8_t>(), src.as<uint32_t>(), count.as<uint32_t>(), (size_t)n, stream));
  uint32_t m = 0;
  PHOTON_TRY(hipMemcpyAsync(&m, count.p, sizeof m, hipMemcpyDeviceToHost, stream));
  PHOTON_TRY(hipStreamSynchronize(stream));
  if (m == 0) return RTX_OK;
  const unsigned grid = (m + 255u) / 256u;
  int levels = 0;   // depth of the deepest range that still holds two photons
  for (uint32_t s = m; s > 1u; s /= 2u) ++levels;
  unsigned id_bits = 1;
  while ((1ull << id_bits) < (unsigned long long)m) ++id_bits;
  Temp keys_a, keys_b, ids_a, ids_b, rank[3], sort_tmp;
  if (!keys_a.alloc((size_t)m * 8) || !keys_b.alloc((size_t)m * 8) || !ids_a.alloc((size_t)m * 4) || !ids_b.alloc((size_t)m * 4) || !rank[0].alloc((size_t)m * 4) ||
      !rank[1].alloc((size_t)m * 4) || !rank[2].alloc((size_t)m * 4)) { err = "photon map: sort buffers"; return RTX_ERR_OUT_OF_MEMORY; }
  size_t tmp_bytes = 0;
  PHOTON_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), ids_a.as<uint32_t>(), ids_b.as<uint32_t>(), (size_t)m, 0u, 64u, stream));
  if (!sort_tmp.alloc(tmp_bytes)) { err = "photon map: sort buffer"; return RTX_ERR_OUT_OF_MEMORY; }
  // 2. every photon's rank along each axis
  for (int a = 0; a < 3; ++a) {
    hipLaunchKernelGGL(coord_keys_kernel<R>, dim3(grid), dim3(256), 0, stream, raw, n, src.as<uint32_t>(), m, a, keys_a.as<uint64_t>(), ids_a.as<uint32_t>());
    PHOTON_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), ids_a.as<uint32_t>(), ids_b.as<uint32_t>(), (size_t)m, 0u, 64u, stream));
    hipLaunchKernelGGL(ranks_kernel, dim3(grid), dim3(256), 0, stream, ids_b.as<uint32_t>(), m, rank[a].as<uint32_t>());
  }
  // 3. one stable sort by (range, rank along the level's axis) per level; at_pos[p] = the photon (its slot in `src`) at position p
  uint32_t* at_pos = ids_a.as<uint32_t>();
  uint32_t* other = ids_b.as<uint32_t>();
  hipLaunchKernelGGL(identity_kernel, dim3(grid), dim3(256), 0, stream, at_pos, m);
  for (int d = 0; d < levels; ++d) {
    hipLaunchKernelGGL(level_keys_kernel, dim3(grid), dim3(256), 0, stream, at_pos, rank[d % 3].as<uint32_t>(), m, d, keys_a.as<uint64_t>());
    PHOTON_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), at_pos, other, (size_t)m, 0u, 32u + id_bits, stream));
    uint32_t* t = at_pos; at_pos = other; other = t;
  }
  // 4. the map's arrays and the boxes of its runs and groups
  const uint32_t n_runs = (m + 63u) / 64u, n_groups = (n_runs + 63u) / 64u;
  if (!nodes.ensure((size_t)m * sizeof(DPhotonNode<R>)) || !data.ensure((size_t)m * sizeof(DPhotonData<R>)) || !boxes.ensure((size_t)(n_runs + n_groups) * sizeof(DPhotonBox<R>))) {
    err = "photon map arrays"; return RTX_ERR_OUT_OF_MEMORY;
  }
  hipLaunchKernelGGL(write_map_kernel<R>, dim3(grid), dim3(256), 0, stream, raw, n, src.as<uint32_t>(), at_pos, m, levels, (DPhotonNode<R>*)nodes.ptr, (DPhotonData<R>*)data.ptr);
  hipLaunchKernelGGL(boxes_kernel<R>, dim3((n_runs + 3u) / 4u), dim3(256), 0, stream, (const DPhotonNode<R>*)nodes.ptr, m, (DPhotonBox<R>*)boxes.ptr, n_runs, n_groups, 0);
  hipLaunchKernelGGL(boxes_kernel<R>, dim3((n_groups + 3u) / 4u), dim3(256), 0, stream, (const DPhotonNode<R>*)nodes.ptr, m, (DPhotonBox<R>*)boxes.ptr, n_runs, n_groups, 1);
  PHOTON_TRY(hipGetLastError());
  PHOTON_TRY(hipStreamSynchronize(stream));   // the temporaries are freed on return
  n_stored = (int32_t)m;
  return RTX_OK;
}

template int build_photon_map<float>(const float*, uint32_t, DeviceBuffer&, DeviceBuffer&, DeviceBuffer&, int32_t&, hipStream_t, std::string&);
template int build_photon_map<double>(const double*, uint32_t, DeviceBuffer&, DeviceBuffer&, DeviceBuffer&, int32_t&, hipStream_t, std::string&);

}  // namespace rtx
