// GPU construction of the eight-wide traversal tree of the single-precision kernels (DNode8 / DLeaf, pt_types.hpp).
//
// The reference builds its BVH on one host thread by recursive median splits (ref: engine/bvh.hpp:148-218); that tree is still
// built on the host for the validation entry point and for the tie ranks.  The tree the integrator walks is built here, in a
// handful of launches whatever the primitive count:
//
//   1. prim_box_kernel      bounding box of every bounded primitive (spheres, triangles, boxes: the reference's list order,
//                           ref: engine/scene.hpp:496-533) + the bounds of their centroids
//   2. morton_kernel        63-bit Morton code of every centroid (21 bits per axis)
//   3. radix sort           (rocPRIM) primitives in Morton order
//   4. split_level_kernel   top-down, one launch per level: a node takes its range of the sorted order and cuts it at the highest
//                           differing code bit (Karras 2012), always the largest remaining piece, until it has eight children or
//                           every piece is a single primitive (one leaf record)
//   5. emit_level_kernel    bottom-up, one launch per level: eight lanes per node gather the children's boxes (writing the leaf
//                           records on the way), reduce the node's box and quantise the children against it, rounding outward
//
// Levels are separated by kernel boundaries: no workgroup reads what another one wrote in the same launch.
#include <hip/hip_runtime.h>

#include <cstring>   // rocPRIM's headers use memset without including it

#include <rocprim/rocprim.hpp>

#include <chrono>
#include <cmath>
#include <cstdio>

#include "pt_host.hpp"

namespace rtx {
namespace {

struct BuildCounters { uint32_t nodes, leaves, units; uint32_t cmin[3], cmax[3], bmin[3], bmax[3]; };   // centroid bounds and primitive bounds as order-preserving integers
struct OriginGrid { float anchor[3], step[3]; };   // the grid node origins are stored on (DNode8::org)
struct NodeMeta { int32_t inner_base, leaf_base; int32_t n_inner, n_leaf; int32_t unit_base, pad; };   // unit_base: where the children's records start, in 64-byte units

__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ inline float ord2f(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

constexpr float kBig = 3.0e38f;

// Grid-stride: the bounds go to the counters with twelve atomics per wave, which serialise on their six addresses (one wave per
// 64 primitives made this kernel 7 ms of a 27 ms build at 10 M triangles), so every wave first reduces all its primitives.
__global__ __launch_bounds__(256) void prim_box_kernel(const DSphere<float>* __restrict__ sph, int ns, const DTri<float>* __restrict__ tris, int nt,
                                                       const DBox<float>* __restrict__ boxes, int nb, const uint8_t* __restrict__ dead,
                                                       float4* __restrict__ lo4, float4* __restrict__ hi4, BuildCounters* ctr) {
  const int n = ns + nt + nb;
  float cmin[3] = {kBig, kBig, kBig}, cmax[3] = {-kBig, -kBig, -kBig}, bmin[3] = {kBig, kBig, kBig}, bmax[3] = {-kBig, -kBig, -kBig};
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    float lo[3], hi[3];
    if (p < ns) {
      const DSphere<float> s = sph[p];
      for (int a = 0; a < 3; ++a) { lo[a] = nextafterf(s.c[a] - s.radius, -INFINITY); hi[a] = nextafterf(s.c[a] + s.radius, INFINITY); }
    } else if (p < ns + nt) {
      const DTri<float>& t = tris[p - ns];
      for (int a = 0; a < 3; ++a) { lo[a] = fminf(fminf(t.v0[a], t.v1[a]), t.v2[a]); hi[a] = fmaxf(fmaxf(t.v0[a], t.v1[a]), t.v2[a]); }
    } else {
      const DBox<float>& b = boxes[p - ns - nt];
      for (int a = 0; a < 3; ++a) { lo[a] = b.lo[a]; hi[a] = b.hi[a]; }
    }
    bool valid = !(dead && dead[p]);
    for (int a = 0; a < 3; ++a) valid = valid && lo[a] <= hi[a] && fabsf(lo[a]) < kBig && fabsf(hi[a]) < kBig;   // also drops NaN
    if (!valid) for (int a = 0; a < 3; ++a) { lo[a] = kBig; hi[a] = -kBig; }   // inverted: takes part in no union, entered by no ray
    lo4[p] = make_float4(lo[0], lo[1], lo[2], 0.f);
    hi4[p] = make_float4(hi[0], hi[1], hi[2], 0.f);
    if (valid)
      for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * lo[a] + 0.5f * hi[a];
        cmin[a] = fminf(cmin[a], c); cmax[a] = fmaxf(cmax[a], c); bmin[a] = fminf(bmin[a], lo[a]); bmax[a] = fmaxf(bmax[a], hi[a]);
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    for (int a = 0; a < 3; ++a) {
      cmin[a] = fminf(cmin[a], __shfl_xor(cmin[a], off, 64)); cmax[a] = fmaxf(cmax[a], __shfl_xor(cmax[a], off, 64));
      bmin[a] = fminf(bmin[a], __shfl_xor(bmin[a], off, 64)); bmax[a] = fmaxf(bmax[a], __shfl_xor(bmax[a], off, 64));
    }
  if ((threadIdx.x & 63) == 0 && cmin[0] <= cmax[0])
    for (int a = 0; a < 3; ++a) {
      atomicMin(&ctr->cmin[a], f2ord(cmin[a])); atomicMax(&ctr->cmax[a], f2ord(cmax[a]));
      atomicMin(&ctr->bmin[a], f2ord(bmin[a])); atomicMax(&ctr->bmax[a], f2ord(bmax[a]));
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ __launch_bounds__(256) void morton_kernel(const float4* __restrict__ lo4, const float4* __restrict__ hi4, int n, const BuildCounters* __restrict__ ctr,
                                                     uint64_t* __restrict__ codes, uint32_t* __restrict__ ids) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float4 lo = lo4[p], hi = hi4[p];
  const float c[3] = {0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z};
  uint32_t q[3] = {0u, 0u, 0u};
  if (lo.x <= hi.x) {
    for (int a = 0; a < 3; ++a) {
      const float mn = ord2f(ctr->cmin[a]), mx = ord2f(ctr->cmax[a]);
      const float ext = mx - mn;
      const float u = ext > 0.f ? (c[a] - mn) / ext : 0.f;
      const float s = fminf(fmaxf(u * 2097152.0f, 0.f), 2097151.0f);
      q[a] = (uint32_t)s;
    }
  }
  codes[p] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  ids[p] = (uint32_t)p;
}

// [b, e) with e - b >= 2 -> m in (b, e): the range is cut where the highest differing bit of its codes flips.
__device__ __forceinline__ int find_split(const uint64_t* __restrict__ codes, int b, int e) {
  const uint64_t first = codes[b], last = codes[e - 1];
  if (first == last) return (b + e) >> 1;
  const int prefix = __clzll((long long)(first ^ last));
  int lo = b, hi = e - 1;   // codes[lo] shares more than `prefix` leading bits with `first`, codes[hi] does not
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    const uint64_t x = first ^ codes[mid];
    if (x == 0ull || __clzll((long long)x) > prefix) lo = mid; else hi = mid;
  }
  return hi;
}

__global__ __launch_bounds__(256) void split_level_kernel(const uint64_t* __restrict__ codes, int2* range, NodeMeta* __restrict__ meta, int2* __restrict__ leaf_range,
                                                          int32_t* __restrict__ unit_of, int first, int count, BuildCounters* ctr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const int node = first + t;
  int rb[8], re[8];
  int n = 1;
  { const int2 r = range[node]; rb[0] = r.x; re[0] = r.y; }
  // (Dividing the two lowest levels evenly by count instead - full leaves, eight children per node - was measured on a 10 M
  // triangle soup: a third fewer records in memory, but runs of the Morton order have larger boxes than pieces cut at code bits,
  // and a ray fetched 10.8 instead of 6.3 leaf records and as many node records as before.)
  while (n < 8) {
    int best = -1, bc = kLeafMaxPrims;
    for (int i = 0; i < n; ++i) { const int cnt = re[i] - rb[i]; if (cnt > bc) { bc = cnt; best = i; } }
    if (best < 0) break;
    const int m = find_split(codes, rb[best], re[best]);
    rb[n] = m; re[n] = re[best]; re[best] = m;
    ++n;
  }
  int n_inner = 0;
  for (int i = 0; i < n; ++i) n_inner += (re[i] - rb[i] > kLeafMaxPrims) ? 1 : 0;
  const int n_leaf = n - n_inner;
  const int ib = n_inner ? (int)atomicAdd(&ctr->nodes, (uint32_t)n_inner) : 0;
  const int lb = n_leaf ? (int)atomicAdd(&ctr->leaves, (uint32_t)n_leaf) : 0;
  // the children's records, one unit each: the inner children's node records, then the leaf records
  const int ub = (int)atomicAdd(&ctr->units, (uint32_t)n);
  int ki = 0, kl = 0;
  for (int i = 0; i < n; ++i) {
    if (re[i] - rb[i] > kLeafMaxPrims) { unit_of[ib + ki] = ub + ki; range[ib + ki++] = make_int2(rb[i], re[i]); }
    else leaf_range[lb + kl++] = make_int2(rb[i], re[i] - rb[i]);
  }
  NodeMeta m;
  m.inner_base = ib; m.leaf_base = lb; m.n_inner = n_inner; m.n_leaf = n_leaf; m.unit_base = ub; m.pad = 0;
  meta[node] = m;
}

// Eight lanes per node, one per child slot.
__global__ __launch_bounds__(256) void emit_level_kernel(int first, int count, const NodeMeta* __restrict__ meta, const int2* __restrict__ leaf_range,
                                                         const uint32_t* __restrict__ order, const float4* __restrict__ plo, const float4* __restrict__ phi,
                                                         float4* nlo, float4* nhi, DNode8* __restrict__ units, const int32_t* __restrict__ unit_of,
                                                         const DTri<float>* __restrict__ tris, int ns, int nt, OriginGrid grid) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = g & 7;
  const bool active = (g >> 3) < count;
  const int node = first + (active ? (g >> 3) : 0);
  const NodeMeta m = meta[node];
  float clo[3] = {kBig, kBig, kBig}, chi[3] = {-kBig, -kBig, -kBig};
  if (active) {
    if (s < m.n_inner) {
      const float4 a = nlo[m.inner_base + s], b = nhi[m.inner_base + s];
      clo[0] = a.x; clo[1] = a.y; clo[2] = a.z; chi[0] = b.x; chi[1] = b.y; chi[2] = b.z;
    } else if (s < m.n_inner + m.n_leaf) {
      const int l = m.leaf_base + (s - m.n_inner);
      const int2 lr = leaf_range[l];
      DLeaf rec;
      for (int j = 0; j < 9; ++j) rec.v[j] = NAN;
      rec.ref = -1;
      for (int j = 0; j < 6; ++j) rec.pad[j] = 0;
      if (lr.y > 0) {
        const int p = (int)order[lr.x];
        const float4 a = plo[p], b = phi[p];
        if (a.x <= b.x) {   // (an unhittable primitive keeps its record, with nothing to test)
          clo[0] = a.x; clo[1] = a.y; clo[2] = a.z; chi[0] = b.x; chi[1] = b.y; chi[2] = b.z;
          if (p >= ns && p < ns + nt) {
            const DTri<float>& t = tris[p - ns];
            for (int j = 0; j < 3; ++j) { rec.v[j] = t.v0[j]; rec.v[3 + j] = t.v1[j] - t.v0[j]; rec.v[6 + j] = t.v2[j] - t.v0[j]; }   // v0 and the two edges from it
            rec.ref = p;
          } else {
            rec.ref = -2 - p;
          }
        }
      }
      *reinterpret_cast<DLeaf*>(units + m.unit_base + s) = rec;
    }
  }
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a) { lo[a] = clo[a]; hi[a] = chi[a]; }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1)
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64)); }
  if (!active) return;
  const bool node_empty = !(lo[0] <= hi[0]);
  DNode8& out = units[unit_of[node]];
  uint32_t exps = 0u;
  uint32_t ou[3] = {0u, 0u, 0u};
  for (int a = 0; a < 3; ++a) {
    // the origin: the grid point at or below the node's lower corner, evaluated as the traversal kernels evaluate it
    float org = grid.anchor[a];
    if (!node_empty && grid.step[a] > 0.f) {
      double u = floor(((double)lo[a] - (double)grid.anchor[a]) / (double)grid.step[a]);
      const double top = (double)((1u << kNodeOriginBits) - 1u);
      u = u < 0.0 ? 0.0 : (u > top ? top : u);
      ou[a] = (uint32_t)u;
      while (ou[a] > 0u && fmaf((float)ou[a], grid.step[a], grid.anchor[a]) > lo[a]) --ou[a];   // what rounding leaves open
      org = fmaf((float)ou[a], grid.step[a], grid.anchor[a]);
    }
    // unit 2^e with 255 units covering the extent above the origin; the loop settles what rounding leaves open
    int e = -126;
    const double ext = node_empty ? 0.0 : (double)hi[a] - (double)org;
    if (ext > 0.0) { (void)frexp(ext / 255.0, &e); if (e < -126) e = -126; }
    while (e < 127 && ceil(ext / ldexp(1.0, e)) > 255.0) ++e;
    exps |= (uint32_t)(e + 127) << (8 * a);
    const double inv_unit = ldexp(1.0, -e);
    int ql = 255, qh = 0;   // an unused slot, or a child without a hittable primitive: upper plane below lower plane, entered by no ray
    if (!node_empty && clo[0] <= chi[0]) {
      ql = (int)floor(((double)clo[a] - (double)org) * inv_unit);
      qh = (int)ceil(((double)chi[a] - (double)org) * inv_unit);
      ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh > 255 ? 255 : (qh < 0 ? 0 : qh);
    }
    reinterpret_cast<uint8_t*>(out.qlo[a])[s] = (uint8_t)ql;
    reinterpret_cast<uint8_t*>(out.qhi[a])[s] = (uint8_t)qh;
  }
  if (s == 0) {
    nlo[node] = node_empty ? make_float4(kBig, kBig, kBig, 0.f) : make_float4(lo[0], lo[1], lo[2], 0.f);
    nhi[node] = node_empty ? make_float4(-kBig, -kBig, -kBig, 0.f) : make_float4(hi[0], hi[1], hi[2], 0.f);
    out.org[0] = ou[0] | (ou[1] << 21);
    out.org[1] = (ou[1] >> 11) | (ou[2] << 10);
    const int n_child = m.n_inner + m.n_leaf;
    out.exps = exps | ((uint32_t)m.n_inner << 24) | ((uint32_t)(n_child - 1) << 28);
    out.inner_base = m.unit_base;
  }
}

struct Temp {
  void* p = nullptr;
  ~Temp() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
  template <class T> T* as() const { return (T*)p; }
};

}  // namespace

#define BUILD_TRY(expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) { err = std::string(#expr ": ") + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP; } \
  } while (0)

int build_wide_tree(const DScene<float>& sc, const uint8_t* d_dead, WideTree& out, std::string& err) {
  const auto t0 = std::chrono::steady_clock::now();
  const int ns = sc.n_spheres, nt = sc.n_tris, nb = sc.n_boxes;
  const int n = ns + nt + nb;
  out.n_nodes = out.n_leaves = out.levels = 0;
  if (n <= 0) return RTX_OK;
  hipStream_t stream = nullptr;
  const unsigned grid = (unsigned)((n + 255) / 256);
  Temp lo4, hi4, codes_a, codes_b, ids_a, ids_b, range, meta, leaf_range, unit_of, ctr, sort_tmp, nlo, nhi;
  if (!lo4.alloc((size_t)n * 16) || !hi4.alloc((size_t)n * 16) || !codes_a.alloc((size_t)n * 8) || !codes_b.alloc((size_t)n * 8) || !ids_a.alloc((size_t)n * 4) ||
      !ids_b.alloc((size_t)n * 4) || !range.alloc((size_t)n * 8) || !meta.alloc((size_t)n * sizeof(NodeMeta)) || !leaf_range.alloc((size_t)n * 8) ||
      !unit_of.alloc((size_t)n * 4) || !ctr.alloc(sizeof(BuildCounters))) { err = "tree build: temporary buffers"; return RTX_ERR_OUT_OF_MEMORY; }
  BuildCounters hc;
  hc.nodes = 1u; hc.leaves = 0u; hc.units = 2u;   // the root's record is unit 0; unit 1 pads to an even start
  for (int a = 0; a < 3; ++a) { hc.cmin[a] = 0xffffffffu; hc.cmax[a] = 0u; hc.bmin[a] = 0xffffffffu; hc.bmax[a] = 0u; }
  BUILD_TRY(hipMemcpyAsync(ctr.p, &hc, sizeof hc, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(prim_box_kernel, dim3(grid < 2048u ? grid : 2048u), dim3(256), 0, stream, sc.spheres, ns, sc.tris, nt, sc.boxes, nb, d_dead, lo4.as<float4>(), hi4.as<float4>(),
                     ctr.as<BuildCounters>());
  hipLaunchKernelGGL(morton_kernel, dim3(grid), dim3(256), 0, stream, lo4.as<float4>(), hi4.as<float4>(), n, ctr.as<BuildCounters>(), codes_a.as<uint64_t>(),
                     ids_a.as<uint32_t>());
  size_t tmp_bytes = 0;
  BUILD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, codes_a.as<uint64_t>(), codes_b.as<uint64_t>(), ids_a.as<uint32_t>(), ids_b.as<uint32_t>(), (size_t)n, 0u, 63u, stream));
  if (!sort_tmp.alloc(tmp_bytes)) { err = "tree build: sort buffer"; return RTX_ERR_OUT_OF_MEMORY; }
  BUILD_TRY(rocprim::radix_sort_pairs(sort_tmp.p, tmp_bytes, codes_a.as<uint64_t>(), codes_b.as<uint64_t>(), ids_a.as<uint32_t>(), ids_b.as<uint32_t>(), (size_t)n, 0u, 63u, stream));
  const uint64_t* codes = codes_b.as<uint64_t>();
  const uint32_t* order = ids_b.as<uint32_t>();
  // top-down: level L's nodes are [level_first[L], level_first[L + 1])
  const int2 root = make_int2(0, n);
  BUILD_TRY(hipMemcpyAsync(range.p, &root, sizeof root, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemsetAsync(unit_of.p, 0, 4, stream));
  std::vector<int> level_first{0};
  int first = 0, count = 1;
  while (count > 0) {
    hipLaunchKernelGGL(split_level_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, codes, range.as<int2>(), meta.as<NodeMeta>(),
                       leaf_range.as<int2>(), unit_of.as<int32_t>(), first, count, ctr.as<BuildCounters>());
    BUILD_TRY(hipMemcpyAsync(&hc, ctr.p, sizeof hc, hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    first += count;
    level_first.push_back(first);
    count = (int)hc.nodes - first;
    if (level_first.size() > 200) { err = "tree build: too many levels"; return RTX_ERR_HIP; }
  }
  const int n_nodes = (int)hc.nodes, n_leaves = (int)hc.leaves, levels = (int)level_first.size() - 1;
  // one array of 64-byte units holds the node records and the leaf records; sixteen spare records of zeros behind it (degenerate
  // triangles, nothing to hit): a node whose box is a single point cannot keep rays out
  // of its unused slots, such a slot then reads a neighbouring record or one of these
  const size_t n_units = (size_t)hc.units + 16;
  if (n_units > 0x7fffffffull) { err = "tree build: more than 2^31 record units"; return RTX_ERR_UNSUPPORTED; }
  if (!out.nodes.ensure(n_units * sizeof(DNode8)) || !nlo.alloc((size_t)n_nodes * 16) || !nhi.alloc((size_t)n_nodes * 16)) {
    err = "tree build: node and leaf records"; return RTX_ERR_OUT_OF_MEMORY;
  }
  BUILD_TRY(hipMemsetAsync(out.nodes.ptr, 0, n_units * sizeof(DNode8), stream));
  OriginGrid og;   // hc holds the primitive bounds since the first level's read-back
  for (int a = 0; a < 3; ++a) {
    const bool any = hc.bmin[a] <= hc.bmax[a] && hc.bmin[a] != 0xffffffffu;
    const float mn = any ? ord2f(hc.bmin[a]) : 0.f, mx = any ? ord2f(hc.bmax[a]) : 0.f;
    og.anchor[a] = mn;
    og.step[a] = mx > mn ? (float)(((double)mx - (double)mn) / (double)(1u << kNodeOriginBits)) : 0.f;
    out.anchor[a] = og.anchor[a]; out.step[a] = og.step[a];
  }
  for (int L = levels - 1; L >= 0; --L) {
    const int f = level_first[(size_t)L], c = level_first[(size_t)L + 1] - f;
    hipLaunchKernelGGL(emit_level_kernel, dim3((unsigned)(((size_t)c * 8 + 255) / 256)), dim3(256), 0, stream, f, c, meta.as<NodeMeta>(), leaf_range.as<int2>(), order,
                       lo4.as<float4>(), hi4.as<float4>(), nlo.as<float4>(), nhi.as<float4>(), (DNode8*)out.nodes.ptr, unit_of.as<int32_t>(), sc.tris, ns, nt, og);
  }
  BUILD_TRY(hipGetLastError());
  float4 root_lo, root_hi;
  BUILD_TRY(hipMemcpyAsync(&root_lo, nlo.p, sizeof root_lo, hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipMemcpyAsync(&root_hi, nhi.p, sizeof root_hi, hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipStreamSynchronize(stream));
  out.lo[0] = root_lo.x; out.lo[1] = root_lo.y; out.lo[2] = root_lo.z; out.hi[0] = root_hi.x; out.hi[1] = root_hi.y; out.hi[2] = root_hi.z;
  out.n_nodes = n_nodes; out.n_leaves = n_leaves; out.levels = levels; out.n_units = n_units;
  out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return RTX_OK;
}

}  // namespace rtx
