// Scene upload: host BVH construction, float/double device copies, environment tables.
#include "pt_host.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <future>
#include <mutex>
#include <thread>
#include <type_traits>

namespace rtx {

// ------------------------------------------------------------------------------------------
bool DeviceBuffer::ensure(size_t bytes) {
  if (bytes <= cap && ptr) return true;
  if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
  if (bytes == 0) bytes = 16;
  if (hipMalloc(&ptr, bytes) != hipSuccess) { ptr = nullptr; return false; }
  cap = bytes;
  return true;
}
bool DeviceBuffer::upload(const void* src, size_t bytes) {
  if (!ensure(bytes)) return false;
  if (bytes == 0) return true;
  return hipMemcpy(ptr, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
}
void DeviceBuffer::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr; cap = 0;
}
void WavefrontPool::release() {
  state.release(); queues.release(); control.release(); sort_tmp.release();
  for (hipEvent_t e : events) (void)hipEventDestroy(e);
  events.clear(); events_used = 0;
  if (side) (void)hipStreamDestroy(side);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_join) (void)hipEventDestroy(ev_join);
  side = nullptr; ev_fork = nullptr; ev_join = nullptr;
}
bool WavefrontPool::ensure_side() {
  if (side && ev_fork && ev_join) return true;
  if (!side && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; return false; }
  if (!ev_fork && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) { ev_fork = nullptr; return false; }
  if (!ev_join && hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) { ev_join = nullptr; return false; }
  return true;
}
bool WavefrontPool::mark(hipStream_t stream) {
  if (events_used == events.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return false;
    events.push_back(e);
  }
  return hipEventRecord(events[events_used++], stream) == hipSuccess;
}
double WavefrontPool::timed_seconds() const {
  double s = 0.0;
  if (!timing_ok) return 0.0;
  for (size_t i = 0; i + 1 < events_used; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, events[i], events[i + 1]) == hipSuccess) s += ms * 1e-3;
  }
  return s;
}

// ------------------------------------------------------------------------------------------
// Reference-order BVH.  Median split on the longest centroid axis with std::nth_element,
// leaves of at most two primitives, nodes stored in preorder (ref: engine/bvh.hpp:148-218).
// The number of nodes under a range depends only on its length, so every subtree's slot in
// the preorder array is known up front and large subtrees can be built concurrently without
// changing a single node.
// ------------------------------------------------------------------------------------------
namespace {

// f(begin, end) over [0, n) on up to 16 host threads (the conversion loops of a large scene).
template <class F> void parallel_for(size_t n, F&& f) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1u ? 1u : nt > 16u ? 16u : nt;
  if (n < (size_t)1 << 16 || nt == 1u) { f((size_t)0, n); return; }
  const size_t chunk = (n + nt - 1) / nt;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t b = (size_t)t * chunk, e = b + chunk < n ? b + chunk : n;
    if (b >= e) break;
    th.emplace_back([&f, b, e] { f(b, e); });
  }
  for (std::thread& t : th) t.join();
}

// (BuildBox / BuildKey, pt_host.hpp: a primitive's box, numbered as the reference's primitive list, and what the median splits
// move around: its centroid (ref: bvh.hpp:89-95) and that number.  32-byte keys instead of the reference's 56-byte BVHPrimitive:
// std::nth_element makes the same comparisons and therefore the same moves, on fewer bytes.)

size_t subtree_nodes(size_t count) {
  if (count <= 2) return 1;
  size_t l = count / 2;
  return 1 + subtree_nodes(l) + subtree_nodes(count - l);
}

struct Builder {
  const Staging<BuildBox>& boxes;
  Staging<BuildKey>& keys;
  Staging<DNodeRef>& nodes;

  void build(size_t me, size_t start, size_t end, int depth) {
    DNodeRef& nd = nodes[me];
    const size_t count = end - start;
    if (count <= 2) {
      double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};   // ref: bvh.hpp:26-27
      for (size_t i = start; i < end; ++i) {
        const BuildBox& p = boxes[(size_t)keys[i].number];
        for (int a = 0; a < 3; ++a) {
          // AABB::expand(other): expand(min) then expand(max), each an fmin + an fmax (bvh.hpp:69-84)
          lo[a] = std::fmin(std::fmin(lo[a], p.lo[a]), p.hi[a]);
          hi[a] = std::fmax(std::fmax(hi[a], p.lo[a]), p.hi[a]);
        }
      }
      for (int a = 0; a < 3; ++a) { nd.lo[a] = lo[a]; nd.hi[a] = hi[a]; }
      nd.left = -1; nd.right = -1; nd.first = (int32_t)start; nd.count = (int32_t)count;
      return;
    }
    double clo[3] = {1e30, 1e30, 1e30}, chi[3] = {-1e30, -1e30, -1e30};
    // min / max are exact and order-independent, so a large range is reduced in slices
    auto bounds = [&](size_t b, size_t e, double* clo, double* chi) {
      for (size_t i = b; i < e; ++i)
        for (int a = 0; a < 3; ++a) { const double c = keys[i].c[a]; clo[a] = std::fmin(clo[a], c); chi[a] = std::fmax(chi[a], c); }
    };
    if (count > ((size_t)1 << 20)) {
      std::mutex mu;
      parallel_for(count, [&](size_t b, size_t e) {
        double cl[3] = {1e30, 1e30, 1e30}, ch[3] = {-1e30, -1e30, -1e30};
        bounds(start + b, start + e, cl, ch);
        std::lock_guard<std::mutex> g(mu);
        for (int a = 0; a < 3; ++a) { clo[a] = std::fmin(clo[a], cl[a]); chi[a] = std::fmax(chi[a], ch[a]); }
      });
    } else {
      bounds(start, end, clo, chi);
    }
    double dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
    int axis = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);   // ref: bvh.hpp:100-107
    size_t mid = start + count / 2;
    std::nth_element(keys.begin() + start, keys.begin() + mid, keys.begin() + end,
                     [axis](const BuildKey& a, const BuildKey& b) { return a.c[axis] < b.c[axis]; });
    size_t left = me + 1, right = me + 1 + subtree_nodes(mid - start);
    nd.left = (int32_t)left; nd.right = (int32_t)right; nd.first = 0; nd.count = 0;
    if (count > 32768 && depth < 6) {
      auto fut = std::async(std::launch::async, [=] { build(left, start, mid, depth + 1); });
      build(right, mid, end, depth + 1);
      fut.get();
    } else {
      build(left, start, mid, depth + 1);
      build(right, mid, end, depth + 1);
    }
    // The reference expands a node's box by every primitive of its range in turn (bvh.hpp:176-180); fmin / fmax are exact,
    // commutative and associative, so the union of the two children's boxes is that box.
    const DNodeRef& l = nodes[left];
    const DNodeRef& r = nodes[right];
    for (int a = 0; a < 3; ++a) { nd.lo[a] = std::fmin(l.lo[a], r.lo[a]); nd.hi[a] = std::fmax(l.hi[a], r.hi[a]); }
  }
};

// round a double to float toward -inf / +inf
float f_down(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; }
float f_up(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; }

template <class T, class S> void cp3(T* dst, const S* src) { dst[0] = (T)src[0]; dst[1] = (T)src[1]; dst[2] = (T)src[2]; }

template <class T, class A> const T* push(std::vector<DeviceBuffer>& pool, const std::vector<T, A>& v, bool& ok) {
  pool.emplace_back();
  if (!pool.back().upload(v.data(), v.size() * sizeof(T))) ok = false;
  return (const T*)pool.back().ptr;
}

// Luminance-weighted sampling tables of the environment map (ref: engine/texture.hpp:403-473).
void build_env_tables(const rtx_environment& e, std::vector<double>& marginal, std::vector<double>& conditional, double& total) {
  const double kPi = 3.14159265358979323846;
  const int w = e.width, h = e.height;
  marginal.assign(h + 1, 0.0);
  conditional.assign((size_t)h * (w + 1), 0.0);
  std::vector<double> rows(h);
  for (int y = 0; y < h; ++y) {
    double st = std::sin(kPi * (y + 0.5) / h);
    double sum = 0.0;
    size_t off = (size_t)y * (w + 1);
    for (int x = 0; x < w; ++x) {
      const float* t = e.rgb + ((size_t)y * w + x) * 3;
      double lum = 0.2126 * t[0] + 0.7152 * t[1] + 0.0722 * t[2];
      sum += lum * st;
      conditional[off + x + 1] = sum;
    }
    if (sum > 0.0) for (int x = 1; x <= w; ++x) conditional[off + x] /= sum;
    else for (int x = 1; x <= w; ++x) conditional[off + x] = (double)x / w;
    rows[y] = sum;
  }
  for (int y = 0; y < h; ++y) marginal[y + 1] = marginal[y] + rows[y];
  total = marginal[h];
  if (total > 0.0) for (int y = 1; y <= h; ++y) marginal[y] /= total;
  else for (int y = 1; y <= h; ++y) marginal[y] = (double)y / h;
}

// Everything of the device scene that does not depend on the reference-order tree.
template <class R> void fill_geometry(DScene<R>& sc, std::vector<DeviceBuffer>& pool, const rtx_scene_desc& d,
                                      const std::vector<int32_t>& emissive, const DImage* d_images,
                                      const float* d_env_texels, const std::vector<double>& marginal, const std::vector<double>& conditional,
                                      double env_total, bool& ok) {
  memset(&sc, 0, sizeof sc);
  std::vector<DSphere<R>> sph(d.n_spheres);
  for (size_t i = 0; i < d.n_spheres; ++i) { cp3(sph[i].c, d.spheres[i].center); sph[i].radius = (R)d.spheres[i].radius; sph[i].mat = d.spheres[i].material_id; sph[i].pad = 0; }
  std::vector<DPlane<R>> pls(d.n_planes);
  for (size_t i = 0; i < d.n_planes; ++i) { cp3(pls[i].p, d.planes[i].point); cp3(pls[i].n, d.planes[i].normal); pls[i].mat = d.planes[i].material_id;
    pls[i].pd = (float)(d.planes[i].normal[0] * d.planes[i].point[0] + d.planes[i].normal[1] * d.planes[i].point[1] + d.planes[i].normal[2] * d.planes[i].point[2]);
  }
  // rtx_triangle and DTri<double> have the same layout: the double-precision scene takes the caller's array as it is
  static_assert(sizeof(DTri<double>) == sizeof(rtx_triangle) && offsetof(DTri<double>, mat) == offsetof(rtx_triangle, material_id), "layout");
  Staging<DTri<R>> tris;
  if constexpr (std::is_same<R, double>::value) {
    pool.emplace_back();
    if (!pool.back().upload(d.triangles, d.n_triangles * sizeof(rtx_triangle))) ok = false;
    sc.tris = (const DTri<R>*)pool.back().ptr;
  } else {
    tris.resize(d.n_triangles);
    parallel_for(d.n_triangles, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) { cp3(tris[i].v0, d.triangles[i].v0); cp3(tris[i].v1, d.triangles[i].v1); cp3(tris[i].v2, d.triangles[i].v2); tris[i].mat = d.triangles[i].material_id; tris[i].pad = 0; }
    });
  }
  std::vector<DBox<R>> boxes(d.n_boxes);
  for (size_t i = 0; i < d.n_boxes; ++i) { cp3(boxes[i].lo, d.boxes[i].box_min); cp3(boxes[i].hi, d.boxes[i].box_max); boxes[i].mat = d.boxes[i].material_id; boxes[i].pad = 0; }
  std::vector<DPoint<R>> pts(d.n_point_lights);
  for (size_t i = 0; i < d.n_point_lights; ++i) { cp3(pts[i].pos, d.point_lights[i].position); cp3(pts[i].I, d.point_lights[i].intensity); }
  std::vector<DQuad<R>> quads(d.n_quad_lights);
  for (size_t i = 0; i < d.n_quad_lights; ++i) {
    const rtx_quad_light& q = d.quad_lights[i];
    cp3(quads[i].corner, q.corner); cp3(quads[i].eu, q.edge_u); cp3(quads[i].ev, q.edge_v); cp3(quads[i].n, q.normal); cp3(quads[i].Le, q.emission);
    quads[i].area = (R)q.area;
  }
  std::vector<DDisk<R>> disks(d.n_disk_lights);
  for (size_t i = 0; i < d.n_disk_lights; ++i) {
    const rtx_disk_light& q = d.disk_lights[i];
    cp3(disks[i].c, q.center); cp3(disks[i].n, q.normal); disks[i].radius = (R)q.radius; cp3(disks[i].Le, q.emission);
    disks[i].area = (R)q.area; cp3(disks[i].tan, q.tangent); cp3(disks[i].bit, q.bitangent);
  }
  std::vector<DMaterial<R>> mats(d.n_materials);
  for (size_t i = 0; i < d.n_materials; ++i) {
    const rtx_material& m = d.materials[i];
    mats[i].type = m.type; mats[i].has_texture = m.has_texture; mats[i].tex_type = m.tex_type; mats[i].tex_image = m.tex_image;
    cp3(mats[i].albedo, m.albedo); cp3(mats[i].emission, m.emission); mats[i].fuzz = (R)m.fuzz; mats[i].ior = (R)m.refraction_index;
    cp3(mats[i].c1, m.tex_color1); cp3(mats[i].c2, m.tex_color2); mats[i].scale = (R)m.tex_scale;
    mats[i].nmap_image = m.has_normal_map ? m.normal_image : -1; mats[i].pad = 0; mats[i].nmap_strength = (R)m.normal_strength;
  }
  sc.spheres = push(pool, sph, ok); sc.planes = push(pool, pls, ok); sc.boxes = push(pool, boxes, ok);
  if constexpr (!std::is_same<R, double>::value) sc.tris = push(pool, tris, ok);
  sc.points = push(pool, pts, ok); sc.quads = push(pool, quads, ok); sc.disks = push(pool, disks, ok); sc.mats = push(pool, mats, ok);
  sc.emissive = push(pool, emissive, ok);
  sc.images = d_images;
  sc.n_spheres = (int)d.n_spheres; sc.n_planes = (int)d.n_planes; sc.n_tris = (int)d.n_triangles; sc.n_boxes = (int)d.n_boxes;
  sc.n_points = (int)d.n_point_lights; sc.n_quads = (int)d.n_quad_lights; sc.n_disks = (int)d.n_disk_lights; sc.n_mats = (int)d.n_materials;
  sc.n_images = (int)d.n_images; sc.n_emissive = (int)emissive.size();
  sc.n_nmaps = 0;
  for (size_t i = 0; i < d.n_materials; ++i) sc.n_nmaps += d.materials[i].has_normal_map ? 1 : 0;
  if (d_env_texels) {
    std::vector<R> mg(marginal.begin(), marginal.end()), cd(conditional.begin(), conditional.end());
    sc.env.texel = d_env_texels;
    sc.env.marginal = push(pool, mg, ok); sc.env.conditional = push(pool, cd, ok);
    // guides (DEnv): from the values as the device holds them, compared as the device compares them
    const int eh = d.environment->height, ew = d.environment->width;
    auto guide = [](const R* cdf, int n_plus_1, int32_t* g) {
      int i = 0;
      for (int k = 0; k <= kEnvGuide; ++k) {
        const R x = (R)k / (R)kEnvGuide;
        while (i < n_plus_1 && !(cdf[i] > x)) ++i;   // first index with cdf[i] > x (the function does not decrease)
        g[k] = i;
      }
    };
    std::vector<int32_t> gm(kEnvGuide + 1), gc((size_t)eh * (kEnvGuide + 1));
    guide(mg.data(), eh + 1, gm.data());
    for (int y = 0; y < eh; ++y) guide(cd.data() + (size_t)y * (ew + 1), ew + 1, gc.data() + (size_t)y * (kEnvGuide + 1));
    sc.env.marginal_guide = push(pool, gm, ok); sc.env.conditional_guide = push(pool, gc, ok);
    sc.env.w = d.environment->width; sc.env.h = d.environment->height;
    sc.env.intensity = (R)d.environment->intensity; sc.env.rotation = (R)d.environment->rotation;
    sc.env.valid = 1; sc.env.importance = env_total > 0.0 ? 1 : 0;   // ref: engine/texture.hpp:478-480, :692-694
  }
}

}  // namespace

// What the device scene takes from the reference-order tree.
template <class R> void fill_tree(DScene<R>& sc, std::vector<DeviceBuffer>& pool, const rtx_scene_desc& d, const HostBvh& bvh, const DNodeRef* d_nodes, const DPrimRef* d_prims,
                                  bool& ok) {
  if (std::is_same<R, float>::value && d.n_triangles == 0 && d.n_boxes == 0 && d.n_spheres <= (size_t)kFlatMaxSpheres) {
    std::vector<DFlatRec> recs;
    auto rec = [&](float a, float b, float c, float w, int mat, int type, int idx) { recs.push_back(DFlatRec{a, b, c, w, mat, type, idx, 0}); };
    for (const DPrimRef& p : bvh.prims) {   // all spheres here, in the reference's leaf order
      const rtx_sphere& s = d.spheres[p.index];
      rec((float)s.center[0], (float)s.center[1], (float)s.center[2], 1.0f / (float)s.radius, s.material_id, PRIM_SPHERE, p.index);
    }
    for (size_t i = 0; i < d.n_planes; ++i) rec((float)d.planes[i].normal[0], (float)d.planes[i].normal[1], (float)d.planes[i].normal[2], 0.f, d.planes[i].material_id, PRIM_PLANE, (int)i);
    for (size_t i = 0; i < d.n_quad_lights; ++i) rec((float)d.quad_lights[i].normal[0], (float)d.quad_lights[i].normal[1], (float)d.quad_lights[i].normal[2], 0.f, -(int)i - 1, PRIM_QUAD, (int)i);
    for (size_t i = 0; i < d.n_disk_lights; ++i) rec((float)d.disk_lights[i].normal[0], (float)d.disk_lights[i].normal[1], (float)d.disk_lights[i].normal[2], 0.f, -1000 - (int)i, PRIM_DISK, (int)i);
    sc.flat = push(pool, recs, ok);
  }
  // Double precision walks the reference-order tree itself (left child = i + 1), so that closest hits keep the reference's bits
  // whatever the visiting order.  Single precision walks the eight-wide tree built on the GPU (HostScene::upload_tree).
  sc.nodes = std::is_same<R, double>::value ? d_nodes : nullptr;
  sc.prims = d_prims;
  sc.n_nodes = (int)bvh.nodes.size(); sc.n_prims = (int)bvh.prims.size();
}

size_t reference_tree_nodes(size_t n_prims) { return n_prims ? subtree_nodes(n_prims) : 0; }

void make_build_input(const rtx_scene_desc& d, BuildInput& out) {
  const auto t0 = std::chrono::steady_clock::now();
  const size_t ns = d.n_spheres, nt = d.n_triangles, nb = d.n_boxes;
  out.n_spheres = ns; out.n_tris = nt; out.n_boxes = nb;
  Staging<BuildBox>& bp = out.boxes;
  bp.resize(ns + nt + nb);
  for (size_t i = 0; i < ns; ++i) {   // ref: engine/scene.hpp:500-510
    BuildBox& p = bp[i];
    for (int a = 0; a < 3; ++a) { p.lo[a] = d.spheres[i].center[a] - d.spheres[i].radius; p.hi[a] = d.spheres[i].center[a] + d.spheres[i].radius; }
  }
  parallel_for(nt, [&](size_t b, size_t e) {   // ref: engine/scene.hpp:513-522
    for (size_t i = b; i < e; ++i) {
      BuildBox& p = bp[ns + i];
      const rtx_triangle& t = d.triangles[i];
      for (int a = 0; a < 3; ++a) {
        p.lo[a] = std::fmin(std::fmin(std::fmin(1e30, t.v0[a]), t.v1[a]), t.v2[a]);
        p.hi[a] = std::fmax(std::fmax(std::fmax(-1e30, t.v0[a]), t.v1[a]), t.v2[a]);
      }
    }
  });
  for (size_t i = 0; i < nb; ++i) {      // ref: engine/scene.hpp:525-533
    BuildBox& p = bp[ns + nt + i];
    for (int a = 0; a < 3; ++a) { p.lo[a] = d.boxes[i].box_min[a]; p.hi[a] = d.boxes[i].box_max[a]; }
  }
  out.keys.resize(bp.size());
  std::atomic<bool> flat{false};
  parallel_for(bp.size(), [&](size_t b, size_t e) {
    bool f = false;
    for (size_t i = b; i < e; ++i) {
      for (int a = 0; a < 3; ++a) {
        out.keys[i].c[a] = (bp[i].lo[a] + bp[i].hi[a]) * 0.5;   // ref: bvh.hpp:89-95
        f = f || !(bp[i].lo[a] < bp[i].hi[a]);                   // flat (or inverted, or NaN) along this axis
      }
      out.keys[i].number = (int32_t)i; out.keys[i].pad = 0;
    }
    if (f) flat.store(true, std::memory_order_relaxed);
  });
  out.any_flat = flat.load();
  out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HostBvh::build(const rtx_scene_desc& d) {
  BuildInput in;
  make_build_input(d, in);
  build_from(in);
}

void HostBvh::build_from(BuildInput& in) {
  auto t0 = std::chrono::steady_clock::now();
  nodes.clear(); prims.clear();
  const size_t ns = in.n_spheres, nt = in.n_tris;
  const Staging<BuildBox>& bp = in.boxes;
  Staging<BuildKey>& keys = in.keys;
  const bool timing = getenv("RTX_TIMING") != nullptr;   // diagnostics: phase times on stderr
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[rtx upload]   reference tree: %-28s %7.1f ms since its start\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
  };
  if (!bp.empty()) {
    nodes.resize(subtree_nodes(bp.size()));
    Builder b{bp, keys, nodes};
    b.build(0, 0, bp.size(), 0);
    lap("median splits");
    prims.resize(bp.size());
    parallel_for(bp.size(), [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        const size_t k = (size_t)keys[i].number;
        prims[i].type = k < ns ? PRIM_SPHERE : k < ns + nt ? PRIM_TRIANGLE : PRIM_BOX;
        prims[i].index = (int32_t)(k < ns ? k : k < ns + nt ? k - ns : k - ns - nt);
      }
    });
  }
  seconds = in.seconds + std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HostScene::release() {
  job.reset(); job_slot = -1;   // (the last scene to let go of a job waits for its thread)
  for (DeviceBuffer& b : pool) b.release();
  pool.clear();
  wide.release();
  memset(&dev_f, 0, sizeof dev_f); memset(&dev_d, 0, sizeof dev_d); memset(&dev_ref, 0, sizeof dev_ref);
  h_spheres.clear(); h_materials.clear();
  n_nodes = n_prims = 0;
}

int validate_scene(const rtx_scene_desc& d, std::string& err) {
  const size_t lim = 0x7fffffff;
  if (d.n_spheres > lim || d.n_planes > lim || d.n_triangles > lim || d.n_boxes > lim || d.n_materials > lim ||
      d.n_spheres + d.n_triangles + d.n_boxes > lim / 2) { err = "too many primitives"; return RTX_ERR_INVALID_ARGUMENT; }
  auto need = [&](const void* p, size_t n, const char* what) { if (n && !p) { err = std::string(what) + " is NULL"; return false; } return true; };
  if (!need(d.spheres, d.n_spheres, "spheres") || !need(d.planes, d.n_planes, "planes") || !need(d.triangles, d.n_triangles, "triangles") ||
      !need(d.boxes, d.n_boxes, "boxes") || !need(d.point_lights, d.n_point_lights, "point_lights") || !need(d.quad_lights, d.n_quad_lights, "quad_lights") ||
      !need(d.disk_lights, d.n_disk_lights, "disk_lights") || !need(d.materials, d.n_materials, "materials") || !need(d.images, d.n_images, "images"))
    return RTX_ERR_INVALID_ARGUMENT;
  // The reference indexes its material array unchecked (engine/scene.hpp:705-707); an id out
  // of range would be a device fault here, so it is rejected up front.
  const int nm = (int)d.n_materials;
  auto bad = [&](int id) { return id < 0 || id >= nm; };
  for (size_t i = 0; i < d.n_spheres; ++i) if (bad(d.spheres[i].material_id)) { err = "sphere material id out of range"; return RTX_ERR_INVALID_ARGUMENT; }
  for (size_t i = 0; i < d.n_planes; ++i) if (bad(d.planes[i].material_id)) { err = "plane material id out of range"; return RTX_ERR_INVALID_ARGUMENT; }
  for (size_t i = 0; i < d.n_triangles; ++i) if (bad(d.triangles[i].material_id)) { err = "triangle material id out of range"; return RTX_ERR_INVALID_ARGUMENT; }
  for (size_t i = 0; i < d.n_boxes; ++i) if (bad(d.boxes[i].material_id)) { err = "box material id out of range"; return RTX_ERR_INVALID_ARGUMENT; }
  for (size_t i = 0; i < d.n_materials; ++i) {
    const rtx_material& m = d.materials[i];
    if (m.type < 0 || m.type > 2) { err = "material type out of range"; return RTX_ERR_INVALID_ARGUMENT; }
    if (m.has_texture && m.tex_type == RTX_TEX_IMAGE && (m.tex_image < 0 || (size_t)m.tex_image >= d.n_images)) { err = "material image index out of range"; return RTX_ERR_INVALID_ARGUMENT; }
    if (m.has_normal_map && (m.normal_image < 0 || (size_t)m.normal_image >= d.n_images)) { err = "material normal map index out of range"; return RTX_ERR_INVALID_ARGUMENT; }
  }
  for (size_t i = 0; i < d.n_images; ++i) if (!d.images[i].rgb || d.images[i].width <= 0 || d.images[i].height <= 0) { err = "bad image"; return RTX_ERR_INVALID_ARGUMENT; }
  return RTX_OK;
}

// The emissive spheres, ascending (ref: engine/scene.hpp:222-230).
static std::vector<int32_t> emissive_spheres(const rtx_scene_desc& d) {
  std::vector<int32_t> out;
  for (size_t i = 0; i < d.n_spheres; ++i) {
    const rtx_material& m = d.materials[d.spheres[i].material_id];
    if (m.emission[0] > 0.0 || m.emission[1] > 0.0 || m.emission[2] > 0.0) out.push_back((int32_t)i);
  }
  return out;
}

// What the GPU-built tree takes from the reference-order tree: every primitive's position in the reference's leaf order (ties
// in t go to the later one), and which primitives the reference can never hit (its slab test rejects a box that is flat along
// an axis, ref: engine/bvh.hpp:43, so the primitives of such a leaf are unreachable there).
static void derive_from_tree(size_t ns, size_t nt, ScenePrep& out) {
  const HostBvh& bvh = out.bvh;
  out.rank.clear(); out.dead.clear();
  if (bvh.prims.size() <= (size_t)kLinearMaxPrims) return;
  const size_t np = bvh.prims.size();
  auto number = [&](const DPrimRef& p) -> size_t { return p.type == PRIM_SPHERE ? (size_t)p.index : p.type == PRIM_TRIANGLE ? ns + (size_t)p.index : ns + nt + (size_t)p.index; };
  out.rank.resize(np);
  parallel_for(np, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) out.rank[number(bvh.prims[i])] = (int32_t)i; });
  std::mutex mu;
  parallel_for(bvh.nodes.size(), [&](size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const DNodeRef& n = bvh.nodes[i];
      if (n.count <= 0 || (n.lo[0] != n.hi[0] && n.lo[1] != n.hi[1] && n.lo[2] != n.hi[2])) continue;
      std::lock_guard<std::mutex> g(mu);
      if (out.dead.empty()) out.dead.assign(np, 0);
      for (int k = 0; k < n.count; ++k) out.dead[number(bvh.prims[(size_t)n.first + k])] = 1;
    }
  });
}

void build_scene_prep(const rtx_scene_desc& d, ScenePrep& out) {
  const bool timing = getenv("RTX_TIMING") != nullptr;   // diagnostics: phase times on stderr
  const auto t0 = std::chrono::steady_clock::now();
  out.bvh.build(d);
  out.emissive = emissive_spheres(d);
  derive_from_tree(d.n_spheres, d.n_triangles, out);
  if (timing) fprintf(stderr, "[rtx upload] %-28s %8.1f ms\n", "reference tree, ranks", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

// ------------------------------------------------------------------------------------------
// The reference-order tree in the background (RefTreeJob, pt_host.hpp)
// ------------------------------------------------------------------------------------------
struct RefTreeJob {
  BuildInput in;
  ScenePrep prep;
  std::vector<int> devices;
  std::vector<RefTreeBuffers> dev;
  std::thread th;
  std::atomic<bool> done{false};
  std::mutex take_mutex;
  bool joined = false;
  int rc = RTX_OK;
  std::string err;
  double seconds = 0.0;
  void run();
  ~RefTreeJob() {
    if (th.joinable()) th.join();
    for (size_t k = 0; k < dev.size(); ++k) {   // what no scene took over
      if (!dev[k].nodes.ptr && !dev[k].prims.ptr && !dev[k].rank.ptr) continue;
      (void)hipSetDevice(devices[k]);
      dev[k].nodes.release(); dev[k].prims.release(); dev[k].rank.release();
    }
  }
};

void RefTreeJob::run() {
  const bool timing = getenv("RTX_TIMING") != nullptr;
  if (const char* e = getenv("RTX_TREE_DELAY_MS")) {   // tests: keep the tree away for a while, so that renders run without it
    const int ms = atoi(e);
    if (ms > 0 && ms <= 60000) std::this_thread::sleep_for(std::chrono::milliseconds(ms));
  }
  const auto t0 = std::chrono::steady_clock::now();
  try {
    const size_t ns = in.n_spheres, nt = in.n_tris;
    prep.bvh.build_from(in);
    in = BuildInput();   // boxes and keys are no longer needed
    derive_from_tree(ns, nt, prep);
    seconds = prep.bvh.seconds;
    // every device gets its copy, over a stream of this thread's own (the devices may be rendering)
    for (size_t k = 0; k < devices.size() && rc == RTX_OK; ++k) {
      hipStream_t stream = nullptr;
      if (hipSetDevice(devices[k]) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { rc = RTX_ERR_HIP; err = "background upload: device or stream"; break; }
      auto put = [&](DeviceBuffer& b, const void* src, size_t bytes) {
        if (!b.ensure(bytes)) { rc = RTX_ERR_OUT_OF_MEMORY; err = "reference-order tree: device allocation"; return; }
        if (bytes && hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) { rc = RTX_ERR_HIP; err = "reference-order tree: copy to the device"; }
      };
      put(dev[k].nodes, prep.bvh.nodes.data(), prep.bvh.nodes.size() * sizeof(DNodeRef));
      if (rc == RTX_OK) put(dev[k].prims, prep.bvh.prims.data(), prep.bvh.prims.size() * sizeof(DPrimRef));
      if (rc == RTX_OK) put(dev[k].rank, prep.rank.data(), prep.rank.size() * sizeof(int32_t));
      if (hipStreamSynchronize(stream) != hipSuccess && rc == RTX_OK) { rc = RTX_ERR_HIP; err = "reference-order tree: copy to the device"; }
      (void)hipStreamDestroy(stream);
      dev[k].seconds = seconds;
    }
    if (rc == RTX_OK && !prep.dead.empty()) { rc = RTX_ERR_HIP; err = "internal: unreachable primitives in a scene without flat boxes"; }
    prep = ScenePrep();   // the host copies are no longer needed
  } catch (const std::bad_alloc&) { rc = RTX_ERR_OUT_OF_MEMORY; err = "host allocation failed (reference-order tree)";
  } catch (const std::exception& e) { rc = RTX_ERR_HIP; err = e.what(); }
  if (timing) fprintf(stderr, "[rtx upload] %-28s %8.1f ms (in the background)\n", "reference tree, ranks, copies", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  done.store(true, std::memory_order_release);
}

std::shared_ptr<RefTreeJob> start_ref_tree_job(BuildInput& in, const std::vector<int>& devices) {
  if (in.boxes.size() < kLazyTreeMinPrims || in.any_flat || devices.empty() || getenv("RTX_EAGER_TREE") != nullptr) return nullptr;
  std::shared_ptr<RefTreeJob> job(new RefTreeJob());
  job->in = std::move(in);
  job->devices = devices;
  job->dev.resize(devices.size());
  RefTreeJob* raw = job.get();   // the job outlives its thread: its destructor joins
  job->th = std::thread([raw] { raw->run(); });
  return job;
}

bool ref_tree_job_done(const RefTreeJob& job) { return job.done.load(std::memory_order_acquire); }

int ref_tree_job_take(RefTreeJob& job, int slot, RefTreeBuffers& out, std::string& err) {
  std::lock_guard<std::mutex> lock(job.take_mutex);   // the scenes of several devices may come at once
  if (!job.joined) { if (job.th.joinable()) job.th.join(); job.joined = true; }
  if (job.rc != RTX_OK) { err = job.err; return job.rc; }
  if (slot < 0 || (size_t)slot >= job.dev.size()) { err = "internal: device slot"; return RTX_ERR_INVALID_ARGUMENT; }
  out = job.dev[(size_t)slot];
  job.dev[(size_t)slot] = RefTreeBuffers();   // ownership moves: DeviceBuffer is a plain handle
  return RTX_OK;
}

int prepare_scene(const rtx_scene_desc& d, ScenePrep& out, std::string& err) {
  const int rc = validate_scene(d, err);
  if (rc != RTX_OK) return rc;
  build_scene_prep(d, out);
  return RTX_OK;
}

namespace {
struct UploadLap {
  bool on = getenv("RTX_TIMING") != nullptr;   // diagnostics: phase times of the upload on stderr
  std::chrono::steady_clock::time_point t_prev = std::chrono::steady_clock::now();
  void operator()(const char* what) {
    if (!on) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[rtx upload] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  }
};
}  // namespace

int HostScene::upload_geometry(const rtx_scene_desc& d, std::string& err) {
  release();
  UploadLap lap;
  const std::vector<int32_t> emissive = emissive_spheres(d);
  bool ok = true;
  std::vector<DImage> imgs(d.n_images);
  for (size_t i = 0; i < d.n_images; ++i) {
    pool.emplace_back();
    if (!pool.back().upload(d.images[i].rgb, (size_t)d.images[i].width * d.images[i].height * 3)) ok = false;
    imgs[i].rgb = (const uint8_t*)pool.back().ptr; imgs[i].w = d.images[i].width; imgs[i].h = d.images[i].height;
  }
  const DImage* d_images = push(pool, imgs, ok);
  const float* d_env = nullptr;
  std::vector<double> marginal, conditional;
  double env_total = 0.0;
  const rtx_environment* e = d.environment;
  if (e && e->rgb && e->width > 0 && e->height > 0) {
    build_env_tables(*e, marginal, conditional, env_total);
    pool.emplace_back();
    if (!pool.back().upload(e->rgb, (size_t)e->width * e->height * 3 * sizeof(float))) ok = false;
    d_env = (const float*)pool.back().ptr;
  }
  lap("images, env");
  fill_geometry(dev_d, pool, d, emissive, d_images, d_env, marginal, conditional, env_total, ok);
  lap("double-precision scene");
  fill_geometry(dev_f, pool, d, emissive, d_images, d_env, marginal, conditional, env_total, ok);
  lap("single-precision scene");
  h_spheres.assign(d.spheres, d.spheres + d.n_spheres);
  h_materials.assign(d.materials, d.materials + d.n_materials);
  if (!ok) { release(); err = "device allocation or copy failed"; return RTX_ERR_OUT_OF_MEMORY; }
  return RTX_OK;
}

int HostScene::upload_tree_deferred(const rtx_scene_desc& d, std::shared_ptr<RefTreeJob> j, int slot, std::string& err) {
  UploadLap lap;
  const size_t n = d.n_spheres + d.n_triangles + d.n_boxes;
  n_prims = (int)n; n_nodes = (int)reference_tree_nodes(n);
  build_seconds = 0.0;
  // the trees themselves arrive with adopt_tree
  dev_d.nodes = nullptr; dev_d.prims = nullptr; dev_d.n_nodes = n_nodes; dev_d.n_prims = n_prims;
  dev_f.nodes = nullptr; dev_f.prims = nullptr; dev_f.n_nodes = n_nodes; dev_f.n_prims = n_prims;
  dev_ref.nodes = nullptr; dev_ref.prims = nullptr; dev_ref.n_nodes = n_nodes; dev_ref.n_prims = n_prims;
  dev_f.rank = nullptr;
  job = std::move(j); job_slot = slot;
  const int rc = build_walk_tree(nullptr, err);
  lap("eight-wide tree (GPU)");
  return rc;
}

int HostScene::adopt_tree(bool wait, std::string& err) {
  if (!job) return RTX_OK;
  if (!wait && !ref_tree_job_done(*job)) return RTX_OK;
  RefTreeBuffers b;
  const int rc = ref_tree_job_take(*job, job_slot, b, err);
  job.reset(); job_slot = -1;
  if (rc != RTX_OK) return rc;
  pool.push_back(b.nodes); pool.push_back(b.prims); pool.push_back(b.rank);   // the scene owns them from here on
  const DNodeRef* d_nodes = (const DNodeRef*)b.nodes.ptr;
  const DPrimRef* d_prims = (const DPrimRef*)b.prims.ptr;
  dev_d.nodes = d_nodes; dev_d.prims = d_prims; dev_f.prims = d_prims;
  dev_ref.nodes = d_nodes; dev_ref.prims = d_prims;
  dev_f.rank = (const int32_t*)b.rank.ptr;
  build_seconds = b.seconds;
  return RTX_OK;
}

// The eight-wide tree of the single-precision kernels, built on the current device from the single-precision tables.
int HostScene::build_walk_tree(const uint8_t* d_dead, std::string& err) {
  std::string berr;
  const int rc = build_wide_tree(dev_f, d_dead, wide, berr);
  if (rc != RTX_OK) { release(); err = berr; return rc; }
  if (wide.levels > kMaxWalkLevels) {
    err = "the traversal tree of this scene is " + std::to_string(wide.levels) + " levels deep (at most " + std::to_string(kMaxWalkLevels) + " are walked)";
    release();
    return RTX_ERR_UNSUPPORTED;
  }
  dev_f.nodes8 = (const DNode8*)wide.nodes.ptr;
  dev_f.n_nodes8 = wide.n_nodes; dev_f.n_leaves8 = wide.n_leaves; dev_f.walk_levels = wide.levels;
  for (int a = 0; a < 3; ++a) {
    dev_f.bb_lo[a] = wide.lo[a]; dev_f.bb_scale[a] = wide.hi[a] > wide.lo[a] ? 128.0f / (wide.hi[a] - wide.lo[a]) : 0.f;
    dev_f.wt_anchor[a] = wide.anchor[a]; dev_f.wt_step[a] = wide.step[a];
  }
  return RTX_OK;
}

int HostScene::upload_tree(const rtx_scene_desc& d, const ScenePrep& prep, std::string& err) {
  UploadLap lap;
  const HostBvh& bvh = prep.bvh;
  build_seconds = bvh.seconds;
  n_nodes = (int)bvh.nodes.size(); n_prims = (int)bvh.prims.size();
  bool ok = true;
  const DPrimRef* d_prims = push(pool, bvh.prims, ok);
  const DNodeRef* d_nodes_ref = push(pool, bvh.nodes, ok);
  fill_tree(dev_d, pool, d, bvh, d_nodes_ref, d_prims, ok);
  fill_tree(dev_f, pool, d, bvh, d_nodes_ref, d_prims, ok);
  dev_ref.nodes = d_nodes_ref; dev_ref.prims = d_prims; dev_ref.n_nodes = n_nodes; dev_ref.n_prims = n_prims;
  lap("reference tree");
  if (ok && bvh.prims.size() > (size_t)kLinearMaxPrims) {
    dev_f.rank = push(pool, prep.rank, ok);
    const uint8_t* d_dead = prep.dead.empty() ? nullptr : push(pool, prep.dead, ok);
    lap("ranks and unreachable primitives");
    if (ok) {
      const int rc = build_walk_tree(d_dead, err);
      if (rc != RTX_OK) return rc;
    }
    lap("eight-wide tree (GPU)");
  }
  if (!ok) { release(); err = "device allocation or copy failed"; return RTX_ERR_OUT_OF_MEMORY; }
  return RTX_OK;
}

}  // namespace rtx
