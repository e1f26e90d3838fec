// Host-side helpers of the backend: device buffers, the uploaded scene, kernel launchers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/rtx_hip.h"
#include "pt_types.hpp"

namespace rtx {

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes);   // grow-only; contents are NOT preserved on growth
  bool upload(const void* src, size_t bytes);
  void release();
};

// Allocator whose construct() default-initialises: a std::vector of PODs built with it does not zero its
// storage, so the pages of a large staging array are first touched by the threads that fill it.
template <class T> struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using Staging = std::vector<T, NoInitAlloc<T>>;

// What the reference-order build takes from the scene description: every bounded primitive's box and the key the median
// splits move around (its centroid and its number in the reference's primitive list: spheres, then triangles, then boxes).
// Once this is made the caller's arrays are no longer needed.
struct BuildBox { double lo[3], hi[3]; };
struct BuildKey { double c[3]; int32_t number, pad; };
struct BuildInput {
  Staging<BuildBox> boxes;
  Staging<BuildKey> keys;
  size_t n_spheres = 0, n_tris = 0, n_boxes = 0;
  bool any_flat = false;    // some primitive's box is flat along an axis (only such primitives can be unreachable for the reference)
  double seconds = 0.0;     // time it took to make
};
void make_build_input(const rtx_scene_desc& d, BuildInput& out);

// The reference-order BVH on the host (ref: engine/bvh.hpp:148-218 over the primitive list
// of engine/scene.hpp:496-533).
struct HostBvh {
  Staging<DNodeRef> nodes;
  Staging<DPrimRef> prims;
  double seconds = 0.0;
  void build(const rtx_scene_desc& d);
  void build_from(BuildInput& in);   // consumes the keys
};
size_t reference_tree_nodes(size_t n_prims);   // node count of the reference-order tree over n primitives (a function of n alone)

// The eight-wide tree of the single-precision kernels, built on the GPU (bvh_build.hip).
struct WideTree {
  DeviceBuffer nodes;           // one array of 64-byte units: DNode8 and DLeaf records
  size_t n_units = 0;
  int n_nodes = 0, n_leaves = 0, levels = 0;
  float lo[3] = {0.f, 0.f, 0.f}, hi[3] = {0.f, 0.f, 0.f};   // the root's box
  float anchor[3] = {0.f, 0.f, 0.f}, step[3] = {0.f, 0.f, 0.f};   // grid of the node origins (DScene::wt_anchor, wt_step)
  double seconds = 0.0;         // device + host time of the build
  void release() { nodes.release(); n_nodes = n_leaves = levels = 0; n_units = 0; }
};
// `sc`: the single-precision scene with its primitive tables already on the device; `d_dead`: optional device array, one byte per
// bounded primitive (numbered spheres, triangles, boxes), non-zero = the reference can never hit it (built in as unhittable).
int build_wide_tree(const DScene<float>& sc, const uint8_t* d_dead, WideTree& out, std::string& err);

// PathPhoton: the photon map (kd-tree in the reference's layout, photon data, run and group boxes) from the records of the photon
// pass, on the device (photon_build.hip).  `raw`: device array of 10 planes of n values (position, -direction, power, stored flag).
// Waits for `stream`.  n_stored = photons in the map.
template <class R>
int build_photon_map(const R* raw, uint32_t n, DeviceBuffer& nodes, DeviceBuffer& data, DeviceBuffer& boxes, int32_t& n_stored, hipStream_t stream, std::string& err);

// What an upload derives from the description on the host alone: computed once, shared by every device that receives the scene.
struct ScenePrep {
  HostBvh bvh;
  Staging<int32_t> rank;        // per bounded primitive (numbered spheres, triangles, boxes): its position in the reference's leaf order
  std::vector<uint8_t> dead;    // per bounded primitive: non-zero = the reference can never hit it; empty when there is none
  std::vector<int32_t> emissive;
};
// Validates the description and fills `out`; RTX_OK or an error code with `err` set.
// prepare_scene = validate_scene, then build_scene_prep.  build_scene_prep makes no device call: a caller may run it on a host
// thread of its own while the scene's tables are on their way to the device (HostScene::upload_geometry).
int validate_scene(const rtx_scene_desc& d, std::string& err);
void build_scene_prep(const rtx_scene_desc& d, ScenePrep& out);
int prepare_scene(const rtx_scene_desc& d, ScenePrep& out, std::string& err);

// The reference-order tree of a LARGE scene, built on a host thread of its own after rtx_upload_scene has returned.  The single-
// precision kernels need that tree only for exact ties in t (DScene::rank) and for the primitives the reference can never hit -
// and those need a primitive with a flat box, which is checked up front -, so such a scene renders at once; the double-precision
// kernels, the validation entry points and a frame in which a tie did occur wait for it.  The thread builds the tree, derives the
// ranks and copies nodes, primitives and ranks to every device of the upload (on a stream of its own); HostScene::adopt_tree then
// only takes the buffers over.
struct RefTreeJob;
struct RefTreeBuffers { DeviceBuffer nodes, prims, rank; double seconds = 0.0; };
// NULL when the scene is built the ordinary way: fewer than kLazyTreeMinPrims bounded primitives, a primitive with a flat box, or
// RTX_EAGER_TREE set in the environment.  `in` is consumed when a job is started.
constexpr size_t kLazyTreeMinPrims = 1u << 18;
std::shared_ptr<RefTreeJob> start_ref_tree_job(BuildInput& in, const std::vector<int>& devices);
bool ref_tree_job_done(const RefTreeJob& job);
// Waits for the job and moves the buffers of device slot `slot` (the position in `devices`) out of it.
int ref_tree_job_take(RefTreeJob& job, int slot, RefTreeBuffers& out, std::string& err);

struct HostScene {
  DScene<float> dev_f;
  DScene<double> dev_d;
  DSceneRef dev_ref;
  int n_nodes = 0, n_prims = 0;
  double build_seconds = 0.0;
  WideTree wide;                    // walked by the single-precision kernels when the scene has more than kLinearMaxPrims bounded primitives
  std::vector<DeviceBuffer> pool;   // owns every other device allocation of the scene
  std::vector<rtx_sphere> h_spheres;   // host copies the PathPhoton photon pass is planned from
  std::vector<rtx_material> h_materials;
  // Copies the scene to the current device and builds the eight-wide tree there, in two steps: the tables that do not depend on
  // the reference-order tree (a validated scene), then the tree, what derives from it, and the eight-wide tree; `prep` =
  // prepare_scene(d).
  int upload_geometry(const rtx_scene_desc& d, std::string& err);
  int upload_tree(const rtx_scene_desc& d, const ScenePrep& prep, std::string& err);
  // The second step for a scene whose reference-order tree is built in the background: counts, the eight-wide tree, no ranks yet.
  int upload_tree_deferred(const rtx_scene_desc& d, std::shared_ptr<RefTreeJob> job, int slot, std::string& err);
  std::shared_ptr<RefTreeJob> job;   // non-NULL until the tree has been adopted
  int job_slot = -1;
  bool tree_pending() const { return job != nullptr; }
  // Takes the finished tree over (device pointers into dev_d / dev_f / dev_ref).  wait = false: only if the job is done.
  // RTX_OK with tree_pending() still true means "not done yet".
  int adopt_tree(bool wait, std::string& err);
  int build_walk_tree(const uint8_t* d_dead, std::string& err);   // the eight-wide tree, on the current device
  int upload(const rtx_scene_desc& d, const ScenePrep& prep, std::string& err) {
    const int rc = upload_geometry(d, err);
    return rc != RTX_OK ? rc : upload_tree(d, prep, err);
  }
  int upload(const rtx_scene_desc& d, std::string& err) {
    ScenePrep prep;
    const int rc = prepare_scene(d, prep, err);
    return rc != RTX_OK ? rc : upload(d, prep, err);
  }
  // Which intersection mode the integrator kernels use for this scene.
  int fused_mode(bool f64) const {
    if (const char* e = getenv("RTX_FORCE_MODE")) { int m = atoi(e); if (m == MODE_TREE || m == MODE_TREE_UV) return m; }   // diagnostics only
    const bool spheres_only = dev_f.n_tris == 0 && dev_f.n_boxes == 0 && dev_f.n_spheres <= kFlatMaxSpheres;
    if (!f64 && spheres_only && dev_f.n_images == 0) return MODE_FLAT;
    return dev_f.n_images > 0 ? MODE_TREE_UV : MODE_TREE;
  }
  void release();
};

// kernel launchers (kernels_f32.hip / kernels_f64.hip)
struct FusedLaunch {
  int mode;             // MODE_TREE / MODE_TREE_UV / MODE_FLAT (pt_device.hpp)
  bool lds_mats;        // material table staged in LDS
  int grid;
  size_t lds_bytes;     // dynamic LDS per workgroup
  int stack_first_int;  // offset (in ints) of the traversal stacks inside the dynamic LDS
  int integ = INTEG_PATH;  // INTEG_*: path tracing, path tracing with MNEE (double precision only), Whitted (TREE_UV instances only)
  bool mnee_batched = true;   // single-precision manifold NEE: the solves of a wave batched (render_fused_mnee_batched_kernel); RTX_MNEE_ON_THE_SPOT=1 in the environment launches the instance that solves at the vertex
  int regroup = 0;         // experiment (RTX_FUSED_REGROUP): > 0 launches the material-coherent variant of the lean flat path tracer; lanes trade places when a wave has at least this many lanes outside its majority class
};
// PathPhoton photon pass (pt_photon.hpp): `out` = 10 planes of n_photons values (position, -direction, power, valid)
void launch_photon_trace(const DScene<float>& sc, const DRenderParams& P, const DPhotonLight* lights, int n_lights, const int32_t* targets, int n_targets,
                         uint32_t n_photons, float* out, DCounters* counters, size_t lds_bytes, hipStream_t stream);
void launch_photon_trace(const DScene<double>& sc, const DRenderParams& P, const DPhotonLight* lights, int n_lights, const int32_t* targets, int n_targets,
                         uint32_t n_photons, double* out, DCounters* counters, size_t lds_bytes, hipStream_t stream);
// BDPT (pt_bdpt.hpp): `light_store` holds kBdptFields * store_cap values per lane of the grid (store_cap = max_depth + 1 kept light vertices)
int launch_render_bdpt(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* partial, DCounters* counters, const FusedLaunch& L,
                       float* light_store, unsigned long long store_cap, hipStream_t stream);
int launch_render_bdpt(const DScene<double>& sc, const DCamera<double>& cam, const DRenderParams& P, double* partial, DCounters* counters, const FusedLaunch& L,
                       double* light_store, unsigned long long store_cap, hipStream_t stream);
int launch_render_fused(const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* partial, DCounters* counters,
                        const FusedLaunch& L, hipStream_t stream);
int launch_render_fused(const DScene<double>& sc, const DCamera<double>& cam, const DRenderParams& P, double* partial, DCounters* counters,
                        const FusedLaunch& L, hipStream_t stream);
void launch_film_reduce(const float* partial, float* film, unsigned long long npix, int n_chunks, int spp, hipStream_t stream);
void launch_film_reduce(const double* partial, float* film, unsigned long long npix, int n_chunks, int spp, hipStream_t stream);
void launch_postprocess(const float* film, unsigned char* out, int width, int height, int mapper, double exposure, hipStream_t stream);
int launch_trace_fast(const DScene<float>& sc, const double* rays8, unsigned long long n, DHitOut* out, int mode, size_t lds_bytes, hipStream_t stream, int n_cus = 256);
int launch_trace_fast(const DScene<double>& sc, const double* rays8, unsigned long long n, DHitOut* out, int mode, size_t lds_bytes, hipStream_t stream, int n_cus = 256);
// Grid of the production closest-hit kernel for a batch of n rays: grid-stride over the batch, a few workgroups per CU.
inline unsigned trace_fast_grid(unsigned long long n, int n_cus) {
  const unsigned long long want = (n + kBlock - 1) / kBlock, cap = (unsigned long long)n_cus * 8ull;
  return (unsigned)(want < cap ? want : cap);
}
void launch_trace_ref(const DScene<double>& sc, const DSceneRef& ref, const double* rays8, unsigned long long n, DHitOut* out, hipStream_t stream);
void launch_selftest(float* out, int n, hipStream_t stream);
void launch_zero_words(uint32_t* p, unsigned n, hipStream_t stream);   // p[0 .. n) = 0 by a kernel (no memset node)

// Wavefront pipeline (wavefront.hip): queue storage + the per-iteration launch loop.
struct WavefrontPool {
  DeviceBuffer state, queues, control, sort_tmp;
  bool sort_rays = false;   // this render sorts its ray queues before the traversal kernels take them
  bool sort_materials = false;   // this render orders the shade kernel's entries by material class (experiment)
  // HIP events around every launch of the closest-hit traversal kernel of the last render (pairs: before, after)
  std::vector<hipEvent_t> events;
  size_t events_used = 0;
  bool timing_ok = true;             // every mark of the last render succeeded (else its pairs would be mismatched)
  // The any-hit walk of a turn runs on a stream of its own, beside the closest-hit walk (the two are independent): made at first use
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap = false;              // this render runs the two walks of a turn side by side
  bool ensure_side();
  bool mark(hipStream_t stream);     // record the next event on `stream`
  double timed_seconds() const;      // sum over the pairs (0 if a mark failed); call after the stream has been synchronised
  void release();
};
int wavefront_render(WavefrontPool& pool, const DScene<float>& sc, const DCamera<float>& cam, const DRenderParams& P, float* d_film,
                     DCounters* counters, int n_cus, int mode, int stack_levels, hipStream_t stream, uint32_t& launches, std::string& err);

}  // namespace rtx
