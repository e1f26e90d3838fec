// Host side of the C ABI declared in include/rtx_hip.h: context, scene upload (both BVHs,
// environment tables, float/double device copies), kernel launches, error reporting.
#include "../../include/rtx_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pt_host.hpp"

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

// No exception may cross the C boundary: host allocations (staging vectors, threads) can throw.
template <class F> int guarded(F&& f) {
  try { return f(); }
  catch (const std::bad_alloc&) { return fail(RTX_ERR_OUT_OF_MEMORY, "host allocation failed"); }
  catch (const std::exception& e) { return fail(RTX_ERR_HIP, "internal error: %s", e.what()); }
  catch (...) { return fail(RTX_ERR_HIP, "internal error"); }
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

}  // namespace

using namespace rtx;

// ------------------------------------------------------------------------------------------
// Scratch memory of rtx_render_device_noalloc: the chunk sums and counters of a render live in this array of the code object
// instead of buffers the library allocates.  Every GPU the library is loaded on holds its own copy at its own address; the
// array's name is only a host-side handle, so the device address is looked up once per device (scratch_pointer) and that
// pointer, never the name, is what kernels and fills receive.
// ------------------------------------------------------------------------------------------
// Samples and work items (pixel, sample chunk) a lane of a fused launch handles at least, which sizes the grid (see the launch in
// render_enqueue).
constexpr unsigned long long kFusedSamplesPerLane = 32, kFusedMinItemsPerLane = 3, kFusedRounds = 16;
constexpr int kAutoQueuesAbovePrims = 1024;   // RTX_PIPELINE_AUTO: scenes of more bounded primitives take the queue pipeline (render_enqueue)
constexpr int kWfTargetChunks = 64;   // chunks the queue pipeline cuts a pixel's sample sum into (render_enqueue); the fused kernels: 16
constexpr size_t kScratchBytes = 64ull << 20;
constexpr size_t kScratchHeader = 256;   // the counters, ahead of the chunk sums
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};

// Device address of `scratch` on `device`, which must be the current device.  nullptr + error message on failure.
unsigned char* scratch_pointer(int device) {
  if (device < 0 || device >= kMaxDevices) { fail(RTX_ERR_INVALID_ARGUMENT, "device %d: the scratch table holds %d devices", device, kMaxDevices); return nullptr; }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) { fail(RTX_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e)); return nullptr; }
    g_scratch_ptr[device] = (unsigned char*)p;
  }
  return g_scratch_ptr[device];
}
}  // namespace

struct rtx_ctx;
static size_t stack_words(const rtx_ctx* c, bool f64);

namespace {
// A host thread that runs one job at a time, kept for the life of a multi-device context: a render hands every other device's
// share to that device's worker instead of starting a thread per call.
class Worker {
 public:
  Worker() : th_([this] { run(); }) {}
  ~Worker() {
    { std::lock_guard<std::mutex> l(m_); quit_ = true; }
    cv_.notify_all();
    th_.join();
  }
  Worker(const Worker&) = delete;
  Worker& operator=(const Worker&) = delete;
  void post(std::function<void()> job) {
    { std::lock_guard<std::mutex> l(m_); job_ = std::move(job); busy_ = true; }
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return !busy_; });
  }

 private:
  void run() {
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      cv_.wait(l, [&] { return quit_ || (busy_ && job_); });
      if (quit_) return;
      std::function<void()> j = std::move(job_);
      job_ = nullptr;
      l.unlock();
      j();   // jobs catch their own exceptions
      l.lock();
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool busy_ = false, quit_ = false;
  std::thread th_;   // last: the thread starts when every other member exists
};

// RCCL, bound at run time (rtx_multi_set_gather): the library is not linked against it.
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void**, int, const int*) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static constexpr int kFloat = 7;   // ncclFloat32 (rccl.h)
};
std::mutex g_rccl_mutex;
Rccl g_rccl;
// nullptr + *why when the library or one of its symbols is missing
const Rccl* load_rccl(std::string* why) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return &g_rccl;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) { if (why) { const char* e = dlerror(); *why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?"); } return nullptr; }
  Rccl r;
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) { ok = false; if (why) *why = std::string("librccl lacks ") + n; } return p; };
  r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(h); return nullptr; }
  r.handle = h;
  g_rccl = r;
  return &g_rccl;
}
}  // namespace

struct rtx_ctx {
  int device = 0;
  unsigned char* scratch_dev = nullptr;   // this device's copy of `scratch`
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evm = nullptr;
  int n_cus = 256;
  size_t lds_per_block = 65536;           // dynamic LDS one workgroup may ask for on this device
  bool has_scene = false, has_camera = false;
  HostScene scene;          // device-resident copies + BVHs
  DCamera<float> cam_f;
  DCamera<double> cam_d;
  DeviceBuffer partial, film, counters, rays, hits, post_in, post_out, bdpt_store;
  DeviceBuffer tie_list, repair_counters;   // work items with a tie decided before the reference-order tree arrived; the counters of their repair pass
  DeviceBuffer ph_lights, ph_targets, ph_out, ph_nodes, ph_data, ph_boxes, ph_cand;   // PathPhoton: photon pass plan, its raw output, the kd-tree
  WavefrontPool wf;         // wavefront-pipeline queues, allocated on first use
  // The photon map in ph_nodes / ph_data / ph_boxes and what it was built from: a render with the same inputs reuses it
  // (the map depends on the scene, the seed and generator, the photon count and the precision, not on camera or samples).
  struct PhotonMapKey {
    bool valid = false;
    unsigned long long scene_generation = 0, photon_count = 0;
    uint32_t seed = 0; int rng = 0, f64 = 0, bucket = 0;
    int32_t n_photons = 0;              // photons stored
    unsigned long long closest = 0;     // closest-hit queries the photon pass made
  } ph_map;
  unsigned long long scene_generation = 0;   // counts rtx_upload_scene calls
  // The render whose kernels are enqueued and whose statistics have not been read yet (render_enqueue -> render_collect).
  // Its counters arrive in h_counters, pinned host memory, so that the copy does not hold the host up.
  DCounters* h_counters = nullptr;
  struct Pending {
    bool active = false;
    hipStream_t stream = nullptr;
    bool photon_map_reused = false, wavefront = false, f64 = false;
    int pipeline = 0;
    std::chrono::steady_clock::time_point wall0;
    rtx_stats st;   // what is known when the kernels are enqueued
    // The render ran while the reference-order tree was still being built (HostScene::tree_pending): exact ties in t were decided
    // by primitive number and counted.  If there was one, render_collect waits for the tree and renders the call again.
    bool optimistic = false;
    rtx_settings settings;
    float* host_out = nullptr;
    void* dev_out = nullptr;
    hipStream_t user_stream = nullptr;
    bool no_alloc = false, stream_as_given = false;
  } pending;
  // Multi-device context (rtx_create_multi): this is the context of the first listed device; `peers` are the contexts of the
  // others (owned), each with a host thread of its own (`workers`, same order); `gather` holds the peers' rows on this device
  // until they are interleaved into the caller's film.
  bool multi = false;
  std::vector<rtx_ctx*> peers;
  std::vector<std::unique_ptr<Worker>> workers;
  DeviceBuffer gather;
  int gather_mode = RTX_GATHER_PEER_COPY;
  std::vector<void*> comms;   // RTX_GATHER_RCCL: one communicator per device, this context's first
};

// Words of LDS per lane the traversal stack of a launch needs (0: the scene is tested linearly): the double-precision kernels
// walk the reference's binary tree (kStackDepth), the single-precision ones the eight-wide tree (one two-word entry per level, walk_stack_slots).
static size_t stack_words(const rtx_ctx* c, bool f64) {
  if (c->scene.n_prims <= kLinearMaxPrims) return 0;
  return f64 ? (size_t)kStackDepth : (size_t)walk_stack_slots(c->scene.wide.levels);
}

// PathPhoton: the caustic photon pass (ref: engine/renderer.cpp:344-372, engine/photon_integrator.hpp:93-137) — plan the
// photons per emissive sphere on the host, trace them on the GPU, build the kd-tree PhotonMap::build makes
// (ref: engine/photon_map.hpp:104-108, :222-245) and the search's boxes there too (photon_build.hip) and hand them to the
// render through P.  The host reads back two numbers: the photons stored and the pass's query count.
namespace {
// `reused` comes back true when the context already held the map of these inputs (nothing was traced or built).
template <class R> int photon_pass(rtx_ctx* c, const rtx_settings* s, DRenderParams& P, const DScene<R>& sc, size_t lds_bytes, hipStream_t stream, bool& reused) {
  reused = false;
  const double kPi = 3.14159265358979323846;
  const std::vector<rtx_sphere>& sph = c->scene.h_spheres;
  const std::vector<rtx_material>& mats = c->scene.h_materials;
  P.ph_nodes = nullptr; P.ph_data = nullptr; P.ph_boxes = nullptr; P.n_photons = 0; P.ph_bucket = kPhotonBucket; P.ph_cand = nullptr; P.ph_cand_cap = 0;
  if (const char* e = std::getenv("RTX_PHOTON_BUCKET")) { const int v = std::atoi(e); if (v >= 1) P.ph_bucket = v; }   // tests: small maps walk the tree too
  P.gather_k = s->photon_gather_count; P.gather_radius = (double)s->caustic_gather_radius;
  std::vector<size_t> lights;
  double total_power = 0.0;
  for (size_t i = 0; i < sph.size(); ++i) {
    const rtx_material& m = mats[sph[i].material_id];
    if (m.emission[0] > 0 || m.emission[1] > 0 || m.emission[2] > 0) { lights.push_back(i); total_power += (m.emission[0] + m.emission[1] + m.emission[2]) / 3.0; }
  }
  std::vector<int32_t> targets;
  for (size_t i = 0; i < sph.size(); ++i) {
    const rtx_material& m = mats[sph[i].material_id];
    if (m.type == RTX_MAT_DIELECTRIC || m.type == RTX_MAT_METAL || m.fuzz < 0.1) targets.push_back((int32_t)i);   // ref: photon_integrator.hpp:110-119
  }
  if (lights.empty() || targets.empty()) return RTX_OK;
  std::vector<DPhotonLight> plan;
  unsigned long long total = 0;
  for (size_t li : lights) {
    const rtx_material& m = mats[sph[li].material_id];
    const double fraction = ((m.emission[0] + m.emission[1] + m.emission[2]) / 3.0) / total_power;
    const size_t n = (size_t)((double)s->caustic_photon_count * fraction);
    if (n == 0) continue;
    const double area = 4.0 * kPi * sph[li].radius * sph[li].radius;
    DPhotonLight L;
    L.sphere = (int32_t)li; L.first = (uint32_t)total; L.count = (uint32_t)n; L.pad = 0;
    for (int a = 0; a < 3; ++a) L.power[a] = m.emission[a] * (area / (double)n);
    plan.push_back(L);
    total += n;
  }
  if (total == 0) return RTX_OK;
  const int is_f64 = sizeof(R) == sizeof(double) ? 1 : 0;
  rtx_ctx::PhotonMapKey& key = c->ph_map;
  if (key.valid && key.scene_generation == c->scene_generation && key.photon_count == (unsigned long long)s->caustic_photon_count && key.seed == P.seed && key.rng == P.rng &&
      key.f64 == is_f64 && key.bucket == P.ph_bucket) {
    reused = true;
    if (key.n_photons > 0) { P.ph_nodes = c->ph_nodes.ptr; P.ph_data = c->ph_data.ptr; P.ph_boxes = c->ph_boxes.ptr; P.n_photons = key.n_photons; }
    return RTX_OK;
  }
  key.valid = false;
  if (total > kPhotonMaxCount) return fail(RTX_ERR_INVALID_ARGUMENT, "caustic_photon_count %llu is too large (at most %llu)", (unsigned long long)s->caustic_photon_count, (unsigned long long)kPhotonMaxCount);
  const uint32_t n = (uint32_t)total;
  if (!c->ph_lights.upload(plan.data(), plan.size() * sizeof(DPhotonLight)) || !c->ph_targets.upload(targets.data(), targets.size() * sizeof(int32_t)) ||
      !c->ph_out.ensure((size_t)n * 10 * sizeof(R)))
    return fail(RTX_ERR_OUT_OF_MEMORY, "photon pass buffers (%u photons)", n);
  launch_photon_trace(sc, P, (const DPhotonLight*)c->ph_lights.ptr, (int)plan.size(), (const int32_t*)c->ph_targets.ptr, (int)targets.size(), n, (R*)c->ph_out.ptr,
                      (DCounters*)c->counters.ptr, lds_bytes, stream);
  HIP_TRY(hipGetLastError());
  DCounters pass_counters;
  HIP_TRY(hipMemcpyAsync(&pass_counters, c->counters.ptr, sizeof pass_counters, hipMemcpyDeviceToHost, stream));
  auto remember = [&](int32_t stored) {
    key.valid = true; key.scene_generation = c->scene_generation; key.photon_count = (unsigned long long)s->caustic_photon_count; key.seed = P.seed; key.rng = P.rng;
    key.f64 = is_f64; key.bucket = P.ph_bucket; key.n_photons = stored; key.closest = pass_counters.closest;
  };
  int32_t stored = 0;
  std::string berr;
  const int brc = build_photon_map((const R*)c->ph_out.ptr, n, c->ph_nodes, c->ph_data, c->ph_boxes, stored, stream, berr);   // (waits for the stream: the counters have arrived)
  if (brc != RTX_OK) return fail(brc, "%s (%u photons)", berr.c_str(), n);
  HIP_TRY(hipStreamSynchronize(stream));
  if (stored == 0) { remember(0); return RTX_OK; }
  P.ph_nodes = c->ph_nodes.ptr; P.ph_data = c->ph_data.ptr; P.ph_boxes = c->ph_boxes.ptr; P.n_photons = stored;
  remember(P.n_photons);
  return RTX_OK;
}
}  // namespace


extern "C" {

const char* rtx_last_error(void) { return g_error.c_str(); }

int rtx_shard_rows(int height, int shard_index, int shard_count) {
  if (shard_count <= 0 || shard_index < 0 || shard_index >= shard_count || height <= 0) return 0;
  return (height - shard_index + shard_count - 1) / shard_count;
}

int rtx_selftest(int device_id) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fail(RTX_ERR_NO_DEVICE, "no HIP device visible");
  if (device_id < 0 || device_id >= n_dev) return fail(RTX_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device_id, n_dev);
  HIP_TRY(hipSetDevice(device_id));
  const int n = 1024;
  float* d = nullptr;
  HIP_TRY(hipMalloc(&d, n * sizeof(float)));
  launch_selftest(d, n, nullptr);
  std::vector<float> h(n);
  hipError_t e = hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RTX_ERR_HIP, "selftest copy: %s", hipGetErrorString(e));
  for (int i : {0, 1, 1023})
    if (std::fabs(h[i] - (float)i * 0.001f) > 1e-6f) return fail(RTX_ERR_HIP, "selftest: out[%d] = %g", i, h[i]);
  return RTX_OK;
}

static int create_impl(rtx_ctx** out_ctx, int device_id) {
  if (!out_ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fail(RTX_ERR_NO_DEVICE, "no HIP device visible");
  if (device_id < 0 || device_id >= n_dev) return fail(RTX_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device_id, n_dev);
  HIP_TRY(hipSetDevice(device_id));
  unsigned char* scratch_dev = scratch_pointer(device_id);   // once per device, with that device current
  if (!scratch_dev) return RTX_ERR_HIP;
  rtx_ctx* c = new rtx_ctx();
  c->device = device_id;
  c->scratch_dev = scratch_dev;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    c->n_cus = prop.multiProcessorCount;
    if (prop.sharedMemPerBlock > 0) c->lds_per_block = prop.sharedMemPerBlock;
  }
  if (hipStreamCreate(&c->stream) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->evm) != hipSuccess ||
      hipHostMalloc((void**)&c->h_counters, sizeof(DCounters), hipHostMallocDefault) != hipSuccess) {
    (void)rtx_destroy(c);
    return fail(RTX_ERR_HIP, "stream / event / pinned-memory creation failed");
  }
  memset(c->h_counters, 0, sizeof(DCounters));
  *out_ctx = c;
  return RTX_OK;
}
int rtx_create(rtx_ctx** out_ctx, int device_id) { return guarded([&] { return create_impl(out_ctx, device_id); }); }

int rtx_create_multi(rtx_ctx** out_ctx, const int* device_ids, int n_devices) {
  if (!out_ctx) return fail(RTX_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > kMaxDevices) return fail(RTX_ERR_INVALID_ARGUMENT, "device list: %d entries (1 to %d)", n_devices, kMaxDevices);
  return guarded([&] {
    rtx_ctx* first = nullptr;
    int rc = create_impl(&first, device_ids[0]);
    if (rc != RTX_OK) return rc;
    first->multi = true;
    for (int k = 1; k < n_devices; ++k) {
      rtx_ctx* p = nullptr;
      rc = create_impl(&p, device_ids[k]);
      if (rc != RTX_OK) { const std::string msg = g_error; (void)rtx_destroy(first); g_error = msg; return rc; }
      first->peers.push_back(p);
      first->workers.emplace_back(new Worker());
      // direct access lets the gather copy run device to device over the links between the GPUs; without it the runtime stages
      // the same copy through the host, so a refusal is not an error
      if (p->device != first->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, first->device, p->device) == hipSuccess && can) {
          (void)hipSetDevice(first->device);
          const hipError_t e = hipDeviceEnablePeerAccess(p->device, 0);
          if (e != hipSuccess) (void)hipGetLastError();   // already enabled, or refused: clear the sticky code
        }
      }
    }
    (void)hipSetDevice(first->device);
    *out_ctx = first;
    return (int)RTX_OK;
  });
}

int rtx_multi_shard(int shard_index, int shard_count, int device_rank, int n_devices, int* out_index, int* out_count) {
  if (shard_count < 1) shard_count = 1;
  if (!out_index || !out_count || shard_index < 0 || shard_index >= shard_count || n_devices < 1 || device_rank < 0 || device_rank >= n_devices)
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_multi_shard(%d, %d, %d, %d)", shard_index, shard_count, device_rank, n_devices);
  if ((long long)shard_count * n_devices > 0x7fffffffLL) return fail(RTX_ERR_INVALID_ARGUMENT, "shard_count * n_devices overflows");
  // row y of the caller's shard: y = shard_index + shard_count * j; device k takes the j with j % n_devices == k
  *out_index = shard_index + shard_count * device_rank;
  *out_count = shard_count * n_devices;
  return RTX_OK;
}

int rtx_destroy(rtx_ctx* c) {
  if (!c) return RTX_OK;
  c->workers.clear();   // joins the peers' host threads (none has a job: every render call waits for its own)
  if (!c->comms.empty()) {
    if (const Rccl* r = load_rccl(nullptr))
      for (void* comm : c->comms) if (comm) (void)r->CommDestroy(comm);
    c->comms.clear();
  }
  for (rtx_ctx* p : c->peers) (void)rtx_destroy(p);
  c->peers.clear();
  (void)hipSetDevice(c->device);
  if (c->pending.active && c->pending.stream) (void)hipStreamSynchronize(c->pending.stream);
  c->pending.active = false;
  c->gather.release();
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  c->scene.release();
  c->tie_list.release(); c->repair_counters.release();
  c->partial.release(); c->film.release(); c->counters.release(); c->rays.release(); c->hits.release(); c->post_in.release(); c->post_out.release(); c->bdpt_store.release();
  c->ph_lights.release(); c->ph_targets.release(); c->ph_out.release(); c->ph_nodes.release(); c->ph_data.release(); c->ph_boxes.release(); c->ph_cand.release();
  c->wf.release();
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->evm) (void)hipEventDestroy(c->evm);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  delete c;
  return RTX_OK;
}

static void settle(rtx_ctx* c);
// The reference-order tree of a large scene is built in the background of rtx_upload_scene: whoever needs it on the device - the
// double-precision kernels, the validation entry points, a frame with an exact tie in t - comes through here.
static int need_tree(rtx_ctx* c) {
  if (!c->scene.tree_pending()) return RTX_OK;
  std::string err;
  HIP_TRY(hipSetDevice(c->device));
  const int rc = c->scene.adopt_tree(true, err);
  if (rc != RTX_OK) { c->has_scene = false; return fail(rc, "reference-order tree: %s", err.c_str()); }
  return RTX_OK;
}
static int upload_impl(rtx_ctx* c, const rtx_scene_desc* d) {
  if (!c || !d) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  settle(c);
  for (rtx_ctx* p : c->peers) settle(p);
  c->has_scene = false;
  for (rtx_ctx* p : c->peers) p->has_scene = false;
  ++c->scene_generation;
  for (rtx_ctx* p : c->peers) ++p->scene_generation;
  std::string err;
  int rc = validate_scene(*d, err);
  if (rc != RTX_OK) return fail(rc, "upload_scene: %s", err.c_str());
  std::vector<rtx_ctx*> all{c};
  all.insert(all.end(), c->peers.begin(), c->peers.end());
  std::vector<int> rcs(all.size(), RTX_OK);
  std::vector<std::string> errs(all.size());
  auto guarded_step = [&](size_t k, auto&& step) {
    if (rcs[k] != RTX_OK) return;
    try {
      if (hipSetDevice(all[k]->device) != hipSuccess) { rcs[k] = RTX_ERR_HIP; errs[k] = "hipSetDevice failed"; return; }
      rcs[k] = step();
    } catch (const std::bad_alloc&) { rcs[k] = RTX_ERR_OUT_OF_MEMORY; errs[k] = "host allocation failed";
    } catch (const std::exception& e) { rcs[k] = RTX_ERR_HIP; errs[k] = e.what(); }
  };
  // every device takes its copy over its own host link and builds its eight-wide tree itself (the build is deterministic,
  // so the trees are identical): one host thread per device
  auto per_device = [&](auto&& body) {
    std::vector<std::thread> th;
    for (size_t k = 1; k < all.size(); ++k) th.emplace_back(body, k);
    body((size_t)0);
    for (std::thread& t : th) t.join();
  };
  auto finish = [&]() -> int {
    (void)hipSetDevice(c->device);
    for (size_t k = 0; k < all.size(); ++k)
      if (rcs[k] != RTX_OK) return fail(rcs[k], "upload_scene (device %d): %s", all[k]->device, errs[k].c_str());
    for (rtx_ctx* x : all) x->has_scene = true;
    return RTX_OK;
  };
  // A large scene without flat primitive boxes: the reference-order tree is built in the background (RefTreeJob) and this call
  // returns as soon as the tables and the eight-wide trees are on the devices.
  if (d->n_spheres + d->n_triangles + d->n_boxes >= kLazyTreeMinPrims && getenv("RTX_EAGER_TREE") == nullptr) {
    const bool timing = getenv("RTX_TIMING") != nullptr;
    BuildInput in;
    make_build_input(*d, in);
    if (timing) fprintf(stderr, "[rtx upload] %-28s %8.1f ms\n", "primitive boxes and keys", in.seconds * 1e3);
    std::vector<int> devices;
    for (rtx_ctx* x : all) devices.push_back(x->device);
    std::shared_ptr<RefTreeJob> job = start_ref_tree_job(in, devices);
    if (job) {
      per_device([&](size_t k) {
        guarded_step(k, [&] { return all[k]->scene.upload_geometry(*d, errs[k]); });
        guarded_step(k, [&] { return all[k]->scene.upload_tree_deferred(*d, job, (int)k, errs[k]); });
      });
      return finish();
    }
    // (a primitive with a flat box: the tree is needed before the first ray, built below as for any other scene)
  }
  // The reference-order tree and what derives from it: built once, whatever the number of devices, on a host thread of its own
  // (it makes no device call) while the scene's tables travel to the devices.
  ScenePrep prep;
  int prep_rc = RTX_OK;
  std::string prep_err;
  std::thread prep_thread([&] {
    try { build_scene_prep(*d, prep);
    } catch (const std::bad_alloc&) { prep_rc = RTX_ERR_OUT_OF_MEMORY; prep_err = "host allocation failed";
    } catch (const std::exception& e) { prep_rc = RTX_ERR_HIP; prep_err = e.what(); }
  });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } join_on_every_path{prep_thread};
  per_device([&](size_t k) { guarded_step(k, [&] { return all[k]->scene.upload_geometry(*d, errs[k]); }); });
  prep_thread.join();
  if (prep_rc != RTX_OK) {
    for (rtx_ctx* x : all) x->scene.release();
    (void)hipSetDevice(c->device);
    return fail(prep_rc, "upload_scene: %s", prep_err.c_str());
  }
  per_device([&](size_t k) { guarded_step(k, [&] { return all[k]->scene.upload_tree(*d, prep, errs[k]); }); });
  return finish();
}
int rtx_upload_scene(rtx_ctx* c, const rtx_scene_desc* d) {
  const int rc = guarded([&] { return upload_impl(c, d); });
  if (rc != RTX_OK && c) { c->has_scene = false; for (rtx_ctx* p : c->peers) p->has_scene = false; }
  return rc;
}

int rtx_set_camera(rtx_ctx* c, const rtx_camera* cam) {
  if (!c || !cam) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  for (int i = 0; i < 3; ++i) {
    c->cam_d.origin[i] = cam->origin[i]; c->cam_d.llc[i] = cam->lower_left_corner[i];
    c->cam_d.horizontal[i] = cam->horizontal[i]; c->cam_d.vertical[i] = cam->vertical[i];
    c->cam_f.origin[i] = (float)cam->origin[i]; c->cam_f.llc[i] = (float)cam->lower_left_corner[i];
    c->cam_f.horizontal[i] = (float)cam->horizontal[i]; c->cam_f.vertical[i] = (float)cam->vertical[i];
  }
  c->has_camera = true;
  for (rtx_ctx* p : c->peers) { p->cam_d = c->cam_d; p->cam_f = c->cam_f; p->has_camera = true; }
  return RTX_OK;
}

int rtx_get_bvh_info(rtx_ctx* c, rtx_bvh_info* out) {
  if (!c || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  if (!c->has_scene) return fail(RTX_ERR_NOT_READY, "no scene uploaded");
  { const int rc = need_tree(c); if (rc != RTX_OK) return rc; }   // build_seconds is known when the tree is
  out->ref_nodes = c->scene.n_nodes; out->ref_prims = c->scene.n_prims;
  // the records the single-precision integrator walks (none for scenes small enough to be tested linearly)
  out->wide_nodes = (uint64_t)c->scene.wide.n_nodes;
  out->wide_bytes = (uint64_t)c->scene.wide.n_nodes * sizeof(DNode8);
  out->wide_leaves = (uint64_t)c->scene.wide.n_leaves; out->wide_levels = (uint64_t)c->scene.wide.levels;
  out->wide_build_seconds = c->scene.wide.seconds;
  out->build_seconds = c->scene.build_seconds;
  out->leaf_bytes = c->scene.wide.n_nodes > 0 ? sizeof(DLeaf) : 0;
  return RTX_OK;
}

int rtx_get_reference_bvh(rtx_ctx* c, rtx_bvh_node* nodes, size_t n_nodes, int32_t* prim_types, int32_t* prim_indices, size_t n_prims) {
  return guarded([&]() -> int {
    if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!c->has_scene) return fail(RTX_ERR_NOT_READY, "no scene uploaded");
    { const int rc = need_tree(c); if (rc != RTX_OK) return rc; }
    if (n_nodes != (size_t)c->scene.n_nodes || n_prims != (size_t)c->scene.n_prims)
      return fail(RTX_ERR_INVALID_ARGUMENT, "the tree has %d nodes and %d primitives (rtx_get_bvh_info), the buffers hold %zu and %zu", c->scene.n_nodes, c->scene.n_prims, n_nodes, n_prims);
    if ((n_nodes && !nodes) || (n_prims && (!prim_types || !prim_indices))) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL buffer");
    static_assert(sizeof(rtx_bvh_node) == sizeof(DNodeRef) && offsetof(rtx_bvh_node, left) == offsetof(DNodeRef, left), "layout");
    HIP_TRY(hipSetDevice(c->device));
    if (n_nodes) HIP_TRY(hipMemcpy(nodes, c->scene.dev_ref.nodes, n_nodes * sizeof(DNodeRef), hipMemcpyDeviceToHost));
    std::vector<DPrimRef> prims(n_prims);
    if (n_prims) HIP_TRY(hipMemcpy(prims.data(), c->scene.dev_ref.prims, n_prims * sizeof(DPrimRef), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_prims; ++i) { prim_types[i] = prims[i].type; prim_indices[i] = prims[i].index; }
    return RTX_OK;
  });
}

constexpr uint32_t kTieListCap = 1u << 16;   // tied work items a frame can note for its repair pass; beyond, the whole frame is rendered again
static int render_enqueue(rtx_ctx* c, const rtx_settings* s, float* host_out, void* dev_out, hipStream_t user_stream, bool no_alloc = false, bool bare = false,
                          bool stream_as_given = false, uint32_t repair_n = 0);

// Second half of a render: waits for the kernels render_enqueue put on the stream and turns the counters and event times into
// statistics.
static int render_collect(rtx_ctx* c, rtx_stats* stats) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  rtx_ctx::Pending& pd = c->pending;
  if (!pd.active) return fail(RTX_ERR_NOT_READY, "no render in flight on this context");
  pd.active = false;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(pd.stream));
  rtx_stats st = pd.st;
  const DCounters hc = *c->h_counters;
  // the times of this pass, read before a repair pass records the same events again
  float ms_render = 0.f, ms_main = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms_render, c->ev0, c->ev1));
  HIP_TRY(hipEventElapsedTime(&ms_main, c->ev0, c->evm));
  if (pd.wavefront) { st.seconds_extend = c->wf.timed_seconds(); if (c->wf.overlap) st.flags |= RTX_STAT_WALKS_SIDE_BY_SIDE; }
  if (pd.optimistic && hc.ties > 0) {
    // Ties in t were decided without the reference's leaf order: wait for that order, then render the work items they occurred in
    // again (a few dozen of the 33 M of a 1080p frame at 64 spp); if there were more than the list holds, the whole frame.
    const rtx_ctx::Pending again = pd;
    const int trc = need_tree(c);
    if (trc != RTX_OK) return trc;
    if (hc.ties <= (unsigned long long)kTieListCap) {
      // every item once: the queue pipeline adds the light samples of an item's last path to its chunk sum, which two concurrent
      // renders of one item would do twice
      std::vector<uint32_t> items((size_t)hc.ties);
      HIP_TRY(hipMemcpy(items.data(), c->tie_list.ptr, items.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      std::sort(items.begin(), items.end());
      items.erase(std::unique(items.begin(), items.end()), items.end());
      HIP_TRY(hipMemcpy(c->tie_list.ptr, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      const int rc = render_enqueue(c, &again.settings, again.host_out, again.dev_out, again.user_stream, again.no_alloc, false, again.stream_as_given, (uint32_t)items.size());
      if (rc != RTX_OK) return rc;
      HIP_TRY(hipStreamSynchronize(pd.stream));
      st.flags |= RTX_STAT_TIES_REPAIRED;
    } else {
      const int rc = render_enqueue(c, &again.settings, again.host_out, again.dev_out, again.user_stream, again.no_alloc, false, again.stream_as_given);
      if (rc != RTX_OK) return rc;
      c->pending.wall0 = again.wall0;
      c->pending.st.flags |= RTX_STAT_BEFORE_TREE | RTX_STAT_RENDERED_TWICE;
      return render_collect(c, stats);
    }
  }
  if (pd.optimistic) st.flags |= RTX_STAT_BEFORE_TREE;
  st.samples = hc.samples; st.rays_closest = hc.closest; st.rays_shadow = hc.shadow; st.segments = hc.segments;
  if (pd.photon_map_reused) st.rays_closest += c->ph_map.closest;   // the queries this film's photon map was traced with, made by an earlier call
  st.nodes_visited = hc.nodes_visited; st.prims_tested = hc.prims_tested;
  st.seconds_render = ms_render * 1e-3;
  st.seconds_main_kernel = ms_main * 1e-3;
  st.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - pd.wall0).count();
  st.pipeline_used = (uint32_t)pd.pipeline;
  st.precision_used = pd.f64 ? RTX_PRECISION_F64 : RTX_PRECISION_F32;
  if (stats) *stats = st;
  return RTX_OK;
}
// A context holds one render in flight: whatever touches its buffers next waits for it first (its statistics are dropped).
static void settle(rtx_ctx* c) {
  if (c && c->pending.active) (void)render_collect(c, nullptr);
}

// First half of a render: validates, sizes the buffers and puts every kernel (and the copies of the film to `host_out` and of the
// counters to pinned host memory) on the stream; render_collect finishes it.  `no_alloc`: the render may not allocate
// (rtx_render_device_noalloc): its chunk sums and counters go to the device's scratch array, and whatever would need another
// buffer is refused; with `bare` as well it records nothing but kernels (no event, no copy) and leaves nothing to collect.
// repair_n > 0: the repair pass of the frame just rendered - the first repair_n work items of c->tie_list are rendered again (with the
// tie order that has arrived meanwhile) into the chunk sums the first pass left, and the film is reduced again; nothing is timed,
// armed or counted for the caller (its counters go to c->repair_counters).
static int render_enqueue(rtx_ctx* c, const rtx_settings* s, float* host_out, void* dev_out, hipStream_t user_stream, bool no_alloc, bool bare,
                          bool stream_as_given /* NULL is HIP's default stream, not the context's own */, uint32_t repair_n) {
  if (!c || !s) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  const bool repair = repair_n > 0;
  if (repair) bare = true;   // no events, no arming
  else settle(c);
  if (!c->has_scene) return fail(RTX_ERR_NOT_READY, "render before upload_scene");
  if (!c->has_camera) return fail(RTX_ERR_NOT_READY, "render before set_camera");
  if (s->width <= 0 || s->height <= 0 || s->samples_per_pixel <= 0 || s->max_depth < 0)
    return fail(RTX_ERR_INVALID_ARGUMENT, "bad settings: %dx%d spp=%d depth=%d", s->width, s->height, s->samples_per_pixel, s->max_depth);
  if (s->width > 65535 || s->height > 65535) return fail(RTX_ERR_INVALID_ARGUMENT, "film too large (max 65535 per side)");
  const int sc_n = s->shard_count > 0 ? s->shard_count : 1;
  if (s->shard_index < 0 || s->shard_index >= sc_n) return fail(RTX_ERR_INVALID_ARGUMENT, "shard %d of %d", s->shard_index, sc_n);
  if (s->precision != RTX_PRECISION_F32 && s->precision != RTX_PRECISION_F64) return fail(RTX_ERR_INVALID_ARGUMENT, "precision %d", s->precision);
  if (s->pipeline < RTX_PIPELINE_AUTO || s->pipeline > RTX_PIPELINE_WAVEFRONT) return fail(RTX_ERR_INVALID_ARGUMENT, "pipeline %d", s->pipeline);
  if (s->mode != RTX_MODE_WHITTED && s->mode != RTX_MODE_PATHTRACE && s->mode != RTX_MODE_BDPT && s->mode != RTX_MODE_PATHPHOTON)
    return fail(RTX_ERR_UNSUPPORTED, "render mode %d: only Whitted (0), PathTrace (1), BDPT (2) and PathPhoton (6) are implemented", s->mode);
  if (s->mode == RTX_MODE_PATHPHOTON && s->photon_gather_count < 1) return fail(RTX_ERR_INVALID_ARGUMENT, "photon_gather_count %d", s->photon_gather_count);
  if (s->mode == RTX_MODE_PATHPHOTON && !(s->caustic_gather_radius >= 0.0f)) return fail(RTX_ERR_INVALID_ARGUMENT, "caustic_gather_radius %g", (double)s->caustic_gather_radius);
  if (s->mode == RTX_MODE_BDPT && !(s->clamp_max == s->clamp_max)) return fail(RTX_ERR_INVALID_ARGUMENT, "clamp_max is NaN");
  if (s->rng < RTX_RNG_AUTO || s->rng > RTX_RNG_PCG4D) return fail(RTX_ERR_INVALID_ARGUMENT, "rng %d", s->rng);
  if (!host_out && !dev_out) return fail(RTX_ERR_INVALID_ARGUMENT, "output buffer is NULL");
  HIP_TRY(hipSetDevice(c->device));
  auto wall0 = std::chrono::steady_clock::now();
  hipStream_t stream = (user_stream || stream_as_given) ? user_stream : c->stream;

  DRenderParams P;
  memset(&P, 0, sizeof P);
  P.width = s->width; P.height = s->height; P.spp = s->samples_per_pixel; P.max_depth = s->max_depth;
  P.use_nee = s->use_nee != 0; P.use_mis = s->use_mis != 0; P.use_gradient = s->use_background_gradient != 0;
  P.shard_index = s->shard_index; P.shard_count = sc_n;
  P.rows = rtx_shard_rows(s->height, s->shard_index, sc_n);
  // Fixed chunking of the sample sum (a function of spp and the pipeline only) keeps the film bit-identical
  // for any grid size or shard count.
  int target_chunks = std::min(P.spp, 16);
  P.chunk_spp = (P.spp + target_chunks - 1) / target_chunks;
  P.n_chunks = (P.spp + P.chunk_spp - 1) / P.chunk_spp;
  P.seed = s->seed;
  P.clamp_max = s->clamp_max;
  P.rng = (s->rng == RTX_RNG_PCG4D || (s->rng == RTX_RNG_AUTO && s->precision == RTX_PRECISION_F32)) ? 1 /* RNG_PCG4D */ : 0;
  for (int i = 0; i < 3; ++i) { P.bg[i] = s->background_color[i]; P.bg_top[i] = s->background_top[i]; P.bg_bottom[i] = s->background_bottom[i]; }

  const unsigned long long npix = (unsigned long long)P.rows * P.width;
  rtx_stats st;
  memset(&st, 0, sizeof st);
  auto arm = [&](int pipeline_used, bool is_f64, bool reused) {   // hands the render over to render_collect
    rtx_ctx::Pending& pd = c->pending;
    pd.active = true; pd.stream = stream; pd.photon_map_reused = reused; pd.wavefront = pipeline_used == RTX_PIPELINE_WAVEFRONT; pd.f64 = is_f64;
    pd.pipeline = pipeline_used; pd.wall0 = wall0; pd.st = st;
    pd.optimistic = false; pd.settings = *s; pd.host_out = host_out; pd.dev_out = dev_out; pd.user_stream = user_stream; pd.no_alloc = no_alloc;
    pd.stream_as_given = stream_as_given;
  };
  if (npix == 0) {   // a shard without rows: nothing to launch, empty statistics
    if (bare) return RTX_OK;
    memset(c->h_counters, 0, sizeof(DCounters));
    HIP_TRY(hipEventRecord(c->ev0, stream)); HIP_TRY(hipEventRecord(c->evm, stream)); HIP_TRY(hipEventRecord(c->ev1, stream));
    arm(s->pipeline == RTX_PIPELINE_WAVEFRONT ? RTX_PIPELINE_WAVEFRONT : RTX_PIPELINE_FUSED, s->precision == RTX_PRECISION_F64, false);
    return RTX_OK;
  }
  // Manifold next event estimation runs in the fused kernels: in double precision as the reference computes it, in single
  // precision with single-precision traces around a double-precision manifold walk.
  const bool whitted = s->mode == RTX_MODE_WHITTED;
  const bool bdpt = s->mode == RTX_MODE_BDPT;
  const bool photon = s->mode == RTX_MODE_PATHPHOTON;
  const bool mnee = !whitted && !bdpt && !photon && s->use_mnee != 0 && s->use_nee != 0;
  const bool f64 = s->precision == RTX_PRECISION_F64;
  const bool f64_req = f64;
  // AUTO: scenes of many bounded primitives run the queue-based pipeline (its traversal kernels hold twice the waves of the fused
  // tree-walking kernel and keep their lanes full by refilling); small ones stay fused (no queue traffic).  Triangle soups at
  // 1080p x 16 spp, fused / queues in ms: 30 triangles 10.0 / 27.2, 300: 19.5 / 28.6, 1000: 29.9 / 31.7, 2000: 38.3 / 34.8, 4000: 48.3 /
  // 38.9, 10 000: 68.0 / 47.2, 100 000: 164 / 88, 1 M: 358 / 133 (scripts/pipeline_crossover.py).
  int pipeline = s->pipeline;
  if (pipeline == RTX_PIPELINE_AUTO) pipeline = (!f64_req && !whitted && !bdpt && !photon && !mnee && c->scene.n_prims > kAutoQueuesAbovePrims) ? RTX_PIPELINE_WAVEFRONT : RTX_PIPELINE_FUSED;
  if (pipeline == RTX_PIPELINE_WAVEFRONT) {
    // The queue pipeline cuts the sample sum finer: a path slot renders the samples of its work item one after the other, so the
    // number of turns (and with it the number of launches, each with a start and an end at which the device is not full) follows
    // the samples per item.  10 M triangles, 1080p, 64 spp, chunks of 4 / 2 / 1 samples: 389.9 / 371.5 / 373.9 ms; every 8th row of
    // that frame: 67.7 / 58.2 / 52.7 ms.  The pool holds the items of a 1080p frame at 32 spp at once; beyond, slots take new items
    // as they finish theirs.  (Still a function of spp and the film's size alone: films do not depend on shards or grids.  RTX_WF_CHUNKS: experiments.)
    int target = kWfTargetChunks;
    if (const char* e = std::getenv("RTX_WF_CHUNKS")) target = std::max(1, std::atoi(e));
    // (a launch indexes fewer than 2^31 work items: very large films fall back towards the fused kernels' 16 chunks - by the size
    // of the whole film, which all shards of a frame share)
    while (target > 16 && (unsigned long long)P.width * (unsigned long long)P.height * (unsigned long long)target >= 0x7fffffffull) target /= 2;
    target = std::min(P.spp, target);
    P.chunk_spp = (P.spp + target - 1) / target;
    P.n_chunks = (P.spp + P.chunk_spp - 1) / P.chunk_spp;
  }
  if (pipeline == RTX_PIPELINE_WAVEFRONT && photon) return fail(RTX_ERR_UNSUPPORTED, "the PathPhoton mode is implemented by the fused pipeline only");
  if (pipeline == RTX_PIPELINE_WAVEFRONT && bdpt) return fail(RTX_ERR_UNSUPPORTED, "the BDPT mode is implemented by the fused pipeline only");
  if (pipeline == RTX_PIPELINE_WAVEFRONT && whitted) return fail(RTX_ERR_UNSUPPORTED, "the Whitted mode is implemented by the fused pipeline only");
  if (pipeline == RTX_PIPELINE_WAVEFRONT && mnee) return fail(RTX_ERR_UNSUPPORTED, "use_mnee is implemented by the fused pipeline only");
  if (pipeline == RTX_PIPELINE_WAVEFRONT && f64) return fail(RTX_ERR_UNSUPPORTED, "the wavefront pipeline is single precision only");

  // A scene whose reference-order tree is still being built: the plain single-precision path tracer renders without it (ties are
  // counted, render_collect renders again if there was one); whatever else needs it waits here.
  bool optimistic = false;
  if (c->scene.tree_pending()) {
    std::string terr;
    int trc = c->scene.adopt_tree(false, terr);   // takes it over if it has arrived meanwhile
    if (trc == RTX_OK && c->scene.tree_pending()) {
      if (f64 || whitted || bdpt || photon || mnee || bare || no_alloc) trc = c->scene.adopt_tree(true, terr);   // (no_alloc: a frame ahead of the tree needs a tie list and a repair pass)
      else optimistic = true;
    }
    if (trc != RTX_OK) { c->has_scene = false; return fail(trc, "reference-order tree: %s", terr.c_str()); }
  }
  if (optimistic) {
    if (!c->tie_list.ensure((size_t)kTieListCap * sizeof(uint32_t))) return fail(RTX_ERR_OUT_OF_MEMORY, "tie list");
    P.tie_list = (uint32_t*)c->tie_list.ptr; P.tie_cap = kTieListCap;
  }
  if (repair) { P.item_list = (const uint32_t*)c->tie_list.ptr; P.n_list = repair_n; }

  float* d_film = (float*)dev_out;
  void* partial_ptr = nullptr;
  DCounters* counters_ptr = nullptr;
  if (no_alloc) {
    if (f64 || mnee || bdpt || photon) return fail(RTX_ERR_UNSUPPORTED, "the no-allocation render runs single-precision PathTrace and Whitted only");
    if (pipeline != RTX_PIPELINE_FUSED) return fail(RTX_ERR_UNSUPPORTED, "the no-allocation render runs the fused pipeline only (this scene and these settings select the queue-based one)");
    static_assert(sizeof(DCounters) <= kScratchHeader, "counters must fit the scratch header");
    counters_ptr = (DCounters*)c->scratch_dev;
    partial_ptr = c->scratch_dev + kScratchHeader;
    if (repair) return fail(RTX_ERR_HIP, "internal: repair pass of a no-allocation render");   // (such renders wait for the tree)
    launch_zero_words((uint32_t*)counters_ptr, (unsigned)(sizeof(DCounters) / 4), stream);   // a fill kernel, not a memset node: callers may capture this call
  } else {
    if (!d_film) {
      if (!c->film.ensure(npix * 3 * sizeof(float))) return fail(RTX_ERR_OUT_OF_MEMORY, "film buffer");
      d_film = (float*)c->film.ptr;
    }
    DeviceBuffer& cb = repair ? c->repair_counters : c->counters;
    if (!cb.ensure(sizeof(DCounters))) return fail(RTX_ERR_OUT_OF_MEMORY, "counters");
    counters_ptr = (DCounters*)cb.ptr;
    HIP_TRY(hipMemsetAsync(counters_ptr, 0, sizeof(DCounters), stream));
  }

  const bool needs_stack = c->scene.n_prims > kLinearMaxPrims;
  bool photon_map_reused = false;
  if (!photon && !bare) HIP_TRY(hipEventRecord(c->ev0, stream));   // PathPhoton records it behind its photon pass
  if (pipeline == RTX_PIPELINE_FUSED) {
    const size_t elem = f64 ? sizeof(double) : sizeof(float);
    const unsigned long long all_items = npix * (unsigned long long)P.n_chunks;
    if (all_items >= 0x7fffffffull) return fail(RTX_ERR_INVALID_ARGUMENT, "film too large for one launch (%llu work items)", all_items);
    const unsigned long long items = repair ? (unsigned long long)repair_n : all_items;   // work items this launch renders
    if (no_alloc) {
      if ((unsigned long long)all_items * 3ull * elem > kScratchBytes - kScratchHeader)
        return fail(RTX_ERR_INVALID_ARGUMENT, "the no-allocation render holds %llu bytes of chunk sums: %d x %d pixels in %d sample chunks need %llu (at most %llu pixel-chunks)",
                    (unsigned long long)(kScratchBytes - kScratchHeader), P.width, P.rows, P.n_chunks, (unsigned long long)all_items * 3ull * elem,
                    (unsigned long long)((kScratchBytes - kScratchHeader) / (3 * elem)));
    } else {
      if (!c->partial.ensure((size_t)all_items * 3 * elem)) return fail(RTX_ERR_OUT_OF_MEMORY, "chunk-sum buffer (%llu bytes)", (unsigned long long)(all_items * 3 * elem));
      partial_ptr = c->partial.ptr;
    }
    FusedLaunch L;
    L.mode = whitted ? MODE_TREE_UV : c->scene.fused_mode(f64);
    if (photon && L.mode != MODE_FLAT) L.mode = MODE_TREE_UV;   // PathPhoton: the flat instance (single precision, small untextured scenes) or the general one
    if (bdpt && L.mode == MODE_TREE) L.mode = MODE_TREE_UV;
    if (mnee && !f64 && L.mode == MODE_TREE) L.mode = MODE_TREE_UV;   // the single-precision MNEE instances: flat or general
    L.integ = whitted ? INTEG_WHITTED : bdpt ? INTEG_BDPT : photon ? INTEG_PATHPHOTON : mnee ? INTEG_PATH_MNEE : INTEG_PATH;
    if (const char* e = std::getenv("RTX_MNEE_ON_THE_SPOT")) L.mnee_batched = std::atoi(e) == 0;
    if (const char* e = std::getenv("RTX_FUSED_REGROUP")) { const int v = std::atoi(e); if (v > 0 && v <= 64) L.regroup = v; }   // experiment, see render_fused_regroup_kernel
    const size_t mat_bytes = (size_t)c->scene.dev_f.n_mats * (f64 ? sizeof(DMaterial<double>) : sizeof(DMaterial<float>));
    L.lds_mats = mat_bytes <= kMaxLdsMaterialBytes;
    if (!L.lds_mats) L.mode = MODE_TREE_UV;
    L.stack_first_int = L.lds_mats ? (int)((mat_bytes / 4 + 3) & ~(size_t)3) : 0;
    const bool stack = L.mode != MODE_FLAT && needs_stack;
    L.lds_bytes = ((size_t)L.stack_first_int + (stack ? stack_words(c, f64) * kBlock : 0)) * sizeof(int);
    if (L.lds_bytes > c->lds_per_block && L.lds_mats && stack) {   // the traversal stacks come first: the material table stays in device memory
      L.lds_mats = false; L.mode = MODE_TREE_UV; L.stack_first_int = 0;
      L.lds_bytes = stack_words(c, f64) * kBlock * sizeof(int);
    }
    if (L.lds_bytes > c->lds_per_block)
      return fail(RTX_ERR_UNSUPPORTED, "this scene's traversal stacks need %zu bytes of LDS per workgroup, the device gives %zu", L.lds_bytes, c->lds_per_block);
    // Persistent grid: enough workgroups to fill every CU at the kernel's occupancy, no more.
    // Grid: at least what fills every CU at the kernel's occupancy; beyond that, so many workgroups that a lane gets about
    // kFusedSamplesPerLane samples (and at least kFusedMinItemsPerLane work items).  A lane's items are fixed by its index
    // (item = lane + k * lanes), so a workgroup keeps returning to the same few strips of the image, and strips through glass and
    // mirrors cost several times what bare walls cost: with exactly the resident number of workgroups the slow ones ran on alone at
    // the end of a launch.  Many short workgroups let the dispatcher even that out; too short ones (a dozen samples per lane) lose
    // again to the spread inside a wave and to the per-workgroup set-up.  Measured, ms per render, resident grid -> best (samples
    // per lane at the best): 1080p Cornell box 1024 spp 202.5 -> 168.4 (192; 128: 169.0, 384: 169.7); cornell_box_quad +
    // caustics_test 800 x 800 x 128 16.8 -> 14.5 (24; 48: 14.8); Cornell box 64 spp 15.5 -> 13.3 (24; 96: 14.2), 16 spp 4.94 -> 4.49
    // (12; 24: 4.65, 6: 5.44); double precision 32 spp 34.0 -> 32.0 (48); manifold NEE 16 spp 338.7 -> 317.2 (24); environment_demo
    // 4K 256 spp 220.5 -> 186.6.  Whitted, BDPT and PathPhoton: below.  The film does not depend on the grid.
    const int blocks_per_cu = stack ? 3 : 6;
    const unsigned long long one_item_each = (items + kBlock - 1) / kBlock, resident = (unsigned long long)c->n_cus * blocks_per_cu;
    unsigned long long per_lane = 0ull;   // 0 = the resident grid
    if (!bdpt && !photon && !whitted) {
      const unsigned long long enough = (kFusedSamplesPerLane + (unsigned long long)P.chunk_spp - 1ull) / (unsigned long long)P.chunk_spp;   // items that hold kFusedSamplesPerLane samples
      per_lane = std::max<unsigned long long>(kFusedMinItemsPerLane, enough);
      // A small launch (a shard of a frame) would have few rounds of such workgroups, and its last, partly filled round would be a
      // large share of it: fewer items per lane then, down to `enough`, as long as the grid has fewer than kFusedRounds rounds.
      // 1080p Cornell box at 1024 spp, every 8th row: 24.3 -> 22.7 ms; every 4th: 44.7 -> 43.4; every 2nd and the whole frame: as before.
      per_lane = std::min(per_lane, std::max(enough, items / (resident * (unsigned long long)kBlock * kFusedRounds)));
      if (const char* e = std::getenv("RTX_ITEMS_PER_LANE")) { const long v = std::atol(e); if (v >= 0 && v <= 4096) per_lane = (unsigned long long)v; }   // experiments
    } else {
      // PathPhoton: a sample costs many times a plain one (a density estimate per Lambertian vertex), a few per lane are enough:
      // 1080p Cornell box, 16 spp, resident grid 129.6 ms, 3 items per lane 102.7, 6: 102.8, 12: 104.2.  BDPT loses (its per-lane
      // light-vertex store grows with the grid: 194.6 ms -> 199.4 at 3, 224.0 at 12) and keeps the resident grid.
      if (photon) per_lane = 4ull;
      // Whitted: with the instance at six waves per SIMD the resident grid (sized for three to six workgroups per CU) no longer
      // fills the device, and short workgroups even out the glass regions here too.  Items per lane, 64 spp: demo.json 2560 x 1440
      // resident grid 15.8 ms, 32: 17.9, 16: 13.1, 8: 12.4, 4: 13.6; earth_demo 4K 23.0 / 24: 19.9 / 16: 16.9 / 8: 18.3; kitchen_sink 4K
      // 85.3 / 16: 80.1 / 8: 82.8.
      if (whitted) per_lane = 16ull;
      if (const char* e = std::getenv("RTX_ITEMS_PER_LANE_MODES")) { const long v = std::atol(e); if (v >= 0 && v <= 4096) per_lane = (unsigned long long)v; }   // experiments
    }
    unsigned long long grid = std::min(one_item_each, resident);
    if (per_lane) grid = std::max(grid, (items + kBlock * per_lane - 1) / (kBlock * per_lane));
    L.grid = (int)grid;
    int lrc;
    const DScene<float>& scf = c->scene.dev_f;
    if (photon) {   // the photon pass comes first (ref: engine/renderer.cpp:344-372); its rays are counted as closest-hit queries
      // the photon kernel is the tree-walking instance whatever the render kernel is (a flat scene of 13 to 16 spheres walks
      // its tree there), so its traversal stacks are sized from the scene alone
      const size_t walk_lds = stack_words(c, f64) * kBlock * sizeof(int);
      const auto ph0 = std::chrono::steady_clock::now();
      const int prc = f64 ? photon_pass(c, s, P, c->scene.dev_d, walk_lds, stream, photon_map_reused) : photon_pass(c, s, P, c->scene.dev_f, walk_lds, stream, photon_map_reused);
      if (prc != RTX_OK) return prc;
      st.seconds_photon_pass = std::chrono::duration<double>(std::chrono::steady_clock::now() - ph0).count();
      // An estimate keeps min(photon_gather_count, photons stored) photons.  Up to kPhotonMaxGather the candidate list of a wave
      // sits in LDS; beyond, every wave of the grid gets two lists in device memory, and the grid shrinks if they would pass 2 GiB.
      const long long keep = std::min<long long>(s->photon_gather_count, P.n_photons);
      if (keep > kPhotonMaxGather) {
        const unsigned long long cap = ((unsigned long long)keep + 63ull) / 64ull * 64ull + 128ull;
        const unsigned long long per_block = (unsigned long long)(kBlock / 64) * cap * 2ull * (elem + sizeof(int));
        const unsigned long long max_blocks = (2ull << 30) / per_block;
        if (max_blocks < 1) return fail(RTX_ERR_OUT_OF_MEMORY, "photon_gather_count %d: the candidate lists of one workgroup need %llu bytes", s->photon_gather_count, per_block);
        if ((unsigned long long)L.grid > max_blocks) L.grid = (int)max_blocks;
        if (!c->ph_cand.ensure((size_t)(per_block * (unsigned long long)L.grid))) return fail(RTX_ERR_OUT_OF_MEMORY, "photon candidate lists (%llu bytes)", per_block * (unsigned long long)L.grid);
        P.ph_cand = c->ph_cand.ptr; P.ph_cand_cap = (int32_t)cap;
      }
      HIP_TRY(hipEventRecord(c->ev0, stream));
    }
    if (bdpt) {
      // every lane keeps the connectible vertices of its light subpath (at most max_depth + 1) in global memory;
      // the grid shrinks when that would pass 2 GiB
      const unsigned long long cap = (unsigned long long)P.max_depth + 1ull;
      const unsigned long long per_lane = (unsigned long long)kBdptStoreFields * cap * elem;
      const unsigned long long max_lanes = std::max<unsigned long long>(kBlock, (2ull << 30) / per_lane / kBlock * kBlock);
      if ((unsigned long long)L.grid * kBlock > max_lanes) L.grid = (int)(max_lanes / kBlock);
      if (!c->bdpt_store.ensure((size_t)((unsigned long long)L.grid * kBlock * per_lane))) return fail(RTX_ERR_OUT_OF_MEMORY, "BDPT light-vertex store (%llu bytes)", (unsigned long long)L.grid * kBlock * per_lane);
      if (f64) lrc = launch_render_bdpt(c->scene.dev_d, c->cam_d, P, (double*)partial_ptr, counters_ptr, L, (double*)c->bdpt_store.ptr, cap, stream);
      else lrc = launch_render_bdpt(scf, c->cam_f, P, (float*)partial_ptr, counters_ptr, L, (float*)c->bdpt_store.ptr, cap, stream);
      if (!bare) HIP_TRY(hipEventRecord(c->evm, stream));
      if (f64) launch_film_reduce((const double*)partial_ptr, d_film, npix, P.n_chunks, P.spp, stream);
      else launch_film_reduce((const float*)partial_ptr, d_film, npix, P.n_chunks, P.spp, stream);
    } else if (f64) {
      lrc = launch_render_fused(c->scene.dev_d, c->cam_d, P, (double*)partial_ptr, counters_ptr, L, stream);
      if (!bare) HIP_TRY(hipEventRecord(c->evm, stream));
      launch_film_reduce((const double*)partial_ptr, d_film, npix, P.n_chunks, P.spp, stream);
    } else {
      lrc = launch_render_fused(scf, c->cam_f, P, (float*)partial_ptr, counters_ptr, L, stream);
      if (!bare) HIP_TRY(hipEventRecord(c->evm, stream));
      launch_film_reduce((const float*)partial_ptr, d_film, npix, P.n_chunks, P.spp, stream);
    }
    if (lrc != 0) return fail(RTX_ERR_UNSUPPORTED, "no kernel instance for mode %d", L.mode);
    st.kernel_launches = photon ? 3 : 2;
  } else {
    std::string err;
    if (stack_words(c, false) * kBlock * sizeof(int) > c->lds_per_block)
      return fail(RTX_ERR_UNSUPPORTED, "this scene's traversal stacks need %zu bytes of LDS per workgroup, the device gives %zu", stack_words(c, false) * kBlock * sizeof(int), c->lds_per_block);
    int rc = wavefront_render(c->wf, c->scene.dev_f, c->cam_f, P, d_film, counters_ptr, c->n_cus, c->scene.fused_mode(false), (int)stack_words(c, false), stream, st.kernel_launches, err);
    if (rc != RTX_OK) return fail(rc, "wavefront: %s", err.c_str());
    HIP_TRY(hipEventRecord(c->evm, stream));
  }
  HIP_TRY(hipGetLastError());
  if (repair) {   // the caller (render_collect) waits for the stream
    if (host_out) HIP_TRY(hipMemcpyAsync(host_out, d_film, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
    return RTX_OK;
  }
  if (bare) return RTX_OK;   // kernels only: the caller synchronises (or replays) its stream
  HIP_TRY(hipEventRecord(c->ev1, stream));
  HIP_TRY(hipMemcpyAsync(c->h_counters, counters_ptr, sizeof(DCounters), hipMemcpyDeviceToHost, stream));
  if (host_out) HIP_TRY(hipMemcpyAsync(host_out, d_film, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, stream));
  arm(pipeline, f64, photon_map_reused);
  c->pending.optimistic = optimistic;
  return RTX_OK;
}

static int render_impl(rtx_ctx* c, const rtx_settings* s, float* host_out, void* dev_out, hipStream_t user_stream, rtx_stats* stats, bool no_alloc = false) {
  const bool bare = no_alloc && !stats;
  const int rc = render_enqueue(c, s, host_out, dev_out, user_stream, no_alloc, bare);
  if (rc != RTX_OK || bare) return rc;
  return render_collect(c, stats);
}

// Render of a multi-device context.  The caller's shard (rows y with y % shard_count == shard_index, packed bottom-up: packed
// row j is y = shard_index + shard_count * j) is dealt to the devices row by row: device k renders the packed rows j with
// j % n == k, which is shard rtx_multi_shard(k) of the film, with the kernels and sample streams of a single-device render, so the
// gathered film equals the single-device film bit for bit.
// Every other device's share runs on that device's own host thread (kept since rtx_create_multi), the first device's on the
// caller's.  A share is: enqueue the kernels on the device's stream; say so; then wait for them and read the statistics.  As soon
// as a device has said so, its rows are ordered behind its render and sent on their way: to host memory by one strided copy per
// device (issued by its own thread on its own stream: every device over its own host link), to device memory by a copy to a
// staging buffer on the first device (a peer copy behind an event of the rendering stream, or ncclSend / ncclRecv) followed by a
// strided copy there - all on the gather stream, which is waited for once, at the end.
static int render_multi(rtx_ctx* c, const rtx_settings* s, float* host_out, void* dev_out, hipStream_t user_stream, rtx_stats* stats) {
  if (!c || !s) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  if (s->width <= 0 || s->height <= 0) return fail(RTX_ERR_INVALID_ARGUMENT, "bad settings: %dx%d", s->width, s->height);
  const auto wall0 = std::chrono::steady_clock::now();
  std::vector<rtx_ctx*> all{c};
  all.insert(all.end(), c->peers.begin(), c->peers.end());
  const int n = (int)all.size();
  const size_t row_bytes = (size_t)s->width * 3 * sizeof(float);
  std::vector<rtx_settings> sk(n, *s);
  std::vector<int> rows(n, 0);
  for (int k = 0; k < n; ++k) {
    const int rc = rtx_multi_shard(s->shard_index, s->shard_count, k, n, &sk[k].shard_index, &sk[k].shard_count);
    if (rc != RTX_OK) return rc;
    rows[k] = rtx_shard_rows(s->height, sk[k].shard_index, sk[k].shard_count);
  }
  struct Share {
    int rc = RTX_OK;
    std::string err;
    rtx_stats st;
    std::mutex m;
    std::condition_variable cv;
    bool enqueued = false;   // the share's kernels are on its stream (or it failed before that)
    bool collected = false;  // ... and already waited for (a frame rendered ahead of the reference-order tree, see enqueue_share)
  };
  std::unique_ptr<Share[]> sh(new Share[n]);
  for (int k = 0; k < n; ++k) memset(&sh[k].st, 0, sizeof(rtx_stats));
  auto announce = [&](int k) {
    { std::lock_guard<std::mutex> l(sh[k].m); sh[k].enqueued = true; }
    sh[k].cv.notify_all();
  };
  auto enqueue_share = [&](int k) {   // -> the share's kernels are on all[k]->stream
    rtx_ctx* x = all[k];
    if (rows[k] == 0) return;
    if (hipSetDevice(x->device) != hipSuccess) { sh[k].rc = RTX_ERR_HIP; sh[k].err = "hipSetDevice failed"; return; }
    if (!x->film.ensure((size_t)rows[k] * row_bytes)) { sh[k].rc = RTX_ERR_OUT_OF_MEMORY; sh[k].err = "film buffer"; return; }
    sh[k].rc = render_enqueue(x, &sk[k], nullptr, x->film.ptr, nullptr);
    if (sh[k].rc != RTX_OK) { sh[k].err = g_error; return; }
    if (x->pending.optimistic) {
      // A frame ahead of the reference-order tree: collecting it renders its tied work items again, so nothing - the copy to the
      // host below, the gather - may be ordered behind the kernels that are on the stream now.  Collect first.
      sh[k].rc = render_collect(x, &sh[k].st);
      sh[k].collected = true;
      if (sh[k].rc != RTX_OK) { sh[k].err = g_error; return; }
    }
    if (host_out) {
      const hipError_t e = hipMemcpy2DAsync((char*)host_out + (size_t)k * row_bytes, (size_t)n * row_bytes, x->film.ptr, row_bytes, row_bytes, (size_t)rows[k], hipMemcpyDeviceToHost, x->stream);
      if (e != hipSuccess) { sh[k].rc = RTX_ERR_HIP; sh[k].err = std::string("film copy to the host: ") + hipGetErrorString(e); }
    }
  };
  auto finish_share = [&](int k) {    // -> its kernels (and its copy to the host) are done, its statistics read
    if (rows[k] == 0) return;
    if (sh[k].collected) {   // (only the copy to the host may still be on its way)
      if (hipSetDevice(all[k]->device) != hipSuccess || hipStreamSynchronize(all[k]->stream) != hipSuccess) { if (sh[k].rc == RTX_OK) { sh[k].rc = RTX_ERR_HIP; sh[k].err = "waiting for the film copy failed"; } }
      return;
    }
    if (!all[k]->pending.active) return;
    const int rc = render_collect(all[k], &sh[k].st);
    if (rc != RTX_OK && sh[k].rc == RTX_OK) { sh[k].rc = rc; sh[k].err = g_error; }
  };
  auto guarded_part = [&](int k, auto&& part) {
    try { part(k);
    } catch (const std::bad_alloc&) { sh[k].rc = RTX_ERR_OUT_OF_MEMORY; sh[k].err = "host allocation failed";
    } catch (const std::exception& e) { sh[k].rc = RTX_ERR_HIP; sh[k].err = e.what(); }
  };
  for (int k = 1; k < n; ++k)
    c->workers[(size_t)k - 1]->post([&, k] {
      guarded_part(k, enqueue_share);
      announce(k);
      guarded_part(k, finish_share);
    });
  guarded_part(0, enqueue_share);
  announce(0);
  // the gather into device memory, issued from this thread as the devices report in
  int gather_rc = RTX_OK;
  std::string gather_err;
  hipStream_t gstream = user_stream ? user_stream : c->stream;
  auto gather = [&]() -> int {
    size_t peer_bytes = 0;
    for (int k = 1; k < n; ++k) peer_bytes += (size_t)rows[k] * row_bytes;
    HIP_TRY(hipSetDevice(c->device));
    if (peer_bytes && !c->gather.ensure(peer_bytes)) return fail(RTX_ERR_OUT_OF_MEMORY, "gather buffer (%zu bytes)", peer_bytes);
    const bool rccl = c->gather_mode == RTX_GATHER_RCCL && n > 1 && (int)c->comms.size() == n;
    const Rccl* R = rccl ? load_rccl(nullptr) : nullptr;
    std::vector<void*> stage(n, nullptr);
    size_t at = 0;
    for (int k = 1; k < n; ++k) { stage[k] = (char*)c->gather.ptr + at; at += (size_t)rows[k] * row_bytes; }
    for (int k = 0; k < n; ++k) {   // every device's kernels must be on its stream before anything is ordered behind them
      std::unique_lock<std::mutex> l(sh[k].m);
      sh[k].cv.wait(l, [&] { return sh[k].enqueued; });
      if (sh[k].rc != RTX_OK) return RTX_OK;   // reported below, with the share's own message
    }
    if (R) {
      int e = R->GroupStart();
      for (int k = 1; k < n && e == 0; ++k) {
        if (rows[k] == 0) continue;
        HIP_TRY(hipSetDevice(all[k]->device));
        e = R->Send(all[k]->film.ptr, (size_t)rows[k] * (size_t)s->width * 3, Rccl::kFloat, 0, c->comms[(size_t)k], all[k]->stream);   // behind the render, in stream order
      }
      HIP_TRY(hipSetDevice(c->device));
      if (e == 0 && gstream != c->stream) {   // the receives are ordered behind nothing of this device; its own rows are copied below
        HIP_TRY(hipStreamWaitEvent(gstream, c->ev1, 0));
      }
      for (int k = 1; k < n && e == 0; ++k)
        if (rows[k]) e = R->Recv(stage[k], (size_t)rows[k] * (size_t)s->width * 3, Rccl::kFloat, k, c->comms[0], gstream);
      const int e2 = R->GroupEnd();
      if (e != 0 || e2 != 0) return fail(RTX_ERR_HIP, "RCCL film gather: %s", R->GetErrorString(e != 0 ? e : e2));
    }
    for (int k = 0; k < n; ++k) {
      if (rows[k] == 0) continue;
      const void* src = all[k]->film.ptr;
      if (k > 0) {
        if (!R) {
          HIP_TRY(hipStreamWaitEvent(gstream, all[k]->ev1, 0));   // recorded behind device k's kernels
          HIP_TRY(hipMemcpyPeerAsync(stage[k], c->device, src, all[k]->device, (size_t)rows[k] * row_bytes, gstream));
        }
        src = stage[k];
      } else if (gstream != c->stream && !R) {
        HIP_TRY(hipStreamWaitEvent(gstream, c->ev1, 0));
      }
      HIP_TRY(hipMemcpy2DAsync((char*)dev_out + (size_t)k * row_bytes, (size_t)n * row_bytes, src, row_bytes, row_bytes, (size_t)rows[k], hipMemcpyDeviceToDevice, gstream));
    }
    return RTX_OK;
  };
  if (dev_out) {
    try { gather_rc = gather();
    } catch (const std::bad_alloc&) { gather_rc = fail(RTX_ERR_OUT_OF_MEMORY, "host allocation failed");
    } catch (const std::exception& e) { gather_rc = fail(RTX_ERR_HIP, "internal error: %s", e.what()); }
    if (gather_rc != RTX_OK) gather_err = g_error;
  }
  guarded_part(0, finish_share);
  for (int k = 1; k < n; ++k) c->workers[(size_t)k - 1]->wait();   // nothing below may run while a worker still uses this frame
  (void)hipSetDevice(c->device);
  if (dev_out) {
    const hipError_t e = hipStreamSynchronize(gstream);
    if (e != hipSuccess && gather_rc == RTX_OK) { gather_rc = RTX_ERR_HIP; gather_err = std::string("film gather: ") + hipGetErrorString(e); }
  }
  for (int k = 0; k < n; ++k)
    if (sh[k].rc != RTX_OK) return fail(sh[k].rc, "render (device %d, shard %d of %d): %s", all[k]->device, sk[k].shard_index, sk[k].shard_count, sh[k].err.c_str());
  if (gather_rc != RTX_OK) return fail(gather_rc, "%s", gather_err.c_str());
  if (stats) {
    rtx_stats t = sh[0].st;
    for (int k = 1; k < n; ++k) {
      const rtx_stats& a = sh[k].st;
      t.samples += a.samples; t.rays_closest += a.rays_closest; t.rays_shadow += a.rays_shadow; t.segments += a.segments;
      t.nodes_visited += a.nodes_visited; t.prims_tested += a.prims_tested; t.kernel_launches += a.kernel_launches;
      // the devices run side by side: the call takes as long as the slowest
      t.seconds_render = std::max(t.seconds_render, a.seconds_render); t.seconds_main_kernel = std::max(t.seconds_main_kernel, a.seconds_main_kernel);
      t.seconds_photon_pass = std::max(t.seconds_photon_pass, a.seconds_photon_pass); t.seconds_extend = std::max(t.seconds_extend, a.seconds_extend);
      t.flags |= a.flags;
      if (t.pipeline_used == 0) { t.pipeline_used = a.pipeline_used; t.precision_used = a.precision_used; }
    }
    t.seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    *stats = t;
  }
  return RTX_OK;
}

// RCCL for the film gather of a multi-device context: one communicator per device, created here.
static int set_gather_impl(rtx_ctx* c, int mode) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL context");
  if (!c->multi) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_multi_set_gather takes a context made by rtx_create_multi");
  if (mode != RTX_GATHER_PEER_COPY && mode != RTX_GATHER_RCCL) return fail(RTX_ERR_INVALID_ARGUMENT, "gather mode %d", mode);
  if (mode == RTX_GATHER_PEER_COPY) { c->gather_mode = mode; return RTX_OK; }
  std::vector<int> devs{c->device};
  for (rtx_ctx* p : c->peers) devs.push_back(p->device);
  for (size_t i = 0; i < devs.size(); ++i)
    for (size_t j = i + 1; j < devs.size(); ++j)
      if (devs[i] == devs[j]) return fail(RTX_ERR_UNSUPPORTED, "an RCCL communicator takes every device once: device %d is listed twice", devs[i]);
  std::string why;
  const Rccl* R = load_rccl(&why);
  if (!R) return fail(RTX_ERR_UNSUPPORTED, "%s", why.c_str());
  if (c->comms.empty()) {
    std::vector<void*> comms(devs.size(), nullptr);
    const int e = R->CommInitAll(comms.data(), (int)devs.size(), devs.data());
    (void)hipSetDevice(c->device);
    if (e != 0) return fail(RTX_ERR_HIP, "ncclCommInitAll over %zu devices: %s", devs.size(), R->GetErrorString(e));
    c->comms = comms;
  }
  c->gather_mode = mode;
  return RTX_OK;
}

int rtx_render(rtx_ctx* c, const rtx_settings* s, float* out_rgb, rtx_stats* stats) {
  if (!out_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "out_rgb is NULL");
  if (c && c->multi) return guarded([&] { return render_multi(c, s, out_rgb, nullptr, nullptr, stats); });
  return guarded([&] { return render_impl(c, s, out_rgb, nullptr, nullptr, stats); });
}

int rtx_render_device_noalloc(rtx_ctx* c, const rtx_settings* s, void* d_out_rgb, void* hip_stream, rtx_stats* stats) {
  if (!d_out_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "d_out_rgb is NULL");
  if (c && c->multi) return fail(RTX_ERR_UNSUPPORTED, "the no-allocation render takes a single-device context (rtx_create)");
  return guarded([&] { return render_impl(c, s, nullptr, d_out_rgb, (hipStream_t)hip_stream, stats, true); });
}

int rtx_render_device(rtx_ctx* c, const rtx_settings* s, void* d_out_rgb, void* hip_stream, rtx_stats* stats) {
  if (!d_out_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "d_out_rgb is NULL");
  if (c && c->multi) return guarded([&] { return render_multi(c, s, nullptr, d_out_rgb, (hipStream_t)hip_stream, stats); });
  return guarded([&] { return render_impl(c, s, nullptr, d_out_rgb, (hipStream_t)hip_stream, stats); });
}

int rtx_render_device_async(rtx_ctx* c, const rtx_settings* s, void* d_out_rgb, void* hip_stream) {
  if (!d_out_rgb) return fail(RTX_ERR_INVALID_ARGUMENT, "d_out_rgb is NULL");
  if (c && c->multi) return fail(RTX_ERR_UNSUPPORTED, "the asynchronous render takes a single-device context (rtx_create)");
  return guarded([&] {
    // The film must be complete in stream order (the caller queues its own work behind the render), so this entry point never
    // renders ahead of the reference-order tree - such a frame has work items rendered again when it is collected: it waits for
    // the tree of a large scene here, once.
    if (c && c->has_scene) { const int rc = need_tree(c); if (rc != RTX_OK) return rc; }
    return render_enqueue(c, s, nullptr, d_out_rgb, (hipStream_t)hip_stream, false, false, true);
  });
}

int rtx_render_wait(rtx_ctx* c, rtx_stats* stats) {
  if (c && c->multi) return fail(RTX_ERR_UNSUPPORTED, "the asynchronous render takes a single-device context (rtx_create)");
  return guarded([&] { return render_collect(c, stats); });
}

int rtx_multi_set_gather(rtx_ctx* c, int mode) { return guarded([&] { return set_gather_impl(c, mode); }); }

static int postprocess_impl(rtx_ctx* c, const float* h_film, const void* d_film, int width, int height, int mapper, double exposure, uint8_t* h_out,
                            void* d_out, hipStream_t user_stream) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL context");
  if (width <= 0 || height <= 0 || (unsigned long long)width * height > 0x7fffffffull / 3ull)
    return fail(RTX_ERR_INVALID_ARGUMENT, "bad image size %dx%d", width, height);
  if (mapper < RTX_TONE_NONE || mapper > RTX_TONE_UNCHARTED2) return fail(RTX_ERR_INVALID_ARGUMENT, "unknown tone mapper %d", mapper);
  if (!(exposure == exposure)) return fail(RTX_ERR_INVALID_ARGUMENT, "exposure is NaN");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t stream = user_stream ? user_stream : c->stream;
  const size_t n = (size_t)width * height * 3;
  const float* in = (const float*)d_film;
  unsigned char* out = (unsigned char*)d_out;
  if (h_film) {
    if (!c->post_in.ensure(n * sizeof(float)) || !c->post_out.ensure(n)) return fail(RTX_ERR_OUT_OF_MEMORY, "post-process buffers (%zu values)", n);
    HIP_TRY(hipMemcpyAsync(c->post_in.ptr, h_film, n * sizeof(float), hipMemcpyHostToDevice, stream));
    in = (const float*)c->post_in.ptr;
    out = (unsigned char*)c->post_out.ptr;
  }
  launch_postprocess(in, out, width, height, mapper, exposure, stream);
  HIP_TRY(hipGetLastError());
  if (h_out) HIP_TRY(hipMemcpyAsync(h_out, out, n, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  return RTX_OK;
}

int rtx_postprocess(rtx_ctx* c, const float* film_rgb, int width, int height, int tone_mapper, double exposure, uint8_t* out_rgb8) {
  if (!film_rgb || !out_rgb8) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL buffer");
  return guarded([&] { return postprocess_impl(c, film_rgb, nullptr, width, height, tone_mapper, exposure, out_rgb8, nullptr, nullptr); });
}

int rtx_postprocess_device(rtx_ctx* c, const void* d_film_rgb, int width, int height, int tone_mapper, double exposure, void* d_out_rgb8, void* hip_stream) {
  if (!d_film_rgb || !d_out_rgb8) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL buffer");
  return guarded([&] { return postprocess_impl(c, nullptr, d_film_rgb, width, height, tone_mapper, exposure, nullptr, d_out_rgb8, (hipStream_t)hip_stream); });
}

static int trace_impl(rtx_ctx* c, const double* rays8, size_t n, rtx_hit* out, int mode /*0 ref, 1 fast f32, 2 fast f64*/) {
  if (!c || (!rays8 && n) || (!out && n)) return fail(RTX_ERR_INVALID_ARGUMENT, "NULL argument");
  if (!c->has_scene) return fail(RTX_ERR_NOT_READY, "trace before upload_scene");
  if (n == 0) return RTX_OK;
  settle(c);
  { const int rc = need_tree(c); if (rc != RTX_OK) return rc; }
  static_assert(sizeof(DHitOut) == sizeof(rtx_hit), "hit layouts must agree");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->rays.ensure(n * 8 * sizeof(double)) || !c->hits.ensure(n * sizeof(DHitOut))) return fail(RTX_ERR_OUT_OF_MEMORY, "ray batch of %zu", n);
  HIP_TRY(hipMemcpyAsync(c->rays.ptr, rays8, n * 8 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  const size_t lds = stack_words(c, mode != 1) * kBlock * sizeof(int);
  if (mode == 0) launch_trace_ref(c->scene.dev_d, c->scene.dev_ref, (const double*)c->rays.ptr, n, (DHitOut*)c->hits.ptr, c->stream);
  else if (mode == 1) launch_trace_fast(c->scene.dev_f, (const double*)c->rays.ptr, n, (DHitOut*)c->hits.ptr, c->scene.fused_mode(false) == MODE_FLAT ? MODE_FLAT : MODE_TREE_UV, lds, c->stream, c->n_cus);
  else launch_trace_fast(c->scene.dev_d, (const double*)c->rays.ptr, n, (DHitOut*)c->hits.ptr, MODE_TREE_UV, lds, c->stream, c->n_cus);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, c->hits.ptr, n * sizeof(DHitOut), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTX_OK;
}

int rtx_trace_closest(rtx_ctx* c, const double* rays8, size_t n, rtx_hit* out) { return guarded([&] { return trace_impl(c, rays8, n, out, 0); }); }

int rtx_trace_closest_fast(rtx_ctx* c, const double* rays8, size_t n, int precision, rtx_hit* out) {
  if (precision != RTX_PRECISION_F32 && precision != RTX_PRECISION_F64) return fail(RTX_ERR_INVALID_ARGUMENT, "precision %d", precision);
  return guarded([&] { return trace_impl(c, rays8, n, out, precision == RTX_PRECISION_F32 ? 1 : 2); });
}

}  // extern "C"
