// C++ host-side mirror of the reference's GPU plug-in class, bound to the C ABI of
// include/rtx_hip.h.  Header-only; link against ray-tracer_amd/lib/librtx_hip.so.
//
// It keeps the surface of  raytracer::cuda::GPUPathTracer  (ref: engine/cuda/cuda_api.hpp:72-110):
// non-copyable, `set_settings` / `upload_scene` / `set_camera` / `render(float*)`, all `void`,
// failures reported on stderr and the output left untouched (ref: engine/cuda/cuda_pathtracer.cu:367-372,
// engine/cuda/cuda_common.cuh:124-131).  In addition the status of the last call can be queried.
#pragma once

#include <cstdio>
#include <vector>

#include "../../include/rtx_hip.h"

namespace raytracer {
namespace hip {

// ref: raytracer::cuda::test_cuda(), engine/cuda/cuda_api.hpp:17
inline bool test_hip(int device_id = 0) {
  if (rtx_selftest(device_id) == RTX_OK) return true;
  std::fprintf(stderr, "HIP self test failed: %s\n", rtx_last_error());
  return false;
}

class HipPathTracer {
 public:
  explicit HipPathTracer(int device_id = 0) { note(rtx_create(&ctx_, device_id), "rtx_create"); }
  // One tracer over several GPUs of the node (rtx_create_multi): the same calls, the film's rows dealt to the devices and gathered.
  explicit HipPathTracer(const std::vector<int>& device_ids) { note(rtx_create_multi(&ctx_, device_ids.data(), (int)device_ids.size()), "rtx_create_multi"); }
  ~HipPathTracer() { if (ctx_) rtx_destroy(ctx_); }
  HipPathTracer(const HipPathTracer&) = delete;
  HipPathTracer& operator=(const HipPathTracer&) = delete;

  void upload_scene(const rtx_scene_desc& scene) { if (ctx_) note(rtx_upload_scene(ctx_, &scene), "upload_scene"); }
  void set_settings(const rtx_settings& settings) { settings_ = settings; have_settings_ = true; }
  void set_camera(const rtx_camera& camera) { if (ctx_) note(rtx_set_camera(ctx_, &camera), "set_camera"); }
  // output: width * height * 3 floats (rows of the shard when sharded), row 0 = bottom,
  // linear averaged radiance (ref: engine/main.cpp:394-397)
  void render(float* output) {
    if (!ctx_) return;
    if (!have_settings_) { status_ = RTX_ERR_NOT_READY; std::fprintf(stderr, "HIP render: set_settings was not called\n"); return; }
    note(rtx_render(ctx_, &settings_, output, &stats_), "render");
  }
  // exposure -> tone map -> sqrt gamma -> 8-bit, rows flipped to top-first
  // (ref: Image::apply_tone_mapping + Image::write_png, engine/renderer.cpp:186-198, :232-258)
  void postprocess(const float* film, int width, int height, int tone_mapper, double exposure, unsigned char* out_rgb8) {
    if (ctx_) note(rtx_postprocess(ctx_, film, width, height, tone_mapper, exposure, out_rgb8), "postprocess");
  }
  // Additions of this backend: the film in device memory (of the first device), also in two halves (enqueue on the caller's
  // stream, wait later), and the choice of the multi-device film gather (RTX_GATHER_PEER_COPY / RTX_GATHER_RCCL).
  void render_device(void* d_output, void* hip_stream = nullptr) {
    if (ready("render_device")) note(rtx_render_device(ctx_, &settings_, d_output, hip_stream, &stats_), "render_device");
  }
  void render_device_async(void* d_output, void* hip_stream) {
    if (ready("render_device_async")) note(rtx_render_device_async(ctx_, &settings_, d_output, hip_stream), "render_device_async");
  }
  void render_wait() { if (ctx_) note(rtx_render_wait(ctx_, &stats_), "render_wait"); }
  void set_gather(int mode) { if (ctx_) note(rtx_multi_set_gather(ctx_, mode), "set_gather"); }
  void trace_closest(const double* rays8, size_t n, rtx_hit* out) { if (ctx_) note(rtx_trace_closest(ctx_, rays8, n, out), "trace_closest"); }

  bool ok() const { return status_ == RTX_OK; }
  int status() const { return status_; }
  const rtx_stats& stats() const { return stats_; }

 private:
  bool ready(const char* what) {
    if (!ctx_) return false;
    if (!have_settings_) { status_ = RTX_ERR_NOT_READY; std::fprintf(stderr, "HIP %s: set_settings was not called\n", what); return false; }
    return true;
  }
  void note(int rc, const char* what) {
    status_ = rc;
    if (rc != RTX_OK) std::fprintf(stderr, "HIP %s failed (%d): %s\n", what, rc, rtx_last_error());
  }
  rtx_ctx* ctx_ = nullptr;
  rtx_settings settings_{};
  rtx_stats stats_{};
  bool have_settings_ = false;
  int status_ = RTX_OK;
};

}  // namespace hip
}  // namespace raytracer
