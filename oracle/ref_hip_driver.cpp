// TEST INFRASTRUCTURE ONLY.  End-to-end drop-in check, built by `make -C oracle ref` where
// /root/reference exists and runnable on a GPU box (the binary travels in oracle/_ref/):
//
//   the REFERENCE's SceneLoader::load  ->  the REFERENCE's Scene / Camera objects
//        -> ray-tracer_amd/host/reference_adapter.hpp -> C ABI -> MI355X kernels      (film A)
//        -> the REFERENCE's own Renderer::render on the host cores (use_mnee = false unless asked for)  (film B)
//
//   ref_hip_driver <scene.json> <W> <H> <spp_gpu> <spp_cpu> <depth> <threads> <out_prefix> [cpu_seed [mode [photons gather_count gather_radius]]]
// (mode: RenderMode value, 1 = PathTrace by default, 0 = Whitted, 2 = BDPT, 6 = PathPhoton; both renderers use it.
//  The three photon arguments replace Renderer::Settings' caustic photon defaults.  A mode written with a trailing 'm',
//  e.g. 1m, switches use_mnee on for both renderers.)
// writes <out_prefix>_gpu.bin (float32) and <out_prefix>_cpu.bin (float64, unclamped) and
// prints timings.  spp_cpu = 0 skips the CPU render.
#include <omp.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "scene_loader.hpp"  // reference, through -I/root/reference/engine
#include "../ray-tracer_amd/host/hip_path_tracer.hpp"
#include "../ray-tracer_amd/host/reference_adapter.hpp"

namespace raytracer { extern thread_local std::mt19937 rng; }
using namespace raytracer;

int main(int argc, char** argv) {
  if (argc < 9) { std::fprintf(stderr, "usage: ref_hip_driver scene W H spp_gpu spp_cpu depth threads out_prefix [cpu_seed]\n"); return 2; }
  const int W = atoi(argv[2]), H = atoi(argv[3]), spp_gpu = atoi(argv[4]), spp_cpu = atoi(argv[5]), depth = atoi(argv[6]), threads = atoi(argv[7]);
  const std::string prefix = argv[8];
  const unsigned cpu_seed = argc > 9 ? (unsigned)strtoul(argv[9], nullptr, 10) : 4711u;
  auto data = SceneLoader::load(argv[1]);
  Renderer::Settings st;
  st.width = W; st.height = H; st.samples_per_pixel = spp_gpu; st.max_depth = depth;
  const int mode = argc > 10 ? atoi(argv[10]) : 1;
  st.mode = mode == 0 ? RenderMode::Whitted : mode == 2 ? RenderMode::BDPT : mode == 6 ? RenderMode::PathPhoton : RenderMode::PathTrace;
  if (argc > 13) { st.caustic_photon_count = (size_t)strtoull(argv[11], nullptr, 10); st.photon_gather_count = atoi(argv[12]); st.caustic_gather_radius = (float)atof(argv[13]); }
  st.use_nee = data.use_nee; st.use_mis = data.use_mis; st.use_mnee = argc > 10 && strchr(argv[10], 'm') != nullptr;
  st.background_color = data.background_color; st.background_top = data.background_top;
  st.background_bottom = data.background_bottom; st.use_background_gradient = data.use_background_gradient;
  Camera cam(data.camera_position, data.camera_target, data.camera_up, data.camera_fov, (double)W / H);

  if (spp_gpu > 0) {
    hip::SceneBuffers buf;
    hip::HipPathTracer gpu(0);
    gpu.set_settings(hip::to_settings(st));          // same order as main.cpp:353-391
    gpu.upload_scene(hip::to_scene_desc(data.scene, buf));
    gpu.set_camera(hip::to_camera(cam));
    std::vector<float> out((size_t)W * H * 3, -1.0f);
    gpu.render(out.data());
    if (!gpu.ok()) return 3;
    FILE* f = fopen((prefix + "_gpu.bin").c_str(), "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    const rtx_stats& s = gpu.stats();
    printf("@@GPU@@ %.6f s kernel  %.3f Msamples/s  %.3f Mrays/s\n", s.seconds_render, s.samples / s.seconds_render * 1e-6,
           (s.rays_closest + s.rays_shadow) / s.seconds_render * 1e-6);
  }
  if (spp_cpu > 0) {
    st.samples_per_pixel = spp_cpu;
    st.tone_mapper = ToneMapper::None;
    const double k = 1048576.0;   // exact power-of-two pre-scale keeps the [0,1] clamp of ToneMapper::None out of the way
    st.exposure = 1.0 / k;
    omp_set_dynamic(0);
    omp_set_num_threads(threads);
#pragma omp parallel
    { raytracer::rng.seed(cpu_seed + (unsigned)omp_get_thread_num()); }
    Renderer renderer(st);
    auto t0 = std::chrono::steady_clock::now();
    Image img = renderer.render(data.scene, cam);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    FILE* f = fopen((prefix + "_cpu.bin").c_str(), "wb");
    for (const color& c : img.pixels) { double v[3] = {c.x * k, c.y * k, c.z * k}; fwrite(v, 8, 3, f); }
    fclose(f);
    printf("\n@@CPU@@ %.6f s  %d threads  %.6f Msamples/s\n", secs, threads, (double)W * H * spp_cpu / secs * 1e-6);
  }
  return 0;
}
