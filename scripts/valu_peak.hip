// Measures what one MI355X sustains in vector ALU wave-instructions per second, the ceiling the path-tracing kernels are
// priced against: independent instruction streams of one opcode at 1, 2, 4 and 8 waves per SIMD, every launch long enough
// (>= 100 ms) for the clocks to settle.  The "lanes" variants run the same stream with only the first 32 / 16 lanes of each
// wave active: whether the hardware skips the empty quarter-wave passes decides what a divergent kernel can reach.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_peak.hip -o scripts/bin/valu_peak && scripts/bin/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 16;   // independent instructions per loop trip

template <int OP, int LANES = 64> __global__ __launch_bounds__(256) void stream_kernel(float* out, int iters, float seed) {
  float a[kUnroll];
  unsigned u[kUnroll];
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) { a[k] = seed + (float)(threadIdx.x + k); u[k] = (unsigned)(threadIdx.x * 2654435761u + k) | 1u; }
  const float b = 1.0000001f, c = 1e-7f;
  if ((int)(threadIdx.x & 63) >= LANES) iters = 0;   // these lanes wait at the end of the loop: the wave runs on with EXEC = the others
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
      else if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) % kUnroll]));
      else if (OP == 2) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[k]) : "v"(u[k]));
      else if (OP == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
      else if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
      else if (OP == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
      else if (OP == 7) asm volatile("v_sin_f32 %0, %0" : "+v"(a[k]));
      else if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(a[(k + 1) % kUnroll]));
      else if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : );
      else if (OP == 10) { unsigned long long w; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w) : "v"(u[k]), "v"(u[(k + 1) % kUnroll]) : "vcc"); u[k] = (unsigned)w; }
      else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(*reinterpret_cast<double*>(&a[k & ~1])) : "v"((double)b), "v"((double)c));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) s += a[k] + (float)u[k];
  if (s == 12345.678f) out[0] = s;   // never true: keeps the streams alive
}

template <int OP, int LANES = 64> void run(const char* name, int cus, float* d_out, bool all_occupancies = true) {
  for (int blocks_per_cu : {1, 2, 4, 8}) {   // 256 threads = one wave per SIMD per block
    if (!all_occupancies && blocks_per_cu != 8) continue;
    const int iters = 4000000 / blocks_per_cu;   // >= 100 ms per launch at any occupancy
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(cus * blocks_per_cu), block(256);
    hipLaunchKernelGGL((stream_kernel<OP, LANES>), grid, block, 0, 0, d_out, iters / 4, 1.0f);   // warm-up: brings the clocks up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((stream_kernel<OP, LANES>), grid, block, 0, 0, d_out, iters, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)grid.x * 4.0 * iters * kUnroll;   // wave-instructions
    const double rate = insts / (ms * 1e-3);
    printf("{\"op\": \"%s\", \"active_lanes\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_insts_per_s\": %.4e, \"cycles_per_inst_per_simd_at_2.4GHz\": %.3f}\n", name, LANES, blocks_per_cu, ms,
           rate, (double)cus * 4.0 * 2.4e9 / rate);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.gcnArchName, cus, prop.clockRate);
  float* d_out = nullptr;
  CHECK(hipMalloc(&d_out, 256));
  run<0>("v_fma_f32", cus, d_out);
  run<0, 32>("v_fma_f32", cus, d_out, false);
  run<0, 16>("v_fma_f32", cus, d_out, false);
  run<3>("v_min_f32", cus, d_out, false);
  run<8>("v_mov_b32", cus, d_out, false);
  run<9>("v_cndmask_b32", cus, d_out, false);
  run<2>("v_cvt_f32_ubyte1", cus, d_out, false);
  run<1>("v_mul_lo_u32", cus, d_out, false);
  run<10>("v_mad_u64_u32", cus, d_out, false);
  run<4>("v_rcp_f32", cus, d_out, false);
  run<6>("v_sqrt_f32", cus, d_out, false);
  run<7>("v_sin_f32", cus, d_out, false);
  run<5>("v_fma_f64", cus, d_out, false);
  CHECK(hipFree(d_out));
  return 0;
}
