// What one returning atomicAdd per workgroup on a shared counter costs a streaming kernel: workgroups of 1024 / 256 threads, each
// with one atomic by thread 0 whose result every thread waits for (the queue append of the queue pipeline's streaming kernels),
// on ONE address or spread over 64 lines; 16 M threads per launch.
//   hipcc --offload-arch=gfx950 -O3 scripts/atomic_rate.hip -o scripts/bin/atomic_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BANKS, int ATOMICS> __global__ void append_kernel(unsigned* counters, unsigned* out) {
  __shared__ unsigned base;
  if (threadIdx.x == 0) {
    unsigned b = 0;
    for (int k = 0; k < ATOMICS; ++k) b += atomicAdd(&counters[(blockIdx.x % BANKS) * 32 + k * 2048], blockDim.x);
    base = b;
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = base + threadIdx.x;
}

template <int BANKS, int ATOMICS> void run(const char* name, int block, unsigned* counters, unsigned* out, size_t n) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid((unsigned)(n / block));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(counters, 0, 65536 * 4));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((append_kernel<BANKS, ATOMICS>), grid, dim3(block), 0, 0, counters, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2)
      printf("{\"case\": \"%s\", \"workgroup\": %d, \"workgroups\": %u, \"atomics_per_workgroup\": %d, \"ms\": %.3f, \"ns_per_atomic_if_serial\": %.1f}\n", name, block, grid.x,
             ATOMICS, ms, ATOMICS ? ms * 1e6 / ((double)grid.x * ATOMICS) : 0.0);
  }
}

int main() {
  const size_t n = (size_t)16 << 20;
  unsigned *counters = nullptr, *out = nullptr;
  CHECK(hipMalloc(&counters, 65536 * 4));
  CHECK(hipMalloc(&out, n * 4));
  run<1, 0>("no atomic", 1024, counters, out, n);
  run<1, 1>("one address", 1024, counters, out, n);
  run<1, 2>("two addresses, one after the other", 1024, counters, out, n);
  run<64, 1>("64 banks", 1024, counters, out, n);
  run<1, 0>("no atomic", 256, counters, out, n);
  run<1, 1>("one address", 256, counters, out, n);
  run<64, 1>("64 banks", 256, counters, out, n);
  return 0;
}
