// Calibrates the memory-side counters (FETCH_SIZE, TCC_EA0_RDREQ) on the access pattern of the tree walk: every lane fetches
// one 64-byte record at a random place of a table far larger than any cache, with four 16-byte loads.  Three kernels, each
// with a byte count known in advance:
//   gather64    one random 64-byte record per lane and turn (the records of a wave share no line)
//   gather64x4  the same with four independent records in flight per lane
//   gather128   one random 128-byte-aligned pair of records per lane and turn
//   stream      every lane reads 16 bytes, consecutive lanes consecutive addresses (the pattern the 2x correction was found on)
// Run each under  rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum  and divide the counters by the bytes printed here.
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_calib.hip -o scripts/bin/gather_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// INFLIGHT independent records per lane and turn: how much the memory system delivers when latency is hidden.
template <int UNITS, int INFLIGHT> __global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ table, unsigned n_records, int turns, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0u;
  for (int t = 0; t < turns; ++t) {
    uint4 v[INFLIGHT][4 * UNITS];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      const unsigned rec = hash32(tid * 977u + (unsigned)(t * INFLIGHT + j) * 0x9e3779b9u) % n_records;
      const uint4* p = table + (size_t)rec * (4 * UNITS);
#pragma unroll
      for (int k = 0; k < 4 * UNITS; ++k) v[j][k] = p[k];
    }
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j)
#pragma unroll
      for (int k = 0; k < 4 * UNITS; ++k) acc += v[j][k].x ^ v[j][k].y ^ v[j][k].z ^ v[j][k].w;
  }
  if (acc == 0x12345678u) out[0] = acc;   // never true for the table's contents
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ table, size_t n16, unsigned* out) {
  unsigned acc = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = table[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "gather64";
  const size_t mib = argc > 2 ? (size_t)atol(argv[2]) : 4096;   // table size: 4 GiB = sixteen times the 256 MiB on-die cache
  const size_t bytes = mib << 20;
  uint4* table = nullptr;
  unsigned* out = nullptr;
  CHECK(hipMalloc(&table, bytes));
  CHECK(hipMalloc(&out, 16));
  CHECK(hipMemset(table, 0x5a, bytes));
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 256 * 8, turns = 64;
  double known = 0.0;
  CHECK(hipEventRecord(e0, 0));
  if (!strcmp(which, "gather64")) {
    hipLaunchKernelGGL((gather_kernel<1, 1>), dim3(blocks), dim3(256), 0, 0, table, (unsigned)(bytes / 64), turns, out);
    known = (double)blocks * 256.0 * turns * 64.0;
  } else if (!strcmp(which, "gather64x4")) {
    hipLaunchKernelGGL((gather_kernel<1, 4>), dim3(blocks), dim3(256), 0, 0, table, (unsigned)(bytes / 64), turns, out);
    known = (double)blocks * 256.0 * turns * 4.0 * 64.0;
  } else if (!strcmp(which, "gather128")) {
    hipLaunchKernelGGL((gather_kernel<2, 1>), dim3(blocks), dim3(256), 0, 0, table, (unsigned)(bytes / 128), turns, out);
    known = (double)blocks * 256.0 * turns * 128.0;
  } else {
    hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, 0, table, bytes / 16, out);
    known = (double)bytes;
  }
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"pattern\": \"%s\", \"table_MiB\": %zu, \"known_bytes\": %.0f, \"ms\": %.3f, \"GBps\": %.1f}\n", which, mib, known, ms, known / (ms * 1e6));
  return 0;
}
