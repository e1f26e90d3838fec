// Which vector-ALU opcodes share an issue slot on one MI355X: streams that alternate two opcodes (16 independent instructions per
// loop trip, 8 of each), 8 waves per SIMD, launches of about 100 ms.  A SIMD can issue two wave64 instructions of different waves
// in one four-cycle slot; whether it does depends on what the two instructions are.  The rate of a mix against the rates of its
// two pure streams says whether the pair co-issues.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_mix.hip -o scripts/bin/valu_mix && scripts/bin/valu_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 16;

enum Op { FMA, MOV, AND, MINU, CVTUB, MAXF, ADDU, LSHL, RCP, MULF, ADDF, MIN3F, CNDMASK, MULLO, FMA64, PERM, ORSDWA, BFE, CMPF, LSHLOR, PKFMA16, PKMAX16, CNDM, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_mov_b32", "v_and_b32", "v_min_u32", "v_cvt_f32_ubyte1", "v_max_f32", "v_add_u32", "v_lshlrev_b32", "v_rcp_f32", "v_mul_f32",
                                    "v_add_f32", "v_min3_f32", "v_cndmask_b32", "v_mul_lo_u32", "v_fma_f64", "v_perm_b32", "v_or_b32_sdwa(byte1)", "v_bfe_u32", "v_cmp_lt_f32", "v_lshl_or_b32", "v_pk_fma_f16", "v_pk_max_f16", "v_cndmask_b32(sgpr mask)"};

template <int OP> __device__ __forceinline__ void issue(float& a, unsigned& u, float b, float c, unsigned m, float other) {
  if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  else if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(other));
  else if (OP == AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u) : "v"(m));
  else if (OP == MINU) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u) : "v"(m));
  else if (OP == CVTUB) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a) : "v"(u));
  else if (OP == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if (OP == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(m));
  else if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u));
  else if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
  else if (OP == MULF) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  else if (OP == ADDF) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
  else if (OP == MIN3F) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
  else if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(m));
  else if (OP == FMA64) { }   // (handled by the caller: needs a register pair)
  else if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(m), "v"(0x07060100u));
  else if (OP == ORSDWA) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(0x4b000000u), "v"(u));
  else if (OP == BFE) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a) : "v"(u));
  else if (OP == CMPF) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
  else if (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u) : "v"(m));
  else if (OP == PKFMA16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u) : "v"(0x3c003c00u), "v"(0x00010001u));
  else if (OP == PKMAX16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u) : "v"(0x3c003c00u));
  else if (OP == CNDM) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(0x5555555555555555ull));
}

template <int A, int B> __global__ __launch_bounds__(256) void mix_kernel(float* out, int iters, float seed) {
  float a[kUnroll];
  unsigned u[kUnroll];
  double d[kUnroll / 2];
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) { a[k] = seed + (float)(threadIdx.x + k); u[k] = (unsigned)(threadIdx.x * 2654435761u + k) | 1u; }
#pragma unroll
  for (int k = 0; k < kUnroll / 2; ++k) d[k] = (double)a[k];
  const float b = 1.0000001f, c = 1e-7f;
  const double bd = 1.0000001, cd = 1e-7;
  const unsigned m = 0xfffffff7u + (unsigned)seed;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      constexpr int dummy = 0;
      if ((k & 1) == 0) {
        if (A == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k / 2]) : "v"(bd), "v"(cd));
        else issue<A>(a[k], u[k], b, c, m, a[(k + 2) % kUnroll]);
      } else {
        if (B == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k / 2]) : "v"(bd), "v"(cd));
        else issue<B>(a[k], u[k], b, c, m, a[(k + 2) % kUnroll]);
      }
      (void)dummy;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) s += a[k] + (float)u[k];
#pragma unroll
  for (int k = 0; k < kUnroll / 2; ++k) s += (float)d[k];
  if (s == 12345.678f) out[0] = s;
}

template <int A, int B> void run(int cus, float* d_out) {
  const int blocks_per_cu = 8, iters = 250000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid(cus * blocks_per_cu), block(256);
  hipLaunchKernelGGL((mix_kernel<A, B>), grid, block, 0, 0, d_out, iters / 4, 1.0f);
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((mix_kernel<A, B>), grid, block, 0, 0, d_out, iters, 1.0f);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double insts = (double)grid.x * 4.0 * iters * kUnroll;
  const double rate = insts / (ms * 1e-3);
  printf("{\"a\": \"%s\", \"b\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_insts_per_s\": %.4e, \"cycles_per_inst_per_simd_at_2.4GHz\": %.3f}\n", kNames[A], kNames[B], blocks_per_cu, ms,
         rate, (double)cus * 4.0 * 2.4e9 / rate);
  fflush(stdout);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, cus);
  float* d_out = nullptr;
  CHECK(hipMalloc(&d_out, 256));
  // pure streams
  run<FMA, FMA>(cus, d_out); run<MOV, MOV>(cus, d_out); run<AND, AND>(cus, d_out); run<MINU, MINU>(cus, d_out); run<CVTUB, CVTUB>(cus, d_out); run<MAXF, MAXF>(cus, d_out);
  run<ADDU, ADDU>(cus, d_out); run<LSHL, LSHL>(cus, d_out); run<RCP, RCP>(cus, d_out); run<MULF, MULF>(cus, d_out); run<ADDF, ADDF>(cus, d_out); run<MIN3F, MIN3F>(cus, d_out);
  run<MULLO, MULLO>(cus, d_out); run<FMA64, FMA64>(cus, d_out); run<PERM, PERM>(cus, d_out);
  // mixes with the multiply-add
  run<FMA, MOV>(cus, d_out); run<FMA, AND>(cus, d_out); run<FMA, MINU>(cus, d_out); run<FMA, CVTUB>(cus, d_out); run<FMA, MAXF>(cus, d_out); run<FMA, ADDU>(cus, d_out);
  run<FMA, LSHL>(cus, d_out); run<FMA, RCP>(cus, d_out); run<FMA, MULF>(cus, d_out); run<FMA, ADDF>(cus, d_out); run<FMA, MIN3F>(cus, d_out); run<FMA, MULLO>(cus, d_out);
  run<FMA, FMA64>(cus, d_out); run<FMA, PERM>(cus, d_out);
  // other pairs the tree walk is made of
  run<CVTUB, MAXF>(cus, d_out); run<MINU, AND>(cus, d_out); run<MAXF, MINU>(cus, d_out); run<CVTUB, MINU>(cus, d_out); run<MOV, AND>(cus, d_out); run<RCP, MOV>(cus, d_out);
  run<RCP, AND>(cus, d_out); run<FMA64, MOV>(cus, d_out); run<FMA64, RCP>(cus, d_out);
  run<ORSDWA, ORSDWA>(cus, d_out); run<BFE, BFE>(cus, d_out); run<CMPF, CMPF>(cus, d_out); run<LSHLOR, LSHLOR>(cus, d_out);
  run<FMA, ORSDWA>(cus, d_out); run<FMA, BFE>(cus, d_out); run<FMA, CMPF>(cus, d_out); run<FMA, LSHLOR>(cus, d_out); run<ORSDWA, MAXF>(cus, d_out); run<ORSDWA, MINU>(cus, d_out);
  run<CMPF, MAXF>(cus, d_out); run<AND, MAXF>(cus, d_out); run<ADDU, MINU>(cus, d_out); run<MULF, MAXF>(cus, d_out);
  run<PKFMA16, PKFMA16>(cus, d_out); run<PKMAX16, PKMAX16>(cus, d_out); run<CNDM, CNDM>(cus, d_out);
  run<PKFMA16, PKMAX16>(cus, d_out); run<PKFMA16, PERM>(cus, d_out); run<FMA, PKMAX16>(cus, d_out); run<FMA, PKFMA16>(cus, d_out); run<FMA, CNDM>(cus, d_out); run<CNDM, MAXF>(cus, d_out);
  CHECK(hipFree(d_out));
  return 0;
}
