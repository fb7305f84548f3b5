// Do MFMA and ordinary VALU instructions of DIFFERENT waves on one SIMD overlap in time, and do they inside ONE wave?
// One block of 256 * k threads per CU (k waves per SIMD).  Roles per wave: M = a chain of independent
// v_mfma_i32_32x32x32_i8 (4 accumulators), V = a chain of independent v_lshl_add_u32, X = both interleaved in one
// instruction stream (1 MFMA : R VALU).  Wall time of the launch / per-wave instruction counts -> SIMD cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

// mode bits per wave index (wave >> 2 = which wave of the SIMD): role[i] in {0 idle, 1 MFMA, 2 VALU, 3 interleaved}
template <int R>
__global__ __launch_bounds__(1024) void k(int* sink, int roles, int iters, int seed) {
  const int wave = threadIdx.x >> 6;
  const int role = (roles >> (4 * (wave >> 2))) & 15;       // waves 0-3 -> SIMDs 0-3 (first per SIMD), 4-7 second, ...
  i32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = seed + r;
  i32x4 a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed + 4, seed + 5, seed + 6, seed + 7};
  int v[8]; for (int i = 0; i < 8; i++) v[i] = seed + i + threadIdx.x;
  int p = seed * 3 + 1, q = seed + 7;
  if (role == 1) {
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  } else if (role == 2) {
    for (int it = 0; it < iters; it++)
      asm volatile(
        "v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
        "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
        "v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
        "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
        "v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
        "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
        "v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
        "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(p), "v"(q));
  } else if (role == 3) {
    for (int it = 0; it < iters; it++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; r++) asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(v[r & 7]) : "v"(p), "v"(q));
      }
  }
  int s = 0; for (int i = 0; i < 8; i++) s += v[i];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 0x1234567) sink[0] = 1;
}

template <int R>
static double run(int* sink, int nwaves_per_simd, int roles, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<R><<<256, 256 * nwaves_per_simd>>>(sink, roles, iters, 3);
  (void)hipEventRecord(e0); k<R><<<256, 256 * nwaves_per_simd>>>(sink, roles, iters, 3); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3;
}

int main() {
  int* sink; (void)hipMalloc(&sink, 64);
  const int it = 8192;                                  // per wave: 4 * it MFMAs (role M), 32 * it VALU (role V)
  printf("per wave: M = %d MFMA 32x32x32 i8, V = %d v_lshl_add_u32; one block of 256*k threads per CU\n", 4 * it, 32 * it);
  const double m1 = run<0>(sink, 1, 0x1, it), v1 = run<0>(sink, 1, 0x2, it);
  printf("1 wave/SIMD   M alone %8.1f us (%.1f ns per MFMA)   V alone %8.1f us (%.2f ns per VALU)\n", m1, m1 * 1e3 / (4 * it), v1, v1 * 1e3 / (32 * it));
  printf("2 waves/SIMD  M+M %8.1f   V+V %8.1f   M+V %8.1f   (sum of alone %8.1f, max %8.1f)\n",
         run<0>(sink, 2, 0x11, it), run<0>(sink, 2, 0x22, it), run<0>(sink, 2, 0x21, it), m1 + v1, m1 > v1 ? m1 : v1);
  printf("4 waves/SIMD  MMMM %8.1f   VVVV %8.1f   MMVV %8.1f   MVVV %8.1f   MMMV %8.1f\n",
         run<0>(sink, 4, 0x1111, it), run<0>(sink, 4, 0x2222, it), run<0>(sink, 4, 0x2211, it), run<0>(sink, 4, 0x2221, it), run<0>(sink, 4, 0x2111, it));
  printf("one wave, interleaved 1 MFMA : R VALU (4*it MFMA + 4*it*R VALU):  R=2 %8.1f   R=4 %8.1f   R=7 %8.1f   R=8 %8.1f   R=12 %8.1f\n",
         run<2>(sink, 1, 0x3, it), run<4>(sink, 1, 0x3, it), run<7>(sink, 1, 0x3, it), run<8>(sink, 1, 0x3, it), run<12>(sink, 1, 0x3, it));
  printf("two waves, both interleaved:  R=4 %8.1f   R=8 %8.1f\n", run<4>(sink, 2, 0x33, it), run<8>(sink, 2, 0x33, it));
  return 0;
}
