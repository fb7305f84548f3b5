// issue rate of v_ashr_pk_i8_i32 against v_lshl_add_u32 and v_med3_i32 / v_perm_b32 (one wave per SIMD and four)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(1024) void k(int* sink, int iters, int seed) {
  int v[8]; for (int i = 0; i < 8; i++) v[i] = seed + i + threadIdx.x;
  int p = seed * 3 + 1, q = seed + 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (OP == 0) asm volatile("v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
                                "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
                                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(p), "v"(q));
      if (OP == 1) asm volatile("v_ashr_pk_i8_i32 %0, %0, %8, 3\n v_ashr_pk_i8_i32 %1, %1, %8, 3\n v_ashr_pk_i8_i32 %2, %2, %8, 3\n v_ashr_pk_i8_i32 %3, %3, %8, 3\n"
                                "v_ashr_pk_i8_i32 %4, %4, %8, 3\n v_ashr_pk_i8_i32 %5, %5, %8, 3\n v_ashr_pk_i8_i32 %6, %6, %8, 3\n v_ashr_pk_i8_i32 %7, %7, %8, 3\n"
                                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(p), "v"(q));
      if (OP == 2) asm volatile("v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n"
                                "v_med3_i32 %4, %4, %8, %9\n v_med3_i32 %5, %5, %8, %9\n v_med3_i32 %6, %6, %8, %9\n v_med3_i32 %7, %7, %8, %9\n"
                                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(p), "v"(q));
      if (OP == 3) asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                                "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(p), "v"(q));
      if (OP == 4) asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3\n"
                                "v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3\n"
                                : "+v"(*(long long*)&v[0]), "+v"(*(long long*)&v[2]), "+v"(*(long long*)&v[4]), "+v"(*(long long*)&v[6]) : "v"(p), "v"(q) : "vcc");
    }
  }
  int s = 0; for (int i = 0; i < 8; i++) s += v[i];
  if (s == 0x1234567) sink[0] = 1;
}
template <int OP> static void run(const char* name, int* sink) {
  const int it = 4096;
  for (int wps : {1, 4}) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<256, 256 * wps>>>(sink, it, 3); (void)hipEventRecord(e0); k<OP><<<256, 256 * wps>>>(sink, it, 3); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %d wave(s)/SIMD: %.2f ns per instruction per SIMD\n", name, wps, ms * 1e6 / (32.0 * it * wps));
  }
}
int main() { int* sink; (void)hipMalloc(&sink, 64);
  run<0>("v_lshl_add_u32", sink); run<1>("v_ashr_pk_i8_i32", sink); run<2>("v_med3_i32", sink); run<3>("v_perm_b32", sink); run<4>("v_mad_i64_i32", sink); return 0; }
