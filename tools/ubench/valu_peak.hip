// VALU throughput per SIMD at 1..8 waves/SIMD: grid of (256 * k) blocks of 1024 threads keeps k blocks per CU
// resident (k = 1, 2); each wave runs a straight-line chain of independent v_lshl_add_u32 / v_mad_i64_i32.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(1024) void k(long long* out, int* sink, int seed, int iters) {
  int a[8]; long long q[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; q[i] = a[i]; }
  int b = seed * 3 + 1, c = seed + 7;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (OP == 0) {
      asm volatile(
        "v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
        "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
        "v_lshl_add_u32 %0, %0, %8, %9\n v_lshl_add_u32 %1, %1, %8, %9\n v_lshl_add_u32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, %8, %9\n"
        "v_lshl_add_u32 %4, %4, %8, %9\n v_lshl_add_u32 %5, %5, %8, %9\n v_lshl_add_u32 %6, %6, %8, %9\n v_lshl_add_u32 %7, %7, %8, %9\n"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
    } else {
      asm volatile(
        "v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n"
        "v_mad_i64_i32 %4, vcc, %8, %9, %4\n v_mad_i64_i32 %5, vcc, %8, %9, %5\n v_mad_i64_i32 %6, vcc, %8, %9, %6\n v_mad_i64_i32 %7, vcc, %8, %9, %7\n"
        "v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n"
        "v_mad_i64_i32 %4, vcc, %8, %9, %4\n v_mad_i64_i32 %5, vcc, %8, %9, %5\n v_mad_i64_i32 %6, vcc, %8, %9, %6\n v_mad_i64_i32 %7, vcc, %8, %9, %7\n"
        : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(b), "v"(c) : "vcc");
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  long long s = 0; for (int i = 0; i < 8; i++) s += a[i] + q[i];
  if (s == 0x1234567) sink[0] = 1;
}
template <int OP>
void run(const char* name, long long* d, int* sink) {
  for (int nthr : {64, 256, 512, 1024}) for (int nblk : {1, 256, 512}) {
    if (nblk == 512 && nthr != 1024) continue;
    const int iters = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<nblk, nthr>>>(d, sink, 3, iters); (void)hipEventRecord(e0); k<OP><<<nblk, nthr>>>(d, sink, 3, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    double per = (double)h / (iters * 16.0);
    printf("   [wall %.1f us for %d instr per wave -> %.2f ns per instr per wave] ", ms * 1e3, iters * 16, ms * 1e6 / (iters * 16.0));
    double wps = nthr / 256.0 * (nblk == 512 ? 2 : 1);
    printf("%-16s blocks %4d threads %4d (%.2f waves/SIMD): %6.2f ticks per instr per wave -> %5.2f ticks per SIMD-instr\n", name, nblk, nthr, wps, per, per / (wps < 1 ? 1 : wps));
  }
}
int main() {
  long long* d; int* sink; (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 64);
  run<0>("v_lshl_add_u32", d, sink); run<1>("v_mad_i64_i32", d, sink);
}
