// Per-wave LDS-DMA issue rate on gfx950: how many `global_load_lds_dwordx4` (1 KiB) a wave gets through per unit time when
// it keeps D of them in flight to DISTINCT LDS slots, as a function of the number of waves per CU.  (dma_rate.hip reuses two
// LDS slots per wave; this one separates "waves issuing" from "depth per wave" cleanly.)  Source: a 256 KiB region (L2 hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
template <int D>
__global__ __launch_bounds__(1024) void k(const char* src, int iters, int slots, int* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* my = lds + wave * slots * 1024;
  size_t off = ((size_t)(blockIdx.x * 16 + wave) * 4096) & 0x3ffff;
  for (int it = 0; it < iters; it++) {
    __builtin_amdgcn_global_load_lds(GP(src + off + lane * 16), LP(my + (it % slots) * 1024), 16, 0, 0);
    if (D == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (D == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (D == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (D == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    off = (off + 1024 * 37) & 0x3ffff;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[threadIdx.x] == 0x77 && iters < 0) sink[0] = 1;
}
template <int D>
void run(const char* src, int blocks, int waves, int* sink) {
  const int iters = 2000, slots = 8;
  const size_t ldsb = (size_t)waves * slots * 1024;
  (void)hipFuncSetAttribute((const void*)k<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<D><<<blocks, waves * 64, ldsb>>>(src, iters, slots, sink);
  (void)hipEventRecord(e0);
  k<D><<<blocks, waves * 64, ldsb>>>(src, iters, slots, sink);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * waves * iters * 1024;
  printf("blocks %4d waves/block %2d depth %2d: %6.1f GB/s per CU (%5.2f TB/s chip), %6.1f ns per instruction per wave\n", blocks, waves, D,
         bytes / (ms * 1e-3) / 1e9 / (blocks < 256 ? blocks : 256), bytes / (ms * 1e-3) / 1e12, ms * 1e6 / iters / (blocks > 256 ? 2.0 : 1.0));
}
int main() {
  char* src; (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 1, 1 << 20);
  int* sink; (void)hipMalloc(&sink, 64);
  for (int waves : {1, 2, 4, 8, 16}) {
    if (waves == 1) { run<0>(src, 256, 1, sink); run<1>(src, 256, 1, sink); run<3>(src, 256, 1, sink); run<7>(src, 256, 1, sink); }
    if (waves == 2) { run<0>(src, 256, 2, sink); run<3>(src, 256, 2, sink); run<7>(src, 256, 2, sink); }
    if (waves == 4) { run<0>(src, 256, 4, sink); run<1>(src, 256, 4, sink); run<3>(src, 256, 4, sink); run<7>(src, 256, 4, sink); }
    if (waves == 8) { run<0>(src, 256, 8, sink); run<1>(src, 256, 8, sink); run<3>(src, 256, 8, sink); run<7>(src, 256, 8, sink); }
    if (waves == 16) { run<0>(src, 256, 16, sink); run<1>(src, 256, 16, sink); run<3>(src, 256, 16, sink); run<7>(src, 256, 16, sink); }
  }
  run<3>(src, 64, 8, sink); run<3>(src, 128, 8, sink); run<3>(src, 196, 8, sink);   // fewer CUs busy: does the per-CU rate go up?
  return 0;
}
