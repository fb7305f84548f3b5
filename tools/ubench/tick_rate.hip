// s_memtime (clock64 / __builtin_readcyclecounter) tick frequency vs the 100 MHz wall clock, and the
// shader clock implied by a dependent v_add chain (1 instruction per 4 cycles at most).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out) {
  long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  int a = threadIdx.x;
  for (int i = 0; i < 200000; i++) asm volatile("v_add_u32 %0, %0, %0\nv_add_u32 %0, %0, %0\nv_add_u32 %0, %0, %0\nv_add_u32 %0, %0, %0" : "+v"(a));
  long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = a; }
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[3];
  for (int nt : {64, 256, 1024}) {
    k<<<1, nt>>>(d); k<<<1, nt>>>(d); hipDeviceSynchronize(); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    double us = h[0] / 100.0;
    printf("threads %4d: wall %.1f us, ticks %lld -> tick %.1f MHz; 800k dependent v_add -> %.2f ns each (%.2f ticks)\n", nt, us, h[1], h[1] / us, us * 1000 / 800000, h[1] / 800000.0);
  }
}
