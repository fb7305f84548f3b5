// Back-to-back dependent launches of a (nearly) empty kernel in one stream: the per-launch cost as a function of
// grid size, kernel-argument bytes and dynamic LDS -- the fixed cost under every conv layer.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[170]; };     // ~680 bytes, like ConvArgs
__global__ void k_small(int* p) { if (p && threadIdx.x == 1000000) p[0] = 1; }
__global__ void k_big(Big b, int* p) { if (p && b.v[threadIdx.x & 127] == 123456789) p[0] = 1; }
__global__ void k_lds(Big b, int* p) { extern __shared__ int s[]; if (p && b.v[threadIdx.x & 127] == 123456789) p[0] = s[threadIdx.x]; }
// a block that lives ~4 us (like a conv block): spin on the clock
__global__ void k_spin(Big b, int* p, int ticks) {
  extern __shared__ int s[];
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks) {}
  if (p && b.v[threadIdx.x & 127] == 123456789) p[0] = s[threadIdx.x];
}
template <typename F> float timeit(F f, int n) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; i++) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < n; i++) f(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1000 / n;
}
int main() {
  int* p; (void)hipMalloc(&p, 64); Big b{};
  (void)hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int grid : {16, 400, 1568, 6272}) {
    printf("grid %5d x 256 thr: small args %.2f us | 680 B args %.2f us | + 52 KiB LDS %.2f us | + blocks spinning 4000 ticks %.2f us | 9600 ticks %.2f us\n", grid,
           timeit([&] { k_small<<<grid, 256>>>(p); }, 2000), timeit([&] { k_big<<<grid, 256>>>(b, p); }, 2000),
           timeit([&] { k_lds<<<grid, 256, 52 * 1024>>>(b, p); }, 2000), timeit([&] { k_spin<<<grid, 256, 52 * 1024>>>(b, p, 4000); }, 1000),
           timeit([&] { k_spin<<<grid, 256, 52 * 1024>>>(b, p, 9600); }, 1000));
  }
  printf("grid   400 x 512 thr + 52 KiB LDS: %.2f us ; 1024 thr: %.2f us\n", timeit([&] { k_lds<<<400, 512, 52 * 1024>>>(b, p); }, 2000), timeit([&] { k_lds<<<400, 1024, 52 * 1024>>>(b, p); }, 2000));
  return 0;
}
