// Workgroup dispatch rate: wall-clock (100 MHz s_memrealtime) spread between the first and the last block START
// of one launch, as a function of dynamic LDS per block, registers per lane and block size.  Blocks live ~3 us.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NV>
__global__ void k(long long* out, int* sink, int spin) {
  extern __shared__ int s[];
  const long long w0 = wall_clock64();
  int v[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = threadIdx.x * i + spin;
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) {
#pragma unroll
    for (int i = 0; i < NV; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(v[i]));
  }
  int acc = 0;
#pragma unroll
  for (int i = 0; i < NV; i++) acc += v[i];
  if (acc == 0x12345) sink[0] = s[threadIdx.x];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = w0; out[blockIdx.x * 2 + 1] = wall_clock64(); }
}
template <int NV>
void run(int grid, int threads, int lds, long long* d, int* sink) {
  (void)hipFuncSetAttribute((const void*)k<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<NV><<<grid, threads, lds>>>(d, sink, 7000); k<NV><<<grid, threads, lds>>>(d, sink, 7000); (void)hipDeviceSynchronize();
  std::vector<long long> h(grid * 2); (void)hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
  long long s0 = h[0], s1 = h[0], e1 = h[1];
  for (int i = 0; i < grid; i++) { s0 = std::min(s0, h[2 * i]); s1 = std::max(s1, h[2 * i]); e1 = std::max(e1, h[2 * i + 1]); }
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, (const void*)k<NV>);
  printf("grid %5d x %4d thr, LDS %3d KiB, %3d VGPR: first->last block start %6.2f us, first start->last end %6.2f us\n",
         grid, threads, lds / 1024, fa.numRegs, (s1 - s0) / 100.0, (e1 - s0) / 100.0);
}
int main() {
  long long* d; int* sink; (void)hipMalloc(&d, 16 * 8192); (void)hipMalloc(&sink, 64);
  for (int grid : {400, 1568}) {
    run<8>(grid, 256, 0, d, sink);
    run<8>(grid, 256, 16 * 1024, d, sink);
    run<8>(grid, 256, 37 * 1024, d, sink);
    run<8>(grid, 256, 52 * 1024, d, sink);
    run<100>(grid, 256, 0, d, sink);
    run<100>(grid, 256, 52 * 1024, d, sink);
    run<8>(grid, 512, 52 * 1024, d, sink);
    run<8>(grid, 1024, 52 * 1024, d, sink);
    run<8>(grid, 64, 0, d, sink);
  }
  return 0;
}
