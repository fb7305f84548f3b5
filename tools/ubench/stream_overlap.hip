// Do small kernels on two HIP streams overlap?  N dependent launches of a 100-block kernel whose blocks spin ~4 us,
// on one stream, and N on each of two / four streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_spin(int* p, int ticks) {
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < ticks) {}
  if (p && threadIdx.x == 100000) p[0] = 1;
}
int main() {
  int* p; (void)hipMalloc(&p, 64);
  for (int ns : {1, 2, 4}) for (int grid : {100, 400}) {
    std::vector<hipStream_t> st(ns);
    for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 300;
    for (int i = 0; i < 20; i++) for (auto s : st) k_spin<<<grid, 256, 40 * 1024, s>>>(p, 9600);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < N; i++) for (auto s : st) k_spin<<<grid, 256, 40 * 1024, s>>>(p, 9600);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%d stream(s), grid %3d: %.2f us per launch-slot (%d launches per stream, %d total)\n", ns, grid, ms * 1000 / N, N, N * ns);
  }
  return 0;
}
