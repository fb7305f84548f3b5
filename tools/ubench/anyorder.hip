// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (hip_ext.h notes the flag as
// unsupported on GFX9 for the module-launch form.)  Two spin kernels of ~T us on 64 blocks each, 200 pairs, with and without
// the flag on the second kernel of every pair.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) *sink = 1;
}
int main() {
  int* sink; hipMalloc(&sink, 4);
  hipStream_t s; hipStreamCreate(&s);
  for (int ticks : {500, 1000, 2000}) {      // 100 MHz: 5, 10, 20 us
    for (int flag = 0; flag < 2; flag++) {
      for (int rep = 0; rep < 2; rep++) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; i++) {
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, (long long)ticks, sink);
          if (flag) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (long long)ticks, sink);
          else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, (long long)ticks, sink);
        }
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
        if (rep) printf("spin %2d us x 2 kernels, any-order flag on the second: %d -> %.2f us per pair\n", ticks / 100, flag, us);
      }
    }
  }
  return 0;
}
