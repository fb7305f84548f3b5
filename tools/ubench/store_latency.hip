// global_store_dwordx4 issue -> vmcnt(0) latency on gfx950: one wave alone, and all CUs busy doing the same.
// pattern 0: 64 lanes x 16 B contiguous (1 KiB); pattern 1: the conv epilogue's (32 pixels x 2 x 16 B at a
// 256-byte pitch: 32 lines touched, 32 B each); `same` = every iteration hits the same lines (L2 write hits).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int PATTERN, bool SAME, int MOD>
__global__ void k(char* dst, size_t span, int iters, long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i v = {1, 2, 3, 4};
  long long tot = 0;
  for (int it = 0; it < iters; it++) {
    size_t base = (gw * 8192 + (SAME ? 0 : (size_t)it * gridDim.x * (blockDim.x >> 6) * 8192)) % span;
    char* p = dst + base + (PATTERN == 0 ? lane * 16 : (lane & 31) * 256 + (lane >> 5) * 16);
    long long t0 = __builtin_readcyclecounter();
    if (MOD == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if (MOD == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    tot += t1 - t0;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = tot;
}
template <int PATTERN, bool SAME, int MOD>
void run(const char* name, char* dst, size_t span, int blocks, int threads, long long* dout) {
  const int iters = 200;
  k<PATTERN, SAME, MOD><<<blocks, threads>>>(dst, span, iters, dout);
  k<PATTERN, SAME, MOD><<<blocks, threads>>>(dst, span, iters, dout);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost);
  double t = 0; for (auto v : h) t += v; t /= blocks;
  printf("%-44s blocks %4d waves/blk %2d: %8.1f ticks store->ack\n", name, blocks, threads / 64, t / iters);
}
int main() {
  const size_t big = (size_t)2 << 30;
  char* dst; (void)hipMalloc(&dst, big + (1 << 20)); (void)hipMemset(dst, 0, big);
  long long* dout; (void)hipMalloc(&dout, 8 * 4096);
  run<0, true, 0>("contiguous 1 KiB, same lines", dst, big, 1, 64, dout);
  run<0, false, 0>("contiguous 1 KiB, fresh lines", dst, big, 1, 64, dout);
  run<1, true, 0>("32 lines x 32 B, same lines", dst, big, 1, 64, dout);
  run<1, false, 0>("32 lines x 32 B, fresh lines", dst, big, 1, 64, dout);
  run<1, false, 1>("32 lines x 32 B, fresh lines, nt", dst, big, 1, 64, dout);
  run<1, false, 2>("32 lines x 32 B, fresh lines, sc0 sc1", dst, big, 1, 64, dout);
  run<0, false, 0>("contiguous 1 KiB, fresh lines", dst, big, 512, 1024, dout);
  run<1, true, 0>("32 lines x 32 B, same lines", dst, big, 512, 1024, dout);
  run<1, false, 0>("32 lines x 32 B, fresh lines", dst, big, 512, 1024, dout);
  run<1, false, 1>("32 lines x 32 B, fresh lines, nt", dst, big, 512, 1024, dout);
  run<1, false, 2>("32 lines x 32 B, fresh lines, sc0 sc1", dst, big, 512, 1024, dout);
  run<1, false, 0>("32 lines x 32 B, fresh lines", dst, big, 256, 256, dout);
  return 0;
}
