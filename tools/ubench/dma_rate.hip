// LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) latency and throughput per CU on gfx950,
// as a function of waves per CU, instructions in flight per wave, and where the data comes from
// (hot = every wave re-reads one 64 KiB region: L2 hits; stream = unique addresses: HBM / MALL).
// For comparison the same with global_load_dwordx4 into registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
template <int D, bool TO_LDS>
__global__ __launch_bounds__(1024) void k(const char* src, size_t span, int iters, long long* out, int* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * (blockDim.x >> 6) + wave;            // global wave id
  char* my = lds + wave * 2048;
  size_t off = ((size_t)gw * 1024 * 37) % span;
  int4 acc = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const char* p = src + off + lane * 16;
    if (TO_LDS) __builtin_amdgcn_global_load_lds(GP(p), LP(my + (it & 1) * 1024), 16, 0, 0);
    else { typedef int v4i __attribute__((ext_vector_type(4))); v4i v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
    if (D == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (D == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (D == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (D == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (D == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    off += (size_t)gridDim.x * (blockDim.x >> 6) * 1024;
    off %= span;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc.x == 0x1234567) sink[0] = 1;
}
template <int D, bool TO_LDS>
void run(const char* name, const char* src, size_t span, int blocks, int threads, long long* dout, int* sink) {
  const int iters = 400;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<D, TO_LDS><<<blocks, threads, 65536>>>(src, span, iters, dout, sink);
  (void)hipEventRecord(e0);
  k<D, TO_LDS><<<blocks, threads, 65536>>>(src, span, iters, dout, sink);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); (void)hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost);
  double ticks = 0; for (auto v : h) ticks += v; ticks /= blocks;
  const double bytes = (double)blocks * (threads / 64) * iters * 1024;
  printf("%-22s blocks %4d waves/blk %2d depth %2d: %7.1f ticks/instr/wave  kernel %.1f us -> %.2f TB/s chip, %.1f GB/s per CU\n",
         name, blocks, threads / 64, D, ticks / iters, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256);
}
int main() {
  const size_t big = (size_t)4 << 30;
  char* src; (void)hipMalloc(&src, big); (void)hipMemset(src, 1, big);
  long long* dout; int* sink; (void)hipMalloc(&dout, 8 * 4096); (void)hipMalloc(&sink, 64);
  (void)hipFuncSetAttribute((const void*)k<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int hot = 1; hot >= 0; hot--) {
    const size_t span = hot ? (64 << 10) : big - (1 << 20);
    const char* nm = hot ? "lds-dma hot(L2)" : "lds-dma stream(HBM)";
    run<0, true>(nm, src, span, 256, 64, dout, sink);
    run<0, true>(nm, src, span, 256, 1024, dout, sink);
    run<1, true>(nm, src, span, 256, 1024, dout, sink);
    run<2, true>(nm, src, span, 256, 1024, dout, sink);
    run<4, true>(nm, src, span, 256, 1024, dout, sink);
    run<8, true>(nm, src, span, 256, 1024, dout, sink);
    run<2, true>(nm, src, span, 512, 1024, dout, sink);
    run<4, true>(nm, src, span, 512, 1024, dout, sink);
    run<8, true>(nm, src, span, 512, 1024, dout, sink);
    run<8, true>(nm, src, span, 256, 256, dout, sink);
    run<15, true>(nm, src, span, 256, 256, dout, sink);
    const char* nr = hot ? "vgpr-load hot(L2)" : "vgpr-load stream(HBM)";
    run<0, false>(nr, src, span, 256, 64, dout, sink);
    run<2, false>(nr, src, span, 512, 1024, dout, sink);
    run<8, false>(nr, src, span, 512, 1024, dout, sink);
    run<8, false>(nr, src, span, 256, 256, dout, sink);
  }
  return 0;
}
