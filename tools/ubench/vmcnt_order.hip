// Does a younger global STORE ever retire (decrement vmcnt) before an older, slower global LOAD on gfx950?
// Each wave: slow load (cold 2 GiB buffer, page-strided), then a store to a hot line, then s_waitcnt vmcnt(1)
// and an immediate copy of the load's destination register.  If vmcnt were decremented out of order, the copy
// would sometimes hold the register's old content (the poison written before the load).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* cold, int* hot, int* bad, size_t n_cold, int rounds, int seed) {
  const int lane = threadIdx.x & 63;
  unsigned h = (blockIdx.x * 977u + (threadIdx.x >> 6) * 131u + seed) * 2654435761u;
  int nbad = 0;
  for (int r = 0; r < rounds; r++) {
    h = h * 1664525u + 1013904223u;
    const size_t idx = ((size_t)(h >> 4) * 1024 + lane * 1031) % n_cold;     // scattered lines, far apart
    const int* src = cold + idx;
    int* dst = hot + ((blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane);
    int v = 0x7eadbeef, copy;
    asm volatile(
      "v_mov_b32 %0, 0x7eadbeef\n"
      "global_load_dword %0, %2, off\n"
      "global_store_dword %3, %4, off\n"
      "s_waitcnt vmcnt(1)\n"
      "v_mov_b32 %1, %0\n"
      "s_waitcnt vmcnt(0)\n"
      : "=&v"(v), "=&v"(copy) : "v"(src), "v"(dst), "v"(r) : "memory");
    if (copy != v) nbad++;
  }
  if (nbad) atomicAdd(bad, nbad);
}
int main() {
  const size_t n_cold = (size_t)512 << 20;     // 2 GiB of ints
  int *cold, *hot, *bad; (void)hipMalloc(&cold, n_cold * 4); (void)hipMalloc(&hot, 1 << 24); (void)hipMalloc(&bad, 4);
  (void)hipMemset(cold, 1, n_cold * 4); (void)hipMemset(bad, 0, 4);
  for (int it = 0; it < 5; it++) { k<<<2048, 256>>>(cold, hot, bad, n_cold, 2000, it); (void)hipDeviceSynchronize(); }
  int h; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("load-then-store, vmcnt(1): %d early reads out of %lld\n", h, 5LL * 2048 * 256 * 2000);
  return 0;
}
