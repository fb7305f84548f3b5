// Does a line that kernel A pulled into an XCD's L2 survive into the dependent kernel B (same stream)?  gfx950, 8 XCDs, 4 MiB of L2 each.
// A: grid of 8 * NB blocks; block b (XCD b % 8) reads slice b / 8 of a 2 MiB "weight" range -- every XCD reads the whole range.
// B: the same grid; every wave times a dependent chain of 16-byte loads over its slice (latency) and then a streaming read (bandwidth):
//    (1) on the range A touched, (2) on a range nobody touched since a 512 MiB sweep (memory-side cache evicted too),
//    (3) on a range another kernel touched on the OTHER XCD assignment (slices rotated by one XCD: resident in a different L2).
// Also: B on range (1) with a kernel in between that WRITES 8 MiB elsewhere (does an unrelated launch evict / invalidate it?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long wall() { long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

__global__ void touch(const char* base, size_t bytes, int nb, int rot, int* sink) {
  const int xcd = (blockIdx.x + rot) & 7, sl = blockIdx.x >> 3;
  const size_t per = bytes / nb;
  const char* p = base + (size_t)sl * per;
  int acc = 0;
  for (size_t o = threadIdx.x * 16; o < per; o += blockDim.x * 16) { v4i v = *reinterpret_cast<const v4i*>(p + o); acc += v[0] + v[3]; }
  if (acc == 0x7fffffff) sink[xcd] = acc;
}
__global__ void sweep(char* base, size_t bytes) {
  for (size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o < bytes; o += (size_t)gridDim.x * blockDim.x * 16) *reinterpret_cast<v4i*>(base + o) = v4i{1, 2, 3, 4};
}
__global__ void probe(const char* base, size_t bytes, int nb, long long* out, int* sink) {
  const int sl = blockIdx.x >> 3;
  const size_t per = bytes / nb;
  const char* p = base + (size_t)sl * per;
  // latency: 32 dependent 16-byte loads by lane 0's address chain (stride 4 KiB inside the slice)
  long long t0 = wall();
  size_t o = threadIdx.x * 16;
  int acc = 0;
  for (int i = 0; i < 32; i++) { v4i v = *reinterpret_cast<const v4i*>(p + o); acc += v[0]; o = (o + 4096 + (size_t)(v[1] & 0)) % per; }
  long long t1 = wall();
  // bandwidth: the whole slice once
  for (size_t q = threadIdx.x * 16; q < per; q += blockDim.x * 16) { v4i v = *reinterpret_cast<const v4i*>(p + q); acc += v[2]; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t2 = wall();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t1; }
  if (acc == 0x7fffffff) sink[0] = acc;
}
int main() {
  const size_t W = 2 << 20, BIG = (size_t)512 << 20;
  const int NB = 16, grid = 8 * NB;
  char *w, *big; long long* out; int* sink;
  (void)hipMalloc(&w, 4 * W); (void)hipMalloc(&big, BIG); (void)hipMalloc(&out, grid * 16); (void)hipMalloc(&sink, 64);
  (void)hipMemset(w, 0, 4 * W);
  auto report = [&](const char* name) {
    (void)hipDeviceSynchronize();
    std::vector<long long> h(grid * 2); (void)hipMemcpy(h.data(), out, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> lat, bw;
    for (int b = 0; b < grid; b++) { lat.push_back(h[2 * b] / 32.0 * 10.0); bw.push_back((double)(W / NB) / (h[2 * b + 1] * 10.0)); }
    std::sort(lat.begin(), lat.end()); std::sort(bw.begin(), bw.end());
    printf("%-72s latency per dependent load: median %6.0f ns (min %5.0f)   slice read: median %6.2f GB/s per block (x %d blocks = %6.2f TB/s)\n", name,
           lat[grid / 2], lat[0], bw[grid / 2], grid, bw[grid / 2] * grid / 1000.0);
  };
  for (int rep = 0; rep < 2; rep++) {
    sweep<<<2048, 256>>>(big, BIG); (void)hipDeviceSynchronize();
    probe<<<grid, 256>>>(w, W, NB, out, sink); report("cold (behind a 512 MiB sweep)");
    probe<<<grid, 256>>>(w, W, NB, out, sink); report("again, same launch shape (touched by the previous probe)");
    touch<<<grid, 256>>>(w + W, W, NB, 0, sink); probe<<<grid, 256>>>(w + W, W, NB, out, sink); report("touched by the launch in front (same XCD assignment)");
    touch<<<grid, 256>>>(w + 2 * W, W, NB, 0, sink); sweep<<<512, 256>>>(big, 8 << 20); probe<<<grid, 256>>>(w + 2 * W, W, NB, out, sink); report("touched, then an 8 MiB writer in between");
    sweep<<<2048, 256>>>(big, BIG); (void)hipDeviceSynchronize();
    touch<<<grid, 256>>>(w + 3 * W, W, NB, 0, sink); (void)hipDeviceSynchronize();
    probe<<<grid, 256>>>(w + 3 * W, W, NB, out, sink); report("touched with the host in between (hipDeviceSynchronize)");
    sweep<<<2048, 256>>>(big, BIG); (void)hipDeviceSynchronize();
    // slices shifted by one block id: block b reads what block b + 1's XCD touched
    touch<<<grid, 256>>>(w, W, NB, 0, sink); probe<<<grid, 256>>>(w + W / NB / 1 * 0, W, NB, out, sink); report("control: touched, same mapping");
  }
  return 0;
}
