// Which CUs does a stream created with hipExtStreamCreateWithCUMask really use?  Every block records its XCC id and HW id; the host
// counts distinct (xcc, se, sh, cu) per mask, and times a fixed amount of spinning work (a restricted stream takes longer).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));       // XCC_ID
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
  }
}
static void run(const char* name, hipStream_t s, unsigned* d, int nblk) {
  std::vector<unsigned> h(2 * nblk);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<<<nblk, 256, 0, s>>>(d, 200); hipStreamSynchronize(s);
  hipEventRecord(e0, s);
  probe<<<nblk, 256, 0, s>>>(d, 200);            // 2 us per block
  hipEventRecord(e1, s); hipStreamSynchronize(s);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::set<unsigned long long> cus; std::set<unsigned> xccs;
  for (int b = 0; b < nblk; b++) {
    const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
  }
  printf("%-28s %5d blocks x 2 us: %8.1f us, %3zu distinct CUs on %zu XCDs:", name, nblk, ms * 1e3, cus.size(), xccs.size());
  for (unsigned x : xccs) printf(" %u", x);
  printf("\n");
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32, nblk = 8192;
  unsigned* d; hipMalloc(&d, 2 * nblk * 4);
  hipStream_t plain; hipStreamCreate(&plain);
  run("plain stream", plain, d, nblk);
  struct { const char* name; int mod, lo, hi; } masks[] = {
    {"bits i%8 in {0,1}", 8, 0, 1}, {"bits i%8 in {0}", 8, 0, 0}, {"bits i%8 in {0..3}", 8, 0, 3}, {"bits i%4 == 0", 4, 0, 0},
    {"bits [0,64)", 0, 0, 63}, {"bits [0,32)", 0, 0, 31} };
  for (auto& m : masks) {
    std::vector<unsigned> w(words, 0);
    for (int i = 0; i < ncu; i++) { const int k = m.mod ? i % m.mod : i; if (k >= m.lo && k <= m.hi) w[i / 32] |= 1u << (i % 32); }
    hipStream_t s; const hipError_t e = hipExtStreamCreateWithCUMask(&s, words, w.data());
    if (e != hipSuccess) { printf("%s: create failed %d\n", m.name, (int)e); continue; }
    unsigned back[16] = {0}; hipExtStreamGetCUMask(s, 16, back);
    int pc = 0; for (unsigned x : back) pc += __builtin_popcount(x);
    char nm[64]; snprintf(nm, sizeof nm, "%s (%d bits)", m.name, pc);
    run(nm, s, d, nblk);
  }
  return 0;
}
