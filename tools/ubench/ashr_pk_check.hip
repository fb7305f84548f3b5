// v_ashr_pk_i8_i32 / v_ashr_pk_u8_i32 (gfx950) against their scalar definition {sat(a >> n), sat(b >> n)} in d[15:0], for
// every shift 0..31 over edge values and a pseudo-random sweep; and the ReLU form used by requant_epilogue.h:
// (pack_u8(a >> (n-1), b >> (n-1)) >> 1) & 0x7f7f == {clamp(a >> n, 0, 127), clamp(b >> n, 0, 127)}.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* v, int n, unsigned long long* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = v[i], b = v[(i * 7 + 3) % n];
  unsigned long long nb = 0;
#define CHK(SH) do { \
    const unsigned gi = (unsigned short)__builtin_amdgcn_ashr_pk_i8_i32(a, b, SH), gu = (unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(a, b, SH); \
    const int sa = a >> SH, sb = b >> SH; \
    const int ia = sa < -128 ? -128 : sa > 127 ? 127 : sa, ib = sb < -128 ? -128 : sb > 127 ? 127 : sb; \
    const int ua = sa < 0 ? 0 : sa > 255 ? 255 : sa, ub = sb < 0 ? 0 : sb > 255 ? 255 : sb; \
    if (gi != (unsigned)((ia & 0xff) | ((ib & 0xff) << 8))) { nb++; if (atomicAdd(bad + 1, 1ull) < 6) printf("i8 sh %d a %d b %d got %04x want %04x\n", SH, a, b, gi, (ia & 0xff) | ((ib & 0xff) << 8)); } \
    if (gu != (unsigned)(ua | (ub << 8))) { nb++; if (atomicAdd(bad + 2, 1ull) < 6) printf("u8 sh %d a %d b %d got %04x want %04x\n", SH, a, b, gu, ua | (ub << 8)); } \
    if (SH >= 1) { const unsigned r = (((unsigned short)__builtin_amdgcn_ashr_pk_u8_i32(a, b, SH >= 1 ? SH - 1 : 0)) >> 1) & 0x7f7fu; \
      const int ra = sa < 0 ? 0 : sa > 127 ? 127 : sa, rb = sb < 0 ? 0 : sb > 127 ? 127 : sb; if (r != (unsigned)(ra | (rb << 8))) { nb++; if (atomicAdd(bad + 3, 1ull) < 6) printf("relu sh %d a %d b %d got %04x want %04x\n", SH, a, b, r, ra | (rb << 8)); } } \
  } while (0)
  CHK(0); CHK(1); CHK(2); CHK(3); CHK(4); CHK(7); CHK(8); CHK(14); CHK(15); CHK(16); CHK(20); CHK(24); CHK(30); CHK(31);
  if (nb) atomicAdd(bad, nb);
}
int main() {
  const int n = 1 << 22;
  int* h = new int[n];
  unsigned x = 12345;
  for (int i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; const int sh = (x >> 27); h[i] = (int)(x ^ (x << 13)) >> (sh & 31); }
  const int edge[] = {0, 1, -1, 127, 128, 255, 256, -128, -129, 1023, 1024, -1024, 2147483647, (int)0x80000000, 32767, -32768, 65535, 8, 7, -8, -7, 2039, 2040, 2047, 2048};
  for (unsigned i = 0; i < sizeof(edge) / 4; i++) for (int j = 0; j < 64; j++) h[(i * 64 + j) * 61 % n] = edge[i] + (j - 32);
  int* d; unsigned long long* bad; (void)hipMalloc(&d, n * 4); (void)hipMalloc(&bad, 64); (void)hipMemset(bad, 0, 64);
  (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d, n, bad);
  unsigned long long nb = 1; (void)hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
  printf("ashr_pk_i8_i32 / ashr_pk_u8_i32 / ReLU form: %d value pairs x 14 shifts, mismatches: %llu\n", n, nb);
  return nb != 0;
}
