// How fast can a wave feed v_mfma_i32_32x32x32_i8 when A comes global -> registers (an L2-resident weight array) and B comes from LDS
// (ds_read_b128), as in conv_c3.hip's K loop -- and what does the register tile decide?
//   form 8: 8 waves per block (two per SIMD), wave tile TR x TC = 2 x 4 MFMA tiles (conv_c3's 256-channel blocks: <= 256 registers)
//   form 4: 4 waves per block (ONE per SIMD), wave tile 4 x 4 (256 accumulator registers: needs the AccVGPR half of the 512-entry file)
// Per K step (one 64-byte slab): two K halves, each TR A fragments (global, one step ahead in a second register set) and TC B fragments
// (LDS, the next half's reads issued before this half's MFMAs), TR * TC MFMAs.  mode 0: loads as described; 1: no loads in the loop
// (operands loop-invariant: the matrix pipe's own ceiling for this instruction stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

template <int NW, int TR, int TC, int MODE, bool BAR = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 1 : 2) void feed(const int8_t* __restrict__ w, int* sink, int steps, int rows_total) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int WR = (NW == 4) ? 2 : 4;                     // waves along the rows
  const int wr = wave % WR, wc = wave / WR;
  for (int i = tid; i < 24 * 1024 / 4; i += NW * 64) reinterpret_cast<int*>(lds)[i] = i * 2654435761u;
  __syncthreads();
  i32x16 acc[TR][TC];
#pragma unroll
  for (int i = 0; i < TR; i++)
#pragma unroll
    for (int j = 0; j < TC; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
  const int8_t* wa = w + ((size_t)(wr * TR * 32 + (lane & 31))) * 64 + half * 16;       // + step * rows_total * 64 + i * 2048 + ks * 32
  const int boff = ((wc * TC * 32 + (lane & 31)) * 64 + half * 16) & (24 * 1024 - 1);
  i32x4 a_cur[2][TR], a_nxt[2][TR], b_cur[TC], b_nxt[TC];
  auto load_a = [&](i32x4 (&f)[2][TR], int step) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < TR; i++) f[ks][i] = *reinterpret_cast<const i32x4*>(wa + (size_t)step * rows_total * 64 + i * 2048 + ks * 32);
  };
  auto load_b = [&](i32x4 (&f)[TC], int step, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TC; j++) f[j] = *reinterpret_cast<const i32x4*>(lds + ((boff + j * 2048 + (step & 7) * 1040 + ks * 32) & (24 * 1024 - 16)));
  };
  load_a(a_cur, 0);
  load_b(b_cur, 0, 0);
  for (int s = 0; s < steps; s++) {
    if (MODE == 0) load_a(a_nxt, s + 1 < steps ? s + 1 : s);
    if (BAR) __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks the loads to their first use: no prefetch left)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      if (MODE == 0) load_b(b_nxt, ks == 0 ? s : s + 1, ks ^ 1);
      if (BAR) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TR; i++)
#pragma unroll
        for (int j = 0; j < TC; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_cur[ks][i], b_cur[j], acc[i][j], 0, 0, 0);
      if (BAR) __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < TC; j++) b_cur[j] = b_nxt[j];
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < TR; i++) a_cur[ks][i] = a_nxt[ks][i];
    }
  }
  int sum = 0;
#pragma unroll
  for (int i = 0; i < TR; i++)
#pragma unroll
    for (int j = 0; j < TC; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) sum += acc[i][j][r];
  if (sum == 0x12345678) sink[0] = sum;
}

template <int NW, int TR, int TC, int MODE, bool BAR = false>
static void run(const int8_t* w, int* sink, int steps, int rows_total, const char* name) {
  auto fn = feed<NW, TR, TC, MODE, BAR>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t lds = 24 * 1024 + (NW == 4 ? 64 * 1024 : 24 * 1024);        // (one block per CU in both forms)
  fn<<<256, NW * 64, lds>>>(w, sink, steps, rows_total);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; r++) fn<<<256, NW * 64, lds>>>(w, sink, steps, rows_total);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 5;
  const double mfma = 256.0 * NW * steps * 2 * TR * TC;
  printf("%-34s %8.1f us  %7.1f TOP/s  (%.0f MFMAs per wave; per SIMD %.1f cycles per MFMA at 2.4 GHz)\n", name, us, mfma * 65536.0 / us * 1e-6,
         (double)steps * 2 * TR * TC, us * 2400.0 / (steps * 2.0 * TR * TC * (NW / 4)));
}

int main() {
  const int steps = 288, rows_total = 256;
  int8_t* w; int* sink;
  (void)hipMalloc(&w, (size_t)steps * rows_total * 64 + 4096); (void)hipMemset(w, 1, (size_t)steps * rows_total * 64 + 4096);
  (void)hipMalloc(&sink, 64);
  run<8, 2, 4, 1>(w, sink, steps, rows_total, "8 waves, 2x4 tiles, no loads");
  run<8, 2, 4, 0>(w, sink, steps, rows_total, "8 waves, 2x4 tiles, A L2 / B LDS");
  run<4, 4, 4, 1>(w, sink, steps, rows_total, "4 waves, 4x4 tiles, no loads");
  run<4, 4, 4, 0>(w, sink, steps, rows_total, "4 waves, 4x4 tiles, A L2 / B LDS");
  run<4, 2, 4, 0>(w, sink, steps, rows_total, "4 waves, 2x4 tiles, A L2 / B LDS");
  run<8, 2, 4, 0, true>(w, sink, steps, rows_total, "8 waves, 2x4, loads pinned ahead");
  run<4, 2, 4, 0, true>(w, sink, steps, rows_total, "4 waves, 2x4, loads pinned ahead");
  run<4, 4, 4, 0, true>(w, sink, steps, rows_total, "4 waves, 4x4, loads pinned ahead");
  run<8, 2, 2, 0>(w, sink, steps, rows_total, "8 waves, 2x2 tiles, A L2 / B LDS");
  return 0;
}
