// conv_bfirst.hip -- the FIRST bottleneck of a stage (ResNet-50 rows 11-14, 24-27, 43-46): projection shortcut S (1x1 / stride 2,
// Cin -> Cout) and reduce A (1x1, Cin -> M) of the same 2H x 2W input, 3x3 / stride 2 / pad 1 (M -> M, output H x W), expand (1x1,
// M -> Cout) + residual from S + ReLU -- four table rows (pe.cl:144-203 four times, feature_writer.cl:119-122 once) in ONE launch
// of independent row bands, Cin = 2 M, Cout = 4 M (gfx950).  Same idea as conv_bband.hip (no exchange between blocks: a band
// recomputes the reduce for the one input row it shares with its neighbour), plus the shortcut:
// A block owns R OUTPUT rows x the full width of one image:
//   phase 0  reduce over the 2 R + 1 input rows the 3x3 reads, in NG groups of TG column tiles (the input map has four times the
//            output's pixels: one group's accumulators at a time; the reduce's weights are re-streamed per group, the input
//            streams through the chunk ring exactly once), each group requantised into the halo tile in LDS;
//   phase 1  3x3 / stride 2: tap (dh, dw) of output pixel (r, c) = halo pixel (2 r + dh, 2 c + dw); requantised into the expand's
//            B tile in LDS;
//   phase 2  Cout / M passes: the SHORTCUT tile of the pass's channels -- K = Cin over the band's even input pixels, gathered once
//            into LDS by DMA (the chunk ring's memory) -- requantised to the int8 values row S would store, then the expand tile,
//            requantised with that as its residual.  S's map is written only for keep_all (tf2_net_read_layer).
// DUAL: S and A are two-window layers (both read the stage input): A keeps two accumulator sets over the one input stream, S sweeps
// its LDS-resident operand window by window with the Horner shift in between.  The pass's header rows (S | E) are double-buffered.
// Weight tiles, header rows and the requantisation are the packed image's and requant_epilogue.h's: bit-identical to the separate
// launches (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void bf_wait_vmcnt() {
  static_assert(N == 2 || N == 4 || N == 8, "prepared immediates");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <int T, int N, class F>
__device__ __forceinline__ void bf_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); bf_static_for<T + 1, N>(fn); }
}

// LDS-DMA as inline assembly (see conv_bband.hip bb_dma16): invisible to the compiler's wait-count pass, every wait is written out
__device__ __forceinline__ void bf_dma16(const int8_t* src, int8_t* lds_dst) {
  const unsigned l = (unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(l) : "memory", "m0");
}

// M: channels of the intermediates; NW waves = WM x WN; TG column tiles per reduce group, NG groups ((2 R + 1) * 2 W <= 32 TG NG);
// NT1 column tiles of the band (R * W <= 32 NT1); SC channel slabs per chunk of the input stream
template <int M, int NW, int WN, int TG, int NG, int NT1, int SC, bool DUAL>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_bfirst_kernel(BFirstArgs a) {
  constexpr int CIN = 2 * M, COUT = 4 * M;
  constexpr int WM = NW / WN, MT = M / (32 * WM);
  static_assert(WM * WN == NW && MT * 32 * WM == M && TG % WN == 0 && NT1 % WN == 0, "wave grid");
  constexpr int J0 = TG / WN, J1 = NT1 / WN;
  constexpr int KSI = CIN / 64, KS2 = M / 64, NE = 9 * KS2;
  constexpr int NPG = 32 * TG, NP1 = 32 * NT1;           // pixels of a reduce group / of the band
  static_assert(KSI % SC == 0 && SC >= 2, "whole chunks; a chunk's DMAs are told from its fragment loads by a counted wait");
  constexpr int NCH = KSI / SC;                          // chunks per group
  constexpr int CHUNK = SC * NPG * 64;
  constexpr int XS = KSI * NP1 * 64;                     // the shortcut's gathered operand
  constexpr int RING = 2 * CHUNK > XS ? 2 * CHUNK : XS;
  constexpr int NPASS = COUT / M;
  constexpr int NSP = (DUAL ? 2 : 1) * KSI + KS2;        // steps of one pass of phase 2
  constexpr int LEAN = 0;

  __shared__ __attribute__((aligned(1024))) int8_t ring[RING];
  extern __shared__ __attribute__((aligned(1024))) int8_t dyn[];     // [mid1 halo tile][mid2][hdrA][hdrB][hdrP x 2]
  int8_t* const xs = ring;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;
  const int W = a.W, H = a.H, R = a.R, Win = 2 * a.W, Hin = 2 * a.H, Wp = 2 * a.W + 2;
  const int n_hr = 2 * R + 1;                            // input rows of the halo band
  const int n_h = n_hr * Wp;
  const int n_grp_h = (n_h + 15) >> 4;
  const int slabb = n_grp_h * 1024;
  int8_t* const mid1 = dyn;
  int8_t* const mid2 = mid1 + KS2 * slabb;
  const int tmsS = a.tms == 128 ? 7 : 6, tms1 = a.tm1 == 128 ? 7 : 6, tms2 = a.tm2 == 128 ? 7 : 6, tms3 = a.tm3 == 128 ? 7 : 6;
  const int hstS = (DUAL ? 28 : 20) << tmsS, hst1 = (DUAL ? 28 : 20) << tms1, hst2 = 20 << tms2, hst3 = 20 << tms3;
  int8_t* const hdr1 = mid2 + KS2 * NP1 * 64;
  int8_t* const hdr2 = hdr1 + (M >> tms1) * hst1;
  int8_t* const hdrP = hdr2 + (M >> tms2) * hst2;        // two buffers of [S rows of the pass | E rows of the pass]
  const int hdrP_E = (M >> tmsS) * hstS;                 // offset of the E part
  const int hdrP_bytes = hdrP_E + (M >> tms3) * hst3;

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int img = bid / a.tiles_per_img;
  const int r0 = (bid - img * a.tiles_per_img) * R;      // first OUTPUT row of the band
  const int rows = (H - r0) < R ? (H - r0) : R;
  const int n_px = rows * W;
  const int n_p0 = n_hr * Win;                           // halo-band input pixels (input row 2 r0 - 1 first)
  const int in_row0 = 2 * r0 - 1;
  const long long pix_base = ((long long)img * H + r0) * W;            // output-map pixel index of band pixel 0
  const long long pin0 = ((long long)img * Hin + in_row0) * Win;       // input-map pixel index of halo-band pixel 0 (row may be -1)

  long long* const dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bband_timeline.py
#define BF_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  BF_STAMP(0);
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;

  // ---- prologue: pad fill of the halo tile, the first two chunks, the reduce's and the 3x3's headers ----
  for (int gi = wave; gi < n_grp_h * KS2; gi += NW) {
    const int s = gi / n_grp_h, grp = gi - s * n_grp_h;
    bf_dma16(a.zero2 + s * 64 + chunk * 16, mid1 + s * slabb + grp * 1024);
  }
  // chunk item ci = (group g, chunk c): slabs [c * SC, (c + 1) * SC) of the group's NPG pixels
  auto issue_chunk = [&](int ci) {
    const int g = ci / NCH, c = ci - g * NCH;
    int8_t* const buf = ring + (ci & 1) * CHUNK;
    for (int gi = wave; gi < SC * (NPG / 16); gi += NW) {
      const int sl = gi / (NPG / 16), grp = gi - sl * (NPG / 16);
      const int p = g * NPG + grp * 16 + drow;
      const int row = in_row0 + p / Win;
      const bool ok = p < n_p0 && (unsigned)row < (unsigned)Hin;
      const int8_t* src = ok ? a.x + (size_t)(pin0 + p) * CIN + (c * SC + sl) * 64 + chunk * 16 : a.zero + chunk * 16;
      bf_dma16(src, buf + sl * (NPG * 64) + grp * 1024);
    }
  };
  issue_chunk(0);
  issue_chunk(1);
  auto hdr_copy = [&](const int32_t* hdr, int hdr_bytes, int tms, int mt0, int n_mt, int8_t* dst, int words_per_row) {
    const int per = words_per_row << (tms - 2);
    for (int i = tid; i < n_mt * per; i += NW * 64) {
      const int mt = i / per, k = i - mt * per;
      const i32x4 v = *reinterpret_cast<const i32x4*>(reinterpret_cast<const int8_t*>(hdr) + (size_t)(mt0 + mt) * hdr_bytes + k * 16);
      *reinterpret_cast<i32x4*>(dst + (size_t)mt * (per * 16) + k * 16) = v;
    }
  };
  hdr_copy(a.hdr1, a.hdr1_bytes, tms1, 0, M >> tms1, hdr1, DUAL ? 7 : 5);
  hdr_copy(a.hdr2, a.hdr2_bytes, tms2, 0, M >> tms2, hdr2, 5);
  auto hdr_pass = [&](int q) {                            // pass q's rows of S and E into buffer q & 1
    int8_t* const d = hdrP + (q & 1) * hdrP_bytes;
    hdr_copy(a.hdrs, a.hdrs_bytes, tmsS, (q * M) >> tmsS, M >> tmsS, d, DUAL ? 7 : 5);
    hdr_copy(a.hdr3, a.hdr3_bytes, tms3, (q * M) >> tms3, M >> tms3, d + hdrP_E, 5);
  };
  hdr_pass(0);

  struct Afr { i32x4 k[MT][2]; };
  const int cb_w = wm * (MT * 32);
  auto load_a = [&](Afr& f, const int8_t* w, int tms, int nslab, int cb, int slab, int wins = 1, int win = 0) {
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int ch = cb + i * 32;
      const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
      const int8_t* p = w + (((((size_t)mt * nslab + slab) * wins + win) << tms) + ro + (lane & 31)) * 64 + half * 16;
      f.k[i][0] = *reinterpret_cast<const i32x4*>(p);
      f.k[i][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
  };
  constexpr int N0 = NG * KSI;                           // steps of phase 0 (group-major)
  // fragments of global step v: phase 0 -- slab v % KSI of the reduce's high window; phase 1 -- (tap, slab) of the 3x3;
  // phase 2 -- pass q: the shortcut's (window, slab) steps, then the expand's slabs
  auto load_step = [&](Afr& f, auto v_c) {
    constexpr int v = decltype(v_c)::value;
    if constexpr (v < N0) load_a(f, a.w1, tms1, KSI, cb_w, v % KSI, DUAL ? 2 : 1, 0);
    else if constexpr (v < N0 + NE) load_a(f, a.w2, tms2, NE, cb_w, v - N0);
    else if constexpr (v < N0 + NE + NPASS * NSP) {
      constexpr int u = v - N0 - NE, q = u / NSP, t = u % NSP;
      if constexpr (t < NSP - KS2) load_a(f, a.ws, tmsS, KSI, q * M + cb_w, t % KSI, DUAL ? 2 : 1, t / KSI);
      else load_a(f, a.w3, tms3, KS2, q * M + cb_w, t - (NSP - KS2));
    }
  };
  Afr f0, f1, f2, f3, g0, g1, g2, g3;                    // g*: DUAL -- the reduce's low-window fragments
#define BF_BUF(v) ((v) % 4 == 0 ? f0 : (v) % 4 == 1 ? f1 : (v) % 4 == 2 ? f2 : f3)
#define BF_BUFL(v) ((v) % 4 == 0 ? g0 : (v) % 4 == 1 ? g1 : (v) % 4 == 2 ? g2 : g3)
  load_step(f0, std::integral_constant<int, 0>{});
  load_step(f1, std::integral_constant<int, 1>{});
  if constexpr (DUAL) { load_a(g0, a.w1, tms1, KSI, cb_w, 0, 2, 1); load_a(g1, a.w1, tms1, KSI, cb_w, 1 % KSI, 2, 1); }

  constexpr int JM = J0 > J1 ? J0 : J1;
  i32x16 acc[MT][JM];
  i32x16 acc2[DUAL ? MT : 1][DUAL ? J0 : 1];
  auto zero_acc = [&](int nj) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < JM; j++)
        if (j < nj)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
  };
  auto zero_acc2 = [&]() __attribute__((always_inline)) {
    if constexpr (DUAL) {
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < J0; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc2[i][j][r] = 0;
    }
  };
  zero_acc(J0); zero_acc2();
  // Horner step of a two-window layer: acc = (acc << dshift[1][row]) [+ acc2]; dshift sits behind rows | lo of the m-tile's image
  auto window_combine = [&](const int8_t* hdr, int tms, int hst, int cb, int nj, bool add_low) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int ch = cb + i * 32;
      const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
      const int* dsh = reinterpret_cast<const int*>(hdr + mt * hst) + (6 << tms) + ro + 4 * half;
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < JM; j++)
            if (j < nj) {
              unsigned vv = (unsigned)acc[i][j][G * 4 + r] << (d[r] & 31);
              if constexpr (DUAL) { if (add_low) vv += (unsigned)acc2[i][j < J0 ? j : 0][G * 4 + r]; }
              acc[i][j][G * 4 + r] = (int)vv;
            }
      }
    }
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  BF_STAMP(1);

  // ---- phase 0: reduce over the halo band, group by group ------------------------------------------------------------------
  int bm0[J0];
#pragma unroll
  for (int j = 0; j < J0; j++) {
    const int row = (wn + j * WN) * 32 + (lane & 31);
    bm0[j] = row * 64 + ((half ^ ((row >> 2) & 3)) << 4);
  }
  const i32x4 nores = {0, 0, 0, 0};
  auto group_out = [&](int g) __attribute__((always_inline)) {                           // requantise group g into the halo tile (rows outside the image keep the pad value)
    if constexpr (DUAL) window_combine(hdr1, tms1, hst1, cb_w, J0, true);
    const int lo_b = a.relu1 ? 0 : -128;
    auto go = [&](auto fast_c) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const int ch = cb_w + i * 32;
        const int mt = ch >> tms1, ro = ch & ((1 << tms1) - 1);
        const int* prm = reinterpret_cast<const int*>(hdr1 + mt * hst1);
        const int chl = ch + 16 * half;
#pragma unroll
        for (int j = 0; j < J0; j++) {
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
          const i32x4 out = requant_tile16<false, LEAN, FAST>(a16, prm, 1 << tms1, ro + 4 * half, lo_b, -128, nores, a.dbl1 != 0, a.fast1 == 2);
          const int p = g * NPG + (wn + j * WN) * 32 + (lane & 31);
          const int hr = p / Win, col = p - hr * Win;
          const int row = in_row0 + hr;
          if (p < n_p0 && (unsigned)row < (unsigned)Hin) {
            const int h = hr * Wp + col + 1;
            const int c = (chl & 63) >> 4;
            *reinterpret_cast<i32x4*>(mid1 + (chl >> 6) * slabb + h * 64 + ((c ^ ((h >> 2) & 3)) << 4)) = out;
            // keep_all: every input row but the band's first is this band's to write out (that one is the previous band's last)
            if (a.keep_mid && hr >= 1 && hr <= 2 * rows)
              *reinterpret_cast<i32x4*>(a.mid1 + (size_t)(pin0 + p) * M + chl) = out;
          }
        }
      }
    };
    if (a.fast1 == 1) go(std::true_type{}); else go(std::false_type{});
    // (keep_all only: the stores above sit in the same counter as the chunk DMAs the next groups wait for by count)
    if (a.keep_mid) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    zero_acc(J0); zero_acc2();
  };
  auto step0 = [&](auto v_c) {
    constexpr int v = decltype(v_c)::value;                // group-major slab index
    constexpr int g = v / KSI, s = v % KSI, ci = v / SC, sl = v % SC;
    Afr& cur = BF_BUF(v);
    Afr& nxt = BF_BUF(v + 2);
    if constexpr (sl == 0 && ci > 0) {
      // chunk item ci landed in every wave, nobody reads buffer (ci + 1) & 1 any more (conv_bband.hip explains the count)
      bf_wait_vmcnt<(SC - 1 < 2 ? SC - 1 : 2) * (DUAL ? 4 : 2) * MT>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    load_step(nxt, std::integral_constant<int, v + 2>{});
    if constexpr (DUAL && v + 2 < N0) load_a(BF_BUFL(v + 2), a.w1, tms1, KSI, cb_w, (v + 2) % KSI, 2, 1);
    if constexpr (sl == 0 && ci > 0 && ci + 1 < NG * NCH) {
      asm volatile("" ::: "memory");
      issue_chunk(ci + 1);
    }
    const int8_t* B = ring + (ci & 1) * CHUNK + sl * (NPG * 64);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x4 bf[J0];
#pragma unroll
      for (int j = 0; j < J0; j++) bf[j] = *reinterpret_cast<const i32x4*>(B + (bm0[j] ^ (ks << 5)));
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < J0; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
          if constexpr (DUAL) acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(BF_BUFL(v).k[i][ks], bf[j], acc2[i][j], 0, 0, 0);
        }
    }
    if constexpr (s == KSI - 1) group_out(g);
    __builtin_amdgcn_sched_barrier(0);
  };
  bf_static_for<0, N0>(step0);
  zero_acc(J1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // the halo tile is complete; the chunk ring is free
  asm volatile("" ::: "memory");
  BF_STAMP(2);
  // the shortcut's operand: the band's own pixels of the INPUT map at (2 r, 2 c), all Cin channels -> xs [slab][pixel][64] (lands during phase 1)
  for (int gi = wave; gi < KSI * (NP1 / 16); gi += NW) {
    const int sl = gi / (NP1 / 16), grp = gi - sl * (NP1 / 16);
    const int p = grp * 16 + drow;
    const int r = p / W, c = p - r * W;
    const bool ok = p < n_px;
    const int8_t* src = ok ? a.x + (size_t)(((long long)img * Hin + 2 * (r0 + r)) * Win + 2 * c) * CIN + sl * 64 + chunk * 16 : a.zero + chunk * 16;
    bf_dma16(src, xs + sl * (NP1 * 64) + grp * 1024);
  }

  // ---- phase 1: the 3x3 / stride 2 over the halo tile ------------------------------------------------------------------------------
  int h0[J1];
#pragma unroll
  for (int j = 0; j < J1; j++) {
    int p = (wn + j * WN) * 32 + (lane & 31);
    if (p >= n_px) p = 0;
    const int r = p / W;
    h0[j] = 2 * r * Wp + 2 * (p - r * W);
  }
  auto step1 = [&](auto e_c) {
    constexpr int e = decltype(e_c)::value;
    constexpr int v = N0 + e;
    constexpr int t = e / KS2, s = e % KS2;
    Afr& cur = BF_BUF(v);
    Afr& nxt = BF_BUF(v + 2);
    load_step(nxt, std::integral_constant<int, v + 2>{});
    const int8_t* B = mid1 + s * slabb;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x4 bf[J1];
#pragma unroll
      for (int j = 0; j < J1; j++) {
        const int h = h0[j] + (t / 3) * Wp + t % 3;
        bf[j] = *reinterpret_cast<const i32x4*>(B + ((h * 64 + ((half ^ ((h >> 2) & 3)) << 4)) ^ (ks << 5)));
      }
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < J1; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  bf_static_for<0, NE>(step1);
  BF_STAMP(3);
  {
    const int lo_b = a.relu2 ? 0 : -128;
    auto to_mid2 = [&](auto fast_c) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const int ch = cb_w + i * 32;
        const int mt = ch >> tms2, ro = ch & ((1 << tms2) - 1);
        const int* prm = reinterpret_cast<const int*>(hdr2 + mt * hst2);
        const int chl = ch + 16 * half;
#pragma unroll
        for (int j = 0; j < J1; j++) {
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
          const i32x4 out = requant_tile16<false, LEAN, FAST>(a16, prm, 1 << tms2, ro + 4 * half, lo_b, -128, nores, a.dbl2 != 0, a.fast2 == 2);
          const int row = (wn + j * WN) * 32 + (lane & 31);
          const int c = (chl & 63) >> 4;
          *reinterpret_cast<i32x4*>(mid2 + (chl >> 6) * (NP1 * 64) + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = out;
          if (a.keep_mid && row < n_px)
            *reinterpret_cast<i32x4*>(a.mid2 + (size_t)(pix_base + row) * M + chl) = out;
        }
      }
    };
    if (a.fast2 == 1) to_mid2(std::true_type{}); else to_mid2(std::false_type{});
  }
  zero_acc(J1);
  BF_STAMP(4);

  // ---- phase 2: per pass of M output channels -- shortcut tile, expand tile, residual add -------------------------------------
  int bm1[J1];
#pragma unroll
  for (int j = 0; j < J1; j++) {
    const int row = (wn + j * WN) * 32 + (lane & 31);
    bm1[j] = row * 64 + ((half ^ ((row >> 2) & 3)) << 4);
  }
  const int lo_bs = a.relu_s ? 0 : -128, lo_b3 = a.relu3 ? 0 : -128, rlo = a.add_relu ? 0 : -128;
  i32x4 sres[MT][J1];                                    // the pass's shortcut values, as the expand's residual
  auto step2 = [&](auto u_c) {
    constexpr int u = decltype(u_c)::value;
    constexpr int v = N0 + NE + u;
    constexpr int q = u / NSP, t = u % NSP;
    constexpr bool is_s = t < NSP - KS2;                   // a shortcut step
    Afr& cur = BF_BUF(v);
    Afr& nxt = BF_BUF(v + 2);
    if constexpr (t == 0) {
      // every wave is past the previous pass (its header buffer may be rewritten) and pass q's header rows are visible;
      // q == 0: the B tile and the gathered shortcut operand are complete as well
      if constexpr (q == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (q + 1 < NPASS) hdr_pass(q + 1);
    }
    load_step(nxt, std::integral_constant<int, v + 2>{});
    if constexpr (DUAL && is_s && t == KSI) window_combine(hdrP + (q & 1) * hdrP_bytes, tmsS, hstS, cb_w, J1, false);      // between S's windows (rows local to the pass buffer)
    const int8_t* B = is_s ? xs + (t % KSI) * (NP1 * 64) : mid2 + (t - (NSP - KS2)) * (NP1 * 64);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x4 bf[J1];
#pragma unroll
      for (int j = 0; j < J1; j++) bf[j] = *reinterpret_cast<const i32x4*>(B + (bm1[j] ^ (ks << 5)));
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < J1; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (t == NSP - KS2 - 1) {
      // the shortcut tile: requantised to row S's int8 values (no residual of its own), kept as the expand's residual
      const int8_t* hb = hdrP + (q & 1) * hdrP_bytes;
      auto sq = [&](auto fast_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
        for (int i = 0; i < MT; i++) {
          const int chp = cb_w + i * 32;                   // channel inside the pass
          const int mt = chp >> tmsS, ro = chp & ((1 << tmsS) - 1);
          const int* prm = reinterpret_cast<const int*>(hb + mt * hstS);
          const int chl = q * M + chp + 16 * half;
#pragma unroll
          for (int j = 0; j < J1; j++) {
            int a16[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
            sres[i][j] = requant_tile16<false, LEAN, FAST>(a16, prm, 1 << tmsS, ro + 4 * half, lo_bs, -128, nores, false, a.fast_s == 2);
            const int p = (wn + j * WN) * 32 + (lane & 31);
            if (a.keep_mid && p < n_px) *reinterpret_cast<i32x4*>(a.ys + (size_t)(pix_base + p) * a.ys_cp + chl) = sres[i][j];
          }
        }
      };
      if (a.fast_s == 1) sq(std::true_type{}); else sq(std::false_type{});
      zero_acc(J1);
    }
    if constexpr (t == NSP - 1) {
      const int8_t* hb = hdrP + (q & 1) * hdrP_bytes + hdrP_E;
      auto eq = [&](auto fast_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
        for (int i = 0; i < MT; i++) {
          const int chp = cb_w + i * 32;
          const int mt = chp >> tms3, ro = chp & ((1 << tms3) - 1);
          const int* prm = reinterpret_cast<const int*>(hb + mt * hst3);
          const int chl = q * M + chp + 16 * half;
#pragma unroll
          for (int j = 0; j < J1; j++) {
            int a16[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
            const i32x4 out = requant_tile16<true, LEAN, FAST>(a16, prm, 1 << tms3, ro + 4 * half, lo_b3, rlo, sres[i][j], false, a.fast3 == 2);
            const int p = (wn + j * WN) * 32 + (lane & 31);
            if (p < n_px) *reinterpret_cast<i32x4*>(a.y + (size_t)(pix_base + p) * a.y_cp + a.y_off + chl) = out;
          }
        }
      };
      if (a.fast3 == 1) eq(std::true_type{}); else eq(std::false_type{});
      zero_acc(J1);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  bf_static_for<0, NPASS * NSP>(step2);
  BF_STAMP(5);
#undef BF_STAMP
#undef BF_BUF
#undef BF_BUFL
}

static size_t bfirst_dyn_lds(int M, int R, int W, int NP1, bool dual) {
  const int n_h = (2 * R + 1) * (2 * W + 2);
  const size_t hA = (size_t)(dual ? 28 : 20) * M, hB = (size_t)20 * M, hP = (size_t)((dual ? 28 : 20) + 20) * M;
  return (size_t)(M / 64) * (((n_h + 15) >> 4) * 1024) + (size_t)(M / 64) * NP1 * 64 + hA + hB + 2 * hP + 64;
}

// LDS of an instantiation: static chunk ring (the shortcut's gathered operand re-uses it) + dynamic tiles and headers
template <int M, int TG, int NT1, int SC>
static bool bfirst_fits(int R, int W, bool dual, int NG) {
  constexpr int KSI = 2 * M / 64;
  constexpr int CHUNK = SC * 32 * TG * 64, XS = KSI * 32 * NT1 * 64;
  const size_t stat = 2 * CHUNK > XS ? 2 * CHUNK : XS;
  return bfirst_dyn_lds(M, R, W, 32 * NT1, dual) + stat <= 160 * 1024 && (2 * R + 1) * 2 * W <= 32 * TG * NG && R * W <= 32 * NT1;
}

template <int M, int NW, int WN, int TG, int NG, int NT1, int SC, bool DUAL>
static int launch_bfirst2(const BFirstArgs& a, hipStream_t s) {
  constexpr int KSI = 2 * M / 64;
  constexpr int CHUNK = SC * 32 * TG * 64, XS = KSI * 32 * NT1 * 64;
  const size_t stat = 2 * CHUNK > XS ? 2 * CHUNK : XS;
  const size_t dyn = bfirst_dyn_lds(M, a.R, a.W, 32 * NT1, DUAL);
  if (!bfirst_fits<M, TG, NT1, SC>(a.R, a.W, DUAL, NG)) return 1;
  auto fn = conv_bfirst_kernel<M, NW, WN, TG, NG, NT1, SC, DUAL>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn), 160 * 1024 - (int)stat)) return -1;
  TF2_LAUNCH_NAME("conv_bfirst_kernel<%dx%d->%dx%d,Cin%d,M%d,R%d%s> (%d bands per image)", 2 * a.H, 2 * a.W, a.H, a.W, 2 * M, M, a.R,
                  DUAL ? ",dual shortcut+reduce" : "", a.tiles_per_img);
  TF2_LAUNCH(fn, dim3(a.B * a.tiles_per_img), dim3(NW * 64), dyn, s, a);
  return launch_ok() ? 0 : -1;
}

// Shapes instantiated (output map H x W, M): ResNet-50 stage 4's first bottleneck (14 x 14, M = 256) and stage 3's (28 x 28, M = 128);
// S and A both two-window or both single.  Tile plans (input pixels of the halo band = groups x tiles; band pixels = tiles):
//   14 x 14, M 256: R = 4: 9 x 28 = 252 = 2 x 4 tiles, 56 = 2 tiles;  R = 2: 5 x 28 = 140 = 2 x 3 tiles, 28 = 1 tile
//   28 x 28, M 128: R = 4: 9 x 56 = 504 = 4 x 4 tiles, 112 = 4 tiles; R = 2: 5 x 56 = 280 = 3 x 4 tiles, 56 = 2 tiles
bool conv_bfirst_shape_ok(int H, int W, int M, int R, int dual) {
  if (H != W || R < 1 || R > H) return false;
  if (M == 256 && W == 14) return R == 4 ? bfirst_fits<256, 4, 2, 2>(R, W, dual, 2) : R == 2 ? bfirst_fits<256, 3, 1, 2>(R, W, dual, 2) : false;
  if (M == 128 && W == 28) return R == 4 ? bfirst_fits<128, 4, 4, 2>(R, W, dual, 4) : R == 2 ? bfirst_fits<128, 4, 2, 2>(R, W, dual, 3) : false;
  return false;
}

int launch_conv_bfirst(const BFirstArgs& a, int M, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!conv_bfirst_shape_ok(a.H, a.W, M, a.R, a.dual)) return 1;
  if (M == 256 && a.W == 14) {
    if (a.R == 4) return a.dual ? launch_bfirst2<256, 8, 1, 4, 2, 2, 2, true>(a, s) : launch_bfirst2<256, 8, 1, 4, 2, 2, 2, false>(a, s);
    if (a.R == 2) return a.dual ? launch_bfirst2<256, 8, 1, 3, 2, 1, 2, true>(a, s) : launch_bfirst2<256, 8, 1, 3, 2, 1, 2, false>(a, s);
  }
  if (M == 128 && a.W == 28) {
    if (a.R == 4) return a.dual ? launch_bfirst2<128, 8, 2, 4, 4, 4, 2, true>(a, s) : launch_bfirst2<128, 8, 2, 4, 4, 4, 2, false>(a, s);
    if (a.R == 2) return a.dual ? launch_bfirst2<128, 8, 2, 4, 3, 2, 2, true>(a, s) : launch_bfirst2<128, 8, 2, 4, 3, 2, 2, false>(a, s);
  }
  return 1;
}

}  // namespace tf2
