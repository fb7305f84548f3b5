// Shared epilogue arithmetic of the MFMA convolution kernels (conv_mfma2 / conv_mfma_sk / conv_pw / conv_bneck / conv_stem).
//
// Reference semantics (device/src/pe.cl:191-194, relu.cl:54, feature_writer.cl:119-122), 32-bit truncation
// and wrap-around kept:
//   v = bias + (sum << lo)                                  (int32 wrap)
//   x = low32((v * alpha + (beta << 20)) >> 20)             == (int)((int64)v * alpha >> 20) + beta (wrapped)
//   y = sat32(x + 2^14) >> 15                               == ((x >> 14) + 1) >> 1 wherever either is < 2^16,
//                                                              and both clamp to 127 where they differ
//   c = clamp(y, relu ? 0 : -128, 127);  with a residual:  c = clamp(c + res, add_relu ? 0 : -128, 127)
//
// FAST (PackLayer::fast == 1, proven per layer at pack time): y = (sum * (alpha << lo) + B') >> 35 with
// B' = bias * alpha + (beta << 20) + 2^34 -- mad_i64_i32, ashr, med3.
// SEMI (PackLayer::fast == 2): v = bias + (sum << lo) MAY wrap (the reference's int32 accumulator, pe.cl:43) and is computed
// as such, but 2^31 * |alpha| + |beta << 20| + 2^34 < 2^51 for every row, so x = (v * alpha + (beta << 20)) >> 20 fits 32 bits
// and x + 2^14 cannot saturate: y = (v * alpha + B'') >> 35 with B'' = (beta << 20) + 2^34 (floor of floor) -- lshl_add,
// mad_i64_i32, ashr, med3: four instead of six.  ResNet-50's three layers that fail the FAST proof (conv1, two long-K layers)
// take this path.
//
// The epilogue is VALU-issue bound on the short-K layers (measured: ~13.5 issue slots per output before this
// header existed, 8.1k of a block's 15.6k cycles at 3 waves/SIMD), so the per-output instruction count is the
// thing to watch here: lshl_add, mad_i64_i32, alignbit, add clamp, ashr, med3 (+ sdwa add, med3 with a
// residual) + 0.75 perm.  The per-m-tile LDS header keeps beta64 as adjacent (lo,hi) words so that the 64-bit
// addend needs no register moves.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "tf2_internal.h"

namespace tf2 {

typedef int rq_i32x4 __attribute__((ext_vector_type(4)));
typedef long long rq_i64x2 __attribute__((ext_vector_type(2)));

// LDS header parameter block of one m-tile (weight_pack.cpp): TM rows of {bias, alpha, beta64.lo, beta64.hi}
// (one 16-byte read per output row, the 64-bit addend already in an aligned register pair), then lo[TM].
constexpr int kPrmWordsPerRow = 5;

// One 32x32 C/D tile: this lane holds rows (reg&3) + 8*(reg>>2) + 4*half of column (lane&31) in a16[reg].
// Returns the lane's 16 contiguous NHWC bytes (lanes 0-31: channels 0..15 of the tile, lanes 32-63: 16..31).
// LEAN (1 or 2): for kernels compiled for 8 waves/SIMD (64 registers); the number of rows read ahead.
// resv: with HAS_RES the 16 residual bytes of the same NHWC position (as loaded, before the swaps).
// RNN ("residual non-negative", round 5): the layer has no ReLU of its own (lo_bound = -128), adds a residual r that is a post-ReLU
// tensor (0 <= r <= 127) and clamps the sum to [0, 127] (add_relu) -- every identity bottleneck's expand.  Then
//     clamp(clamp(y, -128, 127) + r, 0, 127) == clamp(y + r, 0, 127)
// (y > 127: both 127; y < -128: the left side is clamp(r - 128) = 0, the right one clamp(y + r < -1) = 0; else identical), and the
// first v_med3 -- one of the 5.75 VALU instructions per output of these VALU-bound phases -- is left out.  The launch plan sets it
// (Net::res_nonneg_single_clamp); the kernels that always run such rows (conv_bband, the group kernels) are only selected for them.
// DBL (layers without a residual whose output tensor has "doubled" channels, weight_pack.cpp): header word 0 of a row (FAST;
// generic rows: bits 8.. of the row's shift word) is -128 for a doubled channel, 0 otherwise, and the stored value is
// (c << 1) - 128 resp. c.
template <bool HAS_RES, int LEAN, bool FAST, bool DBL, bool SEMI = false, bool RNN = false>
__device__ __forceinline__ rq_i32x4 requant_tile16_impl(const int (&a16)[16], const int* prm, int TM, int row0 /* tile row base + 4*half */,
                                                        int lo_bound, int rlo, const rq_i32x4& resv) {
  unsigned rd[4] = {0, 0, 0, 0};
  if (HAS_RES) {
    // the prefetched 16 contiguous bytes back into the C/D layout (the store swaps are involutions)
    auto r02 = __builtin_amdgcn_permlane32_swap((unsigned)resv[0], (unsigned)resv[1], false, false);
    auto r13 = __builtin_amdgcn_permlane32_swap((unsigned)resv[2], (unsigned)resv[3], false, false);
    rd[0] = r02[0]; rd[2] = r02[1]; rd[1] = r13[0]; rd[3] = r13[1];
  }
  unsigned d[4];
  const rq_i32x4* rowp = reinterpret_cast<const rq_i32x4*>(prm) + row0;
  const int* lop = prm + 4 * TM + row0;
  // LEAN (kernels built for 64 registers): the rows are processed strictly in order with the parameters of the
  // next LEAN rows in flight, instead of letting the scheduler hoist all sixteen reads
  rq_i32x4 pq[3];
  if (LEAN) { pq[0] = rowp[0]; if (LEAN > 1) pq[1] = rowp[1]; }
#pragma unroll
  for (int G = 0; G < 4; G++) {
    rq_i32x4 lo4 = {0, 0, 0, 0};
    if (!FAST) lo4 = *reinterpret_cast<const rq_i32x4*>(lop + 8 * G);
    int q[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int k = G * 4 + r;                             // 0..15, row = row0 + 8 * (k / 4) + k % 4
      rq_i32x4 pr;
      if (LEAN) {
        constexpr int NB = LEAN + 1;                       // row buffers: LEAN rows in flight ahead of the current
        if (k + LEAN < 16) pq[(k + LEAN) % NB] = rowp[8 * ((k + LEAN) / 4) + (k + LEAN) % 4];
        __builtin_amdgcn_sched_barrier(0);
        pr = pq[k % NB];
      } else {
        pr = rowp[8 * G + r];
      }
      const long long b64 = (long long)(((unsigned long long)(unsigned)pr[3] << 32) | (unsigned)pr[2]);
      int y;
      if (FAST) {
        // rows proven at pack time (weight_pack.cpp): y = (acc * (alpha << lo) + B') >> 35, no wrap anywhere
        const long long p = (long long)a16[k] * (long long)pr[1] + b64;
        y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
      } else if (SEMI) {
        // the 32-bit wrap of v kept; rows hold B'' = (beta << 20) + 2^34 as their 64-bit addend (weight_pack.cpp)
        const int v = (int)((unsigned)pr[0] + ((unsigned)a16[k] << (lo4[r] & 31)));
        const long long p = (long long)v * (long long)pr[1] + b64;
        y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
      } else {
        const int v = (int)((unsigned)pr[0] + ((unsigned)a16[k] << (lo4[r] & 31)));
        const long long p = (long long)v * (long long)pr[1] + b64;
        const int x = (int)(p >> kAlphaInflat);
        y = __builtin_elementwise_add_sat(x, 1 << (kInflat - 1)) >> kInflat;
      }
      static_assert(!RNN || (HAS_RES && !DBL), "RNN: a residual layer");
      int c = y;
      if (!RNN) asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(y), "s"(lo_bound), "v"(127));
      if (DBL) {
        const int kd = FAST ? pr[0] : (lo4[r] >> 8);          // -128 or 0 (generic rows keep it above the shift amount)
        c = (int)(((unsigned)c << ((unsigned)kd >> 31)) + (unsigned)kd);
      }
      if (HAS_RES) {
        const int rr = (int)(signed char)((rd[G] >> (8 * r)) & 0xff);
        const int sres = c + rr;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(sres), "s"(rlo), "v"(127));
      }
      q[r] = c;
    }
    const unsigned p01 = __builtin_amdgcn_perm((unsigned)q[1], (unsigned)q[0], 0x0c0c0400u);   // bytes: q0.b0, q1.b0
    const unsigned p23 = __builtin_amdgcn_perm((unsigned)q[3], (unsigned)q[2], 0x0c0c0400u);
    d[G] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
  }
  // d[G] = channel group 2G (lanes 0-31) / 2G+1 (lanes 32-63): two half-wave swaps give lanes 0-31 groups
  // 0..3 and lanes 32-63 groups 4..7 -> 16 contiguous NHWC bytes per lane
  auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
  auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
  return rq_i32x4{(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
}

// ---- NJ column tiles of one row tile in lockstep: every parameter row is read ONCE and applied to all NJ tiles ------------------
// (conv_bneck: two tiles per wave at 128 registers -- the row-by-row order of LEAN costs no register beyond the tiles' own
// packed words, and halves the parameter reads, which were two thirds of that kernel's LDS instructions)
template <int NJ, bool HAS_RES, int LEAN, bool FAST, bool DBL, bool SEMI, bool RNN = false>
__device__ __forceinline__ void requant_tiles16_impl(const int (&a16)[NJ][16], rq_i32x4 (&out)[NJ], const int* prm, int TM, int row0,
                                                     int lo_bound, int rlo, const rq_i32x4 (&resv)[NJ]) {
  static_assert(LEAN >= 1, "row-ordered form");
  unsigned rd[NJ][4], d[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    if (HAS_RES) {
      auto r02 = __builtin_amdgcn_permlane32_swap((unsigned)resv[j][0], (unsigned)resv[j][1], false, false);
      auto r13 = __builtin_amdgcn_permlane32_swap((unsigned)resv[j][2], (unsigned)resv[j][3], false, false);
      rd[j][0] = r02[0]; rd[j][2] = r02[1]; rd[j][1] = r13[0]; rd[j][3] = r13[1];
    } else { rd[j][0] = rd[j][1] = rd[j][2] = rd[j][3] = 0; }
  }
  const rq_i32x4* rowp = reinterpret_cast<const rq_i32x4*>(prm) + row0;
  const int* lop = prm + 4 * TM + row0;
  rq_i32x4 pq[3];
  pq[0] = rowp[0]; if (LEAN > 1) pq[1] = rowp[1];
#pragma unroll
  for (int G = 0; G < 4; G++) {
    rq_i32x4 lo4 = {0, 0, 0, 0};
    if (!FAST) lo4 = *reinterpret_cast<const rq_i32x4*>(lop + 8 * G);
    int q[NJ][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int k = G * 4 + r;
      constexpr int NB = LEAN + 1;
      if (k + LEAN < 16) pq[(k + LEAN) % NB] = rowp[8 * ((k + LEAN) / 4) + (k + LEAN) % 4];
      __builtin_amdgcn_sched_barrier(0);
      const rq_i32x4 pr = pq[k % NB];
      const long long b64 = (long long)(((unsigned long long)(unsigned)pr[3] << 32) | (unsigned)pr[2]);
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        int y;
        if (FAST) {
          const long long p = (long long)a16[j][k] * (long long)pr[1] + b64;
          y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
        } else {
          const int v = (int)((unsigned)pr[0] + ((unsigned)a16[j][k] << (lo4[r] & 31)));
          const long long p = (long long)v * (long long)pr[1] + b64;
          if (SEMI) y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
          else { const int x = (int)(p >> kAlphaInflat); y = __builtin_elementwise_add_sat(x, 1 << (kInflat - 1)) >> kInflat; }
        }
        static_assert(!RNN || (HAS_RES && !DBL), "RNN: a residual layer");
        int c = y;
        if (!RNN) asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(y), "s"(lo_bound), "v"(127));
        if (DBL) {
          const int kd = FAST ? pr[0] : (lo4[r] >> 8);
          c = (int)(((unsigned)c << ((unsigned)kd >> 31)) + (unsigned)kd);
        }
        if (HAS_RES) {
          const int rr = (int)(signed char)((rd[j][G] >> (8 * r)) & 0xff);
          const int sres = c + rr;
          asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(sres), "s"(rlo), "v"(127));
        }
        q[j][r] = c;
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const unsigned p01 = __builtin_amdgcn_perm((unsigned)q[j][1], (unsigned)q[j][0], 0x0c0c0400u);
      const unsigned p23 = __builtin_amdgcn_perm((unsigned)q[j][3], (unsigned)q[j][2], 0x0c0c0400u);
      d[j][G] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    auto s02 = __builtin_amdgcn_permlane32_swap(d[j][0], d[j][2], false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(d[j][1], d[j][3], false, false);
    out[j] = rq_i32x4{(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
  }
}

// (same dispatch as requant_tile16: dbl only without a residual, semi only read when !FAST)
template <int NJ, bool HAS_RES, int LEAN, bool FAST, bool RNN = false>
__device__ __forceinline__ void requant_tiles16(const int (&a16)[NJ][16], rq_i32x4 (&out)[NJ], const int* prm, int TM, int row0, int lo_bound, int rlo,
                                                const rq_i32x4 (&resv)[NJ], bool dbl = false /* wave-uniform */, bool semi = false) {
  if constexpr (!FAST) {
    if (semi) {
      if constexpr (!HAS_RES) {
        if (dbl) return requant_tiles16_impl<NJ, false, LEAN, false, true, true>(a16, out, prm, TM, row0, lo_bound, rlo, resv);
      }
      return requant_tiles16_impl<NJ, HAS_RES, LEAN, false, false, true, RNN>(a16, out, prm, TM, row0, lo_bound, rlo, resv);
    }
  }
  if constexpr (!HAS_RES) {
    if (dbl) return requant_tiles16_impl<NJ, false, LEAN, FAST, true, false>(a16, out, prm, TM, row0, lo_bound, rlo, resv);
  }
  return requant_tiles16_impl<NJ, HAS_RES, LEAN, FAST, false, false, RNN>(a16, out, prm, TM, row0, lo_bound, rlo, resv);
}

// ---- the same arithmetic with the rows' parameters held in registers across column tiles ------------------------------------------
// A wave that requantises several column tiles of ONE row tile (conv_bband's hand-overs: four tiles per wave) reads the same
// sixteen 16-byte parameter rows for each of them; with an LDS write between two tiles the compiler cannot keep them, and the reads
// (1 KiB of LDS return traffic each, eight cycles of the CU's LDS pipe) outweigh the tile's VALU work.  Here the rows are read
// once per row tile, eight at a time (32 registers + 8 for the shift words of rows that are not FAST; all sixteen at once spilled
// in the 7-row band kernels), and every column tile's two packed words of those eight rows are built before the next eight.
template <bool FAST, bool DBL, bool SEMI>
__device__ __forceinline__ unsigned rq_rows4(const int (&a16)[16], int G, const rq_i32x4 (&pr4)[4], const rq_i32x4& lo4, int lo_bound) {
  int q[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const rq_i32x4 pr = pr4[r];
    const int acc = a16[G * 4 + r];
    const long long b64 = (long long)(((unsigned long long)(unsigned)pr[3] << 32) | (unsigned)pr[2]);
    int y, lo = 0;
    if (FAST) {
      const long long p = (long long)acc * (long long)pr[1] + b64;
      y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
    } else {
      lo = lo4[r];
      const int v = (int)((unsigned)pr[0] + ((unsigned)acc << (lo & 31)));
      const long long p = (long long)v * (long long)pr[1] + b64;
      if (SEMI) y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
      else { const int x = (int)(p >> kAlphaInflat); y = __builtin_elementwise_add_sat(x, 1 << (kInflat - 1)) >> kInflat; }
    }
    int c;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(y), "s"(lo_bound), "v"(127));
    if (DBL) {
      const int kd = FAST ? pr[0] : (lo >> 8);
      c = (int)(((unsigned)c << ((unsigned)kd >> 31)) + (unsigned)kd);
    }
    q[r] = c;
  }
  const unsigned p01 = __builtin_amdgcn_perm((unsigned)q[1], (unsigned)q[0], 0x0c0c0400u);
  const unsigned p23 = __builtin_amdgcn_perm((unsigned)q[3], (unsigned)q[2], 0x0c0c0400u);
  return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// NJ column tiles of one row tile, no residual (acc_of(j) returns tile j's sixteen accumulators by reference; out[j] = its 16 bytes)
template <int NJ, bool FAST, class AccOf>
__device__ __forceinline__ void requant_tiles16_rows(AccOf acc_of, rq_i32x4 (&out)[NJ], const int* prm, int TM, int row0, int lo_bound,
                                                     bool dbl /* wave-uniform */, bool semi /* wave-uniform, read when !FAST */) {
  const rq_i32x4* rowp = reinterpret_cast<const rq_i32x4*>(prm) + row0;
  const int* lop = prm + 4 * TM + row0;
  auto run = [&](auto dbl_c, auto semi_c) __attribute__((always_inline)) {
    constexpr bool DBL = decltype(dbl_c)::value, SEMI = decltype(semi_c)::value;
    unsigned d[NJ][4];
#pragma unroll
    for (int gp = 0; gp < 2; gp++) {
      rq_i32x4 pr[2][4], lo4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int g = 0; g < 2; g++) {
#pragma unroll
        for (int r = 0; r < 4; r++) pr[g][r] = rowp[8 * (2 * gp + g) + r];
        if (!FAST) lo4[g] = *reinterpret_cast<const rq_i32x4*>(lop + 8 * (2 * gp + g));
      }
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int g = 0; g < 2; g++) d[j][2 * gp + g] = rq_rows4<FAST, DBL, SEMI>(acc_of(j), 2 * gp + g, pr[g], lo4[g], lo_bound);
      __builtin_amdgcn_sched_barrier(0);                 // (keeps the second eight rows' reads behind the first eight's use)
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      auto s02 = __builtin_amdgcn_permlane32_swap(d[j][0], d[j][2], false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(d[j][1], d[j][3], false, false);
      out[j] = rq_i32x4{(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
    }
  };
  if constexpr (FAST) {
    if (dbl) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{});
  } else {
    if (semi) { if (dbl) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}); }
    else { if (dbl) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}); }
  }
}

// FAST = the layer's PackLayer::fast == 1; semi (wave-uniform, only read when !FAST) = PackLayer::fast == 2
template <bool HAS_RES, int LEAN = 0, bool FAST = false, bool RNN = false>
__device__ __forceinline__ rq_i32x4 requant_tile16(const int (&a16)[16], const int* prm, int TM, int row0, int lo_bound, int rlo, const rq_i32x4& resv,
                                                   bool dbl = false /* wave-uniform */, bool semi = false) {
  if constexpr (!FAST) {
    if (semi) {
      if constexpr (!HAS_RES) {
        if (dbl) return requant_tile16_impl<false, LEAN, false, true, true>(a16, prm, TM, row0, lo_bound, rlo, resv);
      }
      return requant_tile16_impl<HAS_RES, LEAN, false, false, true, RNN>(a16, prm, TM, row0, lo_bound, rlo, resv);
    }
  }
  if constexpr (!HAS_RES) {
    if (dbl) return requant_tile16_impl<false, LEAN, FAST, true>(a16, prm, TM, row0, lo_bound, rlo, resv);
  }
  return requant_tile16_impl<HAS_RES, LEAN, FAST, false, false, RNN>(a16, prm, TM, row0, lo_bound, rlo, resv);
}

}  // namespace tf2
