// conv_bneck.hip -- branch2b (3x3 / stride 1 / pad 1, C -> C) + branch2c (1x1 expand C -> 4C, residual add, ReLU) of a
// ResNet bottleneck in ONE launch, operands fed from registers and LDS-resident tiles -- no LDS-DMA ring, no barrier
// inside the K loops (gfx950).
//
// Why not conv_mfma2 twice (round 1), nor the ring-based fusion tried first in round 2 (profiles/r02_fused_pairs_*):
// measured with in-kernel counters (tools/layer_times.py --stamps), a ring step of the 3x3 layers costs 700-1100 cycles
// for 256-512 cycles of MFMA -- every wave of a block is released by the same barrier, then all of them do their address
// arithmetic / DMA issue / LDS reads together while the matrix pipe idles, and the activation slab of every one of the
// 9 taps is gathered from L2 again (9x the bytes).  The kernel that already beat that structure was conv_pw (weights
// in registers, no barrier).  Same idea here, for the whole pair:
//
//   * a block owns R output rows x the full width W of one image (R * W <= TN pixels) and ALL C channels of the 3x3;
//   * the 3x3's input is fetched ONCE: the (R+2) x (W+2) halo tile (zero border = the padding of sequencer.cl:287) goes
//     global -> LDS by LDS-DMA in the prologue; the B operand of tap (dh, dw) is the same tile read at a shifted pixel
//     address (ds_read_b128, XOR-swizzled rows: any 16 consecutive pixels hit 16 distinct bank groups);
//   * weights (A operand) go global -> registers: a lane's MFMA fragment is 16 contiguous bytes of its row in the packed
//     tile; the next step's fragments are loaded while the current MFMAs run.  Two-window layers (weight_pack.cpp "dual")
//     are swept window by window into ONE accumulator with the Horner shift in between (the B operand sits in LDS, so
//     reading it twice is cheap) -- a second accumulator set would cost the 32 registers that decide between one and two
//     blocks per CU;  Waves never wait for each other inside a
//     phase, so the two waves of a SIMD drift apart and one's loads overlap the other's matrix work;
//   * the 3x3's requantised int8 tile (pe.cl:185-203, relu.cl:54) is written to an LDS "mid" tile in B-operand layout
//     and consumed by the four 1x1 passes (output channels 4C = 4 m-tiles of C rows), each followed by the usual
//     epilogue with the residual tile (feature_writer.cl:119-122) and 16-byte NHWC stores.
//
// Bit-identical to the two separate launches: same Z/2^32 sums, same requantisation, same int8 intermediate (with
// keep_mid it is also written to HBM so that per-layer parity tests can read it).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int T, int N, class F>
__device__ __forceinline__ void bn_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); bn_static_for<T + 1, N>(fn); }
}

constexpr int kBneckPasses = 4;
#ifndef TF2_BNECK_PF
#define TF2_BNECK_PF 1          // weight-fragment prefetch distance (1 or 2 steps)
#endif

// WM x WN = 8 waves, wave tile 32 x (32 * NTN): TM = 32 * WM channels of the 3x3 (= K of the expand), TN = 32 * NTN * WN pixels.
template <int WM, int WN, int NTN, bool DUAL1, bool DUAL2>
__global__ __launch_bounds__(512, 4) void conv_bneck_kernel(BneckArgs a) {
  constexpr int TM = WM * 32, WTN = 32 * NTN, TN = WN * WTN;
  constexpr int NSL = TM / 64;                           // 64-byte channel slabs of the C-channel tensors
  constexpr int A1_BYTES = (DUAL1 ? 2 : 1) * TM * 64, A2_BYTES = (DUAL2 ? 2 : 1) * TM * 64;
  constexpr int NE1 = 9 * NSL;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];

  TF2_PROBE_WORD(a.probe);           // timing probes (tf2_device.h; constant 0 in the product build)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;
  const int W = a.W, Wp = a.W + 2, R = a.R;
  const int n_h = (R + 2) * Wp;                          // halo pixels
  const int slabb = ((n_h + 15) & ~15) * 64;             // bytes of one 64-channel slab of the halo tile
  int8_t* const mid1 = lds;
  int8_t* const mid2 = lds + NSL * slabb;
  int* const prm1 = reinterpret_cast<int*>(mid2 + TN * TM);
  int* const prm2 = reinterpret_cast<int*>(reinterpret_cast<int8_t*>(prm1) + a.hdr1_used);

  // XCD-aware remap: consecutive tiles (neighbouring row bands of one image share halo rows) on one XCD
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int img = bid / a.tiles_per_img;
  const int r0 = (bid - img * a.tiles_per_img) * R;
  const int rows = (a.H - r0) < R ? (a.H - r0) : R;      // valid output rows of this tile
  const int n_px = rows * W;
  const long long pix_base = ((long long)img * a.H + r0) * W;      // NHWC pixel index of tile pixel 0 (rows are contiguous)

  long long* const dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bneck_timeline.py: 100 MHz wall clock per phase
#define BN_STAMP(i) do { if (dbg && (wave == 0 || wave == 7) && lane == 0) dbg[(i) + (wave == 7 ? 8 : 0)] = (long long)wall_clock64(); } while (0)
  BN_STAMP(0);
  // ---- prologue: headers and the halo tile by LDS-DMA, residual tiles and the first weight fragments by ordinary loads ----
  {
    const int8_t* h1 = reinterpret_cast<const int8_t*>(a.hdr1) + lane * 16;
    for (int i = wave; i * 1024 < a.hdr1_used; i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(h1 + i * 1024), TF2_LDS_PTR(reinterpret_cast<int8_t*>(prm1) + i * 1024), 16, 0, 0);
    const int per = a.hdr2_used >> 10;
    for (int i = wave; i < kBneckPasses * per; i += 8) {
      const int mt = i / per, k = i - mt * per;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(reinterpret_cast<const int8_t*>(a.hdr2) + (size_t)mt * a.hdr2_bytes + k * 1024 + lane * 16),
                                       TF2_LDS_PTR(reinterpret_cast<int8_t*>(prm2) + i * 1024), 16, 0, 0);
    }
    // halo tile: lane l of a DMA instruction fills pixel row (l >> 2), 16-byte slot (l & 3) of a 16-pixel group; with the
    // XOR swizzle slot c' of pixel h holds chunk c' ^ ((h >> 2) & 3)
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const int n_grp = (n_h + 15) >> 4;
    for (int gi = wave; gi < n_grp * NSL; gi += 8) {
      const int s = gi / n_grp, grp = gi - s * n_grp;
      const int h = grp * 16 + (lane >> 2);
      const int hr = fast_div(h, a.wp_m, a.wp_s), hc = h - hr * Wp;
      const int row = r0 - 1 + hr, col = hc - 1;
      const bool ok = h < n_h && (unsigned)row < (unsigned)a.H && (unsigned)col < (unsigned)W;
      const int8_t* src = ok ? a.x + (((long long)img * a.H + row) * W + col) * TM + s * 64 + chunk * 16
                             : a.zero + s * 64 + chunk * 16;       // the stored form of x = 0: the zero page, or the 3x3's pad row
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(mid1 + s * slabb + grp * 1024), 16, 0, 0);
    }
  }
  // per-lane B addresses inside the halo tile: pixel p = (r, c) -> halo pixel h0 + dh * (W + 2) + dw for tap (dh, dw), byte
  // address h * 64 + ((chunk ^ ((h >> 2) & 3)) << 4); the second K half (ks = 1) is the same address ^ 32 (chunk 2 + half =
  // half ^ 2 under the swizzle).  Recomputed per step (4 VALU per address) rather than kept in 18 registers.
  int h0[NTN];
#pragma unroll
  for (int j = 0; j < NTN; j++) {
    int p = wn * WTN + j * 32 + (lane & 31);
    if (p >= n_px) p = 0;                               // lanes beyond the tile compute on pixel 0 and are never stored
    const int r = fast_div(p, a.w_m, a.w_s);
    h0[j] = r * Wp + (p - r * W);
  }
  auto baddr = [&](int t, int j) {
    const int h = h0[j] + (t / 3) * Wp + t % 3;
    return h * 64 + ((half ^ ((h >> 2) & 3)) << 4);
  };

  struct Afr { i32x4 k[2]; };                            // the two K halves of one window of one weight tile
  const int a_row_off = (wm * 32 + (lane & 31)) * 64 + half * 16;
  // virtual step v of a phase = (window, entry): window-major, so that one accumulator serves both windows
  constexpr int NW1 = DUAL1 ? 2 : 1, NW2 = DUAL2 ? 2 : 1;
  auto load_a1 = [&](Afr& f, int v) {
    const int win = v / NE1, e = v - win * NE1;
    const int8_t* p = a.w1 + (size_t)e * A1_BYTES + win * (TM * 64) + a_row_off;
    f.k[0] = *reinterpret_cast<const i32x4*>(p); f.k[1] = *reinterpret_cast<const i32x4*>(p + 32);
  };
  Afr f0, f1, f2;                                        // weight fragments: two steps ahead of the MFMAs (PF = 2)
  constexpr int PF = TF2_BNECK_PF;
  load_a1(f0, 0);
  if (PF == 2 && NW1 * NE1 > 1) load_a1(f1, 1);

  i32x16 acc[NTN];
#pragma unroll
  for (int j = 0; j < NTN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0;
  // Horner step between the windows: acc <<= dshift[1][row]  (weight_pack.cpp: hi window first)
  auto window_shift = [&](const int* dsh) {
    const int rb = wm * 32 + 4 * half;
#pragma unroll
    for (int G = 0; G < 4; G++) {
      const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + TM + rb + 8 * G);
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < NTN; j++) acc[j][G * 4 + r] = (int)((unsigned)acc[j][G * 4 + r] << (d[r] & 31));
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BN_STAMP(1);
  __builtin_amdgcn_s_barrier();                          // headers + halo tile complete in every wave
  asm volatile("" ::: "memory");
  BN_STAMP(2);

  // ---- phase 1: the 3x3 over the halo tile; step e = (tap t, slab s) -----------------------------------------------------
  auto step1 = [&](auto v_c) {
    constexpr int v = decltype(v_c)::value;
    constexpr int win = v / NE1, e = v % NE1, t = e / NSL, s = e % NSL;
    Afr& cur = PF == 2 ? (v % 3 == 0 ? f0 : v % 3 == 1 ? f1 : f2) : ((v & 1) ? f1 : f0);
    Afr& nxt = PF == 2 ? ((v + 2) % 3 == 0 ? f0 : (v + 2) % 3 == 1 ? f1 : f2) : ((v & 1) ? f0 : f1);
    if (v + PF < NW1 * NE1 && !(prb & kProbeNoA)) load_a1(nxt, v + PF);
    if (win == 1 && e == 0) window_shift(prm1 + kPrmWordsPerRow * TM);
    const int8_t* B = mid1 + s * slabb;
    int ba[NTN];
#pragma unroll
    for (int j = 0; j < NTN; j++) ba[j] = baddr(t, j);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x4 bf[NTN];
#pragma unroll
      for (int j = 0; j < NTN; j++) bf[j] = *reinterpret_cast<const i32x4*>(B + (ba[j] ^ (ks << 5)));
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        if (prb & kProbeNoMfma) { asm volatile("" :: "v"(cur.k[ks]), "v"(bf[j])); continue; }
        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[ks], bf[j], acc[j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                   // steps stay in order: the unrolled loop must not pile up loads
  };
  bn_static_for<0, NW1 * NE1>(step1);
  BN_STAMP(3);
  // ---- hand-over: requantise the 3x3 (window combine, pe.cl:185-203, ReLU) into the mid tile -------------------------------
  {
    const int lo_bound = a.relu1 ? 0 : -128;
    const i32x4 nores = {0, 0, 0, 0};
    const int chl = wm * 32 + 16 * half;                  // this lane's 16 channels of the intermediate
    auto to_mid = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
      // (the wave's NTN column tiles row by row: each parameter row read once -- requant_epilogue.h requant_tiles16)
      int a16s[NTN][16];
      i32x4 outs[NTN], nores_j[NTN];
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        nores_j[j] = nores;
#pragma unroll
        for (int r = 0; r < 16; r++) a16s[j][r] = acc[j][r];
      }
      requant_tiles16<NTN, false, 1, FAST>(a16s, outs, prm1, TM, wm * 32 + 4 * half, lo_bound, -128, nores_j, a.dbl_mid != 0, a.fast1 == 2);
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        const i32x4 out = outs[j];
        const int row = wn * WTN + j * 32 + (lane & 31);
        const int c = (chl & 63) >> 4;
        *reinterpret_cast<i32x4*>(mid2 + (chl >> 6) * (TN * 64) + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = out;
        if (a.keep_mid && row < n_px)
          *reinterpret_cast<i32x4*>(a.y_mid + (size_t)(pix_base + row) * a.ymid_cp + chl) = out;
      }
    };
    if (a.fast1 == 1) to_mid(std::true_type{}); else to_mid(std::false_type{});
  }
#pragma unroll
  for (int j = 0; j < NTN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  BN_STAMP(4);
  __builtin_amdgcn_s_barrier();                          // the mid tile is complete
  asm volatile("" ::: "memory");
  BN_STAMP(5);

  // ---- phase 2: the 1x1 expand over the mid tile -------------------------------------------------------------------------------
  // Round 6 (conv_bfirst's scheme): a wave owns 32 of the 4 TM output channels per GROUP pass (wave w: channels 32 (w + 8 g) ..; one
  // pass at TM = 64, two at 128) with that group's weight fragments resident in registers, and sweeps the block's column tiles that
  // hold pixels in pairs -- a run-time loop, residual tiles one pair ahead.  (Rounds 2-5: four statically unrolled passes of TM
  // channels over the WM x WN wave grid of phase 1 -- an eighth, empty column tile's worth of work per pass on 56 x 56 maps, four times
  // the code, 52-88 bytes of scratch per lane whose write-back showed up as 10 MB of HBM writes per launch.)
  {
    constexpr int NGRP = (kBneckPasses * TM) / 256;
    static_assert(NGRP * 256 == kBneckPasses * TM, "eight waves x 32 channels per group pass");
    const int lo_bound2 = a.relu2 ? 0 : -128;
    const int rlo = a.add_relu ? 0 : -128;
    const int n_t = (n_px + 31) >> 5;                      // column tiles that hold pixels
    const int frow = lane & 31;
    const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);
    struct Afr2 { i32x4 k[NW2][NSL][2]; };
#pragma unroll 1
    for (int gp = 0; gp < NGRP; gp++) {
      const int ch = (wave + 8 * gp) * 32;                 // first of the wave's 32 output channels
      const int mt = ch / TM, ro = ch - mt * TM;
      // (the group's weight fragments come from the packed image again for every tile pair -- L2-resident, read-only -- instead of staying
      //  in registers: held, the compiler parks part of them in scratch, and spilled registers are dirty lines that reach HBM)
      const int8_t* const wgrp = a.w2 + (size_t)(mt * NSL) * A2_BYTES + ro * 64;
      const unsigned w_lane = (unsigned)(frow * 64 + half * 16);
      auto load_wf = [&](Afr2& f) __attribute__((always_inline)) {
#pragma unroll
        for (int win = 0; win < NW2; win++)
#pragma unroll
          for (int sl = 0; sl < NSL; sl++) {
            const int8_t* p = wgrp + sl * A2_BYTES + win * (TM * 64) + w_lane;
            f.k[win][sl][0] = *reinterpret_cast<const i32x4*>(p); f.k[win][sl][1] = *reinterpret_cast<const i32x4*>(p + 32);
          }
      };
      const int* const pm = reinterpret_cast<const int*>(reinterpret_cast<const int8_t*>(prm2) + (size_t)mt * a.hdr2_used);
      const int chl = ch + 16 * half;
      const bool ch_ok = chl + 16 <= a.y_nvalid;
      // (kernel-argument base + ONE 32-bit offset per lane: the scalar-base form of global loads / stores; the block's part of the offset is
      //  wave-uniform and pinned to a scalar register -- as 64-bit lane addresses these were eight of the registers the compiler spilled)
      const unsigned res_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)pix_base * (unsigned)a.res_cp + (unsigned)a.res_off + (unsigned)ch));
      const unsigned y_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)pix_base * (unsigned)a.y_cp + (unsigned)a.y_off + (unsigned)ch));
      const unsigned reso = (unsigned)(frow * a.res_cp + 16 * half), yo = (unsigned)(frow * a.y_cp + 16 * half);
      auto load_res2 = [&](i32x4 (&rv)[2], int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int p = (t0 + j) * 32 + frow;
          const bool ok = a.has_res && p < n_px && ch_ok;
          const int8_t* rp = ok ? a.res + (res_u + reso + (unsigned)((t0 + j) * 32 * a.res_cp)) : a.zero;
          rv[j] = *reinterpret_cast<const i32x4*>(rp);
        }
      };
      // (the epilogue's form -- residual, FAST rows, single clamp -- is chosen ONCE per group pass, outside the tile loop: five forms inside
      //  the loop body kept their common lane values alive across all of them, in scratch)
      auto sweep = [&](auto has_res_c, auto fast_c, auto rnn_c) __attribute__((always_inline)) {
      auto tiles = [&](auto nj_c, int t0, const i32x4 (&rv2)[2]) __attribute__((always_inline)) {
        constexpr int NJ = decltype(nj_c)::value;
        Afr2 wf;
        load_wf(wf);
        i32x16 ac[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) ac[j][r] = 0;
#pragma unroll
        for (int win = 0; win < NW2; win++) {
          if (win == 1) {
            const int* dsh = pm + (kPrmWordsPerRow + 1) * TM + ro + 4 * half;
#pragma unroll
            for (int G = 0; G < 4; G++) {
              const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
              for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < NJ; j++) ac[j][G * 4 + r] = (int)((unsigned)ac[j][G * 4 + r] << (d[r] & 31));
            }
          }
#pragma unroll
          for (int sl = 0; sl < NSL; sl++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
              i32x4 bf[NJ];
#pragma unroll
              for (int j = 0; j < NJ; j++) bf[j] = *reinterpret_cast<const i32x4*>(mid2 + sl * (TN * 64) + (t0 + j) * 2048 + (fr0 ^ (ks << 5)));
#pragma unroll
              for (int j = 0; j < NJ; j++) {
                if (prb & kProbeNoMfma) { asm volatile("" :: "v"(wf.k[win][sl][ks]), "v"(bf[j])); continue; }
                ac[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[win][sl][ks], bf[j], ac[j], 0, 0, 0);
              }
            }
        }
        if (prb & kProbeNoEpi) { asm volatile("" :: "v"(ac[0])); return; }
        i32x4 rv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) rv[j] = rv2[j];
        {
          constexpr bool HAS_RES = decltype(has_res_c)::value;
          constexpr bool FAST = decltype(fast_c)::value;
          constexpr bool RNN = decltype(rnn_c)::value;        // the residual is a post-ReLU tensor and the sum is clamped to [0, 127]: one clamp (requant_epilogue.h)
          int a16s[NJ][16];
          i32x4 outs[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) a16s[j][r] = ac[j][r];
          requant_tiles16<NJ, HAS_RES, 1, FAST, RNN>(a16s, outs, pm, TM, ro + 4 * half, lo_bound2, rlo, rv, a.dbl_out != 0, a.fast2 == 2);
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            const int p = (t0 + j) * 32 + frow;
            if (prb & kProbeNoStore) { asm volatile("" :: "v"(outs[j])); continue; }
            if (p < n_px && ch_ok) *reinterpret_cast<i32x4*>(a.y + (y_u + yo + (unsigned)((t0 + j) * 32 * a.y_cp))) = outs[j];
          }
        }
      };
      i32x4 ra[2], rb[2];
      load_res2(ra, 0);
      int t0 = 0;
#pragma unroll 1
      for (; t0 + 2 <= n_t; t0 += 2) {
        load_res2(rb, t0 + 2);                              // the next pair's residual tiles (past the block's pixels: the zero page)
        tiles(std::integral_constant<int, 2>{}, t0, ra);
#pragma unroll
        for (int j = 0; j < 2; j++) ra[j] = rb[j];
      }
      if (t0 < n_t) tiles(std::integral_constant<int, 1>{}, t0, ra);
      };
      if (a.fast2 == 1) {
        if (a.has_res) { if (a.rnn) sweep(std::true_type{}, std::true_type{}, std::true_type{}); else sweep(std::true_type{}, std::true_type{}, std::false_type{}); }
        else sweep(std::false_type{}, std::true_type{}, std::false_type{});
      } else { if (a.has_res) sweep(std::true_type{}, std::false_type{}, std::false_type{}); else sweep(std::false_type{}, std::false_type{}, std::false_type{}); }
      if (gp == 0) BN_STAMP(6);
    }
  }
  BN_STAMP(7);
#undef BN_STAMP
}

size_t conv_bneck_lds_bytes(int TM, int TN, int R, int W, size_t hdr1_used, size_t hdr2_used) {
  const int n_h = (R + 2) * (W + 2);
  return (size_t)(TM / 64) * (((n_h + 15) & ~15) * 64) + (size_t)TN * TM + hdr1_used + kBneckPasses * hdr2_used + 64;
}

template <int WM, int WN, int NTN>
static int launch_bneck_shape(const BneckArgs& a, hipStream_t s) {
  constexpr int TM = WM * 32, TN = WN * 32 * NTN;
  const size_t lds = conv_bneck_lds_bytes(TM, TN, a.R, a.W, (size_t)a.hdr1_used, (size_t)a.hdr2_used);
  if (lds > 160 * 1024 || a.R * a.W > TN) return 1;
  const int grid = a.B * a.tiles_per_img;
#define TF2_BN(D1, D2) do { auto fn = conv_bneck_kernel<WM, WN, NTN, D1, D2>; if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1; \
                           TF2_LAUNCH_NAME("conv_bneck_kernel<TM%d,TN%d,%s,%s>", TM, TN, D1 ? "dual" : "single", D2 ? "dual" : "single"); \
                           TF2_LAUNCH(fn, dim3(grid), dim3(512), lds, s, a); } while (0)
  if (a.dual1) { if (a.dual2) TF2_BN(true, true); else TF2_BN(true, false); }
  else { if (a.dual2) TF2_BN(false, true); else TF2_BN(false, false); }
#undef TF2_BN
  return launch_ok() ? 0 : -1;
}

// TM = channels of the 3x3, TN = pixel capacity of a block (a.R * a.W <= TN).  Instantiated: 64 x 256 and 128 x 128 (wave
// tile 32 x 64).  Measured and dropped (profiles/r02_bneck_shapes_b32.txt): the narrow tiles 64 x 128 / 128 x 64 (twice the
// blocks, half the rows: slower than the wide ones on 56x56 and 28x28 maps) and 256-channel pairs on 14x14 maps (128 or 224
// blocks per launch: no faster than the split-K kernel + the 1x1 they would replace).  Returns 1 for anything else.
int launch_conv_bneck(const BneckArgs& a, int TM, int TN, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (TM == 64 && TN == 256) return launch_bneck_shape<2, 4, 2>(a, s);
  if (TM == 128 && TN == 128) return launch_bneck_shape<4, 2, 2>(a, s);
  return 1;
}

}  // namespace tf2
