// conv_bfirst.hip -- the FIRST bottleneck of the 56 x 56 stage (ResNet-50 rows 1-4: projection shortcut 1x1 64 -> 256 | reduce 1x1
// 64 -> 64, 3x3 / 1 / pad 1 64 -> 64, expand 1x1 64 -> 256 + the shortcut as residual + ReLU; pe.cl:144-203 four times,
// feature_writer.cl:119-122 once) in ONE launch of INDEPENDENT row bands, two blocks per CU (gfx950).  Round 6.
//
// Why: with batches in flight those four rows were three launches (conv_pw x 2 + conv_bneck) that move 116 MB per batch of 32 --
// the shortcut's 25.7 MB map written and read back as the residual, the reduce's map written and read back with its halo, the
// pooled input read twice -- on a step whose chip-filling launches are bound by exactly that traffic and by VALU issue
// (profiles/r06_experiments.txt item 1: rows 1-4 cost 38 us of a 368 us step).  The group launch of the one-batch plan
// (conv_bgroup56f_kernel) fuses the same rows, but its 130 KB blocks own their CU and its eight members per image meet.  Here:
//
//   * a block owns R = 4 output rows x the full width of one image (224 pixels; 14 bands per image, 448 blocks per batch of 32: one
//     round at two blocks per CU) and ALL channels; nothing it reads is written by another block of the launch;
//   * the band's input -- rows r0 - 1 .. r0 + R of the pooled map, 64 bytes per pixel -- goes global -> LDS ONCE (21 KB) and serves
//     both the reduce (over the halo rows too: 6 / 4 of a one-slab layer) and, from the same tiles, the shortcut;
//   * reduce -> requantised straight into the 3x3's halo tile in LDS (borders and rows outside the image keep the stored form of
//     x = 0, sequencer.cl:287); 3x3 from that tile (a tap = a shifted address, conv_bneck's scheme) -> requantised into the expand's
//     B tile in LDS;
//   * expand: a wave owns 32 of the 256 output channels (weights of expand and shortcut resident in registers) and sweeps the band's
//     seven column tiles in pairs: the SHORTCUT tile first (K = 64: two MFMAs per window and column tile), requantised with its own header rows into 16 NHWC bytes per lane and column tile -- exactly the form the
//     expand's epilogue takes a residual in --, then the expand and the epilogue; 16-byte NHWC stores.  The shortcut's map never
//     exists (with keep_s it is written, for tf2_net_read_layer), nor do the reduce's and the 3x3's;
//   * weights global -> registers (a lane's MFMA fragment = 16 contiguous bytes of its row), the next phase's fragments in flight
//     while the current one is requantised; two-window layers are swept window by window into ONE accumulator with the Horner shift
//     in between (128 registers: two blocks per CU);
//   * LDS: input 22 KB + halo 22 KB + B tile 14 KB + header images 17.3 KB = 75.3 KB.
//
// HBM per batch of 32: 9.6 MB read + 25.7 MB written instead of 48 + 68.  Bit-identical to the four separate launches (same
// Z/2^32 sums, same requantisation: requant_epilogue.h; tests/test_gpu_parity.py).
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
__device__ __forceinline__ void bf_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); bf_static_for<T + 1, N>(fn); }
}

namespace {
constexpr int kBfHW = 56, kBfR = 4, kBfLead = 8;
constexpr int kBfNT0 = (kBfLead + (kBfR + 2) * kBfHW + 31) / 32;        // 11 tiles of 32 input pixels (8 lead pixels: band pixel 0 = tile 2)
constexpr int kBfNT1 = 7;                                               // column tiles of the band
constexpr int kBfHalo = ((kBfR + 2) * (kBfHW + 2) + 15) / 16 * 16;      // 352 halo pixels
__host__ __device__ constexpr int bf_hdr_bytes(int windows, int rows) { return (5 + windows) * rows * 4; }
__host__ __device__ constexpr size_t bf_lds_bytes(bool dual) {
  return (size_t)kBfNT0 * 2048 + (size_t)kBfHalo * 64 + (size_t)kBfNT1 * 2048 + bf_hdr_bytes(dual ? 2 : 1, 64) + bf_hdr_bytes(1, 64) +
         2 * (size_t)bf_hdr_bytes(dual ? 2 : 1, 256);
}
}  // namespace

// DUAL: reduce, expand and shortcut are two-window layers (ResNet-50 with the shipped Q file), else all single; the 3x3 is single.
template <bool DUAL>
__global__ __launch_bounds__(512, 4) void conv_bfirst_kernel(BGroupArgs a) {
  constexpr int HW = kBfHW, Wp = HW + 2, R = kBfR, NPB = R * HW, NHP = (R + 2) * HW, LEAD = kBfLead;
  constexpr int NT0 = kBfNT0, NT1 = kBfNT1, NHALO = kBfHalo;
  constexpr int NWIN = DUAL ? 2 : 1;
  constexpr int H64 = bf_hdr_bytes(NWIN, 64), H64S = bf_hdr_bytes(1, 64);
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  int8_t* const xt = lds;                                // [tile][32 pixels][64] swizzled: the band's input with its halo rows
  int8_t* const halo = xt + NT0 * 2048;                  // [halo pixel][64] swizzled: the 3x3's input
  int8_t* const mid2 = halo + NHALO * 64;                // [tile][32 pixels][64] swizzled: the 3x3's output, B operand of the expand
  int8_t* const hdr1 = mid2 + NT1 * 2048;
  int8_t* const hdr2 = hdr1 + H64;
  int8_t* const hdr3 = hdr2 + H64S;
  int8_t* const hdrS = hdr3 + 4 * H64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;               // 2 x 4 waves: 32 channels x (column tiles wn, wn + 4, ..)
  const int half = lane >> 5;
  const int frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);        // a lane's fragment of pixel frow of a [32][64] tile, K half 0
  const i32x4 nores = {0, 0, 0, 0};

  // XCD-aware remap: neighbouring bands of one image (they share halo rows of the input) on one XCD
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  constexpr int TPI = HW / R;                            // 14 bands per image
  const int img = bid / TPI;
  const int r0 = (bid - img * TPI) * R;
  const size_t px_img = (size_t)img * (HW * HW);
  const size_t px_band = px_img + (size_t)r0 * HW;

  long long* const dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/block timelines: 100 MHz wall clock per phase
#define BF_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  BF_STAMP(0);
  // (side job: this step's -128 flags of the input preparation -- PrepArgs::q128, read by conv_stem_pool_kernel, the launch in front of this
  //  one -- are cleared for the next step: BGroupArgs::ctr carries them here, B words)
  if (a.ctr && blockIdx.x == 0 && tid < a.B) a.ctr[tid] = 0u;

  // ---- prologue ---------------------------------------------------------------------------------------------------------------
  // LDS-DMA: lane l of an instruction fills pixel row l >> 2, 16-byte slot l & 3 of a 16-pixel group; with the XOR swizzle slot c'
  // of pixel h holds chunk c' ^ ((h >> 2) & 3)
  {
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
    // (1) the halo tile filled with the stored form of x = 0 (the 3x3's pad row)
    for (int gi = wave; gi < NHALO / 16; gi += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(a.zero2 + chunk * 16), TF2_LDS_PTR(halo + gi * 1024), 16, 0, 0);
    // (2) the band's input: LDS pixel q = LEAD + halo-band pixel (row r0 - 1 first)
    for (int gi = wave; gi < NT0 * 2; gi += 8) {
      const int hp = gi * 16 + drow - LEAD;
      const int hr = hp / HW;
      const int row = r0 - 1 + hr;
      const bool ok = hp >= 0 && hp < NHP && (unsigned)row < (unsigned)HW;
      const int8_t* src = ok ? a.x + (px_img + (size_t)(row * HW + (hp - hr * HW))) * 64 + chunk * 16 : a.zero + chunk * 16;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(xt + gi * 1024), 16, 0, 0);
    }
    // (3) header images (rows {bias | dbl, alpha, addend64} | lo | dshift per m-tile), exactly their used bytes
    auto hdr_copy = [&](const int32_t* hdr, int hdr_stride, int n_mt, int bytes, int8_t* dst) {
      const int per = bytes >> 4;
      for (int i = tid; i < n_mt * per; i += 512) {
        const int mt = i / per, k = i - mt * per;
        const i32x4 v = *reinterpret_cast<const i32x4*>(reinterpret_cast<const int8_t*>(hdr) + (size_t)mt * hdr_stride + k * 16);
        *reinterpret_cast<i32x4*>(dst + (size_t)mt * bytes + k * 16) = v;
      }
    };
    hdr_copy(a.hdr1, a.hdr1_bytes, 1, H64, hdr1);
    hdr_copy(a.hdr2, a.hdr2_bytes, 1, H64S, hdr2);
    hdr_copy(a.hdr3, a.hdr3_bytes, 4, H64, hdr3);
    hdr_copy(a.hdrs, a.hdrs_bytes, 256 / a.tms, bf_hdr_bytes(NWIN, 1) * a.tms, hdrS);
  }
  // weight fragments: a lane's MFMA A fragment is 16 contiguous bytes of its row in the packed tile [(window)][rows][64]
  struct Afr { i32x4 k[NWIN][2]; };
  struct Afr1 { i32x4 k[2]; };
  const unsigned a_lane_off = (unsigned)(frow * 64 + half * 16);
  Afr wsf, wf;                                           // shortcut / expand fragments of the current pass (phase 3); the reduce's first
  {
#pragma unroll
    for (int win = 0; win < NWIN; win++) {
      const int8_t* p = a.w1 + (size_t)(win * 64 + wm * 32) * 64 + a_lane_off;
      wf.k[win][0] = *reinterpret_cast<const i32x4*>(p); wf.k[win][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
  }
  auto load_w2 = [&](Afr1& f, int tap) __attribute__((always_inline)) {
    const int8_t* p = a.w2 + (size_t)(tap * 64 + wm * 32) * 64 + a_lane_off;
    f.k[0] = *reinterpret_cast<const i32x4*>(p); f.k[1] = *reinterpret_cast<const i32x4*>(p + 32);
  };
  auto load_ws = [&](Afr& f) __attribute__((always_inline)) {      // the wave's 32 output channels of phase 3
    const int ch = wave * 32;
#pragma unroll
    for (int win = 0; win < NWIN; win++) {
      const int8_t* p = a.ws + (((size_t)(ch / a.tms) * NWIN + win) * a.tms + ch % a.tms) * 64 + a_lane_off;
      f.k[win][0] = *reinterpret_cast<const i32x4*>(p); f.k[win][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
  };
  auto load_w3 = [&](Afr& f) __attribute__((always_inline)) {
#pragma unroll
    for (int win = 0; win < NWIN; win++) {
      const int8_t* p = a.w3 + (((size_t)(wave >> 1) * NWIN + win) * 64 + (wave & 1) * 32) * 64 + a_lane_off;
      f.k[win][0] = *reinterpret_cast<const i32x4*>(p); f.k[win][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
  };
  // Horner step between the windows of a two-window layer: acc <<= dshift[1][row]  (weight_pack.cpp: hi window first)
  auto window_shift = [&](auto& accs, auto nj_c, const int* prm, int tm, int rb) __attribute__((always_inline)) {
    constexpr int NJ = decltype(nj_c)::value;
    const int* dsh = prm + (kPrmWordsPerRow + 1) * tm + rb + 4 * half;
#pragma unroll
    for (int G = 0; G < 4; G++) {
      const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int j = 0; j < NJ; j++) accs[j][G * 4 + r] = (int)((unsigned)accs[j][G * 4 + r] << (d[r] & 31));
    }
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // pad fill, input tiles, headers: complete in every wave
  asm volatile("" ::: "memory");
  BF_STAMP(1);

  // ---- phase 1: reduce (1x1, 64 -> 64, K = one slab) over the band and its halo rows -> the 3x3's halo tile ---------------------
  Afr1 f0, f1;                                           // the 3x3's fragments, one step ahead
  {
    const int* const prm1 = reinterpret_cast<const int*>(hdr1);
    const int lo_b = a.relu1 ? 0 : -128;
    load_w2(f0, 0);                                      // the 3x3's first fragments are on their way while the reduce runs
    const int chl = wm * 32 + 16 * half;                 // this lane's 16 channels of the intermediate
    // (one column tile at a time in a run-time loop: 16 accumulator registers, a third of the code; the phase is 3 % of the block's MFMAs)
#pragma unroll 1
    for (int t = wn; t < NT0; t += 4) {
      const int8_t* B = xt + t * 2048;
      const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + fr0), b1 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
      i32x16 acc1[1];
#pragma unroll
      for (int r = 0; r < 16; r++) acc1[0][r] = 0;
      acc1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[0][0], b0, acc1[0], 0, 0, 0);
      acc1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[0][1], b1, acc1[0], 0, 0, 0);
      if constexpr (DUAL) {
        window_shift(acc1, std::integral_constant<int, 1>{}, prm1, 64, wm * 32);
        acc1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[NWIN - 1][0], b0, acc1[0], 0, 0, 0);
        acc1[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[NWIN - 1][1], b1, acc1[0], 0, 0, 0);
      }
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc1[0][r];
      i32x4 out;
      if (a.fast1 == 1) out = requant_tile16<false, 2, true>(a16, prm1, 64, wm * 32 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, false);
      else out = requant_tile16<false, 2, false>(a16, prm1, 64, wm * 32 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, a.fast1 == 2);
      const int hp = t * 32 + frow - LEAD;
      const int hr = hp / HW, col = hp - hr * HW;
      const int row = r0 - 1 + hr;
      if (hp >= 0 && hp < NHP && (unsigned)row < (unsigned)HW) {
        const int h = hr * Wp + col + 1;
        *reinterpret_cast<i32x4*>(halo + h * 64 + (((chl >> 4) ^ ((h >> 2) & 3)) << 4)) = out;
        if (a.keep_s && hr >= 1 && hr <= R)
          *reinterpret_cast<i32x4*>(a.mid1 + (px_img + (size_t)(row * HW + col)) * 64 + chl) = out;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // the halo tile is complete
  asm volatile("" ::: "memory");
  BF_STAMP(2);

  // ---- phase 2: the 3x3 over the halo tile (step = tap; a tap is the tile at a shifted pixel address) -> the expand's B tile --------
  i32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0;
  {
    int h0[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int p = (wn * 2 + j) * 32 + frow;
      if (p >= NPB) p = 0;                               // lanes beyond the band compute on pixel 0 and are never stored
      const int r = p / HW;
      h0[j] = r * Wp + (p - r * HW);
    }
    auto step = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      Afr1& cur = (t & 1) ? f1 : f0;
      Afr1& nxt = (t & 1) ? f0 : f1;
      if constexpr (t + 1 < 9) load_w2(nxt, t + 1);
      int ba[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int h = h0[j] + (t / 3) * Wp + t % 3;
        ba[j] = h * 64 + ((half ^ ((h >> 2) & 3)) << 4);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        i32x4 bf[2];
#pragma unroll
        for (int j = 0; j < 2; j++) bf[j] = *reinterpret_cast<const i32x4*>(halo + (ba[j] ^ (ks << 5)));
#pragma unroll
        for (int j = 0; j < 2; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[ks], bf[j], acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);                 // steps stay in order: the unrolled loop must not pile up loads
    };
    bf_static_for<0, 9>(step);
  }
  BF_STAMP(3);
  {
    const int* const prm2 = reinterpret_cast<const int*>(hdr2);
    const int lo_b = a.relu2 ? 0 : -128;
    auto to_mid2 = [&](auto fast_c) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_c)::value;
      int a16s[2][16];
      i32x4 outs[2], nores_j[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        nores_j[j] = nores;
#pragma unroll
        for (int r = 0; r < 16; r++) a16s[j][r] = acc[j][r];
      }
      requant_tiles16<2, false, 1, FAST>(a16s, outs, prm2, 64, wm * 32 + 4 * half, lo_b, -128, nores_j, a.dbl2 != 0, a.fast2 == 2);
      const int chl = wm * 32 + 16 * half;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int p = (wn * 2 + j) * 32 + frow;
        if (wn * 2 + j < NT1)                              // (the wave grid's eighth column tile holds no pixel)
          *reinterpret_cast<i32x4*>(mid2 + (wn * 2 + j) * 2048 + frow * 64 + (((chl >> 4) ^ ((frow >> 2) & 3)) << 4)) = outs[j];
        if (a.keep_s && p < NPB) *reinterpret_cast<i32x4*>(a.mid2 + (px_band + p) * 64 + chl) = outs[j];
      }
    };
    if (a.fast2 == 1) to_mid2(std::true_type{}); else to_mid2(std::false_type{});
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // the expand's B tile is complete
  asm volatile("" ::: "memory");
  BF_STAMP(4);

  // ---- phase 3: shortcut tile -> residual, expand + residual -> y -------------------------------------------------------------------
  // Wave w owns output channels [32 w, 32 w + 32) -- its expand and shortcut fragments stay in registers for the whole phase -- and sweeps
  // the band's SEVEN column tiles in pairs (2 + 2 + 2 + 1: every wave does the same 7 tiles of work; four 64-channel passes over the wave
  // grid of phase 2 made waves 0-2 of a SIMD row compute an eighth, empty tile's worth beside a half-loaded fourth)
  {
    const int lo_s = a.relu_s ? 0 : -128, lo_b = a.relu3 ? 0 : -128, rlo = a.add_relu ? 0 : -128;
    const int ch = wave * 32;
    const int* const pm = reinterpret_cast<const int*>(hdr3 + (ch >> 6) * H64);
    const int ro3 = ch & 63;
    const int* const ps = reinterpret_cast<const int*>(hdrS + (ch / a.tms) * (bf_hdr_bytes(NWIN, 1) * a.tms));
    const int ros = ch % a.tms;
    // (wave-uniform base + a 32-bit lane offset: the scalar-base form of global_store, no 64-bit address pair held per column tile)
    int8_t* const yb = a.y + px_band * a.y_cp + a.y_off + ch;
    int8_t* const ysb = a.ys + px_band * a.ys_cp + ch;
    const unsigned yo = (unsigned)(frow * a.y_cp + 16 * half);
    auto tiles = [&](auto nj_c, int t0) __attribute__((always_inline)) {
      constexpr int NJ = decltype(nj_c)::value;
      i32x4 rs[NJ];
      // (the wave's fragments come from the packed image again for every tile pair -- L2-resident, read-only: the shortcut's here, the
      //  expand's behind the shortcut's MFMAs -- instead of staying in registers: held, the compiler parked sixteen of the 32 in scratch,
      //  and spilled registers are DIRTY lines that reach HBM: 84 bytes per lane made 17 MB written + 16 MB fetched per launch in the PMC
      //  passes, half of the kernel's traffic)
      load_ws(wsf);
      // the shortcut convolution of the tiles (1x1 64 -> 256 on the band's input: LDS pixel 64 = band pixel 0), requantised: the residual
      {
        const int8_t* const Bs = xt + (2 + t0) * 2048;
        i32x4 b0[NJ], b1[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          b0[j] = *reinterpret_cast<const i32x4*>(Bs + j * 2048 + fr0); b1[j] = *reinterpret_cast<const i32x4*>(Bs + j * 2048 + (fr0 ^ 32));
#pragma unroll
          for (int r = 0; r < 16; r++) acc[j][r] = 0;
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf.k[0][0], b0[j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf.k[0][1], b1[j], acc[j], 0, 0, 0);
        }
        if constexpr (DUAL) {
          window_shift(acc, nj_c, ps, a.tms, ros);
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf.k[NWIN - 1][0], b0[j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf.k[NWIN - 1][1], b1[j], acc[j], 0, 0, 0);
          }
        }
        load_w3(wf);                                       // (arrives while the shortcut is requantised)
        auto to_res = [&](auto fast_c) __attribute__((always_inline)) {
          constexpr bool FAST = decltype(fast_c)::value;
          int a16s[NJ][16];
          i32x4 nores_j[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            nores_j[j] = nores;
#pragma unroll
            for (int r = 0; r < 16; r++) a16s[j][r] = acc[j][r];
          }
          requant_tiles16<NJ, false, 1, FAST>(a16s, rs, ps, a.tms, ros + 4 * half, lo_s, -128, nores_j, false, a.fast_s == 2);
        };
        if (a.fast_s == 1) to_res(std::true_type{}); else to_res(std::false_type{});
        if (a.keep_s) {
#pragma unroll
          for (int j = 0; j < NJ; j++) *reinterpret_cast<i32x4*>(ysb + (unsigned)(((t0 + j) * 32 + frow) * a.ys_cp + 16 * half)) = rs[j];
        }
      }
      // the expand (1x1 64 -> 256 on the 3x3's tile) + residual
      {
        const int8_t* const Be = mid2 + t0 * 2048;
        i32x4 b0[NJ], b1[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          b0[j] = *reinterpret_cast<const i32x4*>(Be + j * 2048 + fr0); b1[j] = *reinterpret_cast<const i32x4*>(Be + j * 2048 + (fr0 ^ 32));
#pragma unroll
          for (int r = 0; r < 16; r++) acc[j][r] = 0;
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[0][0], b0[j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[0][1], b1[j], acc[j], 0, 0, 0);
        }
        if constexpr (DUAL) {
          window_shift(acc, nj_c, pm, 64, ro3);
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[NWIN - 1][0], b0[j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf.k[NWIN - 1][1], b1[j], acc[j], 0, 0, 0);
          }
        }
        auto epilogue = [&](auto fast_c) __attribute__((always_inline)) {
          constexpr bool FAST = decltype(fast_c)::value;
          int a16s[NJ][16];
          i32x4 outs[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) a16s[j][r] = acc[j][r];
          requant_tiles16<NJ, true, 1, FAST>(a16s, outs, pm, 64, ro3 + 4 * half, lo_b, rlo, rs, false, a.fast3 == 2);
#pragma unroll
          for (int j = 0; j < NJ; j++) *reinterpret_cast<i32x4*>(yb + (yo + (unsigned)((t0 + j) * 32 * a.y_cp))) = outs[j];
        };
        if (a.fast3 == 1) epilogue(std::true_type{}); else epilogue(std::false_type{});
      }
    };
#pragma unroll 1
    for (int t0 = 0; t0 < 6; t0 += 2) tiles(std::integral_constant<int, 2>{}, t0);
    tiles(std::integral_constant<int, 1>{}, 6);
  }
  BF_STAMP(5);
#undef BF_STAMP
}

// Rows this kernel is instantiated for: what Net::bgroup_first_at admits (56 x 56 maps, 64 -> 256 | 64 -> 64 -> 64 -> 256, dense
// tiles, the 3x3 one-window, the other three all one- or all two-window) and nothing else; B images, 14 bands each.
size_t conv_bfirst_lds_bytes(int dual) { return bf_lds_bytes(dual != 0); }

int launch_conv_bfirst(const BGroupArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.tm1 != 64 || a.tm2 != 64 || a.tm3 != 64 || (a.tms != 64 && a.tms != 128) || a.dual2 || a.dual1 != a.dual3) return 1;
  const size_t lds = bf_lds_bytes(a.dual1 != 0);
  const void* fn = a.dual1 ? reinterpret_cast<const void*>(conv_bfirst_kernel<true>) : reinterpret_cast<const void*>(conv_bfirst_kernel<false>);
  if (!lds_attr_once(fn)) return -1;
  const dim3 grid(a.B * (kBfHW / kBfR));
  TF2_LAUNCH_NAME("conv_bfirst_kernel<56x56,shortcut | 64->64->64->256,R%d%s> (%d bands per image)", kBfR, a.dual1 ? ",dual" : "", kBfHW / kBfR);
  if (a.dual1) TF2_LAUNCH((conv_bfirst_kernel<true>), grid, dim3(512), lds, s, a);
  else TF2_LAUNCH((conv_bfirst_kernel<false>), grid, dim3(512), lds, s, a);
  return launch_ok() ? 0 : -1;
}

}  // namespace tf2
