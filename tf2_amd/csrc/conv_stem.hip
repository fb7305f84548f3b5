// conv_stem.hip -- the first convolution of an ImageNet network in its executed form (model_loader.cpp:244-257: the
// 7x7 / stride 2 filter runs as 3x3 / stride 1 / pad 0 over a 27-channel space-to-depth image), N <= 64 output channels,
// on the image tensor with ONE copy of x per pixel (32 bytes) instead of [x | xneg] (64 bytes) -- gfx950.
//
// Why its own kernel: this is the largest single launch of ResNet-50 (401 k output pixels per batch of 32, K = 9 taps),
// and the generic ring kernel spends it on 9 K steps of 64 bytes per pixel -- half of them the xneg copy that exists only
// for one value of x.  pe.cl:32-37 negates the ACTIVATION for a negative weight, (int8)(-x), which differs from -x only at
// x = -128: there the reference adds -128 * 2^s where signed arithmetic gives +128 * 2^s.  So
//
//     sum_ref = sum_k w_k * x_k  +  2 * sum_{k: w_k < 0} |w_k| * x128_k,       x128 = (x == -128) ? -128 : 0,
//
// and the second sum is empty for every image that has no -128 in it (mean-subtracted 0..255 data never has).  The block
// scans its own input tile once: no -128 -> signed weights on x alone, half the K bytes, half the MFMAs; otherwise the
// correction is added with two more MFMAs per step on operands derived in registers (|w_k| of the negative weights,
// x128 from x), bit-exact for any int8 image (tests/test_gpu_parity.py runs uniform int8 images through it).
//
// Structure (conv_bneck's, not the ring's): a block owns R output rows x the full width of one image; its (R+2) x W input
// rows are one contiguous NHWC range that goes global -> LDS once; all 9 x (1 or 2 windows) weight tiles (36 KiB) sit in
// LDS next to it; waves take 64-pixel tiles round-robin and sweep window by window into one accumulator set of
// 64 channels x 64 pixels; no barrier and no memory instruction but ds_read in the K loop.
// Both LDS operands are 32 bytes per row, stored as two planes of 16 bytes per row (K bytes 0-15 | 16-31): a lane's MFMA
// fragment is 16 bytes at plane[lane >> 5] + 16 * row, so the sixteen lanes a ds_read_b128 services together read 256
// contiguous bytes whatever the tap offset -- no bank conflict, no swizzle arithmetic, and a tap is an immediate offset.
//
// UNIT (PackLayer::off_unit): the conv1 rewrite leaves 54 codes per output row at the memset value 0x00 = "+x << 0"
// (model_loader.cpp:246-247) -- the same (tap, channel) positions in EVERY row -- and those shift-0 taps
// are the only reason the layer has a second exponent window.  Their contribution is one number per output PIXEL,
// S(p) = sum over the masked (tap, channel) of x, added to every channel's accumulator.  So the kernel sweeps the high
// window alone (9 steps instead of 18, 18 KiB of weights instead of 36), gathers S on the side with v_dot4c_i32_i8 on the
// B fragments it holds anyway (mask bytes 0/1 from the packed image), and finishes with acc = (acc << dshift) + S -- the
// two-window Horner result, term for term, in Z/2^32.
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

constexpr int kStemTile = 64 * 32;       // bytes of one (window, tap) weight tile: 64 rows x 32 K bytes

// bytes equal to 0x80 keep 0x80, every other byte becomes 0 (exact per byte: no carry crosses a byte)
__device__ __forceinline__ unsigned stem_x128(unsigned w) {
  const unsigned t = w ^ 0x80808080u;                          // zero byte <=> x == -128
  const unsigned m = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t;    // bit 7 of a byte clear <=> that byte of t is zero
  return ~m & 0x80808080u;
}
// |w| of the negative bytes (weights are +-2^k, k <= 6, or 0), 0 elsewhere
__device__ __forceinline__ unsigned stem_negmag(unsigned w) {
  const unsigned m = w & 0x80808080u, s = m >> 7;
  const unsigned mask = (m - s) | m;                           // 0xff in every negative byte
  return (~w & mask) + s;                                      // (~b) + 1 per byte; b != 0 there, so no carry out
}

template <int T, int N, class F>
__device__ __forceinline__ void stem_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); stem_static_for<T + 1, N>(fn); }
}

template <int NWIN, bool UNIT>
__global__ __launch_bounds__(512, 4) void conv_stem_kernel(StemArgs a) {
  static_assert(!UNIT || NWIN == 1, "the unit-tap form sweeps one window");
  TF2_PROBE_WORD(a.probe);           // timing probes (tf2_device.h; constant 0 in the product build)
  if (prb & kProbeExit0) return;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int W = a.W, OW = a.OW, R = a.R;
  const int n_h = (R + 2) * W;                              // input pixels of a full band
  const int plane = ((n_h + 63) & ~63) * 16;                  // bytes of one 16-byte plane of the input tile
  const int halo_bytes = 2 * plane;
  int8_t* const wts = lds;
  int8_t* const halo = lds + NWIN * 9 * kStemTile;
  int* const prm = reinterpret_cast<int*>(halo + halo_bytes);
  int* const flag = prm + (a.hdr_used >> 2);
  int8_t* const unit = reinterpret_cast<int8_t*>(flag + 4);          // UNIT: [9 taps][32] mask bytes

  // XCD-aware remap: the bands of one image (they share two input rows with each neighbour) on one XCD
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int img = bid / a.bands_per_img;
  const int r0 = (bid - img * a.bands_per_img) * R;
  const int rows = (a.OH - r0) < R ? (a.OH - r0) : R;        // valid output rows of this band
  const int n_px = rows * OW;
  const long long pix_base = ((long long)img * a.OH + r0) * OW;
  const int in_rows = (a.H - r0) < R + 2 ? (a.H - r0) : R + 2;
  const int n_valid = in_rows * W;                           // input pixels that exist

  // ---- prologue: header, weights, input rows -- all by LDS-DMA, one wait ------------------------------------------------------
  {
    const int8_t* hs = reinterpret_cast<const int8_t*>(a.hdr) + lane * 16;
    for (int i = wave; i * 1024 < a.hdr_used; i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hs + i * 1024), TF2_LDS_PTR(reinterpret_cast<int8_t*>(prm) + i * 1024), 16, 0, 0);
    for (int i = wave; i < NWIN * 9 * (kStemTile / 1024); i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(a.w + i * 1024 + lane * 16), TF2_LDS_PTR(wts + i * 1024), 16, 0, 0);
    // plane k of the input tile = 16-byte chunk k of every pixel: lane l of an instruction fetches pixel 64 * g + l
    const int8_t* xb = a.x + ((long long)img * a.H + r0) * W * 32;
    const int n_grp = plane >> 10;
    for (int gi = wave; gi < 2 * n_grp; gi += 8) {
      const int k = gi >= n_grp, g = gi - k * n_grp;
      const int h = g * 64 + lane;
      const int8_t* src = h < n_valid ? xb + h * 32 + k * 16 : a.zero;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + k * plane + g * 1024), 16, 0, 0);
    }
    if (tid == 0) *flag = 0;
    if (UNIT && tid < 18) reinterpret_cast<i32x4*>(unit)[tid] = reinterpret_cast<const i32x4*>(a.unit)[tid];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // any x == -128 in this block's input rows?  (padding channels and rows beyond the image are zero)
  {
    unsigned hit = 0;
    for (int o = tid * 16; o < halo_bytes; o += 512 * 16) {
      const i32x4 v = *reinterpret_cast<const i32x4*>(halo + o);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned t = (unsigned)v[i] ^ 0x80808080u;
        hit |= (t - 0x01010101u) & ~t & 0x80808080u;         // non-zero <=> some byte of t is zero
      }
    }
    if (__builtin_amdgcn_ballot_w64(hit != 0) != 0 && lane == 0) *flag = 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const bool quirk = __builtin_amdgcn_readfirstlane(*flag) != 0;
  if (prb & kProbeExit1) return;           // prologue only

  // lane-constant operand addresses: weights [window][tap][K half][64 rows][16]
  const int8_t* const a_base = wts + half * (kStemTile / 2) + (lane & 31) * 16;
  const int8_t* const b_plane = halo + half * plane;
  const int lo_bound = a.relu ? 0 : -128;
  const int n_tiles = (n_px + 63) >> 6;
  const int* const dsh = prm + kPrmWordsPerRow * 64;

  auto run = [&](auto quirk_c) {
    constexpr bool QUIRK = decltype(quirk_c)::value;
    for (int tile = wave; tile < n_tiles; tile += 8) {
      const int8_t* b0[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int p = tile * 64 + j * 32 + (lane & 31);
        if (p >= n_px) p = 0;                                // computed on pixel 0, never stored
        const int r = fast_div(p, a.ow_m, a.ow_s);
        b0[j] = b_plane + (r * W + (p - r * OW)) * 16;
      }
      i32x16 acc[2][2];
      const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      // virtual step v = (window, tap), window-major; the operands of step v + 1 are read while step v's MFMAs run
      struct Fr { i32x4 a[2], b[2]; };
      Fr f0, f1;
      auto load_fr = [&](Fr& f, int v) {
        const int t = v % 9;
#pragma unroll
        for (int rt = 0; rt < 2; rt++) f.a[rt] = *reinterpret_cast<const i32x4*>(a_base + v * kStemTile + rt * 512);
        const int tap_off = ((t / 3) * W + t % 3) * 16;
#pragma unroll
        for (int j = 0; j < 2; j++) f.b[j] = *reinterpret_cast<const i32x4*>(b0[j] + tap_off);
      };
      load_fr(f0, 0);
      int usum[2] = {0, 0};                                  // UNIT: this lane's K half of S(pixel), per pixel column tile
      auto step = [&](auto v_c) {
        constexpr int v = decltype(v_c)::value;
        Fr& cur = (v & 1) ? f1 : f0;
        Fr& nxt = (v & 1) ? f0 : f1;
        if (v + 1 < NWIN * 9) load_fr(nxt, v + 1);
        if (UNIT) {
          const i32x4 m = *reinterpret_cast<const i32x4*>(unit + v * 32 + half * 16);
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int w = 0; w < 4; w++) usum[j] = __builtin_amdgcn_sdot4(cur.b[j][w], m[w], usum[j], false);
        }
        if (v == 9) {
          // Horner step between the windows: acc <<= dshift[1][row]  (weight_pack.cpp: high window first)
#pragma unroll
          for (int rt = 0; rt < 2; rt++) {
            const int rb = rt * 32 + 4 * half;
#pragma unroll
            for (int G = 0; G < 4; G++) {
              const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 64 + rb + 8 * G);
#pragma unroll
              for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[rt][j][G * 4 + r] = (int)((unsigned)acc[rt][j][G * 4 + r] << (d[r] & 31));
            }
          }
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            if (prb & kProbeNoMfma) { if (v == 0) acc[rt][j] = zero16; asm volatile("" :: "v"(cur.a[rt]), "v"(cur.b[j])); continue; }
            acc[rt][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a[rt], cur.b[j], v == 0 ? zero16 : acc[rt][j], 0, 0, 0);
          }
        if (QUIRK) {
          i32x4 an[2], bq[2];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            an[0][i] = (int)stem_negmag((unsigned)cur.a[0][i]); an[1][i] = (int)stem_negmag((unsigned)cur.a[1][i]);
            bq[0][i] = (int)stem_x128((unsigned)cur.b[0][i]); bq[1][i] = (int)stem_x128((unsigned)cur.b[1][i]);
          }
#pragma unroll
          for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
              // |w| * x128 twice = -2 * w * x128 for the negative weights
              acc[rt][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(an[rt], bq[j], acc[rt][j], 0, 0, 0);
              acc[rt][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(an[rt], bq[j], acc[rt][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // steps stay in order: the unrolled sweep must not pile up its reads
      };
      stem_static_for<0, NWIN * 9>(step);
      if (UNIT) {
        // S = both K halves of the pixel (lanes l and l ^ 32); acc = (acc << dshift[1][row]) + S: the Horner step of the two-window
        // form with the low window's sum supplied directly
        int s_px[2];
#pragma unroll
        for (int j = 0; j < 2; j++) s_px[j] = usum[j] + __shfl_xor(usum[j], 32);
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
          const int rb = rt * 32 + 4 * half;
#pragma unroll
          for (int G = 0; G < 4; G++) {
            const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 64 + rb + 8 * G);
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
              for (int j = 0; j < 2; j++)
                acc[rt][j][G * 4 + r] = (int)(((unsigned)acc[rt][j][G * 4 + r] << (d[r] & 31)) + (unsigned)s_px[j]);
          }
        }
      }
      // ---- epilogue: pe.cl:185-203, relu.cl:54; 16 contiguous NHWC bytes per lane and 32x32 tile -----------------------------
      const i32x4 nores = {0, 0, 0, 0};
      auto epilogue = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            int a16[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a16[r] = acc[rt][j][r];
            const i32x4 out = requant_tile16<false, 2, FAST>(a16, prm, 64, rt * 32 + 4 * half, lo_bound, -128, nores, a.dbl_out != 0, a.fast == 2);
            const int p = tile * 64 + j * 32 + (lane & 31);
            const int chl = rt * 32 + 16 * half;
            if (prb & kProbeNoStore) { asm volatile("" :: "v"(out)); continue; }
            if (p < n_px && chl + 16 <= a.y_nvalid)
              *reinterpret_cast<i32x4*>(a.y + (size_t)(pix_base + p) * a.y_cp + a.y_off + chl) = out;
            __builtin_amdgcn_sched_barrier(0);
          }
      };
      if (prb & kProbeNoEpi) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1])); continue; }
      if (a.fast == 1) epilogue(std::true_type{}); else epilogue(std::false_type{});
    }
  };
  if (quirk) run(std::true_type{}); else run(std::false_type{});
}

// ---- conv_stem_pipe_kernel: the UNIT form with the epilogue of one 32 x 32 block interleaved with the MFMA sweep of the next ----
// Measured on conv_stem_kernel: halving its MFMA work (UNIT) did not move its duration -- every wave walks "sweep all four
// accumulator blocks, then requantise all four", the four waves of a SIMD do so in step, and the matrix pipe idles while the
// VALU works and vice versa (MFMA 28 % / VALU 41 % busy).  Here a wave's unit of work is ONE 32-channel x 32-pixel block:
// 9 dependent MFMAs into 16 accumulator registers, and between those MFMAs, in program order, the requantisation of the PREVIOUS
// block (two outputs per tap step), so that inside every wave the matrix pipe and the VALU are busy at the same time.  The
// nine B fragments of a pixel column tile stay in registers for both of its channel halves (and yield the unit-tap sum S);
// A fragments come from LDS one step ahead.
template <bool DBL, int MODE /* PackLayer::fast: 0 generic, 1 fast, 2 semi (requant_epilogue.h) */>
__global__ __launch_bounds__(512, 4) void conv_stem_pipe_kernel(StemArgs a) {
  constexpr bool FAST = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  long long* const adbg2 = a.dbg2;                             // tools/block_timeline.py: per-block stamps, else null
  long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, tw0 = 0;
  if (adbg2) { ts0 = (long long)__builtin_readcyclecounter(); tw0 = (long long)wall_clock64(); }
  const int W = a.W, OW = a.OW, R = a.R;
  const int n_h = (R + 2) * W;
  const int plane = ((n_h + 63) & ~63) * 16;
  const int halo_bytes = 2 * plane;
  int8_t* const wts = lds;
  int8_t* const halo = lds + 9 * kStemTile;
  int* const prm = reinterpret_cast<int*>(halo + halo_bytes);
  int* const flag = prm + (a.hdr_used >> 2);
  int8_t* const unit = reinterpret_cast<int8_t*>(flag + 4);

  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int img = bid / a.bands_per_img;
  const int r0 = (bid - img * a.bands_per_img) * R;
  const int rows = (a.OH - r0) < R ? (a.OH - r0) : R;
  const int n_px = rows * OW;
  const long long pix_base = ((long long)img * a.OH + r0) * OW;
  const int in_rows = (a.H - r0) < R + 2 ? (a.H - r0) : R + 2;
  const int n_valid = in_rows * W;

  {
    const int8_t* hs = reinterpret_cast<const int8_t*>(a.hdr) + lane * 16;
    for (int i = wave; i * 1024 < a.hdr_used; i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hs + i * 1024), TF2_LDS_PTR(reinterpret_cast<int8_t*>(prm) + i * 1024), 16, 0, 0);
    for (int i = wave; i < 9 * (kStemTile / 1024); i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(a.w + i * 1024 + lane * 16), TF2_LDS_PTR(wts + i * 1024), 16, 0, 0);
    const int8_t* xb = a.x + ((long long)img * a.H + r0) * W * 32;
    const int n_grp = plane >> 10;
    for (int gi = wave; gi < 2 * n_grp; gi += 8) {
      const int k = gi >= n_grp, g = gi - k * n_grp;
      const int h = g * 64 + lane;
      const int8_t* src = h < n_valid ? xb + h * 32 + k * 16 : a.zero;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + k * plane + g * 1024), 16, 0, 0);
    }
    if (tid == 0) *flag = 0;
    if (tid < 18) reinterpret_cast<i32x4*>(unit)[tid] = reinterpret_cast<const i32x4*>(a.unit)[tid];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (adbg2) ts1 = (long long)__builtin_readcyclecounter();
  {
    unsigned hit = 0;
    for (int o = tid * 16; o < halo_bytes; o += 512 * 16) {
      const i32x4 v = *reinterpret_cast<const i32x4*>(halo + o);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned t = (unsigned)v[i] ^ 0x80808080u;
        hit |= (t - 0x01010101u) & ~t & 0x80808080u;
      }
    }
    if (__builtin_amdgcn_ballot_w64(hit != 0) != 0 && lane == 0) *flag = 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const bool quirk = __builtin_amdgcn_readfirstlane(*flag) != 0;
  if (adbg2) ts2 = (long long)__builtin_readcyclecounter();

  const int8_t* const a_base = wts + half * (kStemTile / 2) + (lane & 31) * 16;
  const int8_t* const b_plane = halo + half * plane;
  const int lo_bound = a.relu ? 0 : -128;
  const int n_tiles = (n_px + 63) >> 6;
  const int* const dsh = prm + kPrmWordsPerRow * 64;
  const rq_i32x4* const rowp_all = reinterpret_cast<const rq_i32x4*>(prm);

  auto run = [&](auto quirk_c) {
    constexpr bool QUIRK = decltype(quirk_c)::value;
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the block whose requantisation is pending: accumulators (replaced in place by the clamped outputs), pixel, channel half
    i32x16 pend = zero16;
    int pend_p = 0, pend_rt = 0;
    bool has_pend = false;
    // outputs 2t, 2t+1 of the pending block (t = 0..7): row = rt * 32 + 4 * half + 8 * (k / 4) + k % 4
    auto epi_rows = [&](int t) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int k = 2 * t + u;
        const rq_i32x4 pr = rowp_all[pend_rt * 32 + 4 * half + 8 * (k >> 2) + (k & 3)];
        const long long b64 = (long long)(((unsigned long long)(unsigned)pr[3] << 32) | (unsigned)pr[2]);
        int y, kd;
        if (FAST) {                                            // requant_epilogue.h: rows proven at pack time
          const long long pp = (long long)pend[k] * (long long)pr[1] + b64;
          y = (int)(pp >> 32) >> (kAlphaInflat + kInflat - 32);
          kd = pr[0];
        } else {                                               // generic / semi rows: 32-bit wrap of v kept (pe.cl:191-193)
          const int lo = prm[4 * 64 + pend_rt * 32 + 4 * half + 8 * (k >> 2) + (k & 3)];
          const int v = (int)((unsigned)pr[0] + ((unsigned)pend[k] << (lo & 31)));
          const long long pp = (long long)v * (long long)pr[1] + b64;
          if (MODE == 2) y = (int)(pp >> 32) >> (kAlphaInflat + kInflat - 32);
          else { const int x = (int)(pp >> kAlphaInflat); y = __builtin_elementwise_add_sat(x, 1 << (kInflat - 1)) >> kInflat; }
          kd = lo >> 8;
        }
        int c;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(y), "s"(lo_bound), "v"(127));
        if (DBL) c = (int)(((unsigned)c << ((unsigned)kd >> 31)) + (unsigned)kd);
        pend[k] = c;
      }
    };
    auto epi_store = [&]() {
      unsigned d[4];
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const unsigned p01 = __builtin_amdgcn_perm((unsigned)pend[4 * G + 1], (unsigned)pend[4 * G], 0x0c0c0400u);
        const unsigned p23 = __builtin_amdgcn_perm((unsigned)pend[4 * G + 3], (unsigned)pend[4 * G + 2], 0x0c0c0400u);
        d[G] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
      }
      auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
      const i32x4 out = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      const int chl = pend_rt * 32 + 16 * half;
      if (pend_p < n_px && chl + 16 <= a.y_nvalid)
        *reinterpret_cast<i32x4*>(a.y + (size_t)(pix_base + pend_p) * a.y_cp + a.y_off + chl) = out;
    };
    for (int tile = wave; tile < n_tiles; tile += 8) {
#pragma unroll 1
      for (int j = 0; j < 2; j++) {
        const int p_raw = tile * 64 + j * 32 + (lane & 31);
        const int p = p_raw < n_px ? p_raw : 0;                 // computed on pixel 0, never stored
        const int r = fast_div(p, a.ow_m, a.ow_s);
        const int8_t* const b0 = b_plane + (r * W + (p - r * OW)) * 16;
        // S = the unit taps' sum of this pixel column tile: gathered with v_dot4c on the B fragments of the rt = 0 sweep (this
        // lane's K half), both halves added once that sweep is over; the rt = 1 sweep reads the fragments again (LDS has the
        // bandwidth, the register file has not: nine resident fragments spilled)
        int us = 0, s_px = 0;
#pragma unroll 1
        for (int rt = 0; rt < 2; rt++) {
          i32x16 acc = zero16;
          i32x4 af = *reinterpret_cast<const i32x4*>(a_base + rt * 512);
          i32x4 bfr = *reinterpret_cast<const i32x4*>(b0);
#pragma unroll
          for (int t = 0; t < 9; t++) {
            i32x4 af_next = af, bf_next = bfr;
            if (t + 1 < 9) {
              af_next = *reinterpret_cast<const i32x4*>(a_base + (t + 1) * kStemTile + rt * 512);
              bf_next = *reinterpret_cast<const i32x4*>(b0 + (((t + 1) / 3) * W + (t + 1) % 3) * 16);
            }
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bfr, acc, 0, 0, 0);
            if (QUIRK) {
              i32x4 an, bq;
#pragma unroll
              for (int i = 0; i < 4; i++) { an[i] = (int)stem_negmag((unsigned)af[i]); bq[i] = (int)stem_x128((unsigned)bfr[i]); }
              acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(an, bq, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(an, bq, acc, 0, 0, 0);
            }
            if (rt == 0) {
              const i32x4 m = *reinterpret_cast<const i32x4*>(unit + t * 32 + half * 16);
#pragma unroll
              for (int w = 0; w < 4; w++) us = __builtin_amdgcn_sdot4(bfr[w], m[w], us, false);
            }
            if (has_pend) { if (t < 8) epi_rows(t); else epi_store(); }
            af = af_next; bfr = bf_next;
            __builtin_amdgcn_sched_barrier(0);                 // one MFMA, then its share of the pending block's requantisation
          }
          if (rt == 0) s_px = us + __shfl_xor(us, 32);
          // acc = (acc << dshift[1][row]) + S: the two-window Horner result; this block is now the pending one
          {
            const int rb = rt * 32 + 4 * half;
#pragma unroll
            for (int G = 0; G < 4; G++) {
              const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 64 + rb + 8 * G);
#pragma unroll
              for (int r4 = 0; r4 < 4; r4++)
                acc[G * 4 + r4] = (int)(((unsigned)acc[G * 4 + r4] << (d[r4] & 31)) + (unsigned)s_px);
            }
          }
          pend = acc; pend_p = p_raw; pend_rt = rt; has_pend = true;
        }
      }
    }
    if (adbg2) ts3 = (long long)__builtin_readcyclecounter();
    if (has_pend) {
#pragma unroll
      for (int t = 0; t < 8; t++) epi_rows(t);
      epi_store();
    }
  };
  if (quirk) run(std::true_type{}); else run(std::false_type{});
  if (adbg2 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long* d = adbg2 + (size_t)blockIdx.x * 8;
    d[0] = ts0; d[1] = (long long)__builtin_readcyclecounter(); d[4] = ts1; d[5] = ts2; d[6] = ts3;
    d[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    d[3] = tw0; d[7] = (long long)wall_clock64();
  }
}

// ---- conv_stem_pool_kernel: first layer + its 3x3 / stride 2 / pad 1 max pool in one launch ---------------------------------
// The reference runs the pool as a pipeline stage behind the PE array (pool.cl:152-260: running 3-max over columns, then over
// rows, out-of-range taps = 0; pool_tail.cl:91-216: every second row / column) -- the conv map never goes to memory.  Here
// a block owns pk pooled rows x the full width of one image x ONE 32-channel half of the layer: it computes the 2 pk + 1 conv
// rows those pooled rows look at (one row of overlap with the neighbouring band is recomputed), requantised blocks go into an
// LDS tile [conv row][pixel][32 channels] instead of HBM, and after one barrier every thread reduces 3 x 3 windows of 16 channel
// bytes and stores the pooled NHWC bytes.  The channel split halves the tile (25 KiB) and the weight image (9 KiB) so that two
// blocks still share a CU (68 KiB each); the K loop / requantisation pipeline is conv_stem_pipe_kernel's (UNIT form).
// 25.7 MB of conv map per batch of 32 neither written nor read back, one launch less.
template <int MODE /* PackLayer::fast: 0 generic, 1 fast, 2 semi */>
__global__ __launch_bounds__(512, 4) void conv_stem_pool_kernel(StemArgs a) {
  constexpr bool FAST = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  long long* const adbg2 = a.dbg2;
  long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, tw0 = 0;
  if (adbg2) { ts0 = (long long)__builtin_readcyclecounter(); tw0 = (long long)wall_clock64(); }
  const int W = a.W, OW = a.OW, OH = a.OH;
  const int R = 2 * a.pk + 1;                                  // conv rows of a band
  const int n_h = (R + 2) * W;
  const int plane = ((n_h + 63) & ~63) * 16;
  const int halo_bytes = 2 * plane;
  constexpr int kHalfTile = 32 * 32;                           // bytes of one (tap) weight tile of a channel half: [K half][32 rows][16]
  int8_t* const wts = lds;
  int8_t* const halo = lds + 9 * kHalfTile;
  int8_t* const ct = halo + halo_bytes;                        // conv tile [R][OW][32]
  int* const prm = reinterpret_cast<int*>(ct + R * OW * 32);
  int* const flag = prm + (a.hdr_used >> 2);
  int8_t* const unit = reinterpret_cast<int8_t*>(flag + 4);

  // (image, band, channel half).  Consecutive block ids go to consecutive XCDs, each with its own L2: hand every XCD a CONTIGUOUS
  // run of (band, half) so that the two halves of a band and the bands of an image -- which read the same input rows -- meet in
  // one L2 (round 3 measured 39.5 MB fetched for a 13.3 MB input with the halves on neighbouring XCDs).
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int rt = bid & 1;
  const int band_lin = bid >> 1;
  const int img = band_lin / a.bands_per_img;
  const int pj0 = (band_lin - img * a.bands_per_img) * a.pk;   // first pooled row
  const int cr0 = 2 * pj0 - 1;                                 // first conv row (may be -1: above the map)
  const int n_px = R * OW;                                     // band pixels, conv rows cr0 .. cr0 + R - 1 (rows outside the map are never read by the pool)
  const int ir0 = cr0 < 0 ? 0 : cr0;                           // first input row that exists
  const int row_shift = ir0 - cr0;                             // 1 for the first band: tile row 0 is conv row -1
  int in_rows = (cr0 + R + 2 <= a.H ? cr0 + R + 2 : a.H) - ir0;
  if (in_rows < 0) in_rows = 0;
  const int n_valid = in_rows * W;
  const unsigned q128_word = a.q128 ? a.q128[img] : 0u;       // (read ahead of the prologue's wait)

  {
    const int8_t* hs = reinterpret_cast<const int8_t*>(a.hdr) + lane * 16;
    for (int i = wave; i * 1024 < a.hdr_used; i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hs + i * 1024), TF2_LDS_PTR(reinterpret_cast<int8_t*>(prm) + i * 1024), 16, 0, 0);
    // this half's weights: image [tap][K half][64 rows][16] -> LDS [tap][K half][32 rows][16], one tap per DMA instruction
    for (int t = wave; t < 9; t += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(a.w + t * kStemTile + half * (kStemTile / 2) + rt * 512 + (lane & 31) * 16),
                                       TF2_LDS_PTR(wts + t * kHalfTile), 16, 0, 0);
    // input rows ir0 .. : tile row i holds input row cr0 + i; for the first band tile row 0 (conv row -1's first input row) is zeros
    const int8_t* xb = a.x + ((long long)img * a.H + ir0) * W * 32;
    const int n_grp = plane >> 10;
    const int skip = row_shift * W;                            // tile pixels in front of the first existing input row
    for (int gi = wave; gi < 2 * n_grp; gi += 8) {
      const int k = gi >= n_grp, g = gi - k * n_grp;
      const int h = g * 64 + lane - skip;
      const int8_t* src = (h >= 0 && h < n_valid) ? xb + h * 32 + k * 16 : a.zero;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + k * plane + g * 1024), 16, 0, 0);
    }
    if (tid == 0) *flag = 0;
    if (tid < 18) reinterpret_cast<i32x4*>(unit)[tid] = reinterpret_cast<const i32x4*>(a.unit)[tid];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (adbg2) ts1 = (long long)__builtin_readcyclecounter();
  bool quirk;
  if (a.q128) {
    // the step's input preparation has told, per image, whether an element is -128 (PrepArgs::q128): no scan of the tile, no second barrier
    quirk = __builtin_amdgcn_readfirstlane((int)q128_word) != 0;
  } else {
    unsigned hit = 0;
    for (int o = tid * 16; o < halo_bytes; o += 512 * 16) {
      const i32x4 v = *reinterpret_cast<const i32x4*>(halo + o);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned t = (unsigned)v[i] ^ 0x80808080u;
        hit |= (t - 0x01010101u) & ~t & 0x80808080u;
      }
    }
    if (__builtin_amdgcn_ballot_w64(hit != 0) != 0 && lane == 0) *flag = 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    quirk = __builtin_amdgcn_readfirstlane(*flag) != 0;
  }
  if (adbg2) ts2 = (long long)__builtin_readcyclecounter();

  const int8_t* const a_base = wts + half * 512 + (lane & 31) * 16;
  const int8_t* const b_plane = halo + half * plane;
  const int lo_bound = a.relu ? 0 : -128;
  const int n_blk = (n_px + 31) >> 5;                          // 32-pixel blocks of the band
  const int* const dsh = prm + kPrmWordsPerRow * 64;
  const rq_i32x4* const rowp = reinterpret_cast<const rq_i32x4*>(prm) + rt * 32 + 4 * half;

  auto run = [&](auto quirk_c) {
    constexpr bool QUIRK = decltype(quirk_c)::value;
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    i32x16 pend = zero16;
    int pend_p = 0;
    bool has_pend = false;
    auto epi_rows = [&](int t) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int k = 2 * t + u;
        const rq_i32x4 pr = rowp[8 * (k >> 2) + (k & 3)];
        const long long b64 = (long long)(((unsigned long long)(unsigned)pr[3] << 32) | (unsigned)pr[2]);
        int y;
        if (FAST) {
          const long long pp = (long long)pend[k] * (long long)pr[1] + b64;
          y = (int)(pp >> 32) >> (kAlphaInflat + kInflat - 32);
        } else {                                               // the low window's base is 0: v = bias + acc (pe.cl:176-180 wrap kept)
          const int v = (int)((unsigned)pr[0] + (unsigned)pend[k]);
          const long long pp = (long long)v * (long long)pr[1] + b64;
          if (MODE == 2) y = (int)(pp >> 32) >> (kAlphaInflat + kInflat - 32);
          else { const int x = (int)(pp >> kAlphaInflat); y = __builtin_elementwise_add_sat(x, 1 << (kInflat - 1)) >> kInflat; }
        }
        int c;
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(y), "s"(lo_bound), "v"(127));
        pend[k] = c;
      }
    };
    auto epi_store = [&]() {                                   // 16 contiguous channel bytes of one pixel -> the LDS conv tile
      unsigned d[4];
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const unsigned p01 = __builtin_amdgcn_perm((unsigned)pend[4 * G + 1], (unsigned)pend[4 * G], 0x0c0c0400u);
        const unsigned p23 = __builtin_amdgcn_perm((unsigned)pend[4 * G + 3], (unsigned)pend[4 * G + 2], 0x0c0c0400u);
        d[G] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
      }
      auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
      const i32x4 out = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      if (pend_p < n_px) *reinterpret_cast<i32x4*>(ct + pend_p * 32 + 16 * half) = out;
    };
    for (int blk = wave; blk < n_blk; blk += 8) {
      const int p_raw = blk * 32 + (lane & 31);
      const int p = p_raw < n_px ? p_raw : 0;
      const int r = fast_div(p, a.ow_m, a.ow_s);
      const int8_t* const b0 = b_plane + (r * W + (p - r * OW)) * 16;
      i32x16 acc = zero16;
      int us = 0;
      i32x4 af = *reinterpret_cast<const i32x4*>(a_base);
      i32x4 bfr = *reinterpret_cast<const i32x4*>(b0);
#pragma unroll
      for (int t = 0; t < 9; t++) {
        i32x4 af_next = af, bf_next = bfr;
        if (t + 1 < 9) {
          af_next = *reinterpret_cast<const i32x4*>(a_base + (t + 1) * kHalfTile);
          bf_next = *reinterpret_cast<const i32x4*>(b0 + (((t + 1) / 3) * W + (t + 1) % 3) * 16);
        }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bfr, acc, 0, 0, 0);
        if (QUIRK) {
          i32x4 an, bq;
#pragma unroll
          for (int i = 0; i < 4; i++) { an[i] = (int)stem_negmag((unsigned)af[i]); bq[i] = (int)stem_x128((unsigned)bfr[i]); }
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(an, bq, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(an, bq, acc, 0, 0, 0);
        }
        {
          const i32x4 m = *reinterpret_cast<const i32x4*>(unit + t * 32 + half * 16);
#pragma unroll
          for (int w = 0; w < 4; w++) us = __builtin_amdgcn_sdot4(bfr[w], m[w], us, false);
        }
        if (has_pend) { if (t < 8) epi_rows(t); else epi_store(); }
        af = af_next; bfr = bf_next;
        __builtin_amdgcn_sched_barrier(0);
      }
      const int s_px = us + __shfl_xor(us, 32);
      {
        const int rb = rt * 32 + 4 * half;
#pragma unroll
        for (int G = 0; G < 4; G++) {
          const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 64 + rb + 8 * G);
#pragma unroll
          for (int r4 = 0; r4 < 4; r4++)
            acc[G * 4 + r4] = (int)(((unsigned)acc[G * 4 + r4] << (d[r4] & 31)) + (unsigned)s_px);
        }
      }
      pend = acc; pend_p = p_raw; has_pend = true;
    }
    if (adbg2) ts3 = (long long)__builtin_readcyclecounter();
    if (has_pend) {
#pragma unroll
      for (int t = 0; t < 8; t++) epi_rows(t);
      epi_store();
    }
  };
  if (quirk) run(std::true_type{}); else run(std::false_type{});

  // ---- the pool: pooled pixel (pj, pi) = max over conv rows 2 pj - 1 .. 2 pj + 1, columns 2 pi - 1 .. 2 pi + 1; a tap outside the
  // map counts as 0 (pool.cl:119-140 feeds zeros beyond the valid area; the window maximum starts from them)
  __syncthreads();
  const int n_items = a.pk * a.PW * 2;                         // (pooled row, pooled column, 16-channel group)
  for (int it = tid; it < n_items; it += 512) {
    const int g = it & 1;
    const int pjl = fast_div(it >> 1, a.pw_m, a.pw_s);
    const int pi = (it >> 1) - pjl * a.PW;
    const int pj = pj0 + pjl;
    if (pj >= a.PH) continue;
    if (a.relu) {
      // after ReLU every byte is 0..127: the signed maximum is the unsigned one, the out-of-range tap value 0 is the identity, and
      // four bytes take four instructions -- even bytes by v_and, odd bytes by v_perm, two v_pk_max_u16 -- instead of eight
      using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
      u16x2 me[4], mo[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { me[q] = u16x2{0, 0}; mo[q] = u16x2{0, 0}; }
#pragma unroll
      for (int dy = 0; dy < 3; dy++) {
        const int cr = 2 * pj - 1 + dy;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
          const int cc = 2 * pi - 1 + dx;
          if ((unsigned)cr < (unsigned)OH && (unsigned)cc < (unsigned)OW) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(ct + ((cr - cr0) * OW + cc) * 32 + g * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const unsigned e = (unsigned)v[q] & 0x00ff00ffu;
              const unsigned o = __builtin_amdgcn_perm(0u, (unsigned)v[q], 0x0c030c01u);
              me[q] = __builtin_elementwise_max(me[q], __builtin_bit_cast(u16x2, e));
              mo[q] = __builtin_elementwise_max(mo[q], __builtin_bit_cast(u16x2, o));
            }
          }
        }
      }
      i32x4 o;
#pragma unroll
      for (int q = 0; q < 4; q++) o[q] = (int)(__builtin_bit_cast(unsigned, me[q]) | (__builtin_bit_cast(unsigned, mo[q]) << 8));
      *reinterpret_cast<i32x4*>(a.yp + ((size_t)((long long)img * a.PH + pj) * a.PW + pi) * a.yp_cp + a.yp_off + rt * 32 + g * 16) = o;
      continue;
    }
    int m[16];
#pragma unroll
    for (int q = 0; q < 16; q++) m[q] = -128;
    bool any_oob = false;
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
      const int cr = 2 * pj - 1 + dy;                          // conv row; tile row = cr - cr0
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const int cc = 2 * pi - 1 + dx;
        if ((unsigned)cr < (unsigned)OH && (unsigned)cc < (unsigned)OW) {
          const i32x4 v = *reinterpret_cast<const i32x4*>(ct + ((cr - cr0) * OW + cc) * 32 + g * 16);
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const int x = (int)(signed char)((v[q >> 2] >> (8 * (q & 3))) & 0xff);
            m[q] = x > m[q] ? x : m[q];
          }
        } else any_oob = true;
      }
    }
    if (any_oob) {
#pragma unroll
      for (int q = 0; q < 16; q++) m[q] = m[q] > 0 ? m[q] : 0;
    }
    i32x4 o;
#pragma unroll
    for (int q = 0; q < 4; q++)
      o[q] = (m[4 * q] & 0xff) | ((m[4 * q + 1] & 0xff) << 8) | ((m[4 * q + 2] & 0xff) << 16) | ((m[4 * q + 3] & 0xff) << 24);
    *reinterpret_cast<i32x4*>(a.yp + ((size_t)((long long)img * a.PH + pj) * a.PW + pi) * a.yp_cp + a.yp_off + rt * 32 + g * 16) = o;
  }
  if (adbg2 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long* d = adbg2 + (size_t)blockIdx.x * 8;
    d[0] = ts0; d[1] = (long long)__builtin_readcyclecounter(); d[4] = ts1; d[5] = ts2; d[6] = ts3;
    d[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    d[3] = tw0; d[7] = (long long)wall_clock64();
  }
}

size_t conv_stem_pool_lds_bytes(int pk, int W, int OW, size_t hdr_used) {
  const int R = 2 * pk + 1;
  const int n_h = (R + 2) * W;
  return (size_t)9 * 1024 + (size_t)((n_h + 63) & ~63) * 32 + (size_t)R * OW * 32 + hdr_used + 64 + 9 * 32;
}

size_t conv_stem_lds_bytes(int nwin, int R, int W, size_t hdr_used) {
  const int n_h = (R + 2) * W;
  return (size_t)nwin * 9 * kStemTile + (size_t)((n_h + 63) & ~63) * 32 + hdr_used + 64 + 9 * 32;
}

int launch_conv_stem(const StemArgs& a, int nwin, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const bool unit = a.unit != nullptr;                 // the image then holds the high window alone
  const int nw = unit ? 1 : nwin;
  const size_t lds = conv_stem_lds_bytes(nw, a.R, a.W, (size_t)a.hdr_used);
  if (lds > 160 * 1024 || (nwin != 1 && nwin != 2) || (unit && nwin != 2)) return 1;
  const int grid = a.B * a.bands_per_img;
#define TF2_STEM(NW_, U_, label) do { auto fn = conv_stem_kernel<NW_, U_>; if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1; \
                                      TF2_LAUNCH_NAME("conv_stem_kernel<%s>", label); TF2_LAUNCH(fn, dim3(grid), dim3(512), lds, s, a); } while (0)
  if (a.yp) {                                           // fused 3x3 / 2 max pool (net.hip checked the shape)
    if (!unit) return 1;
    const size_t ldp = conv_stem_pool_lds_bytes(a.pk, a.W, a.OW, (size_t)a.hdr_used);
    if (ldp > 160 * 1024) return 1;
    const int gridp = a.B * a.bands_per_img * 2;
#define TF2_STEMQ(M_, label) do { auto fn = conv_stem_pool_kernel<M_>; if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1; \
                                  TF2_LAUNCH_NAME("conv_stem_pool_kernel<%s>", label); TF2_LAUNCH(fn, dim3(gridp), dim3(512), ldp, s, a); } while (0)
    if (a.fast == 1) TF2_STEMQ(1, "fast"); else if (a.fast == 2) TF2_STEMQ(2, "semi"); else TF2_STEMQ(0, "generic");
#undef TF2_STEMQ
    return launch_ok() ? 0 : -1;
  }
  if (unit) {
#define TF2_STEMP(D_, M_) do { auto fn = conv_stem_pipe_kernel<D_, M_>; if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1; \
                               TF2_LAUNCH_NAME("conv_stem_pipe_kernel<%s,%s>", D_ ? "doubled" : "plain", M_ == 1 ? "fast" : M_ == 2 ? "semi" : "generic"); \
                               TF2_LAUNCH(fn, dim3(grid), dim3(512), lds, s, a); } while (0)
    if (a.dbl_out) { if (a.fast == 1) TF2_STEMP(true, 1); else if (a.fast == 2) TF2_STEMP(true, 2); else TF2_STEMP(true, 0); }
    else { if (a.fast == 1) TF2_STEMP(false, 1); else if (a.fast == 2) TF2_STEMP(false, 2); else TF2_STEMP(false, 0); }
#undef TF2_STEMP
  }
  else if (unit) TF2_STEM(1, true, "1 window + unit taps");
  else if (nwin == 2) TF2_STEM(2, false, "2 windows");
  else TF2_STEM(1, false, "1 window");
#undef TF2_STEM
  return launch_ok() ? 0 : -1;
}

}  // namespace tf2
