// conv_stem.hip -- the first convolution of an ImageNet network in its executed form (model_loader.cpp:244-257: the
// 7x7 / stride 2 filter runs as 3x3 / stride 1 / pad 0 over a 27-channel space-to-depth image), N <= 64 output channels,
// on the image tensor with ONE copy of x per pixel (32 bytes) instead of [x | xneg] (64 bytes) -- gfx950.
//
// Why its own kernel: this is the largest single launch of ResNet-50 (401 k output pixels per batch of 32, K = 9 taps),
// and the generic ring kernel spends it on 9 K steps of 64 bytes per pixel -- half of them the xneg copy that exists only
// for one value of x.  pe.cl:32-37 negates the ACTIVATION for a negative weight, (int8)(-x), which differs from -x only at
// x = -128: there the reference adds -128 * 2^s where signed arithmetic gives +128 * 2^s.  So
//
//     sum_ref = sum_k w_k * x_k  -  2 * sum_{k: w_k < 0} |w_k| * x128_k,       x128 = (x == -128) ? -128 : 0,
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

template <int NWIN>
__global__ __launch_bounds__(512, 4) void conv_stem_kernel(StemArgs a) {
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
        const int r = p / OW;
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
      auto step = [&](auto v_c) {
        constexpr int v = decltype(v_c)::value;
        Fr& cur = (v & 1) ? f1 : f0;
        Fr& nxt = (v & 1) ? f0 : f1;
        if (v + 1 < NWIN * 9) load_fr(nxt, v + 1);
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
          for (int j = 0; j < 2; j++)
            acc[rt][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.a[rt], cur.b[j], v == 0 ? zero16 : acc[rt][j], 0, 0, 0);
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
            const i32x4 out = requant_tile16<false, 2, FAST>(a16, prm, 64, rt * 32 + 4 * half, lo_bound, -128, nores);
            const int p = tile * 64 + j * 32 + (lane & 31);
            const int chl = rt * 32 + 16 * half;
            if (p < n_px && chl + 16 <= a.y_nvalid)
              *reinterpret_cast<i32x4*>(a.y + (size_t)(pix_base + p) * a.y_cp + a.y_off + chl) = out;
            __builtin_amdgcn_sched_barrier(0);
          }
      };
      if (a.fast) epilogue(std::true_type{}); else epilogue(std::false_type{});
    }
  };
  if (quirk) run(std::true_type{}); else run(std::false_type{});
}

size_t conv_stem_lds_bytes(int nwin, int R, int W, size_t hdr_used) {
  const int n_h = (R + 2) * W;
  return (size_t)nwin * 9 * kStemTile + (size_t)((n_h + 63) & ~63) * 32 + hdr_used + 64;
}

int launch_conv_stem(const StemArgs& a, int nwin, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = conv_stem_lds_bytes(nwin, a.R, a.W, (size_t)a.hdr_used);
  if (lds > 160 * 1024 || (nwin != 1 && nwin != 2)) return 1;
  const int grid = a.B * a.bands_per_img;
  if (nwin == 2) {
    auto fn = conv_stem_kernel<2>;
    if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
    TF2_LAUNCH_NAME("conv_stem_kernel<%d windows>", nwin); TF2_LAUNCH(fn, dim3(grid), dim3(512), lds, s, a);
  } else {
    auto fn = conv_stem_kernel<1>;
    if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
    TF2_LAUNCH_NAME("conv_stem_kernel<%d windows>", nwin); TF2_LAUNCH(fn, dim3(grid), dim3(512), lds, s, a);
  }
  return launch_ok() ? 0 : -1;
}

}  // namespace tf2
