// conv_fc.hip -- a layer whose whole input is ONE filter window per image (a k x k / pad 0 convolution of a k x k map, or a 1 x 1 layer
// of a 1 x 1 map: VGG16's fc6 7x7x512 -> 4096 and fc7) at batch <= 32: a weight stream (gfx950).
//
// Such a layer is [Np x K] x [K x 32]: 32 pixels = ONE column tile, K = 25088 for fc6 -- 205 MB of two-window weight tiles for 6.6 GMAC.
// conv_mfma_sk gives it Np / 64 = 64 blocks (each splits K over its eight waves through LDS rings of two stages): 64 CUs pull
// 1.95 TB/s, 105 us, the largest launch of VGG16.  Here the stream is split over the whole chip, K included:
//
//   fc_partial_kernel  grid (Np / 128, KSPLIT), four waves per block, a wave = 32 output channels x one K slice: weight fragments
//                      (16 contiguous bytes of a row of the packed tile per lane) and activation fragments (16 bytes of image
//                      lane & 31) straight from memory into v_mfma_i32_32x32x32_i8, no LDS, no barrier; its 32 x 32 int32 partial sums
//                      (per window) go to a scratch area of the workspace;
//   fc_finish_kernel   one thread per (channel, image): adds the KSPLIT partials (Z/2^32: any order), combines the two windows
//                      ((hi << dshift[1]) + lo), requantises with the header rows (the arithmetic of requant_epilogue.h, one output at
//                      a time) and stores the int8 activation.
//
// Reference semantics: pe.cl:27-43 (shift-accumulate), pe.cl:185-203 (requant), relu.cl:54.  Bit-identical to conv_mfma_sk.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "tf2_internal.h"
#include "tf2_device.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

template <bool DUAL>
__global__ __launch_bounds__(256) void fc_partial_kernel(FcArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int n0 = (blockIdx.x * 4 + wave) * 32;             // this wave's 32 output channels
  if (n0 >= a.Np) return;
  const int tms = a.tm == 128 ? 7 : 6;
  const int mt = n0 >> tms, ro = n0 & ((1 << tms) - 1);
  const int wins = DUAL ? 2 : 1;
  const int s0 = blockIdx.y * a.slabs_per_split;
  const int s1 = s0 + a.slabs_per_split < a.nslab ? s0 + a.slabs_per_split : a.nslab;
  // (images beyond the batch read image 0: their columns are never looked at)
  const int8_t* xb = a.x + (size_t)(col < a.B ? col : 0) * a.K + half * 16;
  const int8_t* wb = a.w + ((((size_t)mt * a.nslab) * wins) << tms) * 64 + (size_t)(ro + col) * 64 + half * 16;
  const size_t ent = (size_t)wins << (tms + 6);            // bytes of one (m-tile, slab) entry
  const size_t winb = (size_t)1 << (tms + 6);              // bytes of one window of it
  i32x16 hi, lo;
#pragma unroll
  for (int r = 0; r < 16; r++) { hi[r] = 0; lo[r] = 0; }
  // GS slabs' fragments in flight per wave (the loads of a group are issued before its first MFMA)
  constexpr int GS = 8;
  int s = s0;
  for (; s + GS <= s1; s += GS) {
    i32x4 b0[GS], b1[GS], h0[GS], h1[GS], l0[DUAL ? GS : 1], l1[DUAL ? GS : 1];
#pragma unroll
    for (int u = 0; u < GS; u++) {
      const int8_t* wp = wb + (size_t)(s + u) * ent;
      const int8_t* xp = xb + (size_t)(s + u) * 64;
      b0[u] = *reinterpret_cast<const i32x4*>(xp); b1[u] = *reinterpret_cast<const i32x4*>(xp + 32);
      h0[u] = *reinterpret_cast<const i32x4*>(wp); h1[u] = *reinterpret_cast<const i32x4*>(wp + 32);
      if (DUAL) { l0[u] = *reinterpret_cast<const i32x4*>(wp + winb); l1[u] = *reinterpret_cast<const i32x4*>(wp + winb + 32); }
    }
    __builtin_amdgcn_sched_barrier(0);                     // (else the scheduler sinks the loads next to their MFMAs: ~8 in flight)
#pragma unroll
    for (int u = 0; u < GS; u++) {
      hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(h0[u], b0[u], hi, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(h1[u], b1[u], hi, 0, 0, 0);
      if (DUAL) {
        lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(l0[u], b0[u], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(l1[u], b1[u], lo, 0, 0, 0);
      }
    }
  }
  for (; s < s1; s++) {
    const int8_t* wp = wb + (size_t)s * ent;
    const int8_t* xp = xb + (size_t)s * 64;
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(xp), b1 = *reinterpret_cast<const i32x4*>(xp + 32);
    const i32x4 h0 = *reinterpret_cast<const i32x4*>(wp), h1 = *reinterpret_cast<const i32x4*>(wp + 32);
    hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(h0, b0, hi, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(h1, b1, hi, 0, 0, 0);
    if (DUAL) {
      const i32x4 l0 = *reinterpret_cast<const i32x4*>(wp + winb), l1 = *reinterpret_cast<const i32x4*>(wp + winb + 32);
      lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(l0, b0, lo, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(l1, b1, lo, 0, 0, 0);
    }
  }
  // C/D layout: register k of this lane = row 8 * (k / 4) + 4 * half + k % 4 of column col
  int* const p0 = a.part + ((size_t)(blockIdx.y * wins) * a.Np + n0) * 32 + col;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int row = 8 * (k >> 2) + 4 * half + (k & 3);
    p0[(size_t)row * 32] = hi[k];
    if (DUAL) p0[((size_t)a.Np + row) * 32] = lo[k];
  }
}

// ---- the same stream with the filters as 4-BIT CODES in HBM (PackLayer::fc4, weight_pack.cpp; 4bit_data_format.txt) -----------------
// A lane's 16 bytes of a slab are the 32 codes {sign << 3 | e} of its row's K half (both K steps); they are expanded to the int8 window
// values in registers, in front of the MFMAs: per word of eight codes -- the exponent codes of the low / high nibbles as byte selectors of
// v_perm_b32 into the row's 8-byte table "window value of e" (one table per window and input-channel class; the class of a K position
// picks between two permutes with v_bfi_b32), then the sign applied to all four bytes at once.  Two's complement would be (v ^ m) + s
// with m = 0xff / s = 1 in the bytes of negative weights -- but a negative weight is ZERO in the window it does not belong to, and
// (0 ^ 0xff) + 1 carries into the neighbouring byte (the first version did exactly that: fc6 off by one in 3 % of its outputs).  So
// the operand is the ONE'S complement v ^ m = -v - 1, and the missing "+1 per negative weight" is one more MFMA per K step with the
// sign bytes s themselves as its operand: sum((v ^ m) x) + sum(s x) = sum(+-v x), for v = 0 as well, exact in Z/2^32; its accumulator
// is added to both windows' once, at the end.  ~25 VALU instructions per eight weights and two windows: the kernel
// stays HBM-bound (fc6: 51 MB of codes instead of 205 MB of window tiles).  Image chunks of 32 (batch > 32) are grid.z: a layer stored
// this way has no int8 tiles any other kernel could run on.
struct Fc4Lut { unsigned lo, hi; };      // eight table bytes: e = 0..3 | e = 4..7

template <int NCLS>
__device__ __forceinline__ unsigned fc4_dec(unsigned e, unsigned cm, unsigned m, const Fc4Lut (&t)[2]) {
  unsigned v = __builtin_amdgcn_perm(t[0].hi, t[0].lo, e);
  if constexpr (NCLS == 2) {
    const unsigned v1 = __builtin_amdgcn_perm(t[1].hi, t[1].lo, e);
    v = (v1 & cm) | (v & ~cm);
  }
  return v ^ m;                                           // one's complement for negative weights; the +1 rides in the sign MFMA
}

// the 16 codes of one K step (two words) -> the 16 window values (four dwords) of window `w`'s MFMA operand
template <bool DUAL, int NCLS>
__device__ __forceinline__ void fc4_expand(unsigned w0, unsigned w1, const i32x4& cm, const Fc4Lut (&th)[2], const Fc4Lut (&tl)[2], i32x4& hi, i32x4& lo, i32x4& sg) {
  const unsigned wd[2] = {w0, w1};
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const unsigned x = wd[j];
    const unsigned eL = x & 0x07070707u, eH = (x >> 4) & 0x07070707u;
    const unsigned sL = (x >> 3) & 0x01010101u, sH = (x >> 7) & 0x01010101u;
    // 0xff in the bytes of negative weights: v_perm_b32 selectors 0x0c / 0x0d give the constant bytes 0x00 / 0xff (s * 255 would be a
    // quarter-rate 32-bit multiply, and hipcc turns (s << 8) - s back into one)
    const unsigned mL = __builtin_amdgcn_perm(0u, 0u, sL | 0x0c0c0c0cu), mH = __builtin_amdgcn_perm(0u, 0u, sH | 0x0c0c0c0cu);
    sg[2 * j] = (int)sL; sg[2 * j + 1] = (int)sH;          // 1 in the bytes of negative weights: the sign MFMA's operand
    hi[2 * j] = (int)fc4_dec<NCLS>(eL, (unsigned)cm[2 * j], mL, th);
    hi[2 * j + 1] = (int)fc4_dec<NCLS>(eH, (unsigned)cm[2 * j + 1], mH, th);
    if constexpr (DUAL) {
      lo[2 * j] = (int)fc4_dec<NCLS>(eL, (unsigned)cm[2 * j], mL, tl);
      lo[2 * j + 1] = (int)fc4_dec<NCLS>(eH, (unsigned)cm[2 * j + 1], mH, tl);
    }
  }
}

template <bool DUAL, int NCLS>
__global__ __launch_bounds__(256) void fc4_partial_kernel(FcArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int n0 = (blockIdx.x * 4 + wave) * 32;             // this wave's 32 output channels
  if (n0 >= a.Np) return;
  const int tms = a.tm == 128 ? 7 : 6;
  const int mt = n0 >> tms, ro = n0 & ((1 << tms) - 1);
  const int wins = DUAL ? 2 : 1;
  const int s0 = blockIdx.y * a.slabs_per_split;
  const int s1 = s0 + a.slabs_per_split < a.nslab ? s0 + a.slabs_per_split : a.nslab;
  const int img = blockIdx.z * 32 + col;                   // (images beyond the batch read image 0: their columns are never looked at)
  // wave-uniform bases + 32-bit lane offsets (the scalar-base form of global_load: no 64-bit address arithmetic per lane and load)
  const unsigned x_lane = (unsigned)(img < a.B ? img : 0) * (unsigned)a.K + half * 16;
  const unsigned w_lane = (unsigned)(ro + col) * 32 + half * 16;
  const unsigned c_lane = half * 16;
  const int8_t* const xb = a.x;
  const int8_t* const wb = a.w + (((size_t)mt * a.nslab) << tms) * 32;
  const size_t ent = (size_t)32 << tms;                    // bytes of one (m-tile, slab) nibble tile
  const uint8_t* const cb = a.cls;
  // the row's tables: [class][window][8 bytes]
  Fc4Lut th[2], tl[2];
  {
    const i32x4 q0 = *reinterpret_cast<const i32x4*>(a.lut + (size_t)(n0 + col) * 32);
    const i32x4 q1 = *reinterpret_cast<const i32x4*>(a.lut + (size_t)(n0 + col) * 32 + 16);
    th[0] = {(unsigned)q0[0], (unsigned)q0[1]}; tl[0] = {(unsigned)q0[2], (unsigned)q0[3]};
    th[1] = {(unsigned)q1[0], (unsigned)q1[1]}; tl[1] = {(unsigned)q1[2], (unsigned)q1[3]};
  }
  i32x16 hi, lo, sgn;                                      // sgn: sum over the negative weights' K positions of x (added to both windows at the end)
#pragma unroll
  for (int r = 0; r < 16; r++) { hi[r] = 0; lo[r] = 0; sgn[r] = 0; }
  constexpr int GS = 4;                                    // slabs' loads in flight per wave
  auto body = [&](const i32x4& nb, const i32x4& b0, const i32x4& b1, const i32x4& c0, const i32x4& c1) __attribute__((always_inline)) {
    i32x4 h0, h1, l0, l1, g0, g1;
    fc4_expand<DUAL, NCLS>((unsigned)nb[0], (unsigned)nb[1], c0, th, tl, h0, l0, g0);
    fc4_expand<DUAL, NCLS>((unsigned)nb[2], (unsigned)nb[3], c1, th, tl, h1, l1, g1);
    hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(h0, b0, hi, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_i32_32x32x32_i8(h1, b1, hi, 0, 0, 0);
    sgn = __builtin_amdgcn_mfma_i32_32x32x32_i8(g0, b0, sgn, 0, 0, 0);
    sgn = __builtin_amdgcn_mfma_i32_32x32x32_i8(g1, b1, sgn, 0, 0, 0);
    if constexpr (DUAL) {
      lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(l0, b0, lo, 0, 0, 0);
      lo = __builtin_amdgcn_mfma_i32_32x32x32_i8(l1, b1, lo, 0, 0, 0);
    }
  };
  int s = s0;
  for (; s + GS <= s1; s += GS) {
    i32x4 nb[GS], b0[GS], b1[GS], c0[NCLS == 2 ? GS : 1], c1[NCLS == 2 ? GS : 1];
#pragma unroll
    for (int u = 0; u < GS; u++) {
      nb[u] = *reinterpret_cast<const i32x4*>(wb + (size_t)(s + u) * ent + w_lane);
      const int8_t* xp = xb + (size_t)(s + u) * 64;
      b0[u] = *reinterpret_cast<const i32x4*>(xp + x_lane); b1[u] = *reinterpret_cast<const i32x4*>(xp + 32 + x_lane);
      if constexpr (NCLS == 2) {
        c0[u] = *reinterpret_cast<const i32x4*>(cb + (size_t)(s + u) * 64 + c_lane); c1[u] = *reinterpret_cast<const i32x4*>(cb + (size_t)(s + u) * 64 + 32 + c_lane);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                     // (the loads of a group in front of its first expansion)
#pragma unroll
    for (int u = 0; u < GS; u++) body(nb[u], b0[u], b1[u], c0[NCLS == 2 ? u : 0], c1[NCLS == 2 ? u : 0]);
  }
  for (; s < s1; s++) {
    const i32x4 nb = *reinterpret_cast<const i32x4*>(wb + (size_t)s * ent + w_lane);
    const int8_t* xp = xb + (size_t)s * 64;
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(xp + x_lane), b1 = *reinterpret_cast<const i32x4*>(xp + 32 + x_lane);
    i32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    if constexpr (NCLS == 2) { c0 = *reinterpret_cast<const i32x4*>(cb + (size_t)s * 64 + c_lane); c1 = *reinterpret_cast<const i32x4*>(cb + (size_t)s * 64 + 32 + c_lane); }
    body(nb, b0, b1, c0, c1);
  }
  int* const p0 = a.part + (((size_t)blockIdx.z * a.ksplit + blockIdx.y) * wins * a.Np + n0) * 32 + col;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int row = 8 * (k >> 2) + 4 * half + (k & 3);
    p0[(size_t)row * 32] = (int)((unsigned)hi[k] + (unsigned)sgn[k]);
    if (DUAL) p0[((size_t)a.Np + row) * 32] = (int)((unsigned)lo[k] + (unsigned)sgn[k]);
  }
}

__global__ __launch_bounds__(256) void fc_finish_kernel(FcArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int n = idx >> 5, b32 = idx & 31, b = blockIdx.y * 32 + b32;        // (blockIdx.y: the image chunk)
  if (n >= a.Np || b >= a.B || n >= a.y_nvalid) return;
  const int wins = a.dual ? 2 : 1;
  unsigned hi = 0, lo = 0;
  for (int ks = 0; ks < a.ksplit; ks++) {
    const int* p = a.part + (((size_t)blockIdx.y * a.ksplit + ks) * wins * a.Np + n) * 32 + b32;
    hi += (unsigned)p[0];
    if (a.dual) lo += (unsigned)p[(size_t)a.Np * 32];
  }
  const int tms = a.tm == 128 ? 7 : 6, TM = 1 << tms;
  const int mt = n >> tms, ro = n & (TM - 1);
  const int* prm = reinterpret_cast<const int*>(reinterpret_cast<const int8_t*>(a.hdr) + (size_t)mt * a.hdr_bytes);
  int acc = (int)hi;
  if (a.dual) acc = (int)((hi << (prm[6 * TM + ro] & 31)) + lo);        // (hi << dshift[1][row]) + lo, Z/2^32
  // requant_epilogue.h, one output: rows {bias | dbl, alpha, addend64} | lo[TM]
  const int pr0 = prm[4 * ro], alpha = prm[4 * ro + 1];
  const long long b64 = (long long)(((unsigned long long)(unsigned)prm[4 * ro + 3] << 32) | (unsigned)prm[4 * ro + 2]);
  const int low = prm[4 * TM + ro];
  int y, kd;
  if (a.fast == 1) {
    const long long p = (long long)acc * (long long)alpha + b64;
    y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
    kd = pr0;
  } else {
    const int v = (int)((unsigned)pr0 + ((unsigned)acc << (low & 31)));
    const long long p = (long long)v * (long long)alpha + b64;
    if (a.fast == 2) y = (int)(p >> 32) >> (kAlphaInflat + kInflat - 32);
    else { const int x = (int)(p >> kAlphaInflat); y = __builtin_elementwise_add_sat(x, 1 << (kInflat - 1)) >> kInflat; }
    kd = low >> 8;
  }
  const int lo_b = a.relu ? 0 : -128;
  int c = y < lo_b ? lo_b : (y > 127 ? 127 : y);
  if (a.dbl) c = (int)(((unsigned)c << ((unsigned)kd >> 31)) + (unsigned)kd);
  a.y[(size_t)b * a.y_cp + a.y_off + n] = (int8_t)c;
}

// K split: enough (32-channel wave, K slice) pairs for ~8 waves per CU, at least 8 slabs per slice
int conv_fc_pick_ksplit(int Np, int nslab) {
  const int waves_m = Np / 32;
  int ks = (256 * 8 + waves_m - 1) / waves_m;
  if (ks > nslab / 8) ks = nslab / 8;
  if (ks < 1) ks = 1;
  if (ks > 32) ks = 32;
  return ks;
}

size_t conv_fc_scratch_bytes(int Np, int nslab, int dual, int batch) {
  return (size_t)((batch + 31) / 32) * conv_fc_pick_ksplit(Np, nslab) * (dual ? 2 : 1) * Np * 32 * 4;
}

int launch_conv_fc(const FcArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.B < 1 || a.chunks != (a.B + 31) / 32 || a.Np % 128 != 0 || (a.tm != 64 && a.tm != 128) || a.ksplit < 1 || a.K != a.nslab * 64) return 1;
  if (a.fc4) {
    if (a.n_cls != 1 && a.n_cls != 2) return 1;
    TF2_LAUNCH_NAME("fc4_partial_kernel<4-bit codes,%d slabs in %d slices%s,%d class%s>", a.nslab, a.ksplit, a.dual ? ",dual" : "", a.n_cls, a.n_cls == 2 ? "es" : "");
    const dim3 g(a.Np / 128, a.ksplit, a.chunks);
    if (a.dual) { if (a.n_cls == 2) TF2_LAUNCH((fc4_partial_kernel<true, 2>), g, dim3(256), 0, s, a); else TF2_LAUNCH((fc4_partial_kernel<true, 1>), g, dim3(256), 0, s, a); }
    else { if (a.n_cls == 2) TF2_LAUNCH((fc4_partial_kernel<false, 2>), g, dim3(256), 0, s, a); else TF2_LAUNCH((fc4_partial_kernel<false, 1>), g, dim3(256), 0, s, a); }
  } else {
    if (a.B > 32) return 1;                                // (the int8 form: one chunk; larger batches run on the split-K kernel)
    TF2_LAUNCH_NAME("fc_partial_kernel<%d slabs in %d slices%s>", a.nslab, a.ksplit, a.dual ? ",dual" : "");
    if (a.dual) TF2_LAUNCH((fc_partial_kernel<true>), dim3(a.Np / 128, a.ksplit), dim3(256), 0, s, a);
    else TF2_LAUNCH((fc_partial_kernel<false>), dim3(a.Np / 128, a.ksplit), dim3(256), 0, s, a);
  }
  if (!launch_ok()) return -1;
  TF2_LAUNCH_NAME("fc_finish_kernel");
  TF2_LAUNCH(fc_finish_kernel, dim3((a.Np * 32 + 255) / 256, a.chunks), dim3(256), 0, s, a);
  return launch_ok() ? 0 : -1;
}

}  // namespace tf2
