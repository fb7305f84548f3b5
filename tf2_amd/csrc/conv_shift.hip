// conv_shift.hip -- the literal shift-accumulate convolution on the vector ALUs (gfx950).
//
// This is the direct restatement of the reference's PE datapath for ANY weight code:
//   acc[n,p] = bias[n] + sum_{c,fh,fw} MUL(x, code)     MUL = +-(x << s), s in 0..31
// (device/src/pe.cl:27-49,144-180), including the reference's quirk that a negative
// weight negates the activation in int8, so x = -128 stays -128 (pe.cl:32-37).
//
// Mapping: lane = output pixel (64 consecutive pixels of the batch per block), the four
// waves of a block take 8 output channels each, so a weight is WAVE-UNIFORM: it is
// fetched once through the scalar cache as the integer +-2^s and the whole
// shift-accumulate for 64 pixels is one v_mad (v_mad_i32_i24 when every shift of the
// layer is <= 22, a 32-bit multiply-add otherwise: x * 2^s == x << s in Z/2^32).
// For inputs that may be negative, positive weights multiply x and the magnitudes of
// negative weights multiply xneg = (int8)(-x) (computed in registers), reproducing the
// -128 quirk bit for bit; for post-ReLU inputs one signed weight per tap suffices.
//
// Input staging: per 16-channel chunk, all taps of the 64 pixels are gathered with
// coalesced 16-byte loads into LDS (zero padding applied there, sequencer.cl:287); each
// lane then reads its own 16 bytes per tap (ds_read_b128, lane-linear = conflict-free)
// and unpacks them once for the 8 output channels of its wave.
// Epilogue: the arithmetic of requant_epilogue.h's generic path (bias, BN requant, ReLU, residual), 8-byte stores.
//
// PACKED4 (PackLayer::fast on a shift layer, weight_pack.cpp): the filters stay packed in HBM as 4-bit codes {sign, e} with
// s = A[n] + B[c] - e -- the INQ form of 4bit_data_format.txt, 8x fewer weight bytes than the int32 form -- and every wave
// expands the codes of its 8 output channels for the current 16-channel chunk to +-2^s in LDS (two codes per lane and tap),
// next to the staged input tile; the multiply-accumulate then takes its weight from a broadcast ds_read instead of a
// scalar load.
#include <hip/hip_runtime.h>
#include "tf2_internal.h"
#include "tf2_device.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int requant_i8s(int acc, int alpha, int beta, int relu) {
  long long p = (long long)acc * (long long)alpha;
  int t = (int)(p >> kAlphaInflat);
  t = (int)((unsigned)t + (unsigned)beta);
  int v = ((t >> (kInflat - 1)) + 1) >> 1;
  v = v > 127 ? 127 : (v < -128 ? -128 : v);
  if (relu) v = v > 0 ? v : 0;
  return v;
}

constexpr int kMaxTaps = 49;   // up to 7x7 filters

template <bool SIGNED_IN, bool MUL24, bool PACKED4>
__global__ __launch_bounds__(256) void conv_shift_kernel(ConvArgs a) {
  // LDS: [tap][64 pixels][16 bytes] | PACKED4: [wave][tap][half][8 n][8 c] int32 (+ the same again for the negative magnitudes)
  extern __shared__ __attribute__((aligned(16))) int8_t lds_raw[];
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int taps = a.k * a.k;
  const int px0 = blockIdx.x * 64;
  const int n0 = (blockIdx.y * 4 + wave) * 8;          // first output channel of this wave
  const bool wave_active = n0 < a.Np;

  // geometry of this lane's pixel (for the epilogue) and of the pixels this thread stages
  int acc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) acc[r] = 0;

  // staging assignment: item = tap * 64 + pixel, items strided by 256 threads
  const int n_items = taps * 64;

  for (int cc = 0; cc < a.n_cchunk; cc++) {
    __syncthreads();
    // ---- stage x: all taps of this 16-channel chunk for the block's 64 pixels ----
    for (int it = tid; it < n_items; it += 256) {
      const int tap = it >> 6, pl = it & 63;
      const int p = px0 + pl;
      i32x4 v = {0, 0, 0, 0};
      if (p < g.n_pix) {
        int b = fast_div(p, g.ohw_m, g.ohw_s);
        int rem = p - b * g.OHW;
        int oh = fast_div(rem, g.ow_m, g.ow_s), ow = rem - (fast_div(rem, g.ow_m, g.ow_s)) * g.OW;
        int fh = tap / a.k, fw = tap - fh * a.k;
        int ih = oh * g.stride - g.pad_h + fh * a.dil;
        int iw = ow * g.stride - g.pad_w + fw * a.dil;
        if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) {
          const int8_t* src = a.x + ((size_t)(b * g.H * g.W + ih * g.W + iw) * g.Cp_in + cc * 16);
          v = *reinterpret_cast<const i32x4*>(src);
        }
      }
      *reinterpret_cast<i32x4*>(lds_raw + (size_t)it * 16) = v;
    }
    int* const wl = reinterpret_cast<int*>(lds_raw + (size_t)taps * 64 * 16) + (size_t)wave * taps * 128 * (SIGNED_IN ? 2 : 1);
    if (PACKED4 && wave_active) {
      // expand this wave's codes of chunk cc: weight idx = tap * 128 + half * 64 + r * 8 + c (the int32 layout's order).
      // One dword = 8 codes = the 8 channels c of one (tap, half, r): lane l takes dwords l, l + 64, ... -> its (half, r) never
      // change, so its eight shifts S[c] = A[n0 + r] + B[chunk, half, c] are computed once per chunk.
      const unsigned* nb = reinterpret_cast<const unsigned*>(reinterpret_cast<const uint8_t*>(a.w) + (((size_t)(n0 >> 3) * a.n_cchunk + cc) * taps * 128) / 2);
      const int8_t* ab = a.w2;                                   // A[Np] | B[n_cchunk * 16]
      const int r = lane & 7, h = (lane >> 3) & 1;
      const int An = (int)ab[n0 + r];
      const int2 bw = *reinterpret_cast<const int2*>(ab + a.Np + cc * 16 + h * 8);
      int S[8];
#pragma unroll
      for (int c = 0; c < 8; c++) S[c] = An + (int)(signed char)(((c < 4 ? bw.x : bw.y) >> (8 * (c & 3))) & 0xff);
      for (int j = lane; j < taps * 16; j += 64) {
        const unsigned word = nb[j];
        int wv[8], wn[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const unsigned v = (word >> (4 * c)) & 15u;
          const int e = (int)(v & 7u);
          const int mag = e == 7 ? 0 : (int)(1u << ((S[c] - e) & 31));
          if (SIGNED_IN) { wv[c] = (v & 8u) ? 0 : mag; wn[c] = (v & 8u) ? mag : 0; }
          else wv[c] = (v & 8u) ? (int)(0u - (unsigned)mag) : mag;
        }
        i32x4* d = reinterpret_cast<i32x4*>(wl + j * 8);
        d[0] = i32x4{wv[0], wv[1], wv[2], wv[3]}; d[1] = i32x4{wv[4], wv[5], wv[6], wv[7]};
        if (SIGNED_IN) {
          i32x4* d2 = reinterpret_cast<i32x4*>(wl + taps * 128 + j * 8);
          d2[0] = i32x4{wn[0], wn[1], wn[2], wn[3]}; d2[1] = i32x4{wn[4], wn[5], wn[6], wn[7]};
        }
      }
    }
    __syncthreads();
    if (!wave_active) continue;
    // ---- shift-accumulate ----
    // weights: [n8 tile][cchunk][tap][half][8 n][8 c] int32
    const int* wbase = reinterpret_cast<const int*>(a.w) +
                       ((size_t)(n0 >> 3) * a.n_cchunk + cc) * taps * 128;
    const int* w2base = SIGNED_IN ? reinterpret_cast<const int*>(a.w2) +
                                        ((size_t)(n0 >> 3) * a.n_cchunk + cc) * taps * 128
                                  : nullptr;
    for (int tap = 0; tap < taps; tap++) {
      const i32x4 xv = *reinterpret_cast<const i32x4*>(lds_raw + (size_t)(tap * 64 + lane) * 16);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int xs[8], xn[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
          int word = xv[h * 2 + (c >> 2)];
          xs[c] = (int)(signed char)((word >> (8 * (c & 3))) & 0xff);
          if (SIGNED_IN) xn[c] = (int)(signed char)(-xs[c]);   // int8 negate: -(-128) == -128 (pe.cl:32-37)
        }
        const int* wp = PACKED4 ? wl + (tap * 2 + h) * 64 : wbase + (tap * 2 + h) * 64;      // wave-uniform: scalar loads (LDS broadcast reads with PACKED4)
        const int* wq = SIGNED_IN ? (PACKED4 ? wl + taps * 128 + (tap * 2 + h) * 64 : w2base + (tap * 2 + h) * 64) : nullptr;
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const int w = wp[r * 8 + c];
            if (MUL24) acc[r] += __mul24(xs[c], w);
            else acc[r] = (int)((unsigned)acc[r] + (unsigned)xs[c] * (unsigned)w);
            if (SIGNED_IN) {
              const int w2 = wq[r * 8 + c];
              if (MUL24) acc[r] += __mul24(xn[c], w2);
              else acc[r] = (int)((unsigned)acc[r] + (unsigned)xn[c] * (unsigned)w2);
            }
          }
        }
      }
    }
  }

  if (!wave_active) return;
  const int px = px0 + lane;
  if (px >= g.n_pix) return;
  int q[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    int v = (int)((unsigned)a.bias[n0 + r] + (unsigned)acc[r]);
    q[r] = requant_i8s(v, a.alpha[n0 + r], a.beta[n0 + r], g.relu);
  }
  if (n0 + 8 > g.y_nvalid) return;                       // padding rows beyond the tensor
  if (g.has_res) {
    const int2 rv = *reinterpret_cast<const int2*>(a.res + (size_t)px * g.res_cp + g.res_off + n0);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      int word = r < 4 ? rv.x : rv.y;
      int rr = (int)(signed char)((word >> (8 * (r & 3))) & 0xff);
      int s = q[r] + rr;
      s = s > 127 ? 127 : (s < -128 ? -128 : s);
      if (g.add_relu) s = s > 0 ? s : 0;
      q[r] = s;
    }
  }
  int2 out;
  out.x = (q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((q[3] & 0xff) << 24);
  out.y = (q[4] & 0xff) | ((q[5] & 0xff) << 8) | ((q[6] & 0xff) << 16) | ((q[7] & 0xff) << 24);
  *reinterpret_cast<int2*>(a.y + (size_t)px * g.y_cp + g.y_off + n0) = out;
}


// ---- few pixels, many channels (a fully connected layer behind a global average: SqueezeNet 1.1's 1000 -> 128 on 32 pixels) ----
// conv_shift_kernel walks the channel chunks one after the other behind two barriers each: with one block of pixels that is a chain
// of n_cchunk memory latencies on four blocks (157 us for 63 chunks, measured).  Here one block owns 8 output channels x 32 pixels
// and its sixteen waves take every sixteenth 16-channel chunk each; a lane is (pixel = lane & 31, channel half = lane >> 5) and
// reads its 8 input bytes straight from HBM (one tap: nothing to gather), so all 64 lanes work on 32 pixels; the two halves and the
// sixteen waves' partial sums are added at the end (lane exchange, then LDS).  The sum is in Z/2^32: any order, same bits.
template <bool SIGNED_IN, bool MUL24, bool PACKED4>
__global__ __launch_bounds__(1024) void conv_shift_fc_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) int8_t lds_raw[];      // [16 waves][8][32] int32 partial sums | PACKED4: [wave][2 signs][128] int32
  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;                                                // which 8 channels of a 16-channel chunk
  const int n0 = blockIdx.y * 8;
  const int p = blockIdx.x * 32 + (lane & 31);
  int* const red = reinterpret_cast<int*>(lds_raw);
  int* const wl = red + 16 * 8 * 32 + wave * 256;
  const int8_t* xsrc = nullptr;
  if (p < g.n_pix) {
    const int b = fast_div(p, g.ohw_m, g.ohw_s);
    const int rem = p - b * g.OHW;
    const int oh = fast_div(rem, g.ow_m, g.ow_s), ow = rem - oh * g.OW;
    xsrc = a.x + (size_t)(b * g.H * g.W + oh * g.stride * g.W + ow * g.stride) * g.Cp_in + h * 8;     // k = 1, no padding (checked by the launcher)
  }
  int acc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) acc[r] = 0;
  for (int cc = wave; cc < a.n_cchunk; cc += 16) {
    int2 xv = {0, 0};
    if (xsrc) xv = *reinterpret_cast<const int2*>(xsrc + cc * 16);
    if (PACKED4) {
      // 128 codes of this (n8 tile, chunk) = 16 dwords; lanes 0..15 expand one dword (8 channels of one (half, r)) each
      const unsigned* nb = reinterpret_cast<const unsigned*>(reinterpret_cast<const uint8_t*>(a.w) + (((size_t)(n0 >> 3) * a.n_cchunk + cc) * 128) / 2);
      const int8_t* ab = a.w2;
      if (lane < 16) {
        const int r = lane & 7, hh = (lane >> 3) & 1;
        const int An = (int)ab[n0 + r];
        const int2 bw = *reinterpret_cast<const int2*>(ab + a.Np + cc * 16 + hh * 8);
        const unsigned word = nb[lane];
        int wv[8], wn[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int S = An + (int)(signed char)(((c < 4 ? bw.x : bw.y) >> (8 * (c & 3))) & 0xff);
          const unsigned v = (word >> (4 * c)) & 15u;
          const int e = (int)(v & 7u);
          const int mag = e == 7 ? 0 : (int)(1u << ((S - e) & 31));
          if (SIGNED_IN) { wv[c] = (v & 8u) ? 0 : mag; wn[c] = (v & 8u) ? mag : 0; }
          else { wv[c] = (v & 8u) ? (int)(0u - (unsigned)mag) : mag; wn[c] = 0; }
        }
        i32x4* d = reinterpret_cast<i32x4*>(wl + lane * 8);
        d[0] = i32x4{wv[0], wv[1], wv[2], wv[3]}; d[1] = i32x4{wv[4], wv[5], wv[6], wv[7]};
        if (SIGNED_IN) {
          i32x4* d2 = reinterpret_cast<i32x4*>(wl + 128 + lane * 8);
          d2[0] = i32x4{wn[0], wn[1], wn[2], wn[3]}; d2[1] = i32x4{wn[4], wn[5], wn[6], wn[7]};
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // weights of this lane's channel half: [n8 tile][cchunk][half][8 n][8 c] int32 (two addresses per wave: L1 / LDS broadcast)
    const int* wp = (PACKED4 ? wl : reinterpret_cast<const int*>(a.w) + ((size_t)(n0 >> 3) * a.n_cchunk + cc) * 128) + h * 64;
    const int* wq = nullptr;
    if (SIGNED_IN) wq = (PACKED4 ? wl + 128 : reinterpret_cast<const int*>(a.w2) + ((size_t)(n0 >> 3) * a.n_cchunk + cc) * 128) + h * 64;
    int xs[8], xn[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int word = c < 4 ? xv.x : xv.y;
      xs[c] = (int)(signed char)((word >> (8 * (c & 3))) & 0xff);
      xn[c] = SIGNED_IN ? (int)(signed char)(-xs[c]) : 0;         // int8 negate: -(-128) == -128 (pe.cl:32-37)
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const i32x4 w0 = *reinterpret_cast<const i32x4*>(wp + r * 8), w1 = *reinterpret_cast<const i32x4*>(wp + r * 8 + 4);
      i32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
      if (SIGNED_IN) { v0 = *reinterpret_cast<const i32x4*>(wq + r * 8); v1 = *reinterpret_cast<const i32x4*>(wq + r * 8 + 4); }
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int w = c < 4 ? w0[c] : w1[c - 4];
        if (MUL24) acc[r] += __mul24(xs[c], w);
        else acc[r] = (int)((unsigned)acc[r] + (unsigned)xs[c] * (unsigned)w);
        if (SIGNED_IN) {
          const int w2 = c < 4 ? v0[c] : v1[c - 4];
          if (MUL24) acc[r] += __mul24(xn[c], w2);
          else acc[r] = (int)((unsigned)acc[r] + (unsigned)xn[c] * (unsigned)w2);
        }
      }
    }
    if (PACKED4) __builtin_amdgcn_wave_barrier();                  // the next chunk's expansion overwrites wl
  }
  // the two channel halves of a pixel sit in lanes l and l + 32
#pragma unroll
  for (int r = 0; r < 8; r++) acc[r] = (int)((unsigned)acc[r] + (unsigned)__shfl_xor(acc[r], 32));
  if (h == 0) {
#pragma unroll
    for (int r = 0; r < 8; r++) red[(wave * 8 + r) * 32 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0 || h != 0 || p >= g.n_pix || n0 + 8 > g.y_nvalid) return;
  int q[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    unsigned v = (unsigned)a.bias[n0 + r];
#pragma unroll
    for (int w = 0; w < 16; w++) v += (unsigned)red[(w * 8 + r) * 32 + lane];
    q[r] = requant_i8s((int)v, a.alpha[n0 + r], a.beta[n0 + r], g.relu);
  }
  if (g.has_res) {
    const int2 rv = *reinterpret_cast<const int2*>(a.res + (size_t)p * g.res_cp + g.res_off + n0);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int word = r < 4 ? rv.x : rv.y;
      int s = q[r] + (int)(signed char)((word >> (8 * (r & 3))) & 0xff);
      s = s > 127 ? 127 : (s < -128 ? -128 : s);
      if (g.add_relu) s = s > 0 ? s : 0;
      q[r] = s;
    }
  }
  int2 out;
  out.x = (q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((q[3] & 0xff) << 24);
  out.y = (q[4] & 0xff) | ((q[5] & 0xff) << 8) | ((q[6] & 0xff) << 16) | ((q[7] & 0xff) << 24);
  *reinterpret_cast<int2*>(a.y + (size_t)p * g.y_cp + g.y_off + n0) = out;
}

// the wave-split form pays when the pixel blocks cannot fill the chip and the chunk walk is long
static bool shift_fc_form(const ConvArgs& a) {
  return a.k == 1 && a.g.pad_h == 0 && a.g.pad_w == 0 && a.n_cchunk >= 16 && (a.g.n_pix + 63) / 64 * ((a.Np / 8 + 3) / 4) < 64;
}

size_t conv_shift_lds_bytes(int taps, int signed_in, int packed4) {
  return (size_t)taps * 64 * 16 + (packed4 ? (size_t)4 * taps * 128 * 4 * (signed_in ? 2 : 1) : 0);
}

int launch_conv_shift(const ConvArgs& a, int signed_in, int mul24, int packed4, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int taps = a.k * a.k;
  if (taps > kMaxTaps) return -2;
  if (shift_fc_form(a)) {
    dim3 fgrid((a.g.n_pix + 31) / 32, a.Np / 8);
    const size_t flds = (size_t)16 * 8 * 32 * 4 + (packed4 ? (size_t)16 * 256 * 4 : 0);
#define TF2_SHF(S, M, P) do { TF2_LAUNCH_NAME("conv_shift_fc_kernel"); TF2_LAUNCH((conv_shift_fc_kernel<S, M, P>), fgrid, dim3(1024), flds, s, a); } while (0)
    if (packed4) {
      if (signed_in) { if (mul24) TF2_SHF(true, true, true); else TF2_SHF(true, false, true); }
      else { if (mul24) TF2_SHF(false, true, true); else TF2_SHF(false, false, true); }
    } else {
      if (signed_in) { if (mul24) TF2_SHF(true, true, false); else TF2_SHF(true, false, false); }
      else { if (mul24) TF2_SHF(false, true, false); else TF2_SHF(false, false, false); }
    }
#undef TF2_SHF
    return launch_ok() ? 0 : -1;
  }
  dim3 grid((a.g.n_pix + 63) / 64, (a.Np / 8 + 3) / 4);
  const size_t lds = conv_shift_lds_bytes(taps, signed_in, packed4);
  if (lds > 160 * 1024) return -3;
#define TF2_SH(S, M, P) do { auto fn = conv_shift_kernel<S, M, P>; if (lds > 64 * 1024 && !lds_attr_once(reinterpret_cast<const void*>(fn))) return -1; \
                             TF2_LAUNCH_NAME("conv_shift_kernel"); TF2_LAUNCH(fn, grid, dim3(256), lds, s, a); } while (0)
  if (packed4) {
    if (signed_in) { if (mul24) TF2_SH(true, true, true); else TF2_SH(true, false, true); }
    else { if (mul24) TF2_SH(false, true, true); else TF2_SH(false, false, true); }
  } else {
    if (signed_in) { if (mul24) TF2_SH(true, true, false); else TF2_SH(true, false, false); }
    else { if (mul24) TF2_SH(false, true, false); else TF2_SH(false, false, false); }
  }
#undef TF2_SH
  return launch_ok() ? 0 : -1;
}

}  // namespace tf2
