// misc_kernels.hip -- the HBM-bound glue kernels of the cnn path (gfx950).
//   prep_input : LoadInputImage/feature_trans + input quantisation
//                (host/src/input_loader.cpp:27-118, host/src/runner.cpp:158-164) fused,
//                writing the NHWC int8 tensor [x | xneg] the conv kernels consume.
//   maxpool    : pool.cl:152-260 + pool_tail.cl:91-216 (zero-extended 3x3 / 2x2 max).
//   global_avg : full_size_pool.cl:95-125.
// All are one pass over their tensors with 16-byte (pool) or coalesced accesses.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

// runner.cpp:158-163: tmp = x * trans ; (int)(tmp > 0 ? tmp + 0.5 : tmp - 0.5) ; clamp, with tmp +- 0.5 evaluated in
// double and truncated toward zero.  Restated without double precision (the DP conversions run at a fraction of the
// VALU rate and this kernel is VALU bound): (double)tmp +- 0.5 is exact, so the result is sign * (floor|tmp| +
// (frac|tmp| >= 0.5)); floor and the fraction are exact in float.  |tmp| >= 2^31 or NaN: the reference's x86 cvttsd2si
// returns INT_MIN, which clamps to -128.
__device__ __forceinline__ int quant_input(float x, float trans) {
  const float tmp = x * trans;
  const float m = __builtin_fabsf(tmp);
  if (!(m < 2147483648.0f)) return -128;
  const float f = __builtin_floorf(m);
  float r = f + ((m - f) >= 0.5f ? 1.0f : 0.0f);
  r = tmp > 0 ? r : -r;
  r = r > 127.0f ? 127.0f : (r < -128.0f ? -128.0f : r);
  return (int)r;
}

// side job of the step's first kernel: advance the workspace's step counter (conv_bgroup.hip: the value a set group flag carries)
// and clear every flag word of the step's group launches.  Done EVERY step: the control area sits behind the tensors, and a caller
// that re-uses the buffer for another batch size (or an allocator that hands the same address out again) leaves arbitrary bytes
// there -- the counter may then start anywhere (its low 24 bits just must not be 0, the value of a cleared flag), the flags may not.
__device__ __forceinline__ void prep_zero_ctrl(const PrepArgs& a) {
  if (a.epoch_ptr && blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      unsigned e = a.epoch_ptr[0] + 1u;
      if ((e & 0xffffffu) == 0) e++;
      a.epoch_ptr[0] = e;
      // error word: sticky across steps until tf2_net_poll_error reads it; anything that is no report (a fresh or re-used buffer) is cleared
      if (!bg_err_valid(a.epoch_ptr[1])) a.epoch_ptr[1] = 0u;
      a.epoch_ptr[2] = (unsigned)a.bg_poll_limit;
      a.epoch_ptr[3] = (unsigned)a.bg_withhold;
    }
    unsigned* const flags = a.epoch_ptr + 64;
    for (int i = threadIdx.x; i < a.n_flag_words; i += blockDim.x) flags[i] = 0u;
  }
}

__global__ __launch_bounds__(256) void prep_input_kernel(PrepArgs a) {
  prep_zero_ctrl(a);
  // one thread per (image, output pixel, 16-channel group of the x half): 16 gathered source
  // values -> one 16-byte store of x and one of xneg.  Channel slots >= Cl are the zero padding.
  const int Cl = a.rewrite ? a.C * 9 : a.C;
  const int G16 = a.half / 16;
  const long long total = (long long)a.B * a.OH * a.OW * G16;
  const float trans = a.q0 > 0 ? (1.0f / (float)(1 << a.q0)) : (float)(1 << (-a.q0));
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % G16);
    const long long pix = idx / G16;
    const int ow = (int)(pix % a.OW);
    const long long t = pix / a.OW;
    const int oh = (int)(t % a.OH);
    const int b = (int)(t / a.OH);
    int v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int c = cg * 16 + i;
      int ci, sr, sc;
      if (a.rewrite == 1) {
        // feature_trans (input_loader.cpp:27-73): sub-channel k of image channel ci holds
        // pad3[2*oh + roff][2*ow + coff] with (roff,coff) = k0(0,0) k1(1,0) k2(0,1) k3(1,1)
        // k4(0,2) k5(1,2) k6(2,0) k7(2,1) k8(2,2).
        ci = c / 9;
        const int k = c - ci * 9;
        const int roff = k < 6 ? (k & 1) : 2;
        const int coff = k < 6 ? (k >> 1) : (k - 6);
        sr = 2 * oh + roff - 3;
        sc = 2 * ow + coff - 3;
      } else if (a.rewrite == 2) {
        // im2col of a 3x3 first layer: channel c * 9 + fh * 3 + fw of output pixel (oh, ow)
        ci = c / 9;
        const int k = c - ci * 9;
        sr = oh * a.im_stride - a.im_pad_h + k / 3;
        sc = ow * a.im_stride - a.im_pad_w + k % 3;
      } else {
        ci = c; sr = oh; sc = ow;
      }
      int q = 0;
      if (c < Cl && (unsigned)sr < (unsigned)a.H && (unsigned)sc < (unsigned)a.W) {
        const size_t si = ((size_t)(b * a.C + ci) * a.H + sr) * a.W + sc;
        if (a.src_is_q) q = (int)reinterpret_cast<const int8_t*>(a.img)[si];
        else q = quant_input(reinterpret_cast<const float*>(a.img)[si], trans);
      }
      v[i] = q;
    }
    i32x4 px, nx;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      int p = 0, n = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = v[4 * w + j];
        p |= (q & 0xff) << (8 * j);
        n |= ((-q) & 0xff) << (8 * j);        // (int8)(-x): -128 stays -128 (pe.cl:32-37)
      }
      px[w] = p; nx[w] = n;
    }
    int8_t* dst = a.y + (size_t)pix * a.y_cp + cg * 16;
    *reinterpret_cast<i32x4*>(dst) = px;
    *reinterpret_cast<i32x4*>(dst + a.half) = nx;
  }
}

// The 7x7 / stride 2 rewrite of a 3-channel image (ResNet-50 / GoogLeNet conv1), one thread per output pixel, 256 pixels
// per block.  Sub-channel k of image channel ci is pad3[2*oh + roff][2*ow + coff] with (roff, coff) = k0(0,0) k1(1,0)
// k2(0,1) k3(1,1) k4(0,2) k5(1,2) k6(2,0) k7(2,1) k8(2,2) (feature_trans, input_loader.cpp:27-73): a 3x3 window at
// stride 2 per channel, 27 loads that neighbouring lanes share in L1, 32-bit index arithmetic.  The 64 bytes of
// [x | xneg] per pixel go through LDS so that every wave store is 1 KiB of contiguous NHWC bytes.
// XONLY: 32 bytes of x per pixel and no xneg half -- the input of conv_stem.hip, which handles x = -128 itself.
// IM2COL: the same gather for the im2col form of a 3x3 first layer (PrepArgs::rewrite == 2): window origin oh * stride - pad, taps in
// row-major order.
template <bool SRC_Q, bool XONLY, bool IM2COL = false>
__global__ __launch_bounds__(256) void prep_rewrite3_kernel(PrepArgs a) {
  prep_zero_ctrl(a);
  __shared__ __attribute__((aligned(16))) int tile[256][17];        // 64 B per pixel (+1 word: bank spread)
  const int total = a.B * a.OH * a.OW;
  const float trans = a.q0 > 0 ? (1.0f / (float)(1 << a.q0)) : (float)(1 << (-a.q0));
  const int pix0 = blockIdx.x * 256;
  const int pix = pix0 + (int)threadIdx.x;
  if (pix < total) {
    const int ow = pix % a.OW;
    const int t = pix / a.OW;
    const int oh = t % a.OH;
    const int b = t / a.OH;
    const int r0 = IM2COL ? oh * a.im_stride - a.im_pad_h : 2 * oh - 3, c0 = IM2COL ? ow * a.im_stride - a.im_pad_w : 2 * ow - 3;
    int v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = 0;
    bool rok[3], cok[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { rok[i] = (unsigned)(r0 + i) < (unsigned)a.H; cok[i] = (unsigned)(c0 + i) < (unsigned)a.W; }
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int base = ((b * 3 + ci) * a.H + r0) * a.W + c0;          // < 2^31 for any batch the workspace can hold
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int roff = IM2COL ? k / 3 : (k < 6 ? (k & 1) : 2);
        const int coff = IM2COL ? k % 3 : (k < 6 ? (k >> 1) : (k - 6));
        int q = 0;
        if (rok[roff] && cok[coff]) {
          const int si = base + roff * a.W + coff;
          if (SRC_Q) q = (int)reinterpret_cast<const int8_t*>(a.img)[si];
          else q = quant_input(reinterpret_cast<const float*>(a.img)[si], trans);
        }
        v[ci * 9 + k] = q;
      }
    }
#pragma unroll
    for (int w = 0; w < 8; w++) {
      int p = 0, n = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = v[4 * w + j];
        p |= (q & 0xff) << (8 * j);
        n |= ((-q) & 0xff) << (8 * j);        // (int8)(-x): -128 stays -128 (pe.cl:32-37)
      }
      tile[threadIdx.x][w] = p; if (!XONLY) tile[threadIdx.x][8 + w] = n;
    }
  }
  __syncthreads();
  // 256 pixels x 64 B = 16 KiB, written as 4 x (256 lanes x 16 B): consecutive lanes -> consecutive 16-byte chunks
  // (XONLY: 2 x 16 B per pixel)
  constexpr int CPP = XONLY ? 2 : 4;                   // 16-byte chunks per pixel
#pragma unroll
  for (int r = 0; r < CPP; r++) {
    const int c = r * 256 + (int)threadIdx.x;          // chunk index inside the block's output
    const int pl = c / CPP, ch = c % CPP;
    if (pix0 + pl < total) {
      i32x4 o = {tile[pl][ch * 4], tile[pl][ch * 4 + 1], tile[pl][ch * 4 + 2], tile[pl][ch * 4 + 3]};
      *reinterpret_cast<i32x4*>(a.y + (size_t)(pix0 + pl) * (CPP * 16) + ch * 16) = o;
    }
  }
}

// The x-only space-to-depth input of conv_stem.hip, every image element quantised ONCE.  prep_rewrite3_kernel gathers 27 source
// values per output pixel straight from global memory (each image element is fetched and quantised 2.3 times, scalar loads);
// here a block owns two output rows of one image: the five image rows they look at (3 channels x 224 floats each) come in with
// coalesced 16-byte loads, are quantised (runner.cpp:158-163) into an int8 LDS tile with the zero border of the pad-3 image
// (input_loader.cpp:36-72), and every thread then assembles one output pixel's 27 bytes from LDS and stores its 32 bytes.
// Same bytes as prep_rewrite3_kernel<*, true> (tests/test_gpu_parity.py checks the network input tensor).
template <bool SRC_Q>
__global__ __launch_bounds__(256) void prep_rewrite3_rows_kernel(PrepArgs a) {
  prep_zero_ctrl(a);
  constexpr int kPadL = 4, kMaxW = 256;                    // image width <= 248 (launcher-checked); 4 border columns: aligned word stores
  __shared__ __attribute__((aligned(16))) int8_t img[3][5][kMaxW + 8];
  const int rows_per_img = (a.OH + 1) / 2;
  const int b = blockIdx.x / rows_per_img;
  const int oh0 = (blockIdx.x - b * rows_per_img) * 2;
  const int r_first = 2 * oh0 - 3;                         // image row of tile row 0
  const int tid = threadIdx.x;
  const float trans = a.q0 > 0 ? (1.0f / (float)(1 << a.q0)) : (float)(1 << (-a.q0));
  // zero the borders (image columns -4..-1 and W..W+3 of every tile row)
  for (int i = tid; i < 15 * 8; i += 256) {
    const int rr = i >> 3, k = i & 7;
    img[rr / 5][rr % 5][k < kPadL ? k : kPadL + a.W + (k - kPadL)] = 0;
  }
  // 15 (channel, row) lines of W elements, 4 elements per thread and pass
  const int w4 = a.W >> 2;                                 // W % 4 == 0 (launcher-checked)
  int hit128 = 0;
  for (int i = tid; i < 15 * w4; i += 256) {
    const int line = i / w4, x4 = i - line * w4;
    const int ci = line / 5, rr = line - ci * 5;
    const int sr = r_first + rr;
    int q[4] = {0, 0, 0, 0};
    if ((unsigned)sr < (unsigned)a.H) {
      const size_t si = ((size_t)(b * 3 + ci) * a.H + sr) * a.W + x4 * 4;
      if (SRC_Q) {
        const int v = *reinterpret_cast<const int*>(reinterpret_cast<const int8_t*>(a.img) + si);
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = (int)(signed char)((v >> (8 * j)) & 0xff);
      } else {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.img) + si);
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = quant_input(v[j], trans);
      }
    }
    hit128 |= (q[0] == -128) | (q[1] == -128) | (q[2] == -128) | (q[3] == -128);
    *reinterpret_cast<unsigned*>(&img[ci][rr][kPadL + x4 * 4]) =
        (unsigned)(q[0] & 0xff) | ((unsigned)(q[1] & 0xff) << 8) | ((unsigned)(q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24);
  }
  // an element of this image quantised to -128: conv_stem_pool_kernel takes the (int8)(-x) != -x path for the image's blocks without
  // scanning its input tile (every image row is seen by the blocks that own its output rows, so no -128 escapes)
  if (a.q128 && __builtin_amdgcn_ballot_w64(hit128 != 0) != 0 && (tid & 63) == 0) atomicOr(a.q128 + b, 1u);
  __syncthreads();
  // one thread per output pixel: sub-channel k of image channel ci = pad3[2 oh + roff][2 ow + coff] (feature_trans)
  const int rsel = tid / a.OW;                             // 0 or 1: which of the block's two output rows
  const int ow = tid - rsel * a.OW;
  const int oh = oh0 + rsel;
  if (rsel < 2 && oh < a.OH) {
    unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ci = 0; ci < 3; ci++)
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int roff = k < 6 ? (k & 1) : 2;
        const int coff = k < 6 ? (k >> 1) : (k - 6);
        const unsigned v = (unsigned)(unsigned char)img[ci][2 * rsel + roff][2 * ow + coff + 1];  // tile col = image col + 4 = (2 ow + coff - 3) + 4
        const int c = ci * 9 + k;
        w[c >> 2] |= v << (8 * (c & 3));
      }
    int8_t* dst = a.y + ((size_t)(b * a.OH + oh) * a.OW + ow) * 32;
    *reinterpret_cast<i32x4*>(dst) = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    *reinterpret_cast<i32x4*>(dst + 16) = i32x4{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
  }
}

// The im2col input of a 3x3 first layer (PrepArgs::rewrite == 2), every image element fetched and quantised ONCE (round 5).
// prep_rewrite3_kernel<im2col> gathers an output pixel's 27 source values straight from global memory: each image element is loaded
// and quantised nine times (VGG16 at batch 32: 57 us, 7 % of the step; SSD300 96 us).  Here a block owns R consecutive output rows of
// one image: the (R - 1) * stride + 3 image rows they look at (3 channels) come in once -- 16-byte loads where the image width allows
// -- are quantised (runner.cpp:158-163) into an int8 LDS tile whose border is the zero padding (sequencer.cl:287), and every thread
// then assembles output pixels' 27 bytes c * 9 + fh * 3 + fw and their int8 negations (pe.cl:32-37) from LDS; the 64-byte pixels
// leave through an LDS transpose as contiguous 16-byte pieces.  Same bytes as prep_rewrite3_kernel<*, false, true>
// (tf2_net_read_layer(-1) and every first-layer test compare them).
constexpr int kImPadL = 4;                                  // tile column of image column 0 (pad_w <= 4: launcher-checked)

// The (3 channels x TR rows x WS bytes) int8 image tile of the row-tile kernels below: image rows r_first .. r_first + TR - 1 of image b,
// column 0 at tile column kImPadL, quantised (runner.cpp:158-163) on the way in; rows outside the image and the border columns keep the
// zeros the caller wrote (the padding of sequencer.cl:287).  16-byte source loads where the image width allows.
template <bool SRC_Q, int NT>
__device__ __forceinline__ void fill_image_tile(const PrepArgs& a, int8_t* img, int TR, int WS, int b, int r_first, int tid, float trans) {
  if ((a.W & 3) == 0) {
    const int w4 = a.W >> 2;
    for (int i = tid; i < 3 * TR * w4; i += NT) {
      const int line = i / w4, x4 = i - line * w4;
      const int ci = line / TR, rr = line - ci * TR;
      const int sr = r_first + rr;
      if ((unsigned)sr >= (unsigned)a.H) continue;
      const size_t si = ((size_t)(b * 3 + ci) * a.H + sr) * a.W + x4 * 4;
      int q[4];
      if (SRC_Q) {
        const int v = *reinterpret_cast<const int*>(reinterpret_cast<const int8_t*>(a.img) + si);
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = (int)(signed char)((v >> (8 * j)) & 0xff);
      } else {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.img) + si);
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = quant_input(v[j], trans);
      }
      *reinterpret_cast<unsigned*>(&img[(ci * TR + rr) * WS + kImPadL + x4 * 4]) =
          (unsigned)(q[0] & 0xff) | ((unsigned)(q[1] & 0xff) << 8) | ((unsigned)(q[2] & 0xff) << 16) | ((unsigned)(q[3] & 0xff) << 24);
    }
  } else {
    // rows that are not 16-byte aligned (227-wide images): a wave takes a tile line at a time, five lines' loads (dword / byte per lane,
    // a line = <= 4 coalesced loads per 256 columns) in flight before the first is quantised -- the per-element walk of the first version
    // (one dependent load per thread and trip, two divisions by run-time widths each) was 20 of that kernel's 34 us
    static_assert(NT % 64 == 0, "whole waves");
    constexpr int NW = NT / 64, LPI = 5;          // (SqueezeNet 1.1: 33 tile lines on 8 waves = one trip)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int x0 = 0; x0 < a.W; x0 += 256)
      for (int l0 = wave; l0 < 3 * TR; l0 += LPI * NW) {
        float vf[LPI][4]; int vq[LPI][4]; int dst[LPI];
#pragma unroll
        for (int jl = 0; jl < LPI; jl++) {
          const int line = l0 + jl * NW;
          const int ci = (line >= TR) + (line >= 2 * TR), rr = line - ci * TR;
          const int sr = r_first + rr;
          dst[jl] = (line < 3 * TR && (unsigned)sr < (unsigned)a.H) ? (ci * TR + rr) * WS + kImPadL : -1;
          const size_t si = ((size_t)(b * 3 + ci) * a.H + (dst[jl] >= 0 ? sr : 0)) * a.W;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int x = x0 + k * 64 + lane;
            vf[jl][k] = 0.f; vq[jl][k] = 0;
            if (dst[jl] >= 0 && x < a.W) {
              if (SRC_Q) vq[jl][k] = (int)reinterpret_cast<const int8_t*>(a.img)[si + x];
              else vf[jl][k] = reinterpret_cast<const float*>(a.img)[si + x];
            }
          }
        }
#pragma unroll
        for (int jl = 0; jl < LPI; jl++)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int x = x0 + k * 64 + lane;
            if (dst[jl] >= 0 && x < a.W) img[dst[jl] + x] = (int8_t)(SRC_Q ? vq[jl][k] : quant_input(vf[jl][k], trans));
          }
      }
  }
}
template <bool SRC_Q>
__global__ __launch_bounds__(256) void prep_im2col_rows_kernel(PrepArgs a, int R, int WS) {
  prep_zero_ctrl(a);
  extern __shared__ __attribute__((aligned(16))) int8_t im_lds[];
  const int s_ = a.im_stride, TR = (R - 1) * s_ + 3;
  int8_t* const img = im_lds;                               // [3][TR][WS]
  int (*const tile)[17] = reinterpret_cast<int (*)[17]>(im_lds + ((3 * TR * WS + 15) & ~15));      // [256][17]: 64 B per pixel (+1 word: bank spread)
  const int rows_per_img = (a.OH + R - 1) / R;
  const int b = blockIdx.x / rows_per_img;
  const int oh0 = (blockIdx.x - b * rows_per_img) * R;
  const int r_first = oh0 * s_ - a.im_pad_h;               // image row of tile row 0
  const int tid = threadIdx.x;
  const float trans = a.q0 > 0 ? (1.0f / (float)(1 << a.q0)) : (float)(1 << (-a.q0));
  for (int i = tid; i < 3 * TR * WS / 4; i += 256) reinterpret_cast<int*>(img)[i] = 0;      // borders, rows outside the image
  __syncthreads();
  fill_image_tile<SRC_Q, 256>(a, img, TR, WS, b, r_first, tid, trans);
  __syncthreads();
  const int rows = (a.OH - oh0) < R ? (a.OH - oh0) : R;
  const int n_px = rows * a.OW;
  const size_t px_base = ((size_t)b * a.OH + oh0) * a.OW;   // the block's output pixels are contiguous in the NHWC tensor
  for (int p0 = 0; p0 < n_px; p0 += 256) {
    const int p = p0 + tid;
    if (p < n_px) {
      const int rsel = p / a.OW, ow = p - rsel * a.OW;
      const int8_t* base = img + (rsel * s_) * WS + kImPadL + ow * s_ - a.im_pad_w;
      unsigned wx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int ci = 0; ci < 3; ci++)
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const int q = (int)base[(ci * TR + k / 3) * WS + k % 3];
          const int c = ci * 9 + k;
          wx[c >> 2] |= (unsigned)(q & 0xff) << (8 * (c & 3));
          wn[c >> 2] |= (unsigned)((-q) & 0xff) << (8 * (c & 3));        // (int8)(-x): -128 stays -128 (pe.cl:32-37)
        }
#pragma unroll
      for (int w = 0; w < 8; w++) { tile[tid][w] = (int)wx[w]; tile[tid][8 + w] = (int)wn[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = r * 256 + tid;                          // 16-byte chunk inside this pass's 256 pixels
      const int pl = c >> 2, ch = c & 3;
      if (p0 + pl < n_px) {
        const i32x4 o = {tile[pl][ch * 4], tile[pl][ch * 4 + 1], tile[pl][ch * 4 + 2], tile[pl][ch * 4 + 3]};
        *reinterpret_cast<i32x4*>(a.y + (px_base + p0 + pl) * 64 + ch * 16) = o;
      }
    }
    __syncthreads();
  }
}

// ---- conv_first_kernel: a 3x3 / stride 1 first layer on the 3-channel image in ONE launch (round 5) ---------------------------------
// prep_im2col_rows_kernel writes the im2col image (64 bytes per output pixel: VGG16 103 MB at batch 32) and conv_pw reads it back: 69 of
// VGG16's 735 us per step, 120 of SSD300's 1645.  Here the im2col tile of a block's output rows never leaves the CU: the image rows are
// quantised once into LDS (as in prep_im2col_rows_kernel), every thread assembles its pixels' 64 bytes [x(27) | 0 | xneg(27) | 0] into an
// LDS tile of 80-byte pixels (16-byte aligned, conflict-free for the 16 lanes a ds_read_b128 services together), and the eight waves
// -- two 32-channel row groups x four pixel streams, the layer's one weight slab in registers like conv_pw -- run
// v_mfma_i32_32x32x32_i8 over it, requantise (requant_epilogue.h) and store the layer's NHWC output.  Same sums, term for term
// (pe.cl:27-43, the -128 quirk through the xneg half); with `keep` (per-layer parity runs) the im2col tensor is written as well, so that
// tf2_net_read_layer(-1) still hands back the quantised image.
template <bool SRC_Q, bool DUAL>
__global__ __launch_bounds__(512) void conv_first_kernel(FirstArgs f) {
  const PrepArgs& a = f.p;
  prep_zero_ctrl(a);
  extern __shared__ __attribute__((aligned(16))) int8_t im_lds[];
  const int R = f.R, WS = f.WS, TR = R + 2;
  int8_t* const img = im_lds;                               // [3][TR][WS]
  int8_t* const col = im_lds + ((3 * TR * WS + 15) & ~15);  // [R * OW pixels][80]
  const int n_px_max = R * a.OW;
  int* const prm = reinterpret_cast<int*>(col + (((size_t)n_px_max * 80 + 15) & ~(size_t)15));
  const int rows_per_img = (a.OH + R - 1) / R;
  const int b = blockIdx.x / rows_per_img;
  const int oh0 = (blockIdx.x - b * rows_per_img) * R;
  const int r_first = oh0 - a.im_pad_h;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const float trans = a.q0 > 0 ? (1.0f / (float)(1 << a.q0)) : (float)(1 << (-a.q0));
  for (int i = tid; i < 3 * TR * WS / 4; i += 512) reinterpret_cast<int*>(img)[i] = 0;
  for (int i = tid; i < (f.hdr_used >> 4); i += 512) reinterpret_cast<i32x4*>(prm)[i] = reinterpret_cast<const i32x4*>(f.hdr)[i];
  // this wave's weights: the layer's one K slab, [window][64 rows][64 bytes]
  const int wr = wave & 1, wp = wave >> 1;
  i32x4 wf[DUAL ? 2 : 1][2];
  {
    const int row = wr * 32 + (lane & 31);
#pragma unroll
    for (int h = 0; h < (DUAL ? 2 : 1); h++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) wf[h][ks] = *reinterpret_cast<const i32x4*>(f.w + ((size_t)h * 64 + row) * 64 + (ks * 2 + half) * 16);
  }
  __syncthreads();
  fill_image_tile<SRC_Q, 512>(a, img, TR, WS, b, r_first, tid, trans);
  __syncthreads();
  const int rows = (a.OH - oh0) < R ? (a.OH - oh0) : R;
  const int n_px = rows * a.OW;
  const size_t px_base = ((size_t)b * a.OH + oh0) * a.OW;
  for (int p = tid; p < n_px; p += 512) {
    const int rsel = p / a.OW, ow = p - rsel * a.OW;
    const int8_t* base = img + rsel * WS + kImPadL + ow - a.im_pad_w;
    unsigned wx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ci = 0; ci < 3; ci++)
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int q = (int)base[(ci * TR + k / 3) * WS + k % 3];
        const int c = ci * 9 + k;
        wx[c >> 2] |= (unsigned)(q & 0xff) << (8 * (c & 3));
        wn[c >> 2] |= (unsigned)((-q) & 0xff) << (8 * (c & 3));          // (int8)(-x): -128 stays -128 (pe.cl:32-37)
      }
    i32x4* d = reinterpret_cast<i32x4*>(col + (size_t)p * 80);
    d[0] = i32x4{(int)wx[0], (int)wx[1], (int)wx[2], (int)wx[3]}; d[1] = i32x4{(int)wx[4], (int)wx[5], (int)wx[6], (int)wx[7]};
    d[2] = i32x4{(int)wn[0], (int)wn[1], (int)wn[2], (int)wn[3]}; d[3] = i32x4{(int)wn[4], (int)wn[5], (int)wn[6], (int)wn[7]};
    if (f.keep) {
      i32x4* g = reinterpret_cast<i32x4*>(f.im + (px_base + p) * 64);
      g[0] = d[0]; g[1] = d[1]; g[2] = d[2]; g[3] = d[3];
    }
  }
  __syncthreads();
  const int lo_bound = f.relu ? 0 : -128;
  const int chl = wr * 32 + 16 * half;
  const rq_i32x4 nores = {0, 0, 0, 0};
  for (int t = wp; t * 32 < n_px; t += 4) {
    const int p_raw = t * 32 + (lane & 31);
    const int p = p_raw < n_px ? p_raw : n_px - 1;
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(col + (size_t)p * 80 + half * 16);
    const i32x4 b1 = *reinterpret_cast<const i32x4*>(col + (size_t)p * 80 + half * 16 + 32);
    i32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0; acc2[r] = 0; }
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][0], b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][1], b1, acc, 0, 0, 0);
    int a16[16];
    if constexpr (DUAL) {
      acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[1][0], b0, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[1][1], b1, acc2, 0, 0, 0);
      const int* dsh = prm + kPrmWordsPerRow * 64 + 64 + wr * 32 + 4 * half;       // dshift[1] behind rows | lo | dshift[0]
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++) a16[G * 4 + r] = (int)(((unsigned)acc[G * 4 + r] << (d[r] & 31)) + (unsigned)acc2[G * 4 + r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
    }
    i32x4 out;
    if (f.fast == 1) out = requant_tile16<false, 0, true>(a16, prm, 64, wr * 32 + 4 * half, lo_bound, -128, nores, f.dbl != 0, false);
    else out = requant_tile16<false, 0, false>(a16, prm, 64, wr * 32 + 4 * half, lo_bound, -128, nores, f.dbl != 0, f.fast == 2);
    if (p_raw < n_px && chl + 16 <= f.y_nvalid)
      *reinterpret_cast<i32x4*>(f.y + (px_base + p_raw) * f.y_cp + f.y_off + chl) = out;
  }
}

// the fused first layer fits?  (stride 1, 16-byte image rows, one weight slab of <= 64 rows, up to 320 pixels per block)
bool conv_first_fits(const PrepArgs& a, int* R_out, int* WS_out, size_t* lds_out, int hdr_used) {
  if (a.rewrite != 2 || a.C != 3 || a.half != 32 || a.y_cp != 64 || a.im_stride != 1 || (a.W & 3) || a.im_pad_w > kImPadL || a.im_pad_h > 4) return false;
  if (a.OW < 8 || a.OW > 320) return false;
  const int R = a.OW > 160 ? 1 : (320 / a.OW < a.OH ? 320 / a.OW : a.OH);
  const int span = a.OW - 1 - a.im_pad_w + 3;
  const int WS = (kImPadL + (a.W > span ? a.W : span) + 4 + 3) & ~3;
  const size_t lds = (size_t)((3 * (R + 2) * WS + 15) & ~15) + (((size_t)R * a.OW * 80 + 15) & ~(size_t)15) + (size_t)hdr_used;
  if (lds > 64 * 1024) return false;
  *R_out = R; *WS_out = WS; *lds_out = lds;
  return true;
}

int launch_conv_first(const FirstArgs& f0, void* stream) {
  FirstArgs f = f0;
  int R, WS; size_t lds;
  if (!conv_first_fits(f.p, &R, &WS, &lds, f.hdr_used)) return 1;
  f.R = R; f.WS = WS;
  const PrepArgs& a = f.p;
  if ((long long)a.B * a.OH * a.OW * 64 >= (1ll << 31) || (long long)a.B * 3 * a.H * a.W >= (1ll << 31)) return 1;
  const unsigned grid = (unsigned)(a.B * ((a.OH + R - 1) / R));
  TF2_LAUNCH_NAME("conv_first_kernel<im2col tile in LDS,%d rows per block%s>", R, f.dual ? ",dual" : "");
  if (a.src_is_q) { if (f.dual) TF2_LAUNCH((conv_first_kernel<true, true>), dim3(grid), dim3(512), lds, (hipStream_t)stream, f); else TF2_LAUNCH((conv_first_kernel<true, false>), dim3(grid), dim3(512), lds, (hipStream_t)stream, f); }
  else { if (f.dual) TF2_LAUNCH((conv_first_kernel<false, true>), dim3(grid), dim3(512), lds, (hipStream_t)stream, f); else TF2_LAUNCH((conv_first_kernel<false, false>), dim3(grid), dim3(512), lds, (hipStream_t)stream, f); }
  return launch_ok() ? 0 : -1;
}

// ---- conv_first_pool_kernel: a 3x3 first layer (stride 1 or 2) on the 3-channel image AND its 3x3 / stride 2 max pool in ONE launch ---
// SqueezeNet 1.1's front was three launches -- im2col input preparation 16 us, conv1 as a pointwise layer 12 us, pool1 9.5 us: 19 % of the
// step -- that write and re-read the im2col image (13 MB) and the 113 x 113 x 64 conv map (26 MB per batch of 32).  Here a block owns PR
// pooled rows x the full width of one image: the (PR - 1) * 2 + 3 conv rows under them are computed as in conv_first_kernel (image rows
// quantised once into LDS, the im2col tile in LDS, one weight slab in registers, v_mfma_i32_32x32x32_i8, requant_epilogue.h) into an
// LDS tile of the layer's NHWC bytes, and the pool (pool.cl:152-260 + pool_tail.cl:91-216: zero beyond the valid area) runs from there.
// Neighbouring blocks recompute one conv row each ((2 PR + 1) / 2 PR of the layer).
//   * im2col tile: 48 bytes per pixel -- x(27) | 0 in the first 32 -- conflict-free for the sixteen lanes of a ds_read_b128 (12 i mod 64
//     words are 16 disjoint groups of four); the xneg K half (pe.cl:32-37, (int8)(-x), -128 stays -128) is derived in registers from
//     the fragment: per byte ~x + 1 without a carry into the next byte;
//   * conv tile: 64 bytes per pixel, 16-byte chunk g of pixel p at chunk g ^ ((p >> 2) & 3) (the 16 lanes of a ds_write_b128 -- consecutive
//     pixels, one chunk -- land in 16 different bank groups);
//   * the layer has a ReLU (launcher-checked): every byte is 0..127 and the pool is v_pk_max_u16 on the even / odd bytes, window slots
//     outside the map are skipped (they are zeros, pool.cl:119-140, and no byte is below zero).
// With `keep` (per-layer parity runs) the im2col tensor and the conv map are written as well (valid rows; neighbours write equal bytes).
__device__ __forceinline__ unsigned first_neg_bytes(unsigned x) {
  const unsigned t = ~x;
  return ((t & 0x7f7f7f7fu) + 0x01010101u) ^ (t & 0x80808080u);
}

template <bool SRC_Q, bool DUAL>
__global__ __launch_bounds__(512, 4) void conv_first_pool_kernel(FirstArgs f) {
  const PrepArgs& a = f.p;
  prep_zero_ctrl(a);
  extern __shared__ __attribute__((aligned(16))) int8_t im_lds[];
  const int s_ = a.im_stride, RC = f.R, WS = f.WS, TR = (RC - 1) * s_ + 3;
  const int OW = a.OW, OH = a.OH;
  const int n_px = RC * OW;                                  // conv pixels of the block's tile (rows outside the map included: skipped by the pool)
  int8_t* const img = im_lds;                               // [3][TR][WS]
  int8_t* const col = im_lds + ((3 * TR * WS + 15) & ~15);  // [n_px][48]
  int8_t* const cy = col + (size_t)n_px * 48;               // [n_px][64], chunk-swizzled
  int* const prm = reinterpret_cast<int*>(cy + (size_t)n_px * 64);
  const int bands = (f.PH + f.PR - 1) / f.PR;
  const int b = blockIdx.x / bands;
  const int ph0 = (blockIdx.x - b * bands) * f.PR;
  const int oh0 = ph0 * 2 - f.ppad;                          // conv row of tile row 0
  const int r_first = oh0 * s_ - a.im_pad_h;                 // image row of image-tile row 0
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const float trans = a.q0 > 0 ? (1.0f / (float)(1 << a.q0)) : (float)(1 << (-a.q0));
  for (int i = tid; i < 3 * TR * WS / 4; i += 512) reinterpret_cast<int*>(img)[i] = 0;
  for (int i = tid; i < (f.hdr_used >> 4); i += 512) reinterpret_cast<i32x4*>(prm)[i] = reinterpret_cast<const i32x4*>(f.hdr)[i];
  const int wr = wave & 1, wp = wave >> 1;
  i32x4 wf[DUAL ? 2 : 1][2];
  {
    const int row = wr * 32 + (lane & 31);
#pragma unroll
    for (int h = 0; h < (DUAL ? 2 : 1); h++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) wf[h][ks] = *reinterpret_cast<const i32x4*>(f.w + ((size_t)h * 64 + row) * 64 + (ks * 2 + half) * 16);
  }
  __syncthreads();
  fill_image_tile<SRC_Q, 512>(a, img, TR, WS, b, r_first, tid, trans);
  __syncthreads();
  for (int p = tid; p < n_px; p += 512) {
    const int rsel = p / OW, ow = p - rsel * OW;
    const int8_t* base = img + (rsel * s_) * WS + kImPadL + ow * s_ - a.im_pad_w;
    unsigned wx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ci = 0; ci < 3; ci++)
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int q = (int)base[(ci * TR + k / 3) * WS + k % 3];
        const int c = ci * 9 + k;
        wx[c >> 2] |= (unsigned)(q & 0xff) << (8 * (c & 3));
      }
    i32x4* d = reinterpret_cast<i32x4*>(col + (size_t)p * 48);
    d[0] = i32x4{(int)wx[0], (int)wx[1], (int)wx[2], (int)wx[3]}; d[1] = i32x4{(int)wx[4], (int)wx[5], (int)wx[6], (int)wx[7]};
    if (f.keep && (unsigned)(oh0 + rsel) < (unsigned)OH) {
      i32x4* g = reinterpret_cast<i32x4*>(f.im + (((size_t)b * OH + oh0 + rsel) * OW + ow) * 64);
      g[0] = d[0]; g[1] = d[1];
      g[2] = i32x4{(int)first_neg_bytes(wx[0]), (int)first_neg_bytes(wx[1]), (int)first_neg_bytes(wx[2]), (int)first_neg_bytes(wx[3])};
      g[3] = i32x4{(int)first_neg_bytes(wx[4]), (int)first_neg_bytes(wx[5]), (int)first_neg_bytes(wx[6]), (int)first_neg_bytes(wx[7])};
    }
  }
  __syncthreads();
  const int chl = wr * 32 + 16 * half;
  const int g_out = wr * 2 + half;                           // this lane's 16-byte chunk of a conv pixel
  const rq_i32x4 nores = {0, 0, 0, 0};
  for (int t = wp; t * 32 < n_px; t += 4) {
    const int p_raw = t * 32 + (lane & 31);
    const int p = p_raw < n_px ? p_raw : n_px - 1;
    const i32x4 b0 = *reinterpret_cast<const i32x4*>(col + (size_t)p * 48 + half * 16);
    const i32x4 b1 = {(int)first_neg_bytes((unsigned)b0[0]), (int)first_neg_bytes((unsigned)b0[1]), (int)first_neg_bytes((unsigned)b0[2]), (int)first_neg_bytes((unsigned)b0[3])};
    i32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0; acc2[r] = 0; }
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][0], b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][1], b1, acc, 0, 0, 0);
    int a16[16];
    if constexpr (DUAL) {
      acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[1][0], b0, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[1][1], b1, acc2, 0, 0, 0);
      const int* dsh = prm + kPrmWordsPerRow * 64 + 64 + wr * 32 + 4 * half;       // dshift[1] behind rows | lo | dshift[0]
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++) a16[G * 4 + r] = (int)(((unsigned)acc[G * 4 + r] << (d[r] & 31)) + (unsigned)acc2[G * 4 + r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
    }
    i32x4 out;
    if (f.fast == 1) out = requant_tile16<false, 0, true>(a16, prm, 64, wr * 32 + 4 * half, 0, -128, nores, false, false);
    else out = requant_tile16<false, 0, false>(a16, prm, 64, wr * 32 + 4 * half, 0, -128, nores, false, f.fast == 2);
    if (p_raw < n_px) {
      *reinterpret_cast<i32x4*>(cy + (size_t)p_raw * 64 + ((g_out ^ ((p_raw >> 2) & 3)) << 4)) = out;
      if (f.keep && chl + 16 <= f.y_nvalid) {
        const int rsel = p_raw / OW, ow = p_raw - rsel * OW;
        if ((unsigned)(oh0 + rsel) < (unsigned)OH)
          *reinterpret_cast<i32x4*>(f.y + (((size_t)b * OH + oh0 + rsel) * OW + ow) * f.y_cp + f.y_off + chl) = out;
      }
    }
  }
  __syncthreads();
  const int pr_n = (f.PH - ph0) < f.PR ? (f.PH - ph0) : f.PR;
  const int n_out = pr_n * f.PW * 4;
  for (int idx = tid; idx < n_out; idx += 512) {
    const int g = idx & 3, pix = idx >> 2;
    const int pr = pix / f.PW, pw = pix - pr * f.PW;
    unsigned me[4] = {0, 0, 0, 0}, mo[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int r = pr * 2 + i;
      if ((unsigned)(oh0 + r) >= (unsigned)OH) continue;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int ow = pw * 2 - f.ppad + j;
        if ((unsigned)ow >= (unsigned)OW) continue;
        const int p = r * OW + ow;
        const i32x4 v = *reinterpret_cast<const i32x4*>(cy + (size_t)p * 64 + ((g ^ ((p >> 2) & 3)) << 4));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          me[q] = pk_max_u16(me[q], (unsigned)v[q] & 0x00ff00ffu);
          mo[q] = pk_max_u16(mo[q], ((unsigned)v[q] >> 8) & 0x00ff00ffu);
        }
      }
    }
    if (g * 16 + 16 <= f.y_nvalid) {
      const i32x4 o = {(int)(me[0] | (mo[0] << 8)), (int)(me[1] | (mo[1] << 8)), (int)(me[2] | (mo[2] << 8)), (int)(me[3] | (mo[3] << 8))};
      *reinterpret_cast<i32x4*>(f.yp + (((size_t)b * f.PH + ph0 + pr) * f.PW + pw) * f.yp_cp + f.yp_off + g * 16) = o;
    }
  }
}

// the pooled form fits?  (3x3 layer of stride 1 / 2 with a ReLU, 3x3 / stride 2 pool, one weight slab of <= 64 rows; two blocks per CU)
bool conv_first_pool_fits(const PrepArgs& a, int pool_S, int pool_st, int pool_pad, int PH, int PW, int relu, int hdr_used,
                          int* PR_out, int* WS_out, size_t* lds_out) {
  if (a.rewrite != 2 || a.C != 3 || a.half != 32 || a.y_cp != 64 || (a.im_stride != 1 && a.im_stride != 2) || a.im_pad_w > kImPadL || a.im_pad_h > 4) return false;
  if (pool_S != 3 || pool_st != 2 || pool_pad < 0 || pool_pad > 1 || !relu || a.OW < 8 || PH < 1 || PW < 1) return false;
  if ((PH - 1) * 2 - pool_pad >= a.OH || (PW - 1) * 2 - pool_pad >= a.OW) return false;      // (every window holds a valid element)
  const int span = (a.OW - 1) * a.im_stride - a.im_pad_w + 3;
  const int WS = (kImPadL + (a.W > span ? a.W : span) + 4 + 3) & ~3;
  for (int PR = 4; PR >= 1; PR--) {
    const int RC = (PR - 1) * 2 + 3, TR = (RC - 1) * a.im_stride + 3;
    const size_t lds = (size_t)((3 * TR * WS + 15) & ~15) + (size_t)RC * a.OW * (48 + 64) + (size_t)hdr_used;
    if (lds > 78 * 1024 || PR * PW * 4 > 4096) continue;
    *PR_out = PR < PH ? PR : PH; *WS_out = WS; *lds_out = lds;
    return true;
  }
  return false;
}

int launch_conv_first_pool(const FirstArgs& f0, void* stream) {
  FirstArgs f = f0;
  int PR, WS; size_t lds;
  if (!conv_first_pool_fits(f.p, 3, 2, f.ppad, f.PH, f.PW, f.relu, f.hdr_used, &PR, &WS, &lds)) return 1;
  f.PR = PR; f.R = (PR - 1) * 2 + 3; f.WS = WS;
  const PrepArgs& a = f.p;
  if ((long long)a.B * a.OH * a.OW * 64 >= (1ll << 31) || (long long)a.B * 3 * a.H * a.W >= (1ll << 31)) return 1;
  const unsigned grid = (unsigned)(a.B * ((f.PH + PR - 1) / PR));
  TF2_LAUNCH_NAME("conv_first_pool_kernel<stride %d,%d pooled rows per block%s>", a.im_stride, PR, f.dual ? ",dual" : "");
  // up to 78 KiB of dynamic LDS: above the 64 KiB a kernel may use without asking (every other kernel of the library beyond that asks too)
#define TF2_FP(Q, D) do { auto fn = conv_first_pool_kernel<Q, D>; if (lds > 64 * 1024 && !lds_attr_once(reinterpret_cast<const void*>(fn))) return -1; \
                          TF2_LAUNCH(fn, dim3(grid), dim3(512), lds, (hipStream_t)stream, f); } while (0)
  if (a.src_is_q) { if (f.dual) TF2_FP(true, true); else TF2_FP(true, false); }
  else { if (f.dual) TF2_FP(false, true); else TF2_FP(false, false); }
#undef TF2_FP
  return launch_ok() ? 0 : -1;
}

__global__ __launch_bounds__(256) void maxpool_kernel(PoolArgs a) {
  // one thread per (output pixel, 16-channel group)
  const long long total = (long long)a.B * a.PH * a.PW * a.C16;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int cg = (int)(idx % a.C16);
    long long pix = idx / a.C16;
    int pw = (int)(pix % a.PW);
    long long t = pix / a.PW;
    int ph = (int)(t % a.PH);
    int b = (int)(t / a.PH);
    int m[16];
    const int init = a.S < 3 ? 0 : -128;      // window slots >= S stay 0 (pool.cl:177-186)
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = init;
    for (int i = 0; i < a.S; i++)
      for (int j = 0; j < a.S; j++) {
        int h = ph * a.st - a.pad + i, w = pw * a.st - a.pad + j;
        i32x4 v = {0, 0, 0, 0};               // zero beyond the valid area (pool.cl:119-140)
        if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
          v = *reinterpret_cast<const i32x4*>(a.x + ((size_t)(b * a.H + h) * a.W + w) * a.x_cp + a.x_off + cg * 16);
#pragma unroll
        for (int q = 0; q < 16; q++) {
          int x = (int)(signed char)((v[q >> 2] >> (8 * (q & 3))) & 0xff);
          m[q] = x > m[q] ? x : m[q];
        }
      }
    i32x4 o;
#pragma unroll
    for (int q = 0; q < 4; q++)
      o[q] = (m[4 * q] & 0xff) | ((m[4 * q + 1] & 0xff) << 8) | ((m[4 * q + 2] & 0xff) << 16) | ((m[4 * q + 3] & 0xff) << 24);
    *reinterpret_cast<i32x4*>(a.y + (size_t)pix * a.y_cp + a.y_off + cg * 16) = o;
  }
}

__global__ __launch_bounds__(256) void global_avg_kernel(AvgArgs a) {
  // one block per (image, 8 groups of 16 channels = 128 contiguous bytes per pixel); thread = (pixel part, group):
  // 32 parts stride over the HW pixels with 16-byte loads, partial sums meet in LDS.  The reference accumulates in
  // int16 with wrap-around (Sreal, types.h:30): addition mod 2^16 is associative, the wrap is applied at the end.
  __shared__ int part_sum[32][8][17];
  const int G16 = a.C / 16;
  const int gblocks = (G16 + 7) / 8;
  const int b = blockIdx.x / gblocks, g0 = (blockIdx.x % gblocks) * 8;
  const int tid = threadIdx.x, cgl = tid & 7, part = tid >> 3;
  const int cg = g0 + cgl;
  int s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = 0;
  if (cg < G16) {
    const int8_t* p = a.x + (size_t)b * a.HW * a.x_cp + a.x_off + cg * 16;
    for (int i = part; i < a.HW; i += 32) {
      const i32x4 v = *reinterpret_cast<const i32x4*>(p + (size_t)i * a.x_cp);
#pragma unroll
      for (int q = 0; q < 16; q++) s[q] += (int)(signed char)((v[q >> 2] >> (8 * (q & 3))) & 0xff);
    }
  }
#pragma unroll
  for (int q = 0; q < 16; q++) part_sum[part][cgl][q] = s[q];
  __syncthreads();
  if (tid < 128) {
    const int gl = tid >> 4, ch = tid & 15;
    int t = 0;
#pragma unroll 8
    for (int pp = 0; pp < 32; pp++) t += part_sum[pp][gl][ch];
    const int sv = (int)(short)t;                          // int16 accumulator wrap
    int m = (((sv * a.mult) >> 14) + 1) >> 1;              // full_size_pool.cl:118
    m = m > 127 ? 127 : (m < -128 ? -128 : m);
    if (g0 + gl < G16) a.y[(size_t)b * a.y_cp + a.y_off + (g0 + gl) * 16 + ch] = (int8_t)m;
  }
}

// L2Norm row (l2norm.py:19-24 in the engine's integer form; the arithmetic and its order are those of tf2o_l2norm in
// oracle/tf2_oracle.c): one wavefront per pixel, a lane owns channels 8*lane + 512*k .. +7, the exact integer sum of squares
// meets in the wave through __shfl_xor, the rest is IEEE double (-ffp-contract=off: no fused multiply-add).
__global__ __launch_bounds__(256) void l2norm_kernel(L2NormArgs a) {
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= a.n_pix) return;                               // wave-uniform
  const int8_t* xp = a.x + (size_t)pix * a.x_cp;
  long long S = 0;
  for (int c0 = lane * 8; c0 < a.C; c0 += 512) {
    const unsigned long long v8 = *reinterpret_cast<const unsigned long long*>(xp + c0);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const long long v = (long long)(signed char)((v8 >> (8 * i)) & 0xff) << a.e[c0 + i];
      S += v * v;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) S += __shfl_xor(S, m, 64);
  const double norm = __builtin_sqrt((double)S) * __builtin_ldexp(1.0, -a.qs) + 1e-10;
  int8_t* yp = a.y + (size_t)pix * a.y_cp;
  for (int c0 = lane * 8; c0 < a.C; c0 += 512) {
    const unsigned long long v8 = *reinterpret_cast<const unsigned long long*>(xp + c0);
    unsigned long long o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const double x = (double)(int)(signed char)((v8 >> (8 * i)) & 0xff);
      const double t = (x * a.a[c0 + i]) / norm;
      const double v = t * a.b[c0 + i];
      double r = v > 0 ? __builtin_floor(v + 0.5) : __builtin_ceil(v - 0.5);
      r = r > 127.0 ? 127.0 : (r < -128.0 ? -128.0 : r);
      o |= (unsigned long long)((int)r & 0xff) << (8 * i);
    }
    *reinterpret_cast<unsigned long long*>(yp + c0) = o;
  }
}

int launch_l2norm(const L2NormArgs& a, void* stream) {
  TF2_LAUNCH_NAME("l2norm_kernel"); TF2_LAUNCH(l2norm_kernel, dim3((a.n_pix + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return launch_ok() ? 0 : -1;
}

static inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  long long cap = 256LL * 16;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// launch_prep_input's choice of prep_rewrite3_rows_kernel (net.hip hands PrepArgs::q128 / StemArgs::q128 out only then)
bool prep_takes_rows_kernel(const PrepArgs& a) {
  const long long pixels = (long long)a.B * a.OH * a.OW;
  return a.rewrite == 1 && a.C == 3 && a.half == 32 && a.y_cp == 64 && pixels * 64 < (1ll << 31) && (long long)a.B * 3 * a.H * a.W < (1ll << 31) &&
         a.xonly && a.W % 4 == 0 && a.W <= 248 && 2 * a.OW <= 256 && a.OH == a.H / 2 + 2 && a.OW == a.W / 2 + 2;
}

int launch_prep_input(const PrepArgs& a, void* stream) {
  const long long pixels = (long long)a.B * a.OH * a.OW;
  if (a.rewrite == 2 && a.C == 3 && a.half == 32 && a.y_cp == 64 && pixels * 64 < (1ll << 31) && (long long)a.B * 3 * a.H * a.W < (1ll << 31)) {
    // every image element quantised once through an LDS row tile (prep_im2col_rows_kernel) where the geometry allows it
    // (stride 1 and an image width of whole 16-byte float quadruples: every element is then used nine times and comes in by wide loads --
    //  VGG16 57 -> 29 us; SqueezeNet's stride-2 layer on a 227-wide image measured SLOWER this way, 21.9 against 16.5 us, and keeps the gather)
    if (a.im_stride == 1 && (a.W & 3) == 0 && a.im_pad_w <= kImPadL && a.im_pad_h <= 4 && a.OW >= 8) {
      const int R = a.OW >= 256 ? 1 : (256 / a.OW < a.OH ? 256 / a.OW : a.OH);        // output rows per block: up to 256 pixels
      const int TR = (R - 1) * a.im_stride + 3;
      const int span = (a.OW - 1) * a.im_stride - a.im_pad_w + 3;                        // image columns [-pad_w, span - pad_w) are looked at
      const int WS = (kImPadL + (a.W > span ? a.W : span) + 4 + 3) & ~3;
      const size_t lds = (size_t)((3 * TR * WS + 15) & ~15) + 256 * 17 * 4;
      if (lds <= 64 * 1024) {
        const unsigned gridr = (unsigned)(a.B * ((a.OH + R - 1) / R));
        TF2_LAUNCH_NAME("prep_im2col_rows_kernel<%d rows per block>", R);
        if (a.src_is_q) TF2_LAUNCH((prep_im2col_rows_kernel<true>), dim3(gridr), dim3(256), lds, (hipStream_t)stream, a, R, WS);
        else TF2_LAUNCH((prep_im2col_rows_kernel<false>), dim3(gridr), dim3(256), lds, (hipStream_t)stream, a, R, WS);
        return launch_ok() ? 0 : -1;
      }
    }
    const unsigned grid = (unsigned)((pixels + 255) / 256);
    TF2_LAUNCH_NAME("prep_rewrite3_kernel<im2col>");
    if (a.src_is_q) TF2_LAUNCH((prep_rewrite3_kernel<true, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else TF2_LAUNCH((prep_rewrite3_kernel<false, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return launch_ok() ? 0 : -1;
  }
  if (a.rewrite == 1 && a.C == 3 && a.half == 32 && a.y_cp == 64 && pixels * 64 < (1ll << 31) && (long long)a.B * 3 * a.H * a.W < (1ll << 31)) {
    const unsigned grid = (unsigned)((pixels + 255) / 256);
    if (prep_takes_rows_kernel(a)) {
      const unsigned gridr = (unsigned)(a.B * ((a.OH + 1) / 2));
      TF2_LAUNCH_NAME("prep_rewrite3_rows_kernel");
      if (a.src_is_q) TF2_LAUNCH((prep_rewrite3_rows_kernel<true>), dim3(gridr), dim3(256), 0, (hipStream_t)stream, a);
      else TF2_LAUNCH((prep_rewrite3_rows_kernel<false>), dim3(gridr), dim3(256), 0, (hipStream_t)stream, a);
      return launch_ok() ? 0 : -1;
    }
    if (a.xonly) {
      if (a.src_is_q) { TF2_LAUNCH_NAME("prep_rewrite3_kernel"); TF2_LAUNCH((prep_rewrite3_kernel<true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
      else { TF2_LAUNCH_NAME("prep_rewrite3_kernel"); TF2_LAUNCH((prep_rewrite3_kernel<false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
    } else {
      if (a.src_is_q) { TF2_LAUNCH_NAME("prep_rewrite3_kernel"); TF2_LAUNCH((prep_rewrite3_kernel<true, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
      else { TF2_LAUNCH_NAME("prep_rewrite3_kernel"); TF2_LAUNCH((prep_rewrite3_kernel<false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); }
    }
    return launch_ok() ? 0 : -1;
  }
  if (a.xonly) return -1;                       // only the space-to-depth form has an x-only variant (net.hip checks the same limits)
  long long total = (long long)a.B * a.OH * a.OW * (a.half / 16);
  TF2_LAUNCH_NAME("prep_input_kernel"); TF2_LAUNCH(prep_input_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_ok() ? 0 : -1;
}

int launch_maxpool(const PoolArgs& a, void* stream) {
  long long total = (long long)a.B * a.PH * a.PW * a.C16;
  TF2_LAUNCH_NAME("maxpool_kernel"); TF2_LAUNCH(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_ok() ? 0 : -1;
}

int launch_global_avg(const AvgArgs& a, void* stream) {
  TF2_LAUNCH_NAME("global_avg_kernel"); TF2_LAUNCH(global_avg_kernel, dim3(a.B * ((a.C / 16 + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_ok() ? 0 : -1;
}

const char* device_last_error() { return hipGetErrorString(hipGetLastError()); }

}  // namespace tf2
