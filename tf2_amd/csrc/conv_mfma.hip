// conv_mfma.hip -- implicit-GEMM INT8 convolution on the CDNA4 matrix cores (gfx950).
//
// Replaces the reference's PE array for layers whose effective weights fit int8 tiles:
//   acc[n,p] = bias[n] + sum_k MUL(x[p,k], code[n,k])            (device/src/pe.cl:27-49,144-180)
// A weight code is (zero | sign | left-shift s) (pe.cl:27-40), i.e. the integer +-2^s.
// Because the accumulator lives in Z/2^32 (pe.cl:43 "change from long int to int"),
//   sum_k x*(+-2^s) == sum_phases ( sum_k x*(+-2^(s-lo_p[n])) ) << lo_p[n]        exactly,
// so each row's shifts are covered by windows of 7 exponents (host side, weight_pack.cpp)
// and every window is an ordinary int8 GEMM on v_mfma_i32_32x32x32_i8; windows are
// combined Horner-style by left-shifting the int32 accumulators between phases.
//
// Orientation: A operand = weights (rows = output channels), B operand = activations
// (columns = output pixels, NHWC so K is contiguous), D[row=channel][col=pixel].  With
// the 32x32 C/D layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) a lane
// ends up with 4x4 consecutive channels of one pixel, which two v_permlane32_swap turn
// into one 16-byte NHWC store.
//
// Block = 256 threads = 4 waves; tile = TM channels x TN pixels, K step = one 64-byte
// slab (one filter tap x 64 channels, or several taps when the tensor has fewer
// channels); weights arrive as pre-tiled [TM][64] int8 blocks listed per (m-tile, phase)
// (all-zero blocks are not stored), activations are gathered per pixel with zero padding
// (sequencer.cl:287).  Both go global -> registers -> XOR-swizzled LDS (conflict-free
// ds_read_b128) with the next slab's loads in flight during the current slab's MFMAs.
// Epilogue (per output element): acc = bias + (sum << lo) ; BN requant (pe.cl:185-203);
// ReLU (relu.cl:54); residual add in int16 + clamp + ReLU (feature_writer.cl:119-122).
#include <hip/hip_runtime.h>
#include "tf2_internal.h"
#include "tf2_device.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int requant_i8(int acc, int alpha, int beta, int relu) {
  long long p = (long long)acc * (long long)alpha;           // pe.cl:191
  int t = (int)(p >> kAlphaInflat);                          // pe.cl:192
  t = (int)((unsigned)t + (unsigned)beta);
  int v = ((t >> (kInflat - 1)) + 1) >> 1;                   // pe.cl:193
  v = v > 127 ? 127 : (v < -128 ? -128 : v);                 // pe.cl:194
  if (relu) v = v > 0 ? v : 0;                               // relu.cl:54
  return v;
}

__device__ __forceinline__ int add_res_i8(int v, int r, int add_relu) {
  int s = v + r;                                             // feature_writer.cl:119 (int16 never overflows)
  s = s > 127 ? 127 : (s < -128 ? -128 : s);                 // :120
  if (add_relu) s = s > 0 ? s : 0;                           // :121
  return s;
}

template <int MB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  constexpr int TM = 64 * MB;                 // channels per block
  constexpr int TN = (MB == 2) ? 128 : 256;   // pixels per block
  constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64;
  constexpr int AQ = TM * 4 / 256;            // 16-byte chunks of A per thread
  constexpr int BQ = TN * 4 / 256;            // 16-byte chunks of B per thread
  __shared__ __attribute__((aligned(16))) int8_t lds[2 * (A_BYTES + B_BYTES)];

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (MB == 2) ? (wave >> 1) : 0;
  const int wn = (MB == 2) ? (wave & 1) : wave;

  // XCD-aware block remap: consecutive logical tiles (same pixel tile, all channel
  // tiles; then the neighbouring pixel tile) share one XCD's L2.
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mtile = bid % a.n_mtiles;
  const int ntile = bid / a.n_mtiles;
  const int px0 = ntile * TN;

  // ---- per-thread gather state for the B (activation) rows it stages ----------
  const int seg = tid & 3;
  int brow_h[BQ], brow_w[BQ], brow_base[BQ];
#pragma unroll
  for (int q = 0; q < BQ; q++) {
    int p = px0 + (tid >> 2) + 64 * q;
    if (p < g.n_pix) {
      int b = fast_div(p, g.ohw_m, g.ohw_s);
      int rem = p - b * g.OHW;
      int oh = fast_div(rem, g.ow_m, g.ow_s);
      int ow = rem - oh * g.OW;
      brow_h[q] = oh * g.stride - g.pad_h;
      brow_w[q] = ow * g.stride - g.pad_w;
      brow_base[q] = b * g.H * g.W;
    } else {
      brow_h[q] = -(1 << 20); brow_w[q] = 0; brow_base[q] = 0;   // never in range -> zeros
    }
  }

  const int P = a.n_phases;
  const int* dirp = a.dir + mtile * (P + 1);
  const int e_begin = dirp[0];
  const int e_end = dirp[P];

  i32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

  i32x4 areg[AQ], breg[BQ];

  auto load_global = [&](int e) {
    const int slab = a.entries[e];
    const int8_t* wsrc = a.w + (size_t)e * A_BYTES;
#pragma unroll
    for (int q = 0; q < AQ; q++) {
      int idx = tid + 256 * q;
      areg[q] = *reinterpret_cast<const i32x4*>(wsrc + idx * 16);
    }
    const unsigned ki = (unsigned)a.kinfo[slab * 4 + seg];
    const int dh = (int)((ki >> 16) & 0xff);
    const int dw = (int)(ki >> 24);
    const int coff = (ki & 0xffff) == 0xffff ? -1 : (int)(ki & 0xffff);
#pragma unroll
    for (int q = 0; q < BQ; q++) {
      int ih = brow_h[q] + dh, iw = brow_w[q] + dw;
      bool ok = coff >= 0 && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      i32x4 v = {0, 0, 0, 0};
      if (ok) {
        const int8_t* src = a.x + ((size_t)(brow_base[q] + ih * g.W + iw) * g.Cp_in + coff);
        v = *reinterpret_cast<const i32x4*>(src);
      }
      breg[q] = v;
    }
  };

  auto store_lds = [&](int buf) {
    int8_t* A = lds + buf * (A_BYTES + B_BYTES);
    int8_t* B = A + A_BYTES;
#pragma unroll
    for (int q = 0; q < AQ; q++) {
      int idx = tid + 256 * q;
      int row = idx >> 2, c = idx & 3;
      *reinterpret_cast<i32x4*>(A + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = areg[q];
    }
#pragma unroll
    for (int q = 0; q < BQ; q++) {
      int row = (tid >> 2) + 64 * q;
      *reinterpret_cast<i32x4*>(B + row * 64 + ((seg ^ ((row >> 2) & 3)) << 4)) = breg[q];
    }
  };

  auto compute = [&](int buf) {
    const int8_t* A = lds + buf * (A_BYTES + B_BYTES);
    const int8_t* B = A + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks * 2 + (lane >> 5);
      i32x4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int row = wm * 64 + i * 32 + (lane & 31);
        af[i] = *reinterpret_cast<const i32x4*>(A + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int row = wn * 64 + j * 32 + (lane & 31);
        bf[j] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  // Horner step when entering phase p >= 1: acc <<= dshift[p][channel]
  auto phase_shift = [&](int p) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int chb = mtile * TM + wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(a.dshift + (size_t)p * a.Np + chb + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j][G * 4 + r] = (int)((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31));
      }
    }
  };

  int phase = 0;
  if (e_begin < e_end) {
    load_global(e_begin);
    store_lds(0);
  }
  __syncthreads();
  for (int e = e_begin; e < e_end; e++) {
    const int buf = (e - e_begin) & 1;
    while (phase + 1 < P && e == dirp[phase + 1]) { phase++; phase_shift(phase); }
    if (e + 1 < e_end) load_global(e + 1);
    compute(buf);
    if (e + 1 < e_end) store_lds(buf ^ 1);
    __syncthreads();
  }
  while (phase + 1 < P) { phase++; phase_shift(phase); }

  // ---- epilogue -----------------------------------------------------------------
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int tile_ch = mtile * TM + wm * 64 + i * 32;     // first channel of this 32-row tile
    const int chb = tile_ch + 4 * half;
    i32x4 bias4[4], lo4[4], al4[4], be4[4];
#pragma unroll
    for (int G = 0; G < 4; G++) {
      bias4[G] = *reinterpret_cast<const i32x4*>(a.bias + chb + 8 * G);
      lo4[G] = *reinterpret_cast<const i32x4*>(a.lo + chb + 8 * G);
      al4[G] = *reinterpret_cast<const i32x4*>(a.alpha + chb + 8 * G);
      be4[G] = *reinterpret_cast<const i32x4*>(a.beta + chb + 8 * G);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int px = px0 + wn * 64 + j * 32 + (lane & 31);
      const bool pvalid = px < g.n_pix;
      unsigned d[4];
#pragma unroll
      for (int G = 0; G < 4; G++) {
        int q[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int v = (int)((unsigned)bias4[G][r] + ((unsigned)acc[i][j][G * 4 + r] << (lo4[G][r] & 31)));
          q[r] = requant_i8(v, al4[G][r], be4[G][r], g.relu);
        }
        if (g.has_res) {
          int rv = 0;
          const int chl = chb + 8 * G;                       // layer-local channel of q[0]
          if (pvalid && chl + 4 <= g.y_nvalid)
            rv = *reinterpret_cast<const int*>(a.res + (size_t)px * g.res_cp + g.res_off + chl);
#pragma unroll
          for (int r = 0; r < 4; r++) q[r] = add_res_i8(q[r], (int)(signed char)((rv >> (8 * r)) & 0xff), g.add_relu);
        }
        d[G] = (unsigned)(q[0] & 0xff) | ((unsigned)(q[1] & 0xff) << 8) | ((unsigned)(q[2] & 0xff) << 16) |
               ((unsigned)(q[3] & 0xff) << 24);
      }
      if (g.flags & 1) {
        // debug path: four 4-byte stores straight from the C/D layout
#pragma unroll
        for (int G = 0; G < 4; G++) {
          const int chl = chb + 8 * G;
          if (pvalid && chl + 4 <= g.y_nvalid)
            *reinterpret_cast<unsigned*>(a.y + (size_t)px * g.y_cp + g.y_off + chl) = d[G];
        }
      } else {
        // d[G] holds channel group 2G (lanes 0-31) / 2G+1 (lanes 32-63).  Two half-wave
        // swaps give lanes 0-31 groups 0..3 and lanes 32-63 groups 4..7: one 16-B store.
        auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
        i32x4 out = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
        const int chl = tile_ch + 16 * half;
        if (pvalid && chl + 16 <= g.y_nvalid)
          *reinterpret_cast<i32x4*>(a.y + (size_t)px * g.y_cp + g.y_off + chl) = out;
      }
    }
  }
}

int launch_conv_mfma(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (TM == 128) {
    int ntiles = (a.g.n_pix + 127) / 128;
    hipLaunchKernelGGL(conv_mfma_kernel<2>, dim3(ntiles * a.n_mtiles), dim3(256), 0, s, a);
  } else if (TM == 64) {
    int ntiles = (a.g.n_pix + 255) / 256;
    hipLaunchKernelGGL(conv_mfma_kernel<1>, dim3(ntiles * a.n_mtiles), dim3(256), 0, s, a);
  } else {
    return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace tf2
