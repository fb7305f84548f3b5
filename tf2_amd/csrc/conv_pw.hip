// conv_pw.hip -- pointwise (1x1, stride 1, unpadded) INT8 convolution with the weights held in REGISTERS (gfx950).
//
// The short-K pointwise layers (64 -> 256, 128 -> 512, 64 -> 64 ...: a third of ResNet-50's launches) are
// byte-bound: one or two 64-byte K slabs per output, 64..512 output bytes per input pixel.  In the tiled kernels
// every 128x128 tile is a workgroup with its own prologue, LDS ring, barriers and drain; measured, those layers
// ran at ~20 % of what their bytes and their requantisation arithmetic need.  Here:
//
//  * a wave owns 32 output channels for the whole launch: its A operands (both exponent windows of every K slab,
//    at most 32 registers) are loaded once, straight from the packed tiles in global memory;
//  * it then streams pixel tiles of 32: the B operand of `v_mfma_i32_32x32x32_i8` is exactly 16 contiguous NHWC
//    bytes per lane (pixel = lane & 31, K half = lane >> 5), so activations go global -> register -> MFMA with no
//    LDS staging, no barrier and no other wave involved; the loads of the next tiles (and their residual tiles) are
//    in flight while the current one is multiplied, requantised and stored (a ring of 2-4 register buffers);
//  * LDS holds only the per-m-tile parameter header (requant rows, window shift), read 16 bytes per row;
//  * blocks = 8 waves = (TM / 32 row groups) x (pixel streams); grid ~ 2 blocks per CU, each block walks its own
//    contiguous range of pixel tiles; the channel tiles of one pixel range sit on one XCD (shared L2 lines).
//
// Arithmetic, packed image and epilogue are those of conv_mfma2.hip (reference: pe.cl:27-43 shift-accumulate,
// pe.cl:191-194 requant, relu.cl:54, feature_writer.cl:88-122 residual); eligibility is checked by the launcher.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int TM, int NSLAB, bool DUAL>
__global__ __launch_bounds__(512, 4) void conv_pw_kernel(ConvArgs a, int n_t32, int tiles_per_chunk) {
  constexpr int NWIN = DUAL ? 2 : 1;
  constexpr int RG = TM / 32;                  // row groups (waves along channels)
  constexpr int NPW = 8 / RG;                  // pixel streams (waves along pixels)
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  int* const prm = reinterpret_cast<int*>(lds);

  TF2_PRELOAD_CONV_ARGS(a);          // every kernel argument in SGPRs after two scalar-load round trips (tf2_device.h)
  (void)a_max_ent; (void)P; (void)mt_m; (void)mt_s;
  TF2_PROBE_WORD(g.flags);           // timing probes (tf2_device.h; constant 0 in the product build)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave % RG, wp = wave / RG;
  const int half = lane >> 5;
  int* const dsh = prm + kPrmWordsPerRow * TM;

  const int M = a_n_mtiles;
  const int b = blockIdx.x;
  const int mtile = (b >> 3) % M;
  const int chunk = (b & 7) + 8 * ((b >> 3) / M);
  const int t_begin = chunk * tiles_per_chunk;
  int t_end = t_begin + tiles_per_chunk;
  if (t_end > n_t32) t_end = n_t32;

  // header -> LDS (shared by the block)
  {
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(ahdr) + (size_t)mtile * a_hdr_bytes + lane * 16;
    for (int i = wave; i * 1024 < a_hdr_bytes; i += 8)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hsrc + i * 1024), TF2_LDS_PTR(lds + i * 1024), 16, 0, 0);
  }
  // this wave's weights -> registers: entry s of the m-tile is slab s (launcher-checked), [window][TM rows][64 B]
  i32x4 wf[NSLAB][NWIN][2];
  {
    const int8_t* wt = aw + (size_t)(mtile * NSLAB) * (NWIN * TM * 64);      // dense layer: m-tile mt owns entries mt*NSLAB ..
    const int row = wr * 32 + (lane & 31);
#pragma unroll
    for (int s = 0; s < NSLAB; s++)
#pragma unroll
      for (int h = 0; h < NWIN; h++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
          wf[s][h][ks] = *reinterpret_cast<const i32x4*>(wt + ((size_t)(s * NWIN + h) * TM + row) * 64 + (ks * 2 + half) * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int lo_bound = g.relu ? 0 : -128;
  const int rlo = g.add_relu ? 0 : -128;
  const int chl = mtile * TM + wr * 32 + 16 * half;           // first of this lane's 16 output channels
  const bool ch_ok = chl + 16 <= g.y_nvalid;
  const int last_px = g.n_pix - 1;

  struct Tile { i32x4 bf[NSLAB][2]; i32x4 res; };
  auto load_tile = [&](int t, Tile& T) {
    int px = t * 32 + (lane & 31);
    px = px > last_px ? last_px : px;                         // clamped: out-of-range pixels are never stored
    const int8_t* xp = ax + (size_t)px * g.Cp_in + half * 16;
#pragma unroll
    for (int s = 0; s < NSLAB; s++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
        T.bf[s][ks] = *reinterpret_cast<const i32x4*>(xp + s * 64 + ks * 32);
    // unconditional (zero page without a residual): a branch around the load would make hipcc wait at the join
    const int8_t* rp = (g.has_res && ch_ok) ? ares + (size_t)px * g.res_cp + g.res_off + chl : azero;
    T.res = *reinterpret_cast<const i32x4*>(rp);
  };
  auto compute_tile = [&](int t, const Tile& T) {
    i32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0; acc2[r] = 0; }
#pragma unroll
    for (int s = 0; s < NSLAB; s++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        if (prb & kProbeNoMfma) { asm volatile("" :: "v"(T.bf[s][ks])); continue; }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s][0][ks], T.bf[s][ks], acc, 0, 0, 0);
        if (DUAL) acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s][1][ks], T.bf[s][ks], acc2, 0, 0, 0);
      }
    int a16[16];
    if (DUAL) {
      // (hi << dshift[1][row]) + lo   (Z/2^32)
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + TM + wr * 32 + 4 * half + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
          a16[G * 4 + r] = (int)(((unsigned)acc[G * 4 + r] << (d[r] & 31)) + (unsigned)acc2[G * 4 + r]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
    }
    i32x4 out;
    const int row0 = wr * 32 + 4 * half;
    if (prb & kProbeNoEpi) out = i32x4{a16[0] + a16[4], a16[1] + a16[5], a16[2] + a16[6] + a16[8] + a16[12], a16[3] + a16[7] + T.res[0]};
    else
    if (g.fast == 1) out = g.has_res ? requant_tile16<true, 2, true>(a16, prm, TM, row0, lo_bound, rlo, T.res)
                                     : requant_tile16<false, 2, true>(a16, prm, TM, row0, lo_bound, rlo, T.res, g.dbl_out != 0);
    else out = g.has_res ? requant_tile16<true, 2, false>(a16, prm, TM, row0, lo_bound, rlo, T.res, false, g.fast == 2)
                         : requant_tile16<false, 2, false>(a16, prm, TM, row0, lo_bound, rlo, T.res, g.dbl_out != 0, g.fast == 2);
    const int px = t * 32 + (lane & 31);
    if (prb & kProbeNoStore) { asm volatile("" :: "v"(out)); return; }
    if (px <= last_px && ch_ok)
      *reinterpret_cast<i32x4*>(ay + (size_t)px * g.y_cp + g.y_off + chl) = out;
  };

  // ---- stream this wave's pixel tiles: t_begin + wp, + NPW, ... with the next tile always in flight ----
  int t = t_begin + wp;
  if (t >= t_end) return;
  const int t_last = t_end - 1;                // loads are unconditional (clamped to a valid tile): no joins
  // D tiles in flight per wave
  constexpr int D = 2;                          // measured: 4 in flight is no faster (the kernel is VALU-issue bound)
  Tile T[D];
#pragma unroll
  for (int i = 0; i < D - 1; i++) {
    const int ti = t + i * NPW;
    load_tile(ti < t_end ? ti : t_last, T[i]);
  }
  bool more = true;
  while (more) {
#pragma unroll
    for (int j = 0; j < D; j++) {
      if (more) {
        const int tn = t + (D - 1) * NPW;
        if (!(prb & kProbeNoB)) load_tile(tn < t_end ? tn : t_last, T[(j + D - 1) % D]);
        compute_tile(t, T[j]);
        t += NPW;
        more = t < t_end;
      }
    }
  }
}

template <int TM, int NSLAB, bool DUAL>
static int launch_pw2(const ConvArgs& a, hipStream_t s) {
  auto fn = conv_pw_kernel<TM, NSLAB, DUAL>;
  const size_t lds = (size_t)a.hdr_bytes;
  if (lds > 64 * 1024) return 1;
  const int n_t32 = (a.g.n_pix + 31) / 32;
  const int M = a.n_mtiles;
  constexpr int NPW = 8 / (TM / 32);
  // ~2 blocks per CU; chunks come in multiples of 8 (one per XCD); every pixel stream gets >= 2 tiles when possible
  int k = 512 / (8 * M);
  if (k < 1) k = 1;
  int k_need = (n_t32 + 8 * NPW * 2 - 1) / (8 * NPW * 2);
  if (k_need < 1) k_need = 1;
  if (k > k_need) k = k_need;
  const int chunks = 8 * k;
  int tpc = (n_t32 + chunks - 1) / chunks;
  tpc = (tpc + NPW - 1) / NPW * NPW;               // equal work for the pixel streams of a block
  TF2_LAUNCH_NAME("conv_pw_kernel<TM%d,%d slabs,%s>", TM, NSLAB, DUAL ? "dual" : "single");
  TF2_LAUNCH(fn, dim3(8 * M * k), dim3(512), lds, s, a, n_t32, tpc);
  return launch_ok() ? 0 : -1;
}

// Does the layer qualify?  `dense` = every m-tile's entry list is exactly slabs 0..nslab-1 (checked by the caller on
// the host copy of the packed image).
bool conv_pw_eligible(const ConvArgs& a, int TM, int nslab, int k, int dense, int max_slab, long min_pix) {
  const ConvGeom& g = a.g;
  if (k != 1 || g.stride != 1 || (g.pad_h | g.pad_w) != 0 || g.H * g.W != g.OHW) return false;
  // two-slab layers (K = 128) measured slower here than in conv_mfma2 (twice the weight registers, half the
  // occupancy headroom): one slab only unless asked
  if (nslab > max_slab) return false;
  if (!dense || nslab < 1 || nslab > 2 || g.Cp_in != nslab * 64) return false;
  if (a.n_phases > 2 || (a.n_phases == 2 && !a.dual)) return false;
  // a wave streams 32-pixel tiles one after the other: with few pixels (batch 1-2) the ring kernel's many short blocks finish
  // sooner (batch-1 latency 436 -> 428 us without this kernel, batch 32 49.1 -> 48.8 k img/s)
  if (g.n_pix < min_pix) return false;
  return TM == 128 || TM == 64;
}

int launch_conv_pw(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nslab = a.nslab;
  const bool dual = a.dual != 0;
  if (TM == 128) {
    if (nslab == 1) return dual ? launch_pw2<128, 1, true>(a, s) : launch_pw2<128, 1, false>(a, s);
    return dual ? launch_pw2<128, 2, true>(a, s) : launch_pw2<128, 2, false>(a, s);
  }
  if (nslab == 1) return dual ? launch_pw2<64, 1, true>(a, s) : launch_pw2<64, 1, false>(a, s);
  return dual ? launch_pw2<64, 2, true>(a, s) : launch_pw2<64, 2, false>(a, s);
}

}  // namespace tf2
