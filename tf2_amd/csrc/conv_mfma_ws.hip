// conv_mfma_ws.hip -- weight-stationary implicit-GEMM for SHORT-K pointwise layers (gfx950).
//
// The 1x1 layers of ResNet's early stages (K = 64..256 channels) are byte-bound: one 128x128 output
// tile needs 16 MFMAs but moves 8 KB in, 16 KB out and 16 KB of residual.  With one tile per block
// (conv_mfma2.hip) the block's life is a chain of exposed latencies -- header DMA, first operand
// DMA, a 1-2 step K loop, epilogue -- for 40 KB of traffic.  Here a block keeps its m-tile's
// header and ALL of its weight tiles resident in LDS and walks a run of consecutive pixel tiles:
//
//     wait (everything for tile t landed) ; barrier
//     issue LDS-DMA for tile t+1: its activation slabs (each distinct slab once, also when several
//         Horner phases use it) and its residual tile, into the other half of a double buffer
//     tile t: phases x slabs of v_mfma_i32_32x32x32_i8, then the fused epilogue (residual read
//         from LDS), 16-byte NHWC stores
//
// so the DMA latency of tile t+1 hides behind the MFMAs + epilogue of tile t, and header /
// weights / prologue are paid once per block instead of once per tile.  Arithmetic, packed image,
// LDS swizzle and epilogue are those of conv_mfma2.hip (reference citations there).
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

struct WsGeom {              // derived on the host, passed beside ConvArgs
  int32_t tiles_per_block;   // consecutive pixel tiles one block walks
  int32_t n_ptiles;          // pixel tiles in total
  int32_t n_ent_max;         // resident weight tiles per block (>= every m-tile's entry count)
  int32_t n_dist_max;        // distinct slabs per m-tile (B buffer slots)
  int32_t linear;            // 1: input pixel index == output pixel index (k=1, stride 1, pad 0)
};

// WM x WN waves of 64x64 output tiles; TM = 64*WM channels, TN = 64*WN pixels.
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_mfma_ws_kernel(ConvArgs a, WsGeom w) {
  constexpr int TM = WM * 64, TN = WN * 64;
  constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, R_BYTES = TM * TN;
  constexpr int AI = TM / 64, BI = TN / 64;          // LDS-DMA instructions per wave per tile / slab
  constexpr int RI = R_BYTES / 4096;                 // residual tile: 1 KiB instructions per wave
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  // LDS map: [A: n_ent_max tiles][B: 2 x n_dist_max slabs][R: 2 residual tiles (if any)][header]
  const ConvGeom& g = a.g;
  int8_t* const Abase = lds;
  int8_t* const Bbase = Abase + (size_t)w.n_ent_max * A_BYTES;
  int8_t* const Rbase = Bbase + (size_t)2 * w.n_dist_max * B_BYTES;
  int* const prm = reinterpret_cast<int*>(Rbase + (g.has_res ? 2 * R_BYTES : 0));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int P = a.n_phases;
  int* const dsh = prm + kPrmWordsPerRow * TM;
  int* const steps = dsh + P * TM;
  int* const goff = steps + a.max_ent;
  int* const eslot = goff + 8 * a.max_ent;           // after goff[4*max_ent] and ghw[4*max_ent]
  int* const dfirst = eslot + a.max_ent;

  // block -> (m-tile, run of pixel tiles); m-tile fastest so that the blocks sharing activations
  // are neighbours (same XCD after the remap)
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mtile = bid % a.n_mtiles;
  const int run = bid / a.n_mtiles;
  const int t_begin = run * w.tiles_per_block;
  const int t_end = (t_begin + w.tiles_per_block < w.n_ptiles) ? t_begin + w.tiles_per_block : w.n_ptiles;
  const int e_begin = a.e_start[mtile];
  const int n_ent = a.e_start[mtile + 1] - e_begin;

  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);             // see conv_mfma2.hip
  const int a_lane_off = (lane >> 2) * 64 + chunk * 16;
  const int half = lane >> 5;

  // ---- once per block: header + all weight tiles of this m-tile ------------------------------
  {
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(a.hdr) + (size_t)mtile * a.hdr_bytes + lane * 16;
    int8_t* hdst = reinterpret_cast<int8_t*>(prm);
    for (int i = wave; i * 1024 < a.hdr_bytes; i += 4)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hsrc + i * 1024), TF2_LDS_PTR(hdst + i * 1024), 16, 0, 0);
    for (int e = 0; e < n_ent; e++) {
      const int8_t* wsrc = a.w + (size_t)(e_begin + e) * A_BYTES + a_lane_off;
#pragma unroll
      for (int j = 0; j < AI; j++) {
        const int grp = wave + 4 * j;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + grp * 1024), TF2_LDS_PTR(Abase + (size_t)e * A_BYTES + grp * 1024), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // header readable (the slab tables are needed to issue tile 0)
  asm volatile("" ::: "memory");
  const int n_dist = __builtin_amdgcn_readfirstlane(steps[a.max_ent - 1]);

  // issue every DMA of pixel tile t into buffer half `buf`
  auto issue_tile = [&](int t, int buf) {
    const int px0 = t * TN;
    // activation rows this lane fetches
    const int8_t* bptr[BI];
    bool bok[BI];
#pragma unroll
    for (int j = 0; j < BI; j++) {
      const int p = px0 + (wave + 4 * j) * 16 + (lane >> 2);
      bok[j] = p < g.n_pix;
      if (w.linear) {
        bptr[j] = a.x + (size_t)(bok[j] ? p : 0) * g.Cp_in;
      } else {
        const int pp = bok[j] ? p : 0;
        const int b = fast_div(pp, g.ohw_m, g.ohw_s);
        const int rem = pp - b * g.OHW;
        const int oh = fast_div(rem, g.ow_m, g.ow_s);
        const int ow = rem - oh * g.OW;
        bptr[j] = a.x + ((long long)b * g.H * g.W + (long long)(oh * g.stride) * g.W + ow * g.stride) * g.Cp_in;
      }
    }
    for (int d = 0; d < n_dist; d++) {
      const int off = goff[dfirst[d] * 4 + chunk];
      int8_t* const slot = Bbase + (size_t)(buf * w.n_dist_max + d) * B_BYTES;
#pragma unroll
      for (int j = 0; j < BI; j++) {
        const int grp = wave + 4 * j;
        const int8_t* src = (off >= 0 && bok[j]) ? bptr[j] + off : a.zero;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(slot + grp * 1024), 16, 0, 0);
      }
    }
    if (g.has_res) {
      // residual tile [TN pixels][TM bytes]: one 1 KiB instruction covers 1024/TM pixels
      constexpr int CPP = TM / 16;           // 16-byte chunks per pixel
      constexpr int PPI = 64 / CPP;          // pixels per instruction
#pragma unroll
      for (int j = 0; j < RI; j++) {
        const int grp = wave + 4 * j;
        const int pl = grp * PPI + lane / CPP;
        const int c16 = lane % CPP;
        const int px = px0 + pl;
        const int chl = mtile * TM + c16 * 16;
        const bool ok = px < g.n_pix && chl + 16 <= g.y_nvalid;
        const int8_t* src = ok ? a.res + (size_t)px * g.res_cp + g.res_off + chl : a.zero;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(Rbase + (size_t)buf * R_BYTES + grp * 1024), 16, 0, 0);
      }
    }
  };

  auto phase_shift = [&](i32x16 (&acc)[2][2], int p) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int rb = wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + p * TM + rb + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j][G * 4 + r] = (int)((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31));
      }
    }
  };

  const int lo_bound = g.relu ? 0 : -128;
  const int rlo = g.add_relu ? 0 : -128;

  // Output stores are deferred by one tile: vmcnt also counts stores on gfx9, so stores issued right
  // before the next tile's vmcnt(0) would put their write latency on the critical path.
  i32x4 pend[2][2];
  int pend_px0 = -1;
  auto flush_pending = [&]() {
    if (pend_px0 < 0) return;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int px = pend_px0 + wn * 64 + j * 32 + (lane & 31);
        const int chl = mtile * TM + wm * 64 + i * 32 + 16 * half;
        if (px < g.n_pix && chl + 16 <= g.y_nvalid)
          *reinterpret_cast<i32x4*>(a.y + (size_t)px * g.y_cp + g.y_off + chl) = pend[i][j];
      }
  };

  if (t_begin < t_end) issue_tile(t_begin, 0);
  int buf = 0;
  for (int t = t_begin; t < t_end; t++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t (issued a whole tile earlier) has landed
    __builtin_amdgcn_s_barrier();                      // ... for every wave; buffers of tile t-1 are free
    asm volatile("" ::: "memory");
    if (t + 1 < t_end) issue_tile(t + 1, buf ^ 1);
    flush_pending();                                   // tile t-1's outputs, in flight during tile t

    // ---- all phases x slabs of this tile ----
    i32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
    int phase = 0;
    int next_b = __builtin_amdgcn_readfirstlane(steps[0]);
    for (int e = 0; e < n_ent; e++) {
      while (e == next_b) {
        phase++; phase_shift(acc, phase);
        next_b = __builtin_amdgcn_readfirstlane(steps[phase]);
      }
      const int8_t* A = Abase + (size_t)e * A_BYTES;
      const int8_t* B = Bbase + (size_t)(buf * w.n_dist_max + __builtin_amdgcn_readfirstlane(eslot[e])) * B_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int c = ks * 2 + (lane >> 5);
        i32x4 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const int row = wm * 64 + i * 32 + (lane & 31);
          af[i] = *reinterpret_cast<const i32x4*>(A + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int row = wn * 64 + j * 32 + (lane & 31);
          bf[j] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    while (phase + 1 < P) { phase++; phase_shift(acc, phase); }

    // ---- epilogue (arithmetic as in conv_mfma2.hip) ----
    const int px0 = t * TN;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int rb = wm * 64 + i * 32;
      const int tile_ch = mtile * TM + rb;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int pl = wn * 64 + j * 32 + (lane & 31);
        const int px = px0 + pl;
        i32x4 rv = {0, 0, 0, 0};
        if (g.has_res) rv = *reinterpret_cast<const i32x4*>(Rbase + (size_t)buf * R_BYTES + (size_t)pl * TM + rb + 16 * half);
        int a16[16];
#pragma unroll
        for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
        if (g.fast) pend[i][j] = g.has_res ? requant_tile16<true, 0, true>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, rv)
                                           : requant_tile16<false, 0, true>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, rv);
        else pend[i][j] = g.has_res ? requant_tile16<true>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, rv)
                                    : requant_tile16<false>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, rv);
        (void)px; (void)tile_ch;
      }
    }
    pend_px0 = px0;
    buf ^= 1;
  }
  flush_pending();
}

template <int WM, int WN>
static int launch_ws(const ConvArgs& a, const WsGeom& w, size_t lds, int blocks, hipStream_t s) {
  static bool attr_set = false;
  auto fn = conv_mfma_ws_kernel<WM, WN>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    attr_set = true;
  }
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, s, a, w);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Returns 1 when the layer does not qualify (caller falls back to conv_mfma2), 0 on launch, <0 on error.
// `n_ent_max` / `n_dist_max` come from the packed image (host side knows every m-tile's lists).
int launch_conv_mfma_ws(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const ConvGeom& g = a.g;
  if (a.k != 1 || g.pad_h || g.pad_w || a.n_mtiles > kMaxMtiles) return 1;
  int n_ent_max = 0;
  for (int mt = 0; mt < a.n_mtiles; mt++) n_ent_max = n_ent_max > a.e_start[mt + 1] - a.e_start[mt] ? n_ent_max : a.e_start[mt + 1] - a.e_start[mt];
  const int n_dist_max = a.nslab;                    // distinct slabs of an m-tile <= slabs of the layer
  if (n_ent_max < 1 || n_ent_max > 8 || n_dist_max > 4) return 1;
  const int TN = TM == 128 ? 128 : 256;
  const size_t lds = (size_t)n_ent_max * TM * 64 + (size_t)2 * n_dist_max * TN * 64 + (g.has_res ? (size_t)2 * TM * TN : 0) +
                     (size_t)a.hdr_bytes + 64;
  if (lds > 156 * 1024) return 1;
  WsGeom w{};
  w.n_ptiles = (g.n_pix + TN - 1) / TN;
  // enough blocks to cover the chip twice over, each walking a run of consecutive pixel tiles
  int runs = (2 * 256 + a.n_mtiles - 1) / a.n_mtiles;
  if (const char* e = getenv("TF2_AMD_WS")) if (e[0] == '1') runs = w.n_ptiles / 3 > 0 ? w.n_ptiles / 3 : 1;   // tests: force runs of ~3 tiles
  if (runs > w.n_ptiles) runs = w.n_ptiles;
  w.tiles_per_block = (w.n_ptiles + runs - 1) / runs;
  if (w.tiles_per_block < 2) return 1;               // nothing to amortise: one tile per block
  runs = (w.n_ptiles + w.tiles_per_block - 1) / w.tiles_per_block;
  w.n_ent_max = n_ent_max; w.n_dist_max = n_dist_max;
  w.linear = (g.stride == 1 && g.H == g.OH && g.W == g.OW) ? 1 : 0;
  const int blocks = runs * a.n_mtiles;
  if (TM == 128) return launch_ws<2, 2>(a, w, lds, blocks, s);
  if (TM == 64) return launch_ws<1, 4>(a, w, lds, blocks, s);
  return 1;
}

}  // namespace tf2
