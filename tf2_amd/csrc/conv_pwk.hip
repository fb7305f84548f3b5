// conv_pwk.hip -- pointwise (1x1, unpadded, stride 1 or 2) INT8 convolution of SHORT K (128 .. 512 input channels) with the pixel
// tile's whole K extent resident in LDS (gfx950).  Round 6.
//
// Why: with batches in flight the step waits for epilogue / VALU work first (tools/probe_pipes_inflight.py: -12.6 % without the
// epilogues' arithmetic, -2.3 % without the MFMAs), and the ring kernel spends 12-14.5 VALU instructions per output on exactly these
// rows -- ResNet-50's 256 -> 64, 256 -> 128 | 512, 128 -> 512, 512 -> 256 | 1024 | 2048 layers: 29 % of the step's VALU instructions --
// where the requantisation needs 4-7: a 128 x 128 tile with four K steps pays a block prologue, gather arithmetic and DMA issue per
// step that four steps do not amortise, re-fetches the activations once per 128 output channels, and two-window layers hold two
// accumulator sets.  Here (conv_bfirst's expand phase as a layer of its own):
//
//   * a block walks pixel tiles of TN = 128 (K <= 256) or 64 (K = 512) pixels; a tile's activations -- all K / 64 slabs -- go
//     global -> LDS ONCE by LDS-DMA ([slab][pixel][64] swizzled, two buffers: the next tile lands during the current one's last
//     epilogue) and serve EVERY output channel: M / (32 WM) passes over the resident tile;
//   * a wave owns 32 output channels per pass and NJ column tiles of the pixel tile (8 waves = WM channel groups x 8 / WM pixel
//     groups); weight fragments global -> registers one K step ahead (L2-resident, read-only: nothing is held across passes, nothing
//     spills -- spilled registers are dirty lines that reach HBM, profiles/r06_experiments.txt item 18);
//   * two-window layers are swept window by window into ONE accumulator set with the Horner shift in between: the B operand sits
//     in LDS, reading it twice is cheap;
//   * a wave's 32 header rows (requantisation parameters, final shifts, Horner shifts) are copied per pass into its own 1 KB of LDS in
//     the form requant_epilogue.h reads (an m-tile image of 32 rows); epilogue: column tiles in pairs (parameter rows read once per
//     pair), residual tiles loaded one pair ahead, 16-byte NHWC stores; addresses = kernel-argument base + one 32-bit offset.
//
// Arithmetic, packed image and epilogue are conv_mfma2.hip's (pe.cl:27-43 shift-accumulate as exponent-window int8 GEMMs, pe.cl:185-203
// requantisation, relu.cl:54, feature_writer.cl:88-122 residual); bit-identical to it (tests/test_gpu_parity.py runs both forms).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {
constexpr int kPwkTileBytes = 32 * 1024;                 // one pixel tile's activations: TN pixels x K bytes
constexpr int kPwkHdrSlot = 1024;                        // a wave's 32 header rows (rows | lo | dshift[2]: 7 x 32 words = 896 bytes)

// LDS-DMA hidden from the compiler's wait-count pass (conv_bband.hip bb_dma16): the only waits for these are the vmcnt(0) at a
// tile's start, written out below
__device__ __forceinline__ void pwk_dma16(const int8_t* base, unsigned off, int8_t* lds_dst) {
  const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(l) : "memory", "m0");
}
}  // namespace

// WM: channel groups of 32 per pass (waves along channels); NT: 32-pixel column tiles of a pixel tile (4: TN = 128, 2: TN = 64)
template <int WM, int NT, bool DUAL>
__global__ __launch_bounds__(512, 4) void conv_pwk_kernel(ConvArgs a, int n_tiles, int tm, int csplit) {
  constexpr int WN = 8 / WM, NJ = NT / WN, TN = 32 * NT;
  static_assert(WM * WN == 8 && NJ * WN == NT && NJ >= 1, "wave grid");
  constexpr int NWIN = DUAL ? 2 : 1;
  __shared__ __attribute__((aligned(1024))) int8_t tileb[2][kPwkTileBytes];
  __shared__ __attribute__((aligned(16))) int8_t hdrb[8][kPwkHdrSlot];

  const ConvGeom g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5, frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);        // a lane's fragment of pixel frow of a [32][64] tile, K half 0
  const int KS = a.nslab;                                // 64-byte K slabs (2 .. 8; TN * KS * 64 <= kPwkTileBytes: launcher-checked)
  // (csplit blocks share a pixel tile: block part cp runs passes [cp, cp + 1) * n_pass / csplit -- small maps with many output channels)
  const int cpart = (int)blockIdx.x % csplit;
  const int n_pass = a.Np / (32 * WM) / csplit, pass0 = cpart * n_pass;
  const int tms = tm == 128 ? 7 : 6;
  const unsigned a_lane_off = (unsigned)(frow * 64 + half * 16);
  int* const prm = reinterpret_cast<int*>(hdrb[wave]);

  // ---- the producer side: tile t's activations -> tileb[buf].  A wave issues units u = wave, wave + 8, ..: (16-pixel group, slab)
  const bool contiguous = g.stride == 1 && g.OHW == g.H * g.W;          // output pixel index == input pixel index
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    constexpr int NG = TN / 16;
    const int n_units = NG * KS;
    for (int u = wave; u < n_units; u += 8) {
      const int s = u / NG, grp = u - s * NG;
      const int p = t * TN + grp * 16 + drow;
      unsigned off;
      const int8_t* base = a.x;
      if (p < g.n_pix) {
        int ip = p;
        if (!contiguous) {
          const int b = fast_div(p, g.ohw_m, g.ohw_s);
          const int rem = p - b * g.OHW;
          const int oh = fast_div(rem, g.ow_m, g.ow_s);
          const int ow = rem - oh * g.OW;
          ip = (b * g.H + oh * g.stride) * g.W + ow * g.stride;
        }
        off = (unsigned)ip * (unsigned)g.Cp_in + (unsigned)(s * 64 + chunk * 16);
      } else {
        off = (unsigned)(chunk * 16);
      }
      // (pixels past the launch: the zero page -- a different base, so the whole wave-instruction takes it only when every lane is out;
      //  a group straddling the end reads pixel n_pix - 1's bytes instead: never stored)
      const bool all_out = t * TN + grp * 16 >= g.n_pix;
      if (all_out) base = a.zero;
      else if (p >= g.n_pix) off = (unsigned)(g.n_pix - 1) * (unsigned)g.Cp_in + (unsigned)(s * 64 + chunk * 16);
      pwk_dma16(base, off, tileb[buf] + s * (TN * 64) + grp * 1024);
    }
  };

  const int tstride = (int)gridDim.x / csplit;
  int t = (int)blockIdx.x / csplit;
  if (t >= n_tiles) return;
  issue_tile(t, 0);
  // (the epilogue's form -- residual, FAST rows -- is chosen ONCE, outside the tile loop: four forms inside the loop body kept their common
  //  lane values alive across all of them, in scratch -- conv_bneck's lesson)
  auto run = [&](auto has_res_c, auto fast_c) __attribute__((always_inline)) {
  constexpr bool HAS_RES = decltype(has_res_c)::value;
  constexpr bool FAST = decltype(fast_c)::value;
  int it = 0;
#pragma unroll 1
  for (; t < n_tiles; t += tstride, it++) {
    const int8_t* const B0 = tileb[it & 1];
    // tile t landed in every wave (and every store / load this wave issued before); nobody reads the other buffer any more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int px_t = t * TN;
#pragma unroll 1
    for (int ps = 0; ps < n_pass; ps++) {
      const int ch = ((pass0 + ps) * WM + wm) * 32;                  // first of the wave's 32 output channels of this pass
      const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
      // this wave's header rows -> its LDS slot as an m-tile image of 32 rows: rows {bias | dbl, alpha, addend64} | lo | dshift[P]
      {
        const int8_t* hsrc = reinterpret_cast<const int8_t*>(a.hdr) + (size_t)mt * a.hdr_bytes;
        const int TMr = 1 << tms;
        // rows: 32 x 16 bytes = lanes 0..31; lo: 32 words = lanes 32..39 (16 bytes each); dshift[p]: lanes 40..47, 48..55
        i32x4 v = {0, 0, 0, 0};
        int dst = -1;
        if (lane < 32) { v = *reinterpret_cast<const i32x4*>(hsrc + (size_t)(ro + lane) * 16); dst = lane * 16; }
        else if (lane < 40) { v = *reinterpret_cast<const i32x4*>(hsrc + (size_t)(4 * TMr + ro) * 4 + (lane - 32) * 16); dst = 32 * 16 + (lane - 32) * 16; }
        else if (lane < 40 + 8 * NWIN) {
          const int pz = (lane - 40) >> 3, k = (lane - 40) & 7;
          v = *reinterpret_cast<const i32x4*>(hsrc + (size_t)((5 + pz) * TMr + ro) * 4 + k * 16); dst = (5 + pz) * 32 * 4 + k * 16;
        }
        if (dst >= 0) *reinterpret_cast<i32x4*>(hdrb[wave] + dst) = v;
      }
      // weight fragments of K step v = (window, slab): 16 contiguous bytes of the lane's row per K half
      const int8_t* const wrow = a.w + (size_t)(mt * KS) * (NWIN << tms) * 64 + (size_t)ro * 64;
      auto load_a = [&](i32x4 (&f)[2], int v) __attribute__((always_inline)) {
        const int win = DUAL ? (v >= KS ? 1 : 0) : 0, s = v - win * KS;
        const int8_t* p = wrow + ((size_t)(s * NWIN + win) << tms) * 64 + a_lane_off;
        f[0] = *reinterpret_cast<const i32x4*>(p); f[1] = *reinterpret_cast<const i32x4*>(p + 32);
      };
      i32x4 fa[2], fb[2];
      load_a(fa, 0);
      i32x16 acc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0;
      const int nv = NWIN * KS;
      auto step = [&](i32x4 (&cur)[2], i32x4 (&nxt)[2], int v) __attribute__((always_inline)) {
        if (v + 1 < nv) load_a(nxt, v + 1);
        const int win = DUAL ? (v >= KS ? 1 : 0) : 0, s = v - win * KS;
        if (DUAL && v == KS) {
          // Horner step between the windows: acc <<= dshift[1][row]  (weight_pack.cpp: hi window first); the slot's image has 32 rows
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const int* dsh = prm + (kPrmWordsPerRow + 1) * 32 + 4 * half;
#pragma unroll
          for (int G = 0; G < 4; G++) {
            const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
              for (int j = 0; j < NJ; j++) acc[j][G * 4 + r] = (int)((unsigned)acc[j][G * 4 + r] << (d[r] & 31));
          }
        }
        const int8_t* Bs = B0 + s * (TN * 64) + (wn * NJ) * 2048;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          i32x4 bf[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++) bf[j] = *reinterpret_cast<const i32x4*>(Bs + j * 2048 + (fr0 ^ (ks << 5)));
#pragma unroll
          for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur[ks], bf[j], acc[j], 0, 0, 0);
        }
      };
#pragma unroll 1
      for (int v = 0; v < nv; v += 2) {
        step(fa, fb, v);
        if (v + 1 < nv) step(fb, fa, v + 1);
      }
      // the pass's residual tiles (16 contiguous NHWC bytes per lane and column tile): issued behind the pass's last fragment load -- in
      // front of it every fragment wait of the K loop would have waited for them as well (the VM counter retires in order)
      const int chl = ch + 16 * half;
      const bool ch_ok = chl + 16 <= g.y_nvalid;
      const unsigned res_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)px_t * (unsigned)g.res_cp + (unsigned)g.res_off + (unsigned)ch));
      const unsigned reso = (unsigned)(frow * g.res_cp + 16 * half);
      i32x4 rvs[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int tj = wn * NJ + j;
        const int p = px_t + tj * 32 + frow;
        const bool ok = HAS_RES && p < g.n_pix && ch_ok;
        const int8_t* rp = ok ? a.res + (res_u + reso + (unsigned)(tj * 32 * g.res_cp)) : a.zero;
        rvs[j] = *reinterpret_cast<const i32x4*>(rp);
      }
      // the next tile's activations: issued behind the last pass's K loop (no fragment load follows in this tile whose counted wait would
      // have to let them land first -- the VM counter retires in order), they have the epilogue's time
      if (ps + 1 == n_pass && t + tstride < n_tiles) issue_tile(t + tstride, (it + 1) & 1);

      // ---- epilogue: column tiles in pairs --------------------------------------------------------------------------------------
      const unsigned y_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)px_t * (unsigned)g.y_cp + (unsigned)g.y_off + (unsigned)ch));
      const unsigned yo = (unsigned)(frow * g.y_cp + 16 * half);
      const int lo_bound = g.relu ? 0 : -128, rlo = g.add_relu ? 0 : -128;
      {
        auto pair = [&](auto np_c, auto j0_c) __attribute__((always_inline)) {
          constexpr int NP = decltype(np_c)::value, J0 = decltype(j0_c)::value;
          i32x4 rv[NP], outs[NP];
          int a16s[NP][16];
#pragma unroll
          for (int j = 0; j < NP; j++) {
            rv[j] = rvs[J0 + j];
#pragma unroll
            for (int r = 0; r < 16; r++) a16s[j][r] = acc[J0 + j][r];
          }
          requant_tiles16<NP, HAS_RES, 1, FAST>(a16s, outs, prm, 32, 4 * half, lo_bound, rlo, rv, g.dbl_out != 0, g.fast == 2);
#pragma unroll
          for (int j = 0; j < NP; j++) {
            const int tj = wn * NJ + J0 + j;
            const int p = px_t + tj * 32 + frow;
            if (p < g.n_pix && ch_ok) *reinterpret_cast<i32x4*>(a.y + (y_u + yo + (unsigned)(tj * 32 * g.y_cp))) = outs[j];
          }
        };
        if constexpr (NJ >= 2) pair(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        else pair(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        if constexpr (NJ == 4) pair(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
      }
    }
  }
  };
  if (g.fast == 1) { if (g.has_res) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}); }
  else { if (g.has_res) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}); }
}

// Does the layer qualify?  1x1 / pad 0 / stride 1 or 2 (any dilation of a 1x1 is the plain layer), dense entries (every m-tile holds
// slabs 0 .. nslab - 1 in order, one window or dual) in the layer's OWN tiles (no shared storage), K = 2 .. 8 slabs with a pixel tile
// of at least 64 pixels in 32 KB, output channels a multiple of the 32 x WM a pass covers, no fused global average.
static int pwk_wm(int Np) { return Np % 256 == 0 ? 8 : Np % 128 == 0 ? 4 : Np % 64 == 0 ? 2 : 0; }
bool conv_pwk_eligible(const ConvArgs& a, int TM, int k, int dense, long min_pix) {
  const ConvGeom& g = a.g;
  if (k != 1 || (g.pad_h | g.pad_w) != 0 || (g.stride != 1 && g.stride != 2) || g.avg_mult) return false;
  if (!dense || (TM != 64 && TM != 128) || a.nslab < 2 || a.nslab > 8 || g.Cp_in != a.nslab * 64) return false;
  if (a.n_phases > 2 || (a.n_phases == 2 && !a.dual)) return false;
  const int wins = a.dual ? 2 : 1;
  if (a.w_sub_step || a.e_mt_shr || a.e_mt_shl || a.w_ent_bytes != wins * TM * 64 || a.w_win_stride != TM * 64) return false;
  const int wm = pwk_wm(a.Np);
  if (!wm || a.Np % TM != 0) return false;
  if (a.nslab > 4 && wm != 8) return false;               // (K = 512: 64-pixel tiles are instantiated for the eight-group wave grid only)
  if ((long long)g.n_pix * g.y_cp >= (1ll << 32) || (long long)g.H * g.W * (g.n_pix / std::max(1, g.OHW)) * g.Cp_in >= (1ll << 32)) return false;
  if (g.has_res && (long long)g.n_pix * g.res_cp >= (1ll << 32)) return false;
  return g.n_pix >= min_pix;
}

template <int WM, int NT, bool DUAL>
static int launch_pwk2(const ConvArgs& a, int TM, hipStream_t s) {
  constexpr int TN = 32 * NT;
  if ((size_t)TN * a.nslab * 64 > (size_t)kPwkTileBytes) return 1;
  auto fn = conv_pwk_kernel<WM, NT, DUAL>;
  const int n_tiles = (a.g.n_pix + TN - 1) / TN;
  // two blocks per CU; every block walks the same number of tiles (+- 1); launches of fewer tiles than that split the channel passes
  // over 2 / 4 / 8 blocks per tile (the tile's activations are fetched once per part: L2)
  const int per = (n_tiles + 511) / 512;
  const int tb = (n_tiles + per - 1) / per;
  const int n_pass = a.Np / (32 * WM);
  int csplit = 1;
  while (tb * csplit * 2 <= 512 && n_pass % (csplit * 2) == 0 && csplit < 8) csplit *= 2;
  const int grid = tb * csplit;
  TF2_LAUNCH_NAME("conv_pwk_kernel<%d channel groups x %d pixels,%d slabs,%s>%s", WM, TN, a.nslab, DUAL ? "dual" : "single",
                  csplit == 1 ? "" : csplit == 2 ? " (2 blocks per tile)" : csplit == 4 ? " (4 blocks per tile)" : " (8 blocks per tile)");
  TF2_LAUNCH(fn, dim3(grid), dim3(512), 0, s, a, n_tiles, TM, csplit);
  return launch_ok() ? 0 : -1;
}

int launch_conv_pwk(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int wm = pwk_wm(a.Np);
  const bool dual = a.dual != 0;
#define TF2_PWK(WM_, NT_) do { return dual ? launch_pwk2<WM_, NT_, true>(a, TM, s) : launch_pwk2<WM_, NT_, false>(a, TM, s); } while (0)
  if (a.nslab > 4) { if (wm == 8) TF2_PWK(8, 2); return 1; }
  if (wm == 8) TF2_PWK(8, 4);
  if (wm == 4) TF2_PWK(4, 4);
  if (wm == 2) TF2_PWK(2, 4);
#undef TF2_PWK
  return 1;
}

}  // namespace tf2
