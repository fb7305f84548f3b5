// conv_pwk.hip -- pointwise (1x1, unpadded, stride 1 or 2) INT8 convolution of SHORT K (128 or 256 input channels) with the WEIGHTS of a
// wave's 32 output channels resident in registers and the block's pixels resident in LDS (gfx950).  Round 6.
//
// Why: with batches in flight the step waits for epilogue / VALU work first (tools/probe_pipes_inflight.py: -12.6 % without the
// epilogues' arithmetic, -2.3 % without the MFMAs), and the ring kernel spends 12-14.5 VALU instructions per output on exactly these
// rows -- ResNet-50's 256 -> 64, 256 -> 128 | 512, 128 -> 512, 256 -> 1024 layers -- where the requantisation needs 4-7: a 128 x 128
// tile with two or four K steps pays a block prologue, gather arithmetic and DMA issue per step that so few steps do not amortise.
//
// First form (git history: the pixel tile's K extent in LDS, weight fragments global -> registers ONE K step ahead, passes over the
// channel groups inside a loop over pixel tiles): bit-exact and 40-60 % SLOWER per launch than the ring kernel -- every K step of every
// (tile, pass) waited for an L2 round trip that a 4-MFMA step cannot cover (profiles/r06_experiments.txt item 21).  This form turns
// the loops inside out (conv_pw.hip's order with the B operand in LDS):
//
//   * a block owns T pixel tiles of 128 pixels (T * K <= 64 KB): ALL their K slabs go global -> LDS once by LDS-DMA
//     ([tile][slab][pixel][64] swizzled), one wait, one barrier -- two blocks per CU cover each other's load;
//   * a pass = WM channel groups of 32 (waves along channels) x 8 / WM pixel groups; at the start of a pass a wave loads ALL its weight
//     fragments (32 rows x K x windows: 16 .. 64 registers) -- ONE L2 round trip per pass, issued behind the previous pass's last MFMA
//     so that the epilogue covers it -- and then sweeps its column tiles with no memory instruction but ds_read in the K loop;
//   * two-window layers are swept window by window into ONE accumulator set with the Horner shift in between;
//   * a wave's 32 header rows (requantisation parameters, final shifts, Horner shifts) are copied per pass into its own 1 KB of LDS in
//     the form requant_epilogue.h reads (an m-tile image of 32 rows); residual tiles are loaded in front of a column group's MFMAs,
//     16-byte NHWC stores; addresses = kernel-argument base + one 32-bit offset.
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
constexpr int kPwkPixBytes = 64 * 1024;                  // the block's pixels: T tiles x 128 pixels x K bytes
constexpr int kPwkHdrSlot = 1024;                        // a wave's 32 header rows (rows | lo | dshift[2]: 7 x 32 words = 896 bytes)

// LDS-DMA hidden from the compiler's wait-count pass (conv_bband.hip bb_dma16): the only wait for these is the vmcnt(0) in front of
// the block's one barrier, written out below
__device__ __forceinline__ void pwk_dma16(const int8_t* base, unsigned off, int8_t* lds_dst) {
  const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(l) : "memory", "m0");
}
}  // namespace

// KS: 64-byte K slabs of the layer (2 or 4); WM: channel groups of 32 per pass (waves along channels; 4 / WM pixel groups).
// Four waves per block, two blocks per CU: two waves per SIMD with 256 registers each -- the resident fragments (up to 64 registers)
// beside two accumulator sets and the epilogue's temporaries (eight waves at 128 registers parked 160-470 bytes per lane in scratch)
template <int KS, int WM, bool DUAL>
__global__ __launch_bounds__(256, 2) void conv_pwk_kernel(ConvArgs a, int n_tiles, int tm, int T, int csplit) {
  constexpr int WN = 4 / WM;
  constexpr int NWIN = DUAL ? 2 : 1, NV = KS * NWIN;      // K steps of a column tile: (window, slab)
  constexpr int NJ = 2;                                  // column tiles (accumulator sets) per wave at a time, beside NV * 8 fragment registers
  constexpr int TILE = KS * 128 * 64;                    // bytes of one 128-pixel tile: [slab][pixel][64]
  static_assert(WM * WN == 4 && TILE <= kPwkPixBytes, "wave grid / tile");
  __shared__ __attribute__((aligned(1024))) int8_t pixb[kPwkPixBytes];
  __shared__ __attribute__((aligned(16))) int8_t hdrb[4][kPwkHdrSlot];

  const ConvGeom g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5, frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);        // a lane's fragment of pixel frow of a [32][64] tile, K half 0
  // (csplit blocks share a pixel-tile group: block part cp runs passes [cp, cp + 1) * n_pass / csplit -- small maps with many output channels)
  const int cpart = (int)blockIdx.x % csplit;
  const int n_pass = a.Np / (32 * WM) / csplit, pass0 = cpart * n_pass;
  const int t0 = ((int)blockIdx.x / csplit) * T;         // first 128-pixel tile of the block
  const int nt = n_tiles - t0 < T ? n_tiles - t0 : T;    // its tiles (>= 1: the launcher's grid)
  const int n_ct = 4 * nt;                               // its 32-pixel column tiles
  const int tms = tm == 128 ? 7 : 6;
  const unsigned a_lane_off = (unsigned)(frow * 64 + half * 16);
  int* const prm = reinterpret_cast<int*>(hdrb[wave]);

  // ---- the block's pixels -> LDS: units u = wave, wave + 8, ..: (tile, slab, 16-pixel group) -------------------------------------------
  {
    const bool contiguous = g.stride == 1 && g.OHW == g.H * g.W;        // output pixel index == input pixel index
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
    const int n_units = nt * (KS * 8);
    for (int u = wave; u < n_units; u += 4) {
      const int i = u / (KS * 8), r = u - i * (KS * 8), s = r >> 3, grp = r & 7;
      const int pg = (t0 + i) * 128 + grp * 16;
      const int p = pg + drow;
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
      //  a group straddling the end reads the layer's first pixel instead: never stored)
      if (pg >= g.n_pix) base = a.zero;
      else if (p >= g.n_pix) off = (unsigned)(s * 64 + chunk * 16);
      pwk_dma16(base, off, pixb + i * TILE + s * (128 * 64) + grp * 1024);
    }
  }

  // weight fragments of a pass: K step v = (window, slab), 16 contiguous bytes of the lane's row per K half
  struct Afr { i32x4 k[NV][2]; };
  auto load_a = [&](Afr& f, int ch) __attribute__((always_inline)) {
    const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
    const int8_t* const wrow = a.w + (size_t)(mt * KS) * (NWIN << tms) * 64 + (size_t)ro * 64 + a_lane_off;
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int win = v / KS, sl = v - win * KS;
      const int8_t* p = wrow + ((size_t)(sl * NWIN + win) << tms) * 64;
      f.k[v][0] = *reinterpret_cast<const i32x4*>(p); f.k[v][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
  };
  // a wave's 32 header rows as an m-tile image of 32 rows in its LDS slot: rows {bias | dbl, alpha, addend64} | lo | dshift[P].
  // rows: 32 x 16 bytes = lanes 0..31; lo: 32 words = lanes 32..39 (16 bytes each); dshift[p]: lanes 40..47, 48..55
  int hdst = -1;
  if (lane < 32) hdst = lane * 16;
  else if (lane < 40) hdst = 32 * 16 + (lane - 32) * 16;
  else if (lane < 40 + 8 * NWIN) hdst = (5 + ((lane - 40) >> 3)) * 32 * 4 + ((lane - 40) & 7) * 16;
  auto load_hdr = [&](int ch) __attribute__((always_inline)) -> i32x4 {
    const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(a.hdr) + (size_t)mt * a.hdr_bytes;
    const int TMr = 1 << tms;
    const int8_t* src = hsrc;                              // (lanes without a piece read the m-tile's first bytes: never written)
    if (lane < 32) src = hsrc + (size_t)(ro + lane) * 16;
    else if (lane < 40) src = hsrc + (size_t)(4 * TMr + ro) * 4 + (lane - 32) * 16;
    else if (lane < 40 + 8 * NWIN) src = hsrc + (size_t)((5 + ((lane - 40) >> 3)) * TMr + ro) * 4 + ((lane - 40) & 7) * 16;
    return *reinterpret_cast<const i32x4*>(src);
  };

  // the first pass's fragments and header rows travel beside the pixel DMAs
  Afr fa;
  load_a(fa, (pass0 * WM + wm) * 32);
  i32x4 hv = load_hdr((pass0 * WM + wm) * 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // (the epilogue's form -- residual, FAST rows -- is chosen ONCE, outside the loops: four forms inside the loop body kept their common
  //  lane values alive across all of them, in scratch -- conv_bneck's lesson)
  auto run = [&](auto has_res_c, auto fast_c) __attribute__((always_inline)) {
    constexpr bool HAS_RES = decltype(has_res_c)::value;
    constexpr bool FAST = decltype(fast_c)::value;
    const int lo_bound = g.relu ? 0 : -128, rlo = g.add_relu ? 0 : -128;
    const unsigned reso = (unsigned)(frow * g.res_cp + 16 * half);
    const unsigned yo = (unsigned)(frow * g.y_cp + 16 * half);
#pragma unroll 1
    for (int ps = 0; ps < n_pass; ps++) {
      const int ch = ((pass0 + ps) * WM + wm) * 32;        // first of the wave's 32 output channels of this pass
      if (hdst >= 0) *reinterpret_cast<i32x4*>(hdrb[wave] + hdst) = hv;
      const int chl = ch + 16 * half;
      const bool ch_ok = chl + 16 <= g.y_nvalid;
      const unsigned res_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(t0 * 128) * (unsigned)g.res_cp + (unsigned)g.res_off + (unsigned)ch));
      const unsigned y_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(t0 * 128) * (unsigned)g.y_cp + (unsigned)g.y_off + (unsigned)ch));
#pragma unroll 1
      for (int k0 = wn; k0 < n_ct; k0 += WN * NJ) {        // this wave's column tiles k0, k0 + WN (NJ at a time)
        // the group's residual tiles (16 contiguous NHWC bytes per lane and column tile) land during its MFMAs
        i32x4 rvs[NJ];
        bool okj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int tj = k0 + j * WN;
          const int p = t0 * 128 + tj * 32 + frow;
          okj[j] = tj < n_ct && p < g.n_pix && ch_ok;
          const int8_t* rp = (HAS_RES && okj[j]) ? a.res + (res_u + reso + (unsigned)(tj * 32 * g.res_cp)) : a.zero;
          rvs[j] = *reinterpret_cast<const i32x4*>(rp);
        }
        i32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[j][r] = 0;
        const int8_t* Bj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int tj = (k0 + j * WN) & (kPwkPixBytes / TILE * 4 - 1);      // (a column tile past the block's: some resident tile, never stored)
          Bj[j] = pixb + (tj >> 2) * TILE + (tj & 3) * 2048;
        }
        // B fragments one K step ahead of their MFMAs, no further (sched_barrier: left alone, the scheduler hoists all NV * 2 * NJ reads
        // -- up to 128 registers -- in front of the first MFMA)
        i32x4 bfc[2][NJ], bfn[2][NJ];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int j = 0; j < NJ; j++) bfc[ks][j] = *reinterpret_cast<const i32x4*>(Bj[j] + (fr0 ^ (ks << 5)));
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (v + 1 < NV) {
            const int sn = (v + 1) % KS;
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
              for (int j = 0; j < NJ; j++) bfn[ks][j] = *reinterpret_cast<const i32x4*>(Bj[j] + sn * (128 * 64) + (fr0 ^ (ks << 5)));
          }
          if (DUAL && v == KS) {
            // Horner step between the windows: acc <<= dshift[1][row]  (weight_pack.cpp: hi window first); the slot's image has 32 rows
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
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.k[v][ks], bfc[ks][j], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < NJ; j++) bfc[ks][j] = bfn[ks][j];
        }
        // behind the pass's last MFMA: the next pass's fragments and header rows -- the epilogue below covers their round trip
        if (k0 + WN * NJ >= n_ct && ps + 1 < n_pass) {
          load_a(fa, ch + 32 * WM);
          hv = load_hdr(ch + 32 * WM);
        }
        // ---- epilogue of the group ------------------------------------------------------------------------------------------------
        i32x4 outs[NJ];
        int a16s[NJ][16];
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) a16s[j][r] = acc[j][r];
        requant_tiles16<NJ, HAS_RES, 1, FAST>(a16s, outs, prm, 32, 4 * half, lo_bound, rlo, rvs, g.dbl_out != 0, g.fast == 2);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int tj = k0 + j * WN;
          if (okj[j]) *reinterpret_cast<i32x4*>(a.y + (y_u + yo + (unsigned)(tj * 32 * g.y_cp))) = outs[j];
        }
      }
    }
  };
  if (g.fast == 1) { if (g.has_res) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}); }
  else { if (g.has_res) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}); }
}

// Does the layer qualify?  1x1 / pad 0 / stride 1 or 2 (any dilation of a 1x1 is the plain layer), dense entries (every m-tile holds
// slabs 0 .. nslab - 1 in order, one window or dual) in the layer's OWN tiles (no shared storage), K = 2 or 4 slabs, output channels a
// multiple of the 32 x WM a pass covers, no fused global average.
static int pwk_wm(int Np) { return Np % 128 == 0 ? 4 : Np % 64 == 0 ? 2 : 0; }
bool conv_pwk_eligible(const ConvArgs& a, int TM, int k, int dense, long min_pix) {
  const ConvGeom& g = a.g;
  if (k != 1 || (g.pad_h | g.pad_w) != 0 || (g.stride != 1 && g.stride != 2) || g.avg_mult) return false;
  if (!dense || (TM != 64 && TM != 128) || (a.nslab != 2 && a.nslab != 4) || g.Cp_in != a.nslab * 64) return false;
  if (a.n_phases > 2 || (a.n_phases == 2 && !a.dual)) return false;
  const int wins = a.dual ? 2 : 1;
  if (a.w_sub_step || a.e_mt_shr || a.e_mt_shl || a.w_ent_bytes != wins * TM * 64 || a.w_win_stride != TM * 64) return false;
  const int wm = pwk_wm(a.Np);
  if (!wm || a.Np % TM != 0) return false;
  if ((long long)g.n_pix * g.y_cp >= (1ll << 32) || (long long)g.H * g.W * (g.n_pix / std::max(1, g.OHW)) * g.Cp_in >= (1ll << 32)) return false;
  if (g.has_res && (long long)g.n_pix * g.res_cp >= (1ll << 32)) return false;
  return g.n_pix >= min_pix;
}

static int g_pwk_tiles = 0;                              // pwk_t (test-only): tiles per block, 0 = the launcher's choice
void conv_pwk_set_tiles(int t) { g_pwk_tiles = t; }

template <int KS, int WM, bool DUAL>
static int launch_pwk2(const ConvArgs& a, int TM, hipStream_t s) {
  auto fn = conv_pwk_kernel<KS, WM, DUAL>;
  const int n_tiles = (a.g.n_pix + 127) / 128;
  const int n_pass = a.Np / (32 * WM);
  // blocks per pixel-tile group: the channel passes split over 2 / 4 / 8 blocks while the launch has fewer than two blocks per CU
  // (the group's activations are then fetched once per part: L2); tiles per block: as many as fit the 64 KB while the grid keeps ~400 blocks
  int max_split = 1;
  while (n_pass % (max_split * 2) == 0 && max_split < 8) max_split *= 2;
  int T = kPwkPixBytes / (KS * 128 * 64);
  while (T > 1 && (long)((n_tiles + T - 1) / T) * max_split < 384) T >>= 1;
  if (g_pwk_tiles > 0 && g_pwk_tiles <= kPwkPixBytes / (KS * 128 * 64)) T = g_pwk_tiles;
  const int tb = (n_tiles + T - 1) / T;
  int csplit = 1;
  while (tb * csplit * 2 <= 512 && csplit < max_split) csplit *= 2;
  const int grid = tb * csplit;
  TF2_LAUNCH_NAME("conv_pwk_kernel<%d slabs,%d channel groups,%s> (%d pixels per block%s)", KS, WM, DUAL ? "dual" : "single", 128 * T,
                  csplit == 1 ? "" : csplit == 2 ? ", 2 blocks per tile" : csplit == 4 ? ", 4 blocks per tile" : ", 8 blocks per tile");
  TF2_LAUNCH(fn, dim3(grid), dim3(256), 0, s, a, n_tiles, TM, T, csplit);
  return launch_ok() ? 0 : -1;
}

int launch_conv_pwk(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int wm = pwk_wm(a.Np);
  const bool dual = a.dual != 0;
#define TF2_PWK(KS_, WM_) do { return dual ? launch_pwk2<KS_, WM_, true>(a, TM, s) : launch_pwk2<KS_, WM_, false>(a, TM, s); } while (0)
#define TF2_PWK_KS(KS_) do { if (wm == 4) TF2_PWK(KS_, 4); if (wm == 2) TF2_PWK(KS_, 2); } while (0)
  if (a.nslab == 4) TF2_PWK_KS(4);
  if (a.nslab == 2) TF2_PWK_KS(2);
#undef TF2_PWK_KS
#undef TF2_PWK
  return 1;
}

}  // namespace tf2
