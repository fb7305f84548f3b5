// conv_pwk.hip -- pointwise (1x1, unpadded, stride 1 or 2) INT8 convolution of SHORT K (128 or 256 input channels; 512 built, not the plan's
// choice) by PERSISTENT blocks of four waves: the weights of a wave's 32 output channels resident in registers, pixel tiles streamed
// through two LDS buffers (gfx950).  Round 6; the in-flight plan's kernel for ResNet-50 rows 5, 8, 11 | 12, 14 (option `pwk`).
//
// Why: with batches in flight a launch costs the other three batches what it HOLDS -- registers x time first.  The ring kernel runs these
// rows (256 -> 64, 256 -> 128 | 512 / 2, 128 -> 512) as 392-1568 blocks of eight waves at 128 registers, each living ~9 us of which a third
// is block prologue and first-operand latency, and spends 12-14.5 VALU instructions per output where the requantisation needs 4-7.  This
// kernel runs them as 196-240 blocks of four waves at ~200 registers that walk 3-5 tiles each: a half to a third of the register-file
// time, 6.5 VALU per output.  Alone its launches are SLOWER than the ring kernel's (13.0 against 10.4 us on row 5, 18.1 against 12.3 on
// row 14); in flight the step gains 1.9 % (profiles/r06_experiments.txt item 21: four earlier forms, the rows it lost on, block timelines).
//
// Forms that did not make it (git history): (a) the pixel tile's K extent in LDS, weight fragments global -> registers ONE K step ahead,
// channel passes inside the tile loop -- every K step waited for an L2 round trip that four MFMAs cannot cover (40-60 % slower per launch);
// (b) a pass's fragments in registers, ALL the block's pixels loaded first, then computed -- a one-round grid loads in phase and computes
// in phase (20-45 % slower); (c) this structure at two blocks per CU and one launch per row -- level with the ring kernel; (g) deeper B /
// parameter prefetch -- 25-70 more registers, slower alone and in flight.  The form kept (conv_pw.hip's loop order, B streamed through LDS):
//
//   * a block owns WM x 32 output channels (channel part `part` of the layer) for its whole life: a wave's weight fragments (32 rows x K x
//     windows: 16 .. 64 registers) and its header rows are fetched ONCE, beside the first tile's DMAs;
//   * it walks a stream of 128-pixel tiles (tile = stream, stream + n_streams, ..): a tile's K slabs go global -> LDS by LDS-DMA
//     ([slab][pixel][64] swizzled) into one of two buffers, the next tile's DMAs are issued right behind the barrier that hands over the
//     current one; no memory instruction but ds_read in the K loop; the channel parts of one tile run on the same XCD (ids n_streams apart);
//     the grid aims at ONE block per CU (pwk_slots 256); rows taken only where a block walks several tiles (pwk_units);
//   * two-window layers are swept window by window into ONE accumulator set with the Horner shift in between;
//   * FAST rows without a residual requantise a column group BETWEEN the next group's MFMAs (run_pipe: one wave keeps the matrix pipe and
//     the VALU busy at once); the other rows requantise behind their own MFMAs, residual tiles loaded in front of them;
//   * a wave's 32 header rows (requantisation parameters, final shifts, Horner shifts) sit in its own 1 KB of LDS in the form
//     requant_epilogue.h reads (an m-tile image of 32 rows); 16-byte NHWC stores; addresses = kernel-argument base + one 32-bit offset;
//   * two independent rows of one instantiation (a stage's shortcut convolution | the first 1x1 of its first bottleneck) share a launch
//     (conv_pwk_pair_kernel).
//
// Arithmetic, packed image and epilogue are conv_mfma2.hip's (pe.cl:27-43 shift-accumulate as exponent-window int8 GEMMs, pe.cl:185-203
// requantisation, relu.cl:54, feature_writer.cl:88-122 residual); bit-identical to it (tests/test_gpu_parity.py runs both forms; the whole
// GPU suite passes with the kernel forced onto every eligible row of every network: profiles/r06_pytest_gpu_pwk_everywhere.log).
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
constexpr int kPwkHdrSlot = 1024;                        // a wave's 32 header rows (rows | lo | dshift[2]: 7 x 32 words = 896 bytes)

// LDS-DMA hidden from the compiler's wait-count pass (conv_bband.hip bb_dma16): the only wait for these is the vmcnt(0) in front of
// a tile's barrier, written out below
__device__ __forceinline__ void pwk_dma16(const int8_t* base, unsigned off, int8_t* lds_dst) {
  const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(l) : "memory", "m0");
}
__host__ __device__ constexpr int pwk_tile_px(int ks) { return ks <= 4 ? 128 : 64; }      // pixels of a tile (32 KB at most)
}  // namespace

// KS: 64-byte K slabs of the layer (2, 4 or 8); WM: channel groups of 32 of a block (waves along channels; 4 / WM pixel groups).
// Four waves per block at up to 256 registers (launch bounds (256, 2); the two-window K = 512 instantiation (256, 1): 293) -- the resident
// fragments (up to 64 registers) beside two accumulator sets, the pending group and the epilogue's temporaries (eight waves at 128 registers
// parked 160-470 bytes per lane in scratch)
template <int KS, int WM, bool DUAL>
__device__ __forceinline__ void conv_pwk_body(const ConvArgs& a, const int n_tiles, const int tm, const int n_streams, const int pipe, const int bid,
                                              int8_t (*pixb)[pwk_tile_px(KS) * KS * 64], int8_t (*hdrb)[1024]) {
  constexpr int TP = pwk_tile_px(KS);                    // pixels of a tile: 128 (K <= 256), 64 (K = 512)
  constexpr int WN = 4 / WM, CT = TP / 32 / WN;          // a wave's 32-pixel column tiles of a tile
  constexpr int NWIN = DUAL ? 2 : 1, NV = KS * NWIN;      // K steps of a column tile: (window, slab)
  constexpr int NJ = 2;                                  // column tiles (accumulator sets) per wave at a time
  constexpr int TILE = KS * TP * 64;                     // bytes of one tile: [slab][pixel][64]
  constexpr int NG = TP / 16, UPW = KS * NG / 4;         // 16-pixel groups of a tile; DMA units (slab, group) per wave
  static_assert(WM * WN == 4 && CT >= NJ && CT % NJ == 0 && TILE <= 32 * 1024 && (KS * NG) % 4 == 0, "wave grid / tile");

  const ConvGeom g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5, frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);        // a lane's fragment of pixel frow of a [32][64] tile, K half 0
  const int part = bid / n_streams, stream = bid - part * n_streams;
  const int ch = (part * WM + wm) * 32;                  // first of the wave's 32 output channels
  const int tms = tm == 128 ? 7 : 6;
  int* const prm = reinterpret_cast<int*>(hdrb[wave]);

  // ---- a tile's pixels -> LDS: units u = wave, wave + 4, ..: (slab, 16-pixel group); every wave issues 2 KS DMAs per tile ------------
  const bool contiguous = g.stride == 1 && g.OHW == g.H * g.W;          // output pixel index == input pixel index
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
  auto issue_tile = [&](int t, int8_t* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < UPW; q++) {
      const int u = wave + 4 * q, s = u / NG, grp = u % NG;
      const int pg = t * TP + grp * 16;
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
        off = (unsigned)(s * 64 + chunk * 16);             // (a group straddling the end reads the layer's first pixel instead: never stored)
      }
      // (a group wholly past the launch: the zero page -- a different base, wave-uniform)
      if (pg >= g.n_pix) { base = a.zero; off = (unsigned)(chunk * 16); }
      pwk_dma16(base, off, buf + s * (TP * 64) + grp * 1024);
    }
  };
  int t = stream;
  if (t >= n_tiles) return;
  long long* const dbg = a.dbg2 ? a.dbg2 + (size_t)blockIdx.x * 16 : nullptr;       // tools/pwk_timeline.py: 100 MHz wall clock per phase
#define PWK_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  PWK_STAMP(0);
  issue_tile(t, pixb[0]);

  // the wave's weight fragments: K step v = (window, slab), 16 contiguous bytes of the lane's row per K half
  i32x4 fa[NV][2];
  {
    const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
    const int8_t* const wrow = a.w + (size_t)(mt * KS) * (NWIN << tms) * 64 + (size_t)ro * 64 + (size_t)(frow * 64 + half * 16);
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int win = v / KS, sl = v - win * KS;
      const int8_t* p = wrow + ((size_t)(sl * NWIN + win) << tms) * 64;
      fa[v][0] = *reinterpret_cast<const i32x4*>(p); fa[v][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
    // its 32 header rows as an m-tile image of 32 rows in its LDS slot: rows {bias | dbl, alpha, addend64} | lo | dshift[P].
    // rows: 32 x 16 bytes = lanes 0..31; lo: 32 words = lanes 32..39 (16 bytes each); dshift[p]: lanes 40..47, 48..55
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(a.hdr) + (size_t)mt * a.hdr_bytes;
    const int TMr = 1 << tms;
    int hdst = -1;
    const int8_t* src = hsrc;
    if (lane < 32) { src = hsrc + (size_t)(ro + lane) * 16; hdst = lane * 16; }
    else if (lane < 40) { src = hsrc + (size_t)(4 * TMr + ro) * 4 + (lane - 32) * 16; hdst = 32 * 16 + (lane - 32) * 16; }
    else if (lane < 40 + 8 * NWIN) {
      const int pz = (lane - 40) >> 3, k = (lane - 40) & 7;
      src = hsrc + (size_t)((5 + pz) * TMr + ro) * 4 + k * 16; hdst = (5 + pz) * 32 * 4 + k * 16;
    }
    const i32x4 hv = *reinterpret_cast<const i32x4*>(src);
    if (hdst >= 0) *reinterpret_cast<i32x4*>(hdrb[wave] + hdst) = hv;
  }

  // (the epilogue's form -- residual, FAST rows -- is chosen ONCE, outside the loops: four forms inside the loop body kept their common
  //  lane values alive across all of them, in scratch -- conv_bneck's lesson)
  auto run = [&](auto has_res_c, auto fast_c) __attribute__((always_inline)) {
    constexpr bool HAS_RES = decltype(has_res_c)::value;
    constexpr bool FAST = decltype(fast_c)::value;
    const int lo_bound = g.relu ? 0 : -128, rlo = g.add_relu ? 0 : -128;
    const unsigned reso = (unsigned)(frow * g.res_cp + 16 * half);
    const unsigned yo = (unsigned)(frow * g.y_cp + 16 * half);
    const int chl = ch + 16 * half;
    const bool ch_ok = chl + 16 <= g.y_nvalid;
    int it = 0;
#pragma unroll 1
    for (; t < n_tiles; t += n_streams, it++) {
      // tile t landed in every wave (and every store / load this wave issued before); nobody reads the other buffer any more
      if (it < 3) PWK_STAMP(1 + 3 * it);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (it < 3) PWK_STAMP(2 + 3 * it);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (it < 3) PWK_STAMP(3 + 3 * it);
      if (t + n_streams < n_tiles) issue_tile(t + n_streams, pixb[(it + 1) & 1]);
      const int8_t* const B0 = pixb[it & 1];
      const unsigned res_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(t * TP) * (unsigned)g.res_cp + (unsigned)g.res_off + (unsigned)ch));
      const unsigned y_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(t * TP) * (unsigned)g.y_cp + (unsigned)g.y_off + (unsigned)ch));
#pragma unroll
      for (int k0 = 0; k0 < CT; k0 += NJ) {                // this wave's column tiles wn * CT + k0 .. + NJ
        // the group's residual tiles (16 contiguous NHWC bytes per lane and column tile) land during its MFMAs
        i32x4 rvs[NJ];
        bool okj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int tj = wn * CT + k0 + j;
          const int p = t * TP + tj * 32 + frow;
          okj[j] = p < g.n_pix && ch_ok;
          const int8_t* rp = (HAS_RES && okj[j]) ? a.res + (res_u + reso + (unsigned)(tj * 32 * g.res_cp)) : a.zero;
          rvs[j] = *reinterpret_cast<const i32x4*>(rp);
        }
        i32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[j][r] = 0;
        const int8_t* const Bw = B0 + (wn * CT + k0) * 2048;
        // B fragments one K step ahead of their MFMAs, no further (sched_barrier: left alone, the scheduler hoists all NV * 2 * NJ reads
        // -- up to 128 registers -- in front of the first MFMA)
        i32x4 bfc[2][NJ], bfn[2][NJ];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int j = 0; j < NJ; j++) bfc[ks][j] = *reinterpret_cast<const i32x4*>(Bw + j * 2048 + (fr0 ^ (ks << 5)));
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (v + 1 < NV) {
            const int sn = (v + 1) % KS;
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
              for (int j = 0; j < NJ; j++) bfn[ks][j] = *reinterpret_cast<const i32x4*>(Bw + sn * (TP * 64) + j * 2048 + (fr0 ^ (ks << 5)));
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
            for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[v][ks], bfc[ks][j], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < NJ; j++) bfc[ks][j] = bfn[ks][j];
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
          const int tj = wn * CT + k0 + j;
          if (okj[j]) *reinterpret_cast<i32x4*>(a.y + (y_u + yo + (unsigned)(tj * 32 * g.y_cp))) = outs[j];
        }
      }
    }
  };
  // ---- FAST rows without a residual: a column group's requantisation rides between the NEXT group's MFMAs (one wave keeps the matrix pipe
  // and the VALU busy at once: two waves per SIMD that start together would otherwise contend for the one, then for the other) ------------
  auto run_pipe = [&](auto dbl_c) __attribute__((always_inline)) {
    constexpr bool DBL = decltype(dbl_c)::value;
    const int lo_bound = g.relu ? 0 : -128;
    const unsigned yo = (unsigned)(frow * g.y_cp + 16 * half);
    const bool ch_ok = ch + 16 * half + 16 <= g.y_nvalid;
    const rq_i32x4* const rowp = reinterpret_cast<const rq_i32x4*>(prm) + 4 * half;
    const rq_i32x4 nolo = {0, 0, 0, 0};
    int pend[NJ][16];
    unsigned pd[NJ][4];
    unsigned pend_off[NJ];
    bool pend_ok[NJ];
    bool has_pend = false;
#pragma unroll
    for (int j = 0; j < NJ; j++) { pend_off[j] = 0; pend_ok[j] = false; }
    // rows 4 G .. 4 G + 3 (of the lane's sixteen) of the pending group -> their four packed bytes
    auto epi_rows = [&](int G) __attribute__((always_inline)) {
      rq_i32x4 pr4[4];
#pragma unroll
      for (int r = 0; r < 4; r++) pr4[r] = rowp[8 * G + r];
#pragma unroll
      for (int j = 0; j < NJ; j++) pd[j][G] = rq_rows4<true, DBL, false>(pend[j], G, pr4, nolo, lo_bound);
    };
    auto epi_store = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        auto s02 = __builtin_amdgcn_permlane32_swap(pd[j][0], pd[j][2], false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(pd[j][1], pd[j][3], false, false);
        const i32x4 out = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
        if (pend_ok[j]) *reinterpret_cast<i32x4*>(a.y + pend_off[j]) = out;
      }
    };
    int it = 0;
#pragma unroll 1
    for (; t < n_tiles; t += n_streams, it++) {
      if (it < 3) PWK_STAMP(1 + 3 * it);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (it < 3) PWK_STAMP(2 + 3 * it);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (it < 3) PWK_STAMP(3 + 3 * it);
      if (t + n_streams < n_tiles) issue_tile(t + n_streams, pixb[(it + 1) & 1]);
      const int8_t* const B0 = pixb[it & 1];
      const unsigned y_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(t * TP) * (unsigned)g.y_cp + (unsigned)g.y_off + (unsigned)ch));
#pragma unroll
      for (int k0 = 0; k0 < CT; k0 += NJ) {
        i32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[j][r] = 0;
        const int8_t* const Bw = B0 + (wn * CT + k0) * 2048;
        i32x4 bfc[2][NJ], bfn[2][NJ];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
          for (int j = 0; j < NJ; j++) bfc[ks][j] = *reinterpret_cast<const i32x4*>(Bw + j * 2048 + (fr0 ^ (ks << 5)));
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (v + 1 < NV) {
            const int sn = (v + 1) % KS;
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
#pragma unroll
              for (int j = 0; j < NJ; j++) bfn[ks][j] = *reinterpret_cast<const i32x4*>(Bw + sn * (TP * 64) + j * 2048 + (fr0 ^ (ks << 5)));
          }
          if (DUAL && v == KS) {
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
            for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[v][ks], bfc[ks][j], acc[j], 0, 0, 0);
          // the pending group's rows behind this step's MFMAs: 4 row groups over NV steps
          if (has_pend) {
            if constexpr (NV >= 4) { if (v % (NV / 4) == 0) epi_rows(v / (NV / 4)); }
            else { epi_rows(2 * v); epi_rows(2 * v + 1); }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int j = 0; j < NJ; j++) bfc[ks][j] = bfn[ks][j];
        }
        if (has_pend) epi_store();
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int tj = wn * CT + k0 + j;
#pragma unroll
          for (int r = 0; r < 16; r++) pend[j][r] = acc[j][r];
          pend_ok[j] = t * TP + tj * 32 + frow < g.n_pix && ch_ok;
          pend_off[j] = y_u + yo + (unsigned)(tj * 32 * g.y_cp);
        }
        has_pend = true;
      }
    }
    if (has_pend) {
#pragma unroll
      for (int G = 0; G < 4; G++) epi_rows(G);
      epi_store();
    }
  };
  bool piped = false;
  if constexpr (NV * 8 <= 128) {
    if (g.fast == 1 && !g.has_res && pipe) { piped = true; if (g.dbl_out) run_pipe(std::true_type{}); else run_pipe(std::false_type{}); }
  }
  if (!piped) {
  if (g.fast == 1) { if (g.has_res) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}); }
  else { if (g.has_res) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}); }
  }
  if (dbg) {
    PWK_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PWK_STAMP(11);
    if (tid == 0) dbg[12] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  }
#undef PWK_STAMP
}

template <int KS, int WM, bool DUAL>
__global__ __launch_bounds__(256, KS * (DUAL ? 2 : 1) > 8 ? 1 : 2) void conv_pwk_kernel(ConvArgs a, int n_tiles, int tm, int n_streams, int pipe) {
  __shared__ __attribute__((aligned(1024))) int8_t pixb[2][pwk_tile_px(KS) * KS * 64];
  __shared__ __attribute__((aligned(16))) int8_t hdrb[4][kPwkHdrSlot];
  conv_pwk_body<KS, WM, DUAL>(a, n_tiles, tm, n_streams, pipe, (int)blockIdx.x, pixb, hdrb);
}

// two independent rows of one instantiation in one launch (conv_mfma2_pair_kernel's scheme: a stage's shortcut convolution beside the first
// 1x1 of its first bottleneck -- same input, no dependence): blocks [0, n0) work on the first argument block, the rest on the second
template <int KS, int WM, bool DUAL>
__global__ __launch_bounds__(256, KS * (DUAL ? 2 : 1) > 8 ? 1 : 2) void conv_pwk_pair_kernel(ConvArgs a0, ConvArgs a1, int n_tiles0, int tm0, int n_streams0, int n_tiles1, int tm1, int n_streams1,
                                                               int pipe, int n0) {
  __shared__ __attribute__((aligned(1024))) int8_t pixb[2][pwk_tile_px(KS) * KS * 64];
  __shared__ __attribute__((aligned(16))) int8_t hdrb[4][kPwkHdrSlot];
  if ((int)blockIdx.x < n0) conv_pwk_body<KS, WM, DUAL>(a0, n_tiles0, tm0, n_streams0, pipe, (int)blockIdx.x, pixb, hdrb);
  else conv_pwk_body<KS, WM, DUAL>(a1, n_tiles1, tm1, n_streams1, pipe, (int)blockIdx.x - n0, pixb, hdrb);
}

// Does the layer qualify?  1x1 / pad 0 / stride 1 or 2 (any dilation of a 1x1 is the plain layer), dense entries (every m-tile holds
// slabs 0 .. nslab - 1 in order, one window or dual) in the layer's OWN tiles (no shared storage), K = 2 or 4 slabs, output channels a
// multiple of the 32 x WM a block covers, no fused global average.
static int g_pwk_min_units = 512;                        // pwk_units: fewest (tile, channel part) units of a row
void conv_pwk_set_min_units(int u) { g_pwk_min_units = u; }
static int pwk_wm(int Np) { return Np % 128 == 0 ? 4 : Np % 64 == 0 ? 2 : 0; }
bool conv_pwk_eligible(const ConvArgs& a, int TM, int k, int dense, long min_pix, bool force) {
  const ConvGeom& g = a.g;
  if (k != 1 || (g.pad_h | g.pad_w) != 0 || (g.stride != 1 && g.stride != 2) || g.avg_mult) return false;
  if (!dense || (TM != 64 && TM != 128) || (a.nslab != 2 && a.nslab != 4 && a.nslab != 8) || g.Cp_in != a.nslab * 64) return false;
  if (a.n_phases > 2 || (a.n_phases == 2 && !a.dual)) return false;
  const int wins = a.dual ? 2 : 1;
  if (a.w_sub_step || a.e_mt_shr || a.e_mt_shl || a.w_ent_bytes != wins * TM * 64 || a.w_win_stride != TM * 64) return false;
  const int wm = pwk_wm(a.Np);
  if (!wm || a.Np % TM != 0) return false;
  if (a.nslab == 8 && wm != 4) return false;              // (K = 512: 64-pixel tiles, four channel groups x one pixel group)
  if ((long long)g.n_pix * g.y_cp >= (1ll << 32) || (long long)g.H * g.W * (g.n_pix / std::max(1, g.OHW)) * g.Cp_in >= (1ll << 32)) return false;
  if (g.has_res && (long long)g.n_pix * g.res_cp >= (1ll << 32)) return false;
  // enough (tile, channel part) units that a block walks several tiles: the kernel's gain is the amortised first-operand wait (one tile per block:
  // GoogLeNet's two 196-tile rows ran 1 % slower than on the ring kernel, ResNet-50's row 27 -- 49 tiles x 8 parts -- 10.3 against 8.4 us)
  const int tp = pwk_tile_px(a.nslab);
  if (force) return true;                                  // (pwk_rows: test-only)
  if ((long)((g.n_pix + tp - 1) / tp) * (a.Np / (32 * wm)) < g_pwk_min_units) return false;
  return g.n_pix >= min_pix;
}

static int g_pwk_slots = 256;                            // pwk_slots (test-only): blocks a launch aims at (one per CU: 3-5 tiles per block amortise a block's first-operand wait; 512 measured 0.7 % slower in flight)
static int g_pwk_pipe = 1;                               // pwk_pipe (test-only): 0 = no requantisation between the next group's MFMAs
void conv_pwk_set_slots(int t) { g_pwk_slots = t > 0 ? t : 256; }
void conv_pwk_set_pipe(int p) { g_pwk_pipe = p; }

// tile streams of a layer: every block walks the same number of tiles (+- 1), ~`slots` blocks in all; a multiple of 8 where channel parts share
// tiles (block ids n_streams apart then sit on one XCD)
static int pwk_streams(const ConvArgs& a, int WM, int slots, int* n_tiles_out, int* parts_out) {
  const int tp = pwk_tile_px(a.nslab);
  const int n_tiles = (a.g.n_pix + tp - 1) / tp;
  const int parts = a.Np / (32 * WM);                     // channel parts: blocks that share a pixel tile (its activations come from L2 then)
  const int want = std::max(1, slots / parts);
  const int per = (n_tiles + want - 1) / want;
  int n_streams = (n_tiles + per - 1) / per;
  if (parts > 1 && n_streams >= 8) n_streams = std::min((n_streams + 7) & ~7, n_tiles);
  *n_tiles_out = n_tiles; *parts_out = parts;
  return n_streams;
}

template <int KS, int WM, bool DUAL>
static int launch_pwk2(const ConvArgs& a, int TM, hipStream_t s) {
  auto fn = conv_pwk_kernel<KS, WM, DUAL>;
  int n_tiles, parts;
  const int n_streams = pwk_streams(a, WM, g_pwk_slots, &n_tiles, &parts);
  const int grid = n_streams * parts;
  TF2_LAUNCH_NAME("conv_pwk_kernel<%d slabs,%d channel groups,%s> (%d streams of %d..%d tiles x %d channel parts)", KS, WM, DUAL ? "dual" : "single",
                  n_streams, n_tiles / n_streams, (n_tiles + n_streams - 1) / n_streams, parts);
  TF2_LAUNCH(fn, dim3(grid), dim3(256), 0, s, a, n_tiles, TM, n_streams, g_pwk_pipe);
  return launch_ok() ? 0 : -1;
}

template <int KS, int WM, bool DUAL>
static int launch_pwk_pair2(const ConvArgs& a0, int TM0, const ConvArgs& a1, int TM1, hipStream_t s) {
  auto fn = conv_pwk_pair_kernel<KS, WM, DUAL>;
  int nt0, p0, nt1, p1;
  // the launch aims at the same number of blocks as one row would: each row at half
  const int ns0 = pwk_streams(a0, WM, g_pwk_slots / 2, &nt0, &p0), ns1 = pwk_streams(a1, WM, g_pwk_slots / 2, &nt1, &p1);
  const int n0 = ns0 * p0, grid = n0 + ns1 * p1;
  TF2_LAUNCH_NAME("conv_pwk_pair_kernel<%d slabs,%d channel groups,%s> (%d x %d + %d x %d blocks)", KS, WM, DUAL ? "dual" : "single", ns0, p0, ns1, p1);
  TF2_LAUNCH(fn, dim3(grid), dim3(256), 0, s, a0, a1, nt0, TM0, ns0, nt1, TM1, ns1, g_pwk_pipe, n0);
  return launch_ok() ? 0 : -1;
}

int launch_conv_pwk(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int wm = pwk_wm(a.Np);
  const bool dual = a.dual != 0;
#define TF2_PWK(KS_, WM_) do { return dual ? launch_pwk2<KS_, WM_, true>(a, TM, s) : launch_pwk2<KS_, WM_, false>(a, TM, s); } while (0)
#define TF2_PWK_KS(KS_) do { if (wm == 4) TF2_PWK(KS_, 4); if (wm == 2) TF2_PWK(KS_, 2); } while (0)
  if (a.nslab == 8 && wm == 4) TF2_PWK(8, 4);
  if (a.nslab == 4) TF2_PWK_KS(4);
  if (a.nslab == 2) TF2_PWK_KS(2);
#undef TF2_PWK_KS
#undef TF2_PWK
  return 1;
}

// two rows in one launch: the same instantiation (K slabs, channel groups per block, windows)
bool conv_pwk_pair_eligible(const ConvArgs& a0, const ConvArgs& a1) {
  return a0.nslab == a1.nslab && pwk_wm(a0.Np) == pwk_wm(a1.Np) && (a0.dual != 0) == (a1.dual != 0) && !a0.dbg2 && !a1.dbg2;
}

int launch_conv_pwk_pair(const ConvArgs& a0, int TM0, const ConvArgs& a1, int TM1, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!conv_pwk_pair_eligible(a0, a1)) return 1;
  const int wm = pwk_wm(a0.Np);
  const bool dual = a0.dual != 0;
#define TF2_PWKP(KS_, WM_) do { return dual ? launch_pwk_pair2<KS_, WM_, true>(a0, TM0, a1, TM1, s) : launch_pwk_pair2<KS_, WM_, false>(a0, TM0, a1, TM1, s); } while (0)
#define TF2_PWKP_KS(KS_) do { if (wm == 4) TF2_PWKP(KS_, 4); if (wm == 2) TF2_PWKP(KS_, 2); } while (0)
  if (a0.nslab == 8 && wm == 4) TF2_PWKP(8, 4);
  if (a0.nslab == 4) TF2_PWKP_KS(4);
  if (a0.nslab == 2) TF2_PWKP_KS(2);
#undef TF2_PWKP_KS
#undef TF2_PWKP
  return 1;
}

}  // namespace tf2
