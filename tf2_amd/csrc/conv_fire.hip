// conv_fire.hip -- a whole fire module (squeeze 1x1 C -> S, then expand1x1 | expand3x3 S -> E + E written to adjacent slices of one
// concat tensor; pe.cl:144-203 per row, feature_writer.cl:109,116 for the slices, kNStart / kBranchTail quantization.cpp:42-49) in ONE
// launch (gfx950, round 5).
//
// After the merged rows of weight_pack.cpp (PackLayer::merge_next: the two expand rows are ONE 3x3 layer of 2E output channels whose first
// E rows carry the 1x1 filters as centre taps) a fire module is two launches: the squeeze, 5-7 us for a few MFLOP, and the merged expand.
// This kernel is conv_bband.hip's first two phases for that pair: a block owns R output rows x the full width of ONE image; it computes
// the squeeze for its (R + 2) halo rows from the module's input band (which goes global -> LDS once, in one piece: at most 64 KB),
// requantises straight into the expand's halo tile in LDS (16-byte PLANES per pixel and 16-channel group: a tap is a shifted address),
// and runs the merged expand from that tile, weights global -> registers, following the packed layer's own per-m-tile slab lists (the
// 1x1 rows' m-tiles hold the centre-tap slabs only).  No exchange between blocks, no counted wait (every DMA is drained with vmcnt(0) before
// the one barrier that follows it).
//
// The merged expand's K order is (tap, channel) in 64-byte slabs with Sp = round_up(S, 16) bytes per tap: segment g = 4 slab + 2 ks + half
// (16 bytes, one MFMA operand piece of a lane) is tap g / (Sp / 16), plane g % (Sp / 16); segments past tap 8 are K padding (zero weights).
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

__device__ __forceinline__ void fire_dma16(const int8_t* src, int8_t* lds_dst) {
  const unsigned l = (unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(l) : "memory", "m0");
}

// MT0: 32-row tiles of the squeeze (S <= 32: 1, else 2); DUAL1: the squeeze is a two-window layer; JW: pixel tiles per wave in the expand.
// POOL: the module's 3x3 / stride 2 / pad 0 ceil-mode max pool (pool.cl:152-260 + pool_tail.cl:91-216; SqueezeNet 1.1's fire3 / fire5) in the
// launch: a block owns PR pooled rows, i.e. the R = 2 PR + 1 expand rows under them (one row recomputed per neighbour) and their R + 2 squeeze
// rows; the requantised expand tile goes to LDS (16-byte chunk c of pixel p at chunk c ^ (p mod chunks): conflict-free for the sixteen lanes of
// a ds_write_b128) and the pool runs from there -- every byte is 0..127 behind the expands' ReLU (launcher-checked), so the byte-wise maximum is
// v_pk_max_u16 on the even / odd bytes and window slots outside the map (zeros, pool.cl:119-140) are skipped.
template <int MT0, bool DUAL1, int JW, bool POOL = false>
__global__ __launch_bounds__(512, 2) void conv_fire_kernel(FireArgs a) {
  extern __shared__ __attribute__((aligned(1024))) int8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int W = a.W, Wp = W + 2, H = a.H, R = a.R;
  const int KS1 = a.Cin >> 6;                              // K slabs of the squeeze
  const int NP0 = a.NT0 * 32;                              // halo-band pixels, padded to whole column tiles
  const int n_h = (R + 2) * Wp;
  const int planeb = ((n_h + 63) >> 6) * 1024;             // bytes of one 16-channel plane of the halo tile
  const int NPL = a.Sp >> 4;                               // planes
  const int tms1 = a.tm1 == 128 ? 7 : 6, tms2 = a.tm2 == 128 ? 7 : 6;
  int8_t* const xin = lds;                                 // [KS1][NP0][64] swizzled (conv_bband.hip's chunk layout)
  int8_t* const mid = xin + (size_t)KS1 * NP0 * 64;        // [NPL][planeb]
  int8_t* const hdr1 = mid + (size_t)NPL * planeb;
  const int hst1 = (DUAL1 ? 28 : 20) << tms1;              // bytes of the squeeze's (single) m-tile header: rows | lo | dshift[P]
  int8_t* const hdr2 = hdr1 + hst1;
  const int hst2 = 20 << tms2;
  int8_t* const cy = hdr2 + (size_t)(a.N2 >> tms2) * hst2;      // POOL: the expand tile [R * W pixels][N2 bytes], chunk-swizzled

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;      // the bands of one image on one XCD (conv_bband.hip)
  }
  const int img = bid / a.tiles_per_img;
  const int r0 = (bid - img * a.tiles_per_img) * (POOL ? 2 * a.PR : R);       // first expand row of the band
  const int rows = (H - r0) < R ? (H - r0) : R;
  const int n_px = rows * W;
  const int n_p0 = (R + 2) * W;
  const long long pix_base = ((long long)img * H + r0) * W;
  const long long pix0 = pix_base - W;
  long long* const dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/fire_timeline.py: 100 MHz wall clock per phase
#define FIRE_STAMP(i) do { if (dbg && (tid & 63) == 0) dbg[(i) + (wave == 7 ? 8 : 0)] = (long long)wall_clock64(); } while (0)
  if (wave == 0 || wave == 7) FIRE_STAMP(0);

  // ---- prologue: pad fill of the halo tile, the input band, the headers ----
  for (int gi = wave; gi < NPL * (planeb >> 10); gi += 8) {
    const int k = gi / (planeb >> 10), grp = gi - k * (planeb >> 10);
    fire_dma16(a.zero2 + k * 16, mid + k * planeb + grp * 1024);
  }
  {
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
    for (int gi = wave; gi < KS1 * (NP0 / 16); gi += 8) {
      const int sl = gi / (NP0 / 16), grp = gi - sl * (NP0 / 16);
      const int p = grp * 16 + drow;
      const int row = r0 - 1 + fast_div(p, a.w_m, a.w_s);
      const bool ok = p < n_p0 && (unsigned)row < (unsigned)H;
      const int8_t* src = ok ? a.x + (size_t)(pix0 + p) * a.Cin + sl * 64 + chunk * 16 : a.zero + chunk * 16;
      fire_dma16(src, xin + (size_t)sl * (NP0 * 64) + grp * 1024);
    }
  }
  for (int i = tid; i < (hst1 >> 4); i += 512) reinterpret_cast<i32x4*>(hdr1)[i] = reinterpret_cast<const i32x4*>(a.hdr1)[i];
  {
    const int per = hst2 >> 4, n_mt = a.N2 >> tms2;
    for (int i = tid; i < n_mt * per; i += 512) {
      const int mt = i / per, k = i - mt * per;
      reinterpret_cast<i32x4*>(hdr2)[i] = *reinterpret_cast<const i32x4*>(reinterpret_cast<const int8_t*>(a.hdr2) + (size_t)mt * a.hdr2_bytes + k * 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (wave == 0 || wave == 7) FIRE_STAMP(1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wave == 0 || wave == 7) FIRE_STAMP(2);

  // ---- phase 0: the squeeze over the halo band; wave = (row tile wave % MT0, pixel tiles wave / MT0 + j * (8 / MT0)) ----
  {
    constexpr int WN0 = 8 / MT0;
    int rot0 = bid & 7;
    while (rot0 >= KS1) rot0 -= KS1;
    const int rt = wave % MT0, wn = wave / MT0;
    const unsigned a_lane_off = (unsigned)((lane & 31) * 64 + half * 16);
    const int wins = DUAL1 ? 2 : 1;
    const int lo_b = a.relu1 ? 0 : -128;
    const int* prm = reinterpret_cast<const int*>(hdr1);
    for (int t = wn; t < a.NT0; t += WN0) {
      const int prow = t * 32 + (lane & 31);
      const int bm = prow * 64 + ((half ^ ((prow >> 2) & 3)) << 4);
      i32x16 acc, acc2;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[r] = 0; acc2[r] = 0; }
      // weight fragments PD0 slabs ahead of their MFMAs in rotating register buffers (global -> registers: an L2 round trip per slab
      // otherwise, 4-8 slabs on the 14 x 14 maps); loads unconditional, indices clamped (see phase 1)
      constexpr int PD0 = DUAL1 ? 2 : 4, NWF = DUAL1 ? 4 : 2;      // (two-window squeeze: 16 registers per stage -- three stages would cost the second block per CU)
      const int8_t* pu0 = a.w1 + (size_t)(rt * 32) * 64 + a_lane_off;
      const size_t wstep = ((size_t)wins << tms1) * 64, lo_off = (size_t)64 << tms1;
      i32x4 wr[PD0][NWF];
      // Every block walks the slabs from its OWN starting slab (Z/2^32 sums commute: pe.cl:43): the blocks of a launch start together, and
      // in lockstep each step's fragment was a miss for ALL of them at once (tools/fire_timeline.py: 4.6 us for eight slabs) -- rotated, the
      // slabs are first touched in parallel by different blocks and everybody else's later steps hit
      auto rot_s = [&](int sidx) { int k = (sidx < KS1 - 1 ? sidx : KS1 - 1) + rot0; return k >= KS1 ? k - KS1 : k; };
      auto load_w = [&](i32x4 (&f)[NWF], int sidx) __attribute__((always_inline)) {
        const int8_t* pu = pu0 + (size_t)rot_s(sidx) * wstep;
        f[0] = *reinterpret_cast<const i32x4*>(pu); f[1] = *reinterpret_cast<const i32x4*>(pu + 32);
        if constexpr (DUAL1) { f[2] = *reinterpret_cast<const i32x4*>(pu + lo_off); f[3] = *reinterpret_cast<const i32x4*>(pu + lo_off + 32); }
      };
      auto step0 = [&](int sidx, const i32x4 (&f)[NWF]) __attribute__((always_inline)) {
        const int8_t* B = xin + (size_t)rot_s(sidx) * (NP0 * 64);
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + bm), b1 = *reinterpret_cast<const i32x4*>(B + (bm ^ 32));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[1], b1, acc, 0, 0, 0);
        if constexpr (DUAL1) {
          acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[2], b0, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[3], b1, acc2, 0, 0, 0);
        }
      };
#pragma unroll
      for (int d = 0; d < PD0; d++) load_w(wr[d], d);
      int sb = 0;
      for (; sb + PD0 <= KS1; sb += PD0) {
#pragma unroll
        for (int d = 0; d < PD0; d++) {
          i32x4 cur[NWF];
#pragma unroll
          for (int q = 0; q < NWF; q++) cur[q] = wr[d][q];
          load_w(wr[d], sb + d + PD0);
          step0(sb + d, cur);
        }
      }
#pragma unroll
      for (int d = 0; d < PD0 - 1; d++)
        if (sb + d < KS1) step0(sb + d, wr[d]);
      int a16[16];
      if constexpr (DUAL1) {
        const int* dsh = prm + (6 << tms1) + rt * 32 + 4 * half;          // dshift[1] behind rows | lo | dshift[0]
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
      const rq_i32x4 nores = {0, 0, 0, 0};
      i32x4 out;
      if (a.fast1 == 1) out = requant_tile16<false, 0, true>(a16, prm, 1 << tms1, rt * 32 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, false);
      else out = requant_tile16<false, 0, false>(a16, prm, 1 << tms1, rt * 32 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, a.fast1 == 2);
      const int chl = rt * 32 + 16 * half;
      const int hr = fast_div(prow, a.w_m, a.w_s), col = prow - hr * W;
      const int row = r0 - 1 + hr;
      if (chl < a.Sp && prow < n_p0 && (unsigned)row < (unsigned)H) {
        *reinterpret_cast<i32x4*>(mid + (chl >> 4) * planeb + (hr * Wp + col + 1) * 16) = out;
        if (a.keep_mid && hr >= 1 && hr <= rows) *reinterpret_cast<i32x4*>(a.mid + (size_t)(pix0 + prow) * a.mid_cp + chl) = out;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (wave == 0 || wave == 7) FIRE_STAMP(3);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wave == 0 || wave == 7) FIRE_STAMP(4);

  // ---- phase 1: the merged expand (3x3 / pad 1, N2 output channels) from the halo tile; wave = (row tiles wm + i * WM, pixel tiles wn + j * WN)
  {
    const int RT = a.N2 >> 5;                              // 32-row tiles
    const int WM = a.WM, WN = 8 / WM;
    const int wm = wave % WM, wn = wave / WM;
    int h0[JW];                                            // byte offset of the lane's pixel (tap (0, 0)) inside a plane
#pragma unroll
    for (int j = 0; j < JW; j++) {
      int p = (wn + j * WN) * 32 + (lane & 31);
      if (p >= n_px) p = 0;                                // lanes beyond the band compute on pixel 0 and are never stored
      const int r = fast_div(p, a.w_m, a.w_s);
      h0[j] = (r * Wp + (p - r * W)) * 16;
    }
    const int lo_b2 = a.relu2 ? 0 : -128;
    const int gpt = NPL;                                   // 16-byte segments per tap
    for (int rt = wm; rt < RT; rt += WM) {
      const int ch = rt * 32;
      const int mt = ch >> tms2, ro = ch & ((1 << tms2) - 1);
      const int e0 = a.dir2[mt * 2], e1 = a.dir2[mt * 2 + 1];
      i32x16 acc[JW];
#pragma unroll
      for (int j = 0; j < JW; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0;
      // the entry list and the weight fragments PD entries ahead of their MFMAs, in a rotating set of register buffers; the list is walked
      // from the block's own starting entry (see phase 0).  Every load is unconditional -- entry indices are clamped to the list's last
      // entry -- because hipcc drains the queue, vmcnt(0), at any join behind a conditional global load.
      // (Measured, tools/fire_timeline.py, 14 x 14 / C 512: no ring 20 us per launch, one entry ahead 16 us, two or four: the same; the
      //  first fill hoisted to the kernel's start: the same -- the chain moves into the prologue's wait.)
      constexpr int PD = 2;
      const int8_t* pw = a.w2 + (size_t)ro * 64 + (lane & 31) * 64 + half * 16;
      i32x4 ra0[PD], ra1[PD]; int rsl[PD];
      const int n_e = e1 > e0 ? e1 - e0 : 1;
      int rot = bid & 15;
      while (rot >= n_e) rot -= n_e;
      auto ent = [&](int i) { int k = (i < n_e - 1 ? i : n_e - 1) + rot; return e0 + (k >= n_e ? k - n_e : k); };
      auto ring_load = [&](int d, int i) __attribute__((always_inline)) {
        // (an m-tile whose rows are all zero has an EMPTY list, e1 == e0: n_e is forced to 1 and the unconditional prefetch would read the
        //  next m-tile's first entry -- or, for the last m-tile, one entry past the layer's arrays; the values are never used, the address is
        //  clamped to the layer's last entry all the same)
        const int en = ent(i) < a.n_ent2 ? ent(i) : a.n_ent2 - 1;
        const int8_t* pu = pw + (((size_t)en << tms2) * 64);
        rsl[d] = a.ent2[en]; ra0[d] = *reinterpret_cast<const i32x4*>(pu); ra1[d] = *reinterpret_cast<const i32x4*>(pu + 32);
      };
#pragma unroll
      for (int d = 0; d < PD; d++) ring_load(d, d);
      auto step = [&](int sl, const i32x4& a0, const i32x4& a1) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          // this lane's 16-byte K segment: g = 4 sl + 2 ks + half -> (tap, plane); K padding behind tap 8 has zero weights: any address
          const int g = 4 * sl + 2 * ks + half;
          int t = fast_div(g, a.g_m, a.g_s);
          const int pln = g - t * gpt;
          t = t > 8 ? 8 : t;
          const int th = (t * 11) >> 5;                    // t / 3 for t <= 8
          const int off = pln * planeb + (th * Wp + (t - 3 * th)) * 16;
          i32x4 bf[JW];
#pragma unroll
          for (int j = 0; j < JW; j++) bf[j] = *reinterpret_cast<const i32x4*>(mid + h0[j] + off);
#pragma unroll
          for (int j = 0; j < JW; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ks ? a1 : a0, bf[j], acc[j], 0, 0, 0);
        }
      };
      int eb = e0;
      for (; eb + PD <= e1; eb += PD) {                    // whole groups: branch-free
#pragma unroll
        for (int d = 0; d < PD; d++) {
          const int sl = rsl[d];
          const i32x4 a0 = ra0[d], a1 = ra1[d];
          ring_load(d, eb - e0 + d + PD);
          step(sl, a0, a1);
        }
      }
#pragma unroll
      for (int d = 0; d < PD - 1; d++)                     // the list's last e1 - eb < PD entries: fetched already
        if (eb + d < e1) step(rsl[d], ra0[d], ra1[d]);
      if ((wave == 0 || wave == 7) && rt == wm) FIRE_STAMP(5);          // first row tile's MFMAs issued
      const int* prm = reinterpret_cast<const int*>(hdr2 + mt * hst2);
      const int chl = ch + 16 * half;
      const rq_i32x4 nores = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < JW; j++) {
        int a16[16];
#pragma unroll
        for (int r = 0; r < 16; r++) a16[r] = acc[j][r];
        i32x4 out;
        if (a.fast2 == 1) out = requant_tile16<false, 0, true>(a16, prm, 1 << tms2, ro + 4 * half, lo_b2, -128, nores, false, false);
        else out = requant_tile16<false, 0, false>(a16, prm, 1 << tms2, ro + 4 * half, lo_b2, -128, nores, false, a.fast2 == 2);
        const int p = (wn + j * WN) * 32 + (lane & 31);
        if constexpr (POOL) {
          const int nch = a.N2 >> 4;
          if (p < n_px) *reinterpret_cast<i32x4*>(cy + (size_t)p * a.N2 + ((((chl >> 4) ^ p) & (nch - 1)) << 4)) = out;
        }
        if ((!POOL || a.keep_mid) && p < n_px && chl + 16 <= a.y_nvalid)
          *reinterpret_cast<i32x4*>(a.y + (size_t)(pix_base + p) * a.y_cp + a.y_off + chl) = out;
      }
      if ((wave == 0 || wave == 7) && rt == wm) FIRE_STAMP(6);          // ... and requantised, stores issued
    }
  }
  if constexpr (POOL) {
    // ---- the pool over the block's own expand tile ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nch = a.N2 >> 4;
    const int ph0 = (bid - img * a.tiles_per_img) * a.PR;
    const int pr_n = (a.PH - ph0) < a.PR ? (a.PH - ph0) : a.PR;
    const int n_out = pr_n * a.PW * nch;
    for (int idx = tid; idx < n_out; idx += 512) {
      const int c = idx & (nch - 1), pix = idx / nch;
      const int pr = pix / a.PW, pw = pix - pr * a.PW;
      unsigned me[4] = {0, 0, 0, 0}, mo[4] = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const int r = pr * 2 + i;
        if (r >= rows) continue;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int col = pw * 2 + j;
          if (col >= W) continue;
          const int p = r * W + col;
          const i32x4 v = *reinterpret_cast<const i32x4*>(cy + (size_t)p * a.N2 + (((c ^ p) & (nch - 1)) << 4));
#pragma unroll
          for (int q = 0; q < 4; q++) {
            me[q] = pk_max_u16(me[q], (unsigned)v[q] & 0x00ff00ffu);
            mo[q] = pk_max_u16(mo[q], ((unsigned)v[q] >> 8) & 0x00ff00ffu);
          }
        }
      }
      if (c * 16 + 16 <= a.y_nvalid) {
        const i32x4 o = {(int)(me[0] | (mo[0] << 8)), (int)(me[1] | (mo[1] << 8)), (int)(me[2] | (mo[2] << 8)), (int)(me[3] | (mo[3] << 8))};
        *reinterpret_cast<i32x4*>(a.yp + (((size_t)img * a.PH + ph0 + pr) * a.PW + pw) * a.yp_cp + a.yp_off + c * 16) = o;
      }
    }
  }
  if (wave == 0 || wave == 7) FIRE_STAMP(7);
#undef FIRE_STAMP
}

// geometry of a fire launch: rows per band (halo band and band within the 8 / 4 pixel tiles the waves cover), LDS bytes
bool conv_fire_geometry(int H, int W, int Cin, int Sp, int N2, int tm1, int tm2, int dual1, int pool, FireArgs* f, size_t* lds_out) {
  if (H != W || (W != 56 && W != 28 && W != 14) || Cin % 64 != 0 || Cin < 64 || Cin > 512 || Sp % 16 != 0 || Sp < 16 || Sp > 64) return false;
  if (N2 % 128 != 0 || N2 < 128 || N2 > 512 || (tm1 != 64 && tm1 != 128) || (tm2 != 64 && tm2 != 128)) return false;
  if (pool && (W < 28 || (N2 != 128 && N2 != 256))) return false;      // (pooled form: 56 / 28 wide maps, 8 or 16 chunks per pixel)
  constexpr int kPR = 2;                                   // pooled rows per block
  const int R = pool ? 2 * kPR + 1 : (W == 56 ? 2 : 4);    // 4 x 56, 6 x 28, 6 x 14 halo pixels: 7 / 6 / 3 column tiles of 32 (bands of 4 / 4 / 2); pooled: 7 halo rows
  const int NT0 = ((R + 2) * W + 31) / 32, NT1 = (R * W + 31) / 32;
  const int MT0 = Sp > 32 ? 2 : 1;
  if (NT0 > 8 / MT0 * 2) return false;
  const int RT = N2 / 32;
  const int WM = RT % 8 == 0 ? 8 : 4;                      // 128: 4 x 2, 256: 8 x 1, 384: 4 x 2, 512: 8 x 1
  const int WN = 8 / WM, JW = (NT1 + WN - 1) / WN;
  if (JW > (pool ? 5 : 4)) return false;
  const int n_h = (R + 2) * (W + 2);
  const int tms1 = tm1 == 128 ? 7 : 6, tms2 = tm2 == 128 ? 7 : 6;
  const size_t lds = (size_t)(Cin / 64) * NT0 * 32 * 64 + (size_t)(Sp / 16) * ((n_h + 63) / 64) * 1024 + ((size_t)(dual1 ? 28 : 20) << tms1) +
                     (size_t)(N2 >> tms2) * ((size_t)20 << tms2) + 64 + (pool ? (size_t)R * W * N2 : 0);
  if (lds > 160 * 1024) return false;
  if (f) {
    const int PH = (H - 3 + 1) / 2 + 1;                    // ceil((H - 3) / 2) + 1
    f->R = R; f->NT0 = NT0; f->WM = WM; f->tiles_per_img = pool ? (PH + kPR - 1) / kPR : (H + R - 1) / R;
    f->PR = kPR;
    set_fast_div((uint32_t)W, &f->w_m, &f->w_s); set_fast_div((uint32_t)(Sp / 16), &f->g_m, &f->g_s);
  }
  if (lds_out) *lds_out = lds;
  return true;
}

template <int MT0, bool DUAL1, int JW, bool POOL = false>
static int launch_fire2(const FireArgs& a, size_t lds, hipStream_t s) {
  auto fn = conv_fire_kernel<MT0, DUAL1, JW, POOL>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn), 160 * 1024)) return -1;
  TF2_LAUNCH_NAME("conv_fire_kernel<%dx%d,C%d,S%d,N%d%s%s> (%d bands per image)", a.H, a.W, a.Cin, a.Sp, a.N2, DUAL1 ? ",dual squeeze" : "", POOL ? ",3x3/2 pool" : "", a.tiles_per_img);
  TF2_LAUNCH(fn, dim3(a.B * a.tiles_per_img), dim3(512), lds, s, a);
  return launch_ok() ? 0 : -1;
}

int launch_conv_fire(const FireArgs& a0, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  FireArgs a = a0;
  size_t lds = 0;
  if (!conv_fire_geometry(a.H, a.W, a.Cin, a.Sp, a.N2, a.tm1, a.tm2, a.dual1, a.pool, &a, &lds)) return 1;
  const int NT1 = (a.R * a.W + 31) / 32, JW = (NT1 + 8 / a.WM - 1) / (8 / a.WM);
  const bool mt2 = a.Sp > 32;
  if (a.pool) {
    if (!a.relu2 || a.PH != (a.H - 2) / 2 + 1 || a.PW != a.PH) return 1;
    if (mt2) return a.dual1 ? launch_fire2<2, true, 5, true>(a, lds, s) : launch_fire2<2, false, 5, true>(a, lds, s);
    return a.dual1 ? launch_fire2<1, true, 5, true>(a, lds, s) : launch_fire2<1, false, 5, true>(a, lds, s);
  }
#define TF2_FIRE(J_) do { if (mt2) return a.dual1 ? launch_fire2<2, true, J_>(a, lds, s) : launch_fire2<2, false, J_>(a, lds, s); \
                          return a.dual1 ? launch_fire2<1, true, J_>(a, lds, s) : launch_fire2<1, false, J_>(a, lds, s); } while (0)
  if (JW <= 2) TF2_FIRE(2);
  TF2_FIRE(4);
#undef TF2_FIRE
}

}  // namespace tf2
