// conv_mfma2.hip -- implicit-GEMM INT8 convolution, LDS-DMA pipelined (gfx950).
//
// Arithmetic: the reference's PE adds x * (+-2^s) into an int32 that wraps (pe.cl:27-49); here every weight row is split
// into 7-exponent windows of int8 values that the matrix cores multiply, and the windows are recombined by left shifts --
// the same value in Z/2^32 (proof and packed layout: weight_pack.cpp header).  Operand orientation: A = weights (rows =
// output channels), B = gathered NHWC activations (columns = pixels); both tiles sit in LDS as 64-byte rows whose four
// 16-byte chunks are XOR-swizzled with (row >> 2) & 3; the epilogue is requant_epilogue.h (pe.cl:185-203, relu.cl:54,
// feature_writer.cl:88-122).  How operands reach the matrix cores and how little latency a block exposes:
//
//  * everything the K loop and the epilogue need per m-tile (bias / final shift / alpha /
//    beta, Horner shifts, this m-tile's slab list with the phase boundaries encoded, and the
//    per-segment gather tables) is ONE contiguous header per m-tile in the packed image; it
//    is pulled into LDS by LDS-DMA at block start, together with the first weight tiles
//    (whose addresses need no table), so the block pays ONE memory latency before its
//    first gather instead of a chain (kernarg -> dir -> entries -> kinfo -> activations);
//  * weight tiles and gathered activation rows go HBM/L2 -> LDS directly with
//    `global_load_lds_dwordx4` (1 KiB per wave instruction, no VGPR staging) into a ring
//    of S stages; a stage is waited for with a COUNTED s_waitcnt vmcnt(N) and a raw
//    s_barrier, so S-1 stages stay in flight across barriers (cdna_hip_programming.md
//    "Pipelining across barriers").  The K loop contains no ordinary global load and no
//    64-bit LDS read: either makes hipcc (ROCm 7.2) drain the LDS-DMA queue with vmcnt(0)
//    every iteration.  The LDS destination of an LDS-DMA is lane-linear, so the XOR swizzle
//    is applied on the per-lane SOURCE address (rule 21) and again on the ds_read_b128
//    side; zero padding (sequencer.cl:287) is a read from a zero page;
//  * within an iteration: ds_read fragments, issue the next stage's DMAs (hides the LDS
//    latency), then the MFMAs, which keep running while the wave moves on to the next wait;
//  * the residual tile (feature_writer.cl:88-122) is prefetched into registers before the
//    loop; the epilogue packs with v_perm and stores 16 contiguous NHWC bytes per lane;
//  * three tile shapes: 128x128 (2x2 waves of 64x64), 64x256 (1x4 waves, N_out = 64
//    layers) and 64x64 (2x2 waves of 32x32) for layers whose grid would not fill 256 CUs.
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

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 15, "vmcnt immediate out of the prepared range");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}

// WM x WN waves (4, 8 or 16), each wave a WTM x WTN output tile (multiples of 32), S ring stages, OCC blocks
// per CU the register budget is set for.  A wave's instruction stream issues one instruction per ~5 ticks
// (2.1 ns) whatever the occupancy up to 8 waves/SIMD (tools/ubench/valu_peak.hip), so a block's latency is
// its per-wave instruction count: many light waves (32x32 or 32x64 tiles) beat four waves with 64x64 tiles.
// DUAL: two-phase layers packed with both exponent windows per entry (weight_pack.cpp): [hi TM rows][lo TM rows] of
// weights and ONE activation slab per K step, two accumulators, combined once as (hi << dshift[1]) + lo.
// DENSE (ConvArgs::dense): the gather words of a step come from its index (tf2_device.h dense_gather) instead of the header's
// goff / ghw tables and the m-tile's entry range is mtile * nslab .. + nslab: nothing in front of the first DMAs but the
// kernel arguments (one dependent scalar-load round trip and the LDS table reads of every step less).
// The body is a device function of (argument block, block index, grid size) so that ONE launch can carry two independent layers
// (conv_mfma2_pair_kernel below: a stage's shortcut convolution next to the first 1x1 of its first bottleneck -- same input, no
// dependence, same instantiation): blocks [0, n0) work on the first argument block, the rest on the second.
template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL, bool DENSE>
__device__ __forceinline__ void conv_mfma2_body(const ConvArgs& a, const int blk_x, const int nblk_x) {
  constexpr int NW = WM * WN;                  // waves per block
  constexpr int TM = WM * WTM, TN = WN * WTN;
  constexpr int NTM = WTM / 32, NTN = WTN / 32;   // 32x32 MFMA tiles per wave (rows, columns)
  constexpr int A_BYTES = (DUAL ? 2 : 1) * TM * 64, B_BYTES = TN * 64, STAGE = A_BYTES + B_BYTES;
  // LDS-DMA work of one stage: AG weight + BG activation 16-row groups (1 KiB, one wave instruction each), dealt
  // round-robin to the waves: wave w owns groups w, w + NW, ...  Waves < REM own NI_HI groups, the others NI_LO;
  // the counted waits are per class (wave-uniform branch).
  constexpr int AG = A_BYTES / 1024, BG = TN / 16, NG = AG + BG;
  constexpr int NI_LO = NG / NW, REM = NG % NW, NI_HI = NI_LO + (REM ? 1 : 0);
  // AG a multiple of NW (every shape but the single-window 64 x 256 one): a wave's j-th group is a weight group for
  // j < NA and an activation group after that, for every wave -- decided at compile time, no per-lane select
  constexpr bool STATIC_GRP = (AG % NW) == 0;
  constexpr int NA = AG / NW;
  static_assert((S - 2) * NI_HI <= 15, "vmcnt immediate range");
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  // LDS map: [ring S*STAGE][header: rows {bias, alpha, beta64.lo, beta64.hi} (4*TM) | lo (TM) | dshift (P*TM) | steps[max_ent] |
  //           goff[max_ent*4] | ghw[max_ent*4]]   (gather words resolved per (entry, chunk) at pack time;
  //           steps[p-1] = iteration at which phase p starts, INT_MAX after the last; max_ent includes S spare entries)
  int* const prm = reinterpret_cast<int*>(lds + S * STAGE);

  TF2_PRELOAD_CONV_ARGS(a);          // every kernel argument in SGPRs after two scalar-load round trips (tf2_device.h)
  TF2_PROBE_WORD(g.flags);           // prb: timing probes (tools/probe_run.py, -DTF2_PROBES builds only; constant 0 in the product)
  if (prb & kProbeExit0) return;
  long long* const adbg = a.dbg; long long* const adbg2 = a.dbg2;
  asm volatile("" :: "s"(adbg), "s"(adbg2));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool ni_hi = REM == 0 || wave < REM;         // wave-uniform DMA class
  const int wm = wave / WN, wn = wave % WN;
  const bool stamps = (adbg != nullptr) | (adbg2 != nullptr);     // one test on the production path
  const bool dbg_on = adbg != nullptr && blk_x == 0 && tid == 0;
#define TF2_STAMP(i) do { if (stamps) { if (dbg_on) adbg[i] = (long long)__builtin_readcyclecounter(); if (adbg2) tstamp[i] = (long long)__builtin_readcyclecounter(); } } while (0)
  long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  TF2_STAMP(0);
  const long long wall0 = (stamps && adbg2) ? (long long)wall_clock64() : 0;
  int* const dsh = prm + kPrmWordsPerRow * TM;
  int* const steps = dsh + P * TM;
  int* const goff = steps + a_max_ent;
  int* const ghw = goff + a_max_ent * 4;

  // XCD-aware remap: consecutive logical tiles (same pixel tile, all channel tiles) on one XCD
  const int nblk = nblk_x;
  int bid = blk_x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int ntile = fast_div_u(bid, mt_m, mt_s);                 // bid / n_mtiles
  const int mtile = bid - ntile * a_n_mtiles;
  const int px0 = ntile * TN;

  // LDS-DMA lane l of an instruction fills row (l>>2), 16-byte slot (l&3) of a 16-row group;
  // with the XOR swizzle slot c' of row r holds chunk c = c' ^ ((r>>2)&3), and r>>2 == l>>4
  // inside a group, so every lane always fetches the same chunk index:
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const int a_lane_off = (lane >> 2) * 64 + chunk * 16;       // inside a 16-row group of a weight tile

  // The gather words of the first S-1 stages straight from the header image in global memory with SCALAR loads
  // (constant address space, uniform address): they arrive with the kernel arguments' latency class, so the
  // first activation DMAs do not wait for the header to land in LDS (one memory latency less per block).
  typedef const __attribute__((address_space(4))) i32x4* cvec_p;
  const size_t hdr_words = (size_t)mtile * (size_t)(a_hdr_bytes >> 2);
  cvec_p const hg = (cvec_p)(unsigned long long)(ahdr + hdr_words + kPrmWordsPerRow * TM + P * TM + a_max_ent);
  // this m-tile's {first, end} entry: the last two words of steps[] (weight_pack.cpp), same scalar round trip
  typedef const __attribute__((address_space(4))) int __attribute__((ext_vector_type(2)))* cvec2_p;
  int e_begin, n_ent;
  int pro_off[S - 1], pro_hw[S - 1];
  const DenseGeom dg = {a_cslabs, a_cs_m, a_cs_s, a_k, a_kk_m, a_kk_s, a_dil, g.W, g.Cp_in};
  const int lane_c16 = chunk * 16;
  auto gather_of = [&](int sl, int& off, int& hw) {      // DENSE: this lane's gather words of slab sl
    int o, h;
    dense_gather(dg, sl, o, h);
    off = o + lane_c16; hw = h + (lane_c16 << 16);
  };
  if (DENSE) {
    n_ent = a_nslab; e_begin = ((mtile << a_e_shl) >> a_e_shr) * n_ent;
#pragma unroll
    for (int s = 0; s < S - 1; s++) gather_of(s, pro_off[s], pro_hw[s]);
  } else {
    const auto ee = *(cvec2_p)(unsigned long long)(ahdr + hdr_words + kPrmWordsPerRow * TM + P * TM + a_max_ent - 2);
    e_begin = ee[0];
    n_ent = ee[1] - ee[0];
#pragma unroll
    for (int s = 0; s < S - 1; s++) {
      const i32x4 o = hg[s];                               // s_load_dwordx4, unconditional
      pro_off[s] = chunk == 0 ? o[0] : chunk == 1 ? o[1] : chunk == 2 ? o[2] : o[3];
      pro_hw[s] = 0;
      if (PADCHK) {
        const i32x4 h = hg[a_max_ent + s];
        pro_hw[s] = chunk == 0 ? h[0] : chunk == 1 ? h[1] : chunk == 2 ? h[2] : h[3];
      }
    }
  }
  TF2_STAMP(1);
  if (prb & kProbeExit1) { if (pro_off[0] == 0x7eadbeef) ay[0] = 1; return; }

  auto issue_header = [&]() {
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(ahdr) + hdr_words * 4 + lane * 16;
    int8_t* hdst = reinterpret_cast<int8_t*>(prm);
    for (int i = wave; i * 1024 < a_hdr_bytes; i += NW)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hsrc + i * 1024), TF2_LDS_PTR(hdst + i * 1024), 16, 0, 0);
  };
  // residual tile prefetch FIRST (ordinary loads, the oldest entries of this wave's VMEM queue: every counted
  // wait below covers them; first use is in the epilogue)
  const int half = lane >> 5;
  i32x4 resv[NTM][NTN];
#pragma unroll
  for (int i = 0; i < NTM; i++)
#pragma unroll
    for (int j = 0; j < NTN; j++) {
      // no branch around the load (a join would make hipcc wait for it here): without a residual, or when
      // masked, every lane reads the zero page
      const int px = px0 + wn * WTN + j * 32 + (lane & 31);
      const int chl = mtile * TM + wm * WTM + i * 32 + 16 * half;
      const bool ok = g.has_res && px < g.n_pix && chl + 16 <= g.y_nvalid;
      const int8_t* rp = ok ? ares + (size_t)px * g.res_cp + g.res_off + chl : azero;
      resv[i][j] = *reinterpret_cast<const i32x4*>(rp);
    }
  asm volatile("" ::: "memory");           // keep the residual loads OLDER than every DMA below

  // per-lane gather state of the activation row groups this wave owns
  const int8_t* brow_ptr[NI_HI];
  int brow_h[NI_HI], brow_w[NI_HI];
  bool brow_ok[NI_HI];
#pragma unroll
  for (int j = 0; j < NI_HI; j++) {
    const int gi = wave + NW * j;            // group index: < AG weights, else activations
    brow_h[j] = -(1 << 20); brow_w[j] = 0; brow_ptr[j] = azero; brow_ok[j] = false;
    if (STATIC_GRP && j < NA) continue;
    if (gi >= AG && gi < NG) {
      const int p = px0 + (gi - AG) * 16 + (lane >> 2);
      if (p < g.n_pix) {
        const int b = fast_div(p, g.ohw_m, g.ohw_s);
        const int rem = p - b * g.OHW;
        const int oh = fast_div(rem, g.ow_m, g.ow_s);
        const int ow = rem - oh * g.OW;
        brow_h[j] = oh * g.stride - g.pad_h;
        brow_w[j] = ow * g.stride - g.pad_w;
        brow_ptr[j] = ax + ((long long)b * g.H * g.W + (long long)brow_h[j] * g.W + brow_w[j]) * g.Cp_in;
        brow_ok[j] = true;
      }
    }
  }

  // where this wave's weight row groups sit inside a storage entry (ConvArgs w_*: own tiles, halves of 128-row tiles, pairs of
  // 64-row tiles): group gi = 16 rows = 1 KiB
  const int w_sub = (mtile & 1) * a_w_sub;
  int a_goff[NI_HI];
#pragma unroll
  for (int j = 0; j < NI_HI; j++) {
    const int gi = wave + NW * j;
    constexpr int RG = TM / 16;                      // row groups of one window
    const int r = gi % RG, win = gi / RG;
    a_goff[j] = (r & 3) * 1024 + (TM == 128 ? ((r >> 2) & 1) * a_w_half : 0) + win * a_w_win;
  }
  // one stage = entry e (weights) + this lane's gather words off/hw (activations) into ring slot slot_idx
  auto issue_stage = [&](int e, int off, int hw, int slot_idx, bool in_loop = false) {
    int8_t* const slot = lds + slot_idx * STAGE;
    const int8_t* wsrc = aw + (size_t)e * a_w_ent + a_lane_off + w_sub;
    int dh = 0, dw = 0, pc = 0;              // pc: the segment's channel offset = its place in the layer's pad row
    if (PADCHK) { dh = hw & 0xff; dw = (hw >> 8) & 0xff; pc = (int)((unsigned)hw >> 16); }
#pragma unroll
    for (int j = 0; j < NI_HI; j++) {
      const int gi = wave + NW * j;
      if (STATIC_GRP ? j < NA : gi < AG) {
        if (!((prb & kProbeNoA) && in_loop))
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + a_goff[j]), TF2_LDS_PTR(slot + gi * 1024), 16, 0, 0);
      } else if ((j < NI_LO || ni_hi) && !((prb & kProbeNoB) && in_loop)) {
        bool ok = off >= 0 && brow_ok[j];
        if (PADCHK && !(prb & kProbeNoPad)) {
          const int ih = brow_h[j] + dh, iw = brow_w[j] + dw;
          ok = ok && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        }
        const int8_t* src = ok ? brow_ptr[j] + off : azero + pc;      // out of range: the stored form of x = 0 (weight_pack.cpp off_pad)
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(slot + gi * 1024), 16, 0, 0);
      }
    }
  };

  // ---- block start: header, then the first S-1 stages, all by LDS-DMA and all in flight together ----
  // VMEM queue of a wave: [residual, header, stage 0 .. stage S-2, then one stage per loop iteration]
  issue_header();
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (s < n_ent) issue_stage(e_begin + s, pro_off[s], pro_hw[s], s);
  TF2_STAMP(2);

  i32x16 acc[NTM][NTN], acc2[DUAL ? NTM : 1][DUAL ? NTN : 1];     // acc2: the low exponent window (DUAL)
#pragma unroll
  for (int i = 0; i < NTM; i++)
#pragma unroll
    for (int j = 0; j < NTN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0; if (DUAL) acc2[i][j][r] = 0; }

  auto phase_shift = [&](int p) {       // Horner step: acc <<= dshift[p][channel]
#pragma unroll
    for (int i = 0; i < NTM; i++) {
      const int rb = wm * WTM + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + p * TM + rb + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < NTN; j++)
            acc[i][j][G * 4 + r] = (int)((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31));
      }
    }
  };

  // ---- pipelined K loop ---------------------------------------------------------------------
  const int n_main = n_ent - (S - 1);      // iterations that still issue a stage S-1 ahead
  auto wait_main = [&]() {                 // stage `it` (and everything older) landed; S-2 younger stages may fly
    if (ni_hi) wait_vmcnt<(S - 2) * NI_HI>(); else wait_vmcnt<(S - 2) * NI_LO>();
  };
  // header + stage 0
  if (n_main > 0) wait_main(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  TF2_STAMP(3);
  int phase = 0;
  int cslot = 0;                           // ring slot of the stage being computed
  int islot = S - 1;                       // ring slot the next issued stage goes to
  // gather words of the next stage to issue, read one iteration ahead (tables are padded by S entries)
  int off_nx, hw_nx = 0;
  if (DENSE) gather_of(S - 1, off_nx, hw_nx);
  else { off_nx = goff[(S - 1) * 4 + chunk]; if (PADCHK) hw_nx = ghw[(S - 1) * 4 + chunk]; }
  // iteration at which the next Horner phase starts (steps[] holds the P-1 boundaries, then INT_MAX)
  int next_b = DENSE ? 0x7fffffff : __builtin_amdgcn_readfirstlane(steps[0]);

  auto body = [&](int it, bool issue) {
    if (!DUAL && !DENSE)
      while (it == next_b) {               // rare: phase boundary
        phase++; phase_shift(phase);
        next_b = __builtin_amdgcn_readfirstlane(steps[phase]);
      }
    const int8_t* A = lds + cslot * STAGE;
    const int8_t* B = A + A_BYTES;
    auto issue_next = [&]() {
      if (issue) {
        issue_stage(e_begin + it + S - 1, off_nx, hw_nx, islot, true);
        islot = islot + 1 == S ? 0 : islot + 1;
        if (DENSE) gather_of(it + S, off_nx, hw_nx);
        else { off_nx = goff[(it + S) * 4 + chunk]; if (PADCHK) hw_nx = ghw[(it + S) * 4 + chunk]; }
      }
    };
    if (DUAL) {
      // two accumulator sets leave no room for both K halves' fragments: one half at a time
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int c = ks * 2 + (lane >> 5);
        i32x4 af[NTM], af2[NTM], bf[NTN];
#pragma unroll
        for (int i = 0; i < NTM; i++) {
          const int row = wm * WTM + i * 32 + (lane & 31);
          const int o = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
          af[i] = *reinterpret_cast<const i32x4*>(A + o);
          af2[i] = *reinterpret_cast<const i32x4*>(A + TM * 64 + o);
        }
#pragma unroll
        for (int j = 0; j < NTN; j++) {
          const int row = wn * WTN + j * 32 + (lane & 31);
          bf[j] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
        }
        if (ks == 0) issue_next();
#pragma unroll
        for (int i = 0; i < NTM; i++)
#pragma unroll
          for (int j = 0; j < NTN; j++) {
            if (prb & kProbeNoMfma) { asm volatile("" :: "v"(af[i]), "v"(af2[i]), "v"(bf[j])); continue; }
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
            acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af2[i], bf[j], acc2[i][j], 0, 0, 0);
          }
      }
    } else {
      i32x4 af[2][NTM], bf[2][NTN];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int c = ks * 2 + (lane >> 5);
#pragma unroll
        for (int i = 0; i < NTM; i++) {
          const int row = wm * WTM + i * 32 + (lane & 31);
          af[ks][i] = *reinterpret_cast<const i32x4*>(A + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NTN; j++) {
          const int row = wn * WTN + j * 32 + (lane & 31);
          bf[ks][j] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
        }
      }
      issue_next();
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < NTM; i++)
#pragma unroll
          for (int j = 0; j < NTN; j++) {
            if (prb & kProbeNoMfma) { asm volatile("" :: "v"(af[ks][i]), "v"(bf[ks][j])); continue; }
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
          }
    }
    cslot = cslot + 1 == S ? 0 : cslot + 1;
  };

  int it = 0;
  long long t_wait = 0, t_bar = 0, t_body = 0;     // dbg (tools/layer_times.py --stamps): cycles of block 0 / wave 0 in the vmcnt wait, the barrier, the step body
  for (; it < n_main; it++) {
    if (it) {
      const long long c0 = dbg_on ? (long long)__builtin_readcyclecounter() : 0;
      wait_main();
      const long long c1 = dbg_on ? (long long)__builtin_readcyclecounter() : 0;
      if (!(prb & kProbeNoBar))
      __builtin_amdgcn_s_barrier();          // every wave's part of stage `it` landed; slot (it-1)%S is free
      asm volatile("" ::: "memory");         // compile-time fence: no LDS access may be hoisted above the barrier
      if (dbg_on) { const long long c2 = (long long)__builtin_readcyclecounter(); t_wait += c1 - c0; t_bar += c2 - c1; }
    }
    const long long c3 = dbg_on ? (long long)__builtin_readcyclecounter() : 0;
    body(it, true);
    if (dbg_on) t_body += (long long)__builtin_readcyclecounter() - c3;
  }
  if (dbg_on) { adbg[8] = t_wait; adbg[9] = t_bar; adbg[10] = t_body; adbg[11] = n_main; }
  for (; it < n_ent; it++) {               // tail: nothing left to issue, wait for everything
    if (it) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    body(it, false);
  }
  if (DUAL) {
    // combine the two windows: (hi << dshift[1][row]) + lo   (Z/2^32, as the Horner form)
#pragma unroll
    for (int i = 0; i < NTM; i++) {
      const int rb = wm * WTM + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + TM + rb + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < NTN; j++)
            acc[i][j][G * 4 + r] = (int)(((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31)) + (unsigned)acc2[i][j][G * 4 + r]);
      }
    }
  } else if (!DENSE) {
    while (phase + 1 < P) { phase++; phase_shift(phase); }      // phases that start after the last entry
  }
  TF2_STAMP(5);

  if (prb & kProbeNoEpi) return;
  // ---- epilogue --------------------------------------------------------------------------------
  // per output: v = bias + (sum << lo);  x = low32((v*alpha + (beta << 20)) >> 20)   (pe.cl:191-193, the
  // 32-bit truncation and wrap-around of the reference kept);  y = sat(x + 2^14) >> 15, which equals
  // ((x >> 14) + 1) >> 1 everywhere except where both clamp to 127;  clamp (+ReLU) (pe.cl:194, relu.cl:54);
  // residual: int16 add, clamp, ReLU (feature_writer.cl:119-122) in the C/D register layout.
  const int lo_bound = g.relu ? 0 : -128;
  const int rlo = g.add_relu ? 0 : -128;
  auto epilogue = [&](auto has_res_c, auto fast_c) {
    constexpr bool HAS_RES = decltype(has_res_c)::value;
    constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
    for (int i = 0; i < NTM; i++) {
      const int rb = wm * WTM + i * 32;                        // tile row base inside the block tile
      const int chl = mtile * TM + rb + 16 * half;
      // the NTN column tiles of this row tile row by row in lockstep: every parameter row is read once (requant_epilogue.h)
      int a16s[NTN][16];
      i32x4 outs[NTN];
#pragma unroll
      for (int j = 0; j < NTN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) a16s[j][r] = acc[i][j][r];
      requant_tiles16<NTN, HAS_RES, 1, FAST>(a16s, outs, prm, TM, rb + 4 * half, lo_bound, rlo, resv[i], g.dbl_out != 0, g.fast == 2);
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        const int px = px0 + wn * WTN + j * 32 + (lane & 31);
        const i32x4 out = outs[j];
        if (px < g.n_pix && chl + 16 <= g.y_nvalid) {
          i32x4* dst = reinterpret_cast<i32x4*>(ay + (size_t)px * g.y_cp + g.y_off + chl);
          *dst = out;
        }
      }
    }
  };
  if (g.fast == 1) { if (g.has_res) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{}); }
  else { if (g.has_res) epilogue(std::true_type{}, std::false_type{}); else epilogue(std::false_type{}, std::false_type{}); }
  if (dbg_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TF2_STAMP(6); }
  if (adbg2 && tid == 0) {
    long long* d = adbg2 + (size_t)blk_x * 8;
    d[0] = tstamp[0]; d[1] = (long long)__builtin_readcyclecounter();
    d[4] = tstamp[1]; d[5] = tstamp[2]; d[6] = tstamp[3];
    d[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID (id 4), 32 bits
    d[3] = wall0; d[7] = (long long)wall_clock64();                        // 100 MHz, chip-wide
  }
#undef TF2_STAMP
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL, bool DENSE>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * OCC) / 4) void conv_mfma2_kernel(ConvArgs a) {
  conv_mfma2_body<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, DENSE>(a, (int)blockIdx.x, (int)gridDim.x);
}

// two independent layers of the same instantiation in one launch
template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL, bool DENSE>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * OCC) / 4) void conv_mfma2_pair_kernel(ConvArgs a0, ConvArgs a1, int n0) {
  const int b = (int)blockIdx.x;
  if (b < n0) conv_mfma2_body<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, DENSE>(a0, b, n0);
  else conv_mfma2_body<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, DENSE>(a1, b - n0, (int)gridDim.x - n0);
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL, bool DENSE>
static int launch_cfg3(const ConvArgs& a, hipStream_t s) {
  constexpr int TM = WM * WTM, TN = WN * WTN;
  constexpr int STAGE = ((DUAL ? 2 : 1) * TM + TN) * 64;
  const size_t lds = (size_t)S * STAGE + (size_t)a.hdr_bytes + 64;
  auto fn = conv_mfma2_kernel<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, DENSE>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  const int ntiles = (a.g.n_pix + TN - 1) / TN;
  TF2_LAUNCH_NAME("conv_mfma2_kernel<%dx%d waves of %dx%d,S%d,%s%s%s>", WM, WN, WTM, WTN, S, PADCHK ? "pad," : "", DUAL ? "dual," : "", DENSE ? "dense" : "tables");
  TF2_LAUNCH(fn, dim3(ntiles * a.n_mtiles), dim3(WM * WN * 64), lds, s, a);
  return launch_ok() ? 0 : -1;
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL>
static int launch_cfg2(const ConvArgs& a, hipStream_t s) {
  return a.dense ? launch_cfg3<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, true>(a, s) : launch_cfg3<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, false>(a, s);
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
  // bounds checks on the gathered taps are only needed for padded convolutions
  return (a.g.pad_h | a.g.pad_w) ? launch_cfg2<WM, WN, WTM, WTN, S, OCC, true, false>(a, s) : launch_cfg2<WM, WN, WTM, WTN, S, OCC, false, false>(a, s);
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC>
static int launch_dual(const ConvArgs& a, hipStream_t s) {
  return (a.g.pad_h | a.g.pad_w) ? launch_cfg2<WM, WN, WTM, WTN, S, OCC, true, true>(a, s) : launch_cfg2<WM, WN, WTM, WTN, S, OCC, false, true>(a, s);
}

// TM is fixed by the packed image (64 or 128); the pixel-tile shape is picked per launch so
// that small grids still spread over the 256 CUs: 8-wave blocks with 32x64 wave tiles (measured best: 4 waves of 64x64 tiles were
// equal, 16 waves of 32x32 6-9 % slower -- profiles/r03_experiments.txt item 18; both shapes left the tree in round 4).
int launch_conv_mfma2(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr long t256 = 384;
  const long blocks256 = (long)((a.g.n_pix + 255) / 256) * a.n_mtiles;
  if (a.dual) {
    if (TM == 128) return launch_dual<4, 2, 32, 64, 3, 2>(a, s);
    if (TM == 64) return blocks256 >= t256 ? launch_dual<2, 4, 32, 64, 3, 2>(a, s) : launch_dual<2, 2, 32, 32, 4, 4>(a, s);
    return -1;
  }
  if (TM == 128) {
    return launch_cfg<4, 2, 32, 64, 3, 2>(a, s);
  }
  if (TM == 64) {
    if (blocks256 >= t256) return launch_cfg<2, 4, 32, 64, 3, 2>(a, s);
    return launch_cfg<2, 2, 32, 32, 4, 4>(a, s);
  }
  return -1;
}

// ---- pair launch: the 128-row 8-wave shape, both layers dense and unpadded, same window form ---------------------------------
template <bool DUAL, bool DENSE>
static int launch_pair2(const ConvArgs& a0, const ConvArgs& a1, hipStream_t s) {
  constexpr int TM = 128, TN = 128, S = 3;
  constexpr int STAGE = ((DUAL ? 2 : 1) * TM + TN) * 64;
  const size_t lds = (size_t)S * STAGE + (size_t)(a0.hdr_bytes > a1.hdr_bytes ? a0.hdr_bytes : a1.hdr_bytes) + 64;
  auto fn = conv_mfma2_pair_kernel<4, 2, 32, 64, 3, 2, false, DUAL, DENSE>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  const int n0 = ((a0.g.n_pix + TN - 1) / TN) * a0.n_mtiles, n1 = ((a1.g.n_pix + TN - 1) / TN) * a1.n_mtiles;
  TF2_LAUNCH_NAME("conv_mfma2_pair_kernel<4x2 waves of 32x64,S3,%s%s> (%d + %d blocks)", DUAL ? "dual," : "", DENSE ? "dense" : "tables", n0, n1);
  TF2_LAUNCH(fn, dim3(n0 + n1), dim3(512), lds, s, a0, a1, n0);
  return launch_ok() ? 0 : -1;
}

// ... and the 64-row four-wave shape of the small grids (batch 1-4: ResNet-50 rows 1 | 2 on the 56 x 56 maps, round 6)
template <bool DUAL, bool DENSE>
static int launch_pair2_small(const ConvArgs& a0, const ConvArgs& a1, hipStream_t s) {
  constexpr int TM = 64, TN = 64, S = 4;
  constexpr int STAGE = ((DUAL ? 2 : 1) * TM + TN) * 64;
  const size_t lds = (size_t)S * STAGE + (size_t)(a0.hdr_bytes > a1.hdr_bytes ? a0.hdr_bytes : a1.hdr_bytes) + 64;
  auto fn = conv_mfma2_pair_kernel<2, 2, 32, 32, 4, 4, false, DUAL, DENSE>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  const int n0 = ((a0.g.n_pix + TN - 1) / TN) * a0.n_mtiles, n1 = ((a1.g.n_pix + TN - 1) / TN) * a1.n_mtiles;
  TF2_LAUNCH_NAME("conv_mfma2_pair_kernel<2x2 waves of 32x32,S4,%s%s> (%d + %d blocks)", DUAL ? "dual," : "", DENSE ? "dense" : "tables", n0, n1);
  TF2_LAUNCH(fn, dim3(n0 + n1), dim3(256), lds, s, a0, a1, n0);
  return launch_ok() ? 0 : -1;
}

// both rows take the same instantiation when launched alone (launch_conv_mfma2's choice): 128-row tiles, or 64-row tiles on grids that take
// the four-wave shape
static bool pair_small_shape(const ConvArgs& a) { return (long)((a.g.n_pix + 255) / 256) * a.n_mtiles < 384; }
bool conv_mfma2_pair_eligible(const ConvArgs& a0, int TM0, const ConvArgs& a1, int TM1) {
  if (TM0 != TM1 || (TM0 != 128 && TM0 != 64) || a0.dense != a1.dense || a0.dual != a1.dual) return false;
  if ((a0.g.pad_h | a0.g.pad_w | a1.g.pad_h | a1.g.pad_w) != 0) return false;
  if (TM0 == 64 && !(pair_small_shape(a0) && pair_small_shape(a1))) return false;
  return true;
}

int launch_conv_mfma2_pair(const ConvArgs& a0, const ConvArgs& a1, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (TM == 64) {
    if (a0.dense) return a0.dual ? launch_pair2_small<true, true>(a0, a1, s) : launch_pair2_small<false, true>(a0, a1, s);
    return a0.dual ? launch_pair2_small<true, false>(a0, a1, s) : launch_pair2_small<false, false>(a0, a1, s);
  }
  if (a0.dense) return a0.dual ? launch_pair2<true, true>(a0, a1, s) : launch_pair2<false, true>(a0, a1, s);
  return a0.dual ? launch_pair2<true, false>(a0, a1, s) : launch_pair2<false, false>(a0, a1, s);
}

}  // namespace tf2
