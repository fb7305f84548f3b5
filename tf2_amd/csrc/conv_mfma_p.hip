// conv_mfma_p.hip -- persistent, tile-streaming variant of conv_mfma2.hip (gfx950).
//
// Same arithmetic, packed image, LDS header, swizzle and epilogue as conv_mfma2.hip.  What changes is the
// schedule.  Measured on conv_mfma2 (tools/block_timeline.py, ResNet-50 batch 32): a 128x128 output tile of a
// short-K layer costs a block ~8-12k cycles, of which the MFMAs are a few hundred -- the rest is a serial chain
// (kernel arguments -> header / first operands land -> compute -> requantise -> store -> block exit -> next block
// launch) that two or three co-resident blocks per CU cannot hide.  Here a block stays resident and walks a list
// of pixel tiles of ONE channel tile; (tile, K-step) pairs form one flat stream of stages through the same
// 3-slot LDS ring, so the operands of the next tile(s) are already in flight while a tile is requantised and
// stored, the header is fetched once, and there is no block launch between tiles.
//
//  * 16 waves per block, each a 32x32 output tile (one MFMA accumulator): a wave's instruction stream issues
//    one instruction per ~5 cycles whatever the occupancy (tools/ubench/valu_peak.hip), so the epilogue's
//    ~9 instructions per output are spread over as many waves as the CU holds (2 blocks x 16 waves);
//  * the residual tile (feature_writer.cl:88-122) is prefetched one tile ahead by LDS-DMA into a wave-private
//    1 KiB region per 32x32 sub-tile (lane-linear: every lane later reads back exactly the 16 bytes it fetched),
//    issued right after the previous tile's epilogue has read the region -- no register staging, no extra barrier;
//  * VMEM bookkeeping: LDS-DMA stages, residual DMAs and the output stores (inline asm, every lane stores, masked
//    lanes to a dump line, so the instruction count is static) all count in vmcnt and retire in issue order
//    (tools/ubench/vmcnt_order.hip: 0 early retirements in 5e9 trials), so every wait is an exact count of the
//    operations younger than the stage needed, selected at run time from wave-uniform scalars;
//  * grid = 8 * n_mtiles * k blocks; block b runs on XCD b % 8, takes channel tile (b / 8) % n_mtiles and pixel
//    tiles slot, slot + slots, ... with slot = b % 8 + 8 * (b / (8 * n_mtiles)): the channel tiles of one pixel
//    tile share an XCD (activations hit in its L2).
//
// Reference semantics: see conv_mfma.hip / requant_epilogue.h (pe.cl:191-194, relu.cl:54, feature_writer.cl:88-122,
// sequencer.cl:287 zero padding).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 15]
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
  }
}

template <int WM, int WN, int WTM, int WTN, bool PADCHK, bool HAS_RES>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 2) / 4) void conv_mfma_p_kernel(ConvArgs a, int n_ptiles, int slots) {
  constexpr int S = 3;
  constexpr int NW = WM * WN;
  constexpr int TM = WM * WTM, TN = WN * WTN;
  constexpr int NTM = WTM / 32, NTN = WTN / 32;
  constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int AG = TM / 16, BG = TN / 16, NG = AG + BG;          // 16-row LDS-DMA groups of a stage
  constexpr int NI_LO = NG / NW, REM = NG % NW, NI_HI = NI_LO + (REM ? 1 : 0);
  constexpr int NR = NTM * NTN;                                     // stores (and residual loads) per tile and lane
  constexpr int NRL = HAS_RES ? NR : 0;
  static_assert(NI_HI + 2 * (NR + NRL) <= 15, "vmcnt immediate range");
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  int* const prm = reinterpret_cast<int*>(lds + S * STAGE);
  // LDS map: [ring S*STAGE][header hdr_bytes][residual: NW * NR regions of 1 KiB]

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ni_w = (REM == 0 || wave < REM) ? NI_HI : NI_LO;       // this wave's DMA instructions per stage
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;
  const int P = a.n_phases;
  int* const dsh = prm + kPrmWordsPerRow * TM;
  int* const steps = dsh + P * TM;
  int* const goff = steps + a.max_ent;
  int* const ghw = goff + a.max_ent * 4;

  long long tst[16];
#pragma unroll
  for (int i = 0; i < 16; i++) tst[i] = 0;
  int n_tst = 1;
#define TF2_PST() do { if (a.dbg2 && c_ord == 1 && n_tst < 15) { _Pragma("unroll") for (int i_ = 1; i_ < 15; i_++) if (i_ == n_tst) tst[i_] = (long long)__builtin_readcyclecounter(); n_tst++; } } while (0)
  if (a.dbg2) tst[0] = (long long)__builtin_readcyclecounter();
  const int M = a.n_mtiles;
  const int b = blockIdx.x;
  const int mtile = (b >> 3) % M;
  const int slot = (b & 7) + 8 * ((b >> 3) / M);
  const int n_my = slot < n_ptiles ? (n_ptiles - slot + slots - 1) / slots : 0;     // pixel tiles of this block
  if (n_my == 0) return;
  const int e_begin = a.e_start[mtile];
  const int n_ent = a.e_start[mtile + 1] - e_begin;
  const int Q = n_my * n_ent;                                       // stages of this block's stream

  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const int a_lane_off = (lane >> 2) * 64 + chunk * 16;

  // gather words of entries 0 and 1 by scalar loads from the header image (see conv_mfma2.hip)
  typedef const __attribute__((address_space(4))) i32x4* cvec_p;
  const size_t hdr_words = (size_t)mtile * (size_t)(a.hdr_bytes >> 2);
  cvec_p const hg = (cvec_p)(unsigned long long)(a.hdr + hdr_words + kPrmWordsPerRow * TM + P * TM + a.max_ent);
  int pro_off[2], pro_hw[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const i32x4 o = hg[s];
    pro_off[s] = chunk == 0 ? o[0] : chunk == 1 ? o[1] : chunk == 2 ? o[2] : o[3];
    pro_hw[s] = 0;
    if (PADCHK) {
      const i32x4 h = hg[a.max_ent + s];
      pro_hw[s] = chunk == 0 ? h[0] : chunk == 1 ? h[1] : chunk == 2 ? h[2] : h[3];
    }
  }

  // residual tile of this block's ord-th pixel tile -> this wave's LDS regions (one LDS-DMA per 32x32 sub-tile;
  // every lane fetches the 16 NHWC bytes its own epilogue lane needs; masked lanes read the zero page)
  int8_t* const res_lds = lds + S * STAGE + a.hdr_bytes + (size_t)wave * NR * 1024;
  auto res_issue = [&](int ord) {
    const int px0 = (slot + ord * slots) * TN;
#pragma unroll
    for (int i = 0; i < NTM; i++)
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        const int px = px0 + wn * WTN + j * 32 + (lane & 31);
        const int chl = mtile * TM + wm * WTM + i * 32 + 16 * half;
        const bool ok = px < g.n_pix && chl + 16 <= g.y_nvalid;
        const int8_t* rp = ok ? a.res + (size_t)px * g.res_cp + g.res_off + chl : a.zero;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(rp), TF2_LDS_PTR(res_lds + (i * NTN + j) * 1024), 16, 0, 0);
      }
  };

  // per-lane gather state of the activation row groups this wave owns, for the tile the ISSUE cursor is in
  const int8_t* brow_ptr[NI_HI];
  int brow_h[NI_HI], brow_w[NI_HI];
  bool brow_ok[NI_HI];
  const bool linear = g.stride == 1 && (g.pad_h | g.pad_w) == 0 && g.H * g.W == g.OHW;   // 1x1, stride 1: pixel == pixel
  auto set_rows = [&](int ord) {
    const int px0 = (slot + ord * slots) * TN;
#pragma unroll
    for (int j = 0; j < NI_HI; j++) {
      const int gi = wave + NW * j;
      brow_h[j] = -(1 << 20); brow_w[j] = 0; brow_ptr[j] = a.zero; brow_ok[j] = false;
      if (gi >= AG && gi < NG) {
        const int p = px0 + (gi - AG) * 16 + (lane >> 2);
        if (p < g.n_pix) {
          if (linear) {
            brow_h[j] = 0; brow_w[j] = 0;
            brow_ptr[j] = a.x + (long long)p * g.Cp_in;
          } else {
            const int bb = fast_div(p, g.ohw_m, g.ohw_s);
            const int rem = p - bb * g.OHW;
            const int oh = fast_div(rem, g.ow_m, g.ow_s);
            const int ow = rem - oh * g.OW;
            brow_h[j] = oh * g.stride - g.pad_h;
            brow_w[j] = ow * g.stride - g.pad_w;
            brow_ptr[j] = a.x + ((long long)bb * g.H * g.W + (long long)brow_h[j] * g.W + brow_w[j]) * g.Cp_in;
          }
          brow_ok[j] = true;
        }
      }
    }
  };

  auto issue_stage = [&](int e, int off, int hw, int slot_idx) {
    int8_t* const sl = lds + slot_idx * STAGE;
    const int8_t* wsrc = a.w + (size_t)e * A_BYTES + a_lane_off;
    int dh = 0, dw = 0;
    if (PADCHK) { dh = hw & 0xffff; dw = hw >> 16; }
#pragma unroll
    for (int j = 0; j < NI_HI; j++) {
      const int gi = wave + NW * j;
      if (gi < AG) {
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + gi * 1024), TF2_LDS_PTR(sl + gi * 1024), 16, 0, 0);
      } else if (gi < NG) {
        bool ok = off >= 0 && brow_ok[j];
        if (PADCHK) {
          const int ih = brow_h[j] + dh, iw = brow_w[j] + dw;
          ok = ok && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        }
        const int8_t* src = ok ? brow_ptr[j] + off : a.zero;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(sl + gi * 1024), 16, 0, 0);
      }
    }
  };

  // ---- block start: [residual of tile 0], header, stages 0 and 1 -- all in flight together ----
  if (HAS_RES) res_issue(0);
  {
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(a.hdr) + hdr_words * 4 + lane * 16;
    int8_t* hdst = reinterpret_cast<int8_t*>(prm);
    for (int i = wave; i * 1024 < a.hdr_bytes; i += NW)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hsrc + i * 1024), TF2_LDS_PTR(hdst + i * 1024), 16, 0, 0);
  }
  int i_ord = 0, i_e = 0;                  // issue cursor: this block's tile ordinal, entry inside the m-tile
  set_rows(0);
  auto advance_issue = [&]() {
    i_e++;
    if (i_e == n_ent) { i_e = 0; i_ord++; if (i_ord < n_my) set_rows(i_ord); }
  };
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (s < Q) {
      issue_stage(e_begin + i_e, i_e == 0 ? pro_off[0] : pro_off[1], i_e == 0 ? pro_hw[0] : pro_hw[1], s);
      advance_issue();
    }

  i32x16 acc[NTM][NTN];
  auto clear_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NTM; i++)
#pragma unroll
      for (int j = 0; j < NTN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
  };
  clear_acc();

  auto phase_shift = [&](int p) {       // Horner step: acc <<= dshift[p][channel]
#pragma unroll
    for (int i = 0; i < NTM; i++) {
      const int rb = wm * WTM + i * 32 + 4 * half;
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

  const int lo_bound = g.relu ? 0 : -128;
  const int rlo = g.add_relu ? 0 : -128;

  // ---- the stage stream ------------------------------------------------------------------------
  // VMEM queue of a wave, in issue order:  ... stage q (issued in step q-2) | [stores + next residual DMA, if step
  // q-2 ended a tile] | stage q+1 | [the same for step q-1] | ...; step q needs stage q and everything older.
  int c_ord = 0, c_e = 0;                  // compute cursor
  int cslot = 0, islot = S - 1;
  int phase = 0;
  int ex_prev1 = 0, ex_prev2 = 0;          // VMEM operations steps q-1 / q-2 appended after their stage issue
  int since_rl = 1 << 20;                  // VMEM operations issued after the residual DMA of the tile being computed
  int off_nx = 0, hw_nx = 0, next_b = 0x7fffffff;
  for (int q = 0; q < Q; q++) {
    // younger than stage q: stage q+1 (if it exists), then whatever steps q-2 / q-1 appended after their stage issue
    const int younger = (q + 1 < Q ? ni_w : 0) + ex_prev2 + ex_prev1;
    TF2_PST();
    long long w_in = 0;
    if (a.dbg2 && c_ord == 1) w_in = (long long)__builtin_readcyclecounter();
    wait_vmcnt_dyn(younger);
    if (a.dbg2 && c_ord == 1 && lane == 0 && blockIdx.x < 64) {
      long long* d = a.dbg2 + 8192 * 16 + ((size_t)blockIdx.x * 16 + wave) * 8 + (c_e ? 4 : 0);
      d[0] = w_in; d[1] = (long long)__builtin_readcyclecounter(); d[2] = younger; d[3] = since_rl;
    }
    TF2_PST();
    __builtin_amdgcn_s_barrier();          // every wave's part of stage q landed; ring slot (q-1)%S is free
    asm volatile("" ::: "memory");
    if (q == 0) {                          // the header is in LDS now
      off_nx = goff[i_e * 4 + chunk];
      hw_nx = PADCHK ? ghw[i_e * 4 + chunk] : 0;
      next_b = __builtin_amdgcn_readfirstlane(steps[0]);
    }
    const bool last = c_e == n_ent - 1;
    int ex_now = 0;
    TF2_PST();
    while (c_e == next_b) {                // rare: Horner phase boundary
      phase++; phase_shift(phase);
      next_b = __builtin_amdgcn_readfirstlane(steps[phase]);
    }
    const int8_t* A = lds + cslot * STAGE;
    const int8_t* B = A + A_BYTES;
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
    if (q + S - 1 < Q) {
      issue_stage(e_begin + i_e, off_nx, hw_nx, islot);
      advance_issue();
      islot = islot + 1 == S ? 0 : islot + 1;
      off_nx = goff[i_e * 4 + chunk];
      if (PADCHK) hw_nx = ghw[i_e * 4 + chunk];
      since_rl += ni_w;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < NTM; i++)
#pragma unroll
        for (int j = 0; j < NTN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
    cslot = cslot + 1 == S ? 0 : cslot + 1;
    TF2_PST();

    if (last) {
      // ---- tile done: remaining Horner phases, requantise, store ----
      while (phase + 1 < P) { phase++; phase_shift(phase); }
      // the residual DMA is older than `since_rl` operations; beyond 15 the step waits above have covered it
      TF2_PST();
      if (HAS_RES && since_rl <= 15) wait_vmcnt_dyn(since_rl);
      TF2_PST();
      // lane-derived values of the epilogue are re-derived here from an opaque lane id, so that hipcc does not keep
      // them live (hoisted) across the K loop, where the 64-register budget has no room for them
      int lane_e;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
      const int half_e = lane_e >> 5;
      const int px0 = (slot + c_ord * slots) * TN;
#pragma unroll
      for (int i = 0; i < NTM; i++) {
        const int rb = wm * WTM + i * 32;
        const int chl = mtile * TM + rb + 16 * half_e;
#pragma unroll
        for (int j = 0; j < NTN; j++) {
          const int px = px0 + wn * WTN + j * 32 + (lane_e & 31);
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
          i32x4 rv = {0, 0, 0, 0};
          if (HAS_RES) rv = *reinterpret_cast<const i32x4*>(res_lds + (i * NTN + j) * 1024 + lane_e * 16);
          const i32x4 out = g.fast ? requant_tile16<HAS_RES, (HAS_RES ? 1 : 2), true>(a16, prm, TM, rb + 4 * half_e, lo_bound, rlo, rv)
                                   : requant_tile16<HAS_RES, (HAS_RES ? 1 : 2), false>(a16, prm, TM, rb + 4 * half_e, lo_bound, rlo, rv);
          // every lane stores (a dump line when masked) so that the count of VMEM operations is static
          const bool ok = px < g.n_pix && chl + 16 <= g.y_nvalid;
          int8_t* dst = ok ? a.y + (size_t)px * g.y_cp + g.y_off + chl : a.dump + (size_t)(wave * 64 + lane_e) * 16;
          asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(out) : "memory");
        }
      }
      TF2_PST();
      ex_now = NR;
      if (HAS_RES && c_ord + 1 < n_my) {   // this wave has read its residual regions: refill them for the next tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        res_issue(c_ord + 1);
        ex_now += NRL; since_rl = 0;
      }
      clear_acc();
      phase = 0;
      next_b = __builtin_amdgcn_readfirstlane(steps[0]);
      c_e = 0; c_ord++;
    } else {
      c_e++;
    }
    ex_prev2 = ex_prev1; ex_prev1 = ex_now;
  }
  if (a.dbg2 && tid == 0) {
    long long* d = a.dbg2 + (size_t)blockIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 15; i++) d[i] = tst[i];
    d[15] = (long long)__builtin_readcyclecounter();
  }
}

template <int WM, int WN, int WTM, int WTN, bool PADCHK, bool HAS_RES>
static int launch_p3(const ConvArgs& a, hipStream_t s) {
  constexpr int TM = WM * WTM, TN = WN * WTN;
  constexpr int STAGE = (TM + TN) * 64;
  const size_t lds = (size_t)3 * STAGE + (size_t)a.hdr_bytes + (HAS_RES ? (size_t)WM * WN * (WTM / 32) * (WTN / 32) * 1024 : 0) + 64;
  static int usable = -1;       // 1 usable, 0 this instantiation spills
  auto fn = conv_mfma_p_kernel<WM, WN, WTM, WTN, PADCHK, HAS_RES>;
  if (usable < 0) {
    // a register spill would put scratch loads/stores (VMEM, counted in vmcnt) into the stage stream and break
    // the exact wait counts: such an instantiation is never launched (the caller falls back to conv_mfma2)
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn)) != hipSuccess) return -1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    usable = fa.localSizeBytes == 0 ? 1 : 0;
  }
  if (!usable) return 1;
  if (lds > 160 * 1024) return 1;
  const int n_ptiles = (a.g.n_pix + TN - 1) / TN;
  const int M = a.n_mtiles;
  int k = 512 / (8 * M);                        // two 16-wave blocks per CU at most
  if (k < 1) k = 1;
  const int k_need = (n_ptiles + 7) / 8;
  if (k > k_need) k = k_need;
  const int slots = 8 * k;
  hipLaunchKernelGGL(fn, dim3(8 * M * k), dim3(WM * WN * 64), lds, s, a, n_ptiles, slots);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int WM, int WN, int WTM, int WTN>
static int launch_p(const ConvArgs& a, hipStream_t s) {
  const bool pad = (a.g.pad_h | a.g.pad_w) != 0;
  if (a.g.has_res) return pad ? launch_p3<WM, WN, WTM, WTN, true, true>(a, s) : launch_p3<WM, WN, WTM, WTN, false, true>(a, s);
  return pad ? launch_p3<WM, WN, WTM, WTN, true, false>(a, s) : launch_p3<WM, WN, WTM, WTN, false, false>(a, s);
}

// Persistent tile-streaming kernel; TM fixed by the packed image.  Returns 1 if the layer does not qualify.  Needs ConvArgs::dump (a 16 KiB scratch line
// area for masked stores).
int launch_conv_mfma_p(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.n_mtiles > kMaxMtiles || a.dump == nullptr) return -4;
  if (TM == 128) return launch_p<4, 4, 32, 32>(a, s);
  if (TM == 64) return launch_p<2, 8, 32, 32>(a, s);
  return -1;
}

}  // namespace tf2
