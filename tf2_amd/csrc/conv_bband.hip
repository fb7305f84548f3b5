// conv_bband.hip -- a whole identity bottleneck (1x1 reduce C -> M, 3x3 / stride 1 / pad 1 M -> M, 1x1 expand M -> C + residual +
// ReLU; pe.cl:144-203 three times, feature_writer.cl:119-122 once) in ONE launch with NO exchange between blocks (gfx950).
//
// Why (round 4): with several batches in flight the step is the serialised sum of the chip's pipes -- VALU first: the ring kernels
// spend 11-23 VALU instructions per output on their 1x1 / 3x3 layers where the requantisation itself needs ~7 (profiles/
// r04_pmc_conv_b32_conc1.json) -- and the group launches of conv_bgroup.hip, which remove two thirds of the launches, cannot be
// used there: their eight blocks per image MEET, so they want every CU's LDS and co-residency guarantees.  This
// kernel fuses the same three rows without any meeting: a block owns R output rows x the full width of ONE image and ALL
// channels; it recomputes the reduce for its two halo rows (a 1x1 layer: (R + 2) / R of that layer's work) and keeps both
// intermediates in LDS.  Nothing a block reads is written by another block of the launch, so any number of such launches may
// share the chip with anything else: no flags, no epochs, no traps.
//
//   phase 0  reduce over the (R + 2) x W halo pixels: the band's input (one contiguous NHWC range) streams through two LDS chunk
//            buffers of SC 64-byte channel slabs by LDS-DMA; weights global -> registers, PF steps ahead (conv_bneck's scheme);
//            requantised straight into the 3x3's halo tile in LDS (rows outside the image and the two border columns keep the
//            stored form of x = 0, sequencer.cl:287);
//   phase 1  3x3: a tap is the halo tile at a shifted address; requantised into the expand's B tile in LDS;
//   phase 2  expand in C / M passes of M channels, residual = the band's own input rows (ordinary loads, L2-warm), 16-byte stores.
//
// Wave tiling: 8 waves = WM x WN, a wave owns MT x 32 channels (MT = M / (32 WM)) and every WN-th 32-pixel column tile.
// Weight tiles, header rows and the requantisation are the packed image's and requant_epilogue.h's: bit-identical to the three
// separate launches (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"
#include "vm_track.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

#ifdef TF2_CHECK_DMA
TF2_DMA_CHECK_COUNTERS(g_bband_dma_check);
void conv_bband_check_counts(unsigned long long out[2]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bband_dma_check), 16); }
#endif

// The reduce phase's per-wave VM schedule, stated ONCE: the issue code and the counted wait of step0 both read it (vm_track.h).
//   step v = chunk ch * SC + slab sl, in issue order:
//     [sl == 0 and ch > 0]             WAIT for the DMAs of chunk ch, barrier
//     always                           the fragment loads of step v + 2: frag_loads(v) 16-byte loads
//     [sl == 0 and 0 < ch < NCH - 1]   the DMAs of chunk ch + 1
// (chunks 0 and 1 are issued in the prologue and drained there with vmcnt(0))
template <int KS1, int SC, int MT, bool DUAL1>
struct BbSched {
  static constexpr bool low_at(int v) { return DUAL1 && v + 2 < KS1; }      // step v also loads the LOW window's fragments of step v + 2
  static constexpr int frag_loads(int v) { return 2 * MT + (low_at(v) ? 2 * MT : 0); }
  // VM operations issued BEHIND the DMAs of chunk c (step (c - 1) * SC, behind that step's own fragment loads) when step c * SC waits
  static constexpr int since_dma(int c) {
    int n = 0;
    for (int v = (c - 1) * SC + 1; v < c * SC; v++) n += frag_loads(v);
    return n;
  }
  // ... of which the compiler's own waits for the fragments leave at most the last two steps' in flight: no point in allowing more
  static constexpr int wait_n(int c) { return vm_min(since_dma(c), (SC - 1 < 2 ? SC - 1 : 2) * frag_loads(c * SC - 1)); }
};

template <int T, int N, class F>
__device__ __forceinline__ void bb_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); bb_static_for<T + 1, N>(fn); }
}

constexpr int kBbPF = 2;                     // weight fragments in flight ahead of their MFMAs (steps)

// LDS-DMA as inline assembly: the compiler's wait-count pass does not know these loads, so it neither drains the queue (vmcnt(0))
// in front of the next LDS read nor orders them against anything -- every wait for them is written out below (the counter is in
// order: a counted wait for a younger ordinary load covers every older DMA)
__device__ __forceinline__ void bb_dma16(const int8_t* src, int8_t* lds_dst) {
  const unsigned l = (unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(l) : "memory", "m0");
}

// M: channels of the intermediates; WN: pixel-tile columns of the wave grid; NT0 / NT1: 32-pixel column tiles of the halo band
// ((R + 2) * W <= 32 NT0) and of the band itself (R * W <= 32 NT1); SC: channel slabs per chunk of the input stream
// NW: waves per block (8: two per SIMD, up to 256 registers; 16: four per SIMD, 128 registers -- the requantisation phases are VALU
// work that one or two waves per SIMD cannot issue at rate)
// DUAL1 / DUAL2: the reduce / the 3x3 is a two-window layer (weight_pack.cpp: entries [hi rows | lo rows]).  The reduce keeps two
// accumulator sets over the one input stream and combines them once, (hi << dshift[1]) + lo; the 3x3 sweeps its LDS-resident halo
// tile window by window into ONE set with the Horner shift in between (conv_bneck's scheme) -- both exact in Z/2^32.
// rows per band of an instantiation: (map side, column tiles of the halo band) <-> R (launch_conv_bband's table)
__host__ __device__ constexpr int bband_rows_of(int M, int NT0) { return M == 256 ? (NT0 == 4 ? 7 : NT0 == 3 ? 4 : 2) : (NT0 == 8 ? 7 : 4); }

template <int M, int NW, int WN, int NT0, int NT1, int SC, bool DUAL1, bool DUAL2>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_bband_kernel(BBandArgs a) {
  constexpr int C = 4 * M;
  constexpr int WM = NW / WN, MT = M / (32 * WM);
  static_assert(WM * WN == NW && MT * 32 * WM == M && NT0 % WN == 0 && NT1 % WN == 0, "wave grid");
  constexpr int J0 = NT0 / WN, J1 = NT1 / WN;            // column tiles per wave
  constexpr int LEAN = (NW == 16 || (DUAL1 && J0 >= 4)) ? 2 : 0;                 // requant_epilogue.h: header rows read two ahead instead of all sixteen at once (128-register budget)
  constexpr int KS1 = C / 64, KS2 = M / 64, NE = 9 * KS2;
  constexpr int NP0 = 32 * NT0, NP1 = 32 * NT1;
  static_assert(KS1 % SC == 0 && SC >= 2, "whole chunks; a chunk's DMAs are told from its fragment loads by a counted wait (below)");
  constexpr int NCH = KS1 / SC;                          // chunks of the input stream
  constexpr int CHUNK = SC * NP0 * 64;
  constexpr int MID2 = KS2 * NP1 * 64;
  constexpr int RING = CHUNK > MID2 ? CHUNK : MID2;

  // two chunk buffers as separate LDS objects: a DMA into one is then not ordered against reads of the other
  __shared__ __attribute__((aligned(1024))) int8_t ring0[RING];
  __shared__ __attribute__((aligned(1024))) int8_t ring1[CHUNK];
  extern __shared__ __attribute__((aligned(1024))) int8_t dyn[];     // [mid1 halo tile][hdr1][hdr2][hdr3]
  int8_t* const mid2 = ring0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;
  // the map side is the template's (square maps: 14 for M = 256, 28 for M = 128; the launcher checks): every p / W below is then a
  // multiply-high instead of the ~35-instruction division by a run-time value -- two dozen per lane and block, a fifth of the block's
  // VALU work before (round 4)
  constexpr int W = M == 256 ? 14 : M == 128 ? 28 : 56, Wp = W + 2, H = W;
  // rows per band: the instantiation's (the launcher checks a.R), so that every offset into the halo tile is an immediate
  constexpr int R = bband_rows_of(M, NT0);
  constexpr int n_h = (R + 2) * Wp;
  // the halo tile, per 64-channel slab: four PLANES of 16 bytes per pixel (plane k = K bytes 16 k .. 16 k + 15 of every pixel), each
  // padded to whole 1 KiB DMA groups.  A lane's MFMA fragment of pixel h is 16 bytes at plane[half + 2 ks] + 16 h: the sixteen lanes
  // a ds_read_b128 services together read 256 contiguous bytes whatever the tap -- no bank conflict without a swizzle, and tap,
  // slab and K half are an immediate offset (the swizzled [pixel][64] form cost one address add per fragment read: 288 per wave)
  constexpr int n_grp_h = ((n_h + 63) >> 6) * 4;
  constexpr int slabb = n_grp_h * 1024;                  // bytes of one 64-channel slab of the halo tile
  constexpr int planeb = slabb / 4;
  int8_t* const mid1 = dyn;
  const int tms1 = a.tm1 == 128 ? 7 : 6, tms2 = a.tm2 == 128 ? 7 : 6, tms3 = a.tm3 == 128 ? 7 : 6;
  // bytes of one m-tile's header image: rows {bias, alpha, addend64} | lo | dshift[P] (the Horner shifts: two-window layers only)
  const int hst1 = (DUAL1 ? 28 : 20) << tms1, hst2 = (DUAL2 ? 28 : 20) << tms2, hst3 = 20 << tms3;
  int8_t* const hdr1 = mid1 + KS2 * slabb;
  int8_t* const hdr2 = hdr1 + (M >> tms1) * hst1;
  int8_t* const hdr3 = hdr2 + (M >> tms2) * hst2;

  // XCD-aware remap: the bands of one image on one XCD (they share halo rows of the input and, all of them, the weights)
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int img = bid / a.tiles_per_img;
  const int r0 = (bid - img * a.tiles_per_img) * R;
  const int rows = (H - r0) < R ? (H - r0) : R;          // valid output rows of this band
  const int n_px = rows * W;
  const int n_p0 = (R + 2) * W;                          // halo-band pixels (row r0 - 1 first)
  const long long pix_base = ((long long)img * H + r0) * W;        // NHWC pixel index of band pixel 0
  const long long pix0 = pix_base - W;                   // ... of halo-band pixel 0 (may lie outside the image: never dereferenced then)

  long long* const dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bband_timeline.py: 100 MHz wall clock per phase
#define BB_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  BB_STAMP(0);

  // LDS-DMA: lane l fills pixel row l >> 2, slot l & 3 of a 16-pixel group, which holds chunk slot ^ ((row >> 2) & 3)
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;

  // ---- prologue -----------------------------------------------------------------------------------------------------------
  // (1) the halo tile filled with the stored form of x = 0 (the 3x3's pad row): borders and rows outside the image stay that way
  for (int gi = wave; gi < n_grp_h * KS2; gi += NW) {
    const int s = gi / n_grp_h, grp = gi - s * n_grp_h;
    bb_dma16(a.zero2 + s * 64 + (grp / (n_grp_h / 4)) * 16, mid1 + s * slabb + grp * 1024);
  }
  // (2) the input stream: chunk c = channel slabs [c * SC, (c + 1) * SC) of the NP0 halo-band pixels -> [slab][pixel][64] swizzled
  auto issue_chunk = [&](int c, int8_t* buf) {
    for (int gi = wave; gi < SC * (NP0 / 16); gi += NW) {
      const int sl = gi / (NP0 / 16), grp = gi - sl * (NP0 / 16);
      const int p = grp * 16 + drow;
      const int row = r0 - 1 + p / W;
      const bool ok = p < n_p0 && (unsigned)row < (unsigned)H;
      const int8_t* src = ok ? a.x + (size_t)(pix0 + p) * C + (c * SC + sl) * 64 + chunk * 16 : a.zero + chunk * 16;
#ifdef TF2_CHECK_DMA
      dma_stamp(buf + sl * (NP0 * 64) + grp * 1024);
#endif
      bb_dma16(src, buf + sl * (NP0 * 64) + grp * 1024);
    }
  };
  issue_chunk(0, ring0);
  if (NCH > 1) issue_chunk(1, ring1);
  // (3) header images (rows {bias | dbl, alpha, addend64} | lo per m-tile) by ordinary loads: 20 * tm bytes per m-tile, packed
  {
    auto hdr_copy = [&](const int32_t* hdr, int hdr_bytes, int tms, int n_mt, int8_t* dst, int words_per_row) {
      const int per = words_per_row << (tms - 2);          // 16-byte pieces per m-tile
      for (int i = tid; i < n_mt * per; i += NW * 64) {
        const int mt = i / per, k = i - mt * per;
        const i32x4 v = *reinterpret_cast<const i32x4*>(reinterpret_cast<const int8_t*>(hdr) + (size_t)mt * hdr_bytes + k * 16);
        *reinterpret_cast<i32x4*>(dst + (size_t)mt * (per * 16) + k * 16) = v;
      }
    };
    hdr_copy(a.hdr1, a.hdr1_bytes, tms1, M >> tms1, hdr1, DUAL1 ? 7 : 5);
    hdr_copy(a.hdr2, a.hdr2_bytes, tms2, M >> tms2, hdr2, DUAL2 ? 7 : 5);
    hdr_copy(a.hdr3, a.hdr3_bytes, tms3, C >> tms3, hdr3, 5);
  }

  // weight fragments: a lane's MFMA A fragment is 16 contiguous bytes of its row in the packed tile [tm rows][64]
  struct Afr { i32x4 k[MT][2]; };
  const int cb_w = wm * (MT * 32);                       // this wave's first channel inside an M-channel pass
  // (wins: windows per entry of the layer's packed tiles, win: the one to fetch)
  const unsigned a_lane_off = (unsigned)((lane & 31) * 64 + half * 16);
  auto load_a = [&](Afr& f, const int8_t* w, int tms, int nslab, int cb, int slab, int wins = 1, int win = 0) {
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int ch = cb + i * 32;
      const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
      // (wave-uniform base + a 32-bit lane offset: the scalar-base form of global_load, no 64-bit address arithmetic per lane)
      const int8_t* pu = w + (((((size_t)mt * nslab + slab) * wins + win) << tms) + ro) * 64;
      f.k[i][0] = *reinterpret_cast<const i32x4*>(pu + a_lane_off);
      f.k[i][1] = *reinterpret_cast<const i32x4*>(pu + a_lane_off + 32);
    }
  };
  // fragments of global step v (phase 0: slab v of the reduce's high window; phase 1: (window, tap, slab) of the 3x3; phase 2 loads its own)
  constexpr int N1 = (DUAL2 ? 2 : 1) * NE;               // steps of phase 1
  auto load_step = [&](Afr& f, auto v_c) {
    constexpr int v = decltype(v_c)::value;
    if constexpr (v < KS1) load_a(f, a.w1, tms1, KS1, cb_w, v, DUAL1 ? 2 : 1, 0);
    else if constexpr (v < KS1 + N1) load_a(f, a.w2, tms2, NE, cb_w, (v - KS1) % NE, DUAL2 ? 2 : 1, (v - KS1) / NE);
    else load_a(f, a.w3, tms3, KS2, cb_w, v - KS1 - N1);                                     // the expand's first fragments (pass 0)
  };
  Afr g0, g1, g2, g3;                                    // DUAL1: the reduce's LOW window fragments, rotating like f0..f3
#define BB_BUFL(v) ((v) % 4 == 0 ? g0 : (v) % 4 == 1 ? g1 : (v) % 4 == 2 ? g2 : g3)
  // four rotating buffers (PF = 2 would need three; four divides the step count of every phase, so that the pass loop of phase 2 can
  // be a run-time loop with the same buffer assignment in every iteration)
  Afr f0, f1, f2, f3;
  static_assert(kBbPF == 2 && (2 * KS2) % 4 == 0, "buffer rotation: one pass pair advances the step count by a multiple of four");
#define BB_BUF(v) ((v) % 4 == 0 ? f0 : (v) % 4 == 1 ? f1 : (v) % 4 == 2 ? f2 : f3)
  load_step(f0, std::integral_constant<int, 0>{});
  load_step(f1, std::integral_constant<int, 1>{});
  if constexpr (DUAL1) { load_a(g0, a.w1, tms1, KS1, cb_w, 0, 2, 1); load_a(g1, a.w1, tms1, KS1, cb_w, 1, 2, 1); }

  // (every lambda that touches the accumulators is always_inline: a call would take the arrays by reference, i.e. put them in scratch)
  i32x16 acc[MT][J0 > J1 ? J0 : J1];
  i32x16 acc2[DUAL1 ? MT : 1][DUAL1 ? J0 : 1];           // DUAL1: the reduce's low window
  auto zero_acc = [&](int nj) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < (J0 > J1 ? J0 : J1); j++)
        if (j < nj)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[i][j][r] = 0;
  };
  zero_acc(J0);
  if constexpr (DUAL1) {
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < J0; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc2[i][j][r] = 0;
  }

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // pad fill, chunks 0 and 1, headers: complete in every wave
  asm volatile("" ::: "memory");
  BB_STAMP(1);

  // ---- phase 0: reduce over the halo band ----------------------------------------------------------------------------------
  int bm0[J0];                                           // per-lane B address of column tile j inside a [pixel][64] slab
#pragma unroll
  for (int j = 0; j < J0; j++) {
    const int row = (wn + j * WN) * 32 + (lane & 31);
    bm0[j] = row * 64 + ((half ^ ((row >> 2) & 3)) << 4);
  }
  using Sched = BbSched<KS1, SC, MT, DUAL1>;
  auto step0 = [&](auto v_c) {
    constexpr int v = decltype(v_c)::value;                // slab index
    constexpr int ch = v / SC, sl = v % SC;
    Afr& cur = BB_BUF(v);
    Afr& nxt = BB_BUF(v + 2);
    if constexpr (sl == 0 && ch > 0) {
      // chunk ch landed in every wave and nobody reads buffer (ch + 1) & 1 any more.  This wave's DMAs of chunk ch were issued at
      // the first step of chunk ch - 1, BEHIND that step's fragment loads: younger than them are the fragment loads of that chunk's
      // other SC - 1 steps -- Sched::since_dma(ch) operations; "at most that many outstanding" means every DMA of the chunk has landed
      // (vm_track.h; tests/test_vmcnt_isa.py counts the same in the compiled code).  (Round 4 wrote the bound of two steps as a
      // literal: with SC = 2 it let DMAs fly on -- wrong logits with batches in flight.)
      static_assert(Sched::wait_n(ch) <= Sched::since_dma(ch), "a wait may never allow more than was issued behind the DMAs");
      vm_wait<Sched::wait_n(ch)>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    load_step(nxt, std::integral_constant<int, v + 2>{});
    if constexpr (Sched::low_at(v)) load_a(BB_BUFL(v + 2), a.w1, tms1, KS1, cb_w, v + 2, 2, 1);
    if constexpr (sl == 0 && ch > 0 && ch + 1 < NCH) {
      // (behind this step's fragment loads: the compiler's counted wait for the NEXT step's fragments then still lets these fly)
      asm volatile("" ::: "memory");
      issue_chunk(ch + 1, ((ch + 1) & 1) ? ring1 : ring0);
    }
    const int8_t* B = ((ch & 1) ? ring1 : ring0) + sl * (NP0 * 64);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x4 bf[J0];
#pragma unroll
      for (int j = 0; j < J0; j++) bf[j] = *reinterpret_cast<const i32x4*>(B + (bm0[j] ^ (ks << 5)));
#ifdef TF2_CHECK_DMA
#pragma unroll
      for (int j = 0; j < J0; j++) dma_check(bf[j], g_bband_dma_check);
#endif
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < J0; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
          if constexpr (DUAL1) acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(BB_BUFL(v).k[i][ks], bf[j], acc2[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  bb_static_for<0, KS1>(step0);
  // Horner step of a two-window layer: acc = (acc << dshift[1][row]) [+ the low window's sums]; dshift sits behind rows | lo
  auto window_combine = [&](const int8_t* hdr, int tms, int hst, int nj, bool add_low) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int ch = cb_w + i * 32;
      const int mt = ch >> tms, ro = ch & ((1 << tms) - 1);
      const int* dsh = reinterpret_cast<const int*>(hdr + mt * hst) + (6 << tms) + ro + 4 * half;       // dshift[1]
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < (J0 > J1 ? J0 : J1); j++)
            if (j < nj) {
              unsigned vv = (unsigned)acc[i][j][G * 4 + r] << (d[r] & 31);
              if constexpr (DUAL1) { if (add_low) vv += (unsigned)acc2[i][j < J0 ? j : 0][G * 4 + r]; }
              acc[i][j][G * 4 + r] = (int)vv;
            }
      }
    }
  };
  if constexpr (DUAL1) window_combine(hdr1, tms1, hst1, J0, true);
  BB_STAMP(2);

  // hand-over: requantise (pe.cl:185-203, relu.cl:54) into the halo tile; pixels of rows outside the image keep the pad value
  {
    const int lo_b = a.relu1 ? 0 : -128;
    auto to_mid1 = [&](auto fast_c) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const int ch = cb_w + i * 32;
        const int mt = ch >> tms1, ro = ch & ((1 << tms1) - 1);
        const int* prm = reinterpret_cast<const int*>(hdr1 + mt * hst1);
        const int chl = ch + 16 * half;
        // (the sixteen row-parameter reads once for the J0 column tiles, requant_epilogue.h RqRows: re-read per tile they are 16 KiB of
        //  LDS return traffic per tile and wave, more than the tile's VALU work)
        i32x4 outs[J0];
        int a16s[J0][16];
#pragma unroll
        for (int j = 0; j < J0; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) a16s[j][r] = acc[i][j][r];
        requant_tiles16_rows<J0, FAST>([&](int j) -> const int (&)[16] { return a16s[j]; }, outs, prm, 1 << tms1, ro + 4 * half, lo_b,
                                       a.dbl1 != 0, a.fast1 == 2);
#pragma unroll
        for (int j = 0; j < J0; j++) {
          const i32x4 out = outs[j];
          const int p = (wn + j * WN) * 32 + (lane & 31);
          const int hr = p / W, col = p - hr * W;
          const int row = r0 - 1 + hr;
          if (p < n_p0 && (unsigned)row < (unsigned)H) {
            const int h = hr * Wp + col + 1;
            const int c = (chl & 63) >> 4;
            *reinterpret_cast<i32x4*>(mid1 + (chl >> 6) * slabb + c * planeb + h * 16) = out;
            if (a.keep_mid && hr >= 1 && hr <= rows)
              *reinterpret_cast<i32x4*>(a.mid1 + (size_t)(pix0 + p) * M + chl) = out;
          }
        }
      }
    };
    if (a.fast1 == 1) to_mid1(std::true_type{}); else to_mid1(std::false_type{});
  }
  zero_acc(J1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // the halo tile is complete
  asm volatile("" ::: "memory");
  BB_STAMP(3);

  // ---- phase 1: the 3x3 over the halo tile; step e = (tap t, slab s) -----------------------------------------------------------
  int h0[J1];
#pragma unroll
  for (int j = 0; j < J1; j++) {
    int p = (wn + j * WN) * 32 + (lane & 31);
    if (p >= n_px) p = 0;                                 // lanes beyond the band compute on pixel 0 and are never stored
    const int r = p / W;
    h0[j] = (r * Wp + (p - r * W)) * 16 + half * planeb;  // byte offset of the lane's fragment of tap (0, 0), K half 0
  }
  auto step1 = [&](auto e_c) {
    constexpr int ew = decltype(e_c)::value;               // (window, tap, slab)
    constexpr int v = KS1 + ew;                            // global step index: the fragment buffers keep rotating
    constexpr int e = ew % NE, t = e / KS2, s = e % KS2;
    Afr& cur = BB_BUF(v);
    Afr& nxt = BB_BUF(v + 2);
    load_step(nxt, std::integral_constant<int, v + 2>{});
    if constexpr (DUAL2 && ew == NE) window_combine(hdr2, tms2, hst2, J1, false);        // between the windows
    const int8_t* B = mid1 + s * slabb;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      i32x4 bf[J1];
#pragma unroll
      for (int j = 0; j < J1; j++)
        bf[j] = *reinterpret_cast<const i32x4*>(B + h0[j] + (((t / 3) * Wp + t % 3) * 16 + 2 * ks * planeb));
#pragma unroll
      for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < J1; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  bb_static_for<0, N1>(step1);
  BB_STAMP(4);

  // residual tiles of pass q (16 contiguous NHWC bytes per lane and column tile), loaded one pass ahead
  auto load_res = [&](i32x4 (&rv)[MT][J1], int q) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < J1; j++) {
        const int p = (wn + j * WN) * 32 + (lane & 31);
        const int chl = q * M + cb_w + i * 32 + 16 * half;
        const bool ok = a.has_res && p < n_px;
        const int8_t* rp = ok ? a.res + (size_t)(pix_base + p) * a.res_cp + a.res_off + chl : a.zero;
        rv[i][j] = *reinterpret_cast<const i32x4*>(rp);
      }
  };
  // (two buffers, loaded one pass ahead, where the register budget allows; else one, loaded at the start of its pass)
  constexpr bool RESDB = NW != 16;
  i32x4 res0[MT][J1], res1[MT][J1];
  if constexpr (RESDB) load_res(res0, 0);

  // hand-over: requantise the 3x3 into the expand's B tile [slab][pixel][64]
  {
    const int lo_b = a.relu2 ? 0 : -128;
    auto to_mid2 = [&](auto fast_c) __attribute__((always_inline)) {
      constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const int ch = cb_w + i * 32;
        const int mt = ch >> tms2, ro = ch & ((1 << tms2) - 1);
        const int* prm = reinterpret_cast<const int*>(hdr2 + mt * hst2);
        const int chl = ch + 16 * half;
        i32x4 outs[J1];                                      // (as in the first hand-over: requantise, then write)
        int a16s[J1][16];
#pragma unroll
        for (int j = 0; j < J1; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) a16s[j][r] = acc[i][j][r];
        requant_tiles16_rows<J1, FAST>([&](int j) -> const int (&)[16] { return a16s[j]; }, outs, prm, 1 << tms2, ro + 4 * half, lo_b,
                                       a.dbl2 != 0, a.fast2 == 2);
#pragma unroll
        for (int j = 0; j < J1; j++) {
          const i32x4 out = outs[j];
          const int row = (wn + j * WN) * 32 + (lane & 31);
          const int c = (chl & 63) >> 4;
          *reinterpret_cast<i32x4*>(mid2 + (chl >> 6) * (NP1 * 64) + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = out;
          if (a.keep_mid && row < n_px)
            *reinterpret_cast<i32x4*>(a.mid2 + (size_t)(pix_base + row) * M + chl) = out;
        }
      }
    };
    if (a.fast2 == 1) to_mid2(std::true_type{}); else to_mid2(std::false_type{});
  }
  zero_acc(J1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // the expand's B tile is complete (every wave wrote its channels)
  asm volatile("" ::: "memory");
  BB_STAMP(5);

  // ---- phase 2: C / M passes of the 1x1 expand over the B tile ------------------------------------------------------------------
  int bm1[J1];
#pragma unroll
  for (int j = 0; j < J1; j++) {
    const int row = (wn + j * WN) * 32 + (lane & 31);
    bm1[j] = row * 64 + ((half ^ ((row >> 2) & 3)) << 4);
  }
  const int lo_b3 = a.relu3 ? 0 : -128;
  const int rlo = a.add_relu ? 0 : -128;
  constexpr int NPASS = C / M;
  static_assert(NPASS % 2 == 0, "passes run in pairs");
  // The passes run in PAIRS inside a run-time loop (the body is two passes, statically unrolled: the residual buffers alternate and
  // the fragment buffers rotate the same way in every iteration): a quarter of the code of four unrolled passes.  An identity
  // bottleneck always has its residual (Net::bband_at).
#pragma unroll 1
  for (int qq = 0; qq < NPASS; qq += 2) {
    auto step2 = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;                // step inside the pair
      constexpr int v = KS1 + N1 + u;
      constexpr int ql = u / KS2, s = u % KS2;
      Afr& cur = BB_BUF(v);
      Afr& nxt = BB_BUF(v + 2);
      {
        // fragments of step u + 2 (the next pair's first two steps at the end: loaded past the last pass they read valid memory of
        // pass NPASS - 1 again and are never used)
        constexpr int u2 = (u + 2) % (2 * KS2);
        const int q2 = qq + (u + 2) / KS2;
        load_a(nxt, a.w3, tms3, KS2, (q2 < NPASS ? q2 : NPASS - 1) * M + cb_w, u2 % KS2);
      }
      if constexpr (s == 0) {
        if constexpr (RESDB) {
          const int qn = qq + ql + 1 < NPASS ? qq + ql + 1 : NPASS - 1;
          if constexpr (ql & 1) load_res(res0, qn); else load_res(res1, qn);
        } else {
          load_res(res0, qq + ql);
        }
      }
      const int8_t* B = mid2 + s * (NP1 * 64);
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        i32x4 bf[J1];
#pragma unroll
        for (int j = 0; j < J1; j++) bf[j] = *reinterpret_cast<const i32x4*>(B + (bm1[j] ^ (ks << 5)));
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
          for (int j = 0; j < J1; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
      }
      if constexpr (s == KS2 - 1) {
        i32x4 (&rv)[MT][J1] = (RESDB && (ql & 1)) ? res1 : res0;
        const int q = qq + ql;
        auto epilogue = [&](auto fast_c) __attribute__((always_inline)) {
          constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
          for (int i = 0; i < MT; i++) {
            const int ch = q * M + cb_w + i * 32;
            const int mt = ch >> tms3, ro = ch & ((1 << tms3) - 1);
            const int* prm = reinterpret_cast<const int*>(hdr3 + mt * hst3);
            const int chl = ch + 16 * half;
#pragma unroll
            for (int j = 0; j < J1; j++) {
              int a16[16];
#pragma unroll
              for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
              const i32x4 out = requant_tile16<true, LEAN, FAST, true>(a16, prm, 1 << tms3, ro + 4 * half, lo_b3, rlo, rv[i][j], false, a.fast3 == 2);      // (RNN: Net::bband_at admits only such expands)
              const int p = (wn + j * WN) * 32 + (lane & 31);
              if (p < n_px) *reinterpret_cast<i32x4*>(a.y + (size_t)(pix_base + p) * a.y_cp + a.y_off + chl) = out;
            }
          }
        };
        if (a.fast3 == 1) epilogue(std::true_type{}); else epilogue(std::false_type{});
        zero_acc(J1);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    bb_static_for<0, 2 * KS2>(step2);
  }
  BB_STAMP(6);
#undef BB_STAMP
#undef BB_BUF
#undef BB_BUFL
}

// dynamic LDS of a launch: the halo tile + the three layers' header images (the chunk buffers are static)
static size_t bband_dyn_lds(int M, int R, int W, bool dual1, bool dual2) {
  const int n_h = (R + 2) * (W + 2);
  return (size_t)(M / 64) * (((n_h + 63) >> 6) * 4096) + (size_t)((dual1 ? 28 : 20) + (dual2 ? 28 : 20) + 4 * 20) * M + 64;
}

template <int M, int NW, int WN, int NT0, int NT1, int SC, bool DUAL1, bool DUAL2>
static int launch_bband2(const BBandArgs& a, hipStream_t s) {
  constexpr int C = 4 * M;
  constexpr int CHUNK = SC * 32 * NT0 * 64, MID2 = (M / 64) * 32 * NT1 * 64;
  if (a.R != bband_rows_of(M, NT0)) return 1;
  const size_t dyn = bband_dyn_lds(M, a.R, a.W, DUAL1, DUAL2);
  const size_t stat = (size_t)(CHUNK > MID2 ? CHUNK : MID2) + CHUNK;
  if (dyn + stat > 160 * 1024) return 1;
  auto fn = conv_bband_kernel<M, NW, WN, NT0, NT1, SC, DUAL1, DUAL2>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn), 160 * 1024 - (int)stat)) return -1;
  TF2_LAUNCH_NAME("conv_bband_kernel<%dx%d,C%d,M%d,R%d%s%s> (%d bands per image)", a.H, a.W, C, M, a.R,
                  DUAL1 ? ",dual reduce" : "", DUAL2 ? ",dual 3x3" : "", a.tiles_per_img);
  TF2_LAUNCH(fn, dim3(a.B * a.tiles_per_img), dim3(NW * 64), dyn, s, a);
  return launch_ok() ? 0 : -1;
}

// Shapes instantiated: ResNet-50 stage 4 (14 x 14, C = 1024, M = 256: every row single-window with the shipped Q) and stage 3
// (28 x 28, C = 512, M = 128: two-window reduce, the last bottleneck's 3x3 two-window as well), R rows per band -> column tiles of the
// halo band / the band.  16 waves per block measured SLOWER than 8 (profiles/r04_bband_timeline_r7_w16.txt: the requantisation phases
// are bound by the SIMDs' VALU throughput, not by issue latency, and the 3x3 loop loses: 40 against 34 us per block).
bool conv_bband_shape_ok(int H, int W, int C, int M, int R) {
  if (C != 4 * M || H != W || R < 1 || R > H) return false;
  if (M == 256 && W == 14) return R == 7 || R == 4 || R == 2;
  if (M == 128 && W == 28) return R == 7 || R == 4;
  return false;
}
bool conv_bband_windows_ok(int M, int dual1, int dual2) {
  if (M == 256) return !dual1 && !dual2;
  if (M == 128) return dual1 || !dual2;                   // (single, single), (dual, single), (dual, dual)
  return false;
}

// rows per band actually used for `wanted`.  (Rounds 4-5: the 7-row form of the 28 x 28 kernel with BOTH reduce and 3x3 two-window spilled 23
// registers and that bottleneck took 4-row bands; since the hand-overs read their rows two ahead (LEAN) it compiles to 240 registers without
// scratch -- tests/test_vmcnt_isa.py lists every instantiation's private segment -- and takes 7-row bands like its neighbours: 128 blocks
// instead of 224, the reduce recomputed for 2 of 9 rows instead of 2 of 6.)
// (rows_dd: the test-only option bband_rows_dd keeps the round-5 choice for A/B runs)
int conv_bband_pick_rows(int W, int M, int dual1, int dual2, int wanted, int rows_dd) {
  if (M == 128 && W == 28 && dual1 && dual2 && wanted > rows_dd) return rows_dd;
  return wanted;
}

template <int M, int NW, int WN, int NT0, int NT1, int SC>
static int launch_bband(const BBandArgs& a, hipStream_t s) {
  if constexpr (M == 256) return launch_bband2<M, NW, WN, NT0, NT1, SC, false, false>(a, s);
  else {
    if (a.dual1 && a.dual2) return launch_bband2<M, NW, WN, NT0, NT1, SC, true, true>(a, s);
    if (a.dual1) return launch_bband2<M, NW, WN, NT0, NT1, SC, true, false>(a, s);
    return launch_bband2<M, NW, WN, NT0, NT1, SC, false, false>(a, s);
  }
}

int launch_conv_bband(const BBandArgs& a, int C, int M, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!conv_bband_shape_ok(a.H, a.W, C, M, a.R) || !conv_bband_windows_ok(M, a.dual1, a.dual2)) return 1;
  if (a.H != a.W || a.W != (M == 256 ? 14 : M == 128 ? 28 : 56)) return 1;        // (the kernel's compile-time map side)
#ifdef TF2_PROBES
  // (timing probe: the 7-row bands with one column tile fewer in the 3x3 / expand phases -- the pixels behind 32 * NT1 are then NOT computed)
  if (a.probe == 1) {
    if (M == 256 && a.W == 14 && a.R == 7) return launch_bband<256, 8, 1, 4, 3, 4>(a, s);
    if (M == 128 && a.W == 28 && a.R == 7) return launch_bband<128, 8, 2, 8, 6, 2>(a, s);
  }
#endif
  if (M == 256 && a.W == 14) {
    if (a.R == 7) return launch_bband<256, 8, 1, 4, 4, 4>(a, s);      // 126 / 98 pixels
    if (a.R == 4) return launch_bband<256, 8, 1, 3, 2, 4>(a, s);      // 84 / 56
    if (a.R == 2) return launch_bband<256, 8, 1, 2, 1, 4>(a, s);      // 56 / 28
  }
  if (M == 128 && a.W == 28) {
    if (a.R == 7) return launch_bband<128, 8, 2, 8, 8, 2>(a, s);      // 252 / 196 pixels
    if (a.R == 4) return launch_bband<128, 8, 2, 6, 4, 2>(a, s);      // 168 / 112
  }
  return 1;
}

}  // namespace tf2
