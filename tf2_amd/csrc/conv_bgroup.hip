// conv_bgroup.hip -- a whole identity bottleneck of the small maps (1x1 reduce C -> M, 3x3 / stride 1 / pad 1 M -> M, 1x1 expand
// M -> C + residual + ReLU; ResNet-50 stage 4: C = 1024, M = 256, 14 x 14) in ONE launch, by groups of eight blocks per image
// (gfx950).
//
// Why: one batch at a time the three launches of such a bottleneck take 10.7 + 15.5 + 10.0 us for 2.3 us of matrix-pipe work --
// ~3 us of every launch lie outside any block (dispatch, drain, boundary), ~1.4 us of every block is its latency-bound prologue,
// and 98-392 blocks leave most CUs idle most of the time (tools/block_timeline.py, profiles/r03_experiments.txt item 21).  The
// layers cannot be fused per pixel tile (the 3x3 needs its neighbours, the reduce all input channels), but they can per IMAGE:
// a 14 x 14 map is 196 pixels, its M-channel intermediates are 50 KB.
//
// Shape: the eight blocks with the same b % 8 inside a span of 64 blocks (they share an XCD) form an image's group; member m owns 1/8 of every layer's OUTPUT CHANNELS (M / 8 = 32 of the
// reduce and of the 3x3, C / 8 = 128 of the expand) for all 196 pixels: seven 32-pixel MFMA column tiles, one per wave (wave 7
// only helps loading).  The intermediates go through the workspace tensors the three separate launches would use (mid1, mid2):
// written through (sc1 stores), read from the XCD's L2 when the whole group sits on one XCD (the normal placement) and from the
// memory side otherwise, and between the layers the eight members meet at eight epoch-tagged flag words per image and layer
// (bg_signal / bg_wait below; blocks are dispatched in ascending order, so the members of a group become resident together;
// a stuck wait traps instead of hanging).  No ring shared between waves and no
// block barrier inside a K loop:
//   reduce : every wave streams its own 32 pixels x 64 channels and the member's 32 weight rows through a private 3-stage
//            LDS-DMA ring (counted vmcnt waits only);
//   3x3    : the image's intermediate (16 x 16 halo grid x M channels, 64 KB) is loaded into LDS once, a tap is a shifted
//            address (conv_bneck's scheme), weights go global -> registers one step ahead;
//   expand : every wave keeps its 32 pixels x M channels (8 KB) in LDS, sweeps the member's 128 output channels two 32-row tiles
//            at a time, weights global -> registers, residual (the bottleneck's input) by ordinary loads.
// Weight tiles, header rows and the requantisation are the packed image's and requant_epilogue.h's: bit-identical to the three
// launches (tests/test_gpu_parity.py).
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

template <int T, int N, class F>
__device__ __forceinline__ void bg_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); bg_static_for<T + 1, N>(fn); }
}

constexpr int kBgMembers = 8;
constexpr int kBgHdrSlot = 2048;             // LDS bytes reserved per header image (rows | lo of one m-tile)

template <int N>
__device__ __forceinline__ void bg_wait_vmcnt() {
  static_assert(N == 0 || N == 2 || N == 4 || N == 6, "prepared immediates");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

// The eight members of an image meet at eight flag words.  A flag is (epoch << 8) | XCC id of the member: the epoch is a word of the
// workspace that the step's first kernel increments, so flags need no zeroing and a stale copy of a flag line (an older
// epoch) is never mistaken for a set flag.  Normally the blocks of a launch that fits the chip go to XCD (block % 8), so the
// members of an image -- the blocks with the same b % 8 inside a span of 64 -- share an XCD and its L2: then the exchange
// needs no memory-side round trip (writers' stores reach the L2 through their write-through L1s, readers use sc0 accesses
// after dropping their own L1's lines).  That placement is not guaranteed (tools/bgroup_stress.py saw exceptions), so every
// store of the exchange is ALSO written through to memory (sc1), every fourth poll reads the memory side, and a reader that
// finds a member with another XCC id reads the group's data from the memory side (sc1) instead of its L2.
// Two halves, so that loads for the next phase can be issued between them:
// signal -- every store of every wave acknowledged, then this member's flag; wait -- until all eight flags carry the epoch.
__device__ __forceinline__ void bg_signal(unsigned* flags, int member, unsigned tag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(flags + member), "v"(tag) : "memory");
}
// returns (block-uniform) whether every member runs on this block's XCD
__device__ __forceinline__ bool bg_wait(const unsigned* flags, unsigned tag, int tid, int* lds_word) {
  if (tid < 64) {
    int polls = 0;
    unsigned f = 0;
    for (;;) {
      if (tid < kBgMembers) {
        if ((polls & 3) == 3) {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + tid) : "memory");
        } else {
          // an sc0 load may hit this CU's L1, where the line sits from the previous poll: drop the L1's lines first
          asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + tid) : "memory");
        }
      } else {
        f = tag;
      }
      if (__builtin_amdgcn_ballot_w64((f >> 8) != (tag >> 8)) == 0) break;
      __builtin_amdgcn_s_sleep(1);
      if (++polls > (1 << 24)) __builtin_trap();     // members are dispatched together (ascending order): fail, do not hang
    }
    const bool all_here = __builtin_amdgcn_ballot_w64(tid < kBgMembers && (f & 0xff) != (tag & 0xff)) == 0;
    if (tid == 0) *lds_word = all_here ? 1 : 0;
  }
  __syncthreads();
  return *lds_word != 0;
}

// HW: map side; C: channels of the bottleneck's input / output; M: channels of the intermediates (M / 8 = 32 per member)
template <int HW, int C, int M>
__global__ __launch_bounds__(512, 2) void conv_bgroup_kernel(BGroupArgs a) {
  static_assert(M == 32 * kBgMembers && C % (32 * kBgMembers) == 0 && HW <= 14, "member slices are whole 32-row MFMA tiles; the halo grid is 16 x 16");
  constexpr int NPX = HW * HW;
  constexpr int NT = (NPX + 31) / 32;                    // 32-pixel column tiles = working waves
  static_assert(NT <= 7, "one column tile per wave, wave 7 only loads");
  constexpr int KS1 = C / 64, KS2 = M / 64;              // 64-byte channel slabs of the input / of the intermediates
  constexpr int CT = C / kBgMembers / 32;                // 32-row tiles of the expand per member
  constexpr int NE = 9 * KS2;                            // (tap, slab) steps of the 3x3
  constexpr int S = 5, STAGE = 2048;                     // reduce: private ring of 32 pixels x 64 bytes per stage (in the W region, idle until phase B)
  constexpr int HALO = 256 * 64;                         // one 64-channel slab of the 16 x 16 halo grid
  // LDS map: [4 header slots][W: the 3x3's weights, later the expand's][R: reduce weights + pixel rings | halo | expand tiles]
  constexpr int W_BYTES = NE * 2048, R_BYTES = KS2 * HALO;
  static_assert(W_BYTES >= CT * KS2 * 2048 && W_BYTES >= NT * S * STAGE && R_BYTES >= KS1 * 2048 && R_BYTES >= NT * KS2 * 2048, "phase regions");
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  int8_t* const hdr_lds = lds;
  int8_t* const wreg = lds + 4 * kBgHdrSlot;
  int8_t* const work = wreg + W_BYTES;
  int* const ctl = reinterpret_cast<int*>(work + R_BYTES);      // [0] epoch, [1] / [2]: "group on one XCD" of the two meetings

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  // block b -> XCD b % 8; image = the XCD's (b / 64)-th, member = (b / 8) % 8: a group sits on one XCD
  const int img = a.img0 + ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 6), m = ((int)blockIdx.x >> 3) & 7;
  if (img >= a.B) return;
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);      // LDS-DMA: lane l fills row l >> 2, slot l & 3, which holds chunk slot ^ ((row >> 2) & 3)
  const int drow = lane >> 2;
  const size_t px_img = (size_t)img * NPX;
  unsigned* const ctr = a.ctr + (size_t)img * 16;         // two rows of eight flags

  long long* const dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bgroup_timeline.py: 100 MHz wall clock per phase
#define BG_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  BG_STAMP(0);

  // member's rows inside the packed tiles (TM = 64 or 128 rows per m-tile; dense layers: entry = m-tile * nslab + slab)
  const int c1 = 32 * m;                                 // first output channel of this member in reduce / 3x3
  const int mt1 = c1 / a.tm1, ro1 = c1 % a.tm1;
  const int mt2 = c1 / a.tm2, ro2 = c1 % a.tm2;
  const int c3 = (C / kBgMembers) * m;                   // ... in the expand

  // 32 weight rows x 64 bytes of one (m-tile, entry) -> LDS slab (two 16-row pieces), by the calling wave
  auto w_dma = [&](const int8_t* w, int tm, int entry, int ro, int8_t* dst) {
#pragma unroll
    for (int g2 = 0; g2 < 2; g2++)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(w + ((size_t)entry * tm + ro + 16 * g2 + drow) * 64 + chunk * 16), TF2_LDS_PTR(dst + g2 * 1024), 16, 0, 0);
  };

  // ---- kernel start: headers (rows {bias | dbl, alpha, addend64} and lo of the member's m-tiles), the reduce's weights ----
  {
    auto hdr_dma = [&](const int32_t* hdr, int hdr_bytes, int mt, int tm, int slot) {
      const int used = ((kPrmWordsPerRow * tm * 4) + 1023) & ~1023;
      const int8_t* src = reinterpret_cast<const int8_t*>(hdr) + (size_t)mt * hdr_bytes + lane * 16;
      for (int i = wave; i * 1024 < used; i += 8)
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src + i * 1024), TF2_LDS_PTR(hdr_lds + slot * kBgHdrSlot + i * 1024), 16, 0, 0);
    };
    hdr_dma(a.hdr1, a.hdr1_bytes, mt1, a.tm1, 0);
    hdr_dma(a.hdr2, a.hdr2_bytes, mt2, a.tm2, 1);
    hdr_dma(a.hdr3, a.hdr3_bytes, c3 / a.tm3, a.tm3, 2);
    if (a.tm3 < C / kBgMembers) hdr_dma(a.hdr3, a.hdr3_bytes, c3 / a.tm3 + 1, a.tm3, 3);
    for (int s = wave; s < KS1; s += 8) w_dma(a.w1, a.tm1, mt1 * KS1 + s, ro1, work + s * 2048);
    if (tid == 64 * 7) {                                   // the step counter, from the memory side (wave 7 has no tile of its own)
      unsigned e;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(a.epoch) : "memory");
      ctl[0] = (int)e;
    }
  }
  const int* const prm1 = reinterpret_cast<const int*>(hdr_lds);
  const int* const prm2 = reinterpret_cast<const int*>(hdr_lds + kBgHdrSlot);

  const int t = wave;                                    // this wave's column tile
  const bool worker = wave < NT;
  const int p_lane = 32 * t + (lane & 31);               // the pixel of this lane's MFMA column
  const bool p_ok = worker && p_lane < NPX;
  const i32x4 nores = {0, 0, 0, 0};
  // fragment address inside a [32 rows][64 bytes] swizzled tile: row = lane & 31, chunk c = 2 * ks + half
  const int frow = lane & 31;
  const int fr0 = frow * 64 + (((0 + half) ^ ((frow >> 2) & 3)) << 4);      // ks = 0; ks = 1 is the same address ^ 32

  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID, 4 bits
  unsigned tag = 0;                                         // this member's flag value: (epoch << 8) | XCC id

  // =================================== phase A: reduce, 1x1 C -> M ===================================
  {
    int8_t* const ring = wreg + wave * (S * STAGE);
    auto issue = [&](int s, int slot) {
#pragma unroll
      for (int g2 = 0; g2 < 2; g2++) {
        const int p = 32 * t + 16 * g2 + drow;
        const int8_t* src = p < NPX ? a.x + (px_img + p) * C + s * 64 + chunk * 16 : a.zero + chunk * 16;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(ring + slot * STAGE + g2 * 1024), 16, 0, 0);
      }
    };
    if (worker) {
#pragma unroll
      for (int s = 0; s < S - 1; s++) issue(s, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // headers and the reduce's weights are in LDS (pieces fetched by every wave)
    BG_STAMP(1);
    tag = ((unsigned)ctl[0] << 8) | (xcc & 0xff);          // (the epoch word was stored before the barrier)
    // two accumulators (one per K half): a dependent MFMA would wait out the 16 passes of its predecessor
    i32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0; acc1[r] = 0; }
    if (worker) {
      int cs = 0, is = S - 1;
      for (int s = 0; s < KS1; s++) {
        // stages 0 .. S-2 landed above; later: S-2 younger stages (2 DMAs each) may fly while they have been issued
        if (s >= S - 1) { if (s + S - 2 < KS1) bg_wait_vmcnt<2 * (S - 2)>(); else bg_wait_vmcnt<0>(); }
        const int8_t* A = work + s * 2048;
        const int8_t* B = ring + cs * STAGE;
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + fr0), a1 = *reinterpret_cast<const i32x4*>(A + (fr0 ^ 32));
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + fr0), b1 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
        if (s + S - 1 < KS1) { issue(s + S - 1, is); is = is + 1 == S ? 0 : is + 1; }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc1, 0, 0, 0);
        cs = cs + 1 == S ? 0 : cs + 1;
      }
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] += acc1[r];
      BG_STAMP(2);
      // requantise, write this member's 32 channels of mid1 (agent scope: the other members read them next)
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
      const int lo_b = a.relu1 ? 0 : -128;
      i32x4 out;
      if (a.fast1 == 1) out = requant_tile16<false, 0, true>(a16, prm1, a.tm1, ro1 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, false);
      else out = requant_tile16<false, 0, false>(a16, prm1, a.tm1, ro1 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, a.fast1 == 2);
      if (p_ok) {
        int8_t* dst = a.mid1 + (px_img + p_lane) * M + c1 + 16 * half;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(out) : "memory");
      }
    }
  }
  bg_signal(ctr, m, tag, tid);
  BG_STAMP(3);
  // the 3x3's weights (this member's 32 rows of all 9 x KS2 steps) on their way while the group gathers
  for (int e = wave; e < NE; e += 8) w_dma(a.w2, a.tm2, mt2 * NE + e, ro2, wreg + e * 2048);
  const bool local1 = bg_wait(ctr, tag, tid, ctl + 1);
  BG_STAMP(4);

  // =================================== phase B: 3x3 / pad 1, M -> M ===================================
  {
    int8_t* const halo = work;                             // [KS2][256 halo pixels][64], halo (r, c) = pixel (r - 1, c - 1)
    for (int gi = wave; gi < KS2 * 16; gi += 8) {
      const int s = gi >> 4, grp = gi & 15;
      const int h = grp * 16 + drow;
      const int row = (h >> 4) - 1, col = (h & 15) - 1;
      const bool ok = (unsigned)row < (unsigned)HW && (unsigned)col < (unsigned)HW;
      const int8_t* src = ok ? a.mid1 + (px_img + row * HW + col) * M + s * 64 + chunk * 16
                             : a.zero2 + s * 64 + chunk * 16;       // the 3x3's pad row: the stored form of x = 0
      if (local1) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + s * HALO + grp * 1024), 16, 0, 1);       // this XCD's L2
      else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + s * HALO + grp * 1024), 16, 0, 16);            // memory side
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // halo and weights complete in every wave
    BG_STAMP(5);
    if (worker) {
      const int pq = p_ok ? p_lane : 0;                    // lanes beyond the map compute on pixel 0 and are never stored
      const int oh = pq / HW, ow = pq - oh * HW;
      const int h0 = oh * 16 + ow;
      i32x16 acc, acc1;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[r] = 0; acc1[r] = 0; }
      auto step = [&](auto e_c) {
        constexpr int e = decltype(e_c)::value;
        constexpr int tap = e / KS2, s = e % KS2;
        const int8_t* A = wreg + e * 2048;
        const int h = h0 + (tap / 3) * 16 + tap % 3;
        const int ba = s * HALO + h * 64 + ((half ^ ((h >> 2) & 3)) << 4);
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + fr0), a1 = *reinterpret_cast<const i32x4*>(A + (fr0 ^ 32));
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(halo + ba), b1 = *reinterpret_cast<const i32x4*>(halo + (ba ^ 32));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc1, 0, 0, 0);
      };
      bg_static_for<0, NE>(step);
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] += acc1[r];
      BG_STAMP(6);
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
      const int lo_b = a.relu2 ? 0 : -128;
      i32x4 out;
      if (a.fast2 == 1) out = requant_tile16<false, 0, true>(a16, prm2, a.tm2, ro2 + 4 * half, lo_b, -128, nores, a.dbl2 != 0, false);
      else out = requant_tile16<false, 0, false>(a16, prm2, a.tm2, ro2 + 4 * half, lo_b, -128, nores, a.dbl2 != 0, a.fast2 == 2);
      if (p_ok) {
        int8_t* dst = a.mid2 + (px_img + p_lane) * M + c1 + 16 * half;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(out) : "memory");
      }
    }
  }
  bg_signal(ctr + 8, m, tag, tid);
  BG_STAMP(7);
  // the expand's weights: [32-row tile][slab] of this member's C / 8 rows, into the 3x3's weight region
  for (int u = wave; u < CT * KS2; u += 8) {
    const int ch = c3 + 32 * (u / KS2);
    w_dma(a.w3, a.tm3, (ch / a.tm3) * KS2 + u % KS2, ch % a.tm3, wreg + u * 2048);
  }
  // residual tiles (the bottleneck's input, written before this launch: ordinary loads), all CT of them
  i32x4 rv[CT];
#pragma unroll
  for (int q = 0; q < CT; q++) {
    const int8_t* rp = (a.has_res && p_ok) ? a.res + (px_img + p_lane) * a.res_cp + a.res_off + c3 + 32 * q + 16 * half : a.zero;
    rv[q] = *reinterpret_cast<const i32x4*>(rp);
  }
  const bool local2 = bg_wait(ctr + 8, tag, tid, ctl + 2);
  BG_STAMP(8);

  // =================================== phase C: expand, 1x1 M -> C, + residual ===================================
  {
    int8_t* const tile = work + wave * (KS2 * 2048);       // [KS2][32 pixels][64]
    if (worker) {
#pragma unroll
      for (int s = 0; s < KS2; s++)
#pragma unroll
        for (int g2 = 0; g2 < 2; g2++) {
          const int p = 32 * t + 16 * g2 + drow;
          const int8_t* src = p < NPX ? a.mid2 + (px_img + p) * M + s * 64 + chunk * 16 : a.zero + chunk * 16;
          if (local2) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(tile + s * 2048 + g2 * 1024), 16, 0, 1);
          else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(tile + s * 2048 + g2 * 1024), 16, 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // the expand's weights (fetched by every wave) and this wave's pixel tile
    BG_STAMP(9);
    if (worker) {
      const int lo_b = a.relu3 ? 0 : -128, rlo = a.add_relu ? 0 : -128;
      static_assert(CT % 2 == 0, "the expand is swept two 32-row tiles at a time");
#pragma unroll
      for (int pair = 0; pair < CT / 2; pair++) {
        i32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[q][r] = 0;
#pragma unroll
        for (int s = 0; s < KS2; s++) {
          const i32x4 b0 = *reinterpret_cast<const i32x4*>(tile + s * 2048 + fr0), b1 = *reinterpret_cast<const i32x4*>(tile + s * 2048 + (fr0 ^ 32));
          const int8_t* A0 = wreg + ((2 * pair + 0) * KS2 + s) * 2048;
          const int8_t* A1 = wreg + ((2 * pair + 1) * KS2 + s) * 2048;
          const i32x4 a00 = *reinterpret_cast<const i32x4*>(A0 + fr0), a01 = *reinterpret_cast<const i32x4*>(A0 + (fr0 ^ 32));
          const i32x4 a10 = *reinterpret_cast<const i32x4*>(A1 + fr0), a11 = *reinterpret_cast<const i32x4*>(A1 + (fr0 ^ 32));
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a00, b0, acc[0], 0, 0, 0);      // the two tiles alternate: no MFMA waits for its predecessor
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a10, b0, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a01, b1, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a11, b1, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int ch = c3 + 32 * (2 * pair + q);
          const int mt = ch / a.tm3, ro = ch % a.tm3;
          const int* pm = reinterpret_cast<const int*>(hdr_lds + (2 + (mt - c3 / a.tm3)) * kBgHdrSlot);
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[q][r];
          i32x4 out;
          if (a.fast3 == 1) {
            if (a.has_res) out = requant_tile16<true, 0, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, rv[2 * pair + q], false, false);
            else out = requant_tile16<false, 0, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, false);
          } else {
            if (a.has_res) out = requant_tile16<true, 0, false>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, rv[2 * pair + q], false, a.fast3 == 2);
            else out = requant_tile16<false, 0, false>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, a.fast3 == 2);
          }
          if (p_ok) *reinterpret_cast<i32x4*>(a.y + (px_img + p_lane) * a.y_cp + a.y_off + ch + 16 * half) = out;
        }
      }
    }
  }
  BG_STAMP(10);
#undef BG_STAMP
}

size_t conv_bgroup_lds_bytes(int HW, int C, int M) {
  const int NT = (HW * HW + 31) / 32, KS1 = C / 64, KS2 = M / 64;
  (void)NT; (void)KS1;
  return 4 * (size_t)kBgHdrSlot + (size_t)9 * KS2 * 2048 + (size_t)KS2 * 256 * 64 + 64 + 64;     // + the control words behind the work region
}

bool conv_bgroup_shape_ok(int HW, int C, int M) { return HW == 14 && C == 1024 && M == 256; }

int launch_conv_bgroup(const BGroupArgs& a, int HW, int C, int M, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!conv_bgroup_shape_ok(HW, C, M)) return 1;
  const size_t lds = conv_bgroup_lds_bytes(HW, C, M);
  auto fn = conv_bgroup_kernel<14, 1024, 256>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  // at most 32 images = 256 blocks = one block per CU per launch: every group is resident from the start, and the blocks of a
  // launch that fits the chip go to XCD (block % 8) -- larger grids were seen to place late blocks elsewhere (tools/bgroup_stress.py)
  for (int i0 = 0; i0 < a.B; i0 += 32) {
    BGroupArgs b = a;
    b.img0 = i0;
    const int n = std::min(32, a.B - i0);
    TF2_LAUNCH_NAME("conv_bgroup_kernel<%dx%d,C%d,M%d> (8 blocks per image, images %d..%d)", HW, HW, C, M, i0, i0 + n - 1);
    TF2_LAUNCH(fn, dim3(kBgMembers * ((n + 7) / 8 * 8)), dim3(512), lds, s, b);
    if (!launch_ok()) return -1;
  }
  return 0;
}

}  // namespace tf2
