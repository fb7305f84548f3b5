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
// a stuck wait is reported through the workspace's error word and ends the block, bg_report).  No ring shared between waves and no
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
  static_assert(N == 0 || N == 2 || N == 4 || N == 6 || N == 8, "prepared immediates");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
// A meeting that does not complete (a member that never became resident: the launch preconditions of tf2_amd.h were not met) is
// REPORTED, not trapped (round 6; rounds 3-5 ended the HIP context with __builtin_trap()): after `limit` polls the waiting block writes
// kBgErrMagic | code into the workspace's error word (control word 1, written through to memory), gives up and leaves the kernel;
// the other members of its group do the same, every other group completes, the stream goes on, and tf2_net_poll_error returns
// TF2_ERR_GROUP for that step (its logits are garbage).  Control words of the workspace (net.hip / prep_zero_ctrl): [0] step counter,
// [1] error word, [2] poll limit of this step, [3] test-only: 1 + index of a block that leaves its group at kernel entry.
__device__ __forceinline__ void bg_report(const unsigned* epoch, unsigned code) {
  asm volatile("global_store_dword %0, %1, off sc1" :: "v"(const_cast<unsigned*>(epoch) + 1), "v"(kBgErrMagic | (code & 0xffffu)) : "memory");
}
// returns (block-uniform) 1: every member runs on this block's XCD, 0: not, -1: the meeting timed out (reported; the caller returns)
__device__ __forceinline__ int bg_wait(const unsigned* flags, unsigned tag, int tid, int* lds_word, const unsigned* epoch, int limit, unsigned code) {
  if (tid < 64) {
    int polls = 0;
    unsigned f = 0;
    bool failed = false;
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
      if (++polls > limit) { failed = true; break; }       // members are dispatched together (ascending order): report, do not hang
    }
    const bool all_here = __builtin_amdgcn_ballot_w64(tid < kBgMembers && (f & 0xff) != (tag & 0xff)) == 0;
    if (tid == 0) {
      if (failed) bg_report(epoch, code);
      *lds_word = failed ? -1 : all_here ? 1 : 0;
    }
  }
  __syncthreads();
  return *lds_word;
}

// Roll call: every member posts its flag once at the start of the launch; by the time the first exchange stores are due everybody
// has long arrived, and a member that finds the whole group on its own XCD (the normal case) writes its exchange data with
// ordinary stores -- acknowledged by the L2 in a fraction of the time a write-through to the memory side takes.
__device__ __forceinline__ void bg_rollcall_post(unsigned* flags, int member, unsigned tag, int tid) {
  if (tid == 0) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(flags + member), "v"(tag) : "memory");
}
// the calling WAVE reads the roll call (no block barrier): true = all eight members run on this wave's XCD.  A roll call that times
// out is reported and read as "not local": the block's next meeting (a block barrier) times out as well and ends the block.
__device__ __forceinline__ bool bg_rollcall_wave(const unsigned* flags, unsigned tag, int lane, const unsigned* epoch, int limit, unsigned code) {
  int polls = 0;
  unsigned f = tag;
  for (;;) {
    if (lane < kBgMembers) {
      if ((polls & 3) == 3) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + lane) : "memory");
      else asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flags + lane) : "memory");
    }
    if (__builtin_amdgcn_ballot_w64((f >> 8) != (tag >> 8)) == 0) break;
    __builtin_amdgcn_s_sleep(1);
    if (++polls > limit) { if (lane == 0) bg_report(epoch, code); return false; }
  }
  return __builtin_amdgcn_ballot_w64((f & 0xff) != (tag & 0xff)) == 0;
}
// one 16-byte piece of an exchanged tensor
__device__ __forceinline__ void bg_store_x(int8_t* dst, const i32x4& v, bool local) {
  if (local) *reinterpret_cast<i32x4*>(dst) = v;
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
}

// HW: map side; C: channels of the bottleneck's input / output; M: channels of the intermediates (M / 8 = 32 per member)
// Chained: a launch carries up to kBgMaxChain consecutive bottlenecks (the five of ResNet-50's stage 4): the eight blocks of an
// image run them one after the other with a third meeting in between ("this bottleneck's output is complete") instead of a
// launch boundary -- the next one's headers and reduce weights are fetched while the group gathers.
template <int HW, int C, int M>
__global__ __launch_bounds__(512, 2) void conv_bgroup_kernel(BGroupChain c) {
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

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // block b -> XCD b % 8; image = the XCD's (b / 64)-th, member = (b / 8) % 8: a group sits on one XCD
  const int img = c.b[0].img0 + ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 6), m = ((int)blockIdx.x >> 3) & 7;
  if (img >= c.b[0].B) return;
  const size_t px_img = (size_t)img * NPX;
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID, 4 bits
  unsigned tag = 0;                                         // this member's flag value: (epoch << 8) | XCC id
  bool local0 = false;                                      // the whole group on this XCD (roll call): exchange stores need no write-through
  const int t = wave;                                    // this wave's column tile
  const bool worker = wave < NT;
  const i32x4 nores = {0, 0, 0, 0};

#pragma unroll 1
  for (int kb = 0; kb < c.n; kb++) {
  const BGroupArgs& a = c.b[kb];
  // per-lane basics re-derived from an opaque copy of the thread id: otherwise everything that depends only on the lane is hoisted
  // out of this loop and kept in registers across it
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, half = lane >> 5;
  // LDS-DMA: lane l fills row l >> 2, slot l & 3, which holds chunk slot ^ ((row >> 2) & 3)
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
  const int p_lane = 32 * t + (lane & 31);               // the pixel of this lane's MFMA column
  const bool p_ok = worker && p_lane < NPX;
  // fragment address inside a [32 rows][64 bytes] swizzled tile: row = lane & 31, chunk c = 2 * ks + half
  const int frow = lane & 31;
  const int fr0 = frow * 64 + (((0 + half) ^ ((frow >> 2) & 3)) << 4);      // ks = 0; ks = 1 is the same address ^ 32
  unsigned* const ctr = a.ctr + (size_t)img * 32;         // three rows of eight flags: roll call (kb > 0: "input complete"), two meetings

  long long* const dbg = (a.dbg && kb == 0) ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bgroup_timeline.py: 100 MHz wall clock per phase
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
    if (kb == 0 && tid == 64 * 7) {                        // the step counter, from the memory side (wave 7 has no tile of its own)
      i32x4 e;                                             // control words {step counter, error word, poll limit, test: withheld block + 1}
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(a.epoch) : "memory");
      ctl[0] = e[0]; ctl[4] = e[2]; ctl[5] = e[3];
    }
  }
  const int* const prm1 = reinterpret_cast<const int*>(hdr_lds);
  const int* const prm2 = reinterpret_cast<const int*>(hdr_lds + kBgHdrSlot);


  // kb > 0: the input is the previous bottleneck's output, written by all eight members: they meet at this one's roll-call row
  // (signalled behind the previous expand's stores) before the first pixel is fetched
  bool local_in = false;
  if (kb > 0) { const int r_in = bg_wait(ctr, tag, tid, ctl + 3, a.epoch, ctl[4], 0x10u | ((unsigned)kb << 8)); if (r_in < 0) return; local_in = r_in > 0; }

  // =================================== phase A: reduce, 1x1 C -> M ===================================
  {
    int8_t* const ring = wreg + wave * (S * STAGE);
    auto issue = [&](int s, int slot) {
#pragma unroll
      for (int g2 = 0; g2 < 2; g2++) {
        const int p = 32 * t + 16 * g2 + drow;
        const int8_t* src = p < NPX ? a.x + (px_img + p) * C + s * 64 + chunk * 16 : a.zero + chunk * 16;
        int8_t* const dst = ring + slot * STAGE + g2 * 1024;
        if (kb == 0) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 0);                // written before this launch
        else if (local_in) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 1);        // by this group, in this XCD's L2
        else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 16);
      }
    };
    if (worker) {
#pragma unroll
      for (int s = 0; s < S - 1; s++) issue(s, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // headers and the reduce's weights are in LDS (pieces fetched by every wave)
    BG_STAMP(1);
    if (kb == 0) {
      if (ctl[5] == (int)blockIdx.x + 1) return;            // (test-only: this member leaves its group; the others report and go on)
      tag = ((unsigned)ctl[0] << 8) | (xcc & 0xff);        // (the epoch word was stored before the barrier)
      bg_rollcall_post(ctr, m, tag, tid);
    }
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
      if (kb == 0) local0 = bg_rollcall_wave(ctr, tag, lane, a.epoch, ctl[4], 0x01u);
      // requantise, write this member's 32 channels of mid1 (the other members read them next)
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
      const int lo_b = a.relu1 ? 0 : -128;
      i32x4 out;
      if (a.fast1 == 1) out = requant_tile16<false, 0, true>(a16, prm1, a.tm1, ro1 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, false);
      else out = requant_tile16<false, 0, false>(a16, prm1, a.tm1, ro1 + 4 * half, lo_b, -128, nores, a.dbl1 != 0, a.fast1 == 2);
      if (p_ok) {
        int8_t* dst = a.mid1 + (px_img + p_lane) * M + c1 + 16 * half;
        bg_store_x(dst, out, local0);
      }
    }
  }
  bg_signal(ctr + 8, m, tag, tid);
  BG_STAMP(3);
  // the 3x3's weights (this member's 32 rows of all 9 x KS2 steps) on their way while the group gathers
  for (int e = wave; e < NE; e += 8) w_dma(a.w2, a.tm2, mt2 * NE + e, ro2, wreg + e * 2048);
  const int r_m1 = bg_wait(ctr + 8, tag, tid, ctl + 1, a.epoch, ctl[4], 0x20u | ((unsigned)kb << 8)); if (r_m1 < 0) return; const bool local1 = r_m1 > 0;
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
        bg_store_x(dst, out, local0);
      }
    }
  }
  bg_signal(ctr + 16, m, tag, tid);
  BG_STAMP(7);
  // the expand's weights: [32-row tile][slab] of this member's C / 8 rows, into the 3x3's weight region
  for (int u = wave; u < CT * KS2; u += 8) {
    const int ch = c3 + 32 * (u / KS2);
    w_dma(a.w3, a.tm3, (ch / a.tm3) * KS2 + u % KS2, ch % a.tm3, wreg + u * 2048);
  }
  // residual tiles, all CT of them (the bottleneck's input, written before this launch -- or, inside a chain, by THIS thread in the
  // previous bottleneck's expand: ordinary loads either way)
  i32x4 rv[CT];
#pragma unroll
  for (int q = 0; q < CT; q++) {
    const int8_t* rp = (a.has_res && p_ok) ? a.res + (px_img + p_lane) * a.res_cp + a.res_off + c3 + 32 * q + 16 * half : a.zero;
    rv[q] = *reinterpret_cast<const i32x4*>(rp);
  }
  const int r_m2 = bg_wait(ctr + 16, tag, tid, ctl + 2, a.epoch, ctl[4], 0x30u | ((unsigned)kb << 8)); if (r_m2 < 0) return; const bool local2 = r_m2 > 0;
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
            if (a.has_res) out = requant_tile16<true, 0, true, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, rv[2 * pair + q], false, false);
            else out = requant_tile16<false, 0, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, false);
          } else {
            if (a.has_res) out = requant_tile16<true, 0, false, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, rv[2 * pair + q], false, a.fast3 == 2);
            else out = requant_tile16<false, 0, false>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, a.fast3 == 2);
          }
          // (the launch's last output is read by later kernels: ordinary stores; an inner one by the group, like mid1 / mid2)
          if (p_ok) bg_store_x(a.y + (px_img + p_lane) * a.y_cp + a.y_off + ch + 16 * half, out, local0 || kb + 1 == c.n);
        }
      }
    }
  }
  BG_STAMP(10);
#undef BG_STAMP
  // every store of this member acknowledged and every wave done with the LDS regions, then its flag at the next bottleneck's roll call
  if (kb + 1 < c.n) bg_signal(c.b[kb + 1].ctr + (size_t)img * 32, m, tag, tid);
  }
}

// ---- the FIRST bottleneck of the 56 x 56 stage (ResNet-50 rows 1-4: projection shortcut 64 -> 256 | reduce 64 -> 64, 3x3, expand + residual)
// Eight row bands of seven rows per image (one per member), C_in = 64: the band's input pixels (392 x 64 bytes) stay in LDS from the reduce on, and
// the expand's wave computes the SHORTCUT tile it needs as residual itself (K = 64: two MFMAs per window on the resident input
// tile, its own requantisation) -- the shortcut's 25.7 MB map is neither written nor read back (with keep_s it is written, for
// tf2_net_read_layer).  DUAL: reduce, expand and shortcut are two-window layers (ResNet-50), else all single.
template <bool DUAL>
__global__ __launch_bounds__(512, 2) void conv_bgroup56f_kernel(BGroupArgs a) {
  constexpr int HW = 56, CIN = 64, M = 64, PR = 7;
  constexpr int NPX = HW * HW, NPB = PR * HW;
  constexpr int NT = (NPB + 31) / 32;                    // 13
  constexpr int NWN = DUAL ? 2 : 1;
  constexpr int HC = 64, HALO = (PR + 2) * HC * 64;      // 36 KB
  constexpr int kHdrSlots = 10;                          // reduce, 3x3, four m-tiles of the expand, four of the shortcut
  constexpr int W_BYTES = 36 * 1024;                     // reduce weights (<= 16 KB), then the 3x3's (36 KB)
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  int8_t* const hdr_lds = lds;
  int8_t* const wreg = lds + kHdrSlots * kBgHdrSlot;
  int8_t* const halo = wreg + W_BYTES;                   // 36 KB
  int8_t* const tiles = halo + HALO;                     // 26 KB: the band's 3x3 output, B operand of the expand
  int8_t* const xt = tiles + NT * 2048;                  // 26 KB: the band's input, B operand of reduce and shortcut
  int* const ctl = reinterpret_cast<int*>(xt + NT * 2048);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int img = a.img0 + ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 6), m = ((int)blockIdx.x >> 3) & 7;
  if (img >= a.B) return;
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const int drow = lane >> 2;
  const size_t px_img = (size_t)img * NPX;
  const size_t px_band = px_img + (size_t)m * NPB;
  unsigned* const ctr = a.ctr + (size_t)img * 32;
  const int frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);
  const i32x4 nores = {0, 0, 0, 0};

  auto w_dma = [&](const int8_t* w, size_t row0, int8_t* dst) {
#pragma unroll
    for (int g2 = 0; g2 < 2; g2++)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(w + (row0 + 16 * g2 + drow) * 64 + chunk * 16), TF2_LDS_PTR(dst + g2 * 1024), 16, 0, 0);
  };
  {
    auto hdr_dma = [&](const int32_t* hdr, int hdr_bytes, int mt, int slot) {
      const int8_t* src = reinterpret_cast<const int8_t*>(hdr) + (size_t)mt * hdr_bytes + lane * 16;
      for (int i = wave; i < kBgHdrSlot / 1024; i += 8)
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src + i * 1024), TF2_LDS_PTR(hdr_lds + slot * kBgHdrSlot + i * 1024), 16, 0, 0);
    };
    hdr_dma(a.hdr1, a.hdr1_bytes, 0, 0);
    hdr_dma(a.hdr2, a.hdr2_bytes, 0, 1);
#pragma unroll
    for (int q = 0; q < 4; q++) hdr_dma(a.hdr3, a.hdr3_bytes, q, 2 + q);
    {
      // the shortcut's m-tiles (64 or 128 rows each) share the last four slots
      const int n_mt = 256 / a.tms, per = (4 * kBgHdrSlot / n_mt) >> 10;       // KiB pieces per m-tile
      for (int i = wave; i < n_mt * per; i += 8) {
        const int mt = i / per, kk = i - mt * per;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(reinterpret_cast<const int8_t*>(a.hdrs) + (size_t)mt * a.hdrs_bytes + kk * 1024 + lane * 16),
                                         TF2_LDS_PTR(hdr_lds + 6 * kBgHdrSlot + i * 1024), 16, 0, 0);
      }
    }
    for (int u = wave; u < NWN * 2; u += 8) w_dma(a.w1, (size_t)(u >> 1) * 64 + 32 * (u & 1), wreg + u * 2048);      // reduce: [window][two 32-row tiles]
    // the band's input pixels: thirteen tiles of 32 pixels x 64 channels
    for (int gi = wave; gi < NT * 2; gi += 8) {
      const int p = 16 * gi + drow;
      const int8_t* src = p < NPB ? a.x + (px_band + p) * CIN + chunk * 16 : a.zero + chunk * 16;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(xt + gi * 1024), 16, 0, 0);
    }
    if (tid == 64 * 7) {
      i32x4 e;                                             // control words {step counter, error word, poll limit, test: withheld block + 1}
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(a.epoch) : "memory");
      ctl[0] = e[0]; ctl[4] = e[2]; ctl[5] = e[3];
    }
  }
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  const int* const prm1 = reinterpret_cast<const int*>(hdr_lds);
  const int* const prm2 = reinterpret_cast<const int*>(hdr_lds + kBgHdrSlot);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                         // headers, reduce weights and the band's input are in LDS
  if (ctl[5] == (int)blockIdx.x + 1) return;                // (test-only: this member leaves its group; the others report and go on)
  const unsigned tag = ((unsigned)ctl[0] << 8) | (xcc & 0xff);
  bg_rollcall_post(ctr, m, tag, tid);
  bool local0 = false;

  // two-window combine of a 32 x 32 tile: (hi << dshift[1][row]) + lo
  auto combine = [&](i32x16& hi, const i32x16& lo, const int* prm, int row0, int tm = 64) {
    const int* dsh = prm + (kPrmWordsPerRow + 1) * tm;
#pragma unroll
    for (int G = 0; G < 4; G++) {
      const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + row0 + 4 * half + 8 * G);
#pragma unroll
      for (int r = 0; r < 4; r++) hi[G * 4 + r] = (int)(((unsigned)hi[G * 4 + r] << (d[r] & 31)) + (unsigned)lo[G * 4 + r]);
    }
  };

  // =================================== phase A: reduce, 1x1 64 -> 64 (K = one slab) ===================================
  {
    const int lo_b = a.relu1 ? 0 : -128;
    local0 = bg_rollcall_wave(ctr, tag, lane, a.epoch, ctl[4], 0x01u);
#pragma unroll
    for (int rd = 0; rd < 2; rd++) {
      const int t = wave + 8 * rd;
      if (t < NT) {
        const int8_t* B = xt + t * 2048;
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + fr0), b1 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
        const int p = 32 * t + (lane & 31);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          i32x16 acc, accl;
#pragma unroll
          for (int r = 0; r < 16; r++) { acc[r] = 0; accl[r] = 0; }
          const int8_t* A = wreg + q * 2048;
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + fr0), b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + (fr0 ^ 32)), b1, acc, 0, 0, 0);
          if (DUAL) {
            accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + 4096 + fr0), b0, accl, 0, 0, 0);
            accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + 4096 + (fr0 ^ 32)), b1, accl, 0, 0, 0);
            combine(acc, accl, prm1, 32 * q);
          }
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[r];
          i32x4 out;
          if (a.fast1 == 1) out = requant_tile16<false, 0, true>(a16, prm1, 64, 32 * q + 4 * half, lo_b, -128, nores, a.dbl1 != 0, false);
          else out = requant_tile16<false, 0, false>(a16, prm1, 64, 32 * q + 4 * half, lo_b, -128, nores, a.dbl1 != 0, a.fast1 == 2);
          if (p < NPB) {
            int8_t* dst = a.mid1 + (px_band + p) * M + 32 * q + 16 * half;
            bg_store_x(dst, out, local0);
          }
        }
      }
    }
  }
  bg_signal(ctr + 8, m, tag, tid);
  for (int u = wave; u < 9 * 2; u += 8) w_dma(a.w2, (size_t)(u >> 1) * 64 + 32 * (u & 1), wreg + u * 2048);      // the 3x3's weights: [tap][two 32-row tiles]
  const int r_m1 = bg_wait(ctr + 8, tag, tid, ctl + 1, a.epoch, ctl[4], 0x20u | ((unsigned)0 << 8)); if (r_m1 < 0) return; const bool local1 = r_m1 > 0;

  // =================================== phase B: 3x3 / pad 1; the band's output stays in LDS ===================================
  {
    constexpr int NGRP = (PR + 2) * HC / 16;
    for (int grp = wave; grp < NGRP; grp += 8) {
      const int h = grp * 16 + drow;
      const int row = m * PR - 1 + (h >> 6), col = (h & 63) - 1;
      const bool ok = (unsigned)row < (unsigned)HW && (unsigned)col < (unsigned)HW;
      const int8_t* src = ok ? a.mid1 + (px_img + row * HW + col) * M + chunk * 16 : a.zero2 + chunk * 16;
      if (local1) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + grp * 1024), 16, 0, 1);
      else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + grp * 1024), 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int lo_b = a.relu2 ? 0 : -128;
#pragma unroll
    for (int rd = 0; rd < 2; rd++) {
      const int t = wave + 8 * rd;
      if (t < NT) {
        int p = 32 * t + (lane & 31);
        const bool ok = p < NPB;
        if (!ok) p = 0;
        const int oh = p / HW, ow = p - oh * HW;
        const int h0 = oh * HC + ow;
        i32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[q][r] = 0;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
          const int8_t* A = wreg + tap * 4096;
          const int h = h0 + (tap / 3) * HC + tap % 3;
          const int ba = h * 64 + ((half ^ ((h >> 2) & 3)) << 4);
          const i32x4 b0 = *reinterpret_cast<const i32x4*>(halo + ba), b1 = *reinterpret_cast<const i32x4*>(halo + (ba ^ 32));
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + fr0), b0, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + 2048 + fr0), b0, acc[1], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + (fr0 ^ 32)), b1, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const i32x4*>(A + 2048 + (fr0 ^ 32)), b1, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[q][r];
          i32x4 out;
          if (a.fast2 == 1) out = requant_tile16<false, 0, true>(a16, prm2, 64, 32 * q + 4 * half, lo_b, -128, nores, a.dbl2 != 0, false);
          else out = requant_tile16<false, 0, false>(a16, prm2, 64, 32 * q + 4 * half, lo_b, -128, nores, a.dbl2 != 0, a.fast2 == 2);
          const int row = lane & 31, c = 2 * q + half;
          *reinterpret_cast<i32x4*>(tiles + t * 2048 + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = out;
          if (ok) *reinterpret_cast<i32x4*>(a.mid2 + (px_band + 32 * t + row) * M + 32 * q + 16 * half) = out;
        }
      }
    }
  }
  // this wave's 32-row tile of the 256 output channels: expand and shortcut weights in registers (K = 64 each)
  const int ch3 = 32 * wave;
  const int mt3 = ch3 / 64, ro3 = ch3 % 64;
  i32x4 wf[NWN][2], wsf[NWN][2];
#pragma unroll
  for (int win = 0; win < NWN; win++) {
    const int8_t* p = a.w3 + (((size_t)mt3 * NWN + win) * 64 + ro3 + frow) * 64 + half * 16;
    wf[win][0] = *reinterpret_cast<const i32x4*>(p); wf[win][1] = *reinterpret_cast<const i32x4*>(p + 32);
    const int8_t* ps = a.ws + (((size_t)(ch3 / a.tms) * NWN + win) * a.tms + ch3 % a.tms + frow) * 64 + half * 16;
    wsf[win][0] = *reinterpret_cast<const i32x4*>(ps); wsf[win][1] = *reinterpret_cast<const i32x4*>(ps + 32);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                         // the band's 3x3 output is complete in LDS

  // =================================== phase C: shortcut tile -> residual, expand + residual ===================================
  {
    const int lo_b = a.relu3 ? 0 : -128, rlo = a.add_relu ? 0 : -128, lo_s = a.relu_s ? 0 : -128;
    const int* pm = reinterpret_cast<const int*>(hdr_lds + (2 + mt3) * kBgHdrSlot);
    const int ros = ch3 % a.tms;
    const int* ps = reinterpret_cast<const int*>(hdr_lds + 6 * kBgHdrSlot + (ch3 / a.tms) * (4 * kBgHdrSlot / (256 / a.tms)));
#pragma unroll 1
    for (int tt = 0; tt < NT; tt++) {
      const int p = 32 * tt + (lane & 31);
      // the shortcut convolution of this tile (1x1 64 -> 256 on the band's input), requantised: the residual
      i32x4 rs;
      {
        const int8_t* B = xt + tt * 2048;
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + fr0), b1 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
        i32x16 acc, accl;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[r] = 0; accl[r] = 0; }
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf[0][0], b0, acc, 0, 0, 0);
        if (DUAL) accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf[NWN - 1][0], b0, accl, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf[0][1], b1, acc, 0, 0, 0);
        if (DUAL) { accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(wsf[NWN - 1][1], b1, accl, 0, 0, 0); combine(acc, accl, ps, ros, a.tms); }
        int a16[16];
#pragma unroll
        for (int r = 0; r < 16; r++) a16[r] = acc[r];
        if (a.fast_s == 1) rs = requant_tile16<false, 0, true>(a16, ps, a.tms, ros + 4 * half, lo_s, -128, nores, false, false);
        else rs = requant_tile16<false, 0, false>(a16, ps, a.tms, ros + 4 * half, lo_s, -128, nores, false, a.fast_s == 2);
        if (a.keep_s && p < NPB) *reinterpret_cast<i32x4*>(a.ys + (px_band + p) * a.ys_cp + ch3 + 16 * half) = rs;
      }
      const int8_t* B = tiles + tt * 2048;
      const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + fr0), b1 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
      i32x16 acc, accl;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[r] = 0; accl[r] = 0; }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][0], b0, acc, 0, 0, 0);
      if (DUAL) accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[NWN - 1][0], b0, accl, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[0][1], b1, acc, 0, 0, 0);
      if (DUAL) { accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[NWN - 1][1], b1, accl, 0, 0, 0); combine(acc, accl, pm, ro3); }
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[r];
      i32x4 out;
      if (a.fast3 == 1) out = requant_tile16<true, 0, true>(a16, pm, 64, ro3 + 4 * half, lo_b, rlo, rs, false, false);
      else out = requant_tile16<true, 0, false>(a16, pm, 64, ro3 + 4 * half, lo_b, rlo, rs, false, a.fast3 == 2);
      if (p < NPB) *reinterpret_cast<i32x4*>(a.y + (px_band + p) * a.y_cp + a.y_off + ch3 + 16 * half) = out;
    }
  }
}

// ---- the 28 x 28 maps (ResNet-50 stage 3: C = 512, M = 128) --------------------------------------------------------------------
// 784 pixels are too many for one block's LDS and the intermediates have only 128 channels: the eight members of an image are
// FOUR row bands of seven rows (196 pixels = seven 32-pixel column tiles, as on the 14 x 14 maps) times TWO channel halves (64
// intermediate channels = two 32-row tiles, 256 expand channels = eight).  Reduce and 3x3: wave w < 7 owns column tile w of the
// band and both 32-row tiles (the reduce may be a two-window layer: one accumulator per window, combined (hi << dshift) + lo);
// weights LDS-resident, pixels through private rings (reduce) / the band's halo tile (3x3: 9 rows x 32 columns per slab; the rows
// above and below come from the neighbouring bands' members through the same meeting).  Expand: wave w owns 32-row tile w of the
// member's 256 channels with its weights in REGISTERS (K = 128: four fragments) and sweeps the band's seven column tiles.
// DUAL2: the 3x3 is a two-window layer: its weights are swept window by window (72 KB each: they share the LDS region, the low
// window is fetched while the high one's accumulators wait) and combined like the reduce's.
template <bool DUAL1, bool DUAL2>
__global__ __launch_bounds__(512, 2) void conv_bgroup28_kernel(BGroupChain c) {       // chained like conv_bgroup_kernel
  constexpr int HW = 28, C = 512, M = 128, PR = 7;       // PR: rows of a band
  constexpr int NPX = HW * HW, NPB = PR * HW;            // pixels of the image / of a band (196)
  constexpr int NT = (NPB + 31) / 32;                    // 7 column tiles per band
  constexpr int KS1 = C / 64, KS2 = M / 64;              // 8, 2
  constexpr int NE = 9 * KS2;                            // 18
  constexpr int NW1 = DUAL1 ? 2 : 1;
  constexpr int HC = 32, HALO = (PR + 2) * HC * 64;      // halo tile of a band: 9 rows x 32 columns (30 used) per 64-channel slab
  constexpr int S = 5, STAGE = 2048;
  constexpr int kHdrSlot = 4096;                         // rows | lo | dshift of a 128-row m-tile: <= 3584 bytes
  constexpr int W_BYTES = 72 * 1024, R_BYTES = NT * S * STAGE;
  static_assert(W_BYTES >= KS1 * NW1 * 4096 && W_BYTES >= NE * 4096 && R_BYTES >= KS2 * HALO && R_BYTES >= NT * KS2 * 2048, "phase regions");
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  int8_t* const hdr_lds = lds;                           // slots: reduce, 3x3, two m-tiles of the expand
  int8_t* const wreg = lds + 4 * kHdrSlot;
  int8_t* const work = wreg + W_BYTES;
  int* const ctl = reinterpret_cast<int*>(work + R_BYTES);

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int img = c.b[0].img0 + ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 6), m = ((int)blockIdx.x >> 3) & 7;
  if (img >= c.b[0].B) return;
  const int sp = m >> 1, cm = m & 1;                     // row band, channel half
  const size_t px_img = (size_t)img * NPX;
  const size_t px_band = px_img + (size_t)sp * NPB;      // first pixel of the band
  const i32x4 nores = {0, 0, 0, 0};
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  const int t = wave;
  const bool worker = wave < NT;
  unsigned tag = 0;
  bool local0 = false;                                      // roll call: the whole group on this XCD

#pragma unroll 1
  for (int kb = 0; kb < c.n; kb++) {
  const BGroupArgs& a = c.b[kb];
  const bool last = kb + 1 == c.n;
  // per-lane basics re-derived from an opaque copy of the thread id (see conv_bgroup_kernel)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, half = lane >> 5;
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
  const int frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);
  const int p_lane = 32 * t + (lane & 31);               // pixel of this lane's column inside the band (phases A, B)
  const bool p_ok = worker && p_lane < NPB;
  unsigned* const ctr = a.ctr + (size_t)img * 32;
  long long* const dbg = (a.dbg && kb == 0) ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bgroup_timeline.py
#define BG_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  BG_STAMP(0);
  const int c1 = 64 * cm;                                // first intermediate channel of this member
  const int mt1 = c1 / a.tm1, ro1 = c1 % a.tm1;
  const int mt2 = c1 / a.tm2, ro2 = c1 % a.tm2;
  const int c3 = 256 * cm;

  auto w_dma = [&](const int8_t* w, size_t row0, int8_t* dst) {        // 32 rows x 64 bytes starting at row row0 of the tile storage
#pragma unroll
    for (int g2 = 0; g2 < 2; g2++)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(w + (row0 + 16 * g2 + drow) * 64 + chunk * 16), TF2_LDS_PTR(dst + g2 * 1024), 16, 0, 0);
  };
  {
    auto hdr_dma = [&](const int32_t* hdr, int hdr_bytes, int mt, int tm, int nwords, int slot) {
      const int used = (nwords * tm * 4 + 1023) & ~1023;
      const int8_t* src = reinterpret_cast<const int8_t*>(hdr) + (size_t)mt * hdr_bytes + lane * 16;
      for (int i = wave; i * 1024 < used; i += 8)
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src + i * 1024), TF2_LDS_PTR(hdr_lds + slot * kHdrSlot + i * 1024), 16, 0, 0);
    };
    hdr_dma(a.hdr1, a.hdr1_bytes, mt1, a.tm1, kPrmWordsPerRow + 2, 0);
    hdr_dma(a.hdr2, a.hdr2_bytes, mt2, a.tm2, kPrmWordsPerRow + 2, 1);
    hdr_dma(a.hdr3, a.hdr3_bytes, c3 / a.tm3, a.tm3, kPrmWordsPerRow, 2);
    if (a.tm3 < 256) hdr_dma(a.hdr3, a.hdr3_bytes, c3 / a.tm3 + 1, a.tm3, kPrmWordsPerRow, 3);       // (64-row m-tiles: the kernel needs 4 slots -- not instantiated)
    // the reduce's weights: [slab][window][two 32-row tiles]
    for (int u = wave; u < KS1 * NW1 * 2; u += 8) {
      const int s = u / (NW1 * 2), win = (u / 2) % NW1, ctq = u & 1;
      w_dma(a.w1, (((size_t)mt1 * KS1 + s) * NW1 + win) * a.tm1 + ro1 + 32 * ctq, wreg + u * 2048);
    }
    if (kb == 0 && tid == 64 * 7) {
      i32x4 e;                                             // control words {step counter, error word, poll limit, test: withheld block + 1}
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(a.epoch) : "memory");
      ctl[0] = e[0]; ctl[4] = e[2]; ctl[5] = e[3];
    }
  }
  const int* const prm1 = reinterpret_cast<const int*>(hdr_lds);
  const int* const prm2 = reinterpret_cast<const int*>(hdr_lds + kHdrSlot);

  // requantise [two 32-row tiles] x [column tile t] and store into a mid tensor
  auto store_mid = [&](i32x16 (&acc)[2], const int* prm, int tm, int ro, int fast, int relu, int dbl, int8_t* mid) {
    const int lo_b = relu ? 0 : -128;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[q][r];
      i32x4 out;
      if (fast == 1) out = requant_tile16<false, 0, true>(a16, prm, tm, ro + 32 * q + 4 * half, lo_b, -128, nores, dbl != 0, false);
      else out = requant_tile16<false, 0, false>(a16, prm, tm, ro + 32 * q + 4 * half, lo_b, -128, nores, dbl != 0, fast == 2);
      if (p_ok) {
        int8_t* dst = mid + (px_band + p_lane) * M + c1 + 32 * q + 16 * half;
        bg_store_x(dst, out, local0);
      }
    }
  };

  // kb > 0: the input is the previous bottleneck's output: the members meet at this one's roll-call row first
  bool local_in = false;
  if (kb > 0) { const int r_in = bg_wait(ctr, tag, tid, ctl + 3, a.epoch, ctl[4], 0x10u | ((unsigned)kb << 8)); if (r_in < 0) return; local_in = r_in > 0; }

  // =================================== phase A: reduce, 1x1 C -> M ===================================
  {
    int8_t* const ring = work + wave * (S * STAGE);
    auto issue = [&](int s, int slot) {
#pragma unroll
      for (int g2 = 0; g2 < 2; g2++) {
        const int p = 32 * t + 16 * g2 + drow;
        const int8_t* src = p < NPB ? a.x + (px_band + p) * C + s * 64 + chunk * 16 : a.zero + chunk * 16;
        int8_t* const dst = ring + slot * STAGE + g2 * 1024;
        if (kb == 0) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 0);               // written before this launch
        else if (local_in) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 1);       // by this group, in this XCD's L2
        else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 16);
      }
    };
    if (worker) {
#pragma unroll
      for (int s = 0; s < S - 1; s++) issue(s, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // headers and the reduce's weights are in LDS
    BG_STAMP(1);
    if (kb == 0) {
      if (ctl[5] == (int)blockIdx.x + 1) return;            // (test-only: this member leaves its group; the others report and go on)
      tag = ((unsigned)ctl[0] << 8) | (xcc & 0xff);
      bg_rollcall_post(ctr, m, tag, tid);
    }
    i32x16 acc[2], accl[DUAL1 ? 2 : 1];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[q][r] = 0; if (DUAL1) accl[q][r] = 0; }
    if (worker) {
      int cs = 0, is = S - 1;
      for (int s = 0; s < KS1; s++) {
        if (s >= S - 1) { if (s + 3 < KS1) bg_wait_vmcnt<6>(); else if (s + 2 < KS1) bg_wait_vmcnt<4>(); else if (s + 1 < KS1) bg_wait_vmcnt<2>(); else bg_wait_vmcnt<0>(); }
        const int8_t* A = wreg + s * (NW1 * 4096);         // [window][tile q][32 rows][64]
        const int8_t* B = ring + cs * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          const i32x4 b = *reinterpret_cast<const i32x4*>(B + (fr0 ^ (ks << 5)));
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const i32x4 ah = *reinterpret_cast<const i32x4*>(A + q * 2048 + (fr0 ^ (ks << 5)));
            acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, b, acc[q], 0, 0, 0);
            if (DUAL1) {
              const i32x4 al = *reinterpret_cast<const i32x4*>(A + 4096 + q * 2048 + (fr0 ^ (ks << 5)));
              accl[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, b, accl[q], 0, 0, 0);
            }
          }
        }
        if (s + S - 1 < KS1) { issue(s + S - 1, is); is = is + 1 == S ? 0 : is + 1; }
        cs = cs + 1 == S ? 0 : cs + 1;
      }
      if (DUAL1) {
        const int* dsh = prm1 + (kPrmWordsPerRow + 1) * a.tm1;      // dshift[1][row]
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int G = 0; G < 4; G++) {
            const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + ro1 + 32 * q + 4 * half + 8 * G);
#pragma unroll
            for (int r = 0; r < 4; r++)
              acc[q][G * 4 + r] = (int)(((unsigned)acc[q][G * 4 + r] << (d[r] & 31)) + (unsigned)accl[q][G * 4 + r]);
          }
      }
      if (kb == 0) local0 = bg_rollcall_wave(ctr, tag, lane, a.epoch, ctl[4], 0x01u);
      store_mid(acc, prm1, a.tm1, ro1, a.fast1, a.relu1, a.dbl1, a.mid1);
    }
  }
  BG_STAMP(2);
  bg_signal(ctr + 8, m, tag, tid);
  BG_STAMP(3);
  // the 3x3's weights (of one window): [step e][two 32-row tiles]
  constexpr int NW2 = DUAL2 ? 2 : 1;
  auto load_w2 = [&](int win) {
    for (int u = wave; u < NE * 2; u += 8)
      w_dma(a.w2, (((size_t)mt2 * NE + (u >> 1)) * NW2 + win) * a.tm2 + ro2 + 32 * (u & 1), wreg + u * 2048);
  };
  load_w2(0);
  const int r_m1 = bg_wait(ctr + 8, tag, tid, ctl + 1, a.epoch, ctl[4], 0x20u | ((unsigned)kb << 8)); if (r_m1 < 0) return; const bool local1 = r_m1 > 0;
  BG_STAMP(4);

  // =================================== phase B: 3x3 / pad 1, M -> M ===================================
  {
    int8_t* const halo = work;                             // [KS2][9 x 32 halo pixels][64], halo (r, c) = pixel (7 sp - 1 + r, c - 1)
    constexpr int NGRP = (PR + 2) * HC / 16;               // 18 groups of 16 halo pixels per slab
    for (int gi = wave; gi < KS2 * NGRP; gi += 8) {
      const int s = gi / NGRP, grp = gi - s * NGRP;
      const int h = grp * 16 + drow;
      const int row = sp * PR - 1 + (h >> 5), col = (h & 31) - 1;
      const bool ok = (unsigned)row < (unsigned)HW && (unsigned)col < (unsigned)HW;
      const int8_t* src = ok ? a.mid1 + (px_img + row * HW + col) * M + s * 64 + chunk * 16 : a.zero2 + s * 64 + chunk * 16;
      if (local1) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + s * HALO + grp * 1024), 16, 0, 1);
      else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + s * HALO + grp * 1024), 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // halo and weights complete in every wave
    BG_STAMP(5);
    const int pq = p_ok ? p_lane : 0;
    const int oh = pq / HW, ow = pq - oh * HW;
    const int h0 = oh * HC + ow;
    i32x16 acc[2], acch[DUAL2 ? 2 : 1];
#pragma unroll
    for (int win = 0; win < NW2; win++) {
      if (win == 1) {
        // the high window is done: keep its sums, fetch the low window's weights into the same region
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int r = 0; r < 16; r++) acch[q][r] = acc[q][r];
        __syncthreads();                                   // every wave is done reading the high window
        load_w2(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[q][r] = 0;
    if (worker) {
      auto step = [&](auto e_c) {
        constexpr int e = decltype(e_c)::value;
        constexpr int tap = e / KS2, sl = e % KS2;
        const int8_t* A = wreg + e * 4096;
        const int h = h0 + (tap / 3) * HC + tap % 3;
        const int ba = sl * HALO + h * 64 + ((half ^ ((h >> 2) & 3)) << 4);
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(halo + ba), b1 = *reinterpret_cast<const i32x4*>(halo + (ba ^ 32));
        const i32x4 a00 = *reinterpret_cast<const i32x4*>(A + fr0), a01 = *reinterpret_cast<const i32x4*>(A + (fr0 ^ 32));
        const i32x4 a10 = *reinterpret_cast<const i32x4*>(A + 2048 + fr0), a11 = *reinterpret_cast<const i32x4*>(A + 2048 + (fr0 ^ 32));
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a00, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a10, b0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a01, b1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a11, b1, acc[1], 0, 0, 0);
      };
      bg_static_for<0, NE>(step);
    }
    }   // windows
    if (worker) {
      if (DUAL2) {
        const int* dsh = prm2 + (kPrmWordsPerRow + 1) * a.tm2;      // dshift[1][row]: (hi << d) + lo
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int G = 0; G < 4; G++) {
            const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + ro2 + 32 * q + 4 * half + 8 * G);
#pragma unroll
            for (int r = 0; r < 4; r++)
              acc[q][G * 4 + r] = (int)(((unsigned)acch[q][G * 4 + r] << (d[r] & 31)) + (unsigned)acc[q][G * 4 + r]);
          }
      }
      store_mid(acc, prm2, a.tm2, ro2, a.fast2, a.relu2, a.dbl2, a.mid2);
    }
  }
  BG_STAMP(6);
  bg_signal(ctr + 16, m, tag, tid);
  BG_STAMP(7);
  // the expand's weights of this wave (32-row tile `wave` of the member's 256 channels): K = 128 -> four fragments, in registers
  const int ch3 = c3 + 32 * wave;
  i32x4 wf[KS2][2];
  {
    const int mt = ch3 / a.tm3, ro = ch3 % a.tm3;
#pragma unroll
    for (int s = 0; s < KS2; s++) {
      const int8_t* p = a.w3 + (((size_t)mt * KS2 + s) * a.tm3 + ro + frow) * 64 + half * 16;
      wf[s][0] = *reinterpret_cast<const i32x4*>(p); wf[s][1] = *reinterpret_cast<const i32x4*>(p + 32);
    }
  }
  const int r_m2 = bg_wait(ctr + 16, tag, tid, ctl + 2, a.epoch, ctl[4], 0x30u | ((unsigned)kb << 8)); if (r_m2 < 0) return; const bool local2 = r_m2 > 0;
  BG_STAMP(8);

  // =================================== phase C: expand, 1x1 M -> C, + residual ===================================
  {
    int8_t* const tiles = work;                            // [column tile][KS2][32 pixels][64]: the band's 3x3 output
    for (int gi = wave; gi < NT * KS2 * 2; gi += 8) {
      const int tt = gi / (KS2 * 2), s = (gi >> 1) % KS2, g2 = gi & 1;
      const int p = 32 * tt + 16 * g2 + drow;
      const int8_t* src = p < NPB ? a.mid2 + (px_band + p) * M + s * 64 + chunk * 16 : a.zero + chunk * 16;
      if (local2) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(tiles + (tt * KS2 + s) * 2048 + g2 * 1024), 16, 0, 1);
      else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(tiles + (tt * KS2 + s) * 2048 + g2 * 1024), 16, 0, 16);
    }
    const int lo_b = a.relu3 ? 0 : -128, rlo = a.add_relu ? 0 : -128;
    const int mt = ch3 / a.tm3, ro = ch3 % a.tm3;
    const int* pm = reinterpret_cast<const int*>(hdr_lds + (2 + (mt - c3 / a.tm3)) * kHdrSlot);
    // Residual in, outputs out: through an LDS image of the band's [196 pixels][256 channels of this member] so that the global
    // side moves whole 256-byte pixel rows (a wave-tile's own pieces are 16 bytes at a stride of y_cp: 3.6 TB/s measured that way).
    // Row px keeps its sixteen 16-byte pieces XOR-swizzled with px & 15: a column read (32 pixels, one piece) is conflict-free.
    int8_t* const stage = wreg;                            // (the 3x3's weights are dead)
    constexpr int NCHUNK = (NPB * 256 + 1023) / 1024;      // 1 KiB = four pixel rows
    if (a.has_res) {
      for (int ci = wave; ci < NCHUNK; ci += 8) {
        const int px = 4 * ci + (lane >> 4), q = (lane & 15) ^ (px & 15);
        const int8_t* src = px < NPB ? a.res + (px_band + px) * a.res_cp + a.res_off + c3 + q * 16 : a.zero;
        // (kb > 0: this block's own output of one bottleneck earlier, stored like exchange data)
        if (kb == 0) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(stage + ci * 1024), 16, 0, 0);
        else if (local2) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(stage + ci * 1024), 16, 0, 1);
        else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(stage + ci * 1024), 16, 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // the band's tiles (fetched by every wave)
    BG_STAMP(9);
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
      const int pxl = 32 * tt + (lane & 31);
      int8_t* const slot = stage + pxl * 256 + ((((2 * wave + half) ^ (pxl & 15))) << 4);
      const i32x4 rcur = *reinterpret_cast<const i32x4*>(slot);
      i32x16 acc, acc1;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[r] = 0; acc1[r] = 0; }
#pragma unroll
      for (int s = 0; s < KS2; s++) {
        const int8_t* B = tiles + (tt * KS2 + s) * 2048;
        const i32x4 b0 = *reinterpret_cast<const i32x4*>(B + fr0), b1 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s][0], b0, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[s][1], b1, acc1, 0, 0, 0);
      }
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = (int)((unsigned)acc[r] + (unsigned)acc1[r]);
      i32x4 out;
      if (a.fast3 == 1) {
        if (a.has_res) out = requant_tile16<true, 0, true, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, rcur, false, false);
        else out = requant_tile16<false, 0, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, false);
      } else {
        if (a.has_res) out = requant_tile16<true, 0, false, true>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, rcur, false, a.fast3 == 2);
        else out = requant_tile16<false, 0, false>(a16, pm, a.tm3, ro + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, a.fast3 == 2);
      }
      *reinterpret_cast<i32x4*>(slot) = out;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                       // the band's outputs are complete in LDS
    for (int ci = wave; ci < NCHUNK; ci += 8) {
      const int px = 4 * ci + (lane >> 4), q = (lane & 15) ^ (px & 15);
      const i32x4 v = *reinterpret_cast<const i32x4*>(stage + ci * 1024 + lane * 16);
      if (px < NPB) bg_store_x(a.y + (px_band + px) * a.y_cp + a.y_off + c3 + q * 16, v, local2 || last);      // (an inner output is exchange data)
    }
  }
  BG_STAMP(10);
#undef BG_STAMP
  if (!last) bg_signal(c.b[kb + 1].ctr + (size_t)img * 32, m, tag, tid);      // stores acknowledged, LDS free, then "my output is complete"
  }
}

// ---- the 7 x 7 maps (ResNet-50 stage 5: C = 2048, M = 512) --------------------------------------------------------------------
// 49 pixels are two 32-pixel column tiles, so the parallelism of a member comes from its channels and from K: a member owns
// 64 intermediate channels (one 64-row m-tile) and C / 8 = 256 expand channels; in the reduce and the 3x3 wave w works on
// 32-row tile (w & 1) and on K quarter (w >> 1) for both column tiles, the four quarters meet in LDS; in the expand wave w owns
// the 32-row tile w.  A member's weights do not fit LDS here (the 3x3's 64 rows are 295 KB): every wave streams its own weight
// rows through a private LDS-DMA ring.  The reduce may be a two-window layer (DUAL1: it reads the stage's multi-Q input):
// an entry then holds [hi 64 rows | lo 64 rows], a wave keeps an accumulator per window and combines them (hi << dshift) + lo
// before the quarters are added -- the same value in Z/2^32.  AVG: the bottleneck ends in the global average
// (full_size_pool.cl:95-125): the expand's wave sums its 49 requantised + residual-added columns per channel and stores the
// averaged vector; the 7 x 7 map of the last layer never reaches memory.
template <bool DUAL1, bool AVG>
__global__ __launch_bounds__(512, 2) void conv_bgroup7_kernel(BGroupChain c) {        // chained like conv_bgroup_kernel; AVG: the LAST one averages
  constexpr int HW = 7, C = 2048, M = 512;
  constexpr int NPX = HW * HW;
  constexpr int KS1 = C / 64, KS2 = M / 64;              // 32, 8
  constexpr int NE = 9 * KS2;                            // 72 (tap, slab) steps of the 3x3
  constexpr int NW1 = DUAL1 ? 2 : 1;
  constexpr int HC = 10, NHALO = 96, HALO = NHALO * 64;  // halo grid 9 x 10 (90 of 96 slots) per 64-channel slab
  constexpr int kHdrSlots = 6;                           // reduce, 3x3, four m-tiles of the expand
  constexpr int STA = NW1 * 2048 + 4096;                 // reduce ring stage: 32 weight rows per window | 64 pixels
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  int8_t* const hdr_lds = lds;
  int8_t* const wreg = lds + kHdrSlots * kBgHdrSlot;     // 96 KB: weight rings (phases B, C); reduction partials
  int8_t* const work = wreg + 96 * 1024;                 // 48 KB: halo | expand tiles.  Phase A rings span both regions.
  int* const ctl = reinterpret_cast<int*>(work + 48 * 1024);

  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int img = c.b[0].img0 + ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 6), m = ((int)blockIdx.x >> 3) & 7;
  if (img >= c.b[0].B) return;
  const size_t px_img = (size_t)img * NPX;
  const i32x4 nores = {0, 0, 0, 0};
  const int ct = wave & 1, kq = wave >> 1;               // phases A, B: 32-row tile of the member's m-tile, K quarter
  const int c1 = 64 * m;                                 // first intermediate channel of this member (m-tile m of 64 rows)
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  bool local0 = false;                                      // roll call: the whole group on this XCD
  unsigned tag = 0;

#pragma unroll 1
  for (int kb = 0; kb < c.n; kb++) {
  const BGroupArgs& a = c.b[kb];
  const bool last = kb + 1 == c.n;
  // per-lane basics re-derived from an opaque copy of the thread id: otherwise everything that depends only on the lane is hoisted
  // out of this loop and kept in registers across it (256 VGPRs + scratch instead of 143)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, half = lane >> 5;
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3), drow = lane >> 2;
  const int frow = lane & 31;
  const int fr0 = frow * 64 + ((half ^ ((frow >> 2) & 3)) << 4);
  unsigned* const ctr = a.ctr + (size_t)img * 32;
  long long* const dbg = (a.dbg && kb == 0) ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;       // tools/bgroup_timeline.py
#define BG_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  BG_STAMP(0);

  auto w_dma = [&](const int8_t* w, size_t row0, int8_t* dst) {        // 32 rows x 64 bytes starting at tile row row0
#pragma unroll
    for (int g2 = 0; g2 < 2; g2++)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(w + (row0 + 16 * g2 + drow) * 64 + chunk * 16), TF2_LDS_PTR(dst + g2 * 1024), 16, 0, 0);
  };
  {
    auto hdr_dma = [&](const int32_t* hdr, int hdr_bytes, int mt, int slot) {
      const int8_t* src = reinterpret_cast<const int8_t*>(hdr) + (size_t)mt * hdr_bytes + lane * 16;
      for (int i = wave; i < kBgHdrSlot / 1024; i += 8)              // rows | lo | dshift of a 64-row m-tile: <= 1792 bytes
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src + i * 1024), TF2_LDS_PTR(hdr_lds + slot * kBgHdrSlot + i * 1024), 16, 0, 0);
    };
    hdr_dma(a.hdr1, a.hdr1_bytes, m, 0);
    hdr_dma(a.hdr2, a.hdr2_bytes, m, 1);
#pragma unroll
    for (int q = 0; q < 4; q++) hdr_dma(a.hdr3, a.hdr3_bytes, 4 * m + q, 2 + q);
    if (kb == 0 && tid == 64 * 7) {
      i32x4 e;                                             // control words {step counter, error word, poll limit, test: withheld block + 1}
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(e) : "v"(a.epoch) : "memory");
      ctl[0] = e[0]; ctl[4] = e[2]; ctl[5] = e[3];
    }
  }
  const int* const prm1 = reinterpret_cast<const int*>(hdr_lds);
  const int* const prm2 = reinterpret_cast<const int*>(hdr_lds + kBgHdrSlot);

  // the four K quarters of a 32-row tile meet: quarters 1..3 park their two column tiles in LDS, quarter 0 adds them up
  auto reduce_quarters = [&](i32x16 (&acc)[2], int8_t* park) {
    if (kq > 0) {
#pragma unroll
      for (int pt = 0; pt < 2; pt++)
#pragma unroll
        for (int G = 0; G < 4; G++)
          reinterpret_cast<i32x4*>(park)[(((ct * 3 + kq - 1) * 2 + pt) * 4 + G) * 64 + lane] = i32x4{acc[pt][4 * G], acc[pt][4 * G + 1], acc[pt][4 * G + 2], acc[pt][4 * G + 3]};
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int q = 0; q < 3; q++)
#pragma unroll
        for (int pt = 0; pt < 2; pt++)
#pragma unroll
          for (int G = 0; G < 4; G++) {
            const i32x4 v = reinterpret_cast<const i32x4*>(park)[(((ct * 3 + q) * 2 + pt) * 4 + G) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; r++) acc[pt][4 * G + r] = (int)((unsigned)acc[pt][4 * G + r] + (unsigned)v[r]);
          }
    }
  };
  // requantise the member's [32 rows of tile ct] x [two column tiles] and store them into a mid tensor (M bytes per pixel)
  auto store_mid = [&](i32x16 (&acc)[2], const int* prm, int fast, int relu, int dbl, int8_t* mid) {
    const int lo_b = relu ? 0 : -128;
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[pt][r];
      i32x4 out;
      if (fast == 1) out = requant_tile16<false, 0, true>(a16, prm, 64, 32 * ct + 4 * half, lo_b, -128, nores, dbl != 0, false);
      else out = requant_tile16<false, 0, false>(a16, prm, 64, 32 * ct + 4 * half, lo_b, -128, nores, dbl != 0, fast == 2);
      const int p = 32 * pt + (lane & 31);
      if (p < NPX) {
        int8_t* dst = mid + (px_img + p) * M + c1 + 32 * ct + 16 * half;
        bg_store_x(dst, out, local0);
      }
    }
  };

  // kb > 0: the input is the previous bottleneck's output: the members meet at this one's roll-call row first
  bool local_in = false;
  if (kb > 0) { const int r_in = bg_wait(ctr, tag, tid, ctl + 3, a.epoch, ctl[4], 0x10u | ((unsigned)kb << 8)); if (r_in < 0) return; local_in = r_in > 0; }

  // =================================== phase A: reduce, 1x1 C -> M ===================================
  {
    constexpr int NS = KS1 / 4;                            // slabs of a K quarter
    constexpr int NI = 2 * NW1 + 4;                        // LDS-DMAs of a stage
    int8_t* const ring = wreg + wave * (2 * STA);
    const int8_t* const wbase = a.w1;                      // entry e = m * KS1 + s: [window][64 rows][64]
    auto issue = [&](int s, int slot) {
      int8_t* const st = ring + slot * STA;
      const size_t e = (size_t)m * KS1 + (kq * NS + s);
#pragma unroll
      for (int win = 0; win < NW1; win++) w_dma(wbase, (e * NW1 + win) * 64 + 32 * ct, st + win * 2048);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int p = 16 * g4 + drow;
        const int8_t* src = p < NPX ? a.x + (px_img + p) * C + (kq * NS + s) * 64 + chunk * 16 : a.zero + chunk * 16;
        int8_t* const dst = st + NW1 * 2048 + g4 * 1024;
        if (kb == 0) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 0);               // written before this launch
        else if (local_in) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 1);       // by this group, in this XCD's L2
        else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(dst), 16, 0, 16);
      }
    };
    // (the header pieces and the step counter of this wave are older in its queue than its ring stages)
    issue(0, 0); issue(1, 1);
    i32x16 acc[2], accl[DUAL1 ? 2 : 1];
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[pt][r] = 0; if (DUAL1) accl[pt][r] = 0; }
    for (int s = 0; s < NS; s++) {
      if (s + 1 < NS) bg_wait_vmcnt<NI>(); else bg_wait_vmcnt<0>();
      const int8_t* A = ring + (s & 1) * STA;
      const int8_t* B = A + NW1 * 2048;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const i32x4 ah = *reinterpret_cast<const i32x4*>(A + (fr0 ^ (ks << 5)));
        i32x4 al = ah;
        if (DUAL1) al = *reinterpret_cast<const i32x4*>(A + 2048 + (fr0 ^ (ks << 5)));
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          const i32x4 b = *reinterpret_cast<const i32x4*>(B + pt * 2048 + (fr0 ^ (ks << 5)));
          acc[pt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah, b, acc[pt], 0, 0, 0);
          if (DUAL1) accl[pt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al, b, accl[pt], 0, 0, 0);
        }
      }
      if (s + 2 < NS) issue(s + 2, s & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // every wave is done with its ring; headers (fetched by every wave) are in LDS
    BG_STAMP(1);
    if (kb == 0) {
      if (ctl[5] == (int)blockIdx.x + 1) return;            // (test-only: this member leaves its group; the others report and go on)
      tag = ((unsigned)ctl[0] << 8) | (xcc & 0xff);
      bg_rollcall_post(ctr, m, tag, tid);
    }
    if (DUAL1) {
      // combine the windows: (hi << dshift[1][row]) + lo
      const int* dsh = prm1 + (kPrmWordsPerRow + 1) * 64;
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 32 * ct + 4 * half + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int pt = 0; pt < 2; pt++)
            acc[pt][G * 4 + r] = (int)(((unsigned)acc[pt][G * 4 + r] << (d[r] & 31)) + (unsigned)accl[pt][G * 4 + r]);
      }
    }
    reduce_quarters(acc, wreg);
    if (kq == 0) {
      if (kb == 0) local0 = bg_rollcall_wave(ctr, tag, lane, a.epoch, ctl[4], 0x01u);
      store_mid(acc, prm1, a.fast1, a.relu1, a.dbl1, a.mid1);
    }
    BG_STAMP(2);
  }
  bg_signal(ctr + 8, m, tag, tid);
  BG_STAMP(3);
  const int r_m1 = bg_wait(ctr + 8, tag, tid, ctl + 1, a.epoch, ctl[4], 0x20u | ((unsigned)kb << 8)); if (r_m1 < 0) return; const bool local1 = r_m1 > 0;
  BG_STAMP(4);

  // =================================== phase B: 3x3 / pad 1, M -> M ===================================
  {
    int8_t* const halo = work;                             // [KS2][9 x 10 halo pixels (+ 6 spare)][64], halo (r, c) = pixel (r - 1, c - 1)
    constexpr int NGRP = NHALO / 16;                       // 6 groups of 16 halo pixels per slab
    for (int gi = wave; gi < KS2 * NGRP; gi += 8) {
      const int s = gi / NGRP, grp = gi - s * NGRP;
      const int h = grp * 16 + drow;
      const int hr = h / HC;
      const int row = hr - 1, col = h - hr * HC - 1;
      const bool ok = (unsigned)row < (unsigned)HW && (unsigned)col < (unsigned)HW;
      const int8_t* src = ok ? a.mid1 + (px_img + row * HW + col) * M + s * 64 + chunk * 16 : a.zero2 + s * 64 + chunk * 16;
      if (local1) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + s * HALO + grp * 1024), 16, 0, 1);
      else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(halo + s * HALO + grp * 1024), 16, 0, 16);
    }
    constexpr int NS = NE / 4, S = 6;                      // 18 steps per K quarter, private ring of 32 weight rows per stage
    int8_t* const ring = wreg + wave * (S * 2048);
    auto issue = [&](int s, int slot) {
      const size_t e = (size_t)m * NE + (kq * NS + s);
      w_dma(a.w2, e * 64 + 32 * ct, ring + slot * 2048);
    };
#pragma unroll
    for (int s = 0; s < S - 1; s++) issue(s, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // halo complete in every wave (and this wave's first stages)
    BG_STAMP(5);
    int h0[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
      int p = 32 * pt + (lane & 31);
      if (p >= NPX) p = 0;
      const int oh = p / HW;
      h0[pt] = oh * HC + (p - oh * HW);
    }
    i32x16 acc[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[pt][r] = 0;
    int cs = 0, is = S - 1;
    for (int s = 0; s < NS; s++) {
      // stages 0 .. S-2 landed above; later S-2 younger stages (2 DMAs each) may fly while that many have been issued
      if (s >= S - 1) {
        if (s + 4 < NS) bg_wait_vmcnt<8>(); else if (s + 3 < NS) bg_wait_vmcnt<6>(); else if (s + 2 < NS) bg_wait_vmcnt<4>();
        else if (s + 1 < NS) bg_wait_vmcnt<2>(); else bg_wait_vmcnt<0>();
      }
      const int e = kq * NS + s;
      const int tap = e / KS2, sl = e - tap * KS2;
      const int8_t* A = ring + cs * 2048;
      const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + fr0), a1 = *reinterpret_cast<const i32x4*>(A + (fr0 ^ 32));
      const int toff = (tap / 3) * HC + tap % 3;
      i32x4 b0[2], b1[2];
#pragma unroll
      for (int pt = 0; pt < 2; pt++) {
        const int h = h0[pt] + toff;
        const int ba = sl * HALO + h * 64 + ((half ^ ((h >> 2) & 3)) << 4);
        b0[pt] = *reinterpret_cast<const i32x4*>(halo + ba); b1[pt] = *reinterpret_cast<const i32x4*>(halo + (ba ^ 32));
      }
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0[0], acc[0], 0, 0, 0);      // the two column tiles alternate
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0[1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1[1], acc[1], 0, 0, 0);
      if (s + S - 1 < NS) { issue(s + S - 1, is); is = is + 1 == S ? 0 : is + 1; }
      cs = cs + 1 == S ? 0 : cs + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // rings are dead: the partials may use the region
    BG_STAMP(6);
    reduce_quarters(acc, wreg);
    if (kq == 0) store_mid(acc, prm2, a.fast2, a.relu2, a.dbl2, a.mid2);
  }
  bg_signal(ctr + 16, m, tag, tid);
  BG_STAMP(7);
  // residual tiles of this wave's 32-row tile of the expand (the bottleneck's input, written before this launch or -- in a
  // chain -- by THIS thread one bottleneck earlier: ordinary loads)
  const int ch3 = (C / kBgMembers) * m + 32 * wave;        // first channel of this wave's tile
  i32x4 rv[2];
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    const int p = 32 * pt + (lane & 31);
    const int8_t* rp = (a.has_res && p < NPX) ? a.res + (px_img + p) * a.res_cp + a.res_off + ch3 + 16 * half : a.zero;
    rv[pt] = *reinterpret_cast<const i32x4*>(rp);
  }
  // the expand's weights of this wave: private ring, first stages on their way while the group gathers
  constexpr int S3 = 4;
  int8_t* const ring3 = wreg + wave * (S3 * 2048);
  const int mt3 = ch3 / 64, ro3 = ch3 % 64;
  auto issue3 = [&](int s, int slot) { w_dma(a.w3, ((size_t)mt3 * KS2 + s) * 64 + ro3, ring3 + slot * 2048); };
#pragma unroll
  for (int s = 0; s < S3 - 1; s++) issue3(s, s);
  const int r_m2 = bg_wait(ctr + 16, tag, tid, ctl + 2, a.epoch, ctl[4], 0x30u | ((unsigned)kb << 8)); if (r_m2 < 0) return; const bool local2 = r_m2 > 0;
  BG_STAMP(8);

  // =================================== phase C: expand, 1x1 M -> C, + residual (+ global average) ===================================
  {
    int8_t* const tiles = work;                            // [KS2][64 pixels][64]: both column tiles of the image
    for (int gi = wave; gi < KS2 * 4; gi += 8) {
      const int s = gi >> 2, g4 = gi & 3;
      const int p = 16 * g4 + drow;
      const int8_t* src = p < NPX ? a.mid2 + (px_img + p) * M + s * 64 + chunk * 16 : a.zero + chunk * 16;
      if (local2) __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(tiles + s * 4096 + g4 * 1024), 16, 0, 1);
      else __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(tiles + s * 4096 + g4 * 1024), 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // the image's tiles (fetched by every wave)
    BG_STAMP(9);
    i32x16 acc[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[pt][r] = 0;
    int cs = 0, is = S3 - 1;
    for (int s = 0; s < KS2; s++) {
      if (s >= S3 - 1) { if (s + 2 < KS2) bg_wait_vmcnt<4>(); else if (s + 1 < KS2) bg_wait_vmcnt<2>(); else bg_wait_vmcnt<0>(); }
      const int8_t* A = ring3 + cs * 2048;
      const i32x4 a0 = *reinterpret_cast<const i32x4*>(A + fr0), a1 = *reinterpret_cast<const i32x4*>(A + (fr0 ^ 32));
      const int8_t* B = tiles + s * 4096;
      const i32x4 b00 = *reinterpret_cast<const i32x4*>(B + fr0), b01 = *reinterpret_cast<const i32x4*>(B + (fr0 ^ 32));
      const i32x4 b10 = *reinterpret_cast<const i32x4*>(B + 2048 + fr0), b11 = *reinterpret_cast<const i32x4*>(B + 2048 + (fr0 ^ 32));
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b00, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b10, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b01, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b11, acc[1], 0, 0, 0);
      if (s + S3 - 1 < KS2) { issue3(s + S3 - 1, is); is = is + 1 == S3 ? 0 : is + 1; }
      cs = cs + 1 == S3 ? 0 : cs + 1;
    }
    const int lo_b = a.relu3 ? 0 : -128, rlo = a.add_relu ? 0 : -128;
    const int* pm = reinterpret_cast<const int*>(hdr_lds + (2 + (mt3 - 4 * m)) * kBgHdrSlot);
    int sum16[16];
#pragma unroll
    for (int q = 0; q < 16; q++) sum16[q] = 0;
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
      int a16[16];
#pragma unroll
      for (int r = 0; r < 16; r++) a16[r] = acc[pt][r];
      i32x4 out;
      if (a.fast3 == 1) {
        if (a.has_res) out = requant_tile16<true, 0, true, true>(a16, pm, 64, ro3 + 4 * half, lo_b, rlo, rv[pt], false, false);
        else out = requant_tile16<false, 0, true>(a16, pm, 64, ro3 + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, false);
      } else {
        if (a.has_res) out = requant_tile16<true, 0, false, true>(a16, pm, 64, ro3 + 4 * half, lo_b, rlo, rv[pt], false, a.fast3 == 2);
        else out = requant_tile16<false, 0, false>(a16, pm, 64, ro3 + 4 * half, lo_b, rlo, nores, a.dbl3 != 0, a.fast3 == 2);
      }
      const int p = 32 * pt + (lane & 31);
      if (AVG && a.avg_mult) {
        // per-channel sum over this tile's live columns (lanes 0-31 and 32-63 hold different channels: reduce inside a half)
#pragma unroll
        for (int q = 0; q < 16; q++) {
          int v = p < NPX ? (int)(signed char)(((unsigned)out[q >> 2] >> (8 * (q & 3))) & 0xff) : 0;
#pragma unroll
          for (int mm = 16; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm, 64);
          sum16[q] += v;
        }
      } else if (p < NPX) {
        bg_store_x(a.y + (px_img + p) * a.y_cp + a.y_off + ch3 + 16 * half, out, local2 || last);       // (an inner output is exchange data)
      }
    }
    if (AVG && a.avg_mult && (lane & 31) == 0) {
      unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int sv = (int)(short)sum16[q];                                         // int16 accumulator wrap (types.h:30)
        int mv = (((sv * a.avg_mult) >> 14) + 1) >> 1;                               // full_size_pool.cl:115-118
        mv = mv > 127 ? 127 : (mv < -128 ? -128 : mv);
        o[q >> 2] |= (unsigned)(mv & 0xff) << (8 * (q & 3));
      }
      *reinterpret_cast<i32x4*>(a.y + (size_t)img * a.y_cp + a.y_off + ch3 + 16 * half) = i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
    }
  }
  BG_STAMP(10);
#undef BG_STAMP
  if (!last) bg_signal(c.b[kb + 1].ctr + (size_t)img * 32, m, tag, tid);      // stores acknowledged, LDS free, then "my output is complete"
  }
}

size_t conv_bgroup_lds_bytes(int HW, int C, int M) {
  if (HW == 7) return 6 * (size_t)kBgHdrSlot + 96 * 1024 + 48 * 1024 + 64;
  if (HW == 28) return 4 * 4096 + 72 * 1024 + 7 * 5 * 2048 + 64;
  const int KS2 = M / 64;
  return 4 * (size_t)kBgHdrSlot + (size_t)9 * KS2 * 2048 + (size_t)KS2 * 256 * 64 + 64 + 64;     // + the control words behind the work region
}

bool conv_bgroup_shape_ok(int HW, int C, int M) {
  return (HW == 14 && C == 1024 && M == 256) || (HW == 7 && C == 2048 && M == 512) || (HW == 28 && C == 512 && M == 128);
}

// the first bottleneck of the 56 x 56 stage (rows: shortcut, reduce, 3x3, expand): conv_bgroup56f_kernel
int launch_conv_bgroup_first(const BGroupArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = 10 * (size_t)kBgHdrSlot + 36 * 1024 + 36 * 1024 + 2 * 13 * 2048 + 64;
  const void* fn = a.dual1 ? reinterpret_cast<const void*>(conv_bgroup56f_kernel<true>) : reinterpret_cast<const void*>(conv_bgroup56f_kernel<false>);
  if (!lds_attr_once(fn)) return -1;
  if (lds > 160 * 1024) return -3;
  for (int i0 = 0; i0 < a.B; i0 += 32) {
    BGroupArgs b = a;
    b.img0 = i0;
    const int n = std::min(32, a.B - i0);
    const dim3 grid(kBgMembers * ((n + 7) / 8 * 8));
    TF2_LAUNCH_NAME("conv_bgroup56f_kernel<56x56,shortcut | 64->64->64->256%s> (8 blocks per image, images %d..%d)", a.dual1 ? ",dual" : "", i0, i0 + n - 1);
    if (a.dual1) TF2_LAUNCH((conv_bgroup56f_kernel<true>), grid, dim3(512), lds, s, b);
    else TF2_LAUNCH((conv_bgroup56f_kernel<false>), grid, dim3(512), lds, s, b);
    if (!launch_ok()) return -1;
  }
  return 0;
}

int launch_conv_bgroup(const BGroupArgs* chain, int n_chain, int HW, int C, int M, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!conv_bgroup_shape_ok(HW, C, M)) return 1;
  if (n_chain < 1 || n_chain > kBgMaxChain) return 1;
  const BGroupArgs& a = chain[0];
  const BGroupArgs& z = chain[n_chain - 1];            // (7 x 7: the LAST bottleneck of a chain may end in the global average)
  for (int k = 1; k < n_chain; k++)
    if (chain[k].dual1 != a.dual1 || chain[k].dual2 != a.dual2 || chain[k - 1].avg_mult) return 1;
  const size_t lds = conv_bgroup_lds_bytes(HW, C, M);
  const void* fn = nullptr;
  if (HW == 14) fn = reinterpret_cast<const void*>(conv_bgroup_kernel<14, 1024, 256>);
  else if (HW == 28) fn = a.dual2 ? (a.dual1 ? reinterpret_cast<const void*>(conv_bgroup28_kernel<true, true>) : reinterpret_cast<const void*>(conv_bgroup28_kernel<false, true>))
                                  : (a.dual1 ? reinterpret_cast<const void*>(conv_bgroup28_kernel<true, false>) : reinterpret_cast<const void*>(conv_bgroup28_kernel<false, false>));
  else if (a.dual1) fn = z.avg_mult ? reinterpret_cast<const void*>(conv_bgroup7_kernel<true, true>) : reinterpret_cast<const void*>(conv_bgroup7_kernel<true, false>);
  else fn = z.avg_mult ? reinterpret_cast<const void*>(conv_bgroup7_kernel<false, true>) : reinterpret_cast<const void*>(conv_bgroup7_kernel<false, false>);
  if (!lds_attr_once(fn)) return -1;
  if (lds > 160 * 1024) return -3;
  // at most 32 images = 256 blocks = one block per CU per launch: every group is resident from the start, and the blocks of a
  // launch that fits the chip go to XCD (block % 8) -- larger grids were seen to place late blocks elsewhere (tools/bgroup_stress.py)
  for (int i0 = 0; i0 < a.B; i0 += 32) {
    BGroupArgs b = a;
    b.img0 = i0;
    const int n = std::min(32, a.B - i0);
    const dim3 grid(kBgMembers * ((n + 7) / 8 * 8));
    if (n_chain > 1) TF2_LAUNCH_NAME("conv_bgroup%s_kernel<%dx%d,C%d,M%d%s%s%s> x %d bottlenecks (8 blocks per image, images %d..%d)", HW == 7 ? "7" : HW == 28 ? "28" : "", HW, HW, C, M,
                                     (HW != 14 && a.dual1) ? ",dual reduce" : "", a.dual2 ? ",dual 3x3" : "", z.avg_mult ? ",global average" : "", n_chain, i0, i0 + n - 1);
    else TF2_LAUNCH_NAME("conv_bgroup%s_kernel<%dx%d,C%d,M%d%s%s%s> (8 blocks per image, images %d..%d)", HW == 7 ? "7" : HW == 28 ? "28" : "", HW, HW, C, M,
                    (HW != 14 && a.dual1) ? (a.dual2 ? ",dual reduce,dual 3x3" : ",dual reduce") : (a.dual2 ? ",dual 3x3" : ""), a.dual3 ? ",dual expand" : "",
                    a.avg_mult ? ",global average" : "", i0, i0 + n - 1);
    BGroupChain c;
    c.n = n_chain;
    for (int k = 0; k < n_chain; k++) { c.b[k] = chain[k]; c.b[k].img0 = i0; }
    if (HW == 14) TF2_LAUNCH((conv_bgroup_kernel<14, 1024, 256>), grid, dim3(512), lds, s, c);
    else if (HW == 28 && a.dual2 && a.dual1) TF2_LAUNCH((conv_bgroup28_kernel<true, true>), grid, dim3(512), lds, s, c);
    else if (HW == 28 && a.dual2) TF2_LAUNCH((conv_bgroup28_kernel<false, true>), grid, dim3(512), lds, s, c);
    else if (HW == 28 && a.dual1) TF2_LAUNCH((conv_bgroup28_kernel<true, false>), grid, dim3(512), lds, s, c);
    else if (HW == 28) TF2_LAUNCH((conv_bgroup28_kernel<false, false>), grid, dim3(512), lds, s, c);
    else if (a.dual1 && z.avg_mult) TF2_LAUNCH((conv_bgroup7_kernel<true, true>), grid, dim3(512), lds, s, c);
    else if (a.dual1) TF2_LAUNCH((conv_bgroup7_kernel<true, false>), grid, dim3(512), lds, s, c);
    else if (z.avg_mult) TF2_LAUNCH((conv_bgroup7_kernel<false, true>), grid, dim3(512), lds, s, c);
    else TF2_LAUNCH((conv_bgroup7_kernel<false, false>), grid, dim3(512), lds, s, c);
    if (!launch_ok()) return -1;
  }
  return 0;
}

}  // namespace tf2
