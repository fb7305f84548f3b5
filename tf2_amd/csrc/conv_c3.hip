// conv_c3.hip -- 3x3 / stride 1 / pad 1 convolution of a big map (VGG16 / SSD300's body: C, M in 64..512, 14 x 14 .. 300 x 300) with
// the input's halo tile resident in LDS (gfx950).
//
// Why: the ring kernel (conv_mfma2.hip) gathers every tap's activation slab again -- nine LDS-DMA fetches of the same input bytes
// per output tile, 24 KB of L2 -> LDS traffic per 64-byte K slab of a block -- and on these layers (12 of VGG16's 16 launches,
// 1.0 of its 1.17 ms) it runs at 0.19 of the dense int8 peak.  conv_bband's 3x3 phase does the same arithmetic from a halo tile in
// LDS at 95 % of the matrix pipe's rate; this kernel is that phase as a layer of its own:
//
//   * a block owns a TH x TW pixel tile (<= 256 pixels = 8 column tiles of 32) of ONE image and TMK = 64 / 128 / 256 output channels
//     (grid.y = channel groups); its 8 waves are (row tiles) x (pixel columns), one row tile (two with TMK = 256: one-window layers)
//     and 8 / WN column tiles each; 128-channel blocks (and the one-slab kernel below) walk several tiles, the chunk stream running
//     on across tiles;
//   * the input streams through LDS in chunks of SC 64-channel slabs of the (TH + 2) x (TW + 2) HALO tile (zero border = the stored
//     form of x = 0, sequencer.cl:287), two chunk buffers, LDS-DMA; a slab is four PLANES of 16 bytes per pixel (conv_bband.hip):
//     a lane's MFMA fragment of halo pixel h is 16 bytes at plane[half + 2 ks] + 16 h -- conflict-free without a swizzle, and tap,
//     slab and K half are immediate offsets of the ds_read;
//   * every chunk is swept by all nine taps before the next one is touched: the input is fetched ONCE (plus the halo: 1.3-1.5 x);
//   * weights go global -> registers, a lane's fragment is 16 contiguous bytes of its row in the packed tile, two steps ahead
//     (three rotating buffers: nine steps per slab keep the rotation aligned with the run-time chunk loop); one-slab 64-channel layers
//     hold all nine fragments in registers instead (conv_c3_w9_kernel: no load in the K loop, the input three tiles ahead);
//   * two-window layers keep two accumulator sets over the one input stream and combine them once, (hi << dshift[1]) + lo;
//   * epilogue: requant_tiles16_rows (parameter rows read once per row tile), 16-byte NHWC stores.
//
// Arithmetic, packed image and epilogue are conv_mfma2.hip's (reference: pe.cl:27-43 shift-accumulate, pe.cl:185-203 requant,
// relu.cl:54); bit-identical to it (tests/test_gpu_parity.py runs both forms of every eligible layer).
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

#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int T, int N, class F>
__device__ __forceinline__ void c3_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); c3_static_for<T + 1, N>(fn); }
}

// LDS-DMA hidden from the compiler's wait-count pass (conv_bband.hip bb_dma16): every wait for these is written out below
__device__ __forceinline__ void c3_dma16(const int8_t* src, int8_t* lds_dst) {
  const unsigned l = (unsigned)(unsigned long long)TF2_LDS_PTR(lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(l) : "memory", "m0");
}

#ifdef TF2_CHECK_DMA
TF2_DMA_CHECK_COUNTERS(g_c3_dma_check);
void conv_c3_check_counts(unsigned long long out[2]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c3_dma_check), 16); }
#endif

// ---- the layer's 2x2 / stride 2 / pad 0 max pool inside the launch (round 5; pool.cl:152-260 + pool_tail.cl:91-216 in the reference) --
// Tiles of pooled layers are TH x 32 pixels (host: conv_c3_pick_tile_pool), so that a 32-pixel column tile of the MFMA layout IS one
// tile row: a wave owns CONSECUTIVE tile rows (2 j, 2 j + 1 meet in its registers), the window's other column is the neighbouring lane
// (one DPP quad permute).  After ReLU every byte is 0..127: the byte-wise maximum is two v_pk_max_u16 on the even / odd bytes, and a
// tap outside the map (ceil-mode pools: SSD300's pool3) counts as 0 = the identity (pool.cl:119-140, the S < 3 window slots).  The
// conv map is neither written nor read back (VGG16 at batch 32: 196 MB written + 245 MB through maxpool_kernel, five launches).
struct C3Split { unsigned e[4], o[4]; };                  // a 16-byte vector's even / odd bytes as 16-bit lanes
__device__ __forceinline__ C3Split c3_split(const i32x4& v, bool keep) {
  C3Split r;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const unsigned x = keep ? (unsigned)v[q] : 0u;
    r.e[q] = x & 0x00ff00ffu; r.o[q] = (x >> 8) & 0x00ff00ffu;
  }
  return r;
}
// rows tr / tr + 1 of one pixel column (this lane's tc) -> the 2x2 window maximum of the lane pair (tc, tc ^ 1), valid in both lanes
__device__ __forceinline__ i32x4 c3_pool2x2(const i32x4& top, const i32x4& bot, bool bot_ok, bool self_ok, bool right_ok) {
  const C3Split a = c3_split(top, self_ok), b = c3_split(bot, self_ok && bot_ok);
  i32x4 out;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const unsigned ve = pk_max_u16(a.e[q], b.e[q]), vo = pk_max_u16(a.o[q], b.o[q]);
    // the neighbouring lane's column maxima (quad_perm [1, 0, 3, 2]); a lane whose own column lies beyond the map contributed zeros
    unsigned ne = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ve, 0xb1, 0xf, 0xf, false);
    unsigned no = (unsigned)__builtin_amdgcn_update_dpp(0, (int)vo, 0xb1, 0xf, 0xf, false);
    if (!right_ok) { ne = 0; no = 0; }
    out[q] = (int)(pk_max_u16(ve, ne) | (pk_max_u16(vo, no) << 8));
  }
  return out;
}

constexpr int kC3HaloPx = 384;               // halo pixels of a tile, padded to whole 64-pixel DMA groups (host: c3_pick_tile)
constexpr int kC3PlaneB = kC3HaloPx * 16;    // bytes of one 16-byte plane of a slab
constexpr int kC3SlabB = 4 * kC3PlaneB;      // bytes of one 64-channel slab of the halo tile

// PF: steps the weight fragments are loaded ahead of their MFMAs (PF + 1 rotating buffers; PF + 1 divides the nine steps of a slab's
// chunk walk).  The VM counter retires in order, so a wave's first wait for a fragment loaded AFTER its share of a chunk's DMAs also
// waits for those DMAs: a chunk gets PF + 1 steps to land before its issuing wave stalls -- 0.75 us at PF = 2 against ~2 us of
// latency under load, every chunk.  PF = 5 where the registers allow it (one-window 128-channel blocks).
// MT: row tiles per wave (TMK = 256: two -- a B fragment then feeds two MFMAs, half the LDS reads per MFMA; one-window layers only,
// the accumulators of 2 x 4 tiles are 128 registers)
// PERSIST: a block walks several pixel tiles (the host caps the grid); else one tile per block
template <int TMK, int SC, bool DUAL, int PF, bool PERSIST, bool POOL = false>
__global__ __launch_bounds__(512, (TMK == 64 && !PERSIST) ? 4 : 2) void conv_c3_kernel(C3Args a) {
  constexpr int MT = TMK == 256 ? 2 : 1;
  static_assert(MT == 1 || !DUAL, "two row tiles per wave: one accumulator set only");
  constexpr int NBUF = PF + 1;
  static_assert((SC * 9) % NBUF == 0, "the buffer rotation must close over a chunk");
  constexpr int WM = TMK / (32 * MT), WN = 8 / WM, NT = 8, J = NT / WN;
  constexpr int NG = kC3HaloPx / 64;                       // DMA groups per plane
  constexpr int CHUNK = SC * kC3SlabB;
  constexpr int NSTEP = SC * 9;                            // steps (slab, tap) per chunk
  constexpr int F = DUAL ? 4 : 2;                          // fragment loads per step and wave
  __shared__ __attribute__((aligned(1024))) int8_t ring[2 * CHUNK];
  extern __shared__ __attribute__((aligned(16))) int8_t dyn[];          // header image of the block's channel group

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5;
  const int H = a.H, W = a.W, TH = a.TH, TW = a.TW, HC = TW + 2;
  const int KS = a.C >> 6;                                 // 64-channel slabs of the input
  const int NC = KS / SC;                                  // chunks
  const int tms = a.tm == 128 ? 7 : 6;
  const int hst = (DUAL ? 28 : 20) << tms;                 // bytes of one storage m-tile's header image
  // the bands of one image on one XCD (they share halo rows, and all of them the weights)
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int cb = blockIdx.y * TMK + wm * (32 * MT);        // this wave's first output channel
  // A block walks pixel tiles bid, bid + gridDim.x, ... of its channel group (the host caps the grid at what the chip holds at once):
  // the chunk stream runs on across tiles -- while the last chunk of a tile is multiplied, the first chunk of the NEXT tile lands in
  // the other buffer -- so only a block's first tile pays the DMA latency in the open (3.7 us per tile before: more than the 1.5 us
  // of MFMA work a 64-channel layer's tile has), and the header image is fetched once per block.
  const int n_units = PERSIST ? a.B * a.tiles_per_img : bid + 1;      // (one tile per block: the loops below run once)
  const int ustride = PERSIST ? (int)gridDim.x : 1;
  struct Tile { int r0, c0, rows, cols; long long img_px; };
  auto tile_of = [&](int unit) __attribute__((always_inline)) {
    Tile t;
    const int img = fast_div(unit, a.tpi_m, a.tpi_s);
    const int tile = unit - img * a.tiles_per_img;
    const int ty = fast_div(tile, a.tx_m, a.tx_s), tx = tile - ty * a.tiles_x;
    t.r0 = ty * TH; t.c0 = tx * TW;
    t.rows = (H - t.r0) < TH ? (H - t.r0) : TH; t.cols = (W - t.c0) < TW ? (W - t.c0) : TW;
    t.img_px = (long long)img * H * W;
    return t;
  };

  // ---- the producer side: the tile whose chunks are being fetched, this lane's input pixels of its DMA groups (byte offset into x,
  // or -1 for the zero border / outside the image)
  // (the lane's halo pixels are decoded again at every chunk issue -- a dozen VALU instructions per DMA group -- rather than held in
  //  registers across the K loop)
  const int n_halo = (TH + 2) * HC;
  int unit_p = bid, c_p = 0;                               // producer position: (tile, chunk) of the NEXT chunk to fetch
  // chunk c of tile `unit` -> ring buffer: wave w issues (slab w / 4 of the chunk, plane w % 4) for every group (SC = 1: waves 0..3)
  auto issue_chunk = [&](int unit, int c, int8_t* buf) __attribute__((always_inline)) {
    if (wave < SC * 4) {
      const Tile t = tile_of(unit);
      const int sl = wave >> 2, k = wave & 3;
      const int coff = (c * SC + sl) * 64 + k * 16;
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int hp = g * 64 + lane;
        const int hr = fast_div(hp, a.hc_m, a.hc_s), hc = hp - hr * HC;
        const int r = t.r0 - 1 + hr, cc = t.c0 - 1 + hc;
        const bool in = hp < n_halo && (unsigned)r < (unsigned)H && (unsigned)cc < (unsigned)W;
        const int8_t* src = in ? a.x + (t.img_px + (long long)r * W + cc) * a.x_cp + coff : a.zero2 + coff;
#ifdef TF2_CHECK_DMA
        dma_stamp(buf + sl * kC3SlabB + k * kC3PlaneB + g * 1024);
#endif
        c3_dma16(src, buf + sl * kC3SlabB + k * kC3PlaneB + g * 1024);
      }
    }
  };
  long long* const dbg = a.dbg ? a.dbg + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;     // tools/c3_timeline.py: 100 MHz wall clock
#define C3_STAMP(i) do { if (dbg && tid == 0) dbg[i] = (long long)wall_clock64(); } while (0)
  // fetch the producer's next chunk into `buf` (nothing once the block's last tile is complete) and advance
  auto produce = [&](int8_t* buf) __attribute__((always_inline)) {
    if (unit_p >= n_units) return;
    issue_chunk(unit_p, c_p, buf);
    if (++c_p == NC) { c_p = 0; unit_p += ustride; }
  };
  C3_STAMP(0);
  produce(ring);
  produce(ring + CHUNK);
  // header images of the block's channels (rows {bias | dbl, alpha, addend64} | lo | dshift[P]) by ordinary loads
  {
    const int per = (DUAL ? 7 : 5) << (tms - 2);            // 16-byte pieces per storage m-tile
    const int mt0 = (blockIdx.y * TMK) >> tms, n_mt = TMK > (1 << tms) ? TMK >> tms : 1;
    for (int i = tid; i < n_mt * per; i += 512) {
      const int mt = i / per, k = i - mt * per;
      const i32x4 v = *reinterpret_cast<const i32x4*>(reinterpret_cast<const int8_t*>(a.hdr) + (size_t)(mt0 + mt) * a.hdr_bytes + k * 16);
      *reinterpret_cast<i32x4*>(dyn + (size_t)mt * (per * 16) + k * 16) = v;
    }
  }

  // ---- weight fragments ------------------------------------------------------------------------------------------------------
  struct Afr { i32x4 k[MT][2]; };
  const unsigned a_lane_off = (unsigned)((lane & 31) * 64 + half * 16);
  const int wins = DUAL ? 2 : 1;
  const int nslab = 9 * KS;
  // step (chunk c, e = slab sl * 9 + tap t) -> packed slab t * KS + c * SC + sl
  auto load_a = [&](Afr& f, int c, int e, int win) __attribute__((always_inline)) {
    const int sl = e / 9, t = e - sl * 9;
    const int slab = t * KS + c * SC + sl;
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int ch = cb + i * 32;
      const int mt_w = ch >> tms, ro_w = ch & ((1 << tms) - 1);
      const int8_t* pu = a.w + (((((size_t)mt_w * nslab + slab) * wins + win) << tms) + ro_w) * 64;
      f.k[i][0] = *reinterpret_cast<const i32x4*>(pu + a_lane_off);
      f.k[i][1] = *reinterpret_cast<const i32x4*>(pu + a_lane_off + 32);
    }
  };
  Afr fb[NBUF], gb[DUAL ? NBUF : 1];                        // rotating buffers, static indices (hi window; DUAL: gb = lo window)
#define C3_BUF(v) fb[(v) % NBUF]
#define C3_BUFL(v) gb[DUAL ? (v) % NBUF : 0]
#pragma unroll
  for (int v = 0; v < PF; v++) {
    load_a(fb[v], v / (SC * 9) < NC ? v / (SC * 9) : 0, v % (SC * 9), 0);
    if constexpr (DUAL) load_a(gb[v], v / (SC * 9) < NC ? v / (SC * 9) : 0, v % (SC * 9), 1);
  }

  // ---- this lane's pixels: column tile (wn + j * WN), pixel p = tr * TW + tc of the tile ------------------------------------------
  int h0[J];
  int n_j = 0;                                             // column tiles of this wave that hold pixels of the tile
#pragma unroll
  for (int j = 0; j < J; j++) {
    const int pt = (POOL ? wn * J + j : wn + j * WN) * 32;   // (POOL: consecutive column tiles = consecutive tile rows per wave)
    if (pt < TH * TW) n_j = j + 1;
    int p = pt + (lane & 31);
    if (p >= TH * TW) p = 0;                               // lanes beyond the tile compute on pixel 0 and are never stored
    const int tr = fast_div(p, a.tw_m, a.tw_s), tc = p - tr * TW;
    h0[j] = (tr * HC + tc) * 16 + half * kC3PlaneB;        // tap (0, 0), K half 0
  }
  n_j = __builtin_amdgcn_readfirstlane(n_j);

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // chunks 0 and 1, header: complete in every wave
  asm volatile("" ::: "memory");
  C3_STAMP(1);

  // Everything from here on exists twice: for waves all of whose J column tiles hold pixels of the tile, and for waves whose LAST
  // one lies beyond it (4 x 56 and 7 x 28 pixel tiles fill 7 of 8 column tiles) -- chosen once per wave, so that the K loop is
  // straight-line code over exactly the tiles that count (a branch per column tile inside the loop put every ds_read right in front
  // of its MFMA behind a full wait: 58 % of the matrix rate; computing the empty tile costs an eighth of the MFMAs).  Both versions
  // pass the same barriers.
  auto run = [&](auto nj_c) __attribute__((always_inline)) {
  constexpr int NJ = decltype(nj_c)::value;
  int gc = 0;                                              // chunks consumed so far (ring buffer = gc & 1)
#pragma unroll 1
  for (int unit = bid; unit < n_units; unit += ustride) {
  const Tile T = tile_of(unit);
  const int r0 = T.r0, c0 = T.c0, rows = T.rows, cols = T.cols;
  const long long img_px = T.img_px;
  i32x16 acc[MT][NJ], acc2[DUAL ? NJ : 1];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int i = 0; i < MT; i++) acc[i][j][r] = 0;
      if (DUAL) acc2[j][r] = 0;
    }


  // ---- the K loop: chunks at run time, the NSTEP = SC * 9 steps of a chunk unrolled --------------------------------------------
#pragma unroll 1
  for (int c = 0; c < NC; c++, gc++) {
    const int rb = (gc & 1) * CHUNK;
    auto step = [&](auto e_c) __attribute__((always_inline)) {
      constexpr int e = decltype(e_c)::value;
      constexpr int sl = e / 9, t = e % 9;
      if constexpr (e == 0) {
        if (gc > 0) {
          // chunk c landed in every wave and nobody reads the other buffer any more.  This wave's DMAs of chunk c were issued at the
          // first step of chunk c - 1, BEHIND that step's fragment loads: younger than them are the fragment loads of that chunk's
          // other NSTEP - 1 steps, F each -- every step loads, whatever the chunk (past the last one: a valid address, never used) --
          // plus, across a tile boundary, the previous tile's output stores (VM operations as well: they only make the wait stricter).
          // Of those the compiler's own waits leave at most the last PF steps' in flight: no point in allowing more (vm_track.h;
          // tests/test_vmcnt_isa.py counts the same in the compiled code).
          constexpr int kSinceDma = (NSTEP - 1) * F;
          static_assert(PF * F <= kSinceDma, "a wait may never allow more than was issued behind the DMAs");
          vm_wait<vm_min(kSinceDma, PF * F)>();
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      Afr& cur = C3_BUF(e);
      // fragments of step e + PF (the next chunk's first steps at the end; past the last chunk: a valid address, never used)
      {
        constexpr int e2 = (e + PF) % NSTEP;
        const int c2 = (e + PF >= NSTEP) ? (c + 1 < NC ? c + 1 : 0) : c;      // (past a tile's last chunk: the next tile's chunk 0 -- the same weights)
        load_a(C3_BUF(e + PF), c2, e2, 0);
        if constexpr (DUAL) load_a(C3_BUFL(e + PF), c2, e2, 1);
      }
      if constexpr (e == 0) {
        if (gc > 0) {
          asm volatile("" ::: "memory");
          produce(ring + ((gc + 1) & 1) * CHUNK);           // (behind this step's fragment loads; the buffer chunk gc - 1 was read from)
        }
      }
      // (slab, tap) -> a wave-uniform byte offset; one address add per column tile, the K half is an immediate
      const int soff = rb + sl * kC3SlabB + ((t / 3) * HC + (t % 3)) * 16;
      int addr[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) addr[j] = h0[j] + soff;
      // (straight-line: a branch per column tile put every ds_read right in front of its MFMA behind a full wait -- 58 % of the
      //  matrix rate; a column tile beyond the pixel tile, 1 of 8 for a 4 x 56 / 7 x 28 tile, is computed on pixel 0 and dropped)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        i32x4 bf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) bf[j] = *reinterpret_cast<const i32x4*>(ring + addr[j] + 2 * ks * kC3PlaneB);
#ifdef TF2_CHECK_DMA
#pragma unroll
        for (int j = 0; j < NJ; j++) dma_check(bf[j], g_c3_dma_check);
#endif
#pragma unroll
        for (int j = 0; j < NJ; j++) {
#pragma unroll
          for (int i = 0; i < MT; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.k[i][ks], bf[j], acc[i][j], 0, 0, 0);
          if constexpr (DUAL) acc2[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(C3_BUFL(e).k[0][ks], bf[j], acc2[j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    c3_static_for<0, NSTEP>(step);
    if (gc < 8) C3_STAMP(2 + gc);
  }
  if (unit == bid) C3_STAMP(10);
#undef C3_BUF
#undef C3_BUFL

  // ---- epilogue ----------------------------------------------------------------------------------------------------------------
  const int lo_b = a.relu ? 0 : -128;
  auto finish = [&](auto fast_c) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const int ch = cb + i * 32;
      const int8_t* const hd = dyn + (size_t)((ch - blockIdx.y * TMK) >> tms) * hst;     // this row tile's storage m-tile inside the block's images
      const int* prm = reinterpret_cast<const int*>(hd);
      const int row0 = (ch & ((1 << tms) - 1)) + 4 * half;
      if constexpr (DUAL) {
        // (hi << dshift[1][row]) + lo: the two-window Horner result in Z/2^32; dshift sits behind rows | lo
        const int* dsh = prm + (6 << tms) + row0;
#pragma unroll
        for (int G = 0; G < 4; G++) {
          const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + 8 * G);
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
              acc[i][j][G * 4 + r] = (int)(((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31)) + (unsigned)acc2[j][G * 4 + r]);
        }
      }
      const int chl = ch + 16 * half;
      i32x4 outs[NJ];
      int a16s[NJ][16];
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) a16s[j][r] = acc[i][j][r];
      requant_tiles16_rows<NJ, FAST>([&](int j) -> const int (&)[16] { return a16s[j]; }, outs, prm, 1 << tms, row0, lo_b, a.dbl != 0, a.fast == 2);
      if constexpr (POOL) {
        // TW == 32: column tile (wn * J + j) is tile row tr = wn * J + j, lane & 31 the column; tile rows come in pairs (TH, r0 even)
        const int tc = lane & 31;
        const long long pimg = (long long)fast_div(unit, a.tpi_m, a.tpi_s) * a.PH * a.PW;      // pooled pixels in front of this image
#pragma unroll
        for (int jj = 0; jj < (NJ + 1) / 2; jj++) {
          const int tr = wn * J + 2 * jj;
          constexpr int kNJ = NJ;
          const int jb = 2 * jj + 1 < kNJ ? 2 * jj + 1 : 2 * jj;
          const i32x4 pv = c3_pool2x2(outs[2 * jj], outs[jb], 2 * jj + 1 < kNJ && tr + 1 < rows, tr < rows && tc < cols, tc + 1 < cols);
          if ((tc & 1) == 0 && tr < rows && tc < cols && chl + 16 <= a.y_nvalid)
            *reinterpret_cast<i32x4*>(a.y + (size_t)(pimg + (long long)((r0 + tr) >> 1) * a.PW + ((c0 + tc) >> 1)) * a.y_cp + a.y_off + chl) = pv;
        }
      } else {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int p = (wn + j * WN) * 32 + (lane & 31);
        const int tr = fast_div(p, a.tw_m, a.tw_s), tc = p - tr * TW;
        if (tr < rows && tc < cols && chl + 16 <= a.y_nvalid)
          *reinterpret_cast<i32x4*>(a.y + (size_t)(img_px + (long long)(r0 + tr) * W + c0 + tc) * a.y_cp + a.y_off + chl) = outs[j];
      }
      }
    }
  };
  if (a.fast == 1) finish(std::true_type{}); else finish(std::false_type{});
  }
  };
  if (n_j >= J) run(std::integral_constant<int, J>{}); else run(std::integral_constant<int, (J > 1 ? J - 1 : 1)>{});
  C3_STAMP(11);
#undef C3_STAMP
}

// ---- one-slab layers (C = 64, one window, 64 output channels: VGG16's conv1_2, SSD300's) -------------------------------------------------
// A tile of such a layer is 1.1 us of MFMA work behind the latency of its own input: in conv_c3_kernel the next tile's DMAs have one
// tile's time to land, and issuing them earlier is pointless while weight fragments are loaded inside the loop (the VM counter
// retires in order).  Here the nine weight fragments (72 registers) are loaded ONCE per block, the K loop holds no load at all, and
// the input runs THREE tiles ahead through four 24 KiB slab buffers: the only counted wait is on the DMAs themselves (a tile's output
// stores, younger, only make it stricter).  One block per CU walks every 256th tile.
template <bool POOL>
__global__ __launch_bounds__(512, 2) void conv_c3_w9_kernel(C3Args a) {
  constexpr int WM = 2, WN = 4, J = 2, NB = 4, NG = kC3HaloPx / 64;
  __shared__ __attribute__((aligned(1024))) int8_t ring[NB * kC3SlabB];
  extern __shared__ __attribute__((aligned(16))) int8_t dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5;
  const int H = a.H, W = a.W, TH = a.TH, TW = a.TW, HC = TW + 2;
  const int tms = a.tm == 128 ? 7 : 6;
  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int cb = blockIdx.y * 64 + wm * 32;
  const int n_units = a.B * a.tiles_per_img, ustride = gridDim.x;
  struct Tile { int r0, c0, rows, cols; long long img_px; };
  auto tile_of = [&](int unit) __attribute__((always_inline)) {
    Tile t;
    const int img = fast_div(unit, a.tpi_m, a.tpi_s);
    const int tile = unit - img * a.tiles_per_img;
    const int ty = fast_div(tile, a.tx_m, a.tx_s), tx = tile - ty * a.tiles_x;
    t.r0 = ty * TH; t.c0 = tx * TW;
    t.rows = (H - t.r0) < TH ? (H - t.r0) : TH; t.cols = (W - t.c0) < TW ? (W - t.c0) : TW;
    t.img_px = (long long)img * H * W;
    return t;
  };
  const int n_halo = (TH + 2) * HC;
  int unit_p = bid, gp = 0, n_issued = 0;                  // producer: next tile to fetch, produce() calls so far, tiles really fetched
  auto produce = [&]() __attribute__((always_inline)) {    // waves 0..3: plane `wave` of the tile's slab, six DMA groups each
    if (unit_p < n_units) {
      if (wave < 4) {
        const Tile t = tile_of(unit_p);
        int8_t* const buf = ring + (gp % NB) * kC3SlabB + wave * kC3PlaneB;
#pragma unroll
        for (int g = 0; g < NG; g++) {
          const int hp = g * 64 + lane;
          const int hr = fast_div(hp, a.hc_m, a.hc_s), hc = hp - hr * HC;
          const int r = t.r0 - 1 + hr, cc = t.c0 - 1 + hc;
          const bool in = hp < n_halo && (unsigned)r < (unsigned)H && (unsigned)cc < (unsigned)W;
          const int8_t* src = in ? a.x + (t.img_px + (long long)r * W + cc) * a.x_cp + wave * 16 : a.zero2 + wave * 16;
#ifdef TF2_CHECK_DMA
          dma_stamp(buf + g * 1024);
#endif
          c3_dma16(src, buf + g * 1024);
        }
      }
      unit_p += ustride;
      n_issued++;
    }
    gp++;
  };
  produce(); produce(); produce();
  // header image of the block's 64 channels
  {
    const int per = 5 << (tms - 2);
    const int mt0 = (blockIdx.y * 64) >> tms;
    for (int i = tid; i < per; i += 512)
      *reinterpret_cast<i32x4*>(dyn + i * 16) = *reinterpret_cast<const i32x4*>(reinterpret_cast<const int8_t*>(a.hdr) + (size_t)mt0 * a.hdr_bytes + i * 16);
  }
  // the nine taps' weight fragments of this wave's row tile
  i32x4 fw[9][2];
  {
    const unsigned a_lane_off = (unsigned)((lane & 31) * 64 + half * 16);
    const int mt_w = cb >> tms, ro_w = cb & ((1 << tms) - 1);
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int8_t* pu = a.w + ((((size_t)mt_w * 9 + t) << tms) + ro_w) * 64;
      fw[t][0] = *reinterpret_cast<const i32x4*>(pu + a_lane_off);
      fw[t][1] = *reinterpret_cast<const i32x4*>(pu + a_lane_off + 32);
    }
  }
  int h0[J];
  int n_j = 0;
#pragma unroll
  for (int j = 0; j < J; j++) {
    const int pt = (POOL ? wn * J + j : wn + j * WN) * 32;  // (POOL: a wave's two column tiles are two consecutive tile rows)
    if (pt < TH * TW) n_j = j + 1;
    int p = pt + (lane & 31);
    if (p >= TH * TW) p = 0;
    const int tr = fast_div(p, a.tw_m, a.tw_s), tc = p - tr * TW;
    h0[j] = (tr * HC + tc) * 16 + half * kC3PlaneB;
  }
  n_j = __builtin_amdgcn_readfirstlane(n_j);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // tiles 0..2, header, fragments
  asm volatile("" ::: "memory");

  const int* prm = reinterpret_cast<const int*>(dyn);      // (64-channel blocks: at most one storage m-tile's image, rows by position)
  const int row0 = (cb & ((1 << tms) - 1)) + 4 * half;
  const int lo_b = a.relu ? 0 : -128;
  const int chl = cb + 16 * half;
  auto run = [&](auto nj_c) __attribute__((always_inline)) {
    constexpr int NJ = decltype(nj_c)::value;
    int g = 0;
#pragma unroll 1
    for (int unit = bid; unit < n_units; unit += ustride, g++) {
      if (g > 0) {
        // tile g's DMAs were issued three tiles ago; younger: the DMAs of the tiles REALLY fetched since (NG per issuing wave and tile:
        // two tiles, fewer at the end of the block's walk -- counted by produce(), not assumed) and the output stores in between --
        // "at most NG x (tiles fetched since) outstanding" therefore covers every DMA of tile g, whatever the stores did (they only make
        // the wait stricter; vm_track.h)
        vm_wait_groups<NG, 2>(n_issued - (g + 1));
        __builtin_amdgcn_s_barrier();                      // ... in every issuing wave; and nobody reads tile g - 1's buffer any more
        asm volatile("" ::: "memory");
      }
      produce();                                           // tile g + 3 into tile g - 1's buffer
      const Tile T = tile_of(unit);
      i32x16 acc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0;
      const int rb = (g % NB) * kC3SlabB;
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int soff = rb + ((t / 3) * HC + (t % 3)) * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
          i32x4 bf[NJ];
#pragma unroll
          for (int j = 0; j < NJ; j++) bf[j] = *reinterpret_cast<const i32x4*>(ring + h0[j] + soff + 2 * ks * kC3PlaneB);
#ifdef TF2_CHECK_DMA
#pragma unroll
          for (int j = 0; j < NJ; j++) dma_check(bf[j], g_c3_dma_check);
#endif
#pragma unroll
          for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fw[t][ks], bf[j], acc[j], 0, 0, 0);
        }
      }
      auto finish = [&](auto fast_c) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_c)::value;
        i32x4 outs[NJ];
        int a16s[NJ][16];
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) a16s[j][r] = acc[j][r];
        requant_tiles16_rows<NJ, FAST>([&](int j) -> const int (&)[16] { return a16s[j]; }, outs, prm, 1 << tms, row0, lo_b, a.dbl != 0, a.fast == 2);
        if constexpr (POOL) {
          const int tc = lane & 31, tr = wn * J;
          const long long pimg = (long long)fast_div(unit, a.tpi_m, a.tpi_s) * a.PH * a.PW;
          constexpr int kNJ = NJ;
          const i32x4 pv = c3_pool2x2(outs[0], outs[kNJ > 1 ? 1 : 0], kNJ > 1 && tr + 1 < T.rows, tr < T.rows && tc < T.cols, tc + 1 < T.cols);
          if ((tc & 1) == 0 && tr < T.rows && tc < T.cols && chl + 16 <= a.y_nvalid)
            *reinterpret_cast<i32x4*>(a.y + (size_t)(pimg + (long long)((T.r0 + tr) >> 1) * a.PW + ((T.c0 + tc) >> 1)) * a.y_cp + a.y_off + chl) = pv;
        } else {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int p = (wn + j * WN) * 32 + (lane & 31);
          const int tr = fast_div(p, a.tw_m, a.tw_s), tc = p - tr * TW;
          if (tr < T.rows && tc < T.cols && chl + 16 <= a.y_nvalid)
            *reinterpret_cast<i32x4*>(a.y + (size_t)(T.img_px + (long long)(T.r0 + tr) * W + T.c0 + tc) * a.y_cp + a.y_off + chl) = outs[j];
        }
        }
      };
      if (a.fast == 1) finish(std::true_type{}); else finish(std::false_type{});
    }
  };
  if (n_j >= J) run(std::integral_constant<int, J>{}); else run(std::integral_constant<int, 1>{});
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// the pixel tile of a map: TW columns (the whole width up to 62, else the width in equal parts of <= 62), TH rows such that the
// tile has <= 256 pixels and its halo <= 384, rows spread evenly over the tiles of a column
bool conv_c3_pick_tile(int H, int W, int* TH, int* TW) {
  if (H < 1 || W < 1) return false;
  const int nx = (W + 61) / 62;
  const int tw = (W + nx - 1) / nx;
  int th = std::min(H, 256 / tw);
  while (th > 1 && (th + 2) * (tw + 2) > kC3HaloPx) th--;
  if (th < 1 || (th + 2) * (tw + 2) > kC3HaloPx) return false;
  const int ny = (H + th - 1) / th;
  th = (H + ny - 1) / ny;
  *TH = th; *TW = tw;
  return true;
}

// tiles of a layer whose 2x2 / 2 pool rides in the launch: 32 columns (one column tile of the MFMA layout = one tile row), 8 rows;
// taken where the 32-column tiling wastes at most 15 % of the pixels (28, 56, 112, 150, 224, 300 wide maps; not 38 or 75)
bool conv_c3_pick_tile_pool(int H, int W, int* TH, int* TW) {
  if (H < 2 || W < 28) return false;
  const int nx = (W + 31) / 32;
  if ((long)nx * 32 * 100 > (long)W * 115) return false;
  *TW = 32; *TH = H < 8 ? (H + 1) / 2 * 2 : 8;
  return (*TH + 2) * (*TW + 2) <= kC3HaloPx;
}

bool conv_c3_shape_ok(int H, int W, int C, int Np, int min_hw) {
  int th, tw;
  if (C % 64 != 0 || C < 64 || C > 1024 || Np % 64 != 0) return false;
  if (H < min_hw || W < min_hw) return false;              // (policy, Net::c3_at: 14 -- on small maps the ring / split-K kernels stay; the launcher itself takes any map >= 3 x 3)
  return conv_c3_pick_tile(H, W, &th, &tw);
}

// CUs of the current device (256 when only describing a launch, without a device)
static int tf2_cu_count() {
  if (launch_recorder()) return 256;
  static thread_local int cached_dev = -1, n = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev != cached_dev) {
    hipDeviceProp_t prop;
    n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    cached_dev = dev;
  }
  return n;
}

template <int TMK, int SC, bool DUAL, bool PERSIST>
static int launch_c3p(const C3Args& a, hipStream_t s) {
  constexpr int PF = (TMK == 128 && !DUAL && SC == 2) ? 5 : 2;
  if constexpr (TMK == 256 && DUAL) return 1; else {
  const size_t stat = (size_t)2 * SC * kC3SlabB;
  const int tms = a.tm == 128 ? 7 : 6;
  const size_t dyn = (size_t)(TMK > a.tm ? TMK / a.tm : 1) * ((DUAL ? 28 : 20) << tms);
  if (stat + dyn > 160 * 1024) return 1;
  auto fn = a.pool ? conv_c3_kernel<TMK, SC, DUAL, PF, PERSIST, true> : conv_c3_kernel<TMK, SC, DUAL, PF, PERSIST, false>;
  if (a.pool && (a.TW != 32 || (a.TH & 1))) return 1;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn), 160 * 1024 - (int)stat)) return -1;
  TF2_LAUNCH_NAME("conv_c3_kernel<%d channels x %dx%d pixels per block,C%d,%d slabs per chunk%s%s>", TMK, a.TH, a.TW, a.C, SC, DUAL ? ",dual" : "", a.pool ? ",2x2 pool" : "");
  // as many blocks per channel group as the chip holds at once (TMK = 64: two per CU), each walking every grid-th tile
  const int n_units = a.B * a.tiles_per_img, groups = a.M / TMK;
  const int resident = ((TMK == 64 && !PERSIST) ? 2 : 1) * tf2_cu_count();
  int gx = std::max(1, resident / groups);
  if (gx > n_units || !PERSIST) gx = n_units;
  if (a.dbg) gx = n_units;                                  // (timeline tool: one tile per block)
  TF2_LAUNCH(fn, dim3(gx, groups), dim3(512), dyn, s, a);
  return launch_ok() ? 0 : -1;
  }
}

// the several-tiles form: 128-channel blocks always (no spill); 64-channel blocks where a block would otherwise see a long row of
// short tiles (one K slab: 1.5 us of MFMA work behind a 3.7 us prologue each) -- then at two waves per SIMD, one block per CU;
// 256-channel blocks never (they spill in that form: experiment 20)
template <int TMK, int SC, bool DUAL>
static int launch_c3(const C3Args& a, hipStream_t s) {
  if constexpr (TMK == 128) return launch_c3p<TMK, SC, DUAL, true>(a, s);
  else if constexpr (TMK == 64) {
    const long tiles_per_block = (long)a.B * a.tiles_per_img * (a.M / 64) / std::max(1, tf2_cu_count());
    if (a.C == 64 && tiles_per_block >= 8 && !a.dbg) return launch_c3p<TMK, SC, DUAL, true>(a, s);
    return launch_c3p<TMK, SC, DUAL, false>(a, s);
  } else return launch_c3p<TMK, SC, DUAL, false>(a, s);
}

static int launch_c3_w9(const C3Args& a, hipStream_t s) {
  const size_t stat = (size_t)4 * kC3SlabB;
  const int tms = a.tm == 128 ? 7 : 6;
  const size_t dyn = (size_t)(20 << tms);
  if (stat + dyn > 160 * 1024) return 1;
  auto fn = a.pool ? conv_c3_w9_kernel<true> : conv_c3_w9_kernel<false>;
  if (a.pool && (a.TW != 32 || (a.TH & 1))) return 1;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn), 160 * 1024 - (int)stat)) return -1;
  const int n_units = a.B * a.tiles_per_img, groups = a.M / 64;
  int gx = std::max(1, tf2_cu_count() / groups);
  if (gx > n_units) gx = n_units;
  TF2_LAUNCH_NAME("conv_c3_w9_kernel<64 channels x %dx%d pixels per tile,C64,weights resident%s>", a.TH, a.TW, a.pool ? ",2x2 pool" : "");
  TF2_LAUNCH(fn, dim3(gx, groups), dim3(512), dyn, s, a);
  return launch_ok() ? 0 : -1;
}

// the one-slab kernel: mode 0 never, 1 (default) where a block walks at least eight tiles, 2 wherever the layer allows it (tests).
// Decided when the launch plan is built (C3Args::w9): tf2_net_describe_launches and every later step name the same kernel.
bool conv_c3_takes_w9(const C3Args& a, int mode) {
  return mode && a.tmk == 64 && a.C == 64 && !a.dual && !a.dbg && conv_c3_shape_ok(a.H, a.W, a.C, a.M, 3) &&
         (mode == 2 || (long)a.B * a.tiles_per_img * (a.M / 64) / std::max(1, tf2_cu_count()) >= 8);
}

int launch_conv_c3(const C3Args& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.w9) return launch_c3_w9(a, s);
  if (!conv_c3_shape_ok(a.H, a.W, a.C, a.M, 3) || (a.tm != 64 && a.tm != 128)) return 1;
  const int ks = a.C / 64;
  if ((a.tmk != 64 && a.tmk != 128 && a.tmk != 256) || a.M % a.tmk != 0 || (a.tmk == 256 && a.dual)) return 1;
  const bool m128 = a.tmk == 128;
#define TF2_C3(TMK_, SC_) do { return a.dual ? launch_c3<TMK_, SC_, true>(a, s) : launch_c3<TMK_, SC_, false>(a, s); } while (0)
  if (a.tmk == 256) { if (ks % 2 == 0) return launch_c3<256, 2, false>(a, s); return launch_c3<256, 1, false>(a, s); }
  if (!m128) TF2_C3(64, 1);                                // (64-channel blocks: one slab per chunk, 48 KiB of ring -- two blocks share a CU)
  if (ks % 2 == 0) TF2_C3(128, 2);
  TF2_C3(128, 1);
#undef TF2_C3
}

}  // namespace tf2
