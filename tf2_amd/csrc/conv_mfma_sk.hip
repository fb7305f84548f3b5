// conv_mfma_sk.hip -- implicit-GEMM INT8 convolution for SMALL GRIDS: in-block split-K (gfx950).
//
// Same arithmetic, packed image, per-m-tile LDS header and epilogue as conv_mfma2.hip.  When a
// layer has few output pixels (7x7 / 14x14 maps at small batch) but a long K (3x3 over 256-512
// channels, 1x1 over 1024-2048), conv_mfma2's grid cannot fill the 256 CUs and every block
// walks its 36-72 K slabs serially (~0.4 us each: the whole layer takes as long as one block).
// Here the four waves of a block compute the SAME 64-channel x 64-pixel tile over interleaved
// quarters of the slab list (wave w takes entries w, w+4, ...), each with its own LDS-DMA ring
// and its own counted vmcnt -- no block barrier inside the K loop -- and the four int32 partial
// tiles are summed through LDS at the end.  Exactness: the accumulator of the reference lives
// in Z/2^32 (pe.cl:43), where the Horner-combined partial sums of disjoint slab subsets simply
// add; each wave applies the phase shifts to its own partial sum.
//
// DUAL layers (weight_pack.cpp: both exponent windows per entry) are walked as 2 * n_ent virtual entries
// v = 2 * e + h: wave w takes v = w, w + 4, ... so h = w & 1 is fixed per wave -- waves 0 and 2 accumulate the high
// window, waves 1 and 3 the low one, nobody shifts inside the loop, and the reduction is
// ((p0 + p2) << dshift[1]) + (p1 + p3).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "tf2_internal.h"
#include "tf2_device.h"
#include "requant_epilogue.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void sk_wait_vmcnt() {
  static_assert(N == 0 || N == 1 || N == 4 || N == 8, "vmcnt immediate");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// NWV = 4 or 8 waves split the slab list; with 8 the grid of a 7x7 / 14x14 layer at batch 32 (56..392 blocks) puts
// twice as many waves on the chip and every wave walks half as many K steps; waves 0-3 run the epilogue.
// DENSE (ConvArgs::dense, see conv_mfma2.hip): gather words from the slab index, entry range from the kernel arguments -- the
// activation DMAs of the first stages go out together with the weight DMAs instead of behind the header's landing.
// AVG (ConvGeom::avg_mult != 0, four waves): the layer ends in the reference's full_size_pool stage (full_size_pool.cl:95-125:
// int16 sum over the map, x 669 >> 14, round, clamp).  A pixel tile is then ONE image's OHW <= 64 pixels (columns beyond
// OHW are dead), the requantised + residual-added outputs are summed over the valid pixels in the wave (__shfl_xor over the 32
// lanes of a half), the two pixel halves meet in LDS, and the block stores its 64 averaged channels of that image -- the conv
// map itself never reaches memory and the global_avg_kernel launch disappears.
// The body is a device function of (argument block, block index, grid size), as in conv_mfma2.hip: ONE launch can carry two independent
// rows of the same instantiation (conv_mfma_sk_pair_kernel below, round 6: at batch 1 the shortcut convolution of stages 4 and 5 and the
// first 1x1 of the stage's first bottleneck are both split-K launches of a few dozen blocks -- one launch boundary less each).
template <int S, bool PADCHK, bool DUAL, int NWV, bool DENSE, bool AVG, bool KSP = false>
__device__ __forceinline__ void conv_mfma_sk_body(const ConvArgs& a, const int blk_x, const int nblk_x) {
  static_assert(!KSP || (DENSE && !AVG), "K split over blocks: dense layers, no fused average");
  static_assert(!AVG || NWV == 4, "the fused global average runs with four waves");
  constexpr int TM = 64, TN = 64;
  constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, STAGE = A_BYTES + B_BYTES;   // per wave: 8 KiB
  constexpr int AI = 4, BI = 4, NI = 8;        // LDS-DMA instructions per wave per stage
  constexpr int RING = S * STAGE;              // per wave
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  // LDS map: [4 per-wave rings (reused for the 64 KiB reduction)][header as in conv_mfma2.hip]
  constexpr int RING_ALL = (NWV * RING > NWV * 16384) ? NWV * RING : NWV * 16384;
  int* const prm = reinterpret_cast<int*>(lds + RING_ALL);

  TF2_PRELOAD_CONV_ARGS(a);          // every kernel argument in SGPRs after two scalar-load round trips (tf2_device.h)
  TF2_PROBE_WORD(g.flags);
  if (prb & kProbeExit0) return;
  if ((prb & kProbeQuarterBlocks) && (blk_x & 3)) return;      // (probe: a quarter of the launch's footprint at the same block latency)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int* const dsh = prm + kPrmWordsPerRow * TM;
  int* const steps = dsh + P * TM;
  int* const goff = steps + a_max_ent;
  int* const ghw = goff + a_max_ent * 4;
  int8_t* const ring = lds + wave * RING;

  const int nblk = nblk_x;
  int bid = blk_x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  // KSP: ks_parts consecutive ids (one XCD, mostly) share an output tile
  int kpart = 0, kparts = 1;
  if (KSP) { kparts = a.ks_parts; const int t = bid / kparts; kpart = bid - t * kparts; bid = t; }
  const int tile_lin = bid;
  const int ntile = fast_div_u(bid, mt_m, mt_s);                 // bid / n_mtiles
  const int mtile = bid - ntile * a_n_mtiles;
  const int px0 = AVG ? ntile * g.OHW : ntile * TN;           // AVG: tile = image `ntile`, local pixels 0 .. OHW - 1
  const int px_end = AVG ? px0 + g.OHW : g.n_pix;             // first pixel this tile must not touch
  // this m-tile's {first, end} entry: the last two words of steps[] in its header image (weight_pack.cpp)
  typedef const __attribute__((address_space(4))) int __attribute__((ext_vector_type(2)))* cvec2_p;
  int e_begin, n_ent;
  int ent_off = 0;                                             // KSP: first slab of this block's part of the list
  if (DENSE) {
    n_ent = a_nslab; e_begin = ((mtile << a_e_shl) >> a_e_shr) * n_ent;
    if (KSP) { ent_off = kpart * n_ent / kparts; n_ent = (kpart + 1) * n_ent / kparts - ent_off; }
  }
  else {
    const auto ee = *(cvec2_p)(unsigned long long)(ahdr + (size_t)mtile * (size_t)(a_hdr_bytes >> 2) + kPrmWordsPerRow * TM + P * TM + a_max_ent - 2);
    e_begin = ee[0];
    n_ent = ee[1] - ee[0];
  }
  if (prb & kProbeExit1) { if (n_ent == 0x7eadbeef) ay[0] = 1; return; }
  if ((prb & kProbeQuarterK) && n_ent >= 16) n_ent >>= 2;                // (probe: a quarter of the K walk at the same footprint)
  const int n_virt = DUAL ? 2 * n_ent : n_ent;                         // DUAL: (entry, window) pairs
  const int n_mine = n_virt > wave ? (n_virt - wave + NWV - 1) / NWV : 0;     // (virtual) entries wave, wave+NWV, ...
  // list index of this wave's k-th item: entry, and for DUAL the fixed window h = wave & 1
  auto ent_of = [&](int k) { const int v = wave + NWV * k; return ent_off + (DUAL ? (v >> 1) : v); };

  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);             // see conv_mfma2.hip
  const int a_lane_off = (lane >> 2) * 64 + chunk * 16;
  const DenseGeom dg = {a_cslabs, a_cs_m, a_cs_s, a_k, a_kk_m, a_kk_s, a_dil, g.W, g.Cp_in};
  const int lane_c16 = chunk * 16;
  auto gather_of = [&](int sl, int& off, int& hw) {             // DENSE: this lane's gather words of slab sl (entry index == slab)
    int o, h;
    dense_gather(dg, sl, o, h);
    off = o + lane_c16; hw = h + (lane_c16 << 16);
  };

  auto issue_A = [&](int k, int slot_idx) {                     // k-th entry of this wave
    int8_t* const slot = ring + slot_idx * STAGE;
    // (64-row tiles: own storage, or the halves of the main entry's 128-row tiles -- ConvArgs w_*)
    const int8_t* wsrc = aw + (size_t)(e_begin + ent_of(k)) * a_w_ent + (DUAL ? (wave & 1) * a_w_win : 0) + (mtile & 1) * a_w_sub + a_lane_off;
#pragma unroll
    for (int j = 0; j < AI; j++)
      if (!(prb & kProbeNoA) || k < S - 1)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + j * 1024), TF2_LDS_PTR(slot + j * 1024), 16, 0, 0);
  };

  // header (shared by the four waves) + this wave's first weight tiles
  {
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(ahdr) + (size_t)mtile * a_hdr_bytes + lane * 16;
    int8_t* hdst = reinterpret_cast<int8_t*>(prm);
    for (int i = wave; i * 1024 < a_hdr_bytes; i += NWV)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hsrc + i * 1024), TF2_LDS_PTR(hdst + i * 1024), 16, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (s < n_mine) issue_A(s, s);

  const int8_t* brow_ptr[BI];
  int brow_h[BI], brow_w[BI];
  bool brow_ok[BI];
#pragma unroll
  for (int j = 0; j < BI; j++) {
    const int p = px0 + j * 16 + (lane >> 2);
    if (p < px_end) {
      const int b = fast_div(p, g.ohw_m, g.ohw_s);
      const int rem = p - b * g.OHW;
      const int oh = fast_div(rem, g.ow_m, g.ow_s);
      const int ow = rem - oh * g.OW;
      brow_h[j] = oh * g.stride - g.pad_h;
      brow_w[j] = ow * g.stride - g.pad_w;
      brow_ptr[j] = ax + ((long long)b * g.H * g.W + (long long)brow_h[j] * g.W + brow_w[j]) * g.Cp_in;
      brow_ok[j] = true;
    } else {
      brow_h[j] = -(1 << 20); brow_w[j] = 0; brow_ptr[j] = azero; brow_ok[j] = false;
    }
  }

  // residual prefetch for the 32x32 sub-tile this wave finishes in the epilogue
  const int half = lane >> 5;
  const int ti = (wave >> 1) & 1, tj = wave & 1;     // epilogue sub-tile (waves 0-3)
  i32x4 resv = {0, 0, 0, 0};
  asm volatile("" ::: "memory");
  if (g.has_res && wave < 4) {
    const int px = px0 + tj * 32 + (lane & 31);
    const int chl = mtile * TM + ti * 32 + 16 * half;
    const bool ok = px < px_end && chl + 16 <= g.y_nvalid;
    const int8_t* rp = ok ? ares + (size_t)px * g.res_cp + g.res_off + chl : azero;
    resv = *reinterpret_cast<const i32x4*>(rp);
  }
  asm volatile("" ::: "memory");

  i32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

  if (!DENSE) {
    if (g.has_res && wave < 4) sk_wait_vmcnt<1>(); else sk_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();            // header complete (all four waves' parts): the gather tables are read from it
    asm volatile("" ::: "memory");
  }

  auto issue_B = [&](int off, int hw, int slot_idx, bool in_loop = false) {
    int8_t* const slot = ring + slot_idx * STAGE + A_BYTES;
    int dh = 0, dw = 0, pc = 0;              // pc: the segment's channel offset = its place in the layer's pad row
    if (PADCHK) { dh = hw & 0xff; dw = (hw >> 8) & 0xff; pc = (int)((unsigned)hw >> 16); }
#pragma unroll
    for (int j = 0; j < BI; j++) {
      if ((prb & kProbeNoB) && in_loop) continue;
      bool ok = off >= 0 && brow_ok[j];
      if (PADCHK && !(prb & kProbeNoPad)) {
        const int ih = brow_h[j] + dh, iw = brow_w[j] + dw;
        ok = ok && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      }
      const int8_t* src = ok ? brow_ptr[j] + off : azero + pc;      // out of range: the stored form of x = 0 (off_pad)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(slot + j * 1024), 16, 0, 0);
    }
  };

  auto phase_shift = [&](int p) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int rb = i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + p * TM + rb + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i][j][G * 4 + r] = (int)((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31));
      }
    }
  };

  // ---- per-wave pipelined K loop over entries wave, wave+4, ... (no block barrier) -----------
  // VMEM queue of a wave: [hdr, A_0..A_{S-2}, residual, B_0..B_{S-2}, then per iteration A, B]
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (s < n_mine) {
      const int e = ent_of(s);
      int o, h = 0;
      if (DENSE) gather_of(e, o, h); else { o = goff[e * 4 + chunk]; if (PADCHK) h = ghw[e * 4 + chunk]; }
      issue_B(o, h, s);
    }
  int phase = 0;
  int cslot = 0, islot = S - 1;
  const int n_main = n_mine - (S - 1);
  int off_nx, hw_nx = 0;
  if (DENSE) gather_of(ent_of(S - 1), off_nx, hw_nx);
  else { off_nx = goff[ent_of(S - 1) * 4 + chunk]; if (PADCHK) hw_nx = ghw[ent_of(S - 1) * 4 + chunk]; }
  int next_b = DENSE ? 0x7fffffff : __builtin_amdgcn_readfirstlane(steps[0]);

  auto body = [&](int k, bool issue) {
    const int e = wave + NWV * k;          // index in the m-tile's entry list
    if (!DUAL && !DENSE)
      while (e >= next_b) {                // this wave has crossed into the next phase(s)
        phase++; phase_shift(phase);
        next_b = __builtin_amdgcn_readfirstlane(steps[phase]);
      }
    const int8_t* A = ring + cslot * STAGE;
    const int8_t* B = A + A_BYTES;
    i32x4 af[2][2], bf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int row = i * 32 + (lane & 31);
        af[ks][i] = *reinterpret_cast<const i32x4*>(A + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
        bf[ks][i] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
      }
    }
    if (issue) {
      issue_A(k + S - 1, islot);
      issue_B(off_nx, hw_nx, islot, true);
      islot = islot + 1 == S ? 0 : islot + 1;
      if (DENSE) gather_of(ent_of(k + S), off_nx, hw_nx);
      else { off_nx = goff[ent_of(k + S) * 4 + chunk]; if (PADCHK) hw_nx = ghw[ent_of(k + S) * 4 + chunk]; }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (prb & kProbeNoMfma) { asm volatile("" :: "v"(af[ks][i]), "v"(bf[ks][j])); continue; }
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
        }
    cslot = cslot + 1 == S ? 0 : cslot + 1;
  };

  int k = 0;
  for (; k < n_main; k++) {
    if (k == 0) sk_wait_vmcnt<(S - 2) * BI>();
    else sk_wait_vmcnt<(S - 2) * NI>();
    asm volatile("" ::: "memory");
    body(k, true);
  }
  for (; k < n_mine; k++) {
    sk_wait_vmcnt<0>();
    asm volatile("" ::: "memory");
    body(k, false);
  }
  if (!DUAL && !DENSE) while (phase + 1 < P) { phase++; phase_shift(phase); }

  if (prb & kProbeNoEpi) return;
  // ---- reduce the four partial tiles through LDS (the rings are dead now) ---------------------
  sk_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    i32x4* red = reinterpret_cast<i32x4*>(lds) + (size_t)wave * 1024;       // 16 KiB per wave
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int G = 0; G < 4; G++) {
          const i32x4 v = {acc[i][j][G * 4], acc[i][j][G * 4 + 1], acc[i][j][G * 4 + 2], acc[i][j][G * 4 + 3]};
          red[((i * 2 + j) * 4 + G) * 64 + lane] = v;
        }
  }
  __syncthreads();
  if (NWV > 4 && wave >= 4) return;         // partial tile handed over; waves 0-3 finish
  i32x4 sum[4];
#pragma unroll
  for (int G = 0; G < 4; G++) {
    sum[G] = i32x4{0, 0, 0, 0};
    i32x4 hi = {0, 0, 0, 0};
#pragma unroll
    for (int w = 0; w < NWV; w++) {
      const i32x4 v = reinterpret_cast<const i32x4*>(lds)[(size_t)w * 1024 + ((ti * 2 + tj) * 4 + G) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (DUAL && !(w & 1)) hi[r] = (int)((unsigned)hi[r] + (unsigned)v[r]);
        else sum[G][r] = (int)((unsigned)sum[G][r] + (unsigned)v[r]);
      }
    }
    if (DUAL) {
      const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + TM + ti * 32 + 4 * half + 8 * G);
#pragma unroll
      for (int r = 0; r < 4; r++) sum[G][r] = (int)(((unsigned)hi[r] << (d[r] & 31)) + (unsigned)sum[G][r]);
    }
  }

  // ---- KSP: this block's 64 x 64 partial tile -> memory; the block that draws the tile's last ticket adds the parts up -------------
  // (no block waits for another: stores written through (sc0 sc1) and acknowledged, then ONE device-scope ticket per block; the last
  //  one reads the others' parts past its caches.  (hi << d) + lo is linear in the parts: every block combines its own windows first.)
  if (KSP) {
    i32x4* const mine = reinterpret_cast<i32x4*>(a.ks_part) + ((size_t)(tile_lin * kparts + kpart) * 4 + wave) * 256;
#pragma unroll
    for (int G = 0; G < 4; G++)
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(mine + G * 64 + lane), "v"(sum[G]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                           // waves 0-3 (the others have left): every part of the tile is in memory
    int* const tk = reinterpret_cast<int*>(lds);               // (the reduction operands are consumed)
    if (tid == 0) *tk = (int)__hip_atomic_fetch_add(a.ks_ctr + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *tk;
    if (ticket != kparts - 1) return;
    // (the other parts four at a time: sixteen loads in flight per wait -- one part per wait made seven dependent round trips past the caches)
    const i32x4* const base_w = reinterpret_cast<const i32x4*>(a.ks_part) + ((size_t)tile_lin * kparts * 4 + wave) * 256 + lane;
    for (int p0 = 0; p0 < kparts; p0 += 4) {
      i32x4 v[4][4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        // (this block's own part and parts past the end: its own part again, added with weight 0)
        const int p = p0 + q;
        const int pp = (p < kparts && p != kpart) ? p : kpart;
#pragma unroll
        for (int G = 0; G < 4; G++)
          asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[q][G]) : "v"(base_w + (size_t)pp * 1024 + G * 64) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int p = p0 + q;
        if (p < kparts && p != kpart) {
#pragma unroll
          for (int G = 0; G < 4; G++)
#pragma unroll
            for (int r = 0; r < 4; r++) sum[G][r] = (int)((unsigned)sum[G][r] + (unsigned)v[q][G][r]);
        }
      }
    }
  }

  // ---- epilogue for this wave's 32x32 sub-tile (see conv_mfma2.hip for the arithmetic) ----------
  const int lo_bound = g.relu ? 0 : -128;
  const int rlo = g.add_relu ? 0 : -128;
  const int rb = ti * 32;
  const int tile_ch = mtile * TM + rb;
  const int px = px0 + tj * 32 + (lane & 31);
  int a16[16];
#pragma unroll
  for (int G = 0; G < 4; G++)
#pragma unroll
    for (int r = 0; r < 4; r++) a16[G * 4 + r] = sum[G][r];
  i32x4 out;
  if (g.fast == 1) out = g.has_res ? requant_tile16<true, 0, true>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, resv)
                                   : requant_tile16<false, 0, true>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, resv, g.dbl_out != 0);
  else out = g.has_res ? requant_tile16<true>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, resv, false, g.fast == 2)
                       : requant_tile16<false>(a16, prm, TM, rb + 4 * half, lo_bound, rlo, resv, g.dbl_out != 0, g.fast == 2);
  const int chl = tile_ch + 16 * half;
  if (AVG) {
    // per-channel sum of this wave's 32 pixel columns (dead columns count 0), then over the two pixel halves of the tile
    int sum16[16];
    const bool live = px < px_end;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      int v = live ? (int)(signed char)(((unsigned)out[q >> 2] >> (8 * (q & 3))) & 0xff) : 0;
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);      // lanes 0-31 and 32-63 reduce separately (m < 32)
      sum16[q] = v;
    }
    int* const xch = reinterpret_cast<int*>(lds);              // [ti][half][16] partial sums of the tj = 1 waves (rings are dead)
    __syncthreads();                                           // every wave has read its reduction operands
    if (tj == 1 && (lane & 31) == 0) {
#pragma unroll
      for (int q = 0; q < 16; q++) xch[(ti * 2 + half) * 16 + q] = sum16[q];
    }
    __syncthreads();
    if (tj == 0 && (lane & 31) == 0 && chl + 16 <= g.y_nvalid) {
      unsigned o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int sv = (int)(short)(sum16[q] + xch[(ti * 2 + half) * 16 + q]);     // Sreal: int16 accumulator wrap (types.h:30)
        int m = (((sv * g.avg_mult) >> 14) + 1) >> 1;                                // full_size_pool.cl:115-118
        m = m > 127 ? 127 : (m < -128 ? -128 : m);
        o[q >> 2] |= (unsigned)(m & 0xff) << (8 * (q & 3));
      }
      *reinterpret_cast<i32x4*>(ay + (size_t)ntile * g.y_cp + g.y_off + chl) = i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
    }
    return;
  }
  if (px < g.n_pix && chl + 16 <= g.y_nvalid)
    *reinterpret_cast<i32x4*>(ay + (size_t)px * g.y_cp + g.y_off + chl) = out;
  else if (px < g.n_pix && g.y_tail == 8 && chl == g.y_nvalid) {
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    *reinterpret_cast<i32x2_t*>(ay + (size_t)px * g.y_cp + g.y_off + chl) = i32x2_t{out[0], out[1]};
  }
}

template <int S, bool PADCHK, bool DUAL, int NWV, bool DENSE, bool AVG, bool KSP = false>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void conv_mfma_sk_kernel(ConvArgs a) {
  conv_mfma_sk_body<S, PADCHK, DUAL, NWV, DENSE, AVG, KSP>(a, (int)blockIdx.x, (int)gridDim.x);
}

// two independent layers of the same instantiation in one launch: blocks [0, n0) work on the first argument block, the rest on the second
template <int S, bool PADCHK, bool DUAL, int NWV, bool DENSE>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void conv_mfma_sk_pair_kernel(ConvArgs a0, ConvArgs a1, int n0) {
  const int b = (int)blockIdx.x;
  if (b < n0) conv_mfma_sk_body<S, PADCHK, DUAL, NWV, DENSE, false>(a0, b, n0);
  else conv_mfma_sk_body<S, PADCHK, DUAL, NWV, DENSE, false>(a1, b - n0, (int)gridDim.x - n0);
}

template <int S, bool PADCHK, bool DUAL, int NWV, bool DENSE, bool AVG>
static int launch_sk4(const ConvArgs& a, hipStream_t s) {
  constexpr int RING_ALL = (NWV * S * 8192 > NWV * 16384) ? NWV * S * 8192 : NWV * 16384;
  const size_t lds = (size_t)RING_ALL + (size_t)a.hdr_bytes + 64;
  auto fn = conv_mfma_sk_kernel<S, PADCHK, DUAL, NWV, DENSE, AVG>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  const int ntiles = AVG ? a.g.n_pix / a.g.OHW : (a.g.n_pix + 63) / 64;
  TF2_LAUNCH_NAME("conv_mfma_sk_kernel<S%d,%s%s%d waves,%s%s>", S, PADCHK ? "pad," : "", DUAL ? "dual," : "", NWV, AVG ? "global average," : "", DENSE ? "dense" : "tables");
  TF2_LAUNCH(fn, dim3(ntiles * a.n_mtiles), dim3(NWV * 64), lds, s, a);
  return launch_ok() ? 0 : -1;
}

template <int S, bool PADCHK, bool DUAL, int NWV, bool DENSE>
static int launch_sk3(const ConvArgs& a, hipStream_t s) {
  if constexpr (NWV == 4 && S == 2 && !PADCHK) { if (a.g.avg_mult) return launch_sk4<S, PADCHK, DUAL, NWV, DENSE, true>(a, s); }
  if (a.g.avg_mult) return -1;                       // the fused average exists for the unpadded two-stage four-wave shape only
  return launch_sk4<S, PADCHK, DUAL, NWV, DENSE, false>(a, s);
}

template <int S, bool PADCHK, bool DUAL, int NWV>
static int launch_sk2(const ConvArgs& a, hipStream_t s) {
  return a.dense ? launch_sk3<S, PADCHK, DUAL, NWV, true>(a, s) : launch_sk3<S, PADCHK, DUAL, NWV, false>(a, s);
}

// The instantiation a layer takes (64-row packed layers with a long slab list and a grid that would not fill the chip).
struct SkVariant { int S, pad, dual, nwv, dense, avg; bool operator==(const SkVariant& o) const { return S == o.S && pad == o.pad && dual == o.dual && nwv == o.nwv && dense == o.dense && avg == o.avg; } };
static SkVariant sk_variant(const ConvArgs& a, long sk8_blocks, long s3_blocks) {
  const long blocks = (long)((a.g.n_pix + 63) / 64) * a.n_mtiles;
  const int pad = (a.g.pad_h | a.g.pad_w) != 0, dual = a.dual != 0, dense = a.dense != 0;
  if (a.g.avg_mult) return {2, pad, dual, 4, dense, 1};     // global average fused (net.hip checked: 1x1-style unpadded layer, OHW <= 64)
  // 8-way split only for grids far below one block per CU (7x7 maps at batch 32: measured 16.0 -> 14.7 us) -- at 392
  // blocks its 136 KiB of LDS (one block per CU) costs more than the shorter K walk saves (14.0 -> 17.8 us)
  const long n_virt = (long)a.ent0 * (a.dual ? 2 : 1);
  if (blocks <= sk8_blocks && n_virt >= 16) return {2, pad, dual, 8, dense, 0};
  // three ring stages only for grids of at most one block per CU (their 100 KiB of LDS take the CU): s3_blocks = 256 one batch at a
  // time; the in-flight plan passes its own threshold (Net::launch_plan: a 64 KiB block leaves room for another batch's block)
  return {blocks <= s3_blocks ? 3 : 2, pad, dual, 4, dense, 0};
}

// K split over blocks as well (ConvArgs::ks_parts > 1; Net::launch_plan's choice for grids of a few blocks with long slab lists): the
// eight-wave two-stage dense instantiation, ks_parts blocks per output tile
template <bool PADCHK, bool DUAL>
static int launch_sk_ksp(const ConvArgs& a, hipStream_t s) {
  constexpr int NWV = 8, S = 2;
  constexpr int RING_ALL = NWV * 16384;
  const size_t lds = (size_t)RING_ALL + (size_t)a.hdr_bytes + 64;
  auto fn = conv_mfma_sk_kernel<S, PADCHK, DUAL, NWV, true, false, true>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  const int ntiles = (a.g.n_pix + 63) / 64;
  TF2_LAUNCH_NAME("conv_mfma_sk_kernel<S%d,%s%s%d waves,dense,K over %d blocks>", S, PADCHK ? "pad," : "", DUAL ? "dual," : "", NWV, a.ks_parts);
  TF2_LAUNCH(fn, dim3(ntiles * a.n_mtiles * a.ks_parts), dim3(NWV * 64), lds, s, a);
  return launch_ok() ? 0 : -1;
}

bool conv_mfma_sk_ksplit_eligible(const ConvArgs& a) { return a.dense && !a.g.avg_mult && !a.g.y_tail; }

int launch_conv_mfma_sk(const ConvArgs& a, long sk8_blocks, long s3_blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (a.ks_parts > 1) {
    if (!conv_mfma_sk_ksplit_eligible(a)) return -1;
    const bool pad = (a.g.pad_h | a.g.pad_w) != 0;
    if (a.dual) return pad ? launch_sk_ksp<true, true>(a, s) : launch_sk_ksp<false, true>(a, s);
    return pad ? launch_sk_ksp<true, false>(a, s) : launch_sk_ksp<false, false>(a, s);
  }
  const SkVariant v = sk_variant(a, sk8_blocks, s3_blocks);
  if (v.avg) {
    if (v.pad) return -1;
    return v.dual ? launch_sk2<2, false, true, 4>(a, s) : launch_sk2<2, false, false, 4>(a, s);
  }
  if (v.nwv == 8) {
    if (v.dual) return v.pad ? launch_sk2<2, true, true, 8>(a, s) : launch_sk2<2, false, true, 8>(a, s);
    return v.pad ? launch_sk2<2, true, false, 8>(a, s) : launch_sk2<2, false, false, 8>(a, s);
  }
  if (v.S == 3) {
    if (v.dual) return v.pad ? launch_sk2<3, true, true, 4>(a, s) : launch_sk2<3, false, true, 4>(a, s);
    return v.pad ? launch_sk2<3, true, false, 4>(a, s) : launch_sk2<3, false, false, 4>(a, s);
  }
  if (v.dual) return v.pad ? launch_sk2<2, true, true, 4>(a, s) : launch_sk2<2, false, true, 4>(a, s);
  return v.pad ? launch_sk2<2, true, false, 4>(a, s) : launch_sk2<2, false, false, 4>(a, s);
}

// ---- pair launch: two independent rows that take the SAME unpadded, dense, two-stage instantiation ---------------------------------
bool conv_mfma_sk_pair_eligible(const ConvArgs& a0, const ConvArgs& a1, long sk8_blocks, long s3_blocks) {
  const SkVariant v0 = sk_variant(a0, sk8_blocks, s3_blocks), v1 = sk_variant(a1, sk8_blocks, s3_blocks);
  return v0 == v1 && !v0.avg && !v0.pad && v0.dense && v0.S == 2;
}

template <bool DUAL, int NWV>
static int launch_sk_pair2(const ConvArgs& a0, const ConvArgs& a1, hipStream_t s) {
  constexpr int S = 2;
  constexpr int RING_ALL = (NWV * S * 8192 > NWV * 16384) ? NWV * S * 8192 : NWV * 16384;
  const size_t lds = (size_t)RING_ALL + (size_t)(a0.hdr_bytes > a1.hdr_bytes ? a0.hdr_bytes : a1.hdr_bytes) + 64;
  auto fn = conv_mfma_sk_pair_kernel<S, false, DUAL, NWV, true>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  if (lds > 160 * 1024) return -3;
  const int n0 = ((a0.g.n_pix + 63) / 64) * a0.n_mtiles, n1 = ((a1.g.n_pix + 63) / 64) * a1.n_mtiles;
  TF2_LAUNCH_NAME("conv_mfma_sk_pair_kernel<S%d,%s%d waves,dense> (%d + %d blocks)", S, DUAL ? "dual," : "", NWV, n0, n1);
  TF2_LAUNCH(fn, dim3(n0 + n1), dim3(NWV * 64), lds, s, a0, a1, n0);
  return launch_ok() ? 0 : -1;
}

int launch_conv_mfma_sk_pair(const ConvArgs& a0, const ConvArgs& a1, long sk8_blocks, long s3_blocks, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!conv_mfma_sk_pair_eligible(a0, a1, sk8_blocks, s3_blocks)) return 1;
  const SkVariant v = sk_variant(a0, sk8_blocks, s3_blocks);
  if (v.nwv == 8) return v.dual ? launch_sk_pair2<true, 8>(a0, a1, s) : launch_sk_pair2<false, 8>(a0, a1, s);
  return v.dual ? launch_sk_pair2<true, 4>(a0, a1, s) : launch_sk_pair2<false, 4>(a0, a1, s);
}

}  // namespace tf2
