// conv_mfma2.hip -- implicit-GEMM INT8 convolution, LDS-DMA pipelined (gfx950).
//
// Same arithmetic, operand orientation, LDS swizzle and epilogue as conv_mfma.hip (see the
// header there for the Z/2^32 exponent-window / Horner argument and the reference
// citations); what changes is how operands reach the matrix cores:
//
//  * every table the K loop needs (this m-tile's slab list, the per-slab gather table
//    `kinfo`, the per-channel epilogue parameters and Horner shifts) is copied into LDS
//    once per block, so the loop has NO dependent global-memory chain (v1 paid three
//    serial latencies per slab: entries -> kinfo -> activations);
//  * weight tiles and gathered activation rows go HBM/L2 -> LDS directly with
//    `global_load_lds_dwordx4` (1 KiB per wave instruction, no VGPR staging) into a ring
//    of S stages; a stage is waited for with a COUNTED s_waitcnt vmcnt(N) and a raw
//    s_barrier, so S-1 stages stay in flight across barriers (cdna_hip_programming.md
//    "Pipelining across barriers").  The LDS destination of an LDS-DMA is lane-linear, so
//    the XOR swizzle is applied on the per-lane SOURCE address (rule 21) and again on the
//    ds_read_b128 side; zero padding (sequencer.cl:287) is a read from a zero page;
//  * three tile shapes: 128x128 (2x2 waves of 64x64), 64x256 (1x4 waves, N_out = 64
//    layers) and 64x64 (2x2 waves of 32x32) for layers whose grid would not fill 256 CUs.
#include <hip/hip_runtime.h>
#include "tf2_internal.h"

namespace tf2 {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

#define TF2_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TF2_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int requant2_i8(int acc, int alpha, int beta, int relu) {
  long long p = (long long)acc * (long long)alpha;           // pe.cl:191
  int t = (int)(p >> kAlphaInflat);                          // pe.cl:192
  t = (int)((unsigned)t + (unsigned)beta);
  int v = ((t >> (kInflat - 1)) + 1) >> 1;                   // pe.cl:193
  v = v > 127 ? 127 : (v < -128 ? -128 : v);                 // pe.cl:194
  if (relu) v = v > 0 ? v : 0;                               // relu.cl:54
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else static_assert(N < 0, "add the vmcnt immediate");
}

// WM x WN waves (WM*WN == 4), each wave a WT x WT output tile (WT = 64 or 32), S ring stages.
template <int WM, int WN, int WT, int S>
__global__ __launch_bounds__(256, (WT == 64 ? 3 : 4)) void conv_mfma2_kernel(ConvArgs a) {
  constexpr int TM = WM * WT, TN = WN * WT;
  constexpr int NT = WT / 32;                  // 32x32 MFMA tiles per wave and dimension
  constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int AI = TM / 64, BI = TN / 64;    // LDS-DMA instructions per wave per stage (A, B)
  constexpr int NI = AI + BI;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  // LDS map: [ring S*STAGE][bias|lo|alpha|beta : 4*TM ints][dshift : P*TM ints][entries][kinfo]
  int* const prm = reinterpret_cast<int*>(lds + S * STAGE);

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool dbg_on = a.dbg != nullptr && blockIdx.x == 0 && tid == 0;
#define TF2_STAMP(i) do { if (dbg_on) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
  TF2_STAMP(0);
  const int P = a.n_phases;
  int* const dsh = prm + 4 * TM;
  int* const ent = dsh + P * TM;
  int* const kin = ent + a.max_ent;

  // XCD-aware remap: consecutive logical tiles (same pixel tile, all channel tiles) on one XCD
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mtile = bid % a.n_mtiles;
  const int ntile = bid / a.n_mtiles;
  const int px0 = ntile * TN;
  const int* dirp = a.dir + mtile * (P + 1);
  const int e_begin = dirp[0];
  const int e_end = dirp[P];
  const int n_ent = e_end - e_begin;

  TF2_STAMP(1);
  // ---- one-time table copy into LDS --------------------------------------------------
  for (int i = tid; i < TM; i += 256) {
    const int ch = mtile * TM + i;
    prm[i] = a.bias[ch]; prm[TM + i] = a.lo[ch]; prm[2 * TM + i] = a.alpha[ch]; prm[3 * TM + i] = a.beta[ch];
  }
  for (int i = tid; i < P * TM; i += 256) dsh[i] = a.dshift[(size_t)(i / TM) * a.Np + mtile * TM + (i % TM)];
  // slab id | (number of Horner phase steps to take before this entry) << 24: the K loop must not
  // contain ordinary global loads (they would make hipcc drain the LDS-DMA queue every iteration)
  for (int i = tid; i < n_ent; i += 256) {
    int steps = 0;
    for (int p = 1; p < P; p++) steps += (dirp[p] == e_begin + i && dirp[p] < e_end) ? 1 : 0;
    ent[i] = a.entries[e_begin + i] | (steps << 24);
  }
  for (int i = tid; i < a.nslab * 4; i += 256) kin[i] = a.kinfo[i];

  // ---- per-lane gather state ------------------------------------------------------------
  // LDS-DMA lane l of an instruction fills row (l>>2), 16-byte slot (l&3) of a 16-row group;
  // with the XOR swizzle slot c' of row r holds chunk c = c' ^ ((r>>2)&3), and r>>2 == l>>4
  // inside a group, so every lane always fetches the same chunk index:
  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  int brow_h[BI], brow_w[BI], brow_base[BI];
#pragma unroll
  for (int j = 0; j < BI; j++) {
    const int p = px0 + (wave + 4 * j) * 16 + (lane >> 2);
    if (p < g.n_pix) {
      const int b = p / g.OHW;
      const int rem = p - b * g.OHW;
      const int oh = rem / g.OW;
      const int ow = rem - oh * g.OW;
      brow_h[j] = oh * g.stride - g.pad_h;
      brow_w[j] = ow * g.stride - g.pad_w;
      brow_base[j] = b * g.H * g.W;
    } else {
      brow_h[j] = -(1 << 20); brow_w[j] = 0; brow_base[j] = 0;
    }
  }
  const int a_lane_off = (lane >> 2) * 64 + chunk * 16;       // inside a 16-row group of a weight tile

  i32x16 acc[NT][NT];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

  __syncthreads();     // tables visible; no LDS-DMA outstanding yet, so this is a plain barrier
  TF2_STAMP(2);

  auto issue_stage = [&](int e) {       // e: absolute entry index; fills ring slot (e - e_begin) % S
    int8_t* const slot = lds + ((e - e_begin) % S) * STAGE;
    const int slab = ent[e - e_begin] & 0xffffff;
    const unsigned ki = (unsigned)kin[slab * 4 + chunk];
    const int dh = (int)((ki >> 16) & 0xff);
    const int dw = (int)(ki >> 24);
    const int coff = (ki & 0xffff) == 0xffff ? -1 : (int)(ki & 0xffff);
    const int8_t* wsrc = a.w + (size_t)((g.flags & 4) ? e_begin : e) * A_BYTES + a_lane_off;   // bit 2: perf experiment only
#pragma unroll
    for (int j = 0; j < AI; j++) {
      const int grp = wave + 4 * j;                          // 16-row group of the A tile
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + grp * 1024), TF2_LDS_PTR(slot + grp * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BI; j++) {
      const int grp = wave + 4 * j;
      const int ih = brow_h[j] + dh, iw = brow_w[j] + dw;
      const bool ok = coff >= 0 && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W && !(g.flags & 2);   // bit 1: perf experiment only
      const int8_t* src = ok ? a.x + ((size_t)(brow_base[j] + ih * g.W + iw) * g.Cp_in + coff) : a.zero;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(slot + A_BYTES + grp * 1024), 16, 0, 0);
    }
  };

  auto compute = [&](int e) {
    const int8_t* A = lds + ((e - e_begin) % S) * STAGE;
    const int8_t* B = A + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks * 2 + (lane >> 5);
      i32x4 af[NT], bf[NT];
#pragma unroll
      for (int i = 0; i < NT; i++) {
        const int row = wm * WT + i * 32 + (lane & 31);
        af[i] = *reinterpret_cast<const i32x4*>(A + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int row = wn * WT + j * 32 + (lane & 31);
        bf[j] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  auto phase_shift = [&](int p) {       // Horner step: acc <<= dshift[p][channel]
#pragma unroll
    for (int i = 0; i < NT; i++) {
      const int rb = wm * WT + i * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const i32x4 d = *reinterpret_cast<const i32x4*>(dsh + p * TM + rb + 8 * G);
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int j = 0; j < NT; j++)
            acc[i][j][G * 4 + r] = (int)((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31));
      }
    }
  };

  // ---- pipelined K loop ---------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (e_begin + s < e_end) issue_stage(e_begin + s);
  int phase = 0;
  TF2_STAMP(3);
  for (int e = e_begin; e < e_end; e++) {
    // stages issued beyond e: min(S-2, e_end-1-e); wait until stage e has landed (per wave)
    const int ahead = (e_end - 1 - e) < (S - 2) ? (e_end - 1 - e) : (S - 2);
    if (ahead >= S - 2) wait_vmcnt<(S - 2) * NI>();
    else if (S >= 4 && ahead == 1) wait_vmcnt<NI>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();          // every wave's part of stage e landed; slot (e-1)%S is free
    if (e == e_begin) TF2_STAMP(4);
    asm volatile("" ::: "memory");         // compile-time fence: no LDS access may be hoisted above the barrier
    if (e + S - 1 < e_end && !(g.flags & 8)) issue_stage(e + S - 1);     // bit 3/4: perf experiments only
    for (int st = ent[e - e_begin] >> 24; st > 0; st--) { phase++; phase_shift(phase); }
    if (!(g.flags & 16)) compute(e);
  }
  while (phase + 1 < P) { phase++; phase_shift(phase); }
  TF2_STAMP(5);

  // ---- epilogue --------------------------------------------------------------------------------
  const int half = lane >> 5;
#pragma unroll
  for (int i = 0; i < NT; i++) {
    const int rb = wm * WT + i * 32;                         // tile row base inside the block tile
    const int tile_ch = mtile * TM + rb;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int px = px0 + wn * WT + j * 32 + (lane & 31);
      const bool pvalid = px < g.n_pix;
      unsigned d[4];
#pragma unroll
      for (int G = 0; G < 4; G++) {
        const int r0 = rb + 4 * half + 8 * G;
        const i32x4 bias4 = *reinterpret_cast<const i32x4*>(prm + r0);
        const i32x4 lo4 = *reinterpret_cast<const i32x4*>(prm + TM + r0);
        const i32x4 al4 = *reinterpret_cast<const i32x4*>(prm + 2 * TM + r0);
        const i32x4 be4 = *reinterpret_cast<const i32x4*>(prm + 3 * TM + r0);
        int q[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int v = (int)((unsigned)bias4[r] + ((unsigned)acc[i][j][G * 4 + r] << (lo4[r] & 31)));
          q[r] = requant2_i8(v, al4[r], be4[r], g.relu);
        }
        d[G] = (unsigned)(q[0] & 0xff) | ((unsigned)(q[1] & 0xff) << 8) | ((unsigned)(q[2] & 0xff) << 16) |
               ((unsigned)(q[3] & 0xff) << 24);
      }
      // d[G] = channel group 2G (lanes 0-31) / 2G+1 (lanes 32-63): two half-wave swaps give
      // lanes 0-31 groups 0..3 and lanes 32-63 groups 4..7 -> 16 contiguous NHWC bytes per lane
      auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
      i32x4 out = {(int)s02[0], (int)s02[1], (int)s13[0], (int)s13[1]};
      const int chl = tile_ch + 16 * half;
      if (pvalid && chl + 16 <= g.y_nvalid) {
        if (g.has_res) {
          // residual add in int16, clamp, ReLU (feature_writer.cl:119-122) on the packed bytes
          const i32x4 rv = *reinterpret_cast<const i32x4*>(a.res + (size_t)px * g.res_cp + g.res_off + chl);
#pragma unroll
          for (int w = 0; w < 4; w++) {
            int o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
              int s = (int)(signed char)((out[w] >> (8 * b)) & 0xff) + (int)(signed char)((rv[w] >> (8 * b)) & 0xff);
              s = s > 127 ? 127 : (s < -128 ? -128 : s);
              if (g.add_relu) s = s > 0 ? s : 0;
              o |= (s & 0xff) << (8 * b);
            }
            out[w] = o;
          }
        }
        *reinterpret_cast<i32x4*>(a.y + (size_t)px * g.y_cp + g.y_off + chl) = out;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TF2_STAMP(6);
}

template <int WM, int WN, int WT, int S>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int TM = WM * WT, TN = WN * WT;
  constexpr int STAGE = (TM + TN) * 64;
  const size_t lds = (size_t)S * STAGE + (size_t)(4 + a.n_phases) * TM * 4 + (size_t)a.max_ent * 4 + (size_t)a.nslab * 16 + 64;
  static bool attr_set = false;
  auto fn = conv_mfma2_kernel<WM, WN, WT, S>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    attr_set = true;
  }
  if (lds > 160 * 1024) return -3;
  const int ntiles = (a.g.n_pix + TN - 1) / TN;
  hipLaunchKernelGGL(fn, dim3(ntiles * a.n_mtiles), dim3(256), lds, s, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// TM is fixed by the packed image (64 or 128); the pixel-tile shape is picked per launch so
// that small grids still spread over the 256 CUs.
int launch_conv_mfma2(const ConvArgs& a, int TM, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (TM == 128) return launch_cfg<2, 2, 64, 3>(a, s);
  if (TM == 64) {
    const long blocks256 = (long)((a.g.n_pix + 255) / 256) * a.n_mtiles;
    if (blocks256 >= 512) return launch_cfg<1, 4, 64, 3>(a, s);
    return launch_cfg<2, 2, 32, 4>(a, s);
  }
  return -1;
}

}  // namespace tf2
