// conv_fused.hip -- two convolution blocks in one launch: A (k x k, any stride / padding) -> requant / ReLU ->
// B (1 x 1 expand) + residual add + ReLU, the `branch2b -> branch2c` pair of every ResNet bottleneck (gfx950).
//
// The reference runs each of these as its own pass of the whole FPGA pipeline with the intermediate map in DDR
// (feature_writer.cl:88-137 -> retriever.cl); round 1 did the same on the GPU: two launches, each paying a block's
// serial latency chain, and a write + read of the 64..256-channel intermediate tensor.  Here a block owns ALL output
// channels of layer A for its pixel tile (TM = N_A = 64 / 128 / 256), so A's requantised int8 tile is exactly the
// B operand of layer B's GEMM for the same pixels and never leaves the CU:
//
//   phase 1  = conv_mfma2.hip's K loop, unchanged (LDS-DMA ring, gathered NHWC activations, dual exponent windows);
//   hand-over: requant_tile16 of A (pe.cl:185-203, relu.cl:54) -> ds_write_b128 into an LDS "mid" tile laid out like a
//              ring B tile per 64-channel slab ([slab][pixel row][64 B], same XOR swizzle);
//   phase 2  = 4 passes over B's weight tiles (B's output channels = 4 x TM; one pass = one m-tile of B packed with
//              TM rows), weights streamed through the same ring by LDS-DMA, B operand read from the mid tile, the
//              usual epilogue with the residual tile (feature_writer.cl:119-122), 16-byte NHWC stores.
//
// All four residual tiles are prefetched into registers before phase 1 (they are the oldest entries of the VMEM queue,
// as in conv_mfma2); a pass's packed output replaces its residual tile in the same registers and all stores are issued
// after the last pass, so the counted vmcnt waits of phase 2 only ever see LDS-DMA loads (which retire in order).  A
// first version stored after every pass and counted the stores in the waits: bit-exact at small batch, wrong under
// load -- stores may retire ahead of older loads.  The ring is DEEP (5-6 stages where LDS allows): a block walks its
// 18-36 K steps alone, and a step costs one DMA latency divided by the stages in flight.
// Arithmetic is bit-identical to running conv_mfma2 twice: same accumulators, same requantisation, and the intermediate
// is the same int8 tensor (with keep_mid it is also written to HBM so that per-layer parity tests see it).
#include <hip/hip_runtime.h>
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
__device__ __forceinline__ void fz_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int T, int N, class F>
__device__ __forceinline__ void fz_static_for(F& fn) {
  if constexpr (T < N) { fn(std::integral_constant<int, T>{}); fz_static_for<T + 1, N>(fn); }
}

constexpr int kFusedPasses = 4;      // output channels of B = 4 x TM (ResNet bottleneck expansion)

// WM x WN waves, wave tile WTM x WTN, S ring stages, OCC blocks per CU the register budget is set for.  TM = WM * WTM is
// layer A's (padded) output channel count AND layer B's K; DUAL / DUAL2: dual-window packing of A / B.
template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL, bool DUAL2>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * OCC) / 4) void conv_fused_kernel(FusedArgs f) {
  constexpr int NW = WM * WN;
  constexpr int TM = WM * WTM, TN = WN * WTN;
  constexpr int NTM = WTM / 32, NTN = WTN / 32;
  constexpr int NSL2 = TM / 64;                          // K slabs of layer B
  constexpr int NT2 = kFusedPasses * NSL2;               // phase-2 steps
  constexpr int A_BYTES = (DUAL ? 2 : 1) * TM * 64, B_BYTES = TN * 64;
  constexpr int A2_BYTES = (DUAL2 ? 2 : 1) * TM * 64;
  constexpr int STAGE = (A_BYTES + B_BYTES) > A2_BYTES ? (A_BYTES + B_BYTES) : A2_BYTES;
  constexpr int AG = A_BYTES / 1024, BG = TN / 16, NG = AG + BG;
  constexpr int NI_LO = NG / NW, REM = NG % NW, NI_HI = NI_LO + (REM ? 1 : 0);
  constexpr int AG2 = A2_BYTES / 1024;
  constexpr int NI2_LO = AG2 / NW, REM2 = AG2 % NW, NI2_HI = NI2_LO + (REM2 ? 1 : 0);
  static_assert((S - 2) * NI_HI <= 63 && S >= 2, "ring depth");
  constexpr int MID_BYTES = TN * TM;
  extern __shared__ __attribute__((aligned(16))) int8_t lds[];
  // LDS map: [ring S*STAGE][mid TN*TM][header of A (hdr_bytes)][4 headers of B (hdr2_used each)]
  int8_t* const mid = lds + S * STAGE;
  int* const prm = reinterpret_cast<int*>(mid + MID_BYTES);

  const ConvArgs& a = f.a;
  TF2_PRELOAD_CONV_ARGS(a);
  const int8_t* const aw2 = f.w2; const int32_t* const ahdr2 = f.hdr2; int8_t* const ay2 = f.y2;
  const int hdr2_bytes = f.hdr2_bytes, hdr2_used = f.hdr2_used, P2 = f.P2;
  asm volatile("" :: "s"(aw2), "s"(ahdr2), "s"(ay2), "s"(hdr2_bytes), "s"(hdr2_used), "s"(P2));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool ni_hi = REM == 0 || wave < REM;
  const bool ni2_hi = REM2 == 0 || wave < REM2;
  const int wm = wave / WN, wn = wave % WN;
  int* const dsh = prm + kPrmWordsPerRow * TM;
  int* const steps = dsh + P * TM;
  int* const goff = steps + a_max_ent;
  int* const ghw = goff + a_max_ent * 4;
  int* const prm2 = reinterpret_cast<int*>(reinterpret_cast<int8_t*>(prm) + a_hdr_bytes);

  // XCD-aware remap (one m-tile: consecutive pixel tiles per XCD)
  const int nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int px0 = bid * TN;

  const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
  const int a_lane_off = (lane >> 2) * 64 + chunk * 16;

  typedef const __attribute__((address_space(4))) i32x4* cvec_p;
  typedef const __attribute__((address_space(4))) int __attribute__((ext_vector_type(2)))* cvec2_p;
  const auto ee = *(cvec2_p)(unsigned long long)(ahdr + kPrmWordsPerRow * TM + P * TM + a_max_ent - 2);
  const int e_begin = ee[0];
  const int n_ent = ee[1] - ee[0];
  cvec_p const hg = (cvec_p)(unsigned long long)(ahdr + kPrmWordsPerRow * TM + P * TM + a_max_ent);
  int pro_off[S - 1], pro_hw[S - 1];
#pragma unroll
  for (int s = 0; s < S - 1; s++) {
    const i32x4 o = hg[s];
    pro_off[s] = chunk == 0 ? o[0] : chunk == 1 ? o[1] : chunk == 2 ? o[2] : o[3];
    pro_hw[s] = 0;
    if (PADCHK) {
      const i32x4 h = hg[a_max_ent + s];
      pro_hw[s] = chunk == 0 ? h[0] : chunk == 1 ? h[1] : chunk == 2 ? h[2] : h[3];
    }
  }

  // ---- residual tiles of all four passes (ordinary loads, oldest in this wave's VMEM queue) ----
  const int half = lane >> 5;
  i32x4 resv[kFusedPasses][NTM][NTN];
#pragma unroll
  for (int mt = 0; mt < kFusedPasses; mt++)
#pragma unroll
    for (int i = 0; i < NTM; i++)
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        const int px = px0 + wn * WTN + j * 32 + (lane & 31);
        const int chl = mt * TM + wm * WTM + i * 32 + 16 * half;
        const bool ok = f.has_res && px < g.n_pix && chl + 16 <= f.y2_nvalid;
        const int8_t* rp = ok ? f.res + (size_t)px * f.res_cp + f.res_off + chl : azero;
        resv[mt][i][j] = *reinterpret_cast<const i32x4*>(rp);
      }
  asm volatile("" ::: "memory");

  const int8_t* brow_ptr[NI_HI];
  int brow_h[NI_HI], brow_w[NI_HI];
  bool brow_ok[NI_HI];
#pragma unroll
  for (int j = 0; j < NI_HI; j++) {
    const int gi = wave + NW * j;
    brow_h[j] = -(1 << 20); brow_w[j] = 0; brow_ptr[j] = azero; brow_ok[j] = false;
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

  auto issue_stage = [&](int e, int off, int hw, int slot_idx) {
    int8_t* const slot = lds + slot_idx * STAGE;
    const int8_t* wsrc = aw + (size_t)e * A_BYTES + a_lane_off;
    int dh = 0, dw = 0;
    if (PADCHK) { dh = hw & 0xffff; dw = hw >> 16; }
#pragma unroll
    for (int j = 0; j < NI_HI; j++) {
      const int gi = wave + NW * j;
      if (gi < AG) {
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + gi * 1024), TF2_LDS_PTR(slot + gi * 1024), 16, 0, 0);
      } else if (j < NI_LO || ni_hi) {
        bool ok = off >= 0 && brow_ok[j];
        if (PADCHK) {
          const int ih = brow_h[j] + dh, iw = brow_w[j] + dw;
          ok = ok && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
        }
        const int8_t* src = ok ? brow_ptr[j] + off : azero;
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(src), TF2_LDS_PTR(slot + gi * 1024), 16, 0, 0);
      }
    }
  };

  // ---- block start: header of A, the four parameter headers of B, then the first S-1 stages ----
  {
    const int8_t* hsrc = reinterpret_cast<const int8_t*>(ahdr) + lane * 16;
    int8_t* hdst = reinterpret_cast<int8_t*>(prm);
    for (int i = wave; i * 1024 < a_hdr_bytes; i += NW)
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(hsrc + i * 1024), TF2_LDS_PTR(hdst + i * 1024), 16, 0, 0);
    const int per = hdr2_used >> 10;                      // KiB per B header that the epilogue needs
    for (int i = wave; i < kFusedPasses * per; i += NW) {
      const int mt = i / per, k = i - mt * per;
      __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(reinterpret_cast<const int8_t*>(ahdr2) + (size_t)mt * hdr2_bytes + k * 1024 + lane * 16),
                                       TF2_LDS_PTR(reinterpret_cast<int8_t*>(prm2) + i * 1024), 16, 0, 0);
    }
  }
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (s < n_ent) issue_stage(e_begin + s, pro_off[s], pro_hw[s], s);

  i32x16 acc[NTM][NTN], acc2[(DUAL || DUAL2) ? NTM : 1][(DUAL || DUAL2) ? NTN : 1];
#pragma unroll
  for (int i = 0; i < NTM; i++)
#pragma unroll
    for (int j = 0; j < NTN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0; if (DUAL || DUAL2) acc2[i][j][r] = 0; }

  auto phase_shift = [&](int p) {
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

  // ---- phase 1: pipelined K loop of layer A (conv_mfma2.hip) --------------------------------------------------
  const int n_main = n_ent - (S - 1);
  auto wait_main = [&]() {
    if (ni_hi) fz_wait_vmcnt<(S - 2) * NI_HI>(); else fz_wait_vmcnt<(S - 2) * NI_LO>();
  };
  if (n_main > 0) wait_main(); else fz_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int phase = 0;
  int cslot = 0;
  int islot = S - 1;
  int off_nx = goff[(S - 1) * 4 + chunk];
  int hw_nx = PADCHK ? ghw[(S - 1) * 4 + chunk] : 0;
  int next_b = __builtin_amdgcn_readfirstlane(steps[0]);

  auto body = [&](int it, bool issue) {
    if (!DUAL)
      while (it == next_b) {
        phase++; phase_shift(phase);
        next_b = __builtin_amdgcn_readfirstlane(steps[phase]);
      }
    const int8_t* A = lds + cslot * STAGE;
    const int8_t* B = A + A_BYTES;
    auto issue_next = [&]() {
      if (issue) {
        issue_stage(e_begin + it + S - 1, off_nx, hw_nx, islot);
        islot = islot + 1 == S ? 0 : islot + 1;
        off_nx = goff[(it + S) * 4 + chunk];
        if (PADCHK) hw_nx = ghw[(it + S) * 4 + chunk];
      }
    };
    if (DUAL) {
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
          for (int j = 0; j < NTN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
    }
    cslot = cslot + 1 == S ? 0 : cslot + 1;
  };

  int it = 0;
  for (; it < n_main; it++) {
    if (it) {
      wait_main();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    body(it, true);
  }
  for (; it < n_ent; it++) {
    if (it) {
      fz_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    body(it, false);
  }
  if (DUAL) {
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
  } else {
    while (phase + 1 < P) { phase++; phase_shift(phase); }
  }

  // ---- hand-over: every wave is done with the ring; start streaming B's weights, requantise A into the mid tile ----
  fz_wait_vmcnt<0>();                        // (n_ent == 0 cannot happen: a layer has at least one entry per m-tile)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  auto issue2 = [&](int t) {                 // weight tile of phase-2 step t -> ring slot t % S
    int8_t* const slot = lds + (t % S) * STAGE;
    const int8_t* wsrc = aw2 + (size_t)t * A2_BYTES + a_lane_off;
#pragma unroll
    for (int j = 0; j < NI2_HI; j++) {
      const int gi = wave + NW * j;
      if (j < NI2_LO || ni2_hi)
        __builtin_amdgcn_global_load_lds(TF2_GLOBAL_PTR(wsrc + gi * 1024), TF2_LDS_PTR(slot + gi * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < S - 1; t++)
    if (t < NT2) issue2(t);

  {
    const int lo_bound = g.relu ? 0 : -128;
    const i32x4 nores = {0, 0, 0, 0};
    auto to_mid = [&](auto fast_c) {
      constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
      for (int i = 0; i < NTM; i++) {
        const int rb = wm * WTM + i * 32;
        const int chl = rb + 16 * half;                     // local channel of this lane's 16 bytes
#pragma unroll
        for (int j = 0; j < NTN; j++) {
          int a16[16];
#pragma unroll
          for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
          const i32x4 out = requant_tile16<false, 0, FAST>(a16, prm, TM, rb + 4 * half, lo_bound, -128, nores);
          const int row = wn * WTN + j * 32 + (lane & 31);
          const int c = (chl & 63) >> 4;
          *reinterpret_cast<i32x4*>(mid + (chl >> 6) * (TN * 64) + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = out;
          if (f.keep_mid) {
            const int px = px0 + row;
            if (px < g.n_pix && chl + 16 <= g.y_nvalid)
              *reinterpret_cast<i32x4*>(ay + (size_t)px * g.y_cp + g.y_off + chl) = out;
          }
        }
      }
    };
    if (g.fast) to_mid(std::true_type{}); else to_mid(std::false_type{});
    if (f.keep_mid) fz_wait_vmcnt<0>();      // parity runs only: keeps the store count of phase 2 exact
  }
#pragma unroll
  for (int i = 0; i < NTM; i++)
#pragma unroll
    for (int j = 0; j < NTN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[i][j][r] = 0; if (DUAL || DUAL2) acc2[i][j][r] = 0; }
  // mid tile complete (ds_write retired) before anyone reads it: the barrier of step 0 below follows an lgkmcnt(0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- phase 2: four passes of layer B over the resident mid tile ---------------------------------------------------
  const int lo_bound2 = f.relu2 ? 0 : -128;
  const int rlo = f.add_relu ? 0 : -128;
  auto step2 = [&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    constexpr int mt = t / NSL2, s = t % NSL2;
    // VMEM operations younger than this step's weight DMA: the DMAs of the following in-flight steps (nothing else is
    // issued during phase 2)
    constexpr int last_issued = (t + S - 2) < (NT2 - 1) ? (t + S - 2) : (NT2 - 1);
    constexpr int n_dma_after = last_issued - t;
    if (ni2_hi) fz_wait_vmcnt<n_dma_after * NI2_HI>(); else fz_wait_vmcnt<n_dma_after * NI2_LO>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int8_t* A = lds + (t % S) * STAGE;
    const int8_t* B = mid + s * (TN * 64);
    i32x4 af[2][NTM], af2[DUAL2 ? 2 : 1][DUAL2 ? NTM : 1], bf[2][NTN];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int c = ks * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < NTM; i++) {
        const int row = wm * WTM + i * 32 + (lane & 31);
        const int o = row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
        af[ks][i] = *reinterpret_cast<const i32x4*>(A + o);
        if (DUAL2) af2[ks][i] = *reinterpret_cast<const i32x4*>(A + TM * 64 + o);
      }
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        const int row = wn * WTN + j * 32 + (lane & 31);
        bf[ks][j] = *reinterpret_cast<const i32x4*>(B + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
      }
    }
    if (t + S - 1 < NT2) issue2(t + S - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < NTM; i++)
#pragma unroll
        for (int j = 0; j < NTN; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
          if (DUAL2) acc2[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af2[ks][i], bf[ks][j], acc2[i][j], 0, 0, 0);
        }
    if (s == NSL2 - 1) {
      int* const pm = reinterpret_cast<int*>(reinterpret_cast<int8_t*>(prm2) + (size_t)mt * hdr2_used);
      if (DUAL2) {
        const int* dsh2 = pm + kPrmWordsPerRow * TM;
#pragma unroll
        for (int i = 0; i < NTM; i++) {
          const int rb = wm * WTM + i * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int G = 0; G < 4; G++) {
            const i32x4 d = *reinterpret_cast<const i32x4*>(dsh2 + TM + rb + 8 * G);
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
              for (int j = 0; j < NTN; j++)
                acc[i][j][G * 4 + r] = (int)(((unsigned)acc[i][j][G * 4 + r] << (d[r] & 31)) + (unsigned)acc2[i][j][G * 4 + r]);
          }
        }
      }
      auto epilogue = [&](auto has_res_c, auto fast_c) {
        constexpr bool HAS_RES = decltype(has_res_c)::value;
        constexpr bool FAST = decltype(fast_c)::value;
#pragma unroll
        for (int i = 0; i < NTM; i++) {
          const int rb = wm * WTM + i * 32;
#pragma unroll
          for (int j = 0; j < NTN; j++) {
            int a16[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a16[r] = acc[i][j][r];
            // the packed 16 output bytes take the place of the residual tile; stored after the last pass
            resv[mt][i][j] = requant_tile16<HAS_RES, 0, FAST>(a16, pm, TM, rb + 4 * half, lo_bound2, rlo, resv[mt][i][j]);
          }
        }
      };
      if (f.fast2) { if (f.has_res) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{}); }
      else { if (f.has_res) epilogue(std::true_type{}, std::false_type{}); else epilogue(std::false_type{}, std::false_type{}); }
#pragma unroll
      for (int i = 0; i < NTM; i++)
#pragma unroll
        for (int j = 0; j < NTN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) { acc[i][j][r] = 0; if (DUAL2) acc2[i][j][r] = 0; }
    }
  };
  // fully unrolled: every step has its own compile-time wait count
  fz_static_for<0, NT2>(step2);
  // ---- all stores: 16 contiguous NHWC bytes per lane and tile ----
#pragma unroll
  for (int mt = 0; mt < kFusedPasses; mt++)
#pragma unroll
    for (int i = 0; i < NTM; i++)
#pragma unroll
      for (int j = 0; j < NTN; j++) {
        const int px = px0 + wn * WTN + j * 32 + (lane & 31);
        const int chl = mt * TM + wm * WTM + i * 32 + 16 * half;
        if (px < g.n_pix && chl + 16 <= f.y2_nvalid)
          *reinterpret_cast<i32x4*>(ay2 + (size_t)px * f.y2_cp + f.y2_off + chl) = resv[mt][i][j];
      }
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC, bool PADCHK, bool DUAL, bool DUAL2>
static int launch_fused_cfg(const FusedArgs& f, hipStream_t s) {
  constexpr int TM = WM * WTM, TN = WN * WTN;
  constexpr int A_BYTES = (DUAL ? 2 : 1) * TM * 64, B_BYTES = TN * 64, A2_BYTES = (DUAL2 ? 2 : 1) * TM * 64;
  constexpr int STAGE = (A_BYTES + B_BYTES) > A2_BYTES ? (A_BYTES + B_BYTES) : A2_BYTES;
  const size_t lds = (size_t)S * STAGE + (size_t)TN * TM + (size_t)f.a.hdr_bytes + (size_t)kFusedPasses * f.hdr2_used + 64;
  if (lds > 160 * 1024) return -3;
  auto fn = conv_fused_kernel<WM, WN, WTM, WTN, S, OCC, PADCHK, DUAL, DUAL2>;
  if (!lds_attr_once(reinterpret_cast<const void*>(fn))) return -1;
  const int ntiles = (f.a.g.n_pix + TN - 1) / TN;
  hipLaunchKernelGGL(fn, dim3(ntiles), dim3(WM * WN * 64), lds, s, f);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int WM, int WN, int WTM, int WTN, int S, int OCC>
static int launch_fused_shape(const FusedArgs& f, hipStream_t s) {
  const bool pad = (f.a.g.pad_h | f.a.g.pad_w) != 0;
  const bool d1 = f.a.dual != 0, d2 = f.dual2 != 0;
#define TF2_FZ(P, D1, D2) launch_fused_cfg<WM, WN, WTM, WTN, S, OCC, P, D1, D2>(f, s)
  if (pad) {
    if (d1) return d2 ? TF2_FZ(true, true, true) : TF2_FZ(true, true, false);
    return d2 ? TF2_FZ(true, false, true) : TF2_FZ(true, false, false);
  }
  if (d1) return d2 ? TF2_FZ(false, true, true) : TF2_FZ(false, true, false);
  return d2 ? TF2_FZ(false, false, true) : TF2_FZ(false, false, false);
#undef TF2_FZ
}

// LDS bytes the fused kernel needs for a (TM, TN, S) shape (weight_pack.cpp decides with it whether a pair is fused)
size_t conv_fused_lds_bytes(int TM, int TN, int S, int dual1, int dual2, size_t hdr1_bytes, size_t hdr2_used) {
  const size_t a1 = (size_t)(dual1 ? 2 : 1) * TM * 64 + (size_t)TN * 64, a2 = (size_t)(dual2 ? 2 : 1) * TM * 64;
  return (size_t)S * (a1 > a2 ? a1 : a2) + (size_t)TN * TM + hdr1_bytes + (size_t)kFusedPasses * hdr2_used + 64;
}
static size_t fused_lds(const FusedArgs& f, int TM, int TN, int S) {
  return conv_fused_lds_bytes(TM, TN, S, f.a.dual, f.dual2, (size_t)f.a.hdr_bytes, (size_t)f.hdr2_used);
}

// TM = layer A's padded output channels (128 / 256; 64-channel pairs run faster unfused -- conv_pw takes their expand
// layer).  `shape`: 0 = wide pixel tile (wave tile 32 x 64), 1 = narrow (32 x 32: twice the blocks, for small maps).  The
// deepest ring that fits the 160 KiB of LDS is used.  Returns 1 if no instantiation fits.
int launch_conv_fused(const FusedArgs& f, int TM, int shape, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr size_t kLds = 160 * 1024;
  if (TM == 128) {
    if (shape == 0) {
      if (fused_lds(f, 128, 128, 5) <= kLds) return launch_fused_shape<4, 2, 32, 64, 5, 1>(f, s);
      return fused_lds(f, 128, 128, 3) <= kLds ? launch_fused_shape<4, 2, 32, 64, 3, 1>(f, s) : 1;
    }
    if (fused_lds(f, 128, 64, 5) <= kLds) return launch_fused_shape<4, 2, 32, 32, 5, 1>(f, s);
    return fused_lds(f, 128, 64, 3) <= kLds ? launch_fused_shape<4, 2, 32, 32, 3, 1>(f, s) : 1;
  }
  if (TM == 256) {
    if (shape == 0) {
      if (fused_lds(f, 256, 64, 5) <= kLds) return launch_fused_shape<8, 1, 32, 64, 5, 1>(f, s);
      return fused_lds(f, 256, 64, 2) <= kLds ? launch_fused_shape<8, 1, 32, 64, 2, 1>(f, s) : 1;
    }
    if (fused_lds(f, 256, 32, 6) <= kLds) return launch_fused_shape<8, 1, 32, 32, 6, 1>(f, s);
    return fused_lds(f, 256, 32, 3) <= kLds ? launch_fused_shape<8, 1, 32, 32, 3, 1>(f, s) : 1;
  }
  return 1;
}

}  // namespace tf2
