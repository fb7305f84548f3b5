// Device-side helpers shared by the HIP kernels (include after <hip/hip_runtime.h>).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tf2_internal.h"

namespace tf2 {

// n / d with the (m, s) pair of set_fast_div (tf2_internal.h): one v_mul_hi_u32 and a shift instead of the
// ~25-instruction integer division sequence -- the pixel decode runs once per block and lane, and the short-K
// layers are bound by VALU issue.
__device__ __forceinline__ int fast_div(int n, uint32_t m, int s) {
  return s < 0 ? n : (int)(__umulhi((unsigned)n, m) >> s);
}

// Two unsigned 16-bit maxima per register (v_pk_max_u16): the building block of the byte-wise maximum of post-ReLU bytes (0..127) in the pools
// fused into conv launches -- even bytes (x & 0x00ff00ff) and odd bytes ((x >> 8) & 0x00ff00ff) separately, then e | o << 8.
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// Pull the kernel arguments a conv block needs into SGPRs at kernel entry: hipcc otherwise loads each field right
// before its first use -- a dozen dependent scalar-load round trips spread over the prologue (measured on conv_mfma2:
// 2.8 k -> 1.8 k cycles from block start to the first address computation, and 3.5 k -> 1.7 k for the part after it).
// The empty asm statements pin the values, so the loads are emitted here, merged into wide s_load_dwordxN and waited
// for once.  Declares: g (ConvGeom copy), ax ay ares aw azero ahdr, a_hdr_bytes a_max_ent P a_n_mtiles mt_m mt_s, a_nslab and the
// DENSE gather constants a_cslabs a_cs_m a_cs_s a_k a_kk_m a_kk_s a_dil.
#define TF2_PRELOAD_CONV_ARGS(a)                                                                                          \
  const ConvGeom g = (a).g;                                                                                                \
  const int8_t* const ax = (a).x; int8_t* const ay = (a).y; const int8_t* const ares = (a).res;                            \
  const int8_t* const aw = (a).w; const int8_t* const azero = (a).zero; const int32_t* const ahdr = (a).hdr;               \
  const int a_hdr_bytes = (a).hdr_bytes, a_max_ent = (a).max_ent, P = (a).n_phases, a_n_mtiles = (a).n_mtiles;             \
  const unsigned mt_m = (a).mt_m; const int mt_s = (a).mt_s;                                                               \
  const int a_nslab = (a).nslab, a_cslabs = (a).cslabs, a_cs_s = (a).cs_s, a_k = (a).k, a_kk_s = (a).kk_s, a_dil = (a).dil;   \
  const unsigned a_cs_m = (a).cs_m, a_kk_m = (a).kk_m;                                                                      \
  const int a_w_ent = (a).w_ent_bytes, a_w_win = (a).w_win_stride, a_w_half = (a).w_half_stride, a_w_sub = (a).w_sub_step;    \
  const int a_e_shl = (a).e_mt_shl, a_e_shr = (a).e_mt_shr;                                                                  \
  asm volatile("" :: "s"(a_w_ent), "s"(a_w_win), "s"(a_w_half), "s"(a_w_sub), "s"(a_e_shl), "s"(a_e_shr));                   \
  asm volatile("" :: "s"(a_nslab), "s"(a_cslabs), "s"(a_cs_s), "s"(a_k), "s"(a_kk_s), "s"(a_dil), "s"(a_cs_m), "s"(a_kk_m)); \
  asm volatile("" :: "s"(g.H), "s"(g.W), "s"(g.Cp_in), "s"(g.OW), "s"(g.OHW), "s"(g.ohw_m), "s"(g.ow_m), "s"(g.ohw_s),     \
               "s"(g.ow_s), "s"(g.stride), "s"(g.pad_h), "s"(g.pad_w), "s"(g.n_pix), "s"(g.y_cp), "s"(g.y_off), "s"(g.y_nvalid)); \
  asm volatile("" :: "s"(g.res_cp), "s"(g.res_off), "s"(g.relu), "s"(g.add_relu), "s"(g.has_res), "s"(g.fast), "s"(g.flags), \
               "s"(a_hdr_bytes), "s"(a_max_ent), "s"(P), "s"(a_n_mtiles), "s"(mt_m), "s"(mt_s));                           \
  asm volatile("" :: "s"(ax), "s"(ay), "s"(ares), "s"(aw), "s"(azero), "s"(ahdr))

// Timing probes (tools/probe_run.py): a build with -DTF2_PROBES (tf2_amd/libtf2amd_probe.so, never the product library)
// reads ConvGeom::flags bits 3.. and leaves out one component of a conv kernel -- its results are then WRONG; only the
// change in duration is of interest (which pipe binds a layer).  In the product build prb is the constant 0.
constexpr int kProbeNoB = 8, kProbeNoA = 16, kProbeNoMfma = 32, kProbeNoEpi = 64, kProbeNoPad = 128, kProbeNoBar = 256,
              kProbeNoStore = 2048 /* conv_pw / conv_bneck / conv_stem: keep the arithmetic, skip the output stores */,
              kProbeQuarterBlocks = 4096 /* conv_mfma_sk: three of four blocks return at entry */, kProbeQuarterK = 8192 /* conv_mfma_sk: a quarter of the slab list */,
              kProbeExit0 = 512 /* return at kernel entry */, kProbeExit1 = 1024 /* return once the header's scalar words are there */;
#ifdef TF2_PROBES
#define TF2_PROBE_WORD(f) const int prb = (f)
#else
#define TF2_PROBE_WORD(f) constexpr int prb = 0
#endif

// DENSE layers (ConvArgs::dense): gather words of slab sl from its index, all wave-uniform (scalar ALU).  off = byte offset of the
// slab's 64 bytes from the pixel's tap origin, hw = dh | dw << 8 | (channel offset << 16) -- what weight_pack.cpp stores per
// (entry, 16-byte segment) as goff / ghw, minus the lane's own chunk * 16 (added by the caller).
struct DenseGeom { int cslabs; unsigned cs_m; int cs_s; int k; unsigned kk_m; int kk_s; int dil; int W; int Cp_in; };
__device__ __forceinline__ void dense_gather(const DenseGeom& d, int sl, int& off, int& hw) {
  const int t = d.cs_s < 0 ? sl : (int)(__umulhi((unsigned)sl, d.cs_m) >> d.cs_s);
  const int cs = sl - t * d.cslabs;
  const int th = d.kk_s < 0 ? t : (int)(__umulhi((unsigned)t, d.kk_m) >> d.kk_s);
  const int tw = t - th * d.k;
  const int dh = th * d.dil, dw = tw * d.dil;
  off = (dh * d.W + dw) * d.Cp_in + cs * 64;
  hw = dh | (dw << 8) | ((cs * 64) << 16);
}

// block id -> bid / n_mtiles with the (mt_m, mt_s) pair of set_fast_div (wave-uniform: scalar multiply-high)
__device__ __forceinline__ int fast_div_u(int n, unsigned m, int s) {
  return s < 0 ? n : (int)(__umulhi((unsigned)n, m) >> s);
}

// ---- launch recorder (Net::describe_launches, tf2_net_describe_launches): with a recorder installed on the calling thread the
// launchers below append {kernel, grid, block, LDS bytes, registers} instead of launching -- the one place where the
// library's own kernel selection (tile shapes, fused pairs, split-K variants) can be read back, e.g. to attach rocprofv3
// rows to layers (tools/pmc_summary.py) without re-deriving any threshold.
struct LaunchRecord { char kernel[96]; unsigned grid, block; size_t lds; int vgprs; };
struct LaunchRecorder { std::vector<LaunchRecord> rows; char name[96]; };
LaunchRecorder*& launch_recorder();                     // thread-local slot (net.hip)
inline bool launch_ok() { return launch_recorder() != nullptr || hipGetLastError() == hipSuccess; }   // after TF2_LAUNCH
#define TF2_LAUNCH_NAME(...) do { if (tf2::LaunchRecorder* r_ = tf2::launch_recorder()) snprintf(r_->name, sizeof r_->name, __VA_ARGS__); } while (0)
#define TF2_LAUNCH(fn, grid_, block_, lds_, stream_, ...)                                                                     \
  do {                                                                                                                     \
    if (tf2::LaunchRecorder* r_ = tf2::launch_recorder()) {                                                                \
      tf2::LaunchRecord row_{}; snprintf(row_.kernel, sizeof row_.kernel, "%s", r_->name);                                   \
      row_.grid = dim3(grid_).x; row_.block = dim3(block_).x; row_.lds = (size_t)(lds_); row_.vgprs = -1;                       \
      hipFuncAttributes fa_; if (hipFuncGetAttributes(&fa_, reinterpret_cast<const void*>(fn)) == hipSuccess) row_.vgprs = fa_.numRegs; \
      (void)hipGetLastError(); r_->rows.push_back(row_);                                                                   \
    } else {                                                                                                               \
      hipLaunchKernelGGL(fn, grid_, block_, lds_, stream_, __VA_ARGS__);                                                       \
    }                                                                                                                      \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KiB for a kernel, once per (function, device): a process driving
// several GPUs must set it on each of them.
// (max_dynamic: a kernel with static __shared__ arrays may only ask for the rest of the 160 KiB)
inline bool lds_attr_once(const void* fn, int max_dynamic = 160 * 1024) {
  if (launch_recorder()) return true;            // describing, not launching (works without a device)
  static thread_local const void* seen_fn[64]; static thread_local int seen_dev[64]; static thread_local int n_seen = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  for (int i = 0; i < n_seen; i++) if (seen_fn[i] == fn && seen_dev[i] == dev) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, max_dynamic) != hipSuccess) return false;
  if (n_seen < 64) { seen_fn[n_seen] = fn; seen_dev[n_seen] = dev; n_seen++; }
  return true;
}

}  // namespace tf2
