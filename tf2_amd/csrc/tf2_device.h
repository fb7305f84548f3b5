// Device-side helpers shared by the HIP kernels (include after <hip/hip_runtime.h>).
#pragma once
#include <hip/hip_runtime.h>
#include "tf2_internal.h"

namespace tf2 {

// n / d with the (m, s) pair of set_fast_div (tf2_internal.h): one v_mul_hi_u32 and a shift instead of the
// ~25-instruction integer division sequence -- the pixel decode runs once per block and lane, and the short-K
// layers are bound by VALU issue.
__device__ __forceinline__ int fast_div(int n, uint32_t m, int s) {
  return s < 0 ? n : (int)(__umulhi((unsigned)n, m) >> s);
}

// Pull the kernel arguments a conv block needs into SGPRs at kernel entry: hipcc otherwise loads each field right
// before its first use -- a dozen dependent scalar-load round trips spread over the prologue (measured on conv_mfma2:
// 2.8 k -> 1.8 k cycles from block start to the first address computation, and 3.5 k -> 1.7 k for the part after it).
// The empty asm statements pin the values, so the loads are emitted here, merged into wide s_load_dwordxN and waited
// for once.  Declares: g (ConvGeom copy), ax ay ares aw azero ahdr, a_hdr_bytes a_max_ent P a_n_mtiles mt_m mt_s.
#define TF2_PRELOAD_CONV_ARGS(a)                                                                                          \
  const ConvGeom g = (a).g;                                                                                                \
  const int8_t* const ax = (a).x; int8_t* const ay = (a).y; const int8_t* const ares = (a).res;                            \
  const int8_t* const aw = (a).w; const int8_t* const azero = (a).zero; const int32_t* const ahdr = (a).hdr;               \
  const int a_hdr_bytes = (a).hdr_bytes, a_max_ent = (a).max_ent, P = (a).n_phases, a_n_mtiles = (a).n_mtiles;             \
  const unsigned mt_m = (a).mt_m; const int mt_s = (a).mt_s;                                                               \
  asm volatile("" :: "s"(g.H), "s"(g.W), "s"(g.Cp_in), "s"(g.OW), "s"(g.OHW), "s"(g.ohw_m), "s"(g.ow_m), "s"(g.ohw_s),     \
               "s"(g.ow_s), "s"(g.stride), "s"(g.pad_h), "s"(g.pad_w), "s"(g.n_pix), "s"(g.y_cp), "s"(g.y_off), "s"(g.y_nvalid)); \
  asm volatile("" :: "s"(g.res_cp), "s"(g.res_off), "s"(g.relu), "s"(g.add_relu), "s"(g.has_res), "s"(g.fast), "s"(g.flags), \
               "s"(a_hdr_bytes), "s"(a_max_ent), "s"(P), "s"(a_n_mtiles), "s"(mt_m), "s"(mt_s));                           \
  asm volatile("" :: "s"(ax), "s"(ay), "s"(ares), "s"(aw), "s"(azero), "s"(ahdr))

// block id -> bid / n_mtiles with the (mt_m, mt_s) pair of set_fast_div (wave-uniform: scalar multiply-high)
__device__ __forceinline__ int fast_div_u(int n, unsigned m, int s) {
  return s < 0 ? n : (int)(__umulhi((unsigned)n, m) >> s);
}

// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KiB for a kernel, once per (function, device): a process driving
// several GPUs must set it on each of them.
inline bool lds_attr_once(const void* fn) {
  static thread_local const void* seen_fn[64]; static thread_local int seen_dev[64]; static thread_local int n_seen = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  for (int i = 0; i < n_seen; i++) if (seen_fn[i] == fn && seen_dev[i] == dev) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  if (n_seen < 64) { seen_fn[n_seen] = fn; seen_dev[n_seen] = dev; n_seen++; }
  return true;
}

}  // namespace tf2
