// vm_track.h -- counted waits on LDS-DMAs that are issued as inline assembly (conv_bband.hip, conv_c3.hip), gfx950.
//
// hipcc drains the whole VMEM queue (s_waitcnt vmcnt(0)) in front of the first LDS read behind an LDS-DMA builtin, so those kernels
// issue their DMAs as inline assembly, which the compiler's wait-count pass does not see -- and have to write every wait for them out.
// The VM counter retires IN ORDER: `s_waitcnt vmcnt(N)` returns when at most N of the wave's VM operations are outstanding, i.e. when
// everything but the N youngest has completed.  A wait for a group of DMAs is therefore right exactly when
//
//        N  <=  (VM operations this wave has issued BEHIND the last DMA of the group)
//
// -- one too many and the wait is bit-exact on an idle chip and wrong when the DMAs land late (twice in round 4: b53d5bf, 7173314).
// Round 5: no kernel writes such an N as a literal any more.  Each kernel states its per-wave VM schedule ONCE (a constexpr function
// of the step index that both the issue code and the wait read, or a run-time count of what was really issued) and waits through the
// helpers below; tests/test_vmcnt_isa.py then checks the COMPILED code: for every hand-written vmcnt wait of the shipped kernels, on
// every path of the control-flow graph, at least N VM instructions lie between the last DMA instruction and the wait.
// A -DTF2_CHECK_DMA build (make -C tf2_amd/csrc check -> libtf2amd_check.so, loaded by tools/dma_stress.py only) stamps a sentinel
// into every chunk buffer before its DMA is issued and counts fragment reads that still see it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace tf2 {

// s_waitcnt vmcnt(N), N a compile-time value (the counter has six bits on gfx9: 0..63)
template <int N>
__device__ __forceinline__ void vm_wait() {
  static_assert(N >= 0 && N <= 63, "vmcnt has six bits");
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

constexpr int vm_min(int a, int b) { return a < b ? a : b; }

// at most min(k, KMAX) * STEP operations outstanding, k a wave-uniform RUN-TIME count (e.g. groups of STEP DMAs issued since)
template <int STEP, int KMAX>
__device__ __forceinline__ void vm_wait_groups(int k) {
  if constexpr (KMAX == 0) vm_wait<0>();
  else {
    if (k >= KMAX) vm_wait<vm_min(KMAX * STEP, 63)>();
    else vm_wait_groups<STEP, KMAX - 1>(k);
  }
}

// ---- -DTF2_CHECK_DMA: sentinel stamping ---------------------------------------------------------------------------------------
// The tensors these kernels stream are post-ReLU (0..127) or "doubled" (2x - 128: even), their pad rows hold 0 or -128: the byte
// 0x81 (-127) never occurs in them, so a 16-byte fragment of 0x81 is a chunk buffer whose DMA has not landed.
constexpr unsigned kDmaSentinel = 0x81818181u;
#ifdef TF2_CHECK_DMA
// {fragments that still held the sentinel, fragment groups checked}: one pair per translation unit, summed by tf2_check_dma_errors
#define TF2_DMA_CHECK_COUNTERS(name) __device__ unsigned long long name[2]
// before a 1 KiB DMA whose lanes fill dst + 16 * lane: the same bytes filled with the sentinel (complete before the DMA is issued)
__device__ __forceinline__ void dma_stamp(int8_t* lds_dst) {
  using u32x4 = unsigned __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u32x4*>(lds_dst + 16 * (threadIdx.x & 63)) = u32x4{kDmaSentinel, kDmaSentinel, kDmaSentinel, kDmaSentinel};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <class V4>
__device__ __forceinline__ void dma_check(const V4& frag, unsigned long long* counters) {
  const bool stale = (unsigned)frag[0] == kDmaSentinel && (unsigned)frag[1] == kDmaSentinel && (unsigned)frag[2] == kDmaSentinel && (unsigned)frag[3] == kDmaSentinel;
  if (stale) atomicAdd(&counters[0], 1ull);
  if ((threadIdx.x & 63) == 0) atomicAdd(&counters[1], 1ull);
}
#endif

}  // namespace tf2
