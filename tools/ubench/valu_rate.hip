#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k_v_mad_i64_i32(long long* out, int* sink, int seed) {
  long long a0 = (long long)(seed + 0 + (int)threadIdx.x);
  long long a1 = (long long)(seed + 1 + (int)threadIdx.x);
  long long a2 = (long long)(seed + 2 + (int)threadIdx.x);
  long long a3 = (long long)(seed + 3 + (int)threadIdx.x);
  long long a4 = (long long)(seed + 4 + (int)threadIdx.x);
  long long a5 = (long long)(seed + 5 + (int)threadIdx.x);
  long long a6 = (long long)(seed + 6 + (int)threadIdx.x);
  long long a7 = (long long)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mul_lo_u32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mul_hi_i32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mul_i32_i24(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mad_i32_i24(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mad_u32_u16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_lshl_add_u32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_lshl_add_u32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_alignbit_b32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_add_i32_clamp(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_ashrrev_i32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_med3_i32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_perm_b32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_bfe_i32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_bfe_i32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_fma_f32(long long* out, int* sink, int seed) {
  float a0 = (float)(seed + 0 + (int)threadIdx.x);
  float a1 = (float)(seed + 1 + (int)threadIdx.x);
  float a2 = (float)(seed + 2 + (int)threadIdx.x);
  float a3 = (float)(seed + 3 + (int)threadIdx.x);
  float a4 = (float)(seed + 4 + (int)threadIdx.x);
  float a5 = (float)(seed + 5 + (int)threadIdx.x);
  float a6 = (float)(seed + 6 + (int)threadIdx.x);
  float a7 = (float)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_pk_fma_f32(long long* out, int* sink, int seed) {
  double a0 = (double)(seed + 0 + (int)threadIdx.x);
  double a1 = (double)(seed + 1 + (int)threadIdx.x);
  double a2 = (double)(seed + 2 + (int)threadIdx.x);
  double a3 = (double)(seed + 3 + (int)threadIdx.x);
  double a4 = (double)(seed + 4 + (int)threadIdx.x);
  double a5 = (double)(seed + 5 + (int)threadIdx.x);
  double a6 = (double)(seed + 6 + (int)threadIdx.x);
  double a7 = (double)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_cvt_f32_i32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_cvt_i32_f32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_floor_f32(long long* out, int* sink, int seed) {
  float a0 = (float)(seed + 0 + (int)threadIdx.x);
  float a1 = (float)(seed + 1 + (int)threadIdx.x);
  float a2 = (float)(seed + 2 + (int)threadIdx.x);
  float a3 = (float)(seed + 3 + (int)threadIdx.x);
  float a4 = (float)(seed + 4 + (int)threadIdx.x);
  float a5 = (float)(seed + 5 + (int)threadIdx.x);
  float a6 = (float)(seed + 6 + (int)threadIdx.x);
  float a7 = (float)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_floor_f32 %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f32 %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_fract_f32(long long* out, int* sink, int seed) {
  float a0 = (float)(seed + 0 + (int)threadIdx.x);
  float a1 = (float)(seed + 1 + (int)threadIdx.x);
  float a2 = (float)(seed + 2 + (int)threadIdx.x);
  float a3 = (float)(seed + 3 + (int)threadIdx.x);
  float a4 = (float)(seed + 4 + (int)threadIdx.x);
  float a5 = (float)(seed + 5 + (int)threadIdx.x);
  float a6 = (float)(seed + 6 + (int)threadIdx.x);
  float a7 = (float)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_fract_f32 %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fract_f32 %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_fma_f64(long long* out, int* sink, int seed) {
  double a0 = (double)(seed + 0 + (int)threadIdx.x);
  double a1 = (double)(seed + 1 + (int)threadIdx.x);
  double a2 = (double)(seed + 2 + (int)threadIdx.x);
  double a3 = (double)(seed + 3 + (int)threadIdx.x);
  double a4 = (double)(seed + 4 + (int)threadIdx.x);
  double a5 = (double)(seed + 5 + (int)threadIdx.x);
  double a6 = (double)(seed + 6 + (int)threadIdx.x);
  double a7 = (double)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_cvt_f64_i32(long long* out, int* sink, int seed) {
  double a0 = (double)(seed + 0 + (int)threadIdx.x);
  double a1 = (double)(seed + 1 + (int)threadIdx.x);
  double a2 = (double)(seed + 2 + (int)threadIdx.x);
  double a3 = (double)(seed + 3 + (int)threadIdx.x);
  double a4 = (double)(seed + 4 + (int)threadIdx.x);
  double a5 = (double)(seed + 5 + (int)threadIdx.x);
  double a6 = (double)(seed + 6 + (int)threadIdx.x);
  double a7 = (double)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_cvt_i32_f64(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_i32_f64 %0, %3" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_floor_f64(long long* out, int* sink, int seed) {
  double a0 = (double)(seed + 0 + (int)threadIdx.x);
  double a1 = (double)(seed + 1 + (int)threadIdx.x);
  double a2 = (double)(seed + 2 + (int)threadIdx.x);
  double a3 = (double)(seed + 3 + (int)threadIdx.x);
  double a4 = (double)(seed + 4 + (int)threadIdx.x);
  double a5 = (double)(seed + 5 + (int)threadIdx.x);
  double a6 = (double)(seed + 6 + (int)threadIdx.x);
  double a7 = (double)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_floor_f64 %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_floor_f64 %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_pk_mul_lo_u16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_pk_mad_i16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_pk_add_i16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_pk_ashrrev_i16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_pk_max_i16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_dot4_i32_i8(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mov_b32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mov_b32 %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mov_b32 %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_cmp_gt_f32(long long* out, int* sink, int seed) {
  float a0 = (float)(seed + 0 + (int)threadIdx.x);
  float a1 = (float)(seed + 1 + (int)threadIdx.x);
  float a2 = (float)(seed + 2 + (int)threadIdx.x);
  float a3 = (float)(seed + 3 + (int)threadIdx.x);
  float a4 = (float)(seed + 4 + (int)threadIdx.x);
  float a5 = (float)(seed + 5 + (int)threadIdx.x);
  float a6 = (float)(seed + 6 + (int)threadIdx.x);
  float a7 = (float)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_permlane32_swap(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_cvt_pk_i16_i32(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
__global__ void k_v_mad_i32_i16(long long* out, int* sink, int seed) {
  int a0 = (int)(seed + 0 + (int)threadIdx.x);
  int a1 = (int)(seed + 1 + (int)threadIdx.x);
  int a2 = (int)(seed + 2 + (int)threadIdx.x);
  int a3 = (int)(seed + 3 + (int)threadIdx.x);
  int a4 = (int)(seed + 4 + (int)threadIdx.x);
  int a5 = (int)(seed + 5 + (int)threadIdx.x);
  int a6 = (int)(seed + 6 + (int)threadIdx.x);
  int a7 = (int)(seed + 7 + (int)threadIdx.x);
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a0) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a1) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a2) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a3) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a4) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a5) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a6) : "v"(b), "v"(c), "v"(dd) : "vcc");
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "vcc");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}
int main() {
  long long* out; int* sink; hipMalloc(&out, 64); hipMalloc(&sink, 64);
  long long h;
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_mad_i64_i32<<<1, nt[v]>>>(out, sink, 3); k_v_mad_i64_i32<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_mad_i64_i32", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_mul_lo_u32<<<1, nt[v]>>>(out, sink, 3); k_v_mul_lo_u32<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_mul_lo_u32", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_lshl_add_u32<<<1, nt[v]>>>(out, sink, 3); k_v_lshl_add_u32<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_lshl_add_u32", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_med3_i32<<<1, nt[v]>>>(out, sink, 3); k_v_med3_i32<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_med3_i32", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_fma_f32<<<1, nt[v]>>>(out, sink, 3); k_v_fma_f32<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_fma_f32", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_mov_b32<<<1, nt[v]>>>(out, sink, 3); k_v_mov_b32<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_mov_b32", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  { double r[6]; int nt[6] = {64, 256, 512, 768, 1024, 1024};
    for (int v = 0; v < 5; v++) { k_v_permlane32_swap<<<1, nt[v]>>>(out, sink, 3); k_v_permlane32_swap<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\n", "v_permlane32_swap", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }
  return 0;
}
