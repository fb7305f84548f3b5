#!/usr/bin/env python3
"""Generates valu_rate.hip: issue-rate microbenchmark of VALU ops the requant epilogue can be built from."""
OPS = [  # name, accumulator type, asm template (%0 acc in/out, %1 %2 inputs)
 ("v_mad_i64_i32", "long long", "v_mad_i64_i32 %0, vcc, %1, %2, %0"),
 ("v_mul_lo_u32", "int", "v_mul_lo_u32 %0, %0, %1"),
 ("v_mul_hi_i32", "int", "v_mul_hi_i32 %0, %0, %1"),
 ("v_mul_i32_i24", "int", "v_mul_i32_i24 %0, %0, %1"),
 ("v_mad_i32_i24", "int", "v_mad_i32_i24 %0, %1, %2, %0"),
 ("v_mad_u32_u16", "int", "v_mad_u32_u16 %0, %1, %2, %0"),
 ("v_lshl_add_u32", "int", "v_lshl_add_u32 %0, %0, %1, %2"),
 ("v_alignbit_b32", "int", "v_alignbit_b32 %0, %0, %1, %2"),
 ("v_add_i32_clamp", "int", "v_add_i32 %0, %0, %1 clamp"),
 ("v_ashrrev_i32", "int", "v_ashrrev_i32 %0, %1, %0"),
 ("v_med3_i32", "int", "v_med3_i32 %0, %0, %1, %2"),
 ("v_perm_b32", "int", "v_perm_b32 %0, %0, %1, %2"),
 ("v_bfe_i32", "int", "v_bfe_i32 %0, %0, %1, %2"),
 ("v_fma_f32", "float", "v_fma_f32 %0, %1, %2, %0"),
 ("v_pk_fma_f32", "double", "v_pk_fma_f32 %0, %0, %0, %0"),
 ("v_cvt_f32_i32", "int", "v_cvt_f32_i32 %0, %0"),
 ("v_cvt_i32_f32", "int", "v_cvt_i32_f32 %0, %0"),
 ("v_floor_f32", "float", "v_floor_f32 %0, %0"),
 ("v_fract_f32", "float", "v_fract_f32 %0, %0"),
 ("v_fma_f64", "double", "v_fma_f64 %0, %0, %0, %0"),
 ("v_cvt_f64_i32", "double", "v_cvt_f64_i32 %0, %1"),
 ("v_cvt_i32_f64", "int", "v_cvt_i32_f64 %0, %3"),
 ("v_floor_f64", "double", "v_floor_f64 %0, %0"),
 ("v_pk_mul_lo_u16", "int", "v_pk_mul_lo_u16 %0, %0, %1"),
 ("v_pk_mad_i16", "int", "v_pk_mad_i16 %0, %1, %2, %0"),
 ("v_pk_add_i16", "int", "v_pk_add_i16 %0, %0, %1"),
 ("v_pk_ashrrev_i16", "int", "v_pk_ashrrev_i16 %0, %1, %0"),
 ("v_pk_max_i16", "int", "v_pk_max_i16 %0, %0, %1"),
 ("v_dot4_i32_i8", "int", "v_dot4_i32_i8 %0, %1, %2, %0"),
 ("v_mov_b32", "int", "v_mov_b32 %0, %1"),
 ("v_cmp_gt_f32", "float", "v_cmp_gt_f32 vcc, %0, %1"),
 ("v_permlane32_swap", "int", "v_permlane32_swap_b32 %0, %0"),
 ("v_cvt_pk_i16_i32", "int", "v_cvt_pk_i16_i32 %0, %0, %1"),
 ("v_mad_i32_i16", "int", "v_mad_i32_i16 %0, %1, %2, %0"),
]
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstring>']
for name, ty, tpl in OPS:
    body = []
    for k in range(8):
        body.append(f'    asm volatile("{tpl}" : "+v"(a{k}) : "v"(b), "v"(c), "v"(dd) : "vcc");')
    decl = "\n".join(f"  {ty} a{k} = ({ty})(seed + {k} + (int)threadIdx.x);" for k in range(8))
    src.append(f"""__global__ void k_{name}(long long* out, int* sink, int seed) {{
{decl}
  int b = seed * 3 + 1, c = seed + 7; double dd = (double)seed;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 512; it++) {{
{chr(10).join(body)}
  }}
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if ((long long)a0 + (long long)a1 + (long long)a2 + (long long)a3 + (long long)a4 + (long long)a5 + (long long)a6 + (long long)a7 == 0x1234567) sink[0] = 1;
}}""")
src.append("int main() {\n  long long* out; int* sink; hipMalloc(&out, 64); hipMalloc(&sink, 64);\n  long long h;")
for name, ty, tpl in OPS:
    if name not in ("v_mad_i64_i32","v_lshl_add_u32","v_fma_f32","v_med3_i32","v_mov_b32","v_permlane32_swap","v_mul_lo_u32"): continue
    src.append(f"""  {{ double r[6]; int nt[6] = {{64, 256, 512, 768, 1024, 1024}};
    for (int v = 0; v < 5; v++) {{ k_{name}<<<1, nt[v]>>>(out, sink, 3); k_{name}<<<1, nt[v]>>>(out, sink, 3); hipDeviceSynchronize();
      hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); r[v] = (double)h / (512.0 * 8); }}
    printf("%-20s ticks/instr/wave at 1 wave, 1,2,3,4 waves/SIMD: %6.2f %6.2f %6.2f %6.2f %6.2f  -> per SIMD-instr %5.2f %5.2f %5.2f %5.2f\\n", "{name}", r[0], r[1], r[2], r[3], r[4], r[1], r[2] / 2, r[3] / 3, r[4] / 4); }}""")
src.append("  return 0;\n}")
open(__file__.replace("gen_valu_rate.py", "valu_rate.hip"), "w").write("\n".join(src) + "\n")
