/* TEST INFRASTRUCTURE.  The reference's PE arithmetic -- MUL and DotProduct, device/src/pe.cl:27-49 -- compiled as C from the source where
 * it lies (oracle/Makefile `ref`: gcc -std=gnu11 -fgnu89-inline, -DTF2_PE_CL='"<reference>/cnn/device/src/pe.cl"') and exported for ctypes, so
 * that oracle/tf2_oracle.c's tf2o_mul / dot products are checked LIVE against the reference's own expressions (tests/test_oracle_vs_ref.py),
 * not only against three constants.  The OpenCL compiler the file was written for (Intel's aoc) is
 * absent; what this file adds is the handful of OpenCL-C keywords and the Altera channel intrinsics as macros -- pe.cl's other kernels
 * (PeFunction, channels, autorun attributes) only have to PARSE, they are never called.  No reference text is copied: pe.cl and the headers it
 * includes (host/inc/cnn.h -> archs.h, defines.h, types.h, resnet50.h) are read in place. */
#include <stdbool.h>                                /* OpenCL C has bool / true / false built in */
typedef unsigned char uchar;
typedef unsigned short ushort;
typedef unsigned int uint;
typedef unsigned long ulong;
#define constant static const
#define kernel
#define global
#define __global
#define local
#define restrict __restrict
#define channel static
#define read_channel_altera(c) (c)
#define write_channel_altera(c, v) ((c) = (v))
#define read_channel_nb_altera(c, pvalid) (*(pvalid) = 1, (c))
#define write_channel_nb_altera(c, v) ((c) = (v), 1)
#define read_channel_intel(c) (c)
#define write_channel_intel(c, v) ((c) = (v))
#define mem_fence(x)
#define barrier(x)
#define CLK_CHANNEL_MEM_FENCE 0
#define CLK_GLOBAL_MEM_FENCE 0
#define CLK_LOCAL_MEM_FENCE 0
#define get_compute_id(x) 0
#ifndef max
#define max(a, b) ((a) > (b) ? (a) : (b))
#endif
#ifndef min
#define min(a, b) ((a) < (b) ? (a) : (b))
#endif
#define OPENCL
#ifndef RESNET50
#define RESNET50
#endif
#include TF2_PE_CL

/* MUL(feature, filter): pe.cl:27-40 */
int tf2ref_pe_mul(int feature, int filter) { return (int)MUL((real)feature, (real)filter); }
/* DotProduct over C_VECTOR lanes: pe.cl:42-49 */
int tf2ref_pe_c_vector(void) { return C_VECTOR; }
int tf2ref_pe_dot(const signed char* feature, const unsigned char* filter) {
  DotVector f, w;
  for (int i = 0; i < C_VECTOR; i++) { f.v[i] = (real)feature[i]; w.v[i] = (real)filter[i]; }
  return (int)DotProduct(f, w);
}
