// TEST INFRASTRUCTURE (oracle side) -- not part of the product path.
//
// Table dumper: compiled against the REFERENCE's own headers where they lie
// (/root/reference/Runtime_Engine/cnn/host/inc/{archs,defines,types}.h and a
// network header <net>.h selected with -DNET_HEADER="..."), it prints every
// per-layer k* table and the size macros as JSON.  The output is the golden
// vector for tf2_amd.config (the <net>.h parser and the fpganetwork.bin
// reader).  Built only by oracle/Makefile into oracle/_ref/; no reference
// source is copied into this repository.
#include <cstdio>
#include <cstring>
#include "archs.h"
#include "defines.h"
#include "types.h"
#include NET_HEADER

template <typename T, size_t N>
static void dump(const char* name, const T (&a)[N], bool last = false) {
  printf("  \"%s\": [", name);
  for (size_t i = 0; i < N; i++) printf("%s%ld", i ? "," : "", (long)a[i]);
  printf("]%s\n", last ? "" : ",");
}
#define D(x) dump(#x, x)
#define M(x) printf("  \"%s\": %ld,\n", #x, (long)(x))

int main() {
  printf("{\n");
  M(NUM_LAYER); M(NUM_CONVOLUTIONS); M(NUM_Q_LAYERS);
  M(INPUT_IMAGE_C); M(INPUT_IMAGE_H); M(INPUT_IMAGE_W); M(FIRST_FILTER_SIZE);
  M(MAX_OUT_CHANNEL); M(POOL_WINDOW_MAX); M(MAX_FILTER_SIZE); M(MAX_BIAS_SIZE);
  M(OUTPUT_OFFSET); M(DDR_BLOCK_SIZE); M(CACHE_PAGE_SIZE);
  M(N_VECTOR); M(C_VECTOR); M(OW_VECTOR); M(FW_VECTOR); M(W_VECTOR);
  M(INFLAT); M(ALPHA_INFLAT);
  D(kCacheReadBase); D(kCacheWriteBase); D(kDDRReadBase); D(kDDRWriteBase);
  D(kCacheWriteEnable); D(kDDRWriteEnable); D(kEndPoolEnable);
  D(kAdditionEnable); D(kAdditionReluEnable); D(kReluEnable);
  D(kFilterSize); D(kPadWidth); D(kPadHeight);
  D(kInputWidth); D(kInputHeight); D(kOutputWidth); D(kOutputHeight);
  D(kInputChannels); D(kOutputChannels); D(kConvStride);
  D(kIpoolEnable); D(kPoolEnable); D(kBiasEnable); D(kPoolWindow); D(kPoolType);
  D(kPoolStride2); D(kPoolOutputWidth); D(kPoolOutputHeight); D(kPoolPad);
  D(kNStart); D(kNEnd); D(kBnEnable); D(kInputLayer); D(kBranchTail);
  dump("kConcatLayer", kConcatLayer, true);
  printf("}\n");
  return 0;
}
