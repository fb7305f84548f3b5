#!/bin/bash
# Timing-experiment builds of the chain kernel (conv_mfma2.hip TF2_CHAIN_VAR): tf2_amd/libtf2amd_cv<N>.so, loaded through TF2_AMD_LIB.
# Results of these libraries are wrong by construction (they drop pieces of the coherence protocol); only rates are read.
cd "$(dirname "$0")/../tf2_amd/csrc" || exit 1
make -s ARCH=gfx950 || exit 1
for v in "$@"; do
  ( ( [ build/conv_mfma2_cv$v.o -nt conv_mfma2.hip ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DTF2_CHAIN_VAR=$((v & 15)) -DTF2_CHAIN_SLEEP=$(( (v >> 4) ? (v >> 4) : 2 )) -x hip -c conv_mfma2.hip -o build/conv_mfma2_cv$v.o ) &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libtf2amd_cv$v.so build/conv_mfma2_cv$v.o $(for f in conv_bneck.hip conv_stem.hip conv_mfma_sk.hip conv_pw.hip conv_shift.hip misc_kernels.hip net.hip host_model.cpp model4bit.cpp weight_pack.cpp capi.cpp; do echo build/$f.o; done) ) &
done
wait
ls -la ../libtf2amd_cv*.so
