#!/bin/bash
# build ab_libs/libddk_<name>.so from the tree's objects with k_conv_x.hip recompiled under extra flags:  tools/build_variant.sh <name> [-DFOO ...]
# (-DDDK_VARIANT_BUILD -DDDK_TIMING_ONLY_BUILD unlock the experiment / timing-only switches, which #error in the product build)
set -e
NAME=$1; shift
C=disco_diffdock_amd/csrc
mkdir -p ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -Wno-unused-result -DDDK_VARIANT_BUILD -DDDK_TIMING_ONLY_BUILD "$@" -c $C/k_conv_x.hip -o /tmp/k_conv_x_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libddk_$NAME.so $C/ddk_capi.o $C/k_conv.o $C/k_tp.o $C/k_graph.o $C/k_heads.o $C/k_se3.o $C/model.o $C/conf.o /tmp/k_conv_x_$NAME.o $C/k_conv_x2.o $C/k_ar.o
echo built ab_libs/libddk_$NAME.so
