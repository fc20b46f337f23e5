#!/bin/bash
# build ab_libs/libddk_<name>.so from the tree's objects with model.hip recompiled under extra flags (timing-only ablations of the launch structure):
#   tools/build_variant_model.sh conc -DDDK_ABL_CONCURRENT_LAYERS
set -e
NAME=$1; shift
C=disco_diffdock_amd/csrc
mkdir -p ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result -DDDK_TIMING_ONLY_BUILD "$@" -c $C/model.hip -o /tmp/model_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libddk_$NAME.so $C/ddk_capi.o $C/k_conv.o $C/k_tp.o $C/k_graph.o $C/k_heads.o $C/k_se3.o /tmp/model_$NAME.o $C/conf.o $C/k_conv_x.o $C/k_conv_x2.o $C/k_ar.o
echo built ab_libs/libddk_$NAME.so
