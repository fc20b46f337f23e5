#!/bin/bash
# build ab_libs/libddk_<name>.so WITH tools/variants/k_conv_y.hip (round 5's one-wave-per-SIMD conv kernel; ddk_config.conv_kernel = 2 selects it in this
# build only): k_conv_x.hip and ddk_capi.hip recompiled with -DDDK_VARIANT_CONV_Y, the variant kernel under extra flags:  tools/build_variant_y.sh <name> [-DFOO ...]
set -e
NAME=$1; shift
C=disco_diffdock_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result -DDDK_VARIANT_CONV_Y -I$C -Iinclude"
mkdir -p ab_libs
/opt/rocm/bin/hipcc $F -fno-slp-vectorize "$@" -c tools/variants/k_conv_y.hip -o /tmp/k_conv_y_$NAME.o
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c $C/k_conv_x.hip -o /tmp/k_conv_xy_$NAME.o
/opt/rocm/bin/hipcc $F -c $C/ddk_capi.hip -o /tmp/ddk_capi_y_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libddk_$NAME.so /tmp/ddk_capi_y_$NAME.o $C/k_conv.o $C/k_tp.o $C/k_graph.o $C/k_heads.o $C/k_se3.o $C/model.o $C/conf.o /tmp/k_conv_xy_$NAME.o /tmp/k_conv_y_$NAME.o $C/k_conv_x2.o $C/k_ar.o
echo built ab_libs/libddk_$NAME.so
