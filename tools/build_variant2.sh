set -e
NAME=$1; shift
C=disco_diffdock_amd/csrc
mkdir -p ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -Wno-unused-result -DDDK_VARIANT_BUILD -DDDK_TIMING_ONLY_BUILD "$@" -c $C/k_conv_x2.hip -o /tmp/k_conv_x2_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/libddk_$NAME.so $C/ddk_capi.o $C/k_conv.o $C/k_tp.o $C/k_graph.o $C/k_heads.o $C/k_se3.o $C/model.o $C/conf.o $C/k_conv_x.o /tmp/k_conv_x2_$NAME.o $C/k_ar.o
echo built ab_libs/libddk_$NAME.so
