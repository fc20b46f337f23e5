// The four-product form of the f16-limb conv kernel (ddk_config.conv_kernel, include/ddk.h): k_conv_x.hip compiled a second time with two limbs per
// operand (hi + mid: 22 bits of significand and the sign of the remainder) and the four products hi.hi + hi.mid + mid.hi + mid.mid in one fp32
// accumulator - 18 MFMAs per W2 tile instead of 27, no third limb to split / read, no second accumulator to fold.  Same tile records (the third limb of
// a record is simply not read), same tables, same epilogue stream (tools/gen_conv_x_epi.py with GEN_ONE_ACC=1 -> k_conv_x_epi4_gen.inc).
// Kernels conv_x2_kernel<...>, entry points launch_conv_fused_x4 / conv_prepare_device_x4.  Reference: models/tensor_layers.py:140-143,154-155.
#define X3_P4 1
#include "k_conv_x.hip"
