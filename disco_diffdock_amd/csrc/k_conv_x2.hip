// The two-limb form of the f16-limb conv kernel - the default, ddk_config.conv_kernel = 0 (include/ddk.h): k_conv_x.hip compiled a second time with two fp16
// limbs per fp32 operand (hi = fp16(x), mid = fp16(x - hi), both rounded to nearest: |x - hi - mid| <= 2^-22 |x|) and the three limb products
// hi.hi + hi.mid + mid.hi in one fp32 accumulator - 14 MFMAs per W2 tile instead of 27, no third limb to split / read, no second accumulator to fold.  The
// dropped mid.mid is <= 2^-22 relative, the size of the operands' own truncation: a product is off by <= 3 * 2^-22, a K = 72 dot product by less than the
// classical bound 72 * 2^-24 of an fp32 FMA chain.  Same tile records (the third limb of a record is simply not read), same tables, same epilogue stream
// (tools/gen_conv_x_epi.py with GEN_ONE_ACC=1 -> k_conv_x_epi2_gen.inc).  Kernels conv_x2_kernel<...>, entry points launch_conv_fused_x2 /
// conv_prepare_device_x2.  Reference: models/tensor_layers.py:140-143,154-155.
#define X3_TWO_LIMBS 1
#include "k_conv_x.hip"
