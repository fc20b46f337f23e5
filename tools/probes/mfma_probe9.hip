// Micro-probe 9: the "wide" alternative to k_conv_x.hip's strict alternation - ONE wave per SIMD (4 waves per workgroup, up to 512 registers), every lane
// owns TWO edges (two B-operand sets), the tensor-product epilogue of tile t-1 is interleaved by the compiler into the MFMA stream of tile t.
// Per tile and wave: 5 K steps x 6 limb products x 2 edge sets = 60 MFMAs on six accumulators, fragments streamed from LDS (3 x ds_read_b128 per step),
// then the combine of the six accumulators (2 x 32 FMA, not overlappable: the next burst overwrites them) and NV interleaved epilogue FMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

template <int NV_PER_MFMA, bool COMBINE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const float* in, float* out, unsigned* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 b0[5][3], b1[5][3];      // h limbs of the lane's two edges (B operands)
  for (int s = 0; s < 5; ++s)
    for (int l = 0; l < 3; ++l)
      for (int i = 0; i < 8; ++i) { b0[s][l][i] = (_Float16)in[(lane * 3 + s * 8 + i + l) & 1023]; b1[s][l][i] = (_Float16)in[(lane * 5 + s * 8 + i + 2 * l) & 1023]; }
  for (int i = tid; i < 8192; i += 256) lds[i] = in[i & 1023];
  float acc[32];
  for (int k = 0; k < 32; ++k) acc[k] = in[(lane + k) & 1023];
  f32x16 Dc0, Dc1;
  for (int r = 0; r < 16; ++r) { Dc0[r] = in[(lane + r) & 1023]; Dc1[r] = in[(lane + r + 16) & 1023]; }
  __syncthreads();
  const f16x8* frag = reinterpret_cast<const f16x8*>(lds) + lane;
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    f32x16 D[6];
    for (int a = 0; a < 6; ++a)
      for (int r = 0; r < 16; ++r) D[a][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const f16x8 ah = frag[64 * (3 * s)], am = frag[64 * (3 * s + 1)], al = frag[64 * (3 * s + 2)];
      D[2] = MFMA16(ah, b0[s][2], D[2]); D[5] = MFMA16(ah, b1[s][2], D[5]);
      D[1] = MFMA16(ah, b0[s][1], D[1]); D[4] = MFMA16(ah, b1[s][1], D[4]);
      D[2] = MFMA16(al, b0[s][0], D[2]); D[5] = MFMA16(al, b1[s][0], D[5]);
      D[0] = MFMA16(ah, b0[s][0], D[0]); D[3] = MFMA16(ah, b1[s][0], D[3]);
      D[2] = MFMA16(am, b0[s][1], D[2]); D[5] = MFMA16(am, b1[s][1], D[5]);
      D[1] = MFMA16(am, b0[s][0], D[1]); D[4] = MFMA16(am, b1[s][0], D[4]);
      // the previous tile's tensor-product FMAs in the MFMAs' shadows
#pragma unroll
      for (int k = 0; k < 12 * NV_PER_MFMA; ++k) {
        const int j = (s * 12 * NV_PER_MFMA + k);
        acc[j & 31] = fmaf(j & 1 ? Dc1[(j >> 1) & 15] : Dc0[(j >> 1) & 15], acc[(j + 7) & 31], acc[j & 31]);
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV_PER_MFMA, 0); }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (COMBINE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Dc0[r] = fmaf(fmaf(D[2][r], 0.00048828125f, D[1][r]), 0.00048828125f, D[0][r]);
        Dc1[r] = fmaf(fmaf(D[5][r], 0.00048828125f, D[4][r]), 0.00048828125f, D[3][r]);
      }
    } else {
      for (int r = 0; r < 16; ++r) { Dc0[r] += D[0][r] + D[1][r] + D[2][r]; Dc1[r] += D[3][r] + D[4][r] + D[5][r]; }
    }
    __syncthreads();
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += Dc0[r] + Dc1[r];
  for (int k = 0; k < 32; ++k) sink += acc[k];
  out[blockIdx.x * 256 + tid] = sink;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NV, bool COMBINE>
void run(const char* name, float* in, float* out, unsigned* cyc) {
  const int tiles = 1000, grid = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NV, COMBINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((probe<NV, COMBINE>), dim3(grid), dim3(256), 160 * 1024, 0, in, out, cyc, 20);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<NV, COMBINE>), dim3(grid), dim3(256), 160 * 1024, 0, in, out, cyc, tiles);
  (void)hipDeviceSynchronize();
  unsigned h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 1024; ++i) m += h[i];
  m /= 1024;
  printf("%-70s %7.0f ticks per tile of 256 edges (60 MFMAs per SIMD = 1920)\n", name, m / tiles);
}

int main() {
  float *in, *out; unsigned* cyc;
  (void)hipMalloc(&in, 8192 * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 4);
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 17) * 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, true>("wide: 60 MFMAs + combine, no interleaved VALU", in, out, cyc);
  run<1, true>("wide: + 1 VALU per MFMA (60 per tile)", in, out, cyc);
  run<2, true>("wide: + 2 VALU per MFMA (120 per tile)", in, out, cyc);
  run<3, true>("wide: + 3 VALU per MFMA (180 per tile)", in, out, cyc);
  run<4, true>("wide: + 4 VALU per MFMA (240 per tile)", in, out, cyc);
  return 0;
}
