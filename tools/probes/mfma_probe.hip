// Micro-probe: what the fp32 MFMA pipe sustains for the access patterns of the fused conv kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// MODE 0: one dependent chain of 36, then consume D (16 reads) and restart from a fresh accumulator
// MODE 1: two accumulators alternating (18 + 18), summed at the end
// MODE 2: chain of 36 never consumed inside the loop (pure pipe rate)
// MODE 3: like 0 but the consumer work (16 fma) is placed after the NEXT chain (software pipelined, two acc sets)
template <int MODE>
__global__ __launch_bounds__(64) void probe(const float* in, float* out, int iters) {
  float a[36], b[36];
  for (int i = 0; i < 36; ++i) { a[i] = in[(threadIdx.x + i) & 1023]; b[i] = in[(threadIdx.x * 3 + i) & 1023]; }
  float sink = 0.f;
  f32x16 Dp;
  for (int r = 0; r < 16; ++r) Dp[r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 D;
    for (int r = 0; r < 16; ++r) D[r] = a[r] * (float)it;
    if (MODE == 1) {
      f32x16 E;
      for (int r = 0; r < 16; ++r) E[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 36; s += 2) { D = MFMA(a[s], b[s], D); E = MFMA(a[s + 1], b[s + 1], E); }
      for (int r = 0; r < 16; ++r) sink += D[r] + E[r];
    } else if (MODE == 2) {
#pragma unroll
      for (int s = 0; s < 36; ++s) Dp = MFMA(a[s], b[s], Dp);
    } else if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 36; ++s) D = MFMA(a[s], b[s], D);
      for (int r = 0; r < 16; ++r) sink = fmaf(Dp[r], b[r], sink);
      Dp = D;
    } else {
#pragma unroll
      for (int s = 0; s < 36; ++s) D = MFMA(a[s], b[s], D);
      for (int r = 0; r < 16; ++r) sink = fmaf(D[r], b[r], sink);
    }
  }
  if (MODE == 2 || MODE == 3) for (int r = 0; r < 16; ++r) sink += Dp[r];
  out[blockIdx.x * 64 + threadIdx.x] = sink;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* in, float* out) {
  const int iters = 4000, grid = 256 * 4 * waves_per_simd;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), 0, 0, in, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), 0, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flop = (double)grid * iters * 36 * 4096.0;
  printf("%-44s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", name, waves_per_simd, ms, flop / ms / 1e9);
}

int main() {
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
  hipMemset(in, 0, 4096);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 1000) / 500.f - 1.f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<2>("pure dependent chain (never consumed)", w, in, out);
    run<0>("chain of 36 then consume D", w, in, out);
    run<1>("two alternating accumulators", w, in, out);
    run<3>("chain of 36, consume previous D (pipelined)", w, in, out);
  }
  return 0;
}
