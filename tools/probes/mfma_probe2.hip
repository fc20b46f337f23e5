// Micro-probe 2: does per-tile VALU / LDS / L2-streaming work of ONE wave overlap with the fp32 MFMA burst of the OTHER
// wave of the SIMD?  Each iteration = one "weight tile": 36 dependent MFMAs, then consume D, plus
//   NV independent VALU fmas, NL ds_read_b128, and (STREAM) 9 global_load_dwordx4 of the next tile's fragments (L2-resident).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NV, int NL, bool STREAM>
__global__ __launch_bounds__(64) void probe(const float* in, const float* wts, float* out, int iters, int ntiles) {
  __shared__ float lds[64 * 36];
  float b[36];
  float4 a[9];
  const int lane = threadIdx.x;
  for (int i = 0; i < 36; ++i) { b[i] = in[(lane * 3 + i) & 1023]; lds[lane * 36 + i] = b[i]; }
  const float* w = wts + lane * 4;
  for (int s = 0; s < 9; ++s) a[s] = *reinterpret_cast<const float4*>(w + s * 256);
  __syncthreads();
  float acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = (float)k;
  float sink = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 D;
    for (int r = 0; r < 16; ++r) D[r] = b[r] * (float)it;
    const float* wn = w + (size_t)((it + 1) % ntiles) * (9 * 256);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      D = MFMA(a[s].x, b[4 * s], D); D = MFMA(a[s].y, b[4 * s + 1], D);
      D = MFMA(a[s].z, b[4 * s + 2], D); D = MFMA(a[s].w, b[4 * s + 3], D);
      if (STREAM) a[s] = *reinterpret_cast<const float4*>(wn + s * 256);
    }
    float4 f[NL > 0 ? NL : 1];
#pragma unroll
    for (int k = 0; k < NL; ++k) f[k] = *reinterpret_cast<const float4*>(lds + ((lane + it) & 63) * 36 + 4 * (k % 9));
#pragma unroll
    for (int r = 0; r < 16; ++r) sink = fmaf(D[r], b[r], sink);
#pragma unroll
    for (int k = 0; k < NL; ++k) sink += f[k].x + f[k].w;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k & 7] = fmaf(acc[k & 7], b[k % 36], b[(k + 5) % 36]);
  }
  for (int k = 0; k < 8; ++k) sink += acc[k];
  out[blockIdx.x * 64 + lane] = sink;
}

template <int NV, int NL, bool STREAM>
void run(int waves_per_simd, float* in, float* wts, float* out) {
  const int iters = 4000, grid = 256 * 4 * waves_per_simd, ntiles = 248;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<NV, NL, STREAM>), dim3(grid), dim3(64), 0, 0, in, wts, out, 100, ntiles);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<NV, NL, STREAM>), dim3(grid), dim3(64), 0, 0, in, wts, out, iters, ntiles);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double flop = (double)grid * iters * 36 * 4096.0;
  printf("NV=%4d NL=%2d stream=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", NV, NL, (int)STREAM, waves_per_simd, ms, flop / ms / 1e9);
}

int main() {
  float *in, *out, *wts;
  const size_t wbytes = (size_t)249 * 9 * 256 * 4;
  hipMalloc(&in, 4096); hipMalloc(&out, 256 * 4 * 8 * 64 * 4); hipMalloc(&wts, wbytes);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 1000) / 500.f - 1.f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  hipMemset(wts, 0, wbytes);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, false>(w, in, wts, out);
    run<64, 0, false>(w, in, wts, out);
    run<128, 0, false>(w, in, wts, out);
    run<256, 0, false>(w, in, wts, out);
    run<512, 0, false>(w, in, wts, out);
    run<128, 12, false>(w, in, wts, out);
    run<0, 0, true>(w, in, wts, out);
    run<128, 12, true>(w, in, wts, out);
    run<256, 12, true>(w, in, wts, out);
  }
  return 0;
}
