// Micro-probe 4: the GEMM2 tile loop of the fused conv kernel in isolation (no GEMM1, no F build, no flush):
//   MODE 0: A fragments reloaded in place after the 4 MFMAs that consumed them (v5 scheme)
//   MODE 1: next tile's fragments + bias requested in ONE batch before the burst into a second register set (v6 scheme)
// random (non-zero) weights, NT tiles per pass, bias as the C operand, tiny epilogue.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ void load_frags(float4 (&a)[9], f32x16& B, const float* w, const float* b) {
#pragma unroll
  for (int s4 = 0; s4 < 9; ++s4) a[s4] = ld4(w + s4 * 256);
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float4 v = ld4(b + 4 * j); B[4 * j] = v.x; B[4 * j + 1] = v.y; B[4 * j + 2] = v.z; B[4 * j + 3] = v.w; }
}

template <int MODE, int NV>
__global__ __launch_bounds__(64) void probe(const float* in, const float* wts, const float* bias, float* out, int passes, int nt) {
  float h[36];
  const int lane = threadIdx.x, hh = lane >> 5;
  for (int i = 0; i < 36; ++i) h[i] = in[(lane * 3 + i) & 1023];
  const float* w2 = wts + lane * 4;
  const float* b2 = bias + hh * 16;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = 0; p < passes; ++p) {
    if (MODE == 0) {
      float4 a[9]; f32x16 B;
      load_frags(a, B, w2, b2);
      for (int t = 0; t < nt; ++t) {
        const int tn = min(t + 1, nt - 1);
        const float* wn = w2 + (size_t)tn * 2304;
        f32x16 D = B;
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
          D = MFMA(a[s4].x, h[4 * s4], D); D = MFMA(a[s4].y, h[4 * s4 + 1], D);
          D = MFMA(a[s4].z, h[4 * s4 + 2], D); D = MFMA(a[s4].w, h[4 * s4 + 3], D);
          a[s4] = ld4(wn + s4 * 256);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 v = ld4(b2 + tn * 32 + 4 * j); B[4 * j] = v.x; B[4 * j + 1] = v.y; B[4 * j + 2] = v.z; B[4 * j + 3] = v.w; }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r & 7] = fmaf(D[r], h[r], acc[r & 7]);
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k & 7] = fmaf(acc[k & 7], h[k % 36], h[(k + 5) % 36]);
      }
    } else {
      float4 a0[9], a1[9]; f32x16 B0, B1;
      load_frags(a0, B0, w2, b2);
#define TILE(T, AC, BC, AN, BN)                                              \
      {                                                                      \
        const int tn = min((T) + 1, nt - 1);                                 \
        load_frags(AN, BN, w2 + (size_t)tn * 2304, b2 + tn * 32);            \
        __builtin_amdgcn_sched_barrier(0);                                   \
        f32x16 D = MFMA(AC[0].x, h[0], BC);                                  \
        D = MFMA(AC[0].y, h[1], D); D = MFMA(AC[0].z, h[2], D); D = MFMA(AC[0].w, h[3], D); \
        _Pragma("unroll") for (int s4 = 1; s4 < 9; ++s4) {                   \
          D = MFMA(AC[s4].x, h[4 * s4], D); D = MFMA(AC[s4].y, h[4 * s4 + 1], D); \
          D = MFMA(AC[s4].z, h[4 * s4 + 2], D); D = MFMA(AC[s4].w, h[4 * s4 + 3], D); \
        }                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[r & 7] = fmaf(D[r], h[r], acc[r & 7]); \
        _Pragma("unroll") for (int k = 0; k < NV; ++k) acc[k & 7] = fmaf(acc[k & 7], h[k % 36], h[(k + 5) % 36]); \
      }
      for (int t = 0; t < nt; t += 2) {
        TILE(t, a0, B0, a1, B1)
        if (t + 1 >= nt) break;
        TILE(t + 1, a1, B1, a0, B0)
      }
    }
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 64 + lane] = s;
}

template <int MODE, int NV>
void run(int waves_per_simd, float* in, float* wts, float* bias, float* out) {
  const int nt = 66, passes = 60, grid = 256 * 4 * waves_per_simd;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE, NV>), dim3(grid), dim3(64), 0, 0, in, wts, bias, out, 2, nt);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, NV>), dim3(grid), dim3(64), 0, 0, in, wts, bias, out, passes, nt);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("mode=%d NV=%3d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", MODE, NV, waves_per_simd, ms, (double)grid * passes * nt * 36 * 4096.0 / ms / 1e9);
}

int main() {
  float *in, *out, *wts, *bias;
  const size_t wn = (size_t)66 * 2304, bn = 66 * 32;
  (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4); (void)hipMalloc(&wts, wn * 4); (void)hipMalloc(&bias, bn * 4);
  float* h = (float*)malloc(wn * 4);
  for (size_t i = 0; i < wn; ++i) h[i] = (float)(rand() % 2000) / 8500.f - 0.117f;
  (void)hipMemcpy(wts, h, wn * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(bias, h, bn * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 1000) / 500.f;
  (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>(w, in, wts, bias, out);
    run<1, 0>(w, in, wts, bias, out);
    run<0, 32>(w, in, wts, bias, out);
    run<1, 32>(w, in, wts, bias, out);
    run<1, 96>(w, in, wts, bias, out);
  }
  return 0;
}
