// Micro-probe 5: weight-tile delivery for the GEMM2 loop of the fused conv kernel.
//   MODE 0 (private): every wave streams its own fragments from L2 (in-place reload), waves de-phased (different start
//                     tile / weight group per wave) as in the real kernel.
//   MODE 1 (shared):  NW-wave workgroup; each tile is fetched ONCE per workgroup into a 2-stage LDS ring (every thread
//                     moves its share), every wave reads its MFMA fragments from LDS into a second register set before
//                     the burst; one s_barrier per tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
constexpr int TILE_F = 9 * 256 + 32;   // floats per tile record: fragments [9][64][4] + bias [2][16]

template <int NV>
__device__ __forceinline__ void epilogue(const f32x16& D, const float (&h)[36], float (&acc)[8]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r & 7] = fmaf(D[r], h[r], acc[r & 7]);
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k & 7] = fmaf(acc[k & 7], h[k % 36], h[(k + 5) % 36]);
}

template <int NV>
__global__ __launch_bounds__(64) void probe_private(const float* in, const float* wts, float* out, int passes, int nt) {
  float h[36];
  const int lane = threadIdx.x, hh = lane >> 5;
  for (int i = 0; i < 36; ++i) h[i] = in[(lane * 3 + i) & 1023];
  const int grp = blockIdx.x & 3;
  const float* base = wts + (size_t)grp * nt * TILE_F;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int t0 = (blockIdx.x * 7) % nt;
  float4 a[9]; f32x16 B;
  for (int p = 0; p < passes; ++p) {
    for (int tt = 0; tt < nt; ++tt) {
      const int t = (t0 + tt) % nt, tn = (t + 1) % nt;
      const float* w = base + (size_t)t * TILE_F + lane * 4;
      const float* wn = base + (size_t)tn * TILE_F + lane * 4;
      if (tt == 0 && p == 0) {
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) a[s4] = ld4(w + s4 * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 v = ld4(base + (size_t)t * TILE_F + 2304 + hh * 16 + 4 * j); B[4 * j] = v.x; B[4 * j + 1] = v.y; B[4 * j + 2] = v.z; B[4 * j + 3] = v.w; }
      }
      f32x16 D = B;
#pragma unroll
      for (int s4 = 0; s4 < 9; ++s4) {
        D = MFMA(a[s4].x, h[4 * s4], D); D = MFMA(a[s4].y, h[4 * s4 + 1], D);
        D = MFMA(a[s4].z, h[4 * s4 + 2], D); D = MFMA(a[s4].w, h[4 * s4 + 3], D);
        a[s4] = ld4(wn + s4 * 256);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float4 v = ld4(base + (size_t)tn * TILE_F + 2304 + hh * 16 + 4 * j); B[4 * j] = v.x; B[4 * j + 1] = v.y; B[4 * j + 2] = v.z; B[4 * j + 3] = v.w; }
      epilogue<NV>(D, h, acc);
    }
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 64 + lane] = s;
}

template <int NV, int NW>
__global__ __launch_bounds__(64 * NW) void probe_shared(const float* in, const float* wts, float* out, int passes, int nt, long long* clk = nullptr) {
  const long long c0 = clock64(), w0c = wall_clock64();
  __shared__ __attribute__((aligned(16))) float ring[2][TILE_F];
  float h[36];
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
  for (int i = 0; i < 36; ++i) h[i] = in[(lane * 3 + i + tid) & 1023];
  const int grp = blockIdx.x & 3;
  const float* base = wts + (size_t)grp * nt * TILE_F;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int NT4 = TILE_F / 4;                       // 584 float4 per tile
  constexpr int PER = (NT4 + 64 * NW - 1) / (64 * NW);  // float4 per thread per tile
  float4 stg[PER];
  float4 a0[9], a1[9]; f32x16 B0, B1;
  const int total = passes * nt;
  // prologue: tile 0 -> ring[0] -> regs; tile 1 -> ring[1]
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int q = tid + i * 64 * NW; if (q < NT4) *reinterpret_cast<float4*>(&ring[k][4 * q]) = ld4(base + (size_t)k * TILE_F + 4 * q); }
  }
  __syncthreads();
#define LDS_FRAGS(A_, B_, st)                                                                \
  { const float* r_ = ring[st] + lane * 4;                                                   \
    _Pragma("unroll") for (int s4 = 0; s4 < 9; ++s4) A_[s4] = ld4(r_ + s4 * 256);            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { const float4 v = ld4(ring[st] + 2304 + hh * 16 + 4 * j); B_[4 * j] = v.x; B_[4 * j + 1] = v.y; B_[4 * j + 2] = v.z; B_[4 * j + 3] = v.w; } }
  LDS_FRAGS(a0, B0, 0)
#define TILE(I, AC, BC, AN, BN)                                                              \
  {                                                                                          \
    const int t2 = ((I) + 2) % nt;                                                           \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) { const int q = tid + i * 64 * NW; if (q < NT4) stg[i] = ld4(base + (size_t)t2 * TILE_F + 4 * q); } \
    LDS_FRAGS(AN, BN, ((I) + 1) & 1)                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    f32x16 D = MFMA(AC[0].x, h[0], BC);                                                      \
    D = MFMA(AC[0].y, h[1], D); D = MFMA(AC[0].z, h[2], D); D = MFMA(AC[0].w, h[3], D);      \
    _Pragma("unroll") for (int s4 = 1; s4 < 9; ++s4) {                                       \
      D = MFMA(AC[s4].x, h[4 * s4], D); D = MFMA(AC[s4].y, h[4 * s4 + 1], D);                \
      D = MFMA(AC[s4].z, h[4 * s4 + 2], D); D = MFMA(AC[s4].w, h[4 * s4 + 3], D); }          \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    epilogue<NV>(D, h, acc);                                                                 \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) { const int q = tid + i * 64 * NW; if (q < NT4) *reinterpret_cast<float4*>(&ring[(I) & 1][4 * q]) = stg[i]; } \
    __syncthreads();                                                                         \
  }
  for (int i = 0; i < total; i += 2) {
    TILE(i, a0, B0, a1, B1)
    TILE(i + 1, a1, B1, a0, B0)
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 64 * NW + tid] = s;
  if (clk && blockIdx.x == 7 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0c; }
}

int main() {
  const int nt = 66, passes = 40;
  float *in, *out, *wts;
  const size_t wn = (size_t)4 * nt * TILE_F;
  (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 8 * 64 * 4 * 2); (void)hipMalloc(&wts, wn * 4);
  float* h = (float*)malloc(wn * 4);
  for (size_t i = 0; i < wn; ++i) h[i] = (float)(rand() % 2000) / 8500.f - 0.117f;
  (void)hipMemcpy(wts, h, wn * 4, hipMemcpyHostToDevice);
  for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 1000) / 500.f;
  (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms;
#define RUN(name, waves, launch_warm, launch)                                                   \
  launch_warm; (void)hipDeviceSynchronize(); (void)hipEventRecord(a); launch; (void)hipEventRecord(b); (void)hipEventSynchronize(b); \
  (void)hipEventElapsedTime(&ms, a, b);                                                        \
  printf("%-34s waves/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, waves, ms, 256.0 * (waves) * passes * nt * 36 * 4096.0 / ms / 1e9);
  RUN("private NV=32", 8, hipLaunchKernelGGL((probe_private<32>), dim3(256 * 8), dim3(64), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_private<32>), dim3(256 * 8), dim3(64), 0, 0, in, wts, out, passes, nt))
  RUN("private NV=32", 4, hipLaunchKernelGGL((probe_private<32>), dim3(256 * 4), dim3(64), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_private<32>), dim3(256 * 4), dim3(64), 0, 0, in, wts, out, passes, nt))
  RUN("shared 4-wave WG NV=32", 4, hipLaunchKernelGGL((probe_shared<32, 4>), dim3(256), dim3(256), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_shared<32, 4>), dim3(256), dim3(256), 0, 0, in, wts, out, passes, nt))
  RUN("shared 4-wave WG NV=96", 4, hipLaunchKernelGGL((probe_shared<96, 4>), dim3(256), dim3(256), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_shared<96, 4>), dim3(256), dim3(256), 0, 0, in, wts, out, passes, nt))
  RUN("shared 8-wave WG NV=32", 8, hipLaunchKernelGGL((probe_shared<32, 8>), dim3(256), dim3(512), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_shared<32, 8>), dim3(256), dim3(512), 0, 0, in, wts, out, passes, nt))
  RUN("shared 8-wave WG NV=96", 8, hipLaunchKernelGGL((probe_shared<96, 8>), dim3(256), dim3(512), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_shared<96, 8>), dim3(256), dim3(512), 0, 0, in, wts, out, passes, nt))
  {
    long long* clk; (void)hipMalloc(&clk, 16); long long hc[2];
    hipLaunchKernelGGL((probe_shared<96, 8>), dim3(256), dim3(512), 0, 0, in, wts, out, passes, nt, clk);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("shared 8-wave NV=96: clock64 ticks %lld, wall_clock64 ticks %lld (100 MHz) -> %.3f GHz shader clock\n", hc[0], hc[1], (double)hc[0] / ((double)hc[1] / 100e6) / 1e9);
  }
  RUN("2x shared 4-wave WG / CU NV=32", 8, hipLaunchKernelGGL((probe_shared<32, 4>), dim3(512), dim3(256), 0, 0, in, wts, out, 2, nt), hipLaunchKernelGGL((probe_shared<32, 4>), dim3(512), dim3(256), 0, 0, in, wts, out, passes, nt))
  return 0;
}
