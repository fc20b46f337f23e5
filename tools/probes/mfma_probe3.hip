// Micro-probe 3: which instruction classes steal issue time from the fp32 MFMA pipe (2 waves/SIMD)?
// Each iteration: 36 dependent v_mfma_f32_32x32x2_f32 + N instructions of one class (independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
enum { K_NONE, K_FMA, K_PKFMA, K_MOV, K_CNDMASK, K_IADD, K_DSREAD, K_BPERM, K_SALU, K_MUL, K_DPP, K_ACCRD };

template <int KIND, int N>
__global__ __launch_bounds__(64) void probe(const float* in, float* out, int iters) {
  __shared__ float lds[64 * 36];
  float b[36], a[36];
  const int lane = threadIdx.x;
  for (int i = 0; i < 36; ++i) { b[i] = in[(lane * 3 + i) & 1023]; a[i] = in[(lane + i) & 1023]; lds[lane * 36 + i] = b[i]; }
  __syncthreads();
  float acc[8]; int iacc[8]; f32x2 pacc[8];
  for (int k = 0; k < 8; ++k) { acc[k] = (float)k; iacc[k] = k + lane; pacc[k] = f32x2{(float)k, 1.f}; }
  float sink = 0.f;
  int ssink = 0;
  const int addr = lane * 144;
  for (int it = 0; it < iters; ++it) {
    f32x16 D;
    for (int r = 0; r < 16; ++r) D[r] = b[r] * (float)it;
#pragma unroll
    for (int s = 0; s < 36; ++s) D = MFMA(a[s], b[s], D);
#pragma unroll
    for (int r = 0; r < 16; ++r) sink = fmaf(D[r], b[r], sink);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int j = k & 7;
      if (KIND == K_FMA) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "v"(b[k % 36]), "v"(b[(k + 5) % 36]));
      if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[j]) : "v"(b[k % 36]));
      if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pacc[j]) : "v"(pacc[(j + 3) & 7]));
      if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[j]) : "v"(b[k % 36]));
      if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[j]) : "v"(b[k % 36]));
      if (KIND == K_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iacc[j]) : "v"(iacc[(j + 3) & 7]));
      if (KIND == K_DSREAD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(float4*)&pacc[2 * (k & 3)]) : "v"(addr), "n"(0));
      if (KIND == K_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(acc[j]) : "v"(addr));
      if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ssink));
      if (KIND == K_DPP) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]));
    }
    if (KIND == K_DSREAD || KIND == K_BPERM) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  for (int k = 0; k < 8; ++k) sink += acc[k] + (float)iacc[k] + pacc[k].x + pacc[k].y;
  out[blockIdx.x * 64 + lane] = sink + (float)ssink;
}

template <int KIND, int N>
void run(const char* name, float* in, float* out) {
  const int iters = 4000, grid = 256 * 4 * 2;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((probe<KIND, N>), dim3(grid), dim3(64), 0, 0, in, out, 100);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((probe<KIND, N>), dim3(grid), dim3(64), 0, 0, in, out, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  static float base = 0.f;
  if (KIND == K_NONE) base = ms;
  // cycles each extra instruction costs the SIMD (2 waves/SIMD -> 2*iters*N instructions per SIMD)
  const double cyc = N ? (ms - base) * 1e-3 * 2.4e9 / (2.0 * iters * N) : 0.0;
  printf("%-12s N=%3d  %.3f ms  %.1f TFLOP/s   +%.2f cycles/instr (at 2.4 GHz)\n", name, N, ms, (double)grid * iters * 36 * 4096.0 / ms / 1e9, cyc);
}

int main() {
  float *in, *out;
  (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 1000) / 500.f - 1.f;
  (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  run<K_NONE, 0>("none", in, out);
  run<K_FMA, 256>("v_fmac_f32", in, out);
  run<K_MUL, 256>("v_mul_f32", in, out);
  run<K_PKFMA, 256>("v_pk_fma_f32", in, out);
  run<K_MOV, 256>("v_mov_b32", in, out);
  run<K_CNDMASK, 256>("v_cndmask", in, out);
  run<K_IADD, 256>("v_add_u32", in, out);
  run<K_DPP, 256>("v_add_f32_dpp", in, out);
  run<K_DSREAD, 64>("ds_read_b128", in, out);
  run<K_BPERM, 64>("ds_bpermute", in, out);
  run<K_SALU, 256>("s_add_u32", in, out);
  return 0;
}
