// Micro-probe 7: does the f16 matrix pipe overlap with the OTHER wave's VALU / LDS work on the same SIMD?  (k_conv_x.hip design question)
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a stream of v_mfma_f32_32x32x16_f16 on three accumulators, their SIMD
// partners (waves 4-7) run `N` instructions of one class per `M` partner-MFMAs worth of time, both in an endless loop for `iters` rounds;
// no barriers.  Reports cycles per MFMA of the matrix waves (s_memtime) for: partner idle / partner VALU fma / v_pk_fma / ds_read_b128 / both
// halves running MFMAs.  Also: the matrix wave alone with K VALU instructions interleaved in its own stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA8(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16((a), (b), (c), 0, 0, 0)
enum { P_IDLE, P_FMA, P_PKFMA, P_DSREAD, P_MFMA, P_MOV };

// ORDER 0: three chains round-robin (D2 D1 D2 D0 D2 D1 per step), 1: each accumulator's MFMAs back to back, 2: all on ONE accumulator
template <int PARTNER, int ORDER, int OWN_VALU, bool TAIL8, bool SWAP = false, int PRIO = 0>
__global__ __launch_bounds__(512) void probe(const float* in, float* out, unsigned* cyc, int iters) {
  __shared__ float lds[512 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 a[5], b[5];
  for (int s = 0; s < 5; ++s)
    for (int i = 0; i < 8; ++i) { a[s][i] = (_Float16)in[(lane * 7 + s * 8 + i) & 1023]; b[s][i] = (_Float16)in[(lane * 3 + s * 8 + i + 5) & 1023]; }
  f16x4 a4, b4;
  for (int i = 0; i < 4; ++i) { a4[i] = a[4][i]; b4[i] = b[4][i]; }
  for (int i = 0; i < 8; ++i) lds[tid * 8 + i] = in[(tid + i) & 1023];
  float acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = in[(lane + k) & 1023];
  float2 pacc[4];
  for (int k = 0; k < 4; ++k) pacc[k] = make_float2(acc[k], acc[k + 4]);
  __syncthreads();
  float sink = 0.f;
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  const bool matrix = SWAP ? wave >= 4 : wave < 4;      // SWAP: the matrix stream runs in the YOUNGER half (waves 4-7)
  if (matrix && PRIO) __builtin_amdgcn_s_setprio(PRIO);
  if (matrix || PARTNER == P_MFMA) {
    f32x16 D0, D1, D2;
    for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; D2[r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        if (TAIL8 && s == 4) {
          if (ORDER == 0) { D2 = MFMA8(a4, b4, D2); D1 = MFMA8(a4, b4, D1); D2 = MFMA8(a4, b4, D2); D0 = MFMA8(a4, b4, D0); D2 = MFMA8(a4, b4, D2); D1 = MFMA8(a4, b4, D1); }
          else { D0 = MFMA8(a4, b4, D0); D0 = MFMA8(a4, b4, D0); D0 = MFMA8(a4, b4, D0); D0 = MFMA8(a4, b4, D0); D0 = MFMA8(a4, b4, D0); D0 = MFMA8(a4, b4, D0); }
        } else if (ORDER == 0) {
          D2 = MFMA16(a[s], b[s], D2); D1 = MFMA16(a[s], b[s], D1); D2 = MFMA16(a[s], b[s], D2);
          D0 = MFMA16(a[s], b[s], D0); D2 = MFMA16(a[s], b[s], D2); D1 = MFMA16(a[s], b[s], D1);
        } else if (ORDER == 1) {
          D2 = MFMA16(a[s], b[s], D2); D2 = MFMA16(a[s], b[s], D2); D2 = MFMA16(a[s], b[s], D2);
          D1 = MFMA16(a[s], b[s], D1); D1 = MFMA16(a[s], b[s], D1); D0 = MFMA16(a[s], b[s], D0);
        } else {
          D0 = MFMA16(a[s], b[s], D0); D0 = MFMA16(a[s], b[s], D0); D0 = MFMA16(a[s], b[s], D0);
          D0 = MFMA16(a[s], b[s], D0); D0 = MFMA16(a[s], b[s], D0); D0 = MFMA16(a[s], b[s], D0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < OWN_VALU; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[k & 7]) : "v"(acc[(k + 3) & 7]), "v"(acc[(k + 5) & 7]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int r = 0; r < 16; ++r) sink += D0[r] + D1[r] + D2[r];
  } else {
    const int addr = tid * 32;
    for (int it = 0; it < iters * 8; ++it) {       // the partner runs far longer than the matrix waves: they see it for their whole run
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (PARTNER == P_FMA) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[k & 7]) : "v"(acc[(k + 3) & 7]), "v"(acc[(k + 5) & 7]));
        if (PARTNER == P_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[k & 7]) : "v"(acc[(k + 3) & 7]));
        if (PARTNER == P_PKFMA) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pacc[k & 3]) : "v"(pacc[(k + 1) & 3]));
        if (PARTNER == P_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(*(float4*)&acc[4 * (k & 1)]) : "v"(addr));
      }
      if (PARTNER == P_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
      if (PARTNER == P_IDLE) break;
      if (*(volatile int*)&lds[0] == 0x7fffffff) break;
    }
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int k = 0; k < 8; ++k) sink += acc[k];
  for (int k = 0; k < 4; ++k) sink += pacc[k].x + pacc[k].y;
  out[blockIdx.x * 512 + tid] = sink;
  if (lane == 0 && matrix) cyc[blockIdx.x * 4 + (wave & 3)] = t1 - t0;
}

template <int PARTNER, int ORDER, int OWN_VALU, bool TAIL8, bool SWAP = false, int PRIO = 0>
void run(const char* name, float* in, float* out, unsigned* cyc) {
  const int iters = 2000, grid = 256;
  hipLaunchKernelGGL((probe<PARTNER, ORDER, OWN_VALU, TAIL8, SWAP, PRIO>), dim3(grid), dim3(512), 0, 0, in, out, cyc, 50);
  (void)hipDeviceSynchronize();
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((probe<PARTNER, ORDER, OWN_VALU, TAIL8, SWAP, PRIO>), dim3(grid), dim3(512), 0, 0, in, out, cyc, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  unsigned h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 1024; ++i) m += h[i];
  m /= 1024;
  printf("%-58s %8.1f ticks per MFMA (matrix waves: %.0f ticks for %d MFMAs; kernel %.3f ms)\n", name, m / (iters * 30.0), m, iters * 30, ms);
}

int main() {
  float *in, *out; unsigned* cyc;
  (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 1024 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 17) * 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<P_IDLE, 0, 0, false>("partner idle, 3 chains round-robin", in, out, cyc);
  run<P_IDLE, 1, 0, false>("partner idle, chains back to back (3,2,1 per step)", in, out, cyc);
  run<P_IDLE, 2, 0, false>("partner idle, ONE accumulator", in, out, cyc);
  run<P_IDLE, 0, 0, true>("partner idle, round-robin, 32x32x8 tail step", in, out, cyc);
  run<P_FMA, 0, 0, false>("partner v_fmac_f32 stream", in, out, cyc);
  run<P_MOV, 0, 0, false>("partner v_mov_b32 stream", in, out, cyc);
  run<P_PKFMA, 0, 0, false>("partner v_pk_fma_f32 stream", in, out, cyc);
  run<P_DSREAD, 0, 0, false>("partner ds_read_b128 stream", in, out, cyc);
  run<P_MFMA, 0, 0, false>("partner runs the same MFMA stream", in, out, cyc);
  run<P_IDLE, 0, 2, false>("partner idle, 2 own v_fmac per step (6 MFMAs)", in, out, cyc);
  run<P_IDLE, 0, 6, false>("partner idle, 6 own v_fmac per step", in, out, cyc);
  run<P_IDLE, 0, 12, false>("partner idle, 12 own v_fmac per step", in, out, cyc);
  run<P_IDLE, 0, 24, false>("partner idle, 24 own v_fmac per step", in, out, cyc);
  run<P_FMA, 0, 6, false>("partner v_fmac stream + 6 own v_fmac per step", in, out, cyc);
  run<P_FMA, 0, 0, false, true>("matrix stream in the YOUNGER half, partner v_fmac stream", in, out, cyc);
  run<P_DSREAD, 0, 0, false, true>("matrix stream in the YOUNGER half, partner ds_read stream", in, out, cyc);
  run<P_FMA, 0, 0, false, true, 1>("younger matrix half at s_setprio 1, partner v_fmac stream", in, out, cyc);
  run<P_FMA, 0, 0, false, true, 3>("younger matrix half at s_setprio 3, partner v_fmac stream", in, out, cyc);
  return 0;
}
