// Micro-probe 12: how fast does a VALU stream run on a SIMD whose partner wave streams f16 MFMAs - and does it matter whether the MFMA
// accumulators live in VGPRs (the compiler's form on gfx950) or in AGPRs (inline asm, "+a")?  k_conv_x.hip's epilogue (fold + tensor product,
// ~50 VALU) takes ~1100 cycles beside the partner's burst; alone it would take ~200.  512-thread workgroups: waves 0-3 (one per SIMD) run an
// endless MFMA stream, waves 4-7 (their SIMD partners) time NV dependent-chain FMAs (8 independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// MODE 0: partner idle (barrier-parked), 1: MFMA stream with VGPR accumulators, 2: MFMA stream with AGPR accumulators (inline asm)
template <int MODE, int PRIO>
__global__ __launch_bounds__(512) void probe(const float* in, float* out, unsigned* cyc, int iters, volatile int* stop) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(lane + i) & 255]; b[i] = (_Float16)in[(lane * 3 + i) & 255]; }
  if (wave < 4) {
    if (MODE == 0) { out[blockIdx.x * 512 + tid] = 0.f; return; }
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    if (MODE == 1) {
      f32x16 D0, D1;
      for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; }
      for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
        for (int k = 0; k < 14; ++k) { D0 = MFMA16(a, b, D0); D1 = MFMA16(b, a, D1); }
      }
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += D0[r] + D1[r];
      out[blockIdx.x * 512 + tid] = s;
    } else {
      f32x16 D0, D1;
      for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; }
      for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
        for (int k = 0; k < 14; ++k) {
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(D0) : "v"(a), "v"(b));
          asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(D1) : "v"(b), "v"(a));
        }
      }
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += D0[r] + D1[r];
      out[blockIdx.x * 512 + tid] = s;
    }
  } else {
    float acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = in[(lane + k) & 255];
    const float m = in[lane & 255] * 0.5f;
    // let the partner get going
    for (int w = 0; w < 200; ++w) __builtin_amdgcn_s_sleep(10);
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[k & 7]) : "v"(m), "v"(acc[(k + 3) & 7]));
    }
    const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave - 4] = t1 - t0;
  }
}

template <int MODE, int PRIO>
void run(const char* name, float* in, float* out, unsigned* cyc) {
  const int iters = 400, grid = 256;
  hipLaunchKernelGGL((probe<MODE, PRIO>), dim3(grid), dim3(512), 0, 0, in, out, cyc, iters, (volatile int*)nullptr);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 1024; ++i) m += h[i];
  m /= 1024;
  printf("%-78s %6.2f ticks per v_fmac (64 per iteration, 8 chains)%s\n", name, m / (iters * 64.0), e == hipSuccess ? "" : " ** ERROR **");
}

int main() {
  float *in, *out; unsigned* cyc;
  (void)hipMalloc(&in, 1024 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 1024 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 17) * 0.01f + 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 0>("VALU stream, partner idle", in, out, cyc);
  run<1, 0>("VALU stream beside an MFMA stream, accumulators in VGPRs", in, out, cyc);
  run<1, 1>("VALU stream beside an MFMA stream, accumulators in VGPRs, MFMA wave at prio 1", in, out, cyc);
  run<2, 0>("VALU stream beside an MFMA stream, accumulators in AGPRs", in, out, cyc);
  run<2, 1>("VALU stream beside an MFMA stream, accumulators in AGPRs, MFMA wave at prio 1", in, out, cyc);
  return 0;
}
