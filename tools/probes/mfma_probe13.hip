// Micro-probe 13 (result: NO - the assembler rejects the symbolic name for gfx950 and hwreg(29) reads 0 on the hardware; s_memtime stays the
// only time stamp, and it is an SMEM operation).  Question: is HW_REG_SHADER_CYCLES (s_getreg_b32, a 20-bit cycle counter read by the SALU: no memory operation, no lgkmcnt) usable as a
// non-intrusive time stamp on gfx950?  Compares it with s_memtime over a known stretch of dependent FMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, float* sink, int n) {
  unsigned a, b;
  const unsigned long long m0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_getreg_b32 %0, hwreg(29, 0, 20)" : "=s"(a));
  float x = sink[threadIdx.x];
  for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
  asm volatile("s_getreg_b32 %0, hwreg(29, 0, 20)" : "=s"(b) : "v"(x));
  const unsigned long long m1 = __builtin_amdgcn_s_memtime();
  sink[threadIdx.x] = x;
  if (threadIdx.x == 0) { out[0] = (b - a) & 0xfffff; out[1] = (unsigned)(m1 - m0); }
}
int main() {
  unsigned* out; float* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 1024); (void)hipMemset(sink, 0, 1024);
  for (int n : {100, 1000, 10000}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, n);
    unsigned h[2];
    (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("n = %5d dependent FMAs: SHADER_CYCLES delta %7u, s_memtime delta %7u\n", n, h[0], h[1]);
  }
  return 0;
}
