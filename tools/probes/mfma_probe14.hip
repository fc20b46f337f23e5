// Micro-probe 14 (round 5): what does ONE instruction cost when it is hand-placed (inline asm, no compiler scheduling) in the shadow of an f16 MFMA
// of the SAME wave - and how well do two waves of one SIMD share the matrix pipe when both stream MFMAs?
//   part A: one wave per SIMD (256-thread workgroups, one per CU), NACC independent accumulators round robin, NF fillers of one kind behind every
//           v_mfma_f32_32x32x16_f16: v_fmac (2 VGPR sources), v_fma (3 VGPR sources), v_add, ds_read_b128, a mix shaped like k_conv_x.hip's epilogue.
//           Reported: cycles per MFMA (32 = the pipe's floor).
//   part B: two waves per SIMD (512 threads), both streaming MFMAs (2 accumulators each, as k_conv_x.hip's burst), with / without the burst's LDS
//           fragment reads, same / different priority: cycles per MFMA of the SIMD (32 = floor) - round 4 read "77 %" for two simultaneous bursts.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MF(D, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(D) : "v"(a), "v"(b))
#define FMAC(d, x, y) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(x), "v"(y))
#define FMA3(d, x, y, z) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z))
#define ADD(d, x, y) asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define DSR(v, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(v) : "v"(addr))

// KIND 0 none, 1 v_fmac, 2 v_fma (3 sources), 3 v_add, 4 one ds_read_b128 per MFMA (NF ignored), 5 mix: per 4 MFMAs 1 ds_read_b128 + 4 v_add + 4 v_fmac + 1 v_fma
template <int NACC, int KIND, int NF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probeA(const float* in, float* out, unsigned* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 a[3], b[3];
  for (int l = 0; l < 3; ++l)
    for (int i = 0; i < 8; ++i) { a[l][i] = (_Float16)in[(lane * 3 + i + l) & 1023]; b[l][i] = (_Float16)in[(lane * 5 + i + 2 * l) & 1023]; }
  for (int i = tid; i < 8192; i += 256) lds[i] = in[i & 1023];
  f32x16 D[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) D[k][r] = 0.f;
  float acc[16], src[8];
  for (int k = 0; k < 16; ++k) acc[k] = in[(lane + k) & 1023];
  for (int k = 0; k < 8; ++k) src[k] = in[(lane + 3 * k + 1) & 1023] * 0.01f;
  f32x4 fr[4];
  const unsigned la = (unsigned)(lane * 16);
  __syncthreads();
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 56; ++m) {
      MF(D[m % NACC], a[m % 3], b[(m / 3) % 3]);
      if (KIND == 1) {
#pragma unroll
        for (int f = 0; f < NF; ++f) FMAC(acc[(m * NF + f) & 15], src[(m + f) & 7], src[(m + f + 3) & 7]);
      } else if (KIND == 2) {
#pragma unroll
        for (int f = 0; f < NF; ++f) FMA3(acc[(m * NF + f) & 15], src[(m + f) & 7], src[(m + f + 3) & 7], acc[(m * NF + f + 8) & 15]);
      } else if (KIND == 3) {
#pragma unroll
        for (int f = 0; f < NF; ++f) ADD(acc[(m * NF + f) & 15], src[(m + f) & 7], acc[(m * NF + f + 8) & 15]);
      } else if (KIND == 4) {
        DSR(fr[m & 3], la, 0);
        if ((m & 3) == 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      } else if (KIND == 5) {
        if ((m & 3) == 0) DSR(fr[(m >> 2) & 3], la, 1024);
        ADD(acc[m & 15], src[m & 7], acc[(m + 8) & 15]);
        FMAC(acc[(m + 4) & 15], src[(m + 1) & 7], src[(m + 5) & 7]);
        if ((m & 3) == 1) FMA3(acc[(m + 6) & 15], src[m & 7], src[(m + 2) & 7], acc[(m + 12) & 15]);
        if ((m & 3) == 3) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += D[k][r];
  for (int k = 0; k < 16; ++k) s += acc[k];
  if (KIND >= 4) for (int k = 0; k < 4; ++k) s += fr[k].x;
  out[blockIdx.x * 256 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// two waves per SIMD, both streaming: LDSR = the burst's fragment reads (15 ds_read_b128 per 28 MFMAs), PRIO = waves 4-7 at s_setprio 1, NACC accumulators per wave
template <int NACC, bool LDSR, int PRIO, bool ALONE>
__global__ __launch_bounds__(512) void probeB(const float* in, float* out, unsigned* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 a[3], b[3];
  for (int l = 0; l < 3; ++l)
    for (int i = 0; i < 8; ++i) { a[l][i] = (_Float16)in[(lane * 3 + i + l) & 1023]; b[l][i] = (_Float16)in[(lane * 5 + i + 2 * l) & 1023]; }
  for (int i = tid; i < 8192; i += 512) lds[i] = in[i & 1023];
  f32x16 D[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) D[k][r] = 0.f;
  f32x4 fr[4];
  const unsigned la = (unsigned)(lane * 16 + (wave & 3) * 1024);
  __syncthreads();
  if (ALONE && wave >= 4) { out[blockIdx.x * 512 + tid] = 0.f; return; }
  if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 28; ++m) {
      MF(D[m % NACC], a[m % 3], b[(m / 3) % 3]);
      if (LDSR && m < 15) {
        DSR(fr[m & 3], la, 2048);
        if ((m & 3) == 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += D[k][r];
  if (LDSR) for (int k = 0; k < 4; ++k) s += fr[k].x;
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_in; static float* g_out; static unsigned* g_cyc;

template <int NACC, int KIND, int NF>
void runA(const char* name) {
  const int iters = 300, grid = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probeA<NACC, KIND, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((probeA<NACC, KIND, NF>), dim3(grid), dim3(256), 160 * 1024, 0, g_in, g_out, g_cyc, 10);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((probeA<NACC, KIND, NF>), dim3(grid), dim3(256), 160 * 1024, 0, g_in, g_out, g_cyc, iters);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[1024];
  (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 1024; ++i) m += h[i];
  m /= 1024;
  printf("A %-86s %6.2f cycles per MFMA%s\n", name, m / (iters * 56.0), e == hipSuccess ? "" : " ** ERROR **");
}

template <int NACC, bool LDSR, int PRIO, bool ALONE>
void runB(const char* name) {
  const int iters = 300, grid = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probeB<NACC, LDSR, PRIO, ALONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((probeB<NACC, LDSR, PRIO, ALONE>), dim3(grid), dim3(512), 160 * 1024, 0, g_in, g_out, g_cyc, 10);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((probeB<NACC, LDSR, PRIO, ALONE>), dim3(grid), dim3(512), 160 * 1024, 0, g_in, g_out, g_cyc, iters);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[2048];
  (void)hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; int n = 0;
  for (int i = 0; i < 2048; ++i) if (!ALONE || (i & 7) < 4) { m += h[i]; ++n; }
  m /= n;
  // per SIMD: two waves x 28 MFMAs per iteration (one wave when ALONE)
  printf("B %-86s %6.2f cycles per MFMA of the SIMD%s\n", name, m / (iters * 28.0 * (ALONE ? 1 : 2)), e == hipSuccess ? "" : " ** ERROR **");
}

int main() {
  (void)hipMalloc(&g_in, 1024 * 4); (void)hipMalloc(&g_out, 256 * 512 * 4); (void)hipMalloc(&g_cyc, 2048 * 4);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 37) % 17) * 0.01f + 0.01f;
  (void)hipMemcpy(g_in, h, sizeof(h), hipMemcpyHostToDevice);
  runA<4, 0, 0>("lone wave, 4 accumulators, no filler");
  runA<2, 0, 0>("lone wave, 2 accumulators, no filler");
  runA<1, 0, 0>("lone wave, 1 accumulator, no filler");
  runA<4, 1, 1>("lone wave, 4 acc, 1 v_fmac per MFMA");
  runA<4, 1, 2>("lone wave, 4 acc, 2 v_fmac per MFMA");
  runA<4, 1, 3>("lone wave, 4 acc, 3 v_fmac per MFMA");
  runA<4, 1, 4>("lone wave, 4 acc, 4 v_fmac per MFMA");
  runA<4, 1, 5>("lone wave, 4 acc, 5 v_fmac per MFMA");
  runA<4, 1, 6>("lone wave, 4 acc, 6 v_fmac per MFMA");
  runA<4, 2, 2>("lone wave, 4 acc, 2 v_fma (3 VGPR sources) per MFMA");
  runA<4, 2, 4>("lone wave, 4 acc, 4 v_fma (3 VGPR sources) per MFMA");
  runA<4, 3, 2>("lone wave, 4 acc, 2 v_add per MFMA");
  runA<4, 3, 4>("lone wave, 4 acc, 4 v_add per MFMA");
  runA<4, 4, 0>("lone wave, 4 acc, 1 ds_read_b128 per MFMA");
  runA<4, 5, 0>("lone wave, 4 acc, mix (per 4 MFMAs: 1 ds_read_b128 + 4 v_add + 4 v_fmac + 1 v_fma = 2.5 per MFMA)");
  runA<2, 1, 2>("lone wave, 2 acc, 2 v_fmac per MFMA");
  runA<2, 1, 4>("lone wave, 2 acc, 4 v_fmac per MFMA");
  runA<2, 5, 0>("lone wave, 2 acc, mix (2.5 per MFMA)");
  runA<1, 1, 2>("lone wave, 1 acc, 2 v_fmac per MFMA");
  runB<2, false, 0, true>("one wave per SIMD streaming (partner exited), 2 acc");
  runB<2, true, 0, true>("one wave per SIMD streaming + fragment reads, 2 acc");
  runB<2, false, 0, false>("two waves per SIMD both streaming, 2 acc each, no LDS reads, equal priority");
  runB<2, true, 0, false>("two waves per SIMD both streaming, 2 acc each, + fragment reads, equal priority");
  runB<2, true, 1, false>("two waves per SIMD both streaming, 2 acc each, + fragment reads, waves 4-7 at prio 1");
  runB<2, true, 2, false>("two waves per SIMD both streaming, 2 acc each, + fragment reads, all at prio 1");
  runB<4, true, 0, false>("two waves per SIMD both streaming, 4 acc each, + fragment reads, equal priority");
  runB<1, false, 0, false>("two waves per SIMD both streaming, 1 acc each, no LDS reads");
  return 0;
}
