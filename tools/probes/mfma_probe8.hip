// Micro-probe 8: the strict-alternation skeleton of k_conv_x.hip.  512-thread workgroups, one per CU; waves 0-3 (group A) and their SIMD
// partners 4-7 (group B) alternate: one group runs a burst of 30 f16 MFMAs (three accumulators) while the other runs an "epilogue" of NV VALU
// instructions (+ NL ds_read_b128, + NW ds_write_b128), two barriers per tile.  Features are switched on one by one to find what stretches the
// burst beyond 30 x 32 cycles: LDS fragment reads inside the burst (FRAG), accumulators re-zeroed per tile (ZERO), the epilogue READING the
// accumulators (READD: the burst's results are consumed by VALU right behind the barrier), global loads in flight (GL).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NV, bool FRAG, bool ZERO, bool READD, bool GL, bool BOTH>
__global__ __launch_bounds__(512) void probe(const float* in, float* out, unsigned* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2;
  f16x8 a[5], b[5];
  for (int s = 0; s < 5; ++s)
    for (int i = 0; i < 8; ++i) { a[s][i] = (_Float16)in[(lane * 7 + s * 8 + i) & 1023]; b[s][i] = (_Float16)in[(lane * 3 + s * 8 + i + 5) & 1023]; }
  for (int i = tid; i < 8192; i += 512) lds[i] = in[i & 1023];
  float acc[16];
  for (int k = 0; k < 16; ++k) acc[k] = in[(lane + k) & 1023];
  __syncthreads();
  f32x16 D0, D1, D2;
  for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; D2[r] = 0.f; }
  const f16x8* frag = reinterpret_cast<const f16x8*>(lds) + lane;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  if (grp && !BOTH) lds_barrier();
  for (int t = 0; t < tiles; ++t) {
    // ---- burst ----
    if (GL) g = *reinterpret_cast<const float4*>(in + ((t * 64 + tid) & 1020));
    if (ZERO) for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; D2[r] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      f16x8 ah = a[s], am = a[(s + 1) % 5], al = a[(s + 2) % 5];
      if (FRAG && s >= 2) { ah = frag[64 * (3 * s)]; am = frag[64 * (3 * s + 1)]; al = frag[64 * (3 * s + 2)]; }
      D2 = MFMA16(ah, b[s], D2); D1 = MFMA16(ah, b[(s + 1) % 5], D1); D2 = MFMA16(al, b[s], D2);
      D0 = MFMA16(ah, b[(s + 2) % 5], D0); D2 = MFMA16(am, b[(s + 3) % 5], D2); D1 = MFMA16(am, b[s], D1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!BOTH) lds_barrier();
    // ---- epilogue ----
    if (READD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaf(fmaf(D2[r], 0.00048828125f, D1[r]), 0.00048828125f, fmaf(D0[r], acc[(r + 1) & 15], acc[r]));
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[k & 15]) : "v"(acc[(k + 3) & 15]), "v"(acc[(k + 5) & 15]));
    if (GL) acc[0] += g.x + g.y + g.z + g.w;
    lds_barrier();
  }
  if (!grp && !BOTH) lds_barrier();
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += D0[r] + D1[r] + D2[r] + acc[r];
  out[blockIdx.x * 512 + tid] = sink;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NV, bool FRAG, bool ZERO, bool READD, bool GL, bool BOTH = false>
void run(const char* name, float* in, float* out, unsigned* cyc) {
  const int tiles = 2000, grid = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<NV, FRAG, ZERO, READD, GL, BOTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((probe<NV, FRAG, ZERO, READD, GL, BOTH>), dim3(grid), dim3(512), 160 * 1024, 0, in, out, cyc, 20);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<NV, FRAG, ZERO, READD, GL, BOTH>), dim3(grid), dim3(512), 160 * 1024, 0, in, out, cyc, tiles);
  (void)hipDeviceSynchronize();
  unsigned h[2048];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 2048; ++i) m += h[i];
  m /= 2048;
  printf("%-86s period %7.0f ticks per tile (2 x 30 MFMAs per SIMD = 1920)\n", name, m / tiles);
}

int main() {
  float *in, *out; unsigned* cyc;
  (void)hipMalloc(&in, 8192 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 4);
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 17) * 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, false, false, false, false, true>("both groups burst together, no barrier between burst and epilogue, 0 VALU", in, out, cyc);
  run<0, false, false, false, false>("alternation, bursts only (0 VALU)", in, out, cyc);
  run<100, false, false, false, false>("alternation, 100 VALU epilogue", in, out, cyc);
  run<200, false, false, false, false>("alternation, 200 VALU epilogue", in, out, cyc);
  run<100, true, false, false, false>("alternation, 100 VALU + fragments of K steps 2-4 from LDS", in, out, cyc);
  run<100, false, true, false, false>("alternation, 100 VALU + accumulators zeroed per tile", in, out, cyc);
  run<100, false, true, true, false>("alternation, 100 VALU + zeroed + epilogue reads the accumulators", in, out, cyc);
  run<100, true, true, true, false>("alternation, 100 VALU + fragments + zeroed + reads", in, out, cyc);
  run<100, true, true, true, true>("alternation, 100 VALU + fragments + zeroed + reads + a global load per tile", in, out, cyc);
  return 0;
}
