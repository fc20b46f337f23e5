// Micro-probe 10: the "free-running, software-pipelined" tile loop that round 4 weighs against k_conv_x.hip's strict alternation.
// 512-thread workgroups (two waves per SIMD), every wave runs the SAME stream: the 28 f16 MFMAs of tile t (prescaled limbs: all six limb
// products carry their own weight, two alternating fp32 accumulators) with the fold + tensor-product FMAs of tile t-1 (NV VALU reading the
// previous tile's result) and the LDS fragment reads threaded between them by sched_group_barrier; ONE barrier per tile (2-stage ring: every
// wave writes its share of tile t+1 into the other stage during burst t).  Questions: (1) does the period stay near the matrix pipe's
// 2 x 28 x 32 = 1792 cycles when each wave carries NV = 32..128 VALU per tile in its MFMA shadows; (2) what do the barrier, the ring traffic
// (2 global loads + 2 ds_write_b128 per thread and tile) and the exposed first fragment read cost; (3) the same with four accumulator sets
// (no end-of-tile fold stall).  Also: do fp16 SUBNORMAL inputs survive v_mfma_f32_32x32x16_f16 (prescaled low limbs rely on them)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA8(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16((a), (b), (c), 0, 0, 0)

constexpr int TILE_BYTES = 13968, LIMB_BYTES = 4608, FS = 132;

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct Frag16 { f16x8 h, m, l; };
__device__ __forceinline__ Frag16 lds_frag16(const char* stage, int s, int lane) {
  Frag16 f;
  f.h = *reinterpret_cast<const f16x8*>(stage + s * 1024 + lane * 16);
  f.m = *reinterpret_cast<const f16x8*>(stage + LIMB_BYTES + s * 1024 + lane * 16);
  f.l = *reinterpret_cast<const f16x8*>(stage + 2 * LIMB_BYTES + s * 1024 + lane * 16);
  return f;
}

// six limb products of one K step, alternating accumulators
#define STEP6(f, s)                                  \
  D0 = MFMA16(f.h, bl[s], D0);                       \
  D1 = MFMA16(f.h, bm[s], D1);                       \
  D0 = MFMA16(f.l, bh[s], D0);                       \
  D1 = MFMA16(f.h, bh[s], D1);                       \
  D0 = MFMA16(f.m, bm[s], D0);                       \
  D1 = MFMA16(f.m, bh[s], D1);

// NV: VALU per tile reading the previous tile's result; RING: ring traffic; BAR: barrier per tile; VPM: VALU slots per MFMA in the pinned order
template <int NV, bool RING, bool BAR, int VPM>
__global__ __launch_bounds__(512) void probe(const float* in, const float4* w, float* out, unsigned* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c1 = min(tid + 512, 872);
  char* ring = reinterpret_cast<char*>(lds);
  float* F = lds + (2 * TILE_BYTES + 16) / 4 + wave * (32 * FS) + (lane & 31) * FS;
  f16x8 bh[4], bm[4], bl[4], tmh, thl;
  f16x4 th, tm;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      bh[s][i] = (_Float16)in[(lane * 7 + s * 8 + i) & 1023];
      bm[s][i] = (_Float16)(in[(lane * 3 + s * 8 + i + 5) & 1023] * 0.001f);
      bl[s][i] = (_Float16)(in[(lane * 5 + s * 8 + i + 9) & 1023] * 0.000001f);
    }
  for (int i = 0; i < 8; ++i) { tmh[i] = (_Float16)in[(lane + i) & 1023]; thl[i] = (_Float16)in[(lane + i + 11) & 1023]; }
  for (int i = 0; i < 4; ++i) { th[i] = (_Float16)in[(lane + i + 3) & 1023]; tm[i] = (_Float16)(in[(lane + i + 17) & 1023] * 0.001f); }
  for (int i = tid; i < 40000; i += 512) lds[i] = in[i & 1023] * 0.01f;
  float acc[16];
  for (int k = 0; k < 16; ++k) acc[k] = in[(lane + k) & 1023];
  f32x16 Dp;
  for (int r = 0; r < 16; ++r) Dp[r] = in[(lane + r + 7) & 1023];
  __syncthreads();
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    const char* stage = ring + (t & 1) * TILE_BYTES;
    char* other = ring + ((t + 1) & 1) * TILE_BYTES;
    f32x16 D0, D1;
    for (int r = 0; r < 16; ++r) { D0[r] = 0.f; D1[r] = 0.f; }
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    const float* Fp = F + ((t * 4) & 127);
    __builtin_amdgcn_sched_barrier(0);
    if (RING) {
      g0 = w[((t & 31) * 1024 + tid)];
      g1 = w[((t & 31) * 1024 + c1)];
    }
    const Frag16 f0 = lds_frag16(stage, 0, lane);
    const f32x4 fv = *reinterpret_cast<const f32x4*>(Fp);
    const Frag16 f1 = lds_frag16(stage, 1, lane);
    const Frag16 f2 = lds_frag16(stage, 2, lane);
    const Frag16 f3 = lds_frag16(stage, 3, lane);
    const f16x4 ath = *reinterpret_cast<const f16x4*>(stage + 4096 + lane * 8);
    const f16x4 atm = *reinterpret_cast<const f16x4*>(stage + LIMB_BYTES + 4096 + lane * 8);
    const f16x4 atl = *reinterpret_cast<const f16x4*>(stage + 2 * LIMB_BYTES + 4096 + lane * 8);
    STEP6(f0, 0) STEP6(f1, 1) STEP6(f2, 2) STEP6(f3, 3)
    {
      const f16x8 a_hm = __builtin_shufflevector(ath, atm, 0, 1, 2, 3, 4, 5, 6, 7), a_lh = __builtin_shufflevector(atl, ath, 0, 1, 2, 3, 4, 5, 6, 7);
      D0 = MFMA16(a_lh, thl, D0);
      D1 = MFMA16(a_hm, tmh, D1);
      D0 = MFMA8(ath, th, D0);
      D1 = MFMA8(atm, tm, D1);
    }
    // the previous tile's tensor-product FMAs
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k & 15] = fmaf(Dp[(k * 5) & 15], fv[k & 3], acc[k & 15]);
    if (RING) {
      *reinterpret_cast<float4*>(other + 16 * tid) = g0;
      *reinterpret_cast<float4*>(other + 16 * c1) = g1;      // threads past the record's end rewrite its last chunk with the same bytes
    }
    // pinned order: three fragment reads ahead, then per MFMA one LDS read (while there are any) and VPM VALU
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int k = 0; k < 28; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (k < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      if (k == 20 || k == 22) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) Dp[r] = D0[r] + D1[r];
    if (BAR) lds_barrier();
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += Dp[r] + acc[r];
  out[blockIdx.x * 512 + tid] = sink;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// four accumulator sets: even tiles accumulate into (E0, E1) while the VALU consumes the odd tile's (O0 += O1, then O0), and vice versa -
// no fold between the last MFMA of a tile and the barrier
#define STEP6X(f, s, A0, A1)                         \
  A0 = MFMA16(f.h, bl[s], A0);                       \
  A1 = MFMA16(f.h, bm[s], A1);                       \
  A0 = MFMA16(f.l, bh[s], A0);                       \
  A1 = MFMA16(f.h, bh[s], A1);                       \
  A0 = MFMA16(f.m, bm[s], A0);                       \
  A1 = MFMA16(f.m, bh[s], A1);

template <int NV, bool RING, bool BAR, int VPM>
__global__ __launch_bounds__(512) void probe4(const float* in, const float4* w, float* out, unsigned* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c1 = min(tid + 512, 872);
  char* ring = reinterpret_cast<char*>(lds);
  float* F = lds + (2 * TILE_BYTES + 16) / 4 + wave * (32 * FS) + (lane & 31) * FS;
  f16x8 bh[4], bm[4], bl[4], tmh, thl;
  f16x4 th, tm;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      bh[s][i] = (_Float16)in[(lane * 7 + s * 8 + i) & 1023];
      bm[s][i] = (_Float16)(in[(lane * 3 + s * 8 + i + 5) & 1023] * 0.001f);
      bl[s][i] = (_Float16)(in[(lane * 5 + s * 8 + i + 9) & 1023] * 0.000001f);
    }
  for (int i = 0; i < 8; ++i) { tmh[i] = (_Float16)in[(lane + i) & 1023]; thl[i] = (_Float16)in[(lane + i + 11) & 1023]; }
  for (int i = 0; i < 4; ++i) { th[i] = (_Float16)in[(lane + i + 3) & 1023]; tm[i] = (_Float16)(in[(lane + i + 17) & 1023] * 0.001f); }
  for (int i = tid; i < 40000; i += 512) lds[i] = in[i & 1023] * 0.01f;
  float acc[16];
  for (int k = 0; k < 16; ++k) acc[k] = in[(lane + k) & 1023];
  f32x16 E0, E1, O0, O1;
  for (int r = 0; r < 16; ++r) { O0[r] = in[(lane + r + 7) & 1023]; O1[r] = 0.f; E0[r] = 0.f; E1[r] = 0.f; }
  __syncthreads();
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
#define HALF(t, A0, A1, P0, P1)                                                                                                     \
  {                                                                                                                                 \
    const char* stage = ring + ((t) & 1) * TILE_BYTES;                                                                              \
    char* other = ring + (((t) + 1) & 1) * TILE_BYTES;                                                                              \
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;                                                                           \
    const float* Fp = F + (((t) * 4) & 127);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    if (RING) {                                                                                                                     \
      g0 = w[(((t) & 31) * 1024 + tid)];                                                                                            \
      g1 = w[(((t) & 31) * 1024 + c1)];                                                                                             \
    }                                                                                                                               \
    const Frag16 f0 = lds_frag16(stage, 0, lane);                                                                                   \
    const f32x4 fv = *reinterpret_cast<const f32x4*>(Fp);                                                                           \
    const Frag16 f1 = lds_frag16(stage, 1, lane);                                                                                   \
    const Frag16 f2 = lds_frag16(stage, 2, lane);                                                                                   \
    const Frag16 f3 = lds_frag16(stage, 3, lane);                                                                                   \
    const f16x4 ath = *reinterpret_cast<const f16x4*>(stage + 4096 + lane * 8);                                                     \
    const f16x4 atm = *reinterpret_cast<const f16x4*>(stage + LIMB_BYTES + 4096 + lane * 8);                                        \
    const f16x4 atl = *reinterpret_cast<const f16x4*>(stage + 2 * LIMB_BYTES + 4096 + lane * 8);                                    \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) P0[r] += P1[r];                                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) { A0[r] = 0.f; A1[r] = 0.f; }                                                    \
    STEP6X(f0, 0, A0, A1) STEP6X(f1, 1, A0, A1) STEP6X(f2, 2, A0, A1) STEP6X(f3, 3, A0, A1)                                         \
    {                                                                                                                               \
      const f16x8 a_hm = __builtin_shufflevector(ath, atm, 0, 1, 2, 3, 4, 5, 6, 7), a_lh = __builtin_shufflevector(atl, ath, 0, 1, 2, 3, 4, 5, 6, 7); \
      A0 = MFMA16(a_lh, thl, A0);                                                                                                   \
      A1 = MFMA16(a_hm, tmh, A1);                                                                                                   \
      A0 = MFMA8(ath, th, A0);                                                                                                      \
      A1 = MFMA8(atm, tm, A1);                                                                                                      \
    }                                                                                                                               \
    _Pragma("unroll") for (int k = 0; k < NV; ++k) acc[k & 15] = fmaf(P0[(k * 5) & 15], fv[k & 3], acc[k & 15]);                    \
    if (RING) {                                                                                                                     \
      *reinterpret_cast<float4*>(other + 16 * tid) = g0;                                                                            \
      *reinterpret_cast<float4*>(other + 16 * c1) = g1;                                                                             \
    }                                                                                                                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                                              \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                                              \
    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);                                                                             \
    _Pragma("unroll") for (int k = 0; k < 28; ++k) {                                                                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                            \
      if (k < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                \
      if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                                                             \
      if (k == 20 || k == 22) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                    \
    }                                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    if (BAR) lds_barrier();                                                                                                         \
  }
  for (int t = 0; t < tiles; t += 2) {
    HALF(t, E0, E1, O0, O1)
    HALF(t + 1, O0, O1, E0, E1)
  }
#undef HALF
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += E0[r] + E1[r] + O0[r] + O1[r] + acc[r];
  out[blockIdx.x * 512 + tid] = sink;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, float* in, float4* w, float* out, unsigned* cyc) {
  const int tiles = 2000, grid = 256;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, in, w, out, cyc, 20);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, in, w, out, cyc, tiles);
  hipError_t e = hipDeviceSynchronize();
  unsigned h[2048];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 2048; ++i) m += h[i];
  m /= 2048;
  printf("%-84s period %7.0f ticks per tile (floor 1792)%s\n", name, m / tiles, e == hipSuccess ? "" : "  ** ERROR **");
}

// fp16 subnormal inputs through the f16 MFMA: A = 2^-20 (subnormal) on the diagonal-ish, B = 2^10: expect 16 * 2^-10 per element when honoured
__global__ void denorm_probe(float* out) {
  const int lane = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1024.0f; }   // 2^-20, 2^10
  f32x16 D;
  for (int r = 0; r < 16; ++r) D[r] = 0.f;
  D = MFMA16(a, b, D);
  out[lane] = D[0];
  // conversion: does (_Float16)x produce subnormals?
  const float x = 3.0e-6f * (float)(lane + 1);
  out[64 + lane] = (float)(_Float16)x;
  // mixed: subnormal x normal with a large accumulator (does the small term vanish or round in?)
  f32x16 E;
  for (int r = 0; r < 16; ++r) E[r] = 16777216.0f;     // 2^24
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.25f; b[i] = (_Float16)0.25f; }   // 16 products of 2^-4 = 1.0 exactly: 2^24 + 1 is not representable
  E = MFMA16(a, b, E);
  out[128 + lane] = E[0] - 16777216.0f;
  for (int r = 0; r < 16; ++r) E[r] = 16777216.0f;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.5f; b[i] = (_Float16)0.375f; }   // 16 x 0.1875 = 3.0: 2^24 + 3 -> rounds to +4 (RNE) or +2 (truncate)
  E = MFMA16(a, b, E);
  out[192 + lane] = E[0] - 16777216.0f;
}

int main() {
  float *in, *out; unsigned* cyc; float4* w;
  (void)hipMalloc(&in, 8192 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 4); (void)hipMalloc(&w, 32 * 1024 * 16);
  (void)hipMemset(w, 0, 32 * 1024 * 16);
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 17) * 0.01f + 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  {
    hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, out);
    float r[256];
    (void)hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    printf("denorm: MFMA(2^-20 x 2^10, K=16) = %g (expect %g if fp16 subnormal inputs are honoured)\n", r[0], 16.0 / 1024.0);
    printf("denorm: (_Float16)3e-6 -> %g, (_Float16)6e-6 -> %g (fp16 subnormal step 5.96e-8)\n", r[64], r[65]);
    printf("round: 2^24 + 16 x 2^-4 -> +%g ; 2^24 + 3 -> +%g (4 = round to nearest even, 2 = truncation)\n", r[128], r[192]);
  }
  run("B2: 28 MFMA, NV=0, no ring, no barrier", probe<0, false, false, 0>, in, w, out, cyc);
  run("B2: 28 MFMA, NV=0, no ring, barrier", probe<0, false, true, 0>, in, w, out, cyc);
  run("B2: NV=32 (1/MFMA), barrier", probe<32, false, true, 1>, in, w, out, cyc);
  run("B2: NV=64 (2/MFMA), barrier", probe<64, false, true, 2>, in, w, out, cyc);
  run("B2: NV=96 (3/MFMA), barrier", probe<96, false, true, 3>, in, w, out, cyc);
  run("B2: NV=128 (4/MFMA), barrier", probe<128, false, true, 4>, in, w, out, cyc);
  run("B2: NV=64, ring traffic, barrier", probe<64, true, true, 2>, in, w, out, cyc);
  run("B2: NV=96, ring traffic, barrier", probe<96, true, true, 3>, in, w, out, cyc);
  run("B2: NV=64, ring traffic, NO barrier (invalid ring, timing only)", probe<64, true, false, 2>, in, w, out, cyc);
  run("B4 (four accumulator sets): NV=0, barrier", probe4<0, false, true, 0>, in, w, out, cyc);
  run("B4: NV=64 (2/MFMA), barrier", probe4<64, false, true, 2>, in, w, out, cyc);
  run("B4: NV=96 (3/MFMA), barrier", probe4<96, false, true, 3>, in, w, out, cyc);
  run("B4: NV=64, ring traffic, barrier", probe4<64, true, true, 2>, in, w, out, cyc);
  run("B4: NV=96, ring traffic, barrier", probe4<96, true, true, 3>, in, w, out, cyc);
  run("B4: NV=96, ring traffic, NO barrier (timing only)", probe4<96, true, false, 3>, in, w, out, cyc);
  return 0;
}
