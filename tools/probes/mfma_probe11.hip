// Micro-probe 11 (from probe 10): the "free-running, software-pipelined" tile loop that round 4 weighs against k_conv_x.hip's strict alternation.
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


// Probe 11: probe 10's streams with the per-tile s_barrier replaced by LDS counters over a FOUR-stage ring (SYNC = 2): every wave adds one to
// ready[stage] behind its two ds_write_b128 of tile t+2 and one to done[stage] behind its last fragment read of tile t; a tile is consumed when
// ready[t & 3] >= 8 (uses + 1) and its stage is refilled when done[stage] >= 8 uses.  The waves drift up to two tiles apart instead of meeting
// at a barrier every tile.  SYNC = 1: one s_barrier per tile (2 of the 4 stages used); SYNC = 0: nothing (timing only).
constexpr int NST = 4;

__device__ __forceinline__ void lds_signal(int* p) {      // one lane adds 1 (no branch: EXEC is set inside the statement)
  asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" ::"v"((unsigned)(size_t)p), "v"(1) : "memory");
}
__device__ __forceinline__ void lds_wait_ge(int* p, int need) {
  while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) __builtin_amdgcn_s_sleep(1);
}

#define STEP6X(f, s, A0, A1)                         \
  A0 = MFMA16(f.h, bl[s], A0);                       \
  A1 = MFMA16(f.h, bm[s], A1);                       \
  A0 = MFMA16(f.l, bh[s], A0);                       \
  A1 = MFMA16(f.h, bh[s], A1);                       \
  A0 = MFMA16(f.m, bm[s], A0);                       \
  A1 = MFMA16(f.m, bh[s], A1);

template <int NV, int SYNC, int VPM, bool FOUR>
__global__ __launch_bounds__(512) void probe(const float* in, const float4* w, float* out, unsigned* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c1 = min(tid + 512, 872);
  char* ring = reinterpret_cast<char*>(lds);
  int* ready = reinterpret_cast<int*>(ring + NST * 14080);
  int* done = ready + NST;
  float* F = reinterpret_cast<float*>(ring + NST * 14080 + 64) + wave * (32 * 96) + (lane & 31) * 96;
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
  __syncthreads();
  if (tid < 2 * NST) ready[tid] = (tid < 2) ? 8 : 0;
  float acc[16];
  for (int k = 0; k < 16; ++k) acc[k] = in[(lane + k) & 1023];
  f32x16 E0, E1, O0, O1;
  for (int r = 0; r < 16; ++r) { O0[r] = in[(lane + r + 7) & 1023]; O1[r] = 0.f; E0[r] = 0.f; E1[r] = 0.f; }
  __syncthreads();
  Frag16 nf0 = lds_frag16(ring, 0, lane);
  const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
#define TILE(t, A0, A1, P0, P1)                                                                                                     \
  {                                                                                                                                 \
    const int st = SYNC >= 2 ? ((t) & 3) : ((t) & 1), sf = SYNC >= 2 ? (((t) + 2) & 3) : (((t) + 1) & 1);                           \
    const char* stage = ring + st * 14080;                                                                                          \
    char* other = ring + sf * 14080;                                                                                                \
    const float* Fp = F + (((t) * 4) & 63);                                                                                         \
    const float4 g0 = w[(((t) & 31) * 1024 + tid)];                                                                                 \
    const float4 g1 = w[(((t) & 31) * 1024 + c1)];                                                                                  \
    if (SYNC == 2) {                                                                                                                \
      lds_wait_ge(ready + st, 8 * (((t) >> 2) + 1));                                                                                \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                                               \
    }                                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    const Frag16 f0 = SYNC == 3 ? nf0 : lds_frag16(stage, 0, lane);                                                                 \
    const f32x4 fv = *reinterpret_cast<const f32x4*>(Fp);                                                                           \
    const Frag16 f1 = lds_frag16(stage, 1, lane);                                                                                   \
    const Frag16 f2 = lds_frag16(stage, 2, lane);                                                                                   \
    const Frag16 f3 = lds_frag16(stage, 3, lane);                                                                                   \
    const f16x4 ath = *reinterpret_cast<const f16x4*>(stage + 4096 + lane * 8);                                                     \
    const f16x4 atm = *reinterpret_cast<const f16x4*>(stage + LIMB_BYTES + 4096 + lane * 8);                                        \
    const f16x4 atl = *reinterpret_cast<const f16x4*>(stage + 2 * LIMB_BYTES + 4096 + lane * 8);                                    \
    if (FOUR) { _Pragma("unroll") for (int r = 0; r < 16; ++r) P0[r] += P1[r]; }                                                    \
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
    __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);                                                                              \
    if (FOUR) __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);                                                                   \
    _Pragma("unroll") for (int k = 0; k < 28; ++k) {                                                                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                            \
      if (k < 11) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                \
      if (VPM > 0) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                                                             \
    }                                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                              \
    if (SYNC == 2) {                                                                                                                \
      lds_signal(done + st);                            /* every fragment read of this tile has returned (they fed the MFMAs above) */ \
      lds_wait_ge(done + sf, 8 * (((t) + 2) >> 2));     /* the stage's previous tile has been consumed by all eight waves */        \
    }                                                                                                                               \
    *reinterpret_cast<float4*>(other + 16 * tid) = g0;                                                                              \
    *reinterpret_cast<float4*>(other + 16 * c1) = g1;                                                                               \
    if (SYNC == 2) {                                                                                                                \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                                               \
      lds_signal(ready + sf);                                                                                                       \
    }                                                                                                                               \
    if (!FOUR) { _Pragma("unroll") for (int r = 0; r < 16; ++r) P0[r] = A0[r] + A1[r]; }                                            \
    if (SYNC == 3) nf0 = lds_frag16(ring + (((t) + 1) & 3) * 14080, 0, lane);                                                       \
    if (SYNC == 1 || SYNC == 3) lds_barrier();                                                                                      \
  }
  if (FOUR) {
    for (int t = 0; t < tiles; t += 2) {
      TILE(t, E0, E1, O0, O1)
      TILE(t + 1, O0, O1, E0, E1)
    }
  } else {
    for (int t = 0; t < tiles; ++t) TILE(t, E0, E1, O0, O1)
  }
#undef TILE
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
  double m = 0, mx = 0;
  for (int i = 0; i < 2048; ++i) { m += h[i]; if (h[i] > mx) mx = h[i]; }
  m /= 2048;
  printf("%-72s %7.0f ticks per tile (max wave %7.0f)%s\n", name, m / tiles, mx / tiles, e == hipSuccess ? "" : "  ** ERROR **");
}

int main() {
  float *in, *out; unsigned* cyc; float4* w;
  (void)hipMalloc(&in, 8192 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 2048 * 4); (void)hipMalloc(&w, 32 * 1024 * 16);
  (void)hipMemset(w, 0, 32 * 1024 * 16);
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 37) % 17) * 0.01f + 0.01f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run("two acc sets, NV=96, ring, no sync (timing only)", probe<96, 0, 3, false>, in, w, out, cyc);
  run("two acc sets, NV=96, ring, barrier per tile", probe<96, 1, 3, false>, in, w, out, cyc);
  run("two acc sets, NV=96, ring, LDS counters (4 stages)", probe<96, 2, 3, false>, in, w, out, cyc);
  run("two acc sets, NV=64, ring, LDS counters", probe<64, 2, 2, false>, in, w, out, cyc);
  run("two acc sets, NV=128, ring, LDS counters", probe<128, 2, 4, false>, in, w, out, cyc);
  run("four acc sets, NV=96, ring, no sync (timing only)", probe<96, 0, 3, true>, in, w, out, cyc);
  run("four acc sets, NV=96, ring, barrier per tile", probe<96, 1, 3, true>, in, w, out, cyc);
  run("four acc sets, NV=96, ring, LDS counters (4 stages)", probe<96, 2, 3, true>, in, w, out, cyc);
  run("four acc sets, NV=64, ring, LDS counters", probe<64, 2, 2, true>, in, w, out, cyc);
  run("four acc sets, NV=128, ring, LDS counters", probe<128, 2, 4, true>, in, w, out, cyc);
  run("four acc sets, NV=0, ring, LDS counters", probe<0, 2, 0, true>, in, w, out, cyc);
  run("four acc sets, NV=96, 4-stage ring, barrier per tile, next K step 0 prefetched", probe<96, 3, 3, true>, in, w, out, cyc);
  run("four acc sets, NV=64, 4-stage ring, barrier per tile, next K step 0 prefetched", probe<64, 3, 2, true>, in, w, out, cyc);
  run("two acc sets, NV=96, 4-stage ring, barrier per tile, next K step 0 prefetched", probe<96, 3, 3, false>, in, w, out, cyc);
  return 0;
}
