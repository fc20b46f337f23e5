// FasterTensorProduct.forward at the reference's op boundary (models/tensor_layers.py:65-116):
//   out[e] = TP(x_dst[e], sh[e], w[e])  with the per-edge weights w [E, W] resident in HBM.
// HBM-bound by construction (4 W B of weights per edge for ~5 kFLOP, 0.7 FLOP/B): the kernel is a read stream with a column walk in its shadow.
// Used by the drop-in FasterTensorProduct module and as an on-GPU cross-check of the fused kernel.
//
// tp_col_kernel (round 6).  One wave per workgroup, persistent, one edge at a time:
//  * the weight row of an edge is ONE contiguous piece of 4 W bytes: the wave reads it flat, 16 B per lane and 1 KB per wave instruction whatever the block
//    shapes (W = 1872: 8 instructions, W = 936: 4; round 5 read row-shaped pieces, 10 / 6), D rows ahead in registers, and parks it in LDS in the same flat order
//    (a read stream alone, with or without the LDS parking, runs at 6.2 - 6.6 TB/s on MI355X: the ceiling this kernel is measured against);
//  * every OUTPUT column of the edge has one lane (24 + 6 + 6 + 24 = 60 lanes) that walks its column down the block's rows: one ds_read_b32 per lane and row, no
//    partial sums between lanes, no reduction, one wave-level barrier per edge;
//  * the per-row factors that do not depend on the row (s0, v, the cross product with v) are applied ONCE per column (tensor_layers.py:75-92 is linear in the
//    rows), so the row operand of a walk step is a raw input value, the same for the whole block: the lane that loaded it from the node row hands it over with
//    v_readlane_b32 and the FMA reads it as a scalar operand - no operand table in LDS.  Every lane runs the FMAs of both operand classes of a step into separate
//    accumulators (a lane's weight is its own; the class that is not its block's is dropped at the end): no select, no divergence.
// Rows are walked in three phases that line up across the four blocks:
//   A (scalar operand; A or C rows):  0e: a_i s0 | 1o: a_i (x) v | 1e: c_i (x) v | 0o: c_i s0
//   B (P rows):                        0e: (p_i . v)/sqrt3 | 1o: p_i s0 | 1e: (p_i x v)/sqrt2
//   C (Q rows):                        1o: (q_i x v)/sqrt2 | 1e: q_i s0 | 0o: (q_i . v)/sqrt3
// Measured (tools/bench_tp.py, E = 800 000): 5.0 - 5.5 TB/s on all four layer shapes (round 5: 3.1 - 4.3); what was tried on the way, each with its number, is in
// profiles/r06_tp_boundary_kernel_stats.md (operands through LDS broadcasts, 1 / 2 / 3 rows in flight, non-temporal loads, grid sizes, occupancy targets).
#include <stdlib.h>

#include "ddk_internal.h"

namespace ddk {

struct TpFArgs {
  const float* x; const float* sh; const float* w; float* out;
  int64_t E;
};
template <int A_, int P_, int Q_, int C_, int O0_, int O1_, int O2_, int O3_>
struct TpShape {
  static constexpr int A = A_, P = P_, Q = Q_, C = C_, O0 = O0_, O1 = O1_, O2 = O2_, O3 = O3_;
  static constexpr int R0 = O0 ? A + P : 0, R1 = O1 ? A + P + Q : 0, R2 = O2 ? P + Q + C : 0, R3 = O3 ? Q + C : 0;      // tensor_layers.py:56-61
  static constexpr int B0 = 0, B1 = B0 + R0 * O0, B2 = B1 + R1 * O1, B3 = B2 + R2 * O2, W = B3 + R3 * O3;
  static constexpr int DIN = A + 3 * P + 3 * Q + C, DOUT = O0 + 3 * O1 + 3 * O2 + O3;
  static constexpr int NV = (W / 4 + 63) / 64;       // 16-B loads per lane and row
  static constexpr int XC = A + 3 * P + 3 * Q;       // first 0o input
  static_assert(A % 8 == 0 && C % 8 == 0 && (A == C || A == 0 || C == 0) && (C == 0 || XC % 4 == 0), "phase A is walked in two halves");
  static_assert(W % 4 == 0 && DIN <= 128 && O0 + O1 + O2 + O3 <= 64 && P <= 8 && Q <= 8, "shape outside the kernel's layout");
};
typedef float tp_f4 __attribute__((ext_vector_type(4), aligned(4)));
template <class S> struct TpRow { float4 w[S::NV]; float x0, x1; float4 sh; };

template <class S>
__device__ __forceinline__ void tpf_request(const TpFArgs& a, int64_t e, int lane, TpRow<S>& R) {
  const float4* wr = reinterpret_cast<const float4*>(a.w + e * S::W);
#pragma unroll
  for (int t = 0; t < S::NV; ++t) {
    const int f = 64 * t + lane;
    if ((t + 1) * 64 <= S::W / 4 || f < S::W / 4) {
      const tp_f4* pw = reinterpret_cast<const tp_f4*>(wr + f);
      const tp_f4 q = *pw;      // (default cache policy: non-temporal loads measured 1-3 % slower on this stream)
      R.w[t] = make_float4(q.x, q.y, q.z, q.w);
    }
  }
  const float* xr = a.x + e * S::DIN;
  R.x0 = lane < S::DIN ? xr[lane] : 0.f;
  R.x1 = lane + 64 < S::DIN ? xr[lane + 64] : 0.f;
  R.sh = *reinterpret_cast<const float4*>(a.sh + e * 4);
}

__device__ __forceinline__ float tp_rl(float v, int k) {      // v_readlane_b32 on the bit pattern (the builtin is typed int: a plain call would convert the VALUE)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
template <class S>
__device__ __forceinline__ float tp_xval(float x0, float x1, int k) {      // x[k] of the node row as a wave-uniform value (k is a compile-time constant)
  return tp_rl(k < 64 ? x0 : x1, k < 64 ? k : k - 64);
}
template <class S, int D, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void tp_col_kernel(TpFArgs a) {
  __shared__ float4 Wst[S::NV * 64];
  const char* Wb = reinterpret_cast<const char*>(Wst);
  const int lane = threadIdx.x;
  const int64_t stride = gridDim.x;
  int64_t e = blockIdx.x;
  const int blk = lane < S::O0 ? 0 : (lane < S::O0 + S::O1 ? 1 : (lane < S::O0 + S::O1 + S::O2 ? 2 : (lane < S::O0 + S::O1 + S::O2 + S::O3 ? 3 : 4)));
  const int col = lane - (blk == 0 ? 0 : (blk == 1 ? S::O0 : (blk == 2 ? S::O0 + S::O1 : S::O0 + S::O1 + S::O2)));
  const int nout = blk == 0 ? S::O0 : (blk == 1 ? S::O1 : (blk == 2 ? S::O2 : (blk == 3 ? S::O3 : 0)));
  const int bbase = blk == 0 ? S::B0 : (blk == 1 ? S::B1 : (blk == 2 ? S::B2 : S::B3));
  const bool lowblk = blk <= 1;                                                  // phase A operand: a_i (blocks 0e, 1o) or c_i (1e, 0o)
  const bool onA = nout > 0 && (lowblk ? S::A > 0 : S::C > 0), onB = blk <= 2 && S::P > 0 && nout > 0, onC = blk >= 1 && blk <= 3 && S::Q > 0 && nout > 0;
  // byte addresses of the lane's first weight of each phase, byte stride between rows (0: the lane idles through the phase on weight 0)
  const int sA = onA ? 4 * nout : 0, sB = onB ? 4 * nout : 0, sC = onC ? 4 * nout : 0;
  const int wA0 = onA ? 4 * (bbase + (blk <= 1 ? 0 : (blk == 2 ? S::P + S::Q : S::Q)) * nout + col) : 0;
  const int wB0 = onB ? 4 * (bbase + (blk == 2 ? 0 : S::A) * nout + col) : 0;
  const int wC0 = onC ? 4 * (bbase + (blk == 1 ? S::A + S::P : (blk == 2 ? S::P : 0)) * nout + col) : 0;
  const int nrow = blk == 0 ? S::R0 : (blk == 1 ? S::R1 : (blk == 2 ? S::R2 : (blk == 3 ? S::R3 : 1)));
  const float rs = 1.0f / sqrtf((float)(nrow > 0 ? nrow : 1));                   // tensor_layers.py:89-92
  const int ooff = blk == 0 ? col : (blk == 1 ? S::O0 + 3 * col : (blk == 2 ? S::O0 + 3 * S::O1 + 3 * col : S::O0 + 3 * S::O1 + 3 * S::O2 + col));
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  constexpr int NA = S::A > S::C ? S::A : S::C;
  // which component of v multiplies this lane's node-row value in the (p . v), (q . v) sums: lanes A .. A + 3P + 3Q - 1 of x0 hold p then q, xyz interleaved
  static_assert(S::A + 3 * S::P + 3 * S::Q <= 64 && NA % 2 == 0, "the vector inputs of the node row sit in the first 64 lanes");
  const int vsel = (lane >= S::A && lane < S::XC) ? (lane - S::A) % 3 : 3;

  TpRow<S> R[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (e + d * stride < a.E) tpf_request<S>(a, e + d * stride, lane, R[d]);
  for (;;) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (e >= a.E) return;
#pragma unroll
      for (int t = 0; t < S::NV; ++t) Wst[64 * t + lane] = R[d].w[t];
      const float x0 = R[d].x0, x1 = R[d].x1;
      const float s0 = R[d].sh.x, vx = R[d].sh.y, vy = R[d].sh.z, vz = R[d].sh.w;
      if (e + (int64_t)D * stride < a.E) tpf_request<S>(a, e + (int64_t)D * stride, lane, R[d]);
      // (p_i . v)/sqrt3 in lane A + 3i, (q_i . v)/sqrt3 in lane A + 3P + 3i: three neighbouring lanes' products through the lane crossbar
      float pvq = 0.f;
      if (S::P + S::Q > 0) {
        const float t = x0 * (vsel == 0 ? vx : (vsel == 1 ? vy : (vsel == 2 ? vz : 0.f)));
        pvq = (t + __shfl_down(t, 1) + __shfl_down(t, 2)) * inv_s3;
      }
      __syncthreads();
      float accA = 0.f, accC = 0.f, bs = 0.f, bx = 0.f, by = 0.f, bz = 0.f, cs = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
      // The walk as a two-stage software pipeline over half phases: the column reads of the next half are issued before the FMAs of the current one
      // (the accumulator chains are serial, the reads are not).  The row addresses advance in ONE register per phase: left to itself the compiler keeps all
      // 36 of them, loop-invariant, in VGPRs - the opaque asm pins the chain.
      constexpr int HA = NA / 2;
      float wa[HA > 0 ? HA : 1], wb[HA > 0 ? HA : 1], wp[S::P > 0 ? S::P : 1], wq[S::Q > 0 ? S::Q : 1];
      int ad = wA0;
      asm volatile("" : "+v"(ad));
#pragma unroll
      for (int i = 0; i < HA; ++i) { wa[i] = *reinterpret_cast<const float*>(Wb + ad); ad += sA; asm volatile("" : "+v"(ad)); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < HA; ++i) { wb[i] = *reinterpret_cast<const float*>(Wb + ad); ad += sA; asm volatile("" : "+v"(ad)); }
#pragma unroll
      for (int i = 0; i < HA; ++i) {
        if (i < S::A) accA = fmaf(tp_xval<S>(x0, x1, i), wa[i], accA);
        if (i < S::C) accC = fmaf(tp_xval<S>(x0, x1, S::XC + i), wa[i], accC);
      }
      __builtin_amdgcn_sched_barrier(0);
      ad = wB0;
      asm volatile("" : "+v"(ad));
#pragma unroll
      for (int i = 0; i < S::P; ++i) { wp[i] = *reinterpret_cast<const float*>(Wb + ad); ad += sB; asm volatile("" : "+v"(ad)); }
      ad = wC0;
      asm volatile("" : "+v"(ad));
#pragma unroll
      for (int i = 0; i < S::Q; ++i) { wq[i] = *reinterpret_cast<const float*>(Wb + ad); ad += sC; asm volatile("" : "+v"(ad)); }
#pragma unroll
      for (int i = 0; i < HA; ++i) {
        if (HA + i < S::A) accA = fmaf(tp_xval<S>(x0, x1, HA + i), wb[i], accA);
        if (HA + i < S::C) accC = fmaf(tp_xval<S>(x0, x1, S::XC + HA + i), wb[i], accC);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < S::P; ++i) {
        bs = fmaf(tp_rl(pvq, S::A + 3 * i), wp[i], bs);
        bx = fmaf(tp_xval<S>(x0, x1, S::A + 3 * i), wp[i], bx);
        by = fmaf(tp_xval<S>(x0, x1, S::A + 3 * i + 1), wp[i], by);
        bz = fmaf(tp_xval<S>(x0, x1, S::A + 3 * i + 2), wp[i], bz);
      }
#pragma unroll
      for (int i = 0; i < S::Q; ++i) {
        cs = fmaf(tp_rl(pvq, S::A + 3 * S::P + 3 * i), wq[i], cs);
        cx = fmaf(tp_xval<S>(x0, x1, S::A + 3 * S::P + 3 * i), wq[i], cx);
        cy = fmaf(tp_xval<S>(x0, x1, S::A + 3 * S::P + 3 * i + 1), wq[i], cy);
        cz = fmaf(tp_xval<S>(x0, x1, S::A + 3 * S::P + 3 * i + 2), wq[i], cz);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float sa = onA ? (lowblk ? accA : accC) : 0.f;
      if (!onB) { bs = 0.f; bx = 0.f; by = 0.f; bz = 0.f; }
      if (!onC) { cs = 0.f; cx = 0.f; cy = 0.f; cz = 0.f; }
      float* o = a.out + e * S::DOUT + ooff;
      if (blk == 0) {
        o[0] = (s0 * sa + bs) * rs;
      } else if (blk == 1) {      // a (x) v + p s0 + (q x v)/sqrt2
        o[0] = (vx * sa + s0 * bx + (cy * vz - cz * vy) * inv_s2) * rs;
        o[1] = (vy * sa + s0 * by + (cz * vx - cx * vz) * inv_s2) * rs;
        o[2] = (vz * sa + s0 * bz + (cx * vy - cy * vx) * inv_s2) * rs;
      } else if (blk == 2) {      // (p x v)/sqrt2 + q s0 + c (x) v
        o[0] = ((by * vz - bz * vy) * inv_s2 + s0 * cx + vx * sa) * rs;
        o[1] = ((bz * vx - bx * vz) * inv_s2 + s0 * cy + vy * sa) * rs;
        o[2] = ((bx * vy - by * vx) * inv_s2 + s0 * cz + vz * sa) * rs;
      } else if (blk == 3) {
        o[0] = (cs + s0 * sa) * rs;
      }
      e += stride;
      __syncthreads();
    }
  }
}

hipError_t launch_tp_forward(const ConvLayerDev& L, const float* x_dst, const float* sh, const float* w, int64_t E,
                             float* out, hipStream_t s) {
  if (E == 0) return hipSuccess;
  const TpFArgs a{x_dst, sh, w, out, E};
  // persistent waves, one per workgroup: 16 per CU cover the occupancy of every instantiation (3 .. 5 waves per SIMD), each with one row in hand and two in flight
  const int64_t cap = 256 * 16;
  const dim3 g((unsigned)(E < cap ? E : cap)), b(64);
  const int* im = L.in_mul;
  const int* om = L.out_mul;
  auto is = [&](int a0, int a1, int a2, int a3, int o0, int o1, int o2, int o3) {
    return im[0] == a0 && im[1] == a1 && im[2] == a2 && im[3] == a3 && om[0] == o0 && om[1] == o1 && om[2] == o2 && om[3] == o3;
  };
  // the conv layers of the score model (tensor_layers.py:12-27): in = 24x0e [+ 6x1o [+ 6x1e [+ 24x0o]]], out = the next entry of the sequence.  Rows in
  // flight per wave and waves per SIMD as measured on MI355X (profiles/r06_tp_boundary_kernel_stats.md): the W = 1872 row wants its registers (2 waves
  // per SIMD requested, 3 fit), the shorter rows want the occupancy
  if (is(24, 6, 6, 24, 24, 6, 6, 24)) hipLaunchKernelGGL((tp_col_kernel<TpShape<24, 6, 6, 24, 24, 6, 6, 24>, 2, 2>), g, b, 0, s, a);
  else if (is(24, 6, 6, 0, 24, 6, 6, 24)) hipLaunchKernelGGL((tp_col_kernel<TpShape<24, 6, 6, 0, 24, 6, 6, 24>, 2, 4>), g, b, 0, s, a);
  else if (is(24, 6, 0, 0, 24, 6, 6, 0)) hipLaunchKernelGGL((tp_col_kernel<TpShape<24, 6, 0, 0, 24, 6, 6, 0>, 2, 4>), g, b, 0, s, a);
  else if (is(24, 0, 0, 0, 24, 6, 0, 0)) hipLaunchKernelGGL((tp_col_kernel<TpShape<24, 0, 0, 0, 24, 6, 0, 0>, 2, 4>), g, b, 0, s, a);
  else return hipErrorInvalidValue;      // not a FasterTensorProduct of this model family (ddk_create refuses other ns / nv)
  return hipGetLastError();
}

}  // namespace ddk
