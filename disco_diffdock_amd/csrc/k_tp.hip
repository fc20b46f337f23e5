// FasterTensorProduct.forward at the reference's op boundary (models/tensor_layers.py:65-116):
//   out[e] = TP(x_dst[e], sh[e], w[e])  with the per-edge weights w [E, W] resident in HBM.
// HBM-bound by construction (W*4 B of weights per edge for ~5 kFLOP): one wave per edge streams the weight row with vector loads
// (tp_stream_kernel below); the <=276 row operands u_i live in an LDS table per wave.
// Used by the drop-in FasterTensorProduct module and as an on-GPU cross-check of the fused kernel.
#include <stdlib.h>

#include "ddk_internal.h"

namespace ddk {

struct TpKArgs {
  const float* x;    // [E, din]
  const float* sh;   // [E, 4]
  const float* w;    // [E, W]
  float* out;        // [E, dout]
  int64_t E;
  int din, dout, W;
  int n_in[4], n_out[4], blk_off[4], out_off[4];
  int in_mul[4];     // multiplicities 0e,1o,1e,0o of the input irreps
};

constexpr int U_MAX = 2 * (NS + NV) + 2 * 3 * (NS + 2 * NV);   // 30+30+108+108 = 276
constexpr int TP_ITS = 4;          // weight rows a lane holds per block (n_in <= TP_ITS * rows per instruction)
constexpr int TP_PART = 768;       // partial sums of one block: 64 lanes x vector width 4 x 3 components

// Streaming form (round 5).  ONE wave per workgroup, one edge at a time, the NEXT edge's weight row requested before the current
// one is consumed (two register sets: ~15 KB in flight per wave, 8 waves per CU).  A block's weights [n_in, n_out] are read with the widest vector
// that keeps a lane inside one row (n_out = 24: float4, 6 lanes per row, 10 rows = 960 contiguous bytes per wave instruction; n_out = 6: float2, 3 lanes
// per row, 21 rows): 10 wave instructions per edge at W = 1872 instead of 38 dword reads.  A lane accumulates its columns over its rows (row operand
// u_i from the wave's LDS table), the row groups' partial sums meet in LDS (plain stores, one reader per output: no atomics), outputs are stored
// straight from the reducing lanes.  Barriers are workgroup = wave wide and every lane runs the same trip count.
struct TpBlk { int v, lpr, R, its; };      // vector width, lanes per row, rows per instruction, instructions per block

template <int V> struct TpVec;
template <> struct TpVec<4> { typedef float4 T; };
template <> struct TpVec<2> { typedef float2 T; };
template <> struct TpVec<1> { typedef float T; };

// geometry of a block with NO output columns (compile time: 0 / 6 / 24 columns in the score model's conv layers)
template <int NO> struct TpGeo {
  static constexpr int v = NO % 4 == 0 ? 4 : (NO % 2 == 0 ? 2 : 1), lpr = NO > 0 ? NO / v : 1, R = 64 / lpr;
};
template <int NO>
__device__ __forceinline__ void tp_load(const float* wb, int n_out, int n_in, const TpBlk& Bd, int lane, float4 (&r)[TP_ITS]) {
  if (NO == 0) return;
  const int v = NO > 0 ? TpGeo<NO>::v : Bd.v, lpr = NO > 0 ? TpGeo<NO>::lpr : Bd.lpr, R = NO > 0 ? TpGeo<NO>::R : Bd.R;
  const int rg = lane / lpr, kq = lane - rg * lpr;
  const bool on = lane < R * lpr;
#pragma unroll
  for (int t = 0; t < TP_ITS; ++t) {
    r[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int i = rg + R * t;
    if (on && i < n_in) {
      const float* p = wb + (size_t)i * n_out + v * kq;
      if (v == 4) r[t] = *reinterpret_cast<const float4*>(p);
      else if (v == 2) { const float2 q = *reinterpret_cast<const float2*>(p); r[t].x = q.x; r[t].y = q.y; }
      else r[t].x = *p;
    }
  }
}

struct TpSArgs {
  TpKArgs k;
  TpBlk blk[4];
};

// everything one edge needs from global memory (requested one edge ahead, in this order: the row operands' inputs are used first)
struct TpEdge {
  float x0, x1;          // x_dst[e][lane], x_dst[e][lane + 64]
  float4 sh;
  float4 w[4][TP_ITS];
};
template <int N0, int N1, int N2, int N3>
__device__ __forceinline__ void tp_request(const TpSArgs& S, int64_t e, int lane, TpEdge& Q) {
  const TpKArgs& A = S.k;
  const float* xr = A.x + e * A.din;
  Q.x0 = lane < A.din ? xr[lane] : 0.f;
  Q.x1 = lane + 64 < A.din ? xr[lane + 64] : 0.f;
  Q.sh = *reinterpret_cast<const float4*>(A.sh + e * 4);
  const float* wr = A.w + e * A.W;
  tp_load<N0>(wr + A.blk_off[0], A.n_out[0], A.n_in[0], S.blk[0], lane, Q.w[0]);
  tp_load<N1>(wr + A.blk_off[1], A.n_out[1], A.n_in[1], S.blk[1], lane, Q.w[1]);
  tp_load<N2>(wr + A.blk_off[2], A.n_out[2], A.n_in[2], S.blk[2], lane, Q.w[2]);
  tp_load<N3>(wr + A.blk_off[3], A.n_out[3], A.n_in[3], S.blk[3], lane, Q.w[3]);
}

// one block of one edge: every lane its columns over its rows, the row groups' partial sums through LDS, one lane per output
template <int NO, bool VEC>
__device__ __forceinline__ void tp_block_acc(const TpKArgs& A, int b, const TpBlk& Bd, int lane, const float4 (&w)[TP_ITS], const float* U, int ubb, float* Pb) {
  if (NO == 0) return;
  const int n_in = A.n_in[b], n_out = A.n_out[b];
  if (NO < 0 && (n_in == 0 || n_out == 0)) return;
  const int v = NO > 0 ? TpGeo<NO>::v : Bd.v, lpr = NO > 0 ? TpGeo<NO>::lpr : Bd.lpr, R = NO > 0 ? TpGeo<NO>::R : Bd.R;
  const int rg = lane / lpr, kq = lane - rg * lpr;
  if (lane < R * lpr) {
    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; }
#pragma unroll
    for (int t = 0; t < TP_ITS; ++t) {
      const int i = rg + R * t;
      if (i < n_in) {
        const float wv[4] = {w[t].x, w[t].y, w[t].z, w[t].w};
        if (VEC) {
          const float u0 = U[ubb + 3 * i], u1 = U[ubb + 3 * i + 1], u2 = U[ubb + 3 * i + 2];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < v) { acc[j][0] = fmaf(u0, wv[j], acc[j][0]); acc[j][1] = fmaf(u1, wv[j], acc[j][1]); acc[j][2] = fmaf(u2, wv[j], acc[j][2]); }
        } else {
          const float u0 = U[ubb + i];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < v) acc[j][0] = fmaf(u0, wv[j], acc[j][0]);
        }
      }
    }
    constexpr int C = VEC ? 3 : 1;
    float* pr = Pb + (rg * n_out + v * kq) * C;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < v) {
        pr[j * C] = acc[j][0];
        if (VEC) { pr[j * C + 1] = acc[j][1]; pr[j * C + 2] = acc[j][2]; }
      }
  }
}
template <int NO, bool VEC>
__device__ __forceinline__ void tp_block_out(const TpKArgs& A, int b, const TpBlk& Bd, int lane, int64_t e, const float* Pb) {
  if (NO == 0) return;
  const int n_in = A.n_in[b], n_out = A.n_out[b];
  if (NO < 0 && (n_in == 0 || n_out == 0)) return;
  constexpr int C = VEC ? 3 : 1;
  const int n = n_out * C, R = NO > 0 ? TpGeo<NO>::R : Bd.R;
  const float rs = 1.0f / sqrtf((float)n_in);
  for (int idx = lane; idx < n; idx += 64) {
    float sum = 0.f;
    if (NO > 0) {
#pragma unroll
      for (int g = 0; g < TpGeo<NO>::R; ++g) sum += Pb[g * n + idx];
    } else {
      for (int g = 0; g < R; ++g) sum += Pb[g * n + idx];
    }
    A.out[e * A.dout + A.out_off[b] + idx] = sum * rs;
  }
}

template <int N0, int N1, int N2, int N3>
__device__ __forceinline__ void tp_edge(const TpSArgs& S, int64_t e, int lane, const TpEdge& Q, float* X, float* U, float (*P)[TP_PART]) {
  const TpKArgs& A = S.k;
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  const int n0e = A.in_mul[0], n1o = A.in_mul[1], n1e = A.in_mul[2], n0o = A.in_mul[3];
  const int ub[4] = {0, A.n_in[0], A.n_in[0] + 3 * A.n_in[1], A.n_in[0] + 3 * A.n_in[1] + 3 * A.n_in[2]};
  X[lane] = Q.x0;
  X[lane + 64] = Q.x1;
  __syncthreads();
  // ---- row operands of this edge (tensor_layers.py:75-92): u0e[n_in0], u1o[n_in1][3], u1e[n_in2][3], u0o[n_in3] ----
  const float s0 = Q.sh.x, vx = Q.sh.y, vy = Q.sh.z, vz = Q.sh.w;
  const float* pa = X;
  const float* pp = X + n0e;
  const float* pq = pp + 3 * n1o;
  const float* pc = pq + 3 * n1e;
  for (int i = lane; i < n0e; i += 64) {
    const float a = pa[i];
    U[ub[0] + i] = a * s0;
    U[ub[1] + 3 * i] = a * vx; U[ub[1] + 3 * i + 1] = a * vy; U[ub[1] + 3 * i + 2] = a * vz;
  }
  for (int i = lane; i < n0o; i += 64) {
    const float c = pc[i];
    const int r1e = n1o + n1e + i, r0o = n1e + i;
    U[ub[2] + 3 * r1e] = c * vx; U[ub[2] + 3 * r1e + 1] = c * vy; U[ub[2] + 3 * r1e + 2] = c * vz;
    U[ub[3] + r0o] = c * s0;
  }
  for (int m = lane; m < n1o; m += 64) {
    const float px = pp[3 * m], py = pp[3 * m + 1], pz = pp[3 * m + 2];
    U[ub[0] + n0e + m] = (px * vx + py * vy + pz * vz) * inv_s3;
    const int r1o = n0e + m;
    U[ub[1] + 3 * r1o] = px * s0; U[ub[1] + 3 * r1o + 1] = py * s0; U[ub[1] + 3 * r1o + 2] = pz * s0;
    U[ub[2] + 3 * m] = (py * vz - pz * vy) * inv_s2;
    U[ub[2] + 3 * m + 1] = (pz * vx - px * vz) * inv_s2;
    U[ub[2] + 3 * m + 2] = (px * vy - py * vx) * inv_s2;
  }
  for (int m = lane; m < n1e; m += 64) {
    const float qx = pq[3 * m], qy = pq[3 * m + 1], qz = pq[3 * m + 2];
    const int r1o = n0e + n1o + m, r1e = n1o + m;
    U[ub[1] + 3 * r1o] = (qy * vz - qz * vy) * inv_s2;
    U[ub[1] + 3 * r1o + 1] = (qz * vx - qx * vz) * inv_s2;
    U[ub[1] + 3 * r1o + 2] = (qx * vy - qy * vx) * inv_s2;
    U[ub[2] + 3 * r1e] = qx * s0; U[ub[2] + 3 * r1e + 1] = qy * s0; U[ub[2] + 3 * r1e + 2] = qz * s0;
    U[ub[3] + m] = (qx * vx + qy * vy + qz * vz) * inv_s3;
  }
  __syncthreads();
  tp_block_acc<N0, false>(A, 0, S.blk[0], lane, Q.w[0], U, ub[0], P[0]);
  tp_block_acc<N1, true>(A, 1, S.blk[1], lane, Q.w[1], U, ub[1], P[1]);
  tp_block_acc<N2, true>(A, 2, S.blk[2], lane, Q.w[2], U, ub[2], P[2]);
  tp_block_acc<N3, false>(A, 3, S.blk[3], lane, Q.w[3], U, ub[3], P[3]);
  __syncthreads();
  // ---- one lane per output: the row groups' partial sums in group order, the block's 1/sqrt(n_in) (tensor_layers.py:89-92), the store ----
  tp_block_out<N0, false>(A, 0, S.blk[0], lane, e, P[0]);
  tp_block_out<N1, true>(A, 1, S.blk[1], lane, e, P[1]);
  tp_block_out<N2, true>(A, 2, S.blk[2], lane, e, P[2]);
  tp_block_out<N3, false>(A, 3, S.blk[3], lane, e, P[3]);
}

template <int N0, int N1, int N2, int N3>
__global__ __launch_bounds__(64) void tp_stream_kernel(TpSArgs S) {
  const TpKArgs& A = S.k;
  __shared__ float X[128];
  __shared__ float U[U_MAX + 4];
  __shared__ __attribute__((aligned(16))) float P[4][TP_PART];
  const int lane = threadIdx.x;
  const int64_t stride = gridDim.x;
  int64_t e = blockIdx.x;
  if (e >= A.E) return;
  TpEdge Q0, Q1;
  tp_request<N0, N1, N2, N3>(S, e, lane, Q0);
  // two register sets, roles swapped by the unrolled loop (no copies); every lane of the wave runs the same trip count
  for (;;) {
    const bool more1 = e + stride < A.E;
    if (more1) tp_request<N0, N1, N2, N3>(S, e + stride, lane, Q1);
    tp_edge<N0, N1, N2, N3>(S, e, lane, Q0, X, U, P);
    if (!more1) break;
    e += stride;
    __syncthreads();
    const bool more0 = e + stride < A.E;
    if (more0) tp_request<N0, N1, N2, N3>(S, e + stride, lane, Q0);
    tp_edge<N0, N1, N2, N3>(S, e, lane, Q1, X, U, P);
    if (!more0) break;
    e += stride;
    __syncthreads();
  }
}


// ---- Column-owner form (round 6) ------------------------------------------------------------------------------------
// The weight row of an edge is ONE contiguous piece of 4 W bytes: the wave reads it flat, 16 B per lane and 1 KB per wave instruction whatever the
// block shapes (W = 1872: 8 instructions, W = 936: 4; the row-shaped reads above needed 10 / 6), D rows ahead in registers, and parks it in LDS in the
// same flat order.  Then every OUTPUT column of the edge has one lane (24 + 6 + 6 + 24 = 60 lanes) that walks its column down the block's rows:
// no partial sums between lanes, no reduction, no barrier beyond the wave's own LDS order.  The per-row factors that do not depend on the row
// (s0, v, the cross product with v) are applied ONCE per column (tensor_layers.py:75-92 is linear in the rows), so the row operand of a lane is a raw
// input value: a_i / c_i (one LDS word, the same address for the whole block: a broadcast) or the raw p_i / q_i vector (one 16-B record).
// Rows are walked in three phases that line up across the four blocks:
//   A (scalar operand; A or C rows):  0e: a_i s0 | 1o: a_i (x) v | 1e: c_i (x) v | 0o: c_i s0
//   B (P rows):                        0e: (p_i . v)/sqrt3 | 1o: p_i s0 | 1e: (p_i x v)/sqrt2
//   C (Q rows):                        1o: (q_i x v)/sqrt2 | 1e: q_i s0 | 0o: (q_i . v)/sqrt3
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
  static_assert(A % 8 == 0 && C % 8 == 0 && (A == C || A == 0 || C == 0) && (C == 0 || XC % 4 == 0), "phase A walks 8 rows at a time, operands in 16-B groups");
  static_assert(W % 4 == 0 && DIN <= 128 && O0 + O1 + O2 + O3 <= 64 && P <= 8 && Q <= 8, "shape outside the flat kernel's layout");
};
typedef float tp_f4 __attribute__((ext_vector_type(4), aligned(4)));
template <class S> struct TpRow { float4 w[S::NV]; float x0, x1; float4 sh; };

template <class S, bool NT>
__device__ __forceinline__ void tpf_request(const TpFArgs& a, int64_t e, int lane, TpRow<S>& R) {
  const float4* wr = reinterpret_cast<const float4*>(a.w + e * S::W);
#pragma unroll
  for (int t = 0; t < S::NV; ++t) {
    const int f = 64 * t + lane;
    if ((t + 1) * 64 <= S::W / 4 || f < S::W / 4) {
      const tp_f4* pw = reinterpret_cast<const tp_f4*>(wr + f);
      const tp_f4 q = NT ? __builtin_nontemporal_load(pw) : *pw;
      R.w[t] = make_float4(q.x, q.y, q.z, q.w);
    }
  }
  const float* xr = a.x + e * S::DIN;
  R.x0 = lane < S::DIN ? (NT ? __builtin_nontemporal_load(xr + lane) : xr[lane]) : 0.f;
  R.x1 = lane + 64 < S::DIN ? (NT ? __builtin_nontemporal_load(xr + lane + 64) : xr[lane + 64]) : 0.f;
  R.sh = *reinterpret_cast<const float4*>(a.sh + e * 4);
}

#ifndef TP_FLAT_ATTR
#define TP_FLAT_ATTR
#endif
template <class S, int D, bool NT, int PROBE = 0>
__global__ __launch_bounds__(64) TP_FLAT_ATTR void tp_flat_kernel(TpFArgs a) {
  __shared__ float4 Wst[S::NV * 64];
  __shared__ __attribute__((aligned(16))) float X[128];
  __shared__ float4 RBv[8], RBs[8], RCv[8], RCs[8];
  const float* Wf = reinterpret_cast<const float*>(Wst);
  const int lane = threadIdx.x;
  const int64_t stride = gridDim.x;
  int64_t e = blockIdx.x;
  // ---- the lane's column: block, first weight of each phase, row stride, operand addresses ----
  const int blk = lane < S::O0 ? 0 : (lane < S::O0 + S::O1 ? 1 : (lane < S::O0 + S::O1 + S::O2 ? 2 : (lane < S::O0 + S::O1 + S::O2 + S::O3 ? 3 : 4)));
  const int col = lane - (blk == 0 ? 0 : (blk == 1 ? S::O0 : (blk == 2 ? S::O0 + S::O1 : S::O0 + S::O1 + S::O2)));
  const int nout = blk == 0 ? S::O0 : (blk == 1 ? S::O1 : (blk == 2 ? S::O2 : (blk == 3 ? S::O3 : 0)));
  const int bbase = blk == 0 ? S::B0 : (blk == 1 ? S::B1 : (blk == 2 ? S::B2 : S::B3));
  const int nA = blk <= 1 ? S::A : (blk <= 3 ? S::C : 0);
  const int rowA = blk <= 1 ? 0 : (blk == 2 ? S::P + S::Q : S::Q);
  const bool onA = nA > 0 && nout > 0, onB = blk <= 2 && S::P > 0 && nout > 0, onC = blk >= 1 && blk <= 3 && S::Q > 0 && nout > 0;
  const int wA = onA ? bbase + rowA * nout + col : 0, sA = onA ? nout : 0, xA = (onA && blk >= 2) ? S::XC : 0;
  const int wB = onB ? bbase + (blk == 2 ? 0 : S::A) * nout + col : 0, sB = onB ? nout : 0;
  const int wC = onC ? bbase + (blk == 1 ? S::A + S::P : (blk == 2 ? S::P : 0)) * nout + col : 0, sC = onC ? nout : 0;
  const float4* recB = blk == 0 ? RBs : RBv;
  const float4* recC = blk == 3 ? RCs : RCv;
  const int nrow = blk == 0 ? S::R0 : (blk == 1 ? S::R1 : (blk == 2 ? S::R2 : (blk == 3 ? S::R3 : 1)));
  const float rs = 1.0f / sqrtf((float)(nrow > 0 ? nrow : 1));                       // tensor_layers.py:89-92
  const int ooff = blk == 0 ? col : (blk == 1 ? S::O0 + 3 * col : (blk == 2 ? S::O0 + 3 * S::O1 + 3 * col : S::O0 + 3 * S::O1 + 3 * S::O2 + col));
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  constexpr int NA = S::A > S::C ? S::A : S::C;

  TpRow<S> R[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (e + d * stride < a.E) tpf_request<S, NT>(a, e + d * stride, lane, R[d]);
  for (;;) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (e >= a.E) return;
      if (PROBE == 2) {      // (measurement only: the read stream alone - no LDS, no arithmetic but a sum, 4 B written per edge)
        float acc = R[d].x0 + R[d].x1 + R[d].sh.x;
#pragma unroll
        for (int t = 0; t < S::NV; ++t) acc += R[d].w[t].x + R[d].w[t].y + R[d].w[t].z + R[d].w[t].w;
        if (e + (int64_t)D * stride < a.E) tpf_request<S, NT>(a, e + (int64_t)D * stride, lane, R[d]);
        if (acc == 123.456f) a.out[e * S::DOUT + lane] = acc;
        e += stride;
        continue;
      }
      // ---- park the row (flat), the node row and the per-edge operands in LDS; ask for the row D edges ahead ----
#pragma unroll
      for (int t = 0; t < S::NV; ++t) Wst[64 * t + lane] = R[d].w[t];
      X[lane] = R[d].x0;
      X[lane + 64] = R[d].x1;
      const float s0 = R[d].sh.x, vx = R[d].sh.y, vy = R[d].sh.z, vz = R[d].sh.w;
      if (e + (int64_t)D * stride < a.E) tpf_request<S, NT>(a, e + (int64_t)D * stride, lane, R[d]);
      __syncthreads();
      if (lane < S::P) {
        const float px = X[S::A + 3 * lane], py = X[S::A + 3 * lane + 1], pz = X[S::A + 3 * lane + 2];
        RBv[lane] = make_float4(px, py, pz, 0.f);
        RBs[lane] = make_float4((px * vx + py * vy + pz * vz) * inv_s3, 0.f, 0.f, 0.f);
      } else if (lane >= 32 && lane < 32 + S::Q) {
        const int m = lane - 32;
        const float qx = X[S::A + 3 * S::P + 3 * m], qy = X[S::A + 3 * S::P + 3 * m + 1], qz = X[S::A + 3 * S::P + 3 * m + 2];
        RCv[m] = make_float4(qx, qy, qz, 0.f);
        RCs[m] = make_float4((qx * vx + qy * vy + qz * vz) * inv_s3, 0.f, 0.f, 0.f);
      }
      __syncthreads();
      if (PROBE == 1) {      // (measurement only: read stream + LDS staging, no column walk)
        if (Wf[lane * 7] == 123.456f) a.out[e * S::DOUT + lane] = 1.f;
        e += stride;
        __syncthreads();
        continue;
      }
      float accA = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
      // (chunks of 8 rows: 2 broadcast reads of 4 operands + 8 column reads in flight; a fully unrolled walk parks ~100 LDS results in registers)
      const float4* X4 = reinterpret_cast<const float4*>(X) + (xA >> 2);
#pragma unroll 1
      for (int i0 = 0; i0 < NA; i0 += 8) {
        const float4 ua = X4[(i0 >> 2)], ub = X4[(i0 >> 2) + 1];
        const float uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
        const float* wp = Wf + wA + i0 * sA;
#pragma unroll
        for (int j = 0; j < 8; ++j) accA = fmaf(uu[j], wp[j * sA], accA);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < S::P; ++i) {
        const float4 u = recB[i];
        const float w = Wf[wB + i * sB];
        b0 = fmaf(u.x, w, b0); b1 = fmaf(u.y, w, b1); b2 = fmaf(u.z, w, b2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < S::Q; ++i) {
        const float4 u = recC[i];
        const float w = Wf[wC + i * sC];
        c0 = fmaf(u.x, w, c0); c1 = fmaf(u.y, w, c1); c2 = fmaf(u.z, w, c2);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!onA) accA = 0.f;
      if (!onB) { b0 = 0.f; b1 = 0.f; b2 = 0.f; }
      if (!onC) { c0 = 0.f; c1 = 0.f; c2 = 0.f; }
      float* o = a.out + e * S::DOUT + ooff;
      if (blk == 0) {
        o[0] = (s0 * accA + b0) * rs;
      } else if (blk == 1) {      // a (x) v + p s0 + (q x v)/sqrt2
        o[0] = (vx * accA + s0 * b0 + (c1 * vz - c2 * vy) * inv_s2) * rs;
        o[1] = (vy * accA + s0 * b1 + (c2 * vx - c0 * vz) * inv_s2) * rs;
        o[2] = (vz * accA + s0 * b2 + (c0 * vy - c1 * vx) * inv_s2) * rs;
      } else if (blk == 2) {      // (p x v)/sqrt2 + q s0 + c (x) v
        o[0] = ((b1 * vz - b2 * vy) * inv_s2 + s0 * c0 + vx * accA) * rs;
        o[1] = ((b2 * vx - b0 * vz) * inv_s2 + s0 * c1 + vy * accA) * rs;
        o[2] = ((b0 * vy - b1 * vx) * inv_s2 + s0 * c2 + vz * accA) * rs;
      } else if (blk == 3) {
        o[0] = (c0 + s0 * accA) * rs;
      }
      e += stride;
      __syncthreads();
    }
  }
}


// ---- the same column walk with the row operands in SGPRs (round 6, second form) ------------------------------------------------------------
// The row operand of a walk step is one value for the whole block (a_i, c_i, the components of p_i / q_i, (p_i . v)/sqrt3): the lane that loaded it
// from the node row hands it over with v_readlane_b32 and the FMA reads it as a scalar operand - no LDS traffic but the weights themselves (one
// ds_read_b32 per lane and row), no operand records, one wave-level barrier per edge.  Every lane runs the FMAs of both operand classes of a step into
// separate accumulators (a lane's weight is its own; the class that is not its block's is dropped at the end): no select, no divergence.
template <class S>
__device__ __forceinline__ float tp_xval(float x0, float x1, int k) {      // x[k] of the node row as a wave-uniform value (k is a compile-time constant)
  return k < 64 ? __builtin_amdgcn_readlane(x0, k) : __builtin_amdgcn_readlane(x1, k - 64);
}
template <class S, int D, bool NT, int WPE>
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
  static_assert(S::A + 3 * S::P + 3 * S::Q <= 64, "the vector inputs of the node row sit in the first 64 lanes");
  const int vsel = (lane >= S::A && lane < S::XC) ? (lane - S::A) % 3 : 3;

  TpRow<S> R[D];
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (e + d * stride < a.E) tpf_request<S, NT>(a, e + d * stride, lane, R[d]);
  for (;;) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (e >= a.E) return;
#pragma unroll
      for (int t = 0; t < S::NV; ++t) Wst[64 * t + lane] = R[d].w[t];
      const float x0 = R[d].x0, x1 = R[d].x1;
      const float s0 = R[d].sh.x, vx = R[d].sh.y, vy = R[d].sh.z, vz = R[d].sh.w;
      if (e + (int64_t)D * stride < a.E) tpf_request<S, NT>(a, e + (int64_t)D * stride, lane, R[d]);
      // (p_i . v)/sqrt3 in lane A + 3i, (q_i . v)/sqrt3 in lane A + 3P + 3i: three neighbouring lanes' products through the lane crossbar
      float pvq = 0.f;
      if (S::P + S::Q > 0) {
        const float t = x0 * (vsel == 0 ? vx : (vsel == 1 ? vy : (vsel == 2 ? vz : 0.f)));
        pvq = (t + __shfl_down(t, 1) + __shfl_down(t, 2)) * inv_s3;
      }
      __syncthreads();
      float accA = 0.f, accC = 0.f, bs = 0.f, bx = 0.f, by = 0.f, bz = 0.f, cs = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
      // (the row addresses advance in ONE register per phase: left to itself the compiler keeps all 36 of them, loop-invariant, in VGPRs - the opaque asm pins the chain)
      int ad = wA0;
      asm volatile("" : "+v"(ad));
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const float w = *reinterpret_cast<const float*>(Wb + ad);
        ad += sA;
        asm volatile("" : "+v"(ad));
        if (i < S::A) accA = fmaf(tp_xval<S>(x0, x1, i), w, accA);
        if (i < S::C) accC = fmaf(tp_xval<S>(x0, x1, S::XC + i), w, accC);
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // (eight rows' reads in flight, not all of them: registers)
      }
      ad = wB0;
      asm volatile("" : "+v"(ad));
#pragma unroll
      for (int i = 0; i < S::P; ++i) {
        const float w = *reinterpret_cast<const float*>(Wb + ad);
        ad += sB;
        asm volatile("" : "+v"(ad));
        bs = fmaf(__builtin_amdgcn_readlane(pvq, S::A + 3 * i), w, bs);
        bx = fmaf(tp_xval<S>(x0, x1, S::A + 3 * i), w, bx);
        by = fmaf(tp_xval<S>(x0, x1, S::A + 3 * i + 1), w, by);
        bz = fmaf(tp_xval<S>(x0, x1, S::A + 3 * i + 2), w, bz);
      }
      __builtin_amdgcn_sched_barrier(0);
      ad = wC0;
      asm volatile("" : "+v"(ad));
#pragma unroll
      for (int i = 0; i < S::Q; ++i) {
        const float w = *reinterpret_cast<const float*>(Wb + ad);
        ad += sC;
        asm volatile("" : "+v"(ad));
        cs = fmaf(__builtin_amdgcn_readlane(pvq, S::A + 3 * S::P + 3 * i), w, cs);
        cx = fmaf(tp_xval<S>(x0, x1, S::A + 3 * S::P + 3 * i), w, cx);
        cy = fmaf(tp_xval<S>(x0, x1, S::A + 3 * S::P + 3 * i + 1), w, cy);
        cz = fmaf(tp_xval<S>(x0, x1, S::A + 3 * S::P + 3 * i + 2), w, cz);
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

template <class S>
static hipError_t launch_tp_flat(const TpFArgs& a, int variant, int64_t blocks, hipStream_t s) {
  const dim3 g((unsigned)blocks), b(64);
  switch (variant) {
    case 1: hipLaunchKernelGGL((tp_flat_kernel<S, 2, false>), g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL((tp_flat_kernel<S, 2, true>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((tp_flat_kernel<S, 3, false>), g, b, 0, s, a); break;
    case 4: hipLaunchKernelGGL((tp_flat_kernel<S, 3, true>), g, b, 0, s, a); break;
    case 5: hipLaunchKernelGGL((tp_flat_kernel<S, 1, false>), g, b, 0, s, a); break;
    case 7: hipLaunchKernelGGL((tp_flat_kernel<S, 2, false, 1>), g, b, 0, s, a); break;
    case 8: hipLaunchKernelGGL((tp_flat_kernel<S, 2, false, 2>), g, b, 0, s, a); break;
    case 9: hipLaunchKernelGGL((tp_flat_kernel<S, 4, false, 2>), g, b, 0, s, a); break;
    case 11: hipLaunchKernelGGL((tp_col_kernel<S, 2, false, 4>), g, b, 0, s, a); break;
    case 12: hipLaunchKernelGGL((tp_col_kernel<S, 2, false, 3>), g, b, 0, s, a); break;
    case 13: hipLaunchKernelGGL((tp_col_kernel<S, 1, false, 4>), g, b, 0, s, a); break;
    case 14: hipLaunchKernelGGL((tp_col_kernel<S, 1, false, 5>), g, b, 0, s, a); break;
    case 15: hipLaunchKernelGGL((tp_col_kernel<S, 3, false, 3>), g, b, 0, s, a); break;
    case 16: hipLaunchKernelGGL((tp_col_kernel<S, 2, false, 2>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((tp_flat_kernel<S, 1, true>), g, b, 0, s, a); break;
  }
  return hipGetLastError();
}

hipError_t launch_tp_forward(const ConvLayerDev& L, const float* x_dst, const float* sh, const float* w, int64_t E,
                             float* out, hipStream_t s) {
  if (E == 0) return hipSuccess;
  TpKArgs k;
  k.x = x_dst; k.sh = sh; k.w = w; k.out = out; k.E = E; k.din = L.din; k.dout = L.dout; k.W = L.W;
  int off = 0;
  const int dims[4] = {1, 3, 3, 1};
  for (int b = 0; b < 4; ++b) {
    k.n_in[b] = L.n_in[b]; k.n_out[b] = L.n_out[b]; k.blk_off[b] = L.blk_off[b]; k.in_mul[b] = L.in_mul[b];
    k.out_off[b] = off;
    off += L.out_mul[b] * dims[b];
  }
  static const int variant = getenv("DDK_TP_VARIANT") ? atoi(getenv("DDK_TP_VARIANT")) : 1;
  static const int grid_env = getenv("DDK_TP_GRID") ? atoi(getenv("DDK_TP_GRID")) : 0;
  if (variant > 0) {
    TpFArgs a{x_dst, sh, w, out, E};
    const int64_t cap = grid_env > 0 ? grid_env : 256 * 16;
    const int64_t nb = E < cap ? E : cap;
    const int im[4] = {L.in_mul[0], L.in_mul[1], L.in_mul[2], L.in_mul[3]}, om[4] = {L.out_mul[0], L.out_mul[1], L.out_mul[2], L.out_mul[3]};
    auto is = [&](int a0, int a1, int a2, int a3, int o0, int o1, int o2, int o3) {
      return im[0] == a0 && im[1] == a1 && im[2] == a2 && im[3] == a3 && om[0] == o0 && om[1] == o1 && om[2] == o2 && om[3] == o3;
    };
    if (is(24, 6, 6, 24, 24, 6, 6, 24)) return launch_tp_flat<TpShape<24, 6, 6, 24, 24, 6, 6, 24>>(a, variant, nb, s);
    if (is(24, 6, 6, 0, 24, 6, 6, 24)) return launch_tp_flat<TpShape<24, 6, 6, 0, 24, 6, 6, 24>>(a, variant, nb, s);
    if (is(24, 6, 0, 0, 24, 6, 6, 0)) return launch_tp_flat<TpShape<24, 6, 0, 0, 24, 6, 6, 0>>(a, variant, nb, s);
    if (is(24, 0, 0, 0, 24, 6, 0, 0)) return launch_tp_flat<TpShape<24, 0, 0, 0, 24, 6, 0, 0>>(a, variant, nb, s);
    return hipErrorInvalidValue;
  }
  TpSArgs S;
  S.k = k;
  for (int b = 0; b < 4; ++b) S.blk[b] = TpBlk{0, 1, 64, 0};      // (geometry is compile time: TpGeo)
  int64_t blocks = E < 256 * 32 ? E : 256 * 32;      // one wave per workgroup; 8 resident per CU, each an edge in hand and one in flight
  // the block shapes of the score model's conv layers (tensor_layers.py:12-27: out = 24x0e + 6x1o [+ 6x1e [+ 24x0o]]) at compile time.  The float4 / float2
  // reads need dword alignment only (global memory, unaligned access mode: tests/test_gpu_round5.py reads a row that starts 4 B behind a 16-B boundary)
  const int no[4] = {k.n_out[0], k.n_out[1], k.n_out[2], k.n_out[3]};
  for (int b = 0; b < 4; ++b)
    if (no[b] > 0 && (k.n_in[b] + (64 / (no[b] == 24 ? 6 : 3)) - 1) / (64 / (no[b] == 24 ? 6 : 3)) > TP_ITS) return hipErrorInvalidValue;
  if (no[0] == 24 && no[1] == 6 && no[2] == 6 && no[3] == 24) hipLaunchKernelGGL((tp_stream_kernel<24, 6, 6, 24>), dim3((unsigned)blocks), dim3(64), 0, s, S);
  else if (no[0] == 24 && no[1] == 6 && no[2] == 6 && no[3] == 0) hipLaunchKernelGGL((tp_stream_kernel<24, 6, 6, 0>), dim3((unsigned)blocks), dim3(64), 0, s, S);
  else if (no[0] == 24 && no[1] == 6 && no[2] == 0 && no[3] == 0) hipLaunchKernelGGL((tp_stream_kernel<24, 6, 0, 0>), dim3((unsigned)blocks), dim3(64), 0, s, S);
  else return hipErrorInvalidValue;      // not a FasterTensorProduct of this model family (ddk_create refuses other ns / nv)
  return hipGetLastError();
}

}  // namespace ddk
