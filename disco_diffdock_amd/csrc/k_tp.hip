// FasterTensorProduct.forward at the reference's op boundary (models/tensor_layers.py:65-116):
//   out[e] = TP(x_dst[e], sh[e], w[e])  with the per-edge weights w [E, W] resident in HBM.
// HBM-bound by construction (W*4 B of weights per edge for ~5 kFLOP): one wave per edge streams the weight row with vector loads
// (tp_stream_kernel below); the <=276 row operands u_i live in an LDS table per wave.
// Used by the drop-in FasterTensorProduct module and as an on-GPU cross-check of the fused kernel.
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
