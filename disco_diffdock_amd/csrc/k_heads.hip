// Output heads of the score model (reference models/score_model.py:269-307, 410-438):
//   * centre convolution (final_conv, e3nn FullyConnectedTensorProduct 84 x (0e+1o) -> 2x1o + 2x1e with per-edge
//     weights from a 48->48->144 MLP), mean over the ligand atoms of a graph, BatchNorm, tr/rot magnitude MLPs;
//   * torsion convolution (tor_bond_conv: FCTP 84 x [(0e+1o) (x) 2e] -> 24x0o + 24x0e, 72->72->288 MLP) around
//     every rotatable bond, mean over the bond's <=32 neighbour atoms, BatchNorm, tor_final_layer.
// Tiny work (B*n_lig resp. B*R*~20 edges): one thread per edge with wave-uniform weights through the scalar cache.
// The Clebsch-Gordan constants below are e3nn's real-basis wigner 3j (restated in oracle/e3nn_lite.py):
//   w3j(0,1,1) = w3j(1,0,1) = delta/sqrt3,  w3j(1,1,1) = eps_ijk/sqrt6,  w3j(1,1,0) = delta/sqrt3,  w3j(1,2,1) below.
#include "model.h"

namespace ddk {

// dot product of a 16-B aligned global weight row with an LDS activation row: all N/4 128-bit loads of the row are issued
// before the first use (the heads are L2-latency bound: few threads, long dependent chains), four independent accumulators
template <int N>
__device__ __forceinline__ float row_dot(const float* __restrict__ w, const float* act) {
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 r[N / 4];
#pragma unroll
  for (int i = 0; i < N / 4; ++i) r[i] = w4[i];
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    a0 = fmaf(r[i].x, act[4 * i + 0], a0); a1 = fmaf(r[i].y, act[4 * i + 1], a1);
    a2 = fmaf(r[i].z, act[4 * i + 2], a2); a3 = fmaf(r[i].w, act[4 * i + 3], a3);
  }
  __builtin_amdgcn_sched_barrier(0);   // one row in flight at a time: keeps the register count of the unrolled callers in check
  return (a0 + a1) + (a2 + a3);
}


__device__ __forceinline__ void smear(float d, const EdgeMlpDev& m, float* gs) {
#pragma unroll
  for (int k = 0; k < DE; ++k) {
    const float t = d - m.offset[k];
    gs[k] = expf(m.coeff * (t * t));
  }
}

// out[NS] = W2 . relu(W1d . gs + b1) + b2
__device__ __forceinline__ void edge_mlp(const EdgeMlpDev& m, const float* b1, const float* gs, float* out) {
  float h[NS];
#pragma unroll
  for (int o = 0; o < NS; ++o) {
    float a = b1[o];
#pragma unroll
    for (int k = 0; k < DE; ++k) a += m.w1d[o * DE + k] * gs[k];
    h[o] = fmaxf(a, 0.0f);
  }
#pragma unroll
  for (int o = 0; o < NS; ++o) {
    float a = m.b2[o];
#pragma unroll
    for (int k = 0; k < NS; ++k) a += m.w2[o * NS + k] * h[k];
    out[o] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Both heads use 8 threads per edge (lanes 8e..8e+7 of a 256-thread workgroup own one edge): the per-edge MLPs
// are split by output row so that a thread does ~1/8 of the multiply-adds, activations are exchanged through LDS.
// ---------------------------------------------------------------------------------------------------------
constexpr int EPB = 32;   // edges per 256-thread workgroup

__device__ void center_head_block(const HeadArgs& A, int b) {
  __shared__ float lp[MAX_LIG * 3];
  __shared__ float act[EPB][2 * NS + 1];     // per-edge activations (h / in48 / h2), padded against bank conflicts
  __shared__ float tot[12];
  __shared__ float ctr[3];
  const int tid = threadIdx.x, n = A.n_lig;
  const int el = tid >> 3, p = tid & 7;
  for (int i = tid; i < n * 3; i += 256) lp[i] = A.lig_pos[(size_t)b * n * 3 + i];
  if (tid < 12) tot[tid] = 0.0f;
  __syncthreads();
  if (tid < 3) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += lp[3 * i + tid];
    ctr[tid] = s / (float)n;
  }
  __syncthreads();
  float part12[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) part12[k] = 0.0f;
  for (int a0 = 0; a0 < n; a0 += EPB) {
    const int i = a0 + el;
    const bool live = i < n;
    const int ii = live ? i : n - 1;
    const float vx = lp[3 * ii] - ctr[0], vy = lp[3 * ii + 1] - ctr[1], vz = lp[3 * ii + 2] - ctr[2];
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
    const float s1[3] = {vx * inv, vy * inv, vz * inv};
    const float* xr = A.x + ((size_t)b * n + ii) * XW;
    float gs[DE];
    smear(d, A.md.center_edge, gs);
    // center_edge_embedding layer 1: 3 of the 24 hidden units per thread
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int o = 3 * p + q;
      float a = A.sp.center_edge_sigb[o];
#pragma unroll
      for (int k = 0; k < DE; ++k) a += A.md.center_edge.w1d[o * DE + k] * gs[k];
      act[el][o] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    float e3[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int o = 3 * p + q;
      float a = A.md.center_edge.b2[o];
#pragma unroll
      for (int k = 0; k < NS; ++k) a += A.md.center_edge.w2[o * NS + k] * act[el][k];
      e3[q] = a;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      act[el][3 * p + q] = e3[q];                 // in48 = [edge embedding | x_atom[:ns]]
      act[el][NS + 3 * p + q] = xr[3 * p + q];
    }
    __syncthreads();
    float h6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int o = 6 * p + q;
      const float a = A.md.fc_b0[o] + row_dot<2 * NS>(A.md.fc_w0 + o * 2 * NS, act[el]);
      h6[q] = fmaxf(a, 0.0f);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 6; ++q) act[el][6 * p + q] = h6[q];
    __syncthreads();
    // 144 per-edge weights, rows p, p+8, ...: accumulate the six FCTP paths (see the weight layout above)
    float sA = 0.f, sF = 0.f, vB[3] = {}, vC[3] = {}, vD[3] = {}, vE[3] = {};
    auto wrow = [&](int row) {
      return A.md.fc_b4[row] + row_dot<2 * NS>(A.md.fc_w4 + row * 2 * NS, act[el]);
    };
    // rows of a path are dealt round-robin to the 8 threads of the edge; every row of a thread has w = row & 1 = p & 1
#pragma unroll 1
    for (int it = 0; it < 6; ++it) {          // A: 0e (x) 1o -> 1o, rows [0,48)
      const int r = p + 8 * it;
      sA += wrow(r) * xr[r >> 1];
    }
#pragma unroll 1
    for (int it = 0; it < 6; ++it) {          // F: 0o (x) 1o -> 1e, rows [96,144)
      const int r = p + 8 * it;
      sF += wrow(96 + r) * xr[OFF_C + (r >> 1)];
    }
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {          // B, C (1o features), D, E (1e features): 12 rows each
      const int r = p + 8 * it;
      if (r < 12) {
        const int uu = r >> 1;
        const float wb = wrow(48 + r), wc = wrow(60 + r), wd = wrow(72 + r), we = wrow(84 + r);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float pk = xr[OFF_P + 3 * uu + k], qk = xr[OFF_Q + 3 * uu + k];
          vB[k] += wb * pk; vC[k] += wc * pk; vD[k] += wd * qk; vE[k] += we * qk;
        }
      }
    }
    const float cS = 0.28867513459481288f * 0.57735026918962576f;   // 1/sqrt12 * 1/sqrt3
    const float cX = 0.28867513459481288f * 0.40824829046386302f;   // 1/sqrt12 * 1/sqrt6
    if (live) {
      const float cEx = vE[1] * s1[2] - vE[2] * s1[1], cEy = vE[2] * s1[0] - vE[0] * s1[2], cEz = vE[0] * s1[1] - vE[1] * s1[0];
      const float cCx = vC[1] * s1[2] - vC[2] * s1[1], cCy = vC[2] * s1[0] - vC[0] * s1[2], cCz = vC[0] * s1[1] - vC[1] * s1[0];
      const float o1[3] = {cS * (sA * s1[0] + vB[0]) + cX * cEx, cS * (sA * s1[1] + vB[1]) + cX * cEy, cS * (sA * s1[2] + vB[2]) + cX * cEz};
      const float e1[3] = {cS * (sF * s1[0] + vD[0]) + cX * cCx, cS * (sF * s1[1] + vD[1]) + cX * cCy, cS * (sF * s1[2] + vD[2]) + cX * cCz};
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const float m = ((p & 1) == w) ? 1.0f : 0.0f;      // this thread's rows all belong to output multiplicity p & 1
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          part12[3 * w + k] += m * o1[k];
          part12[6 + 3 * w + k] += m * e1[k];
        }
      }
    }
    __syncthreads();
  }
  // (everything is linear in the per-row partial sums, so summing the 8 threads of an edge and the edges is one reduction)
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    float v = part12[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) atomicAdd(&tot[k], v);
  }
  __syncthreads();
  __shared__ float g12[12];
  if (tid < 12) g12[tid] = (tot[tid] / (float)n) * A.md.fc_bn_scale[tid / 3];
  __syncthreads();
  if (tid < 2) {   // tid 0: translation, tid 1: rotation  (score_model.py:274-286)
    const int o = 3 * tid;
    const float px = g12[o] + g12[6 + o], py = g12[o + 1] + g12[7 + o], pz = g12[o + 2] + g12[8 + o];
    const float nrm = sqrtf(px * px + py * py + pz * pz);
    const float* w0n = tid == 0 ? A.md.tr_w0n : A.md.rot_w0n;
    const float* w3 = tid == 0 ? A.md.tr_w3 : A.md.rot_w3;
    const float* sb = tid == 0 ? A.sp.tr_sigb : A.sp.rot_sigb;
    float s = tid == 0 ? A.md.tr_b3 : A.md.rot_b3;
    for (int k = 0; k < NS; ++k) s += w3[k] * fmaxf(w0n[k] * nrm + sb[k], 0.0f);
    float f = s / nrm;
    if (A.scale_by_sigma) f = tid == 0 ? f / A.sp.tr_sigma : f * A.sp.so3_norm;
    float* out = (tid == 0 ? A.tr_out : A.rot_out) + 3 * (size_t)b;
    out[0] = px * f; out[1] = py * f; out[2] = pz * f;
  }
}

__device__ void torsion_head_block(const HeadArgs& A, int b, int r) {
  __shared__ float lp[MAX_LIG * 3];
  __shared__ int nb[BOND_CAP];
  __shared__ float act[EPB][NE + 1];
  __shared__ float outp[EPB][2 * NS + 1];
  __shared__ float v48[2 * NS];
  __shared__ int n_nb;
  const int tid = threadIdx.x, n = A.n_lig;
  const int el = tid >> 3, p = tid & 7;
  for (int i = tid; i < n * 3; i += 256) lp[i] = A.lig_pos[(size_t)b * n * 3 + i];
  __syncthreads();
  const int u = A.rot_u[r], v = A.rot_v[r];
  const float cx = (lp[3 * u] + lp[3 * v]) * 0.5f, cy = (lp[3 * u + 1] + lp[3 * v + 1]) * 0.5f, cz = (lp[3 * u + 2] + lp[3 * v + 2]) * 0.5f;
  if (tid < 64) {   // neighbour atoms of the bond centre: ascending index, first BOND_CAP (score_model.py:430)
    int cnt = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
      const int k = k0 + tid;
      bool in = false;
      if (k < n) {
        const float dx = lp[3 * k] - cx, dy = lp[3 * k + 1] - cy, dz = lp[3 * k + 2] - cz;
        in = dx * dx + dy * dy + dz * dz < A.lig_r2;
      }
      const unsigned long long mask = __ballot(in);
      const int rank = cnt + __popcll(mask & ((1ull << tid) - 1ull));
      if (in && rank < BOND_CAP) nb[rank] = k;
      cnt += __popcll(mask);
    }
    if (tid == 0) n_nb = cnt < BOND_CAP ? cnt : BOND_CAP;
  }
  __syncthreads();
  const int ne = n_nb;
  const bool live = el < ne;
  const int k = ne > 0 ? nb[live ? el : 0] : u;
  const float vx = lp[3 * k] - cx, vy = lp[3 * k + 1] - cy, vz = lp[3 * k + 2] - cz;
  const float d = sqrtf(vx * vx + vy * vy + vz * vz);
  const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
  const float s1[3] = {vx * inv, vy * inv, vz * inv};
  float bx = lp[3 * v] - lp[3 * u], by = lp[3 * v + 1] - lp[3 * u + 1], bz = lp[3 * v + 2] - lp[3 * u + 2];
  const float bn = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-12f);
  bx /= bn; by /= bn; bz /= bn;
  const float s3 = 1.7320508075688772f, s5 = 2.2360679774997897f;
  const float y0 = s5 * s3 * bx * bz, y1 = s5 * s3 * bx * by, y2 = s5 * (by * by - 0.5f * (bx * bx + bz * bz)),
              y3 = s5 * s3 * by * bz, y4 = s5 * (s3 * 0.5f) * (bz * bz - bx * bx);
  const float ca = 0.31622776601683794f, cb = 0.18257418583505536f;   // 1/sqrt10, 1/sqrt30
  const float T0 = s3 * (-cb * s1[0] * y2 - ca * s1[0] * y4 + ca * s1[1] * y1 + ca * s1[2] * y0);
  const float T1 = s3 * (ca * s1[0] * y1 + 2.0f * cb * s1[1] * y2 + ca * s1[2] * y3);
  const float T2 = s3 * (ca * s1[0] * y0 + ca * s1[1] * y3 - cb * s1[2] * y2 + ca * s1[2] * y4);
  const float* xk = A.x + ((size_t)b * n + k) * XW;
  const float* xu = A.x + ((size_t)b * n + u) * XW;
  const float* xv = A.x + ((size_t)b * n + v) * XW;
  float gs[DE];
  smear(d, A.md.final_edge, gs);
  // final_edge_embedding: 3 of 24 outputs per thread in both layers
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int o = 3 * p + q;
    float a = A.md.final_edge_b1[o];
#pragma unroll
    for (int j = 0; j < DE; ++j) a += A.md.final_edge.w1d[o * DE + j] * gs[j];
    act[el][o] = fmaxf(a, 0.0f);
  }
  __syncthreads();
  float e3[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int o = 3 * p + q;
    float a = A.md.final_edge.b2[o];
#pragma unroll
    for (int j = 0; j < NS; ++j) a += A.md.final_edge.w2[o * NS + j] * act[el][j];
    e3[q] = a;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int o = 3 * p + q;
    act[el][o] = e3[q];
    act[el][NS + o] = xk[o];
    act[el][2 * NS + o] = xu[o] + xv[o];
  }
  __syncthreads();
  float h9[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const int o = 9 * p + q;
    const float a = A.md.tb_b0[o] + row_dot<NE>(A.md.tb_w0 + o * NE, act[el]);
    h9[q] = fmaxf(a, 0.0f);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 9; ++q) act[el][9 * p + q] = h9[q];
  __syncthreads();
  // paths: [1o (x) 1o -> 0e : [6][24]] then [1e (x) 1o -> 0o : [6][24]]; output irreps 24x0o + 24x0e.
  // thread p owns output multiplicities w = 3p..3p+2 of both paths (private accumulators, no atomics)
  const float c = 0.40824829046386302f * 0.57735026918962576f;   // sqrt(1/6) * 1/sqrt3
#pragma unroll 1
  for (int path = 0; path < 2; ++path) {
    const int xo = path == 0 ? OFF_P : OFF_Q;
    float o3[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int uu = 0; uu < NV; ++uu) {
      const float dt = (xk[xo + 3 * uu] * T0 + xk[xo + 3 * uu + 1] * T1 + xk[xo + 3 * uu + 2] * T2) * c;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int row = path * (NV * NS) + uu * NS + 3 * p + q;
        const float a = A.md.tb_b4[row] + row_dot<NE>(A.md.tb_w4 + row * NE, act[el]);
        o3[q] += a * dt;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) outp[el][(path == 0 ? NS : 0) + 3 * p + q] = o3[q];
  }
  __syncthreads();
  if (tid < 2 * NS) {
    float s = 0.0f;
    for (int e = 0; e < ne; ++e) s += outp[e][tid];
    s = s / (float)(ne > 1 ? ne : 1);
    v48[tid] = (s - A.md.tb_bn_mean[tid]) * A.md.tb_bn_scale[tid] + A.md.tb_bn_bias[tid];
  }
  __syncthreads();
  if (tid < NS) {   // tor_final_layer: Linear(48,24,no bias) -> tanh -> Linear(24,1,no bias)
    float a = 0.0f;
    for (int j = 0; j < 2 * NS; ++j) a += A.md.tf_w0[tid * 2 * NS + j] * v48[j];
    outp[0][tid] = A.md.tf_w3[tid] * tanhf(a);
  }
  __syncthreads();
  if (tid == 0) {
    float o = 0.0f;
    for (int j = 0; j < NS; ++j) o += outp[0][j];
    if (A.scale_by_sigma) o *= A.sp.torus_norm_sqrt;
    A.tor_out[(size_t)b * A.R + r] = o;
  }
}

// one launch for both heads: blocks [0, B*R) = rotatable bonds, blocks [B*R, B*R + B) = graph centres
__global__ __launch_bounds__(256, 2) void heads_kernel(HeadArgs A, int n_tor_blocks) {
  const int blk = blockIdx.x;
  if (blk < n_tor_blocks) torsion_head_block(A, blk / A.R, blk % A.R);
  if (blk >= n_tor_blocks) center_head_block(A, blk - n_tor_blocks);
}

hipError_t launch_heads(const HeadArgs& A, bool torsion, hipStream_t s) {
  const int n_tor = (torsion && A.R > 0) ? A.B * A.R : 0;
  hipLaunchKernelGGL(heads_kernel, dim3(n_tor + A.B), dim3(256), 0, s, A, n_tor);
  return hipGetLastError();
}

}  // namespace ddk
