// Output heads of the score model (reference models/score_model.py:269-307, 410-438):
//   * centre convolution (final_conv, e3nn FullyConnectedTensorProduct 84 x (0e+1o) -> 2x1o + 2x1e with per-edge
//     weights from a 48->48->144 MLP), mean over the ligand atoms of a graph, BatchNorm, tr/rot magnitude MLPs;
//   * torsion convolution (tor_bond_conv: FCTP 84 x [(0e+1o) (x) 2e] -> 24x0o + 24x0e, 72->72->288 MLP) around
//     every rotatable bond, mean over the bond's <=32 neighbour atoms, BatchNorm, tor_final_layer.
// Both are tensor-product convolutions and run through the fused conv kernel (k_conv.hip, explicit edge attributes) with their own
// tile tables (ddk_capi.hip: build_head_layer): tor_bond_conv keeps only the two dot-product parts with the 1o block T of
// sh (x) sh_2e(bond axis) in the place of the edge's sh[1:4] (W = 288: 12 tiles), final_conv is the two vector blocks with two output
// channels each (W = 144, MLP width 48 zero padded to 72).  Here: the kernel that builds both edge sets with their 72-wide edge attributes
// (heads_pre_kernel) and the one that finishes the two outputs (heads_post_kernel).  A launch with a few thousand edges is spread over
// the CUs by the fused kernel's column split of short work queues.
// The Clebsch-Gordan constants are e3nn's real-basis wigner 3j (restated in oracle/e3nn_lite.py):
//   w3j(0,1,1) = w3j(1,0,1) = delta/sqrt3,  w3j(1,1,1) = eps_ijk/sqrt6,  w3j(1,1,0) = delta/sqrt3,  w3j(1,2,1) below.
#include "model.h"

namespace ddk {

// 2-layer edge-embedding MLP of one edge by the 8 threads that own it (thread p: Gaussians 4p..4p+3, hidden units and outputs 3p..3p+2;
// activations exchanged through the edge's LDS row).  All 256 threads call it (barriers inside); returns this thread's three outputs.
constexpr int EPB = 32;   // edges per 256-thread workgroup
__device__ __forceinline__ void edge_embed8(const EdgeMlpDev& m, const float* b1, float d, float (*act)[DE + 1], int el, int p, float* e3) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float t = d - m.offset[4 * p + q];
    act[el][4 * p + q] = expf(m.coeff * (t * t));
  }
  __syncthreads();
  float h3[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int o = 3 * p + q;
    float a = b1[o];
#pragma unroll
    for (int k = 0; k < DE; ++k) a += m.w1d[o * DE + k] * act[el][k];
    h3[q] = fmaxf(a, 0.0f);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) act[el][3 * p + q] = h3[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int o = 3 * p + q;
    float a = m.b2[o];
#pragma unroll
    for (int k = 0; k < NS; ++k) a += m.w2[o * NS + k] * act[el][k];
    e3[q] = a;
  }
  __syncthreads();
}

// blocks [0, B*R): one rotatable bond each (its <= 32 neighbour atoms); blocks [B*R, B*R + B): the centre edges of one graph, 32 atoms per
// pass; 8 threads per edge.  Edge e: src = row of the head's accumulator (graph / bond), dst = ligand node, attr [72], sh [4].
__global__ __launch_bounds__(256) void heads_pre_kernel(HeadArgs A, int n_tor_blocks) {
  __shared__ float lp[MAX_LIG * 3];
  __shared__ float act[EPB][DE + 1];
  __shared__ int nb[BOND_CAP];
  __shared__ int s_cnt, s_base;
  __shared__ float ctr[3];
  const int tid = threadIdx.x, n = A.n_lig;
  const int el = tid >> 3, p = tid & 7;
  const bool tor = (int)blockIdx.x < n_tor_blocks;
  const int b = tor ? blockIdx.x / A.R : blockIdx.x - n_tor_blocks;
  for (int i = tid; i < n * 3; i += 256) lp[i] = A.lig_pos[(size_t)b * n * 3 + i];
  __syncthreads();
  if (!tor) {
    // ---- centre graph: edge (graph b <- atom i), vec = pos_i - centroid (score_model.py:410-423) ----
    if (tid < 3) {
      float s = 0.0f;
      for (int i = 0; i < n; ++i) s += lp[3 * i + tid];
      ctr[tid] = s / (float)n;
    }
    __syncthreads();
    for (int a0 = 0; a0 < n; a0 += EPB) {
      const int i = a0 + el;
      const bool live = i < n;
      const int ii = live ? i : n - 1;
      const float vx = lp[3 * ii] - ctr[0], vy = lp[3 * ii + 1] - ctr[1], vz = lp[3 * ii + 2] - ctr[2];
      const float d = sqrtf(vx * vx + vy * vy + vz * vz);
      float e3[3];
      edge_embed8(A.md.center_edge, A.sp.center_edge_sigb, d, act, el, p, e3);
      if (live) {
        const int e = b * n + i;
        float* at = A.h_attr + (size_t)e * NE;
        const float* xr = A.x + ((size_t)b * n + i) * XW;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          at[3 * p + q] = e3[q];
          at[NS + 3 * p + q] = xr[3 * p + q];
          at[2 * NS + 3 * p + q] = 0.0f;           // the MLP is 48 wide: zero padded to the kernel's 72
        }
        if (p == 0) {
          const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
          *reinterpret_cast<float4*>(A.h_sh + (size_t)e * 4) = make_float4(1.0f, vx * inv, vy * inv, vz * inv);
          A.h_src[e] = b;
          A.h_dst[e] = b * n + i;
        }
      }
    }
    return;
  }
  // ---- bond graph: neighbours of the bond centre (ascending index, first BOND_CAP: score_model.py:430), edge (bond <- atom k) ----
  const int r = blockIdx.x - b * A.R;
  const int u = A.rot_u[r], v = A.rot_v[r];
  const float cx = (lp[3 * u] + lp[3 * v]) * 0.5f, cy = (lp[3 * u + 1] + lp[3 * v + 1]) * 0.5f, cz = (lp[3 * u + 2] + lp[3 * v + 2]) * 0.5f;
  if (tid < 64) {
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
    if (tid == 0) {
      s_cnt = cnt < BOND_CAP ? cnt : BOND_CAP;
      if (A.deterministic) {      // fixed ranges of BOND_CAP edges per bond (the unused tail becomes null edges): the same tiles every run
        s_base = A.B * n + (int)blockIdx.x * BOND_CAP;
        if (blockIdx.x == 0) A.h_info[3] = A.B * n + n_tor_blocks * BOND_CAP;
      } else {
        s_base = atomicAdd(A.h_info + 3, s_cnt);     // this bond's contiguous edge range (the order of the bonds in the list is irrelevant)
      }
      A.h_deg[blockIdx.x] = s_cnt;
    }
  }
  __syncthreads();
  const int ne = s_cnt;
  const bool live = el < ne;
  const int k = ne > 0 ? nb[live ? el : 0] : u;
  const float vx = lp[3 * k] - cx, vy = lp[3 * k + 1] - cy, vz = lp[3 * k + 2] - cz;
  const float d = sqrtf(vx * vx + vy * vy + vz * vz);
  float e3[3];
  edge_embed8(A.md.final_edge, A.md.final_edge_b1, d, act, el, p, e3);
  const int e = s_base + el;
  float* at = A.h_attr + (size_t)e * NE;
  if (!live) {
    if (A.deterministic) {      // null edge: zero attributes and sh, accumulates into the scratch row behind the bonds' rows
#pragma unroll
      for (int q = 0; q < 9; ++q) at[9 * p + q] = 0.0f;
      if (p == 0) {
        *reinterpret_cast<float4*>(A.h_sh + (size_t)e * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        A.h_src[e] = A.B + n_tor_blocks;
        A.h_dst[e] = b * n + u;
      }
    }
    return;
  }
  const float* xk = A.x + ((size_t)b * n + k) * XW;
  const float* xu = A.x + ((size_t)b * n + u) * XW;
  const float* xv = A.x + ((size_t)b * n + v) * XW;
#pragma unroll
  for (int q = 0; q < 3; ++q) {     // [edge embedding | x_atom[:ns] | (x_u + x_v)[:ns]]  (score_model.py:299-300)
    const int o = 3 * p + q;
    at[o] = e3[q];
    at[NS + o] = xk[o];
    at[2 * NS + o] = xu[o] + xv[o];
  }
  if (p != 0) return;
  const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
  const float s1[3] = {vx * inv, vy * inv, vz * inv};
  float bx = lp[3 * v] - lp[3 * u], by = lp[3 * v + 1] - lp[3 * u + 1], bz = lp[3 * v + 2] - lp[3 * u + 2];
  const float bn = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-12f);
  bx /= bn; by /= bn; bz /= bn;
  // sh_2e of the bond axis (component normalised) and the 1o block of FullTensorProduct(sh, sh_2e): sqrt3 * sum w3j(1,2,1) s1 y
  const float s3 = 1.7320508075688772f, s5 = 2.2360679774997897f;
  const float y0 = s5 * s3 * bx * bz, y1 = s5 * s3 * bx * by, y2 = s5 * (by * by - 0.5f * (bx * bx + bz * bz)),
              y3 = s5 * s3 * by * bz, y4 = s5 * (s3 * 0.5f) * (bz * bz - bx * bx);
  const float ca = 0.31622776601683794f, cb = 0.18257418583505536f;   // 1/sqrt10, 1/sqrt30
  const float T0 = s3 * (-cb * s1[0] * y2 - ca * s1[0] * y4 + ca * s1[1] * y1 + ca * s1[2] * y0);
  const float T1 = s3 * (ca * s1[0] * y1 + 2.0f * cb * s1[1] * y2 + ca * s1[2] * y3);
  const float T2 = s3 * (ca * s1[0] * y0 + ca * s1[1] * y3 - cb * s1[2] * y2 + ca * s1[2] * y4);
  *reinterpret_cast<float4*>(A.h_sh + (size_t)e * 4) = make_float4(1.0f, T0, T1, T2);   // v := T for the kernel's (p.v)/sqrt3, (q.v)/sqrt3 rows
  A.h_src[e] = A.B + (int)blockIdx.x;          // accumulator rows: [B graphs | B*R bonds]
  A.h_dst[e] = b * n + k;
}

// blocks [0, B*R): tor_bond_conv's mean / BatchNorm / tor_final_layer (score_model.py:302-307); blocks [B*R, B*R + B): final_conv's mean /
// BatchNorm and the tr / rot magnitude MLPs (:272-286).  Every accumulator that is read is cleared behind the read.
__global__ __launch_bounds__(64) void heads_post_kernel(HeadArgs A, int n_tor_blocks) {
  __shared__ float v48[2 * NS], hid[NS], g12[12];
  const int tid = threadIdx.x;
  if (A.prof_out != nullptr && blockIdx.x == gridDim.x - 1 && tid < PROF_INTS) A.prof_out[tid] = A.exec_info[tid];   // profile mode: edge counts of this forward -> pinned host slot
  if ((int)blockIdx.x < n_tor_blocks) {
    float* row = A.h_sum + (size_t)(A.B + blockIdx.x) * XW;
    const int ne = A.h_deg[blockIdx.x];
    if (tid < 2 * NS) {
      const float s = row[tid] / (float)(ne > 1 ? ne : 1);
      row[tid] = 0.0f;
      v48[tid] = (s - A.md.tb_bn_mean[tid]) * A.md.tb_bn_scale[tid] + A.md.tb_bn_bias[tid];
    }
    __syncthreads();
    if (tid < NS) {   // tor_final_layer: Linear(48,24,no bias) -> tanh -> Linear(24,1,no bias)
      float a = 0.0f;
      for (int j = 0; j < 2 * NS; ++j) a += A.md.tf_w0[tid * 2 * NS + j] * v48[j];
      hid[tid] = A.md.tf_w3[tid] * tanhf(a);
    }
    __syncthreads();
    if (tid == 0) {
      float o = 0.0f;
      for (int j = 0; j < NS; ++j) o += hid[j];
      if (A.scale_by_sigma) o *= A.sp.torus_norm_sqrt;
      A.tor_out[blockIdx.x] = o;          // [b * R + r]
    }
    return;
  }
  const int b = blockIdx.x - n_tor_blocks;
  float* row = A.h_sum + (size_t)b * XW;
  if (tid < 12) {
    g12[tid] = (row[tid] / (float)A.n_lig) * A.md.fc_bn_scale[tid / 3];
    row[tid] = 0.0f;
  }
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

hipError_t launch_heads_pre(const HeadArgs& A, bool torsion, hipStream_t s) {
  const int n_tor = (torsion && A.R > 0) ? A.B * A.R : 0;
  hipLaunchKernelGGL(heads_pre_kernel, dim3(n_tor + A.B), dim3(256), 0, s, A, n_tor);
  return hipGetLastError();
}

hipError_t launch_heads_post(const HeadArgs& A, bool torsion, hipStream_t s) {
  const int n_tor = (torsion && A.R > 0) ? A.B * A.R : 0;
  hipLaunchKernelGGL(heads_post_kernel, dim3(n_tor + A.B), dim3(64), 0, s, A, n_tor);
  return hipGetLastError();
}

}  // namespace ddk
