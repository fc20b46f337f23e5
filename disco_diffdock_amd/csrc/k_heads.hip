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

__global__ __launch_bounds__(256) void center_head_kernel(HeadArgs A) {
  __shared__ float lp[MAX_LIG * 3];
  __shared__ float part[MAX_LIG][12];
  __shared__ float ctr[3];
  __shared__ float g12[12];
  const int b = blockIdx.x, tid = threadIdx.x, n = A.n_lig;
  for (int i = tid; i < n * 3; i += 256) lp[i] = A.lig_pos[(size_t)b * n * 3 + i];
  __syncthreads();
  if (tid < 3) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += lp[3 * i + tid];
    ctr[tid] = s / (float)n;
  }
  __syncthreads();
  if (tid < n) {
    const int i = tid;
    const float vx = lp[3 * i] - ctr[0], vy = lp[3 * i + 1] - ctr[1], vz = lp[3 * i + 2] - ctr[2];
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
    const float s1[3] = {vx * inv, vy * inv, vz * inv};   // l=1 spherical harmonics; s0 = 1
    float gs[DE];
    smear(d, A.md.center_edge, gs);
    float in48[2 * NS];
    edge_mlp(A.md.center_edge, A.sp.center_edge_sigb, gs, in48);
    const float* xr = A.x + ((size_t)b * n + i) * XW;
#pragma unroll
    for (int k = 0; k < NS; ++k) in48[NS + k] = xr[k];
    float h2[2 * NS];
#pragma unroll 4
    for (int o = 0; o < 2 * NS; ++o) {
      float a = A.md.fc_b0[o];
#pragma unroll
      for (int k = 0; k < 2 * NS; ++k) a += A.md.fc_w0[o * 2 * NS + k] * in48[k];
      h2[o] = fmaxf(a, 0.0f);
    }
    // weight vector layout (e3nn instruction order): A 0e(x)1o->1o [24][2] | B 1o(x)0e->1o [6][2] | C 1o(x)1o->1e [6][2]
    //                                               | D 1e(x)0e->1e [6][2] | E 1e(x)1o->1o [6][2] | F 0o(x)1o->1e [24][2]
    float sA[2] = {0, 0}, sF[2] = {0, 0}, vB[2][3] = {}, vC[2][3] = {}, vD[2][3] = {}, vE[2][3] = {};
    int idx = 0;
    auto wgt = [&](int row) {
      float a = A.md.fc_b4[row];
#pragma unroll
      for (int k = 0; k < 2 * NS; ++k) a += A.md.fc_w4[row * 2 * NS + k] * h2[k];
      return a;
    };
    for (int u = 0; u < NS; ++u)
      for (int w = 0; w < 2; ++w) sA[w] += wgt(idx++) * xr[u];
    for (int u = 0; u < NV; ++u)
      for (int w = 0; w < 2; ++w) {
        const float wv = wgt(idx++);
        for (int k = 0; k < 3; ++k) vB[w][k] += wv * xr[OFF_P + 3 * u + k];
      }
    for (int u = 0; u < NV; ++u)
      for (int w = 0; w < 2; ++w) {
        const float wv = wgt(idx++);
        for (int k = 0; k < 3; ++k) vC[w][k] += wv * xr[OFF_P + 3 * u + k];
      }
    for (int u = 0; u < NV; ++u)
      for (int w = 0; w < 2; ++w) {
        const float wv = wgt(idx++);
        for (int k = 0; k < 3; ++k) vD[w][k] += wv * xr[OFF_Q + 3 * u + k];
      }
    for (int u = 0; u < NV; ++u)
      for (int w = 0; w < 2; ++w) {
        const float wv = wgt(idx++);
        for (int k = 0; k < 3; ++k) vE[w][k] += wv * xr[OFF_Q + 3 * u + k];
      }
    for (int u = 0; u < NS; ++u)
      for (int w = 0; w < 2; ++w) sF[w] += wgt(idx++) * xr[OFF_C + u];
    // path coefficient sqrt(3/36) times the 3j normalisation
    const float cS = 0.28867513459481288f * 0.57735026918962576f;   // 1/sqrt12 * 1/sqrt3
    const float cX = 0.28867513459481288f * 0.40824829046386302f;   // 1/sqrt12 * 1/sqrt6
    for (int w = 0; w < 2; ++w) {
      const float cEx = vE[w][1] * s1[2] - vE[w][2] * s1[1], cEy = vE[w][2] * s1[0] - vE[w][0] * s1[2],
                  cEz = vE[w][0] * s1[1] - vE[w][1] * s1[0];
      const float cCx = vC[w][1] * s1[2] - vC[w][2] * s1[1], cCy = vC[w][2] * s1[0] - vC[w][0] * s1[2],
                  cCz = vC[w][0] * s1[1] - vC[w][1] * s1[0];
      part[i][3 * w + 0] = cS * (sA[w] * s1[0] + vB[w][0]) + cX * cEx;
      part[i][3 * w + 1] = cS * (sA[w] * s1[1] + vB[w][1]) + cX * cEy;
      part[i][3 * w + 2] = cS * (sA[w] * s1[2] + vB[w][2]) + cX * cEz;
      part[i][6 + 3 * w + 0] = cS * (sF[w] * s1[0] + vD[w][0]) + cX * cCx;
      part[i][6 + 3 * w + 1] = cS * (sF[w] * s1[1] + vD[w][1]) + cX * cCy;
      part[i][6 + 3 * w + 2] = cS * (sF[w] * s1[2] + vD[w][2]) + cX * cCz;
    }
  }
  __syncthreads();
  if (tid < 12) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += part[i][tid];
    g12[tid] = (s / (float)n) * A.md.fc_bn_scale[tid / 3];
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

__global__ __launch_bounds__(64) void torsion_head_kernel(HeadArgs A) {
  __shared__ float lp[MAX_LIG * 3];
  __shared__ int nb[BOND_CAP];
  __shared__ float red[BOND_CAP][2 * NS];
  __shared__ float v48[2 * NS];
  __shared__ int n_nb;
  const int b = blockIdx.x / A.R, r = blockIdx.x % A.R, lane = threadIdx.x, n = A.n_lig;
  for (int i = lane; i < n * 3; i += 64) lp[i] = A.lig_pos[(size_t)b * n * 3 + i];
  __syncthreads();
  const int u = A.rot_u[r], v = A.rot_v[r];
  const float cx = (lp[3 * u] + lp[3 * v]) * 0.5f, cy = (lp[3 * u + 1] + lp[3 * v + 1]) * 0.5f, cz = (lp[3 * u + 2] + lp[3 * v + 2]) * 0.5f;
  // neighbour atoms of the bond centre: ascending index, first BOND_CAP (radius(..., max_num_neighbors=32), score_model.py:430)
  int cnt = 0;
  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    bool in = false;
    if (k < n) {
      const float dx = lp[3 * k] - cx, dy = lp[3 * k + 1] - cy, dz = lp[3 * k + 2] - cz;
      in = dx * dx + dy * dy + dz * dz < A.lig_r2;
    }
    const unsigned long long mask = __ballot(in);
    const int rank = cnt + __popcll(mask & ((1ull << lane) - 1ull));
    if (in && rank < BOND_CAP) nb[rank] = k;
    cnt += __popcll(mask);
  }
  if (lane == 0) n_nb = cnt < BOND_CAP ? cnt : BOND_CAP;
  __syncthreads();
  const int ne = n_nb;
  if (lane < ne) {
    const int k = nb[lane];
    const float vx = lp[3 * k] - cx, vy = lp[3 * k + 1] - cy, vz = lp[3 * k + 2] - cz;
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
    const float s1[3] = {vx * inv, vy * inv, vz * inv};
    // l=2 spherical harmonics of the bond axis (component normalised), e3nn basis
    float bx = lp[3 * v] - lp[3 * u], by = lp[3 * v + 1] - lp[3 * u + 1], bz = lp[3 * v + 2] - lp[3 * u + 2];
    const float bn = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-12f);
    bx /= bn; by /= bn; bz /= bn;
    const float s3 = 1.7320508075688772f, s5 = 2.2360679774997897f;
    const float y0 = s5 * s3 * bx * bz, y1 = s5 * s3 * bx * by, y2 = s5 * (by * by - 0.5f * (bx * bx + bz * bz)),
                y3 = s5 * s3 * by * bz, y4 = s5 * (s3 * 0.5f) * (bz * bz - bx * bx);
    // 1o part of FullTensorProduct((0e+1o), 2e): sqrt3 * sum_ij w3j(1,2,1)[i,j,k] s1_i y_j
    const float ca = 0.31622776601683794f, cb = 0.18257418583505536f;   // 1/sqrt10, 1/sqrt30
    const float T0 = s3 * (-cb * s1[0] * y2 - ca * s1[0] * y4 + ca * s1[1] * y1 + ca * s1[2] * y0);
    const float T1 = s3 * (ca * s1[0] * y1 + 2.0f * cb * s1[1] * y2 + ca * s1[2] * y3);
    const float T2 = s3 * (ca * s1[0] * y0 + ca * s1[1] * y3 - cb * s1[2] * y2 + ca * s1[2] * y4);
    float gs[DE];
    smear(d, A.md.final_edge, gs);
    float attr[NE];
    edge_mlp(A.md.final_edge, A.md.final_edge_b1, gs, attr);
    const float* xk = A.x + ((size_t)b * n + k) * XW;
    const float* xu = A.x + ((size_t)b * n + u) * XW;
    const float* xv = A.x + ((size_t)b * n + v) * XW;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      attr[NS + j] = xk[j];
      attr[2 * NS + j] = xu[j] + xv[j];
    }
    float h[NE];
#pragma unroll 4
    for (int o = 0; o < NE; ++o) {
      float a = A.md.tb_b0[o];
#pragma unroll
      for (int j = 0; j < NE; ++j) a += A.md.tb_w0[o * NE + j] * attr[j];
      h[o] = fmaxf(a, 0.0f);
    }
    // paths: [1o (x) 1o -> 0e : [6][24]] then [1e (x) 1o -> 0o : [6][24]]; output irreps 24x0o + 24x0e
    const float c = 0.40824829046386302f * 0.57735026918962576f;   // sqrt(1/6) * 1/sqrt3
    float* outp = red[lane];
    for (int w = 0; w < 2 * NS; ++w) outp[w] = 0.0f;
    for (int path = 0; path < 2; ++path) {
      const int xo = path == 0 ? OFF_P : OFF_Q;
      const int oo = path == 0 ? NS : 0;
      for (int uu = 0; uu < NV; ++uu) {
        const float dt = (xk[xo + 3 * uu] * T0 + xk[xo + 3 * uu + 1] * T1 + xk[xo + 3 * uu + 2] * T2) * c;
        for (int w = 0; w < NS; ++w) {
          const int row = path * (NV * NS) + uu * NS + w;
          float a = A.md.tb_b4[row];
#pragma unroll
          for (int j = 0; j < NE; ++j) a += A.md.tb_w4[row * NE + j] * h[j];
          outp[oo + w] += a * dt;
        }
      }
    }
  }
  __syncthreads();
  if (lane < 2 * NS) {
    float s = 0.0f;
    for (int e = 0; e < ne; ++e) s += red[e][lane];
    s = s / (float)(ne > 1 ? ne : 1);
    v48[lane] = (s - A.md.tb_bn_mean[lane]) * A.md.tb_bn_scale[lane] + A.md.tb_bn_bias[lane];
  }
  __syncthreads();
  if (lane == 0) {   // tor_final_layer: Linear(48,24,no bias) -> tanh -> Linear(24,1,no bias)
    float o = 0.0f;
    for (int j = 0; j < NS; ++j) {
      float a = 0.0f;
      for (int k = 0; k < 2 * NS; ++k) a += A.md.tf_w0[j * 2 * NS + k] * v48[k];
      o += A.md.tf_w3[j] * tanhf(a);
    }
    if (A.scale_by_sigma) o *= A.sp.torus_norm_sqrt;
    A.tor_out[(size_t)b * A.R + r] = o;
  }
}

hipError_t launch_heads(const HeadArgs& A, bool torsion, hipStream_t s) {
  hipLaunchKernelGGL(center_head_kernel, dim3(A.B), dim3(256), 0, s, A);
  if (torsion && A.R > 0) hipLaunchKernelGGL(torsion_head_kernel, dim3(A.B * A.R), dim3(64), 0, s, A);
  return hipGetLastError();
}

}  // namespace ddk
