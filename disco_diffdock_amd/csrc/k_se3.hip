// SE(3) x torus conformer update of one reverse-diffusion step, one workgroup per sample
// (reference utils/diffusion_utils.py:37-55 modify_conformer_batch, utils/geometry.py:6-85,126-156,
//  utils/torsion.py:71-86, and the perturbation arithmetic of utils/sampling.py:137-192):
//   upd   = score_coeff * score + noise_coeff * z                       (tr, rot, tor)
//   rigid = (pos - centroid) R(rot)^T + tr + centroid
//   flex  = sequential torsion rotations (in bond order, on already-updated coordinates)
//   out   = Kabsch-align(flex -> rigid)
// The 3x3 SVD of the reference is replaced by Horn's closed form (largest eigenvector of a 4x4 symmetric
// matrix, Jacobi in fp64 on one lane): same optimal proper rotation, reflection case included.
#include "model.h"

namespace ddk {


__device__ void axis_angle_to_matrix_dev(float ax, float ay, float az, float* R) {
  // utils/geometry.py:38-85 (quaternion route, small-angle series below 1e-6)
  const float ang = sqrtf(ax * ax + ay * ay + az * az);
  const float half = 0.5f * ang;
  const float s = fabsf(ang) < 1e-6f ? 0.5f - ang * ang / 48.0f : sinf(half) / ang;
  const float qr = cosf(half), qi = ax * s, qj = ay * s, qk = az * s;
  const float two_s = 2.0f / (qr * qr + qi * qi + qj * qj + qk * qk);
  R[0] = 1 - two_s * (qj * qj + qk * qk); R[1] = two_s * (qi * qj - qk * qr); R[2] = two_s * (qi * qk + qj * qr);
  R[3] = two_s * (qi * qj + qk * qr); R[4] = 1 - two_s * (qi * qi + qk * qk); R[5] = two_s * (qj * qk - qi * qr);
  R[6] = two_s * (qi * qk - qj * qr); R[7] = two_s * (qj * qk + qi * qr); R[8] = 1 - two_s * (qi * qi + qj * qj);
}

// rotation R minimising sum |R a_i - b_i|^2 given S[a][b] = sum a_i[a] b_i[b]  (Horn 1987)
__device__ void horn_rotation(const double* S, float* R) {
  double N[4][4] = {
      {S[0] + S[4] + S[8], S[5] - S[7], S[6] - S[2], S[1] - S[3]},
      {S[5] - S[7], S[0] - S[4] - S[8], S[1] + S[3], S[6] + S[2]},
      {S[6] - S[2], S[1] + S[3], -S[0] + S[4] - S[8], S[5] + S[7]},
      {S[1] - S[3], S[6] + S[2], S[5] + S[7], -S[0] - S[4] + S[8]}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  // Cyclic Jacobi.  A sweep costs six rotations of one sqrt + one rsqrt each on a single lane (the kernel's serial tail), so: the rotation
  // comes from (c, s) = (|r|, sgn(r) x) / sqrt(r^2 + x^2) with r = d + sgn(d) sqrt(d^2 + x^2), d = (N_qq - N_pp) / 2 - the textbook
  // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) without its three divisions; an off-diagonal element below 2^-60 of its diagonal pair is left
  // alone (the rotation would be the identity in fp64); the sweeps end when the off-diagonal mass is below 1e-32 of the diagonal's.
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0, dg = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      dg += N[p][p] * N[p][p];
#pragma unroll
      for (int q = p + 1; q < 4; ++q) off += N[p][q] * N[p][q];
    }
    if (off <= 1e-32 * dg || off < 1e-300) break;
    // (every index below is a compile-time constant after unrolling: N and V stay in registers - with rolled loops they lived in scratch memory)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double x = N[p][q];
        if (fabs(x) <= 8.7e-19 * (fabs(N[p][p]) + fabs(N[q][q]))) continue;
        const double d = 0.5 * (N[q][q] - N[p][p]);
        const double r = d + (d >= 0 ? 1.0 : -1.0) * sqrt(d * d + x * x);
        const double inv = rsqrt(r * r + x * x);
        const double c = fabs(r) * inv, s = (r >= 0 ? x : -x) * inv;
        // N <- G^T N G on the symmetric matrix: the two diagonal elements in closed form, the (p, q) element is zero by construction, the two
        // other rows / columns once (mirrored)
        const double app = N[p][p], aqq = N[q][q], cc = c * c, ss = s * s, cs2 = 2.0 * c * s * x;
        N[p][p] = cc * app - cs2 + ss * aqq;
        N[q][q] = ss * app + cs2 + cc * aqq;
        N[p][q] = 0.0; N[q][p] = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k == p || k == q) continue;
          const double a = N[k][p], b = N[k][q];
          const double np_ = c * a - s * b, nq_ = s * a + c * b;
          N[k][p] = np_; N[p][k] = np_; N[k][q] = nq_; N[q][k] = nq_;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double a = V[k][p], b = V[k][q];
          V[k][p] = c * a - s * b; V[k][q] = s * a + c * b;
        }
      }
  }
  double top = N[0][0], w = V[0][0], x = V[1][0], y = V[2][0], z = V[3][0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (N[k][k] > top) { top = N[k][k]; w = V[0][k]; x = V[1][k]; y = V[2][k]; z = V[3][k]; }
  const double nn = w * w + x * x + y * y + z * z, s2 = 2.0 / nn;
  R[0] = (float)(1 - s2 * (y * y + z * z)); R[1] = (float)(s2 * (x * y - z * w)); R[2] = (float)(s2 * (x * z + y * w));
  R[3] = (float)(s2 * (x * y + z * w)); R[4] = (float)(1 - s2 * (x * x + z * z)); R[5] = (float)(s2 * (y * z - x * w));
  R[6] = (float)(s2 * (x * z - y * w)); R[7] = (float)(s2 * (y * z + x * w)); R[8] = (float)(1 - s2 * (x * x + y * y));
}

// rigid_transform_Kabsch_3D_torch_batch (utils/geometry.py:126-156) for one point-set pair held in LDS: R, t with R a + t ~ b, proper
// rotation also in the reflection case.  All threads of the block call it (barriers inside); cA, cB [3], S [9], Rk [9], tk [3] in LDS.
__device__ void kabsch_block(const float* a, const float* bpts, int n, float* cA, float* cB, double* S, float* Rk, float* tk) {
  const int tid = threadIdx.x;
  if (tid < 3) {
    float sa = 0.0f, sb = 0.0f;
    for (int i = 0; i < n; ++i) { sa += a[3 * i + tid]; sb += bpts[3 * i + tid]; }
    cA[tid] = sa / (float)n; cB[tid] = sb / (float)n;
  }
  __syncthreads();
  if (tid < 9) {
    const int r = tid / 3, c = tid % 3;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)(a[3 * i + r] - cA[r]) * (double)(bpts[3 * i + c] - cB[c]);
    S[tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    horn_rotation(S, Rk);
    for (int k = 0; k < 3; ++k) tk[k] = -(Rk[3 * k] * cA[0] + Rk[3 * k + 1] * cA[1] + Rk[3 * k + 2] * cA[2]) + cB[k];
  }
  __syncthreads();
}

// test hooks (include/ddk_debug.h): the device functions above on caller-supplied inputs
__global__ __launch_bounds__(256) void debug_kabsch_kernel(const float* A, const float* Bp, int n, float* R_out, float* t_out) {
  __shared__ float a[MAX_LIG * 3], bb[MAX_LIG * 3], cA[3], cB[3], Rk[9], tk[3];
  __shared__ double S[9];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < n * 3; i += 256) { a[i] = A[(size_t)b * n * 3 + i]; bb[i] = Bp[(size_t)b * n * 3 + i]; }
  __syncthreads();
  kabsch_block(a, bb, n, cA, cB, S, Rk, tk);
  if (tid < 9) R_out[9 * b + tid] = Rk[tid];
  if (tid < 3) t_out[3 * b + tid] = tk[tid];
}

__global__ void debug_axis_angle_kernel(const float* aa, int n, float* R_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) axis_angle_to_matrix_dev(aa[3 * i], aa[3 * i + 1], aa[3 * i + 2], R_out + 9 * i);
}

hipError_t launch_debug_kabsch(const float* A, const float* Bp, int B, int n, float* R_out, float* t_out, hipStream_t s) {
  hipLaunchKernelGGL(debug_kabsch_kernel, dim3(B), dim3(256), 0, s, A, Bp, n, R_out, t_out);
  return hipGetLastError();
}

hipError_t launch_debug_axis_angle(const float* aa, int n, float* R_out, hipStream_t s) {
  hipLaunchKernelGGL(debug_axis_angle_kernel, dim3((n + 63) / 64), dim3(64), 0, s, aa, n, R_out);
  return hipGetLastError();
}

// POST: the workgroup of sample b first finishes the sample's scores from the heads' accumulators - heads_post_kernel's arithmetic (k_heads.hip:
// tor_bond_conv's mean / BatchNorm / tor_final_layer, final_conv's mean / BatchNorm and the tr / rot magnitude MLPs; same summation orders) - so that
// a reverse step needs no launch between the head convolutions and the update (ddk_sample without classifier-free guidance)
template <bool POST>
__global__ __launch_bounds__(256) void se3_update_kernel(Se3Args A, HeadArgs H) {
  __shared__ float rig[MAX_LIG * 3], flx_a[MAX_LIG * 3], flx2[MAX_LIG * 3];
  __shared__ float upd[6], ctr[3], Rm[9], cA[3], cB[3], Rk[9], tk[3], rot_th[64];
  __shared__ int2 rot_uv[64];
  __shared__ double S[9];
  float* flx = flx_a;
  const int b = blockIdx.x, tid = threadIdx.x, n = A.n_lig, R = A.R;
  if constexpr (POST) {
    __shared__ float v48[64][2 * NS], hid[64][NS], g12[12];
    if (H.prof_out != nullptr && b == (int)gridDim.x - 1 && tid < PROF_INTS) H.prof_out[tid] = H.exec_info[tid];
    if (A.tor != nullptr && R > 0) {
      for (int r0 = 0; r0 < R; r0 += 64) {
        const int chunk = min(64, R - r0);
        for (int i = tid; i < chunk * 2 * NS; i += 256) {
          const int rr = i / (2 * NS), c = i - rr * 2 * NS;
          const int bond = b * R + r0 + rr;
          float* row = H.h_sum + (size_t)(H.B + bond) * XW;
          const int ne = H.h_deg[bond];
          const float sv = row[c] / (float)(ne > 1 ? ne : 1);
          row[c] = 0.0f;
          v48[rr][c] = (sv - H.md.tb_bn_mean[c]) * H.md.tb_bn_scale[c] + H.md.tb_bn_bias[c];
        }
        __syncthreads();
        for (int i = tid; i < chunk * NS; i += 256) {      // tor_final_layer: Linear(48,24,no bias) -> tanh -> Linear(24,1,no bias)
          const int rr = i / NS, k = i - rr * NS;
          float a = 0.0f;
          for (int j = 0; j < 2 * NS; ++j) a += H.md.tf_w0[k * 2 * NS + j] * v48[rr][j];
          hid[rr][k] = H.md.tf_w3[k] * tanhf(a);
        }
        __syncthreads();
        if (tid < chunk) {
          float o = 0.0f;
          for (int j = 0; j < NS; ++j) o += hid[tid][j];
          if (H.scale_by_sigma) o *= H.sp.torus_norm_sqrt;
          H.tor_out[(size_t)b * R + r0 + tid] = o;
        }
        __syncthreads();
      }
    }
    float* row = H.h_sum + (size_t)b * XW;
    if (tid < 12) {
      g12[tid] = (row[tid] / (float)H.n_lig) * H.md.fc_bn_scale[tid / 3];
      row[tid] = 0.0f;
    }
    __syncthreads();
    if (tid < 2) {   // tid 0: translation, tid 1: rotation  (score_model.py:274-286)
      const int o = 3 * tid;
      const float px = g12[o] + g12[6 + o], py = g12[o + 1] + g12[7 + o], pz = g12[o + 2] + g12[8 + o];
      const float nrm = sqrtf(px * px + py * py + pz * pz);
      const float* w0n = tid == 0 ? H.md.tr_w0n : H.md.rot_w0n;
      const float* w3 = tid == 0 ? H.md.tr_w3 : H.md.rot_w3;
      const float* sb = tid == 0 ? H.sp.tr_sigb : H.sp.rot_sigb;
      float sv = tid == 0 ? H.md.tr_b3 : H.md.rot_b3;
      for (int k = 0; k < NS; ++k) sv += w3[k] * fmaxf(w0n[k] * nrm + sb[k], 0.0f);
      float f = sv / nrm;
      if (H.scale_by_sigma) f = tid == 0 ? f / H.sp.tr_sigma : f * H.sp.so3_norm;
      float* out = (tid == 0 ? H.tr_out : H.rot_out) + 3 * (size_t)b;
      out[0] = px * f; out[1] = py * f; out[2] = pz * f;
    }
    __syncthreads();      // the scores written above are read back through A.tr / A.rot / A.tor below (same workgroup)
  }
  const float* nz = A.noise ? A.noise + (size_t)b * (6 + R) : nullptr;
  for (int i = tid; i < n * 3; i += 256) rig[i] = A.pos[(size_t)b * n * 3 + i];
  if (tid < 6) {
    const int k = tid / 3;
    const float sc = (tid < 3 ? A.tr : A.rot)[3 * b + tid % 3];
    upd[tid] = A.sc[k] * sc + (nz ? A.nc[k] * nz[tid] : 0.0f);
  }
  __syncthreads();
  if (tid < 3) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += rig[3 * i + tid];
    ctr[tid] = s / (float)n;
  }
  if (tid == 32) axis_angle_to_matrix_dev(upd[3], upd[4], upd[5], Rm);
  __syncthreads();
  if (tid < n) {
    const float x = rig[3 * tid] - ctr[0], y = rig[3 * tid + 1] - ctr[1], z = rig[3 * tid + 2] - ctr[2];
    const float rx = Rm[0] * x + Rm[1] * y + Rm[2] * z + upd[0] + ctr[0];
    const float ry = Rm[3] * x + Rm[4] * y + Rm[5] * z + upd[1] + ctr[1];
    const float rz = Rm[6] * x + Rm[7] * y + Rm[8] * z + upd[2] + ctr[2];
    // every thread owns its atom: safe to overwrite in place after the barrier above
    rig[3 * tid] = rx; rig[3 * tid + 1] = ry; rig[3 * tid + 2] = rz;
    flx[3 * tid] = rx; flx[3 * tid + 1] = ry; flx[3 * tid + 2] = rz;
  }
  __syncthreads();
  if (A.tor == nullptr || R == 0) {
    for (int i = tid; i < n * 3; i += 256) A.pos_out[(size_t)b * n * 3 + i] = rig[i];
    return;
  }
  // Sequential torsion rotations, one barrier per rotor: the rotor table (u, v, angle) and every atom's mask bits are fetched for up to 64 rotors at
  // once (one exposed memory latency per chunk instead of three per rotor), every thread builds the rotor's matrix itself from the CURRENT
  // coordinates (same instruction sequence on every lane) and writes its atom into the other of two coordinate buffers.
  float* cur = flx;
  float* nxt = flx2;
  for (int r0 = 0; r0 < R; r0 += 64) {
    const int chunk = min(64, R - r0);
    if (tid < chunk) {
      const int r = r0 + tid;
      rot_uv[tid] = make_int2(A.rot_u[r], A.rot_v[r]);
      rot_th[tid] = A.sc[2] * A.tor[(size_t)b * R + r] + (nz ? A.nc[2] * nz[6 + r] : 0.0f);
    }
    unsigned long long mbits = 0ull;
    if (tid < n)
      for (int k = 0; k < chunk; ++k) mbits |= (unsigned long long)(A.mask_rotate[(size_t)(r0 + k) * n + tid] != 0) << k;
    __syncthreads();
    for (int k = 0; k < chunk; ++k) {
      if (tid < n) {
        const int2 uv = rot_uv[k];
        const float th = rot_th[k];
        const float px = cur[3 * uv.y], py = cur[3 * uv.y + 1], pz = cur[3 * uv.y + 2];
        float x = cur[3 * tid], y = cur[3 * tid + 1], z = cur[3 * tid + 2];
        if ((mbits >> k) & 1ull) {
          const float ax = cur[3 * uv.x] - px, ay = cur[3 * uv.x + 1] - py, az = cur[3 * uv.x + 2] - pz;
          const float nn = sqrtf(ax * ax + ay * ay + az * az);
          float Rl[9];
          axis_angle_to_matrix_dev(ax / nn * th, ay / nn * th, az / nn * th, Rl);
          x -= px; y -= py; z -= pz;
          const float nx = Rl[0] * x + Rl[1] * y + Rl[2] * z + px, ny = Rl[3] * x + Rl[4] * y + Rl[5] * z + py, nz_ = Rl[6] * x + Rl[7] * y + Rl[8] * z + pz;
          x = nx; y = ny; z = nz_;
        }
        nxt[3 * tid] = x; nxt[3 * tid + 1] = y; nxt[3 * tid + 2] = z;
      }
      __syncthreads();
      float* t_ = cur; cur = nxt; nxt = t_;
    }
  }
  flx = cur;
  kabsch_block(flx, rig, n, cA, cB, S, Rk, tk);
  if (tid < n) {
    const float x = flx[3 * tid], y = flx[3 * tid + 1], z = flx[3 * tid + 2];
    float* o = A.pos_out + ((size_t)b * n + tid) * 3;
    o[0] = Rk[0] * x + Rk[1] * y + Rk[2] * z + tk[0];
    o[1] = Rk[3] * x + Rk[4] * y + Rk[5] * z + tk[1];
    o[2] = Rk[6] * x + Rk[7] * y + Rk[8] * z + tk[2];
  }
}

// randomize_position (utils/sampling.py:12-34) for B copies of one conformer, one workgroup per sample:
// sequential torsion updates in bond order (utils/torsion.py:48-68: axis pos_u - pos_v, pivot pos_v, zero updates skipped),
// then (pos - centroid) R^T + tr with the caller's rotation matrix and translation draw.
__global__ __launch_bounds__(256) void randomize_kernel(RandPosArgs A) {
  __shared__ float p[MAX_LIG * 3];
  __shared__ float Rt[9], piv[3], ctr[3];
  __shared__ int skip;
  const int b = blockIdx.x, tid = threadIdx.x, n = A.n_lig, R = A.R;
  for (int i = tid; i < n * 3; i += 256) p[i] = A.pos0[i];
  __syncthreads();
  if (A.tor != nullptr) {
    for (int r = 0; r < R; ++r) {
      if (tid == 0) {
        const float th = A.tor[(size_t)b * R + r];
        skip = th == 0.0f;
        const int u = A.rot_u[r], v = A.rot_v[r];
        const float ax = p[3 * u] - p[3 * v], ay = p[3 * u + 1] - p[3 * v + 1], az = p[3 * u + 2] - p[3 * v + 2];
        const float nn = sqrtf(ax * ax + ay * ay + az * az);
        axis_angle_to_matrix_dev(ax / nn * th, ay / nn * th, az / nn * th, Rt);
        piv[0] = p[3 * v]; piv[1] = p[3 * v + 1]; piv[2] = p[3 * v + 2];
      }
      __syncthreads();
      if (!skip && tid < n && A.mask_rotate[(size_t)r * n + tid]) {
        const float x = p[3 * tid] - piv[0], y = p[3 * tid + 1] - piv[1], z = p[3 * tid + 2] - piv[2];
        p[3 * tid] = Rt[0] * x + Rt[1] * y + Rt[2] * z + piv[0];
        p[3 * tid + 1] = Rt[3] * x + Rt[4] * y + Rt[5] * z + piv[1];
        p[3 * tid + 2] = Rt[6] * x + Rt[7] * y + Rt[8] * z + piv[2];
      }
      __syncthreads();
    }
  }
  if (tid < 3) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += p[3 * i + tid];
    ctr[tid] = s / (float)n;
  }
  __syncthreads();
  if (tid < n) {
    const float* M = A.rot + (size_t)b * 9;
    const float x = p[3 * tid] - ctr[0], y = p[3 * tid + 1] - ctr[1], z = p[3 * tid + 2] - ctr[2];
    float ox = M[0] * x + M[1] * y + M[2] * z, oy = M[3] * x + M[4] * y + M[5] * z, oz = M[6] * x + M[7] * y + M[8] * z;
    if (A.tr != nullptr) { ox += A.tr[3 * b]; oy += A.tr[3 * b + 1]; oz += A.tr[3 * b + 2]; }
    float* o = A.pos_out + ((size_t)b * n + tid) * 3;
    o[0] = ox; o[1] = oy; o[2] = oz;
  }
}

hipError_t launch_randomize(const RandPosArgs& A, hipStream_t s) {
  hipLaunchKernelGGL(randomize_kernel, dim3(A.B), dim3(256), 0, s, A);
  return hipGetLastError();
}

// pose metrics of evaluate.py:297-338, one workgroup per pose (see include/ddk.h: ddk_pose_metrics)
__global__ __launch_bounds__(256) void pose_metrics_kernel(const float* pos, const float* ref, const uint8_t* mask, const int32_t* perms, int n_perms,
                                                           const float* rec_pos, int n_lig, int n_rec, float* out) {
  __shared__ float p[MAX_LIG * 3], rf[MAX_LIG * 3];
  __shared__ unsigned char keep[MAX_LIG];
  __shared__ float red[4][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < n_lig * 3; i += 256) { p[i] = pos[(size_t)b * n_lig * 3 + i]; rf[i] = ref[i]; }
  for (int i = tid; i < n_lig; i += 256) keep[i] = mask ? (mask[i] != 0) : 1;
  __syncthreads();
  float cnt = 0.f, cx = 0.f, cy = 0.f, cz = 0.f, rx = 0.f, ry = 0.f, rz = 0.f;
  float mcross = INFINITY, mself = INFINITY;
  for (int i = tid; i < n_lig; i += 256)
    if (keep[i]) {
      cnt += 1.f;
      cx += p[3 * i]; cy += p[3 * i + 1]; cz += p[3 * i + 2];
      rx += rf[3 * i]; ry += rf[3 * i + 1]; rz += rf[3 * i + 2];
      for (int j = 0; j < n_lig; ++j)
        if (j != i && keep[j]) {
          const float ex = p[3 * i] - p[3 * j], ey = p[3 * i + 1] - p[3 * j + 1], ez = p[3 * i + 2] - p[3 * j + 2];
          mself = fminf(mself, ex * ex + ey * ey + ez * ez);
        }
    }
  // RMSD: the symmetry-corrected one of evaluate.py:308-310 (spyrmsd symmrmsd without minimisation) = the minimum over the graph
  // automorphisms G of sqrt(mean_i |pos[G(i)] - ref[i]|^2); n_perms = 0 / perms = null: identity only (the fallback of :313).
  // One thread per permutation (each sum is n_lig terms in atom order: deterministic), then a block minimum.
  float best = INFINITY;
  const int K = perms ? n_perms : 1;
  for (int k = tid; k < K; k += 256) {
    float s2 = 0.f;
    for (int i = 0; i < n_lig; ++i)
      if (keep[i]) {
        const int j = perms ? perms[(size_t)k * n_lig + i] : i;
        if ((unsigned)j >= (unsigned)n_lig) { s2 = INFINITY; break; }      // not a ligand atom: this table row can never be the minimum (rmsd = inf if none is valid)
        const float dx = p[3 * j] - rf[3 * i], dy = p[3 * j + 1] - rf[3 * i + 1], dz = p[3 * j + 2] - rf[3 * i + 2];
        s2 += dx * dx + dy * dy + dz * dz;
      }
    best = fminf(best, s2);
  }
  for (int r = tid; r < n_rec; r += 256) {
    const float qx = rec_pos[3 * r], qy = rec_pos[3 * r + 1], qz = rec_pos[3 * r + 2];
    for (int i = 0; i < n_lig; ++i)
      if (keep[i]) {
        const float ex = qx - p[3 * i], ey = qy - p[3 * i + 1], ez = qz - p[3 * i + 2];
        mcross = fminf(mcross, ex * ex + ey * ey + ez * ez);
      }
  }
  // block reductions: sums of (cnt, centroid components) and the three minima
  float vals[7] = {cnt, cx, cy, cz, rx, ry, rz};
  float sums[7];
  for (int k = 0; k < 7; ++k) {
    red[0][tid] = vals[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[0][tid] += red[0][tid + o]; __syncthreads(); }
    sums[k] = red[0][0];
    __syncthreads();
  }
  red[1][tid] = mcross; red[2][tid] = mself; red[3][tid] = best;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      red[1][tid] = fminf(red[1][tid], red[1][tid + o]); red[2][tid] = fminf(red[2][tid], red[2][tid + o]);
      red[3][tid] = fminf(red[3][tid], red[3][tid + o]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float n = fmaxf(sums[0], 1.f);
    const float dx = (sums[1] - sums[4]) / n, dy = (sums[2] - sums[5]) / n, dz = (sums[3] - sums[6]) / n;
    out[4 * b + 0] = sqrtf(red[3][0] / n);
    out[4 * b + 1] = sqrtf(dx * dx + dy * dy + dz * dz);
    out[4 * b + 2] = sqrtf(red[1][0]);
    out[4 * b + 3] = sqrtf(red[2][0]);
  }
}

hipError_t launch_pose_metrics(const float* pos, const float* ref, const uint8_t* mask, const int32_t* perms, int n_perms, const float* rec_pos,
                               int B, int n_lig, int n_rec, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pose_metrics_kernel, dim3(B), dim3(256), 0, s, pos, ref, mask, perms, n_perms, rec_pos, n_lig, n_rec, out);
  return hipGetLastError();
}

// classifier-free guidance: score <- score + w * (score - score_unconditional)   (utils/sampling.py:131-133)
__global__ void cfg_combine_kernel(float* score, const float* uncond, float w, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) score[i] = score[i] + w * (score[i] - uncond[i]);
}

hipError_t launch_cfg_combine(float* score, const float* uncond, float weight, int64_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, score, uncond, weight, n);
  return hipGetLastError();
}

hipError_t launch_se3(const Se3Args& A, hipStream_t s, const HeadArgs* post) {
  if (post) hipLaunchKernelGGL(se3_update_kernel<true>, dim3(A.B), dim3(256), 0, s, A, *post);
  else { HeadArgs H = {}; hipLaunchKernelGGL(se3_update_kernel<false>, dim3(A.B), dim3(256), 0, s, A, H); }
  return hipGetLastError();
}

}  // namespace ddk
