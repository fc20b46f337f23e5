// FasterTensorProduct.forward at the reference's op boundary (models/tensor_layers.py:65-116):
//   out[e] = TP(x_dst[e], sh[e], w[e])  with the per-edge weights w [E, W] resident in HBM.
// HBM-bound by construction (W*4 B of weights per edge for ~5 kFLOP): one wave per edge streams the
// weight row with fully coalesced 256-B reads; the <=276 row operands u_i live in an LDS table per wave.
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

constexpr int TP_WAVES = 4;
constexpr int U_MAX = 2 * (NS + NV) + 2 * 3 * (NS + 2 * NV);   // 30+30+108+108 = 276

__global__ __launch_bounds__(64 * TP_WAVES) void tp_forward_kernel(TpKArgs A) {
  __shared__ float U[TP_WAVES][U_MAX + 4];
  __shared__ float O[TP_WAVES][XW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* u = U[wave];
  float* o = O[wave];
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  const int n0e = A.in_mul[0], n1o = A.in_mul[1], n1e = A.in_mul[2], n0o = A.in_mul[3];
  // row tables: u0e[n_in0], u1o[n_in1][3], u1e[n_in2][3], u0o[n_in3]
  const int b0 = 0, b1 = A.n_in[0], b2 = b1 + 3 * A.n_in[1], b3 = b2 + 3 * A.n_in[2];
  for (int64_t e = (int64_t)blockIdx.x * TP_WAVES + wave; e < A.E; e += (int64_t)gridDim.x * TP_WAVES) {
    const float* xr = A.x + e * A.din;
    const float s0 = A.sh[e * 4], vx = A.sh[e * 4 + 1], vy = A.sh[e * 4 + 2], vz = A.sh[e * 4 + 3];
    const float* pa = xr;
    const float* pp = xr + n0e;
    const float* pq = pp + 3 * n1o;
    const float* pc = pq + 3 * n1e;
    for (int i = lane; i < XW; i += 64) o[i] = 0.0f;
    // scalars a (0e) and c (0o)
    for (int i = lane; i < n0e; i += 64) {
      const float a = pa[i];
      u[b0 + i] = a * s0;
      u[b1 + 3 * i] = a * vx; u[b1 + 3 * i + 1] = a * vy; u[b1 + 3 * i + 2] = a * vz;
    }
    for (int i = lane; i < n0o; i += 64) {
      const float c = pc[i];
      const int r1e = n1o + n1e + i, r0o = n1e + i;
      u[b2 + 3 * r1e] = c * vx; u[b2 + 3 * r1e + 1] = c * vy; u[b2 + 3 * r1e + 2] = c * vz;
      u[b3 + r0o] = c * s0;
    }
    for (int m = lane; m < n1o; m += 64) {
      const float px = pp[3 * m], py = pp[3 * m + 1], pz = pp[3 * m + 2];
      u[b0 + n0e + m] = (px * vx + py * vy + pz * vz) * inv_s3;
      const int r1o = n0e + m;
      u[b1 + 3 * r1o] = px * s0; u[b1 + 3 * r1o + 1] = py * s0; u[b1 + 3 * r1o + 2] = pz * s0;
      u[b2 + 3 * m] = (py * vz - pz * vy) * inv_s2;
      u[b2 + 3 * m + 1] = (pz * vx - px * vz) * inv_s2;
      u[b2 + 3 * m + 2] = (px * vy - py * vx) * inv_s2;
    }
    for (int m = lane; m < n1e; m += 64) {
      const float qx = pq[3 * m], qy = pq[3 * m + 1], qz = pq[3 * m + 2];
      const int r1o = n0e + n1o + m, r1e = n1o + m;
      u[b1 + 3 * r1o] = (qy * vz - qz * vy) * inv_s2;
      u[b1 + 3 * r1o + 1] = (qz * vx - qx * vz) * inv_s2;
      u[b1 + 3 * r1o + 2] = (qx * vy - qy * vx) * inv_s2;
      u[b2 + 3 * r1e] = qx * s0; u[b2 + 3 * r1e + 1] = qy * s0; u[b2 + 3 * r1e + 2] = qz * s0;
      u[b3 + m] = (qx * vx + qy * vy + qz * vz) * inv_s3;
    }
    __syncthreads();
    const float* wr = A.w + e * A.W;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int n_in = A.n_in[blk], n_out = A.n_out[blk];
      if (n_in == 0 || n_out == 0) continue;
      const bool vec = (blk == 1 || blk == 2);
      const int ub = blk == 0 ? b0 : (blk == 1 ? b1 : (blk == 2 ? b2 : b3));
      const int rows_per_it = 64 / n_out;           // lanes cover whole rows so that a lane keeps its k
      const int active = rows_per_it * n_out;
      const float rs = 1.0f / sqrtf((float)n_in);
      if (lane < active) {
        const int k = lane % n_out, i0 = lane / n_out;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        const float* wb = wr + A.blk_off[blk];
        for (int i = i0; i < n_in; i += rows_per_it) {
          const float wv = wb[i * n_out + k];
          if (vec) {
            a0 += u[ub + 3 * i] * wv; a1 += u[ub + 3 * i + 1] * wv; a2 += u[ub + 3 * i + 2] * wv;
          } else {
            a0 += u[ub + i] * wv;
          }
        }
        if (vec) {
          atomicAdd(&o[A.out_off[blk] + 3 * k], a0 * rs);
          atomicAdd(&o[A.out_off[blk] + 3 * k + 1], a1 * rs);
          atomicAdd(&o[A.out_off[blk] + 3 * k + 2], a2 * rs);
        } else {
          atomicAdd(&o[A.out_off[blk] + k], a0 * rs);
        }
      }
    }
    __syncthreads();
    for (int i = lane; i < A.dout; i += 64) A.out[e * A.dout + i] = o[i];
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
  int64_t blocks = (E + TP_WAVES - 1) / TP_WAVES;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(tp_forward_kernel, dim3((unsigned)blocks), dim3(64 * TP_WAVES), 0, s, k);
  return hipGetLastError();
}

}  // namespace ddk
