// Fused tensor-product convolution for gfx950 (MI355X): the star kernel of the hot path.
//
// Replaces TensorProductConvLayer.forward (reference models/tensor_layers.py:147-168) with the
// FasterTensorProduct of :65-116 for one conv layer:
//     w   = fc[g](edge_attr)          Linear(72,72) + ReLU + Linear(72,W)      [E, W] never hits HBM
//     m   = TP(x[dst], sh, w)         l<=1 Clebsch-Gordan products, dense contraction with w
//     sum = scatter_sum(m, src)       wave-level segmented reduction + fp32 atomics on the segment tails
// (mean / BatchNorm / residual are applied by node_finalize_kernel below.)
//
// Mapping onto CDNA4 (one 64-lane wave == one workgroup, 32 edges per wave iteration):
//   * both GEMMs run on the fp32 matrix cores: v_mfma_f32_32x32x2_f32, D[row][col] with
//       rows  = 32 output features of the GEMM (hidden units / per-edge weights)   -> A operand = packed weights
//       cols  = the 32 edges of the tile                                           -> B operand = per-edge activations
//     so lane l always "owns" edge (l & 31); lane-half (l >> 5) selects the K pair of the A/B operands and
//     the 4-row groups of D.  The hidden vector h = relu(W1 e + b1) therefore comes out of GEMM1 already in
//     the register layout GEMM2 needs for its B operand (the K order of GEMM2 is permuted at weight-packing
//     time to make this true) - h never leaves the VGPRs.
//   * the [32 rows x 72] W2 tiles are streamed from L2 in MFMA fragment order (9 x 16 B per lane per tile) into ONE
//     register set: each 16-B fragment is reloaded with the next tile's data right after the 4 MFMAs that consumed it,
//     so the stream for tile t+1 lands under the rest of tile t's 36-MFMA burst; the F operands of the tile's 4 units
//     are requested from LDS before the burst and the epilogue arithmetic is branch-free (unit kinds select multipliers);
//     the per-edge weights w[row] appear in D and are consumed immediately: each group of 4 D registers is a
//     "unit" = 4 consecutive TP rows i for the output-channel pair (2k, 2k+1) (k = 2*kpair + lane-half);
//     the row operands u_i = f(x[dst], sh) are read from a per-edge LDS table (F row, 140 floats, stride chosen
//     bank-conflict free for ds_read_b128).
//   * when a (block, kpair) finishes, the value is reduced over runs of equal edge_src inside the wave with a
//     5-step segmented scan (ds_bpermute) and only the run tails issue global fp32 atomics.
//   * bias vectors ride in the MFMA C operand (accumulator init), so no separate bias pass exists.
//
// Roofline: MFMA-bound (2*72*(72+W) flop per edge vs ~650 B per edge of HBM traffic), see DESIGN.md.
#include <stdlib.h>

#include "ddk_internal.h"

namespace ddk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvKArgs {
  const float* x;
  const int32_t* src;
  const int32_t* dst;
  const float* edge_attr;
  const float* sh;
  float* sum;
  const int32_t* tile_info;
  int32_t* counter;
  const float* w1p;   // [4][3][9][64][4]
  const float* b1p;   // [4][3][2][16]
  const float* w2p;   // [4][n_tiles][9][64][4]
  const float* b2p;   // [4][n_tiles][2][16]
  const Unit* units;  // [n_tiles*4]
  int n_tiles;
  int lig_side_only;  // evaluate groups 0,1 only
  int g2_limit;       // >= 0: evaluate only the first g2_limit edges of group 2 (see ConvLaunch)
  float* sum_g2;
  int g2_node_off;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct UnitQuad { Unit u[4]; };
// unit descriptors are wave-uniform: read them through the constant address space so that they take the scalar
// (s_load_dwordx16) path and end up in SGPRs
typedef const int32_t __attribute__((address_space(4))) cint32;
__device__ __forceinline__ UnitQuad load_unit_quad(const Unit* p) {
  cint32* q = (cint32*)(uintptr_t)p;
  UnitQuad r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.u[i].w0 = q[4 * i + 0];
    r.u[i].w1 = q[4 * i + 1];
    r.u[i].scale = __int_as_float(q[4 * i + 2]);
    r.u[i].pad = 0;
  }
  return r;
}

template <bool GATHER>
__global__ __launch_bounds__(64) void conv_fused_kernel(ConvKArgs A) {
  __shared__ __attribute__((aligned(16))) float F[32 * F_STRIDE + 16];
  const int lane = threadIdx.x;
  if (lane < 16) F[32 * F_STRIDE + lane] = 0.0f;   // pad: the v5 epilogue always reads 12 floats per unit
  const int el = lane & 31;
  const int hh = lane >> 5;
  const int ts1 = A.tile_info[1], ts2 = A.tile_info[2];
  int ts3 = A.tile_info[3], ts4 = A.tile_info[4];
  const int go0 = A.tile_info[5], go1 = A.tile_info[6], go2 = A.tile_info[7], go3 = A.tile_info[8], go4 = A.tile_info[9];
  int g2_end = go3;
  const bool g2_shared = A.g2_limit >= 0;
  if (A.g2_limit >= 0) {   // shortened group 2: shift the tile ranges of groups 2 and 3
    const int len = min(A.g2_limit, go3 - go2);
    const int delta = (ts3 - ts2) - (len + 31) / 32;
    ts3 -= delta; ts4 -= delta;
    g2_end = go2 + len;
  }
  if (A.lig_side_only) ts4 = ts2;
  float* Fr = F + el * F_STRIDE;
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;

  for (;;) {
    int tile = 0;
    if (lane == 0) tile = atomicAdd(A.counter, 1);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (tile >= ts4) break;
    const int g = (tile >= ts1) + (tile >= ts2) + (tile >= ts3);
    const int tstart = g == 0 ? 0 : (g == 1 ? ts1 : (g == 2 ? ts2 : ts3));
    const int gbeg = g == 0 ? go0 : (g == 1 ? go1 : (g == 2 ? go2 : go3));
    const int gend = g == 0 ? go1 : (g == 1 ? go2 : (g == 2 ? g2_end : go4));
    const int e0 = gbeg + 32 * (tile - tstart);
    const int nvalid = min(32, gend - e0);
    const bool valid = el < nvalid;
    const int e = e0 + min(el, nvalid - 1);
    const int sn = A.src[e], dn = A.dst[e];

    // ---- segmented-scan control words (identical for every output channel of this edge tile) ----
    bool m1, m2, m4, m8, m16, tail;
    {
      const int prev = __shfl_up(sn, 1, 32);
      const int next = __shfl_down(sn, 1, 32);
      int f = (el == 0) || (prev != sn);
      tail = (el == 31) || (next != sn);
      int fu;
      fu = __shfl_up(f, 1, 32);  m1 = (el >= 1) && !f;   if (m1) f |= fu;
      fu = __shfl_up(f, 2, 32);  m2 = (el >= 2) && !f;   if (m2) f |= fu;
      fu = __shfl_up(f, 4, 32);  m4 = (el >= 4) && !f;   if (m4) f |= fu;
      fu = __shfl_up(f, 8, 32);  m8 = (el >= 8) && !f;   if (m8) f |= fu;
      fu = __shfl_up(f, 16, 32); m16 = (el >= 16) && !f; (void)fu;
    }

    // ---- GEMM1: h = relu(W1 [edge_emb | x_src[:ns] | x_dst[:ns]] + b1), K order kappa(s,hh) = 24*(s/12)+12*hh+s%12 ----
    float bin[36];
    {
      const float *pe, *pxs, *pxd;
      if (GATHER) {
        pe = A.edge_attr + (size_t)e * NS + 12 * hh;
        pxs = A.x + (size_t)sn * XW + 12 * hh;
        pxd = A.x + (size_t)dn * XW + 12 * hh;
      } else {
        pe = A.edge_attr + (size_t)e * NE + 12 * hh;
        pxs = pe + NS;
        pxd = pe + 2 * NS;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float4 a = ld4(pe + 4 * j), b = ld4(pxs + 4 * j), c = ld4(pxd + 4 * j);
        bin[4 * j + 0] = a.x; bin[4 * j + 1] = a.y; bin[4 * j + 2] = a.z; bin[4 * j + 3] = a.w;
        bin[12 + 4 * j + 0] = b.x; bin[12 + 4 * j + 1] = b.y; bin[12 + 4 * j + 2] = b.z; bin[12 + 4 * j + 3] = b.w;
        bin[24 + 4 * j + 0] = c.x; bin[24 + 4 * j + 1] = c.y; bin[24 + 4 * j + 2] = c.z; bin[24 + 4 * j + 3] = c.w;
      }
    }
    float h[36];
    {
      const float* w1 = A.w1p + (size_t)g * (3 * 9 * 64 * 4);
      const float* b1 = A.b1p + (size_t)g * (3 * 2 * 16);
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        f32x16 acc;
        const float* bp = b1 + (T * 2 + hh) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 b = ld4(bp + 4 * j);
          acc[4 * j + 0] = b.x; acc[4 * j + 1] = b.y; acc[4 * j + 2] = b.z; acc[4 * j + 3] = b.w;
        }
        const float* wp = w1 + ((size_t)T * 9 * 64 + lane) * 4;
#pragma unroll
        for (int s4 = 0; s4 < 9; ++s4) {
          const float4 a = ld4(wp + s4 * 64 * 4);
          acc = MFMA(a.x, bin[4 * s4 + 0], acc);
          acc = MFMA(a.y, bin[4 * s4 + 1], acc);
          acc = MFMA(a.z, bin[4 * s4 + 2], acc);
          acc = MFMA(a.w, bin[4 * s4 + 3], acc);
        }
        if (T < 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) h[16 * T + r] = fmaxf(acc[r], 0.0f);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) h[32 + r] = fmaxf(acc[r], 0.0f);
        }
      }
    }

    // ---- F row of this edge: TP row operands derived from x[dst] and sh (written by both lane halves) ----
    const float4 shv = ld4(A.sh + (size_t)e * 4);
    const float s0 = shv.x, vx = shv.y, vy = shv.z, vz = shv.w;
    {
      const float* xr = A.x + (size_t)dn * XW;
      // half 0: a -> F_A, p: dot -> F_PV, p*s0 -> T1O[0..], (p x v)/sqrt2 -> T1E[0..]
      // half 1: c -> F_C, q: dot -> F_QV, (q x v)/sqrt2 -> T1O[3nv..], q*s0 -> T1E[3nv..]
      const int o_main_src = hh ? OFF_C : 0, o_main_dst = hh ? F_C : F_A;
      const int o_vec_src = hh ? OFF_Q : OFF_P, o_dot = hh ? F_QV : F_PV;
      const int o_s0 = hh ? (F_T1E + 3 * NV) : F_T1O, o_cross = hh ? (F_T1O + 3 * NV) : F_T1E;
#pragma unroll
      for (int j = 0; j < NS / 4; ++j) *reinterpret_cast<float4*>(Fr + o_main_dst + 4 * j) = ld4(xr + o_main_src + 4 * j);
      float pv[3 * NV];
#pragma unroll
      for (int j = 0; j < 3 * NV / 2; ++j) {
        const float2 t = ld2(xr + o_vec_src + 2 * j);
        pv[2 * j] = t.x; pv[2 * j + 1] = t.y;
      }
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const float px = pv[3 * m], py = pv[3 * m + 1], pz = pv[3 * m + 2];
        Fr[o_dot + m] = (px * vx + py * vy + pz * vz) * inv_s3;
        Fr[o_s0 + 3 * m + 0] = px * s0;
        Fr[o_s0 + 3 * m + 1] = py * s0;
        Fr[o_s0 + 3 * m + 2] = pz * s0;
        Fr[o_cross + 3 * m + 0] = (py * vz - pz * vy) * inv_s2;
        Fr[o_cross + 3 * m + 1] = (pz * vx - px * vz) * inv_s2;
        Fr[o_cross + 3 * m + 2] = (px * vy - py * vx) * inv_s2;
      }
      Fr[o_dot + NV] = 0.0f;
      Fr[o_dot + NV + 1] = 0.0f;
      if (hh) *reinterpret_cast<float4*>(Fr + F_SH) = shv;
    }
    __syncthreads();

    // ---- GEMM2 over the W2 tiles + fused tensor-product epilogue ----
    const float* w2 = A.w2p + (size_t)g * A.n_tiles * (9 * 64 * 4) + (size_t)lane * 4;
    const float* b2 = A.b2p + (size_t)g * A.n_tiles * 32 + hh * 16;
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    // v5 loop: ONE set of A-fragment registers, reloaded in place right after the 4 MFMAs that consumed them
    // (the next tile's fragments stream in under the rest of the burst); the F operands of the tile's 4 units
    // are requested BEFORE the burst; the epilogue arithmetic is branch-free (kinds select multipliers).
    float4 a[9], bn[4];
    UnitQuad un = load_unit_quad(A.units);
#pragma unroll
    for (int s4 = 0; s4 < 9; ++s4) a[s4] = ld4(w2 + s4 * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) bn[j] = ld4(b2 + 4 * j);
    for (int t = 0; t < A.n_tiles; ++t) {
      const UnitQuad uc = un;
      float4 f[4][3];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const float* Fp = Fr + (uc.u[rq].w0 >> 16);
        f[rq][0] = ld4(Fp); f[rq][1] = ld4(Fp + 4); f[rq][2] = ld4(Fp + 8);
      }
      f32x16 D;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        D[4 * j + 0] = bn[j].x; D[4 * j + 1] = bn[j].y; D[4 * j + 2] = bn[j].z; D[4 * j + 3] = bn[j].w;
      }
      const int tn = min(t + 1, A.n_tiles - 1);
      const float* wn = w2 + (size_t)tn * (9 * 64 * 4);
      const float* bp = b2 + (size_t)tn * 32;
#pragma unroll
      for (int s4 = 0; s4 < 9; ++s4) {
        D = MFMA(a[s4].x, h[4 * s4 + 0], D);
        D = MFMA(a[s4].y, h[4 * s4 + 1], D);
        D = MFMA(a[s4].z, h[4 * s4 + 2], D);
        D = MFMA(a[s4].w, h[4 * s4 + 3], D);
        a[s4] = ld4(wn + s4 * 256);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] = ld4(bp + 4 * j);
      un = load_unit_quad(A.units + 4 * tn);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int w0 = uc.u[rq].w0, w1 = uc.u[rq].w1;
        const int kind = w0 & 15, flags = (w0 >> 4) & 15;
        const float d0 = D[4 * rq + 0], d1 = D[4 * rq + 1], d2 = D[4 * rq + 2], d3 = D[4 * rq + 3];
        const float4 f0 = f[rq][0], f1 = f[rq][1], f2 = f[rq][2];
        const float ps = f0.x * d0 + f0.y * d1 + f0.z * d2 + f0.w * d3;
        const float pv0 = f0.x * d0 + f0.w * d1 + f1.z * d2 + f2.y * d3;
        const float pv1 = f0.y * d0 + f1.x * d1 + f1.w * d2 + f2.z * d3;
        const float pv2 = f0.z * d0 + f1.y * d1 + f2.x * d2 + f2.w * d3;
        const bool kS0 = kind == U_R1_S0, kV = kind == U_R1_V, kTS = kind == U_T_S, kTV = kind == U_T_V;
        const float ms0 = kS0 ? s0 : (kV ? vx : (kTS ? 1.0f : 0.0f));
        const float ms1 = kV ? vy : 0.0f, ms2 = kV ? vz : 0.0f, mv = kTV ? 1.0f : 0.0f;
        acc0 = fmaf(ms0, ps, fmaf(mv, pv0, acc0));
        acc1 = fmaf(ms1, ps, fmaf(mv, pv1, acc1));
        acc2 = fmaf(ms2, ps, fmaf(mv, pv2, acc2));
        if (flags & 2) {
          const float scale = uc.u[rq].scale;
          const int ncomp = (w0 >> 8) & 15;
          float* dstp = (g2_shared && g == 2 ? A.sum_g2 + (size_t)(sn - A.g2_node_off) * XW : A.sum + (size_t)sn * XW) +
                        (w1 & 0xffff) + hh * (w1 >> 16);
          float vals[3] = {acc0 * scale, acc1 * scale, acc2 * scale};
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            if (c < ncomp) {
              float xv = valid ? vals[c] : 0.0f;
              float up;
              up = __shfl_up(xv, 1, 32);  if (m1) xv += up;
              up = __shfl_up(xv, 2, 32);  if (m2) xv += up;
              up = __shfl_up(xv, 4, 32);  if (m4) xv += up;
              up = __shfl_up(xv, 8, 32);  if (m8) xv += up;
              up = __shfl_up(xv, 16, 32); if (m16) xv += up;
              if (tail) unsafeAtomicAdd(dstp + c, xv);
            }
          }
          acc0 = acc1 = acc2 = 0.0f;
        }
      }
    }
    __syncthreads();   // F is rewritten by the next edge tile
  }
}

// tile_info[0..4] = prefix of ceil(E_g/32), [5..9] = edge offsets of the groups, [10] = tile counter
__global__ void conv_setup_kernel(int32_t* tile_info, int g0, int g1, int g2, int g3, int g4) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int go[5] = {g0, g1, g2, g3, g4};
    int ts = 0;
    tile_info[0] = 0;
    for (int g = 0; g < 4; ++g) {
      ts += (go[g + 1] - go[g] + 31) / 32;
      tile_info[g + 1] = ts;
    }
    for (int g = 0; g < 5; ++g) tile_info[5 + g] = go[g];
    tile_info[10] = 0;
  }
}

__global__ void pad_rows_kernel(const float* x, int64_t n, int din, float* xpad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * XW) return;
  const int64_t r = i / XW;
  const int c = (int)(i % XW);
  xpad[i] = c < din ? x[r * din + c] : 0.0f;
}

__global__ void count_deg_kernel(const int32_t* src, int64_t E, int32_t* deg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) atomicAdd(deg + src[i], 1);
}

// scatter 'mean' divisor, e3nn BatchNorm (eval) and the residual of tensor_layers.py:159-166:
//   out = ((sum/max(deg,1) - mean) * scale + bias) + pad(x_in)
__global__ void node_finalize_kernel(const float* sum, const int32_t* deg, const float* x_in, const float* bn_mean,
                                     const float* bn_scale, const float* bn_bias, int64_t n, int dout, int out_stride,
                                     float* out, const float* sum_rr0, int64_t n_lig_total, int n_rec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * out_stride) return;
  const int64_t r = i / out_stride;
  const int c = (int)(i % out_stride);
  float v = 0.0f;
  if (c < dout) {
    const int d = deg[r];
    float sv = sum[r * XW + c];
    if (sum_rr0 != nullptr && r >= n_lig_total) sv += sum_rr0[((r - n_lig_total) % n_rec) * XW + c];   // shared layer-0 rec-rec messages
    v = sv / (float)(d > 1 ? d : 1);
    v = (v - bn_mean[c]) * bn_scale[c] + bn_bias[c];
  }
  if (x_in != nullptr && c < XW) v += x_in[r * XW + c];
  out[i] = v;
}

hipError_t launch_conv_fused(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s) {
  ConvKArgs k;
  k.x = a.x; k.src = a.src; k.dst = a.dst; k.edge_attr = a.edge_attr; k.sh = a.sh; k.sum = a.sum;
  k.tile_info = a.tile_info; k.counter = a.counter;
  k.w1p = L.w1p[0]; k.b1p = L.b1p[0]; k.w2p = L.w2p[0]; k.b2p = L.b2p[0]; k.units = L.units; k.n_tiles = L.n_tiles;
  k.lig_side_only = a.lig_side_only; k.g2_limit = a.g2_limit; k.sum_g2 = a.sum_g2; k.g2_node_off = a.g2_node_off;
  const int grid = n_cu * 8;   // 8 single-wave workgroups per CU (2 per SIMD), persistent, dynamic tile queue
  if (a.gather)
    hipLaunchKernelGGL(conv_fused_kernel<true>, dim3(grid), dim3(64), 0, s, k);
  else
    hipLaunchKernelGGL(conv_fused_kernel<false>, dim3(grid), dim3(64), 0, s, k);
  return hipGetLastError();
}

hipError_t launch_conv_setup(int32_t* tile_info, const int64_t* go, hipStream_t s) {
  hipLaunchKernelGGL(conv_setup_kernel, dim3(1), dim3(64), 0, s, tile_info, (int)go[0], (int)go[1], (int)go[2], (int)go[3],
                     (int)go[4]);
  return hipGetLastError();
}

hipError_t launch_pad_rows(const float* x, int64_t n, int din, float* xpad, hipStream_t s) {
  const int64_t tot = n * XW;
  if (tot == 0) return hipSuccess;
  hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, x, n, din, xpad);
  return hipGetLastError();
}

hipError_t launch_count_deg(const int32_t* src, int64_t E, int32_t* deg, hipStream_t s) {
  if (E == 0) return hipSuccess;
  hipLaunchKernelGGL(count_deg_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, src, E, deg);
  return hipGetLastError();
}

hipError_t launch_node_finalize(const float* sum, const int32_t* deg, const float* x_in, const float* bn_mean,
                                const float* bn_scale, const float* bn_bias, int64_t n, int dout, int out_stride,
                                float* out, hipStream_t s, const float* sum_rr0, int64_t n_lig_total, int n_rec) {
  const int64_t tot = n * out_stride;
  if (tot == 0) return hipSuccess;
  hipLaunchKernelGGL(node_finalize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, sum, deg, x_in, bn_mean,
                     bn_scale, bn_bias, n, dout, out_stride, out, sum_rr0, n_lig_total, n_rec);
  return hipGetLastError();
}

}  // namespace ddk
