// Fused tensor-product convolution, error-compensated 3 x f16 variant (ddk_config.conv_f16x3 = 1).
//
// Same algorithm, tile tables, LDS layout and epilogue as k_conv.hip; only the two radial-MLP GEMMs change: every fp32 operand is
// split as x = x_hi + x_lo / 2^11 with x_hi = fp16(x), x_lo = fp16((x - x_hi) 2^11) (weights at pack time, activations in registers)
// and the product is formed on the f16 matrix pipe as
//        W.x = W_hi.x_hi  +  (W_hi.x_lo + W_lo.x_hi) / 2^11          (v_mfma_f32_32x32x16_f16, fp32 accumulators)
// The products of two fp16 numbers are exact in fp32; the dropped W_lo.x_lo term is 2^-22 relative, so the result carries fp32-level
// accuracy (tools/probes/mfma_probe6.hip: 1.1e-7 vs 1.9e-7 for the fp32 MFMA chain, against fp64).  15 MFMAs of 32 cycles replace
// 36 of 64 per tile, and the f16 (XDL) pipe does not share issue time with the VALU the way the fp32 MFMA does.
// K = 72 is padded to 80: register 8*s+i (< 36) of a lane half is element i of MFMA step s; slot 36 of lane half 0 carries the
// constant 1 and the bias as its weight column (a bias read from the ring would race with the publication of tile t+2).
#include <stdlib.h>

#include "k_conv_common.h"

namespace ddk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// staged tile record -> registers: hi / lo fragments (lane-contiguous 16-B reads), bias of this lane half
__device__ __forceinline__ void lds_frags_h(f16x8 (&ah)[5], f16x8 (&al)[5], const float* stage, int lane) {
  const char* st = reinterpret_cast<const char*>(stage);
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    ah[s] = *reinterpret_cast<const f16x8*>(st + (s * 64 + lane) * 16);
    al[s] = *reinterpret_cast<const f16x8*>(st + W2H_FRAG_BYTES + (s * 64 + lane) * 16);
  }
}


__device__ __forceinline__ f32x16 burst_h(const f16x8 (&ah)[5], const f16x8 (&al)[5], const f16x8 (&hhi)[5], const f16x8 (&hlo)[5],
                                          int /*unused*/) {
  const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  f32x16 Dm = MFMA16(ah[0], hhi[0], zero);
  f32x16 Dc = MFMA16(ah[0], hlo[0], zero);
  Dc = MFMA16(al[0], hhi[0], Dc);
#pragma unroll
  for (int s = 1; s < 5; ++s) {
    Dm = MFMA16(ah[s], hhi[s], Dm);
    Dc = MFMA16(ah[s], hlo[s], Dc);
    Dc = MFMA16(al[s], hhi[s], Dc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) Dm[r] = fmaf(Dc[r], 1.0f / 2048.0f, Dm[r]);
  return Dm;
}

// Exact power-of-two range scaling of the split operands: returns 2^(13 - floor(log2 max(m, 2^-40))), i.e. max|x| lands in
// [2^13, 2^14) -- inside the fp16 range, with the 11 + 11 bits of the hi/lo split above the fp16 subnormals for everything within
// 2^-27 of the maximum -- and its inverse.  Without it a hidden unit above 65504 would overflow and tiny checkpoints would lose bits.
__device__ __forceinline__ float range_scale(float m, float& inv) {
  const uint32_t eb = max((__float_as_uint(m) >> 23) & 0xffu, 87u);   // biased exponent; 0 and subnormals map to the 2^-40 floor
  inv = __uint_as_float((eb - 13u) << 23);
  return __uint_as_float((267u - eb) << 23);
}

// scalar-accumulator epilogue (the f16 pipe does not compete with the VALU for issue: fewer registers beat fewer instructions here)
__device__ __forceinline__ void tile_epilogue_s(int kind, const f32x16& D, const float* Fp, f32x4 f0, float (&accA)[4], float (&accV)[4][3]) {
  if (kind == T_TV) {
    const f32x4 f1 = ldv4(Fp + 4), f2 = ldv4(Fp + 8);      // y / z components of the 4 feature rows
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const float d0 = D[4 * rq], d1 = D[4 * rq + 1], d2 = D[4 * rq + 2], d3 = D[4 * rq + 3];
      accV[rq][0] = fmaf(f0.x, d0, fmaf(f0.y, d1, fmaf(f0.z, d2, fmaf(f0.w, d3, accV[rq][0]))));
      accV[rq][1] = fmaf(f1.x, d0, fmaf(f1.y, d1, fmaf(f1.z, d2, fmaf(f1.w, d3, accV[rq][1]))));
      accV[rq][2] = fmaf(f2.x, d0, fmaf(f2.y, d1, fmaf(f2.z, d2, fmaf(f2.w, d3, accV[rq][2]))));
    }
  } else if (kind == T_RA) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      accA[rq] = fmaf(f0.x, D[4 * rq], fmaf(f0.y, D[4 * rq + 1], fmaf(f0.z, D[4 * rq + 2], fmaf(f0.w, D[4 * rq + 3], accA[rq]))));
  } else {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      accV[rq][0] = fmaf(f0.x, D[4 * rq], fmaf(f0.y, D[4 * rq + 1], fmaf(f0.z, D[4 * rq + 2], fmaf(f0.w, D[4 * rq + 3], accV[rq][0]))));
  }
}

template <bool GATHER>
__global__ __launch_bounds__(64 * CONV_WAVES) void conv_fused_h_kernel(ConvKArgs A) {
  constexpr int MODE = 0, WAVES = CONV_WAVES, FS = F_STRIDE, BLOCK_EDGES = 32 * WAVES;
  constexpr int STAGE_F = W2H_TILE_BYTES / 4;                  // floats per ring stage
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* F = lds + wave * (32 * FS);                          // this wave's 32 F rows
  float* ring = lds + WAVES * (32 * FS);                      // [2][W2H_TILE_BYTES]
  int* blk_slot = reinterpret_cast<int*>(ring + 2 * STAGE_F);
  const int el = lane & 31;
  const int hh = lane >> 5;
  const bool g2_shared = A.sum_g2 != nullptr;                 // group 2 is the shared rec-rec copy (layer-0 de-duplication): its own accumulator
  // work unit: a block of BLOCK_EDGES consecutive edges of ONE edge group (its radial-MLP weights are shared by the workgroup)
  // lane g < n_active keeps group g's edge range and its block range [pbeg, pend) of the work queue; a block index is mapped to
  // its group with one ballot (no dependent scalar loads per block)
  int gb_v = 0, ge_v = 0;
  if (lane < A.n_active) {
    gb_v = A.gbeg[lane];
    ge_v = A.gend[lane];
  }
  const int nb_v = (ge_v - gb_v + BLOCK_EDGES - 1) / BLOCK_EDGES;
  int pend_v = nb_v;
#pragma unroll
  for (int d = 1; d < 16; d *= 2) {
    const int t = __shfl_up(pend_v, d, 64);
    if (lane >= d) pend_v += t;
  }
  const int pbeg_v = pend_v - nb_v;
  const int bs4 = __builtin_amdgcn_readlane(pend_v, 15);
  float* Fr = F + el * FS;
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  const int n_tiles = A.n_tiles;
  constexpr int REC4 = W2H_TILE_BYTES / 16;                   // 649 x 16 B per tile record
  const bool second = tid < REC4 - 64 * WAVES;                 // threads that move a second float4 of the record

  // work units: whole blocks, except that the last (bs4 mod #workgroups) blocks are split into column chunks so that the
  // final round of the persistent workgroups is a fraction of a block long (tail of the dynamic queue)
  const int nwg = gridDim.x;
  const int full = bs4 >= nwg ? (bs4 / nwg) * nwg : 0;
  const int rest = bs4 - full;
  // column chunks per block of the tail: the value that minimises (rounds of the persistent workgroups over the chunk units) / split, with 5 %
  // per extra chunk for the GEMM1 + gather prologue every chunk repeats (rest = 131 of 256: one whole extra round unsplit, 0.7 of one in 5 chunks)
  int split = 1;
  if (rest > 0) {
    float best = 1e30f;
    for (int sp = 1; sp <= A.n_cols; ++sp) {
      const float cost = (float)((rest * sp + nwg - 1) / nwg) / (float)sp * (1.0f + 0.05f * (float)(sp - 1));
      if (cost < best - 1e-6f) { best = cost; split = sp; }
    }
  }
  const int n_units = full + rest * split;

  // work queue: the first unit of workgroup w is w itself; every later one comes from the device counter (zeroed by the caller),
  // fetched by thread 0 at the START of the previous unit so that the atomic's round trip hides under that unit's tile loop
  int unit = blockIdx.x;
  for (;;) {
    if (unit >= n_units) break;
    int unit_next = 0;
    if (tid == 0) unit_next = nwg + atomicAdd(A.counter, 1);
    int blk = unit, t_begin = 0, t_end = n_tiles;
    if (unit >= full) {
      const int r = unit - full, c = r % split;
      blk = full + r / split;
      t_begin = A.col_start[(c * A.n_cols) / split];
      t_end = A.col_start[((c + 1) * A.n_cols) / split];
    }
    const int g = __popcll(__ballot(lane < 16 && blk >= pend_v));
    const int gbeg = __builtin_amdgcn_readlane(gb_v, g), gend = __builtin_amdgcn_readlane(ge_v, g);
    const int bstart = __builtin_amdgcn_readlane(pbeg_v, g);
    const int e0 = gbeg + BLOCK_EDGES * (blk - bstart) + 32 * wave;
    const int nvalid = min(32, gend - e0);                    // <= 0: this wave's slice lies past the end of the group
    const bool valid = el < nvalid;
    const int e = nvalid > 0 ? e0 + min(el, nvalid - 1) : gend - 1;
    const int sn = A.src[e], dn = A.dst[e];

    // ---- stage the first two W2 tiles of this unit (the ring is idle: the previous block ended with a barrier) ----
    const float* wrec = reinterpret_cast<const float*>(A.w2h + (size_t)g * n_tiles * W2H_TILE_BYTES);
    {
      const float* wr0 = wrec + (size_t)t_begin * STAGE_F;
      const float* wr1 = wrec + (size_t)min(t_begin + 1, t_end - 1) * STAGE_F;
      const float4 r0 = ld4(wr0 + 4 * tid), r1 = ld4(wr1 + 4 * tid);
      *reinterpret_cast<float4*>(ring + 4 * tid) = r0;
      *reinterpret_cast<float4*>(ring + STAGE_F + 4 * tid) = r1;
      if (second) {
        const int q = 4 * (tid + 64 * WAVES);
        const float4 r2 = ld4(wr0 + q), r3 = ld4(wr1 + q);
        *reinterpret_cast<float4*>(ring + q) = r2;
        *reinterpret_cast<float4*>(ring + STAGE_F + q) = r3;
      }
    }

    // ---- segmented-scan control words (identical for every output channel of this wave's 32 edges) ----
    const SegCtl seg = make_segctl(sn, el, nvalid, valid);

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
    // h = relu(W1 in + b1) as an error-compensated 3 x f16 product (see the header)
    f16x8 hhi[5], hlo[5];
    float osc;        // GEMM2 outputs of this edge are (s2 h) x (w2s W2): the factor that takes a flushed sum back
    {
      f16x8 bhi[5], blo[5];
      float m1 = 0.0f;
#pragma unroll
      for (int j = 0; j < 36; ++j) m1 = fmaxf(m1, fabsf(bin[j]));
      m1 = fmaxf(m1, __shfl_xor(m1, 32));          // both lane halves hold K slices of the same edge
      float inv1;
      const float s1 = range_scale(m1, inv1);
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float v = 8 * s + i < 36 ? bin[(8 * s + i) % 36] * s1 : 0.0f;
          const _Float16 t = (_Float16)v;
          bhi[s][i] = t; blo[s][i] = (_Float16)((v - (float)t) * 2048.0f);
        }
      const uint16_t* w1 = A.w1h + (size_t)g * 3 * (W1H_TILE_BYTES / 2);
      const float* b1 = A.b1p + (size_t)g * (3 * 2 * 16);
      const float bsc = s1 * A.w1s[g], usc = inv1 * A.w1u[g];   // the accumulators hold (s1 in) x (w1s W1): bias in, result out of that scale
      float h[36];
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        f32x16 am, ac;
        const float* bp = b1 + (T * 2 + hh) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 b = ld4(bp + 4 * j);
          am[4 * j + 0] = b.x * bsc; am[4 * j + 1] = b.y * bsc; am[4 * j + 2] = b.z * bsc; am[4 * j + 3] = b.w * bsc;
          ac[4 * j + 0] = 0.0f; ac[4 * j + 1] = 0.0f; ac[4 * j + 2] = 0.0f; ac[4 * j + 3] = 0.0f;
        }
        const uint16_t* wt = w1 + (size_t)T * (W1H_TILE_BYTES / 2) + lane * 8;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const f16x8 ah = *reinterpret_cast<const f16x8*>(wt + s * 512);
          const f16x8 al = *reinterpret_cast<const f16x8*>(wt + W2H_FRAG_BYTES / 2 + s * 512);
          am = MFMA16(ah, bhi[s], am);
          ac = MFMA16(ah, blo[s], ac);
          ac = MFMA16(al, bhi[s], ac);
        }
        if (T < 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) h[16 * T + r] = fmaxf(fmaf(ac[r], 1.0f / 2048.0f, am[r]), 0.0f) * usc;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) h[32 + r] = fmaxf(fmaf(ac[r], 1.0f / 2048.0f, am[r]), 0.0f) * usc;
        }
      }
      float m2 = 1.0f;                                 // the constant 1 of the bias slot takes part in the range
#pragma unroll
      for (int j = 0; j < 36; ++j) m2 = fmaxf(m2, h[j]);
      m2 = fmaxf(m2, __shfl_xor(m2, 32));
      float inv2;
      const float s2 = range_scale(m2, inv2);
      osc = inv2 * A.w2u[g];
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // K slot 36 of lane half 0 is the constant 1: its weight column holds the bias of the tile row (K = 72 is padded to 80 anyway)
          const float v = 8 * s + i < 36 ? h[(8 * s + i) % 36] * s2 : ((8 * s + i == 36 && hh == 0) ? s2 : 0.0f);
          const _Float16 t = (_Float16)v;
          hhi[s][i] = t; hlo[s][i] = (_Float16)((v - (float)t) * 2048.0f);
        }
    }

    // ---- F row of this edge: TP row operands derived from x[dst] and sh (written by both lane halves) ----
    const float4 shv = ld4(A.sh + (size_t)e * 4);
    const float s0 = shv.x, vx = shv.y, vy = shv.z, vz = shv.w;
    {
      const float* xr = A.x + (size_t)dn * XW;
      // half 0: a -> F_A, p: (p.v) -> F_PQ, p*s0 -> rows 0..nv-1 of T1O, (p x v)/sqrt2 -> rows 0..nv-1 of T1E
      // half 1: c -> F_C, q: (q.v) -> F_PQ, q*s0 -> rows nv.. of T1E, (q x v)/sqrt2 -> rows nv.. of T1O
      const int o_main_src = hh ? OFF_C : 0, o_main_dst = hh ? F_C : F_A;
      const int o_vec_src = hh ? OFF_Q : OFF_P;
      const int o_vs = hh ? F_T1E : F_T1O, o_vc = hh ? F_T1O : F_T1E, r0 = hh ? NV : 0;
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
        // F_PQ = [pv0..3 | qv0..3 | pv4 pv5 qv4 qv5]
        Fr[F_PQ + (m < 4 ? 4 * hh + m : 8 + 2 * hh + (m - 4))] = (px * vx + py * vy + pz * vz) * inv_s3;
        // vector parts: row r of the 12-row part lives at 12*(r/4) + 4*c + r%4 (component-major inside a quad of rows)
        const int r = r0 + m;
        float* Ps = Fr + o_vs + 12 * (r >> 2) + (r & 3);
        float* Pc = Fr + o_vc + 12 * (r >> 2) + (r & 3);
        Ps[0] = px * s0;
        Ps[4] = py * s0;
        Ps[8] = pz * s0;
        Pc[0] = (py * vz - pz * vy) * inv_s2;
        Pc[4] = (pz * vx - px * vz) * inv_s2;
        Pc[8] = (px * vy - py * vx) * inv_s2;
        if (MODE == 1) {   // 1o(x)2e->1o / 1e(x)2e->1e: (v^ v^T - |v^|^2 I/3) p with v^ = sh[1:4]/sqrt3  (constants folded into the packed weights)
          // v = sqrt3 v^ ; |v^| is 1, or 0 for a zero-length edge (a C-alpha atom and its own residue: Y2 = 0 there)
          const float dv = (px * vx + py * vy + pz * vz) * (1.0f / 3.0f);
          const float n3 = (vx * vx + vy * vy + vz * vz) * (1.0f / 9.0f);       // |v^|^2 / 3
          float* P2 = Fr + (hh ? F_T2E : F_T2O) + 12 * (m >> 2) + (m & 3);
          P2[0] = dv * vx - px * n3;
          P2[4] = dv * vy - py * n3;
          P2[8] = dv * vz - pz * n3;
        }
      }
      if (MODE == 1) {     // pad rows 6,7 of the second quad (their weights are zero; keep them finite)
        float* P2 = Fr + (hh ? F_T2E : F_T2O) + 12;
        P2[2] = 0.0f; P2[3] = 0.0f; P2[6] = 0.0f; P2[7] = 0.0f; P2[10] = 0.0f; P2[11] = 0.0f;
      }
    }
    __syncthreads();   // ring stages 0/1 and the F rows are visible

    // ---- GEMM2 over the W2 tiles + fused tensor-product epilogue ----
    float* const node_row = (g2_shared && g == 2) ? A.sum_g2 + (size_t)(sn - A.g2_node_off) * XW
                                                  : A.sum + ((size_t)sn * A.n_slots + ((A.slots >> (2 * g)) & 3)) * XW;
    float accA[4], accV[4][3];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) { accA[rq] = 0.0f; accV[rq][0] = 0.0f; accV[rq][1] = 0.0f; accV[rq][2] = 0.0f; }
    f16x8 a0h[5], a0l[5], a1h[5], a1l[5];
    lds_frags_h(a0h, a0l, ring, lane);
    int2 tqv = *reinterpret_cast<const int2*>(ring + (2 * W2H_FRAG_BYTES + 128) / 4);
    // every wave must have taken tile t_begin out of stage 0 before the first iteration's publication of tile t_begin+2 overwrites it
    // (a fast wave reaches that store one burst after this point; the second wave of a SIMD can still be reading)
    __syncthreads();
    TileQ tq;
    tq.w0 = __builtin_amdgcn_readfirstlane(tqv.x); tq.chan0 = __builtin_amdgcn_readfirstlane(tqv.y);
#define PSUM(p) (p)
#define DDK_EPILOGUE tile_epilogue_s(w0 & 3, D, Fp, f0, accA, accV);
#define DDK_FLUSH_COND ((w0 >> 2) & 3)
#define DDK_STAGE_LOAD const float4 st0 = ld4(rec2 + 4 * tid); \
      float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f);   /* (a copy of st0 here would wait for the load) */ \
      if (second) st1 = ld4(rec2 + 4 * (tid + 64 * WAVES));
#define DDK_STAGE_STORE *reinterpret_cast<float4*>(stg + 4 * tid) = st0; \
      if (second) *reinterpret_cast<float4*>(stg + 4 * (tid + 64 * WAVES)) = st1;
#define DDK_FRAGS(ANH, ANL, T) lds_frags_h(ANH, ANL, ring + (((T) + 1 - t_begin) & 1) * STAGE_F, lane);
#define DDK_TILE_BARRIER __syncthreads();
// one W2 tile: (1) request this thread's share of tile t+2 from L2 and the descriptor of tile t+1, (2) read tile t+1's
// fragments from the ring into the other register set, (3) the uninterrupted 36-MFMA burst of tile t, (4) epilogue and,
// at the end of a column, the flush, (5) publish tile t+2 into the ring stage tile t came from, (6) barrier.
#define DDK_TILE(T, ACH, ACL, ANH, ANL) \
    { \
      const int w0 = tq.w0, chan0 = tq.chan0; \
      const int t2 = min((T) + 2, t_end - 1); \
      const float* rec2 = wrec + (size_t)t2 * STAGE_F; \
      DDK_STAGE_LOAD \
      const float* Fp = Fr + (w0 >> 16); \
      const f32x4 f0 = ldv4(Fp); \
      DDK_FRAGS(ANH, ANL, T) \
      tqv = *reinterpret_cast<const int2*>(ring + (((T) + 1 - t_begin) & 1) * STAGE_F + (2 * W2H_FRAG_BYTES + 128) / 4); \
      __builtin_amdgcn_sched_barrier(0); \
      const f32x16 D = burst_h(ACH, ACL, hhi, hlo, 0); \
      __builtin_amdgcn_sched_barrier(0); \
      DDK_EPILOGUE \
      const int fl = DDK_FLUSH_COND; \
      if (fl) { \
        const int nrq = (w0 >> 4) & 7; \
        _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) { \
          if (rq < nrq) { \
            if (fl == FL_S) { \
              seg_add(node_row + chan0 + 2 * rq + hh, osc * fmaf(PSUM(accA[rq]), s0, PSUM(accV[rq][0])), seg); \
            } else { \
              float* d = node_row + chan0 + 3 * (2 * rq + hh); \
              const float sa = osc * PSUM(accA[rq]); \
              seg_add(d + 0, fmaf(sa, vx, osc * PSUM(accV[rq][0])), seg); \
              seg_add(d + 1, fmaf(sa, vy, osc * PSUM(accV[rq][1])), seg); \
              seg_add(d + 2, fmaf(sa, vz, osc * PSUM(accV[rq][2])), seg); \
            } \
          } \
          accA[rq] = 0.0f; accV[rq][0] = 0.0f; accV[rq][1] = 0.0f; accV[rq][2] = 0.0f; \
        } \
      } \
      float* stg = ring + (((T) - t_begin) & 1) * STAGE_F; \
      DDK_STAGE_STORE \
      tq.w0 = __builtin_amdgcn_readfirstlane(tqv.x); tq.chan0 = __builtin_amdgcn_readfirstlane(tqv.y); \
      DDK_TILE_BARRIER \
    }
    for (int t = t_begin; t < t_end; t += 2) {
      DDK_TILE(t, a0h, a0l, a1h, a1l)
      if (t + 1 >= t_end) break;
      DDK_TILE(t + 1, a1h, a1l, a0h, a0l)
    }
#undef DDK_TILE
#undef PSUM
#undef DDK_EPILOGUE
#undef DDK_FLUSH_COND
#undef DDK_TILE_BARRIER
#undef DDK_STAGE_LOAD
#undef DDK_STAGE_STORE
#undef DDK_FRAGS
    // hand the next unit to the workgroup; this barrier also retires the ring (every wave has finished reading it) before the next
    // unit's staging writes
    if (tid == 0) *blk_slot = unit_next;
    __syncthreads();
    unit = __builtin_amdgcn_readfirstlane(*blk_slot);
  }
}


template <bool GATHER>
static hipError_t launch_h_t(const ConvKArgs& k, int n_cu, hipStream_t s) {
  hipLaunchKernelGGL((conv_fused_h_kernel<GATHER>), dim3(n_cu), dim3(64 * CONV_WAVES), CONV_H_LDS_BYTES, s, k);
  return hipGetLastError();
}

hipError_t conv_prepare_device_h() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fused_h_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)CONV_H_LDS_BYTES);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fused_h_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)CONV_H_LDS_BYTES);
  return e;
}

hipError_t launch_conv_fused_h(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s) {
  ConvKArgs k;
  k.x = a.x; k.src = a.src; k.dst = a.dst; k.edge_attr = a.edge_attr; k.sh = a.sh; k.sum = a.sum;
  k.counter = a.counter;
  k.w1p = L.w1p[0]; k.b1p = L.b1p[0]; k.w2r = nullptr; k.w1h = L.w1h; k.w2h = L.w2h; k.n_tiles = L.n_tiles;
  for (int g = 0; g < 4; ++g) { k.w1s[g] = L.w1s[g]; k.w1u[g] = 1.0f / L.w1s[g]; k.w2u[g] = 1.0f / L.w2s[g]; }
  k.n_cols = L.n_cols;
  for (int c = 0; c <= L.n_cols; ++c) k.col_start[c] = L.col_start[c];
  k.sum_g2 = a.sum_g2; k.g2_node_off = a.g2_node_off; k.pre = nullptr; k.part = nullptr;
  k.n_groups = 4; k.n_active = a.n_active; k.n_slots = 1; k.slots = 0; k.wmap = 0x3210ull;
  if (a.gbeg) { k.gbeg = a.gbeg; k.gend = a.gend; }
  else { k.gbeg = a.tile_info + 5; k.gend = a.tile_info + 6; }
  return a.gather ? launch_h_t<true>(k, n_cu, s) : launch_h_t<false>(k, n_cu, s);
}

}  // namespace ddk
