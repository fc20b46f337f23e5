// Fused tensor-product convolution for gfx950 (MI355X): the star kernel of the hot path.
//
// Replaces TensorProductConvLayer.forward (reference models/tensor_layers.py:147-168) with the
// FasterTensorProduct of :65-116 for one conv layer:
//     w   = fc[g](edge_attr)          Linear(72,72) + ReLU + Linear(72,W)      [E, W] never hits HBM
//     m   = TP(x[dst], sh, w)         l<=1 Clebsch-Gordan products, dense contraction with w
//     sum = scatter_sum(m, src)       wave-level segmented reduction + fp32 atomics on the segment tails
// (mean / BatchNorm / residual are applied by node_finalize_kernel below.)
//
// Mapping onto CDNA4 (one persistent 8-wave workgroup per CU; work unit = 256 consecutive edges of one edge group, 32 per wave):
//   * both GEMMs run on the fp32 matrix cores: v_mfma_f32_32x32x2_f32, D[row][col] with
//       rows  = 32 output features of the GEMM (hidden units / per-edge weights)   -> A operand = packed weights
//       cols  = the 32 edges of the wave                                           -> B operand = per-edge activations
//     so lane l always "owns" edge (l & 31); lane-half (l >> 5) selects the K pair of the A/B operands and
//     the 4-row groups of D.  The hidden vector h = relu(W1 e + b1) therefore comes out of GEMM1 already in
//     the register layout GEMM2 needs for its B operand (the K order of GEMM2 is permuted at weight-packing
//     time to make this true) - h never leaves the VGPRs.
//   * on CDNA4 the fp32 MFMA and the VALU share issue time (tools/probes/mfma_probe3.hip: every VALU instruction costs the
//     matrix pipe 1.7-3.7 cycles even when it comes from the other wave of the SIMD), and any instruction placed between two
//     dependent MFMAs breaks the accumulator forwarding.  So a [32 rows x 72] W2 tile is ONE uninterrupted burst of 36
//     v_mfma_f32_32x32x2_f32, everything else happens at the tile boundary, and the epilogue is as few VALU instructions as
//     the algebra allows:
//       - tile row 8*rq + 4*hh + j = weight of TP input row (row0 + j) for output channel 8*col + 2*rq + hh: all four
//         accumulator quads of a tile consume the SAME four feature rows (one 16-B / 48-B LDS read of the per-edge "F row")
//         and the tile kind (scalar rows -> 2 v_pk_fma per quad, vector rows -> 6) is wave-uniform: one scalar branch per tile;
//       - the a*s0 / a(x)v / c*s0 / c(x)v rows accumulate the plain dot product; s0 or v is applied once per channel when
//         the column is flushed; the 1/sqrt(n_in) of tensor_layers.py:89-92 and the bias are folded into the packed W2 / C operand.
//   * the 9.4 KB record of a W2 tile (fragments + bias + descriptor) is fetched from L2 ONCE per workgroup and handed to the eight
//     waves through a 2-stage LDS ring (private per-wave streams cap the loop near 100 TFLOP/s, probe5): during tile t every
//     thread requests its 16-32 B of tile t+2, the waves read tile t+1 from the ring into a second register set, and the share of
//     tile t+2 is published into the stage tile t came from, one barrier per tile.
//   * when a column (8 output channels) finishes, each value is reduced over runs of equal edge_src inside the wave with a
//     5-step segmented scan and only the run tails issue global fp32 atomics.
//
// Roofline: MFMA-bound (2*72*(72+W) flop per edge vs ~650 B per edge of HBM traffic), see DESIGN.md.
#include <stdlib.h>

#include "k_conv_common.h"
#include "k_node.h"
#include "model.h"

namespace ddk {

__device__ __forceinline__ void load_frags(float4 (&a)[9], f32x16& B, const float* w, const float* b) {
#pragma unroll
  for (int s4 = 0; s4 < 9; ++s4) a[s4] = ld4(w + s4 * 256);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = ld4(b + 4 * j);
    B[4 * j + 0] = v.x; B[4 * j + 1] = v.y; B[4 * j + 2] = v.z; B[4 * j + 3] = v.w;
  }
}

__device__ __forceinline__ f32x16 burst(const float4 (&a)[9], const float (&h)[36], const f32x16& C) {
  f32x16 D = MFMA(a[0].x, h[0], C);
  D = MFMA(a[0].y, h[1], D);
  D = MFMA(a[0].z, h[2], D);
  D = MFMA(a[0].w, h[3], D);
#pragma unroll
  for (int s4 = 1; s4 < 9; ++s4) {
    D = MFMA(a[s4].x, h[4 * s4 + 0], D);
    D = MFMA(a[s4].y, h[4 * s4 + 1], D);
    D = MFMA(a[s4].z, h[4 * s4 + 2], D);
    D = MFMA(a[s4].w, h[4 * s4 + 3], D);
  }
  return D;
}

// LDS fragments of one staged W2 tile -> registers (lane-contiguous 16-B reads: conflict free)
__device__ __forceinline__ void lds_frags(float4 (&a)[9], f32x16& B, const float* stage, int lane, int hh) {
#pragma unroll
  for (int s4 = 0; s4 < 9; ++s4) a[s4] = ld4(stage + s4 * 256 + lane * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = ld4(stage + 2304 + hh * 16 + 4 * j);
    B[4 * j + 0] = v.x; B[4 * j + 1] = v.y; B[4 * j + 2] = v.z; B[4 * j + 3] = v.w;
  }
}

// SPLIT (score model, gather mode): the x[src][:ns] / x[dst][:ns] columns of GEMM1 are per-NODE terms computed once per layer
// (node_finalize_pre_kernel below); GEMM1 here contracts the 24 edge-embedding inputs only and starts from their sum (K 72 -> 24)
template <bool GATHER, int MODE, bool SPLIT, bool DET>
__global__ __launch_bounds__(64 * ConvTraits<MODE>::WAVES) void conv_fused_kernel(ConvKArgs A) {
  static_assert(!SPLIT || (GATHER && MODE == 0), "the GEMM1 split exists for the score model's gather path");
  static_assert(!DET || MODE == 0, "the deterministic scatter exists for the score model");
  constexpr int WAVES = ConvTraits<MODE>::WAVES, FS = ConvTraits<MODE>::FS, BLOCK_EDGES = 32 * WAVES;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* F = lds + wave * (32 * FS);                          // this wave's 32 F rows
  float* ring = lds + WAVES * (32 * FS);                      // [2][W2_TILE_FLOATS]
  int* blk_slot = reinterpret_cast<int*>(ring + 2 * W2_TILE_FLOATS);
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
  int bs4 = __builtin_amdgcn_readlane(pend_v, 15);
  if constexpr (DET) {
    if (A.det_nr > 0) bs4 = A.det_rng[A.det_nr];      // sample-aligned units: the work queue walks the ranges' blocks (k_conv_common.h)
  }
  float* Fr = F + el * FS;
  const float inv_s3 = 0.57735026918962576451f, inv_s2 = 0.70710678118654752440f;
  const int n_tiles = A.n_tiles;
  constexpr int REC4 = W2_TILE_FLOATS / 4;                    // 584 float4 per tile record
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
    int g = __popcll(__ballot(lane < 16 && blk >= pend_v));
    int gbeg = __builtin_amdgcn_readlane(gb_v, g), gend = __builtin_amdgcn_readlane(ge_v, g);
    int bstart = __builtin_amdgcn_readlane(pbeg_v, g);
    if constexpr (DET) {
      if (A.det_nr > 0) {
        const DetRange R_ = det_find(A.det_rng, A.det_nr, blk);
        g = R_.g; gbeg = R_.beg; gend = R_.end; bstart = R_.bstart;
      }
    }
    const int e0 = gbeg + BLOCK_EDGES * (blk - bstart) + 32 * wave;
    const int nvalid = min(32, gend - e0);                    // <= 0: this wave's slice lies past the end of the group
    const bool valid = el < nvalid;
    const int e = nvalid > 0 ? e0 + min(el, nvalid - 1) : gend - 1;
    const int sn = A.src[e], dn = A.dst[e];

    // ---- stage the first two W2 tiles of this unit (the ring is idle: the previous block ended with a barrier) ----
    const int gw = (int)((A.wmap >> (4 * g)) & 15);                 // weight set / node-term roles of this group
    const float* wrec = A.w2r + (size_t)gw * n_tiles * W2_TILE_FLOATS;
    {
      const float* wr0 = wrec + (size_t)t_begin * W2_TILE_FLOATS;
      const float* wr1 = wrec + (size_t)min(t_begin + 1, t_end - 1) * W2_TILE_FLOATS;
      const float4 r0 = ld4(wr0 + 4 * tid), r1 = ld4(wr1 + 4 * tid);
      *reinterpret_cast<float4*>(ring + 4 * tid) = r0;
      *reinterpret_cast<float4*>(ring + W2_TILE_FLOATS + 4 * tid) = r1;
      if (second) {
        const int q = 4 * (tid + 64 * WAVES);
        const float4 r2 = ld4(wr0 + q), r3 = ld4(wr1 + q);
        *reinterpret_cast<float4*>(ring + q) = r2;
        *reinterpret_cast<float4*>(ring + W2_TILE_FLOATS + q) = r3;
      }
    }

    // ---- segmented-scan control words (identical for every output channel of this wave's 32 edges) ----
    const SegCtl seg = make_segctl(sn, el, nvalid, valid);

    // ---- GEMM1: h = relu(W1 [edge_emb | x_src[:ns] | x_dst[:ns]] + b1), K order kappa(s,hh) = 24*(s/12)+12*hh+s%12 ----
    float h[36];
    if constexpr (SPLIT) {
      // W1a edge_emb (12 MFMAs per 32-row tile) on top of the per-node terms (W1b x[src][:ns] + b1) + W1c x[dst][:ns], which arrive in
      // the accumulator's own register order: role slot of the receiving node = g & 1, of the sending node = 2 + (g >> 1)
      float bin[12];
      {
        const float* pe = A.edge_attr + (size_t)e * NS + 12 * hh;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float4 a = ld4(pe + 4 * j);
          bin[4 * j + 0] = a.x; bin[4 * j + 1] = a.y; bin[4 * j + 2] = a.z; bin[4 * j + 3] = a.w;
        }
      }
      const float* ps = A.pre + ((size_t)sn * 4 + (gw & 1)) * NE + 36 * hh;
      const float* pd = A.pre + ((size_t)dn * 4 + 2 + (gw >> 1)) * NE + 36 * hh;
      const float* w1 = A.w1p + (size_t)gw * (3 * 9 * 64 * 4);
#pragma unroll
      for (int T = 0; T < 3; ++T) {
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (T < 2 || j == 0) {
            const float4 u = ld4(ps + 16 * T + 4 * j), w = ld4(pd + 16 * T + 4 * j);
            acc[4 * j + 0] = u.x + w.x; acc[4 * j + 1] = u.y + w.y; acc[4 * j + 2] = u.z + w.z; acc[4 * j + 3] = u.w + w.w;
          } else {
            acc[4 * j + 0] = 0.0f; acc[4 * j + 1] = 0.0f; acc[4 * j + 2] = 0.0f; acc[4 * j + 3] = 0.0f;
          }
        }
        const float* wp = w1 + ((size_t)T * 9 * 64 + lane) * 4;
#pragma unroll
        for (int s4 = 0; s4 < 3; ++s4) {
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
    } else {
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
    {
      const float* w1 = A.w1p + (size_t)gw * (3 * 9 * 64 * 4);
      const float* b1 = A.b1p + (size_t)gw * (3 * 2 * 16);
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
    float* node_row = (g2_shared && g == 2) ? A.sum_g2 + (size_t)(sn - A.g2_node_off) * XW
                                            : A.sum + ((size_t)sn * A.n_slots + ((A.slots >> (2 * g)) & 3)) * XW;
    if constexpr (DET) {
      // runs of equal edge_src are contiguous inside a group: only the tile's first / last run can continue in the neighbouring tile
      if (nvalid > 0) {
        const int sn0 = __shfl(sn, 0, 32), snl = __shfl(sn, nvalid - 1, 32);
        const bool first_cont = e0 > gbeg && A.src[e0 - 1] == sn0;
        const bool last_cont = e0 + nvalid < gend && A.src[e0 + nvalid] == snl;
        float* prow = A.part + ((size_t)(blk * WAVES + wave) * 2) * XW;    // this tile's two partial rows (tile id = global block index x 8 + wave)
        if (sn == sn0 && first_cont) node_row = prow;
        else if (sn == snl && last_cont) node_row = prow + XW;
      }
    }
    f32x2 accA[4], accV[4][3];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) { accA[rq] = 0.0f; accV[rq][0] = 0.0f; accV[rq][1] = 0.0f; accV[rq][2] = 0.0f; }
    float4 a0[9], a1[9];
    f32x16 B0, B1;
    lds_frags(a0, B0, ring, lane, hh);
    int2 tqv = *reinterpret_cast<const int2*>(ring + 2336);
    // every wave must have taken tile t_begin out of stage 0 before the first iteration's publication of tile t_begin+2 overwrites it
    // (a fast wave reaches that store one burst after this point; the second wave of a SIMD can still be reading)
    __syncthreads();
    TileQ tq;
    tq.w0 = __builtin_amdgcn_readfirstlane(tqv.x); tq.chan0 = __builtin_amdgcn_readfirstlane(tqv.y);
#define PSUM(p) ((p).x + (p).y)
#define DDK_EPILOGUE tile_epilogue(w0 & 3, D, f0, f1, f2, accA, accV);
#define DDK_FLUSH_COND ((w0 >> 2) & 3)
#define DDK_TILE_BARRIER __syncthreads();
// one W2 tile: (1) request this thread's share of tile t+2 from L2 and the descriptor of tile t+1, (2) read tile t+1's
// fragments from the ring into the other register set, (3) the uninterrupted 36-MFMA burst of tile t, (4) epilogue and,
// at the end of a column, the flush, (5) publish tile t+2 into the ring stage tile t came from, (6) barrier.
#define DDK_TILE(T, AC, BC, AN, BN)                                                                            \
    {                                                                                                          \
      const int w0 = tq.w0, chan0 = tq.chan0;                                                                  \
      const int t2 = min((T) + 2, t_end - 1);                                                                  \
      const float* rec2 = wrec + (size_t)t2 * W2_TILE_FLOATS;                                                  \
      const float4 st0 = ld4(rec2 + 4 * tid);                                                                  \
      float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f);   /* (a copy of st0 here would wait for the load) */     \
      if (second) st1 = ld4(rec2 + 4 * (tid + 64 * WAVES));                                                    \
      const float* Fp = Fr + (w0 >> 16);                                                                       \
      const f32x4 f0 = ldv4(Fp), f1 = ldv4(Fp + 4), f2 = ldv4(Fp + 8);                                        \
      lds_frags(AN, BN, ring + (((T) + 1 - t_begin) & 1) * W2_TILE_FLOATS, lane, hh);                                    \
      tqv = *reinterpret_cast<const int2*>(ring + (((T) + 1 - t_begin) & 1) * W2_TILE_FLOATS + 2336);          \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      const f32x16 D = burst(AC, h, BC);                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      DDK_EPILOGUE                                                                                             \
      if (w0 & 0x80) {   /* 6-channel column: accumulator quad 3 holds another a / c row quad for channel pair xp */ \
        const f32x4 g0 = ldv4(Fr + ((w0 >> 8) & 0x3c));                                                        \
        const f32x2 xv = __builtin_elementwise_fma(D_HI(D, 3), V_HI(g0), D_LO(D, 3) * V_LO(g0));               \
        const int xp = (w0 >> 8) & 3;                                                                          \
        if (xp == 0) accA[0] += xv; else if (xp == 1) accA[1] += xv; else accA[2] += xv;                       \
      }                                                                                                        \
      const int fl = DDK_FLUSH_COND;                                                                           \
      if (fl) {                                                                                                \
        const int nrq = (w0 >> 4) & 7;                                                                         \
        if (fl == FL_S && nrq == 4) {   /* a full scalar column: its four channels in one pass */              \
          float m4[4];                                                                                         \
          _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) m4[rq] = fmaf(PSUM(accA[rq]), s0, PSUM(accV[rq][0])); \
          seg_add_n<DET, 4>(node_row + chan0 + hh, 2, m4, seg);                                                \
        }                                                                                                      \
        _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) {                                                     \
          if (rq < nrq && !(fl == FL_S && nrq == 4)) {                                                         \
            if (fl == FL_S) {                                                                                  \
              seg_add<DET>(node_row + chan0 + 2 * rq + hh, fmaf(PSUM(accA[rq]), s0, PSUM(accV[rq][0])), seg);       \
            } else {                                                                                           \
              float* d = node_row + chan0 + 3 * (2 * rq + hh);                                                 \
              const float sa = PSUM(accA[rq]);                                                                 \
              float m3[3] = {fmaf(sa, vx, PSUM(accV[rq][0])), fmaf(sa, vy, PSUM(accV[rq][1])), fmaf(sa, vz, PSUM(accV[rq][2]))}; \
              seg_add_n<DET, 3>(d, 1, m3, seg);                                                                \
            }                                                                                                  \
          }                                                                                                    \
          accA[rq] = 0.0f; accV[rq][0] = 0.0f; accV[rq][1] = 0.0f; accV[rq][2] = 0.0f;                         \
        }                                                                                                      \
        if ((w0 & 3) == T_RTS) {   /* rows j = 2,3 of the shared tail open the next (0o) column */              \
          _Pragma("unroll") for (int rq = 0; rq < 4; ++rq) accV[rq][0] = D_HI(D, rq) * V_HI(f0);                \
        }                                                                                                      \
      }                                                                                                        \
      float* stg = ring + (((T) - t_begin) & 1) * W2_TILE_FLOATS;                                              \
      *reinterpret_cast<float4*>(stg + 4 * tid) = st0;                                                         \
      if (second) *reinterpret_cast<float4*>(stg + 4 * (tid + 64 * WAVES)) = st1;                              \
      tq.w0 = __builtin_amdgcn_readfirstlane(tqv.x); tq.chan0 = __builtin_amdgcn_readfirstlane(tqv.y);       \
      DDK_TILE_BARRIER                                                                                         \
    }
    for (int t = t_begin; t < t_end; t += 2) {
      DDK_TILE(t, a0, B0, a1, B1)
      if (t + 1 >= t_end) break;
      DDK_TILE(t + 1, a1, B1, a0, B0)
    }
#undef DDK_TILE
#undef PSUM
#undef DDK_EPILOGUE
#undef DDK_FLUSH_COND
#undef DDK_TILE_BARRIER
    // hand the next unit to the workgroup; this barrier also retires the ring (every wave has finished reading it) before the next
    // unit's staging writes
    if (tid == 0) *blk_slot = unit_next;
    __syncthreads();
    unit = __builtin_amdgcn_readfirstlane(*blk_slot);
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

// group table of a single-conv launch (ddk_conv_forward on an all-atom context): gbeg[g] = 0, gend[g] = E for g == k else 0
__global__ void conv_one_group_kernel(int32_t* gt, int n_groups, int k, int E) {
  const int g = threadIdx.x;
  if (g < n_groups) { gt[g] = 0; gt[n_groups + g] = g == k ? E : 0; }
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
// (clear_sum: every accumulator that is read is zeroed behind the read, so the next layer / forward finds a clean buffer without a
// hipMemsetAsync of [N, 84] per layer; zero_extra: a second buffer to clear - the shared layer-0 rec-rec rows, read by every sample
// of layer 0's finalize and therefore cleared one launch later)
__global__ void node_finalize_kernel(float* sum, const int32_t* deg, const float* x_in, const float* bn_mean,
                                     const float* bn_scale, const float* bn_bias, int64_t n, int dout, int out_stride,
                                     float* out, const float* sum_rr0, int64_t n_lig_total, int n_rec, int clear_sum,
                                     float* zero_extra, int64_t n_extra, int n_slots, const uint8_t* rr0_mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (zero_extra != nullptr && i < n_extra) zero_extra[i] = 0.0f;
  if (i >= n * out_stride) return;
  const int64_t r = i / out_stride;
  const int c = (int)(i % out_stride);
  float v = 0.0f;
  if (c < dout) {
    const int d = deg[r];
    float sv = 0.0f;
    for (int sl = 0; sl < n_slots; ++sl) {      // deterministic mode keeps one accumulator per (node, receiving group): added in group order
      sv += sum[(r * n_slots + sl) * XW + c];
      if (clear_sum) sum[(r * n_slots + sl) * XW + c] = 0.0f;
    }
    if (sum_rr0 != nullptr && r >= n_lig_total && !(rr0_mask != nullptr && rr0_mask[r - n_lig_total]))
      sv += sum_rr0[((r - n_lig_total) % n_rec) * XW + c];   // shared layer-0 rec-rec messages
    v = sv / (float)(d > 1 ? d : 1);
    v = (v - bn_mean[c]) * bn_scale[c] + bn_bias[c];
  }
  if (x_in != nullptr && c < XW) v += x_in[r * XW + c];
  out[i] = v;
}

// the kernel around k_node.h's body: MODE 0 the node terms of the rows of x_out; 1: node_finalize of a layer first; 2: the node embedding first
template <int MODE>
__global__ __launch_bounds__(PRE_W) void node_finalize_pre_kernel(NodePreArgs A, NodeEmbedArgs E) {
  node_finalize_pre_body<MODE>(A, E, (int)blockIdx.x, (int)gridDim.x);
}

hipError_t launch_node_finalize_pre(const NodePreArgs& a, bool finalize, hipStream_t s, const NodeEmbedArgs* embed) {
  const int tiles = node_pre_tiles(a);
  if (tiles == 0) return hipSuccess;
  NodeEmbedArgs e = {};
  if (embed) e = *embed;
  if (finalize) hipLaunchKernelGGL(node_finalize_pre_kernel<1>, dim3(tiles), dim3(PRE_W), 0, s, a, e);
  else if (embed) hipLaunchKernelGGL(node_finalize_pre_kernel<2>, dim3(tiles), dim3(PRE_W), 0, s, a, e);
  else hipLaunchKernelGGL(node_finalize_pre_kernel<0>, dim3(tiles), dim3(PRE_W), 0, s, a, e);
  return hipGetLastError();
}

template <bool GATHER, int MODE, bool SPLIT, bool DET = false>
static hipError_t launch_conv_t(const ConvKArgs& k, int n_cu, hipStream_t s) {
  // one persistent workgroup per CU (its F rows + the W2 ring fill the LDS), dynamic block queue
  hipLaunchKernelGGL((conv_fused_kernel<GATHER, MODE, SPLIT, DET>), dim3(n_cu), dim3(64 * ConvTraits<MODE>::WAVES), conv_lds_bytes<MODE>(), s, k);
  return hipGetLastError();
}

template <bool GATHER, int MODE, bool SPLIT, bool DET = false>
static hipError_t conv_attr_t() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fused_kernel<GATHER, MODE, SPLIT, DET>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)conv_lds_bytes<MODE>());
}

// opt the fused kernels into > 64 KB of dynamic LDS on the CURRENT device (ddk_create calls this after hipSetDevice: the attribute
// is per device, and a failure is reported by that context instead of being cached for the process)
hipError_t conv_prepare_device() {
  hipError_t e = conv_attr_t<true, 0, true>();
  if (e == hipSuccess) e = conv_attr_t<true, 0, false>();
  if (e == hipSuccess) e = conv_attr_t<false, 0, false>();
  if (e == hipSuccess) e = conv_attr_t<true, 1, false>();
  if (e == hipSuccess) e = conv_attr_t<false, 1, false>();
  if (e == hipSuccess) e = conv_attr_t<true, 0, true, true>();
  if (e == hipSuccess) e = conv_attr_t<false, 0, false, true>();
  if (e == hipSuccess) e = conv_prepare_device_x();
  if (e == hipSuccess) e = conv_prepare_device_x2();
  return e;
}

// deterministic mode, after a conv launch: fold the partial rows of every run that straddles 32-edge tiles, in tile order, into the run's
// accumulator row.  One wave per tile; the wave whose tile holds the START of a straddling run walks the chain.
__global__ __launch_bounds__(256) void conv_det_fix_kernel(ConvKArgs A, int dout) {
  const int lane = threadIdx.x & 63;
  int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  int g = 0, gb = 0, ge = 0, bstart = 0;                 // bstart: first 256-edge block of the group (or range) in the launch's work queue
  if (A.det_nr > 0) {                                    // sample-aligned units: tile = (work-queue block, wave) of the conv launch
    const int blk = tile / CONV_WAVES;
    if (blk >= A.det_rng[A.det_nr]) return;
    const DetRange R_ = det_find(A.det_rng, A.det_nr, blk);
    g = R_.g; gb = R_.beg; ge = R_.end; bstart = R_.bstart;
    tile -= bstart * CONV_WAVES;
    if (gb + 32 * tile >= ge) return;
  } else {
    for (; g < A.n_active; ++g) {
      gb = A.gbeg[g]; ge = A.gend[g];
      const int nt = (ge - gb + 31) / 32;
      if (tile < nt) break;
      tile -= nt;
      bstart += (ge - gb + CONV_BLOCK_EDGES - 1) / CONV_BLOCK_EDGES;
    }
    if (g >= A.n_active) return;
  }
  auto prow_of = [&](int t, int which) { return A.part + ((size_t)((bstart + t / CONV_WAVES) * CONV_WAVES + t % CONV_WAVES) * 2 + which) * XW; };
  int e0 = gb + 32 * tile;
  int n = min(32, ge - e0);
  const int s_last = A.src[e0 + n - 1];
  if (!(e0 + n < ge && A.src[e0 + n] == s_last)) return;                         // the last run ends here
  if (A.src[e0] == s_last && e0 > gb && A.src[e0 - 1] == s_last) return;          // ... and did not start here: not the head of its chain
  float acc0 = 0.0f, acc1 = 0.0f;
  const float* prow = prow_of(tile, 1);
  if (lane < dout) acc0 = prow[lane];
  if (lane + 64 < dout) acc1 = prow[lane + 64];
  for (;;) {
    ++tile;
    e0 += 32;
    n = min(32, ge - e0);
    prow = prow_of(tile, 0);
    if (lane < dout) acc0 += prow[lane];
    if (lane + 64 < dout) acc1 += prow[lane + 64];
    if (!(A.src[e0 + n - 1] == s_last && e0 + n < ge && A.src[e0 + n] == s_last)) break;   // the run ends inside this tile
  }
  float* row = (A.sum_g2 != nullptr && g == 2) ? A.sum_g2 + (size_t)(s_last - A.g2_node_off) * XW
                                               : A.sum + ((size_t)s_last * A.n_slots + ((A.slots >> (2 * g)) & 3)) * XW;
  if (lane < dout) row[lane] = acc0;
  if (lane + 64 < dout) row[lane + 64] = acc1;
}

// deterministic mode: the fix-up pass behind a conv launch of either kernel
void conv_det_fix(const ConvKArgs& k, const ConvLaunch& a, int dout, hipStream_t s) {
  const int64_t tiles = a.edge_bound / 32 + 8 * CONV_MAX_GROUPS + (int64_t)CONV_WAVES * a.det_nr;      // (every range may end in a partial block)
  hipLaunchKernelGGL(conv_det_fix_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, k, dout);
}

hipError_t launch_conv_fused(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s) {
  // the default: exact three-limb product on the f16 matrix pipe - the score model's launches and, for the confidence model's layers (mode 1, no node
  // terms), both the gather path of its forward and ddk_conv_forward's explicit-boundary entry (test_confidence_conv_layer_vs_oracle[kernel 0])
  if (L.w2x != nullptr && (a.mode == 0 || (a.mode == 1 && a.pre == nullptr)))
    return L.limbs == 2 ? launch_conv_fused_x2(L, a, n_cu, s) : launch_conv_fused_x(L, a, n_cu, s);
  ConvKArgs k;
  k.w1x = nullptr; k.w2x = nullptr;
  for (int g = 0; g < CONV_MAX_GROUPS; ++g) { k.w1s[g] = 1.0f; k.w1u[g] = 1.0f; k.w2s[g] = 1.0f; k.w2u[g] = 1.0f; }
  k.x = a.x; k.src = a.src; k.dst = a.dst; k.edge_attr = a.edge_attr; k.sh = a.sh; k.sum = a.sum;
  k.counter = a.counter;
  k.w1p = L.w1p[0]; k.b1p = L.b1p[0]; k.w2r = L.w2r[0]; k.n_tiles = L.n_tiles;
  k.n_cols = L.n_cols;
  for (int c = 0; c <= L.n_cols; ++c) k.col_start[c] = L.col_start[c];
  k.sum_g2 = a.sum_g2; k.g2_node_off = a.g2_node_off;
  k.n_groups = a.n_groups; k.n_active = a.n_active; k.n_slots = a.n_slots; k.slots = a.slots; k.wmap = a.wmap;
  if (a.gbeg) { k.gbeg = a.gbeg; k.gend = a.gend; }
  else { k.gbeg = a.tile_info + 5; k.gend = a.tile_info + 6; }   // 4 contiguous groups go[g] .. go[g+1] (explicit-boundary entry point)
  k.pre = a.pre; k.part = a.part; k.det_rng = a.det_rng; k.det_nr = a.det_nr;
  if (a.part != nullptr) {       // deterministic scatter (score model paths only)
    if (a.mode != 0) return hipErrorInvalidValue;
    hipError_t e = (a.gather && a.pre != nullptr) ? launch_conv_t<true, 0, true, true>(k, n_cu, s)
                                                  : (!a.gather ? launch_conv_t<false, 0, false, true>(k, n_cu, s) : hipErrorInvalidValue);
    if (e != hipSuccess) return e;
    conv_det_fix(k, a, L.dout, s);
    return hipGetLastError();
  }
  if (a.mode == 1) return a.gather ? launch_conv_t<true, 1, false>(k, n_cu, s) : launch_conv_t<false, 1, false>(k, n_cu, s);
  if (a.gather && a.pre != nullptr) return launch_conv_t<true, 0, true>(k, n_cu, s);
  return a.gather ? launch_conv_t<true, 0, false>(k, n_cu, s) : launch_conv_t<false, 0, false>(k, n_cu, s);
}

hipError_t launch_conv_setup(int32_t* tile_info, const int64_t* go, hipStream_t s) {
  hipLaunchKernelGGL(conv_setup_kernel, dim3(1), dim3(64), 0, s, tile_info, (int)go[0], (int)go[1], (int)go[2], (int)go[3],
                     (int)go[4]);
  return hipGetLastError();
}

hipError_t launch_conv_one_group(int32_t* gt, int n_groups, int k, int64_t E, hipStream_t s) {
  hipLaunchKernelGGL(conv_one_group_kernel, dim3(1), dim3(64), 0, s, gt, n_groups, k, (int)E);
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

hipError_t launch_node_finalize(float* sum, const int32_t* deg, const float* x_in, const float* bn_mean,
                                const float* bn_scale, const float* bn_bias, int64_t n, int dout, int out_stride,
                                float* out, hipStream_t s, const float* sum_rr0, int64_t n_lig_total, int n_rec, int clear_sum,
                                float* zero_extra, int64_t n_extra, int n_slots, const uint8_t* rr0_mask) {
  int64_t tot = n * out_stride;
  if (zero_extra != nullptr && n_extra > tot) tot = n_extra;
  if (tot == 0) return hipSuccess;
  hipLaunchKernelGGL(node_finalize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, sum, deg, x_in, bn_mean,
                     bn_scale, bn_bias, n, dout, out_stride, out, sum_rr0, n_lig_total, n_rec, clear_sum, zero_extra, n_extra, n_slots, rr0_mask);
  return hipGetLastError();
}

}  // namespace ddk
