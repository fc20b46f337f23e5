// node_finalize_pre: the body shared by k_conv.hip's kernels (one launch per conv layer) and k_graph.hip's fused edge-feature + node-embedding launch
// (round 6: the node embedding and layer 0's node terms depend on the diffusion time only, so they ride beside the edge features instead of behind them).
#pragma once
#include "k_conv_common.h"
#include "model.h"

namespace ddk {

// node_finalize of one layer (FINALIZE) + the per-node terms of the NEXT layer's GEMM1 (ConvLayerDev::wn): one workgroup = PRE_TILE nodes
// of one node type, thread t = (role slot t / 72, hidden position t % 72) with its 24 weights in registers; the node scalars are read
// from LDS as broadcast 16-B words (one LDS instruction per four FMAs).  Bound by the 1152 B per node it writes.
#ifndef PRE_TILE_N
#define PRE_TILE_N 16      // nodes per workgroup: 16 -> 10.2 us, 32 -> 12.5 us, 64 -> 17.9 us, 8 -> 9.7 us per launch at 13 200 nodes (the phase-2 loop is a serial chain per thread)
#endif
constexpr int PRE_TILE = PRE_TILE_N;
// MODE 0: the node terms of the rows of x_out; 1: node_finalize of a layer first; 2: the node embedding first (node_embed_kernel's arithmetic: layer 0's
// terms without a launch of their own between the embedding and the first conv)
template <int MODE>
__device__ __forceinline__ void node_finalize_pre_body(const NodePreArgs& A, const NodeEmbedArgs& E, const int blk, const int nblk) {
  constexpr bool FINALIZE = MODE == 1;
  __shared__ __attribute__((aligned(16))) float xs[PRE_TILE][NS];
  __shared__ unsigned char dead[PRE_TILE];      // residues outside the heads' backward receptive field at this depth: nothing reads their rows
  const int tid = threadIdx.x;
  if (FINALIZE && A.zero_extra != nullptr)
    for (int64_t i = (int64_t)blk * PRE_W + tid; i < A.n_extra; i += (int64_t)nblk * PRE_W) A.zero_extra[i] = 0.0f;
  const int lig_tiles = (A.n_lig_total + PRE_TILE - 1) / PRE_TILE;
  const bool lig = blk < lig_tiles;
  const int tile = lig ? blk : blk - lig_tiles;
  const int node0 = (lig ? 0 : A.n_lig_total) + tile * PRE_TILE;
  const int cnt = min(PRE_TILE, (lig ? A.n_lig_total : A.n_lig_total + A.n_rec_total) - node0);
  if (tid < PRE_TILE) {
    bool d = false;
    if (!lig && A.levels != nullptr && tid < cnt) {
      const int r = node0 - A.n_lig_total + tid;              // residue row: sample r / n_rec, residue r % n_rec
      d = A.levels[r] > A.max_level;
    }
    dead[tid] = d;
  }
  __syncthreads();
  // this thread's weights: requested first, they arrive while the finalize phase runs
  float w[NS];
  float bias = 0.0f;
  if (A.pre != nullptr) {
    const float* wrow = A.wn + ((size_t)(lig ? 0 : 4 * NE) + tid) * NS;
#pragma unroll
    for (int k4 = 0; k4 < NS / 4; ++k4) {
      const float4 t = ld4(wrow + 4 * k4);
      w[4 * k4] = t.x; w[4 * k4 + 1] = t.y; w[4 * k4 + 2] = t.z; w[4 * k4 + 3] = t.w;
    }
    bias = A.bnp[(lig ? 0 : 4 * NE) + tid];
  }
  if (FINALIZE) {
    // 16-B words: 21 per node row, 2-3 per thread (the scalar version walked 10 dependent load -> store rounds per thread)
    constexpr int XW4 = XW / 4;
    for (int idx = tid; idx < cnt * XW4; idx += PRE_W) {
      const int n = idx / XW4, c = 4 * (idx - n * XW4);
      if (dead[n]) continue;                    // (it received no message in this layer either: its accumulators are still zero)
      const int64_t r = node0 + n;
      const float4 xin = ld4(A.x_in + r * XW + c);
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (c < A.dout) {
        const int d = A.deg[r];
        const float dd = (float)(d > 1 ? d : 1);
        float sv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int sl = 0; sl < A.n_slots; ++sl) {
          float4* ps = reinterpret_cast<float4*>(A.sum + (r * A.n_slots + sl) * XW + c);
          const float4 t = *ps;
          *ps = make_float4(0.0f, 0.0f, 0.0f, 0.0f);       // every accumulator that is read is cleared behind the read (see node_finalize_kernel)
          sv[0] += t.x; sv[1] += t.y; sv[2] += t.z; sv[3] += t.w;
        }
        if (A.sum_rr0 != nullptr && !lig && !(A.rr0_mask != nullptr && A.rr0_mask[r - A.n_lig_total])) {
          const float4 t = ld4(A.sum_rr0 + ((r - A.n_lig_total) % A.n_rec) * XW + c);
          sv[0] += t.x; sv[1] += t.y; sv[2] += t.z; sv[3] += t.w;
        }
        const float4 bm = ld4(A.bn_mean + c), bs = ld4(A.bn_scale + c), bb = ld4(A.bn_bias + c);
        const float m4[4] = {bm.x, bm.y, bm.z, bm.w}, s4[4] = {bs.x, bs.y, bs.z, bs.w}, b4[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k < A.dout) v[k] = (sv[k] / dd - m4[k]) * s4[k] + b4[k];
      }
      const float4 o = make_float4(v[0] + xin.x, v[1] + xin.y, v[2] + xin.z, v[3] + xin.w);
      *reinterpret_cast<float4*>(A.x_out + r * XW + c) = o;
      if (c < NS) *reinterpret_cast<float4*>(&xs[n][c]) = o;
    }
  } else if (MODE == 2) {
    // AtomEncoder's static part + the per-step sigma part (+ latent columns), zero padded to XW (k_graph.hip: node_embed_kernel)
    constexpr int XW4 = XW / 4;
    for (int idx = tid; idx < cnt * XW4; idx += PRE_W) {
      const int n = idx / XW4, c = 4 * (idx - n * XW4);
      const int64_t r = node0 + n;
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (c < NS) {
        const float* st = lig ? E.lig_static + (r % E.n_lig) * NS : E.rec_static + ((r - A.n_lig_total) % E.n_rec) * NS;
        const float* sg = lig ? E.sp.lig_node_sig : E.sp.rec_node_sig;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = st[c + k] + sg[c + k];
        if (E.latent_dim > 0) {
          const float* lat = lig ? E.lig_latent + r * E.latent_dim : E.rec_latent + (r - A.n_lig_total) * E.latent_dim;
          const float* u = lig ? E.lig_unc : E.rec_unc;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float* w = (lig ? E.lig_w_lat : E.rec_w_lat) + (c + k) * E.latent_dim;
            for (int j = 0; j < E.latent_dim; ++j) v[k] += w[j] * lat[j];
            if (u != nullptr) v[k] += E.unconditional * u[c + k];
          }
        }
      }
      const float4 o = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(A.x_out + r * XW + c) = o;
      if (c < NS) *reinterpret_cast<float4*>(&xs[n][c]) = o;
    }
  } else {
    for (int idx = tid; idx < cnt * NS; idx += PRE_W) {
      const int n = idx / NS, c = idx - n * NS;
      xs[n][c] = A.x_out[(size_t)(node0 + n) * XW + c];
    }
  }
  if (A.pre == nullptr) return;
  __syncthreads();
  if (!(((lig ? A.lig_roles : A.rec_roles) >> (tid / NE)) & 1)) return;      // a role the next layer does not evaluate (its last layer: ligand side only)
  float* out = A.pre + (size_t)node0 * PRE_W + tid;
#pragma unroll 4
  for (int n = 0; n < cnt; ++n) {
    if (dead[n]) continue;
    float a0 = bias, a1 = 0.0f;
#pragma unroll
    for (int k4 = 0; k4 < NS / 4; ++k4) {
      const float4 x = *reinterpret_cast<const float4*>(&xs[n][4 * k4]);
      a0 = fmaf(w[4 * k4], x.x, a0); a1 = fmaf(w[4 * k4 + 1], x.y, a1);
      a0 = fmaf(w[4 * k4 + 2], x.z, a0); a1 = fmaf(w[4 * k4 + 3], x.w, a1);
    }
    out[(size_t)n * PRE_W] = a0 + a1;
  }
}


inline int node_pre_tiles(const NodePreArgs& a) { return (a.n_lig_total + PRE_TILE - 1) / PRE_TILE + (a.n_rec_total + PRE_TILE - 1) / PRE_TILE; }

}  // namespace ddk
