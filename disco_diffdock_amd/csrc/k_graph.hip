// Graph construction and edge / node featurisation for one batch of B samples of one complex
// (reference models/score_model.py:310-408 + 218-225): ligand radius graph + bonds, ligand<->receptor cross
// edges with the time-dependent cutoff, the static receptor kNN edges, merged into ONE edge list
//   [ lig-lig | lig->rec | rec-rec | rec->lig ]   (group order of score_model.py:220-224)
// with every group sorted by edge_src so that the fused conv kernel's segmented reduction sees long runs.
// One workgroup per sample; ligand and receptor coordinates live in LDS (n_rec*12 B: 3.6 KB for 300 residues,
// 24 KB for 2000), neighbour tests are brute force over the sample (30 x 300 .. 30 x 2000 pairs).
#include <stdlib.h>
#include "model.h"
#include "k_node.h"

namespace ddk {


// The neighbour tests below are evaluated at SEVERAL places for the same pair - the counting kernel, the fill kernel's own counts, its write loops,
// the pair matrix of cross_mirror - and the edge lists are only consistent if every place gets the same answer for a pair that sits on the cutoff.
// With floating-point contraction on, the compiler is free to fuse the sum of squares differently at every inlined copy (fma chains vs mul + add):
// a pair within an ulp of the cutoff was then counted by one kernel and not written by the other, which left ONE slot of a cross-edge group with
// whatever the reused device chunk held - a garbage node index, and a GPU memory fault once in a few hundred complexes (round 4, the 363-complex
// stream; chunks fresh from hipMalloc are zero, which hid it).  Hence: no contraction in these three functions - plain IEEE mul / add in source order.
// torch_cluster.radius on coordinates rescaled by the per-graph cutoff (score_model.py:379-381): |x/c - y/c|^2 < 1
__device__ __forceinline__ bool cross_within(const float* lp, const float* rp, float c) {
#pragma clang fp contract(off)
  const float dx = rp[0] / c - lp[0] / c, dy = rp[1] / c - lp[1] / c, dz = rp[2] / c - lp[2] / c;
  return dx * dx + dy * dy + dz * dz < 1.0f;
}

// the same test on coordinates that were divided by c ONCE when they were staged in LDS (bit-identical quotients; the six IEEE
// divisions per pair were ~90 % of the instructions of the counting / fill loops): the ligand atom as one 16-B LDS word
// (x, y, z, -), the residue in registers
__device__ __forceinline__ bool cross_within4(const float4 a, float rx, float ry, float rz) {
#pragma clang fp contract(off)
  const float dx = rx - a.x, dy = ry - a.y, dz = rz - a.z;
  return dx * dx + dy * dy + dz * dz < 1.0f;
}

__device__ __forceinline__ float dist2(const float* a, const float* b) {
#pragma clang fp contract(off)
  const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return dx * dx + dy * dy + dz * dz;
}

// The two per-sample graph kernels are chains of short phases whose length is a few LDS / L2 latencies times the trips of their loops:
// 16 waves per workgroup cut the trips by four against the 4 waves they ran with.
constexpr int GT = 1024;

// adjacency of the capped radius graph: bit (i, j) set <=> j is among the first LIG_CAP (incl. self) atoms within
// lig_max_radius of centre i, j != i  (radius_graph -> edge (src=j, dst=i), score_model.py:315)
__device__ void build_lig_adj(const float* lp, int n_lig, float r2, unsigned (*adj)[MAX_LIG / 32]) {
  // one wave per centre, lane = candidate: the cap keeps the first LIG_CAP candidates (ascending index, self included) inside the radius, i.e. a
  // candidate survives when fewer than LIG_CAP in-radius candidates precede it - a ballot and a prefix popcount instead of a serial walk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = wave; i < n_lig; i += nw) {
    int cnt = 0;
    if (lane < MAX_LIG / 32) adj[i][lane] = 0u;          // (LDS writes of one wave land in program order: the chunks below overwrite their words)
    for (int j0 = 0; j0 < n_lig; j0 += 64) {
      const int j = j0 + lane;
      const bool in = j < n_lig && dist2(lp + 3 * i, lp + 3 * j) < r2;
      const unsigned long long m = __ballot(in);
      const bool keep = in && cnt + __popcll(m & ((1ull << lane) - 1ull)) < LIG_CAP && j != i;
      const unsigned long long kept = __ballot(keep);
      if (lane == 0) { adj[i][j0 >> 5] = (unsigned)kept; adj[i][(j0 >> 5) + 1] = (unsigned)(kept >> 32); }
      cnt += __popcll(m);
    }
  }
}

// In-place exclusive prefix sum of a[0..n) in LDS by the whole block (any multiple of 64 threads up to 1024; thread t owns a contiguous chunk; wave shuffles across
// the chunk totals); tmp: 16 ints of LDS.  Ends with a barrier.  (A single thread walking 2000 residues took 85 us.)
__device__ void block_exclusive_scan(int* a, int n, int* tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockDim.x;
  const int per = (n + nt - 1) / nt;
  const int beg = min(tid * per, n), end = min(beg + per, n);
  int s = 0;
  for (int k = beg; k < end; ++k) s += a[k];
  int v = s;
#pragma unroll
  for (int d = 1; d < 64; d *= 2) {
    const int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  if (lane == 63) tmp[wave] = v;
  __syncthreads();
  int base = v - s;                     // exclusive prefix inside the wave
  for (int w = 0; w < wave; ++w) base += tmp[w];
  for (int k = beg; k < end; ++k) { const int t = a[k]; a[k] = base; base += t; }
  __syncthreads();
}

// ---- backward receptive-field levels of the residues of one sample ---------------------------------------------------------
// The heads read ligand rows only, so the LAST conv layer evaluates only the messages into ligand nodes; those read receptor rows
// only at the residues that carry a cross edge in this sample (level A).  Walking back through the stack, conv layer L-2 therefore
// has to produce only the level-A receptor rows: of its rec-rec messages only those RECEIVED by a level-A residue; they read the
// rows of the sending residues, so layer L-3 has to produce level B = A + {senders of rec-rec edges into A}, layer L-4 level
// C = B + {senders into B}.  Exact: a pruned message never reaches anything the heads read.  (Messages are received at edge_src
// and sent from edge_dst, tensor_layers.py:153-159.)  lvl[j]: 0 = A, 1 = B \ A, 2 = C \ B, 3 = the rest.
// On entry lvl[j] is 0 or 3 and a barrier has passed; ends with a barrier.
// (called by the GT threads of graph_count_kernel)
__device__ void propagate_levels(uint8_t* lvl, const GraphArgs& G) {
  constexpr int UC = 8;                             // edges per thread the register path holds (src | dst << 16: n_rec <= MAX_REC < 65536)
  if (G.E_rr <= GT * UC) {
    // both passes walk the same static edge list: fetched ONCE (all loads in flight together), then two LDS-only passes
    unsigned e[UC];
#pragma unroll
    for (int u = 0; u < UC; ++u) {
      const int k = threadIdx.x + GT * u;
      e[u] = k < G.E_rr ? (unsigned)G.rr_src[k] | ((unsigned)G.rr_dst[k] << 16) : 0xffffffffu;
    }
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        const int r = (int)(e[u] & 0xffffu), d = (int)(e[u] >> 16);
        if (e[u] != 0xffffffffu && lvl[r] == pass && lvl[d] == 3) lvl[d] = (uint8_t)(pass + 1);   // (concurrent writers store the same value)
      }
      __syncthreads();
    }
    return;
  }
  constexpr int U = 8;                              // independent index loads in flight per thread (the loop is L2-latency bound)
  for (int pass = 0; pass < 2; ++pass) {
    for (int k0 = threadIdx.x; k0 < G.E_rr; k0 += GT * U) {
      int r[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + GT * u;
        r[u] = k < G.E_rr ? G.rr_src[k] : -1;
        d[u] = k < G.E_rr ? G.rr_dst[k] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r[u] >= 0 && lvl[r[u]] == pass && lvl[d[u]] == 3) lvl[d[u]] = (uint8_t)(pass + 1);   // (concurrent writers store the same value)
    }
    __syncthreads();
  }
}

// Per level: the number of static rec-rec edges received by the residues of that level (tot[0..3]) and, when pre != nullptr, for
// every residue the count over the residues before it of the SAME level (pre[j]) - a 4-component block scan by the whole block
// (thread t owns a contiguous chunk; wave shuffles across the chunk totals).  tmp: [16 waves][4], tot: [4] in LDS.  Ends with a barrier.
__device__ void level_scan(const uint8_t* lvl, const int32_t* outdeg, int n, int* pre, int (*tmp)[4], int* tot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockDim.x, nw = nt >> 6;
  const int per = (n + nt - 1) / nt;
  const int beg = min(tid * per, n), end = min(beg + per, n);
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int k = beg; k < end; ++k) {
    const int l = lvl[k], od = outdeg[k];
    s0 += l == 0 ? od : 0; s1 += l == 1 ? od : 0; s2 += l == 2 ? od : 0; s3 += l == 3 ? od : 0;
  }
  int v0 = s0, v1 = s1, v2 = s2, v3 = s3;
#pragma unroll
  for (int d = 1; d < 64; d *= 2) {
    const int t0 = __shfl_up(v0, d, 64), t1 = __shfl_up(v1, d, 64), t2 = __shfl_up(v2, d, 64), t3 = __shfl_up(v3, d, 64);
    if (lane >= d) { v0 += t0; v1 += t1; v2 += t2; v3 += t3; }
  }
  if (lane == 63) { tmp[wave][0] = v0; tmp[wave][1] = v1; tmp[wave][2] = v2; tmp[wave][3] = v3; }
  __syncthreads();
  if (pre != nullptr) {
    int b0 = v0 - s0, b1 = v1 - s1, b2 = v2 - s2, b3 = v3 - s3;     // exclusive prefix inside the wave
    for (int w = 0; w < wave; ++w) { b0 += tmp[w][0]; b1 += tmp[w][1]; b2 += tmp[w][2]; b3 += tmp[w][3]; }
    for (int k = beg; k < end; ++k) {
      const int l = lvl[k], od = outdeg[k];
      pre[k] = l == 0 ? b0 : (l == 1 ? b1 : (l == 2 ? b2 : b3));
      b0 += l == 0 ? od : 0; b1 += l == 1 ? od : 0; b2 += l == 2 ? od : 0; b3 += l == 3 ? od : 0;
    }
  }
  if (tid < 4) { int t_ = 0; for (int w = 0; w < nw; ++w) t_ += tmp[w][tid]; tot[tid] = t_; }
  __syncthreads();
}

__global__ __launch_bounds__(GT) void graph_count_kernel(GraphArgs G) {
  extern __shared__ float smem[];
  float* lp = smem;                              // [MAX_LIG*3]
  float* rp = lp + MAX_LIG * 3;                  // [n_rec*3]
  uint8_t* lvl = reinterpret_cast<uint8_t*>(rp + 3 * G.n_rec + G.n_rec);   // [n_rec] (behind the int array only the fill kernel uses)
  __shared__ unsigned adj[MAX_LIG][MAX_LIG / 32];
  __shared__ float4 lps[MAX_LIG];                // ligand coordinates / cross cutoff, one 16-B word per atom
  __shared__ int s_cnt[2], lvl_tmp[GT / 64][4], lvl_tot[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < G.n_lig * 3; i += GT) {
    const float v = G.lig_pos[(size_t)b * G.n_lig * 3 + i];
    lp[i] = v;
    reinterpret_cast<float*>(lps)[4 * (i / 3) + i % 3] = v / G.cross_cutoff;
  }
  for (int i = tid; i < G.n_rec * 3; i += GT) rp[i] = G.rec_pos[i] / G.cross_cutoff;      // only the cross test reads the residues here
  if (tid < 2) s_cnt[tid] = 0;
  __syncthreads();
  build_lig_adj(lp, G.n_lig, G.lig_r2, adj);
  __syncthreads();
  int c_ll = 0, c_lr = 0;
  for (int i = tid; i < G.n_lig; i += GT)
    for (int w = 0; w < MAX_LIG / 32; ++w) c_ll += __popc(adj[i][w]);
  for (int j = tid; j < G.n_rec; j += GT) {
    const float rx = rp[3 * j], ry = rp[3 * j + 1], rz = rp[3 * j + 2];
    int cnt = 0;
#pragma unroll 4
    for (int i = 0; i < G.n_lig; ++i) cnt += cross_within4(lps[i], rx, ry, rz) ? 1 : 0;
    c_lr += cnt;
    lvl[j] = (G.prune && cnt == 0) ? 3 : 0;
  }
  // one LDS atomic per wave (every thread adding to the same two words serialised 2 x 1024 updates: the longest phase of the kernel)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { c_ll += __shfl_down(c_ll, d, 64); c_lr += __shfl_down(c_lr, d, 64); }
  if ((tid & 63) == 0) { atomicAdd(&s_cnt[0], c_ll); atomicAdd(&s_cnt[1], c_lr); }
  __syncthreads();
  if (G.prune) propagate_levels(lvl, G);
  level_scan(lvl, G.rr_outdeg, G.n_rec, nullptr, lvl_tmp, lvl_tot);
  if (tid < 2) G.counts[CNT_STRIDE * b + tid] = s_cnt[tid];
  if (tid < 3) G.counts[CNT_STRIDE * b + 2 + tid] = lvl_tot[tid];
  for (int j = tid; j < G.n_rec; j += GT) G.levels[(size_t)b * G.n_rec + j] = lvl[j];      // the fill kernel's four slices read them back
}

// Prefixes over the samples of the five per-sample counts, by ONE WAVE (lane = sample, chunks of 64 for larger batches; every wave of every
// fill workgroup repeats it: 5 x B coalesced ints and a few shuffles - a separate one-wave kernel between count and fill cost a launch and
// its two dependency stalls).  ex[0..4]: the exclusive prefixes at sample b, ex[5]: its offset in the "rest" level segment; tot[0..4]: the sums.
__device__ __forceinline__ void sample_prefix(const GraphArgs& G, int b, int* ex, int* tot) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 5; ++k) { tot[k] = 0; ex[k] = 0; }
  for (int b0 = 0; b0 < G.B; b0 += 64) {
    const int bb = b0 + lane;
    int v[5], inc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = bb < G.B ? G.counts[CNT_STRIDE * bb + k] + (k == 0 ? G.M : 0) : 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int x = v[k];
#pragma unroll
      for (int d = 1; d < 64; d *= 2) { const int t = __shfl_up(x, d, 64); if (lane >= d) x += t; }
      inc[k] = x;
    }
    if (b >= b0 && b < b0 + 64) {
#pragma unroll
      for (int k = 0; k < 5; ++k) ex[k] = tot[k] + __shfl(inc[k] - v[k], b - b0, 64);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) tot[k] += __shfl(inc[k], 63, 64);
  }
  ex[5] = b * G.E_rr - (ex[2] + ex[3] + ex[4]);
}

// one thread: group offsets and the per-layer group tables (InfoSlot / GroupTable in model.h) from the sums over the samples
__device__ void write_group_tables(const GraphArgs& G, const int* tot) {
  const int ll = tot[0], lr = tot[1], ra = tot[2], rb = tot[3], rc = tot[4];
  int go[5];
  go[0] = 0; go[1] = ll; go[2] = ll + lr; go[3] = go[2] + G.B * G.E_rr; go[4] = go[3] + lr;
  const int seg[4] = {go[2], go[2] + ra, go[2] + ra + rb, go[2] + ra + rb + rc};     // first edge of the level segments of group 2
  const int n_shared = G.shared_rr ? G.E_rr : 0;
  int bs = 0;
  G.info[I_FB] = 0;
  for (int g = 0; g < 5; ++g) {
    bs += (g == 3 && G.cross_mirror) ? 0 : ((g < 4 ? go[g + 1] - go[g] : n_shared) + 255) / 256;      // (mirror: the lig->rec blocks write the rec->lig features too)
    G.info[I_FB + 1 + g] = bs;
  }
  for (int g = 0; g < 5; ++g) G.info[I_GO + g] = go[g];
  for (int k = 0; k < 8; ++k) G.info[I_CNT + k] = 0;
  for (int l = 0; l < 4; ++l) G.info[I_SEG + l] = seg[l];
  G.info[I_E] = go[4];
  G.info[I_OVF] = ((int64_t)go[4] + n_shared > G.edge_cap) ? 1 : 0;
  G.info[I_SHARED] = G.shared_rr ? go[4] : -1;
  G.info[I_HEAD] = 0; G.info[I_HEAD + 1] = G.B * G.n_lig; G.info[I_HEAD + 2] = G.B * G.n_lig; G.info[I_HEAD + 3] = G.B * G.n_lig;
  G.info[I_HEAD + 4] = 0; G.info[I_HEAD + 5] = 0;
  G.info[I_EXEC] = go[4];
  G.info[I_EXEC + 1] = go[2];
  G.info[I_EXEC + 7] = lr;        // cross edges of this forward (ddk_profile_read_forwards)
  for (int k = 0; k < N_TAB; ++k) {
    int32_t* tb = G.info + I_TAB + 8 * k;
    int tot = 0;
    for (int g = 0; g < 4; ++g) { tb[g] = go[g]; tb[4 + g] = go[g + 1]; }
    if (k == TAB_A) tb[4 + 2] = seg[1];
    if (k == TAB_B) tb[4 + 2] = seg[2];
    if (k == TAB_C) tb[4 + 2] = seg[3];
    if (k == TAB_SHARED) { tb[2] = go[4]; tb[4 + 2] = go[4] + n_shared; }
    for (int g = 0; g < 4; ++g) tot += tb[4 + g] - tb[g];
    if (k == TAB_SHARED && G.patch_off >= 0 && G.shared_rr) tot += G.info[I_PATCH];
    G.info[I_EXEC + 2 + k] = tot;
  }
  if (G.patch_off >= 0 && G.shared_rr) {      // latent-conditioned model: [ll | lr | shared rr | rl | per-sample patches] for layer 0
    const int np = G.info[I_PATCH];
    const int32_t* ts = G.info + I_TAB + 8 * TAB_SHARED;
    for (int g = 0; g < 4; ++g) { G.info[I_TABX + g] = ts[g]; G.info[I_TABX + 5 + g] = ts[4 + g]; }
    G.info[I_TABX + 4] = (int)G.patch_off; G.info[I_TABX + 9] = (int)G.patch_off + np;
    G.info[I_FBX] = bs + (np + 255) / 256;
  } else {
    G.info[I_FBX] = bs;
  }
}

// ---------------------------------------------------------------------------------------------------
// DisCo layer-0 patches.  The shared rec-rec pass runs on sample 0's receptor rows (its latents included).  For sample s > 0 a
// residue j is "marked" when its latent row differs from zero in sample s OR in sample 0; every receiver that is marked or has a marked
// sender gets ALL its rec-rec messages evaluated per sample (patch group, sorted by (sample, receiver, static order)) and takes no
// shared row (rr_mask).  With one-hot latents at latent_dim nodes per sample that is ~1/3 of the rec-rec edges.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool latent_nonzero(const float* lat, int64_t row, int ld) {
  bool nz = false;
  for (int j = 0; j < ld; ++j) nz |= lat[row * ld + j] != 0.0f;
  return nz;
}

// one workgroup per sample: rr_mask + the sample's patch-edge count
__global__ __launch_bounds__(256) void disco_patch_count_kernel(PatchArgs A) {
  extern __shared__ unsigned char mk[];          // [n_rec] marked
  __shared__ int red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < A.n_rec; j += 256)
    mk[j] = b > 0 && (latent_nonzero(A.rec_latent, (int64_t)b * A.n_rec + j, A.latent_dim) || latent_nonzero(A.rec_latent, j, A.latent_dim));
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < A.n_rec; i += 256) {
    bool a = mk[i];
    const int k0 = A.rr_start[i], k1 = k0 + A.rr_outdeg[i];
    for (int k = k0; k < k1 && !a; ++k) a = mk[A.rr_dst[k]];
    A.rr_mask[(size_t)b * A.n_rec + i] = a;
    if (a) cnt += k1 - k0;
  }
  red[tid] = cnt;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if (tid < d) red[tid] += red[tid + d]; __syncthreads(); }
  if (tid == 0) A.patch_cnt[b] = red[0];
}

// one wave: exclusive prefix of the per-sample counts, total -> info[I_PATCH]
__global__ void disco_patch_scan_kernel(PatchArgs A) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int tot = 0;
  for (int b = 0; b < A.B; ++b) { const int c = A.patch_cnt[b]; A.patch_cnt[b] = tot; tot += c; }
  A.patch_cnt[A.B] = tot;
  A.info[I_PATCH] = tot;
}

// one workgroup per sample: the patch edges of the masked receivers in (receiver, static order)
__global__ __launch_bounds__(256) void disco_patch_fill_kernel(PatchArgs A) {
  extern __shared__ int pre_[];                  // [n_rec] exclusive prefix of the masked receivers' edge counts
  __shared__ int scan_tmp[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < A.n_rec; i += 256) pre_[i] = A.rr_mask[(size_t)b * A.n_rec + i] ? A.rr_outdeg[i] : 0;
  __syncthreads();
  block_exclusive_scan(pre_, A.n_rec, scan_tmp);
  const int rec0 = A.B * A.n_lig + b * A.n_rec;
  const int64_t base = A.patch_off + A.patch_cnt[b];
  for (int i = tid; i < A.n_rec; i += 256) {
    if (!A.rr_mask[(size_t)b * A.n_rec + i]) continue;
    const int k0 = A.rr_start[i], n = A.rr_outdeg[i];
    for (int q = 0; q < n; ++q) {
      const int64_t pos = base + pre_[i] + q;
      A.e_src[pos] = rec0 + i; A.e_dst[pos] = rec0 + A.rr_dst[k0 + q]; A.e_aux[pos] = k0 + q;
    }
  }
}

hipError_t launch_disco_patch(const PatchArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(disco_patch_count_kernel, dim3(a.B), dim3(256), (size_t)a.n_rec, s, a);
  hipLaunchKernelGGL(disco_patch_scan_kernel, dim3(1), dim3(64), 0, s, a);
  hipLaunchKernelGGL(disco_patch_fill_kernel, dim3(a.B), dim3(256), (size_t)a.n_rec * sizeof(int), s, a);
  return hipGetLastError();
}

constexpr int FILL_SLICES = 4;

__global__ __launch_bounds__(GT) void graph_fill_kernel(GraphArgs G) {
  extern __shared__ float smem[];
  float* lp = smem;                                   // [MAX_LIG*3]
  float* rp = lp + MAX_LIG * 3;                       // [n_rec*3]
  int* c_rl = reinterpret_cast<int*>(rp + 3 * G.n_rec);   // [n_rec] -> exclusive prefix; later the per-level prefix of the rec-rec edges
  uint8_t* lvl = reinterpret_cast<uint8_t*>(c_rl + G.n_rec);   // [n_rec] receptive-field level of the residue
  // cross_mirror: bit i of row j = ligand atom i is within the cross cutoff of residue j (the rank of a pair inside the residue's run of the rec->lig group)
  unsigned long long* cmask = reinterpret_cast<unsigned long long*>(lvl + ((G.n_rec + 15) & ~15));   // [n_rec][LW]
  const int LW = (G.n_lig + 63) >> 6;
  __shared__ unsigned adj[MAX_LIG][MAX_LIG / 32];
  __shared__ int bdeg[MAX_LIG], odeg[MAX_LIG], c_lr[MAX_LIG], ll_pre[MAX_LIG], lr_pre[MAX_LIG], scan_tmp[GT / 64], lvl_tmp[GT / 64][4], lvl_tot[4];
  // FILL_SLICES workgroups per sample: each repeats the (cheap) counting / prefix phase and writes one slice of the edge list --
  // one workgroup per sample left 216 CUs idle while 256 threads issued ~300 scattered stores each
  constexpr int BOND_LDS = 1024;                    // directed bonds staged in LDS for slice 0's bond-ordered walk (more: read from global)
  __shared__ short2 bond_sd[BOND_LDS];
  const int b = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_lig = G.n_lig, n_rec = G.n_rec;
  // group offsets of the merged edge list and this sample's place in every group (no separate scan kernel: see sample_prefix)
  int ex[6], tot[5];
  sample_prefix(G, b, ex, tot);
  const int g1 = tot[0], g2 = tot[0] + tot[1], g3 = g2 + G.B * G.E_rr, g4 = g3 + tot[1];
  if (b == 0 && slice == 0 && tid == 0) write_group_tables(G, tot);
  if ((int64_t)g4 + (G.shared_rr ? G.E_rr : 0) > G.edge_cap) return;   // capacity overflow (I_OVF is set): reported by the host wrapper
  const int off0 = ex[0], off1 = ex[1];
  const int segoff[4] = {g2 + ex[2], g2 + tot[2] + ex[3], g2 + tot[2] + tot[3] + ex[4], g2 + tot[2] + tot[3] + tot[4] + ex[5]};   // level segment start + this sample's offset inside it (group 2)
  const bool bonds_in_lds = G.M <= BOND_LDS;
  if (bonds_in_lds)
    for (int m = tid; m < G.M; m += GT) bond_sd[m] = make_short2((short)G.bond_src[m], (short)G.bond_dst[m]);
  __shared__ float4 lps[MAX_LIG];                // ligand coordinates / cross cutoff, one 16-B word per atom
  for (int i = tid; i < n_lig * 3; i += GT) {
    const float v = G.lig_pos[(size_t)b * n_lig * 3 + i];
    lp[i] = v;
    reinterpret_cast<float*>(lps)[4 * (i / 3) + i % 3] = v / G.cross_cutoff;
  }
  for (int i = tid; i < n_rec * 3; i += GT) rp[i] = G.rec_pos[i] / G.cross_cutoff;      // only the cross test reads the residues here
  for (int i = tid; i < MAX_LIG; i += GT) { bdeg[i] = 0; c_lr[i] = 0; }
  for (int j = tid; j < n_rec; j += GT) c_rl[j] = 0;
  __syncthreads();
  build_lig_adj(lp, n_lig, G.lig_r2, adj);
  for (int m = tid; m < G.M; m += GT) atomicAdd(&bdeg[G.bond_src[m]], 1);
  // cross-edge counts without LDS atomics (at t ~ 1 every pair is an edge: 2 x n_lig x n_rec atomics on n_lig + n_rec addresses
  // serialised): per ligand atom one wave + ballots, per residue one thread
  for (int i = wave; i < n_lig; i += GT / 64) {
    const float4 a = lps[i];
    int cnt = 0;
    for (int j0 = 0; j0 < n_rec; j0 += 64) {
      const int j = min(j0 + lane, n_rec - 1);
      cnt += __popcll(__ballot(j0 + lane < n_rec && cross_within4(a, rp[3 * j], rp[3 * j + 1], rp[3 * j + 2])));
    }
    if (lane == 0) c_lr[i] = cnt;
  }
  for (int j = tid; j < n_rec; j += GT) {
    const float rx = rp[3 * j], ry = rp[3 * j + 1], rz = rp[3 * j + 2];
    int cnt = 0;
    if (G.cross_mirror) {
      for (int w = 0; w < LW; ++w) {
        unsigned long long word = 0ull;
        const int i1 = min(n_lig, 64 * (w + 1));
        for (int i = 64 * w; i < i1; ++i) word |= (unsigned long long)(cross_within4(lps[i], rx, ry, rz) ? 1 : 0) << (i & 63);
        cmask[(size_t)j * LW + w] = word;
        cnt += __popcll(word);
      }
    } else {
#pragma unroll 4
      for (int i = 0; i < n_lig; ++i) cnt += cross_within4(lps[i], rx, ry, rz) ? 1 : 0;
    }
    c_rl[j] = cnt;
    lvl[j] = G.levels[(size_t)b * n_rec + j];        // receptive-field level, computed by graph_count_kernel
  }
  __syncthreads();
  for (int j = tid; j < n_lig; j += GT) {
    int od = 0;
    for (int i = 0; i < n_lig; ++i) od += (adj[i][j >> 5] >> (j & 31)) & 1u;
    odeg[j] = od;
  }
  __syncthreads();
  const int rec_base = G.rec_node_base >= 0 ? G.rec_node_base : G.B * n_lig;
  const int lig0 = b * n_lig, rec0 = rec_base + b * n_rec;
  // degrees (scatter 'mean' divisor: all incoming groups together, tensor_layers.py:159)
  if (slice == 0) {
    for (int i = tid; i < n_lig; i += GT) G.deg[lig0 + i] = bdeg[i] + odeg[i] + c_lr[i];
    for (int j = tid; j < n_rec; j += GT) G.deg[rec0 + j] = G.rr_outdeg[j] + c_rl[j];
  }
  __syncthreads();
  if (wave == 0) {                                  // exclusive prefixes over the atoms: one wave, shuffles, chunks of 64
    int a = 0, c = 0;
    for (int i0 = 0; i0 < n_lig; i0 += 64) {
      const int i = i0 + lane;
      const int va = i < n_lig ? bdeg[i] + odeg[i] : 0, vc = i < n_lig ? c_lr[i] : 0;
      int xa = va, xc = vc;
#pragma unroll
      for (int d = 1; d < 64; d *= 2) {
        const int ta = __shfl_up(xa, d, 64), tc = __shfl_up(xc, d, 64);
        if (lane >= d) { xa += ta; xc += tc; }
      }
      if (i < n_lig) { ll_pre[i] = a + xa - va; lr_pre[i] = c + xc - vc; }
      a += __shfl(xa, 63, 64); c += __shfl(xc, 63, 64);
    }
    // guard: the group offsets come from graph_count_kernel's counts, the slots are written from THIS kernel's recount - the two must agree
    // (the neighbour predicates compile without fp contraction for that reason); a disagreement would leave a slot of a reused chunk unwritten
    if (lane == 0 && slice == 0 && (a != G.counts[CNT_STRIDE * b] + G.M || c != G.counts[CNT_STRIDE * b + 1])) atomicOr(&G.info[I_MISMATCH], 1);
  }
  block_exclusive_scan(c_rl, n_rec, scan_tmp);   // exclusive prefix of the per-residue rec->lig counts (in place; ends with a barrier)
  // ---- group 0: lig-lig, sorted by src: bonds of the atom (bond order) then radius edges (ascending dst); one wave per atom, ballot compaction
  // (a thread per atom walking M bonds and n_lig candidates with a conditional store each was the longest phase of the kernel)
  for (int j = wave + (GT / 64) * slice; j < n_lig; j += (GT / 64) * FILL_SLICES) {
    int pos = off0 + ll_pre[j];
    for (int m0 = 0; m0 < G.M; m0 += 64) {
      const int m = m0 + lane;
      int src = -1, dst = 0;
      if (m < G.M) {
        if (bonds_in_lds) { const short2 sd = bond_sd[m]; src = sd.x; dst = sd.y; }
        else { src = G.bond_src[m]; dst = G.bond_dst[m]; }
      }
      const bool hit = src == j;
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int p = pos + __popcll(mask & ((1ull << lane) - 1ull));
        G.e_src[p] = lig0 + j; G.e_dst[p] = lig0 + dst; G.e_aux[p] = m;
      }
      pos += __popcll(mask);
    }
    for (int i0 = 0; i0 < n_lig; i0 += 64) {
      const int i = i0 + lane;
      const bool hit = i < n_lig && ((adj[i][j >> 5] >> (j & 31)) & 1u);
      const unsigned long long mask = __ballot(hit);
      if (hit) {
        const int p = pos + __popcll(mask & ((1ull << lane) - 1ull));
        G.e_src[p] = lig0 + j; G.e_dst[p] = lig0 + i; G.e_aux[p] = -1;
      }
      pos += __popcll(mask);
    }
  }
  // ---- group 1: lig->rec, sorted by ligand atom then residue: one wave per ligand atom, ballot compaction
  for (int i = wave + (GT / 64) * slice; i < n_lig; i += (GT / 64) * FILL_SLICES) {
    int pos = g1 + off1 + lr_pre[i];
    const float4 a = lps[i];
    for (int j0 = 0; j0 < n_rec; j0 += 64) {
      const int j = j0 + lane;
      const int jc = min(j, n_rec - 1);
      const bool in = j < n_rec && cross_within4(a, rp[3 * jc], rp[3 * jc + 1], rp[3 * jc + 2]);
      const unsigned long long mask = __ballot(in);
      if (in) {
        const int p = pos + __popcll(mask & ((1ull << lane) - 1ull));
        int aux = -1;
        if (G.cross_mirror) {      // slot of the flipped copy: start of residue j's run (c_rl: exclusive prefix) + the ligand atoms before i in it
          int rank = __popcll(cmask[(size_t)j * LW + (i >> 6)] & ((1ull << (i & 63)) - 1ull));
          for (int w = 0; w < (i >> 6); ++w) rank += __popcll(cmask[(size_t)j * LW + w]);
          aux = g3 + off1 + c_rl[j] + rank;
        }
        G.e_src[p] = lig0 + i; G.e_dst[p] = rec0 + j; G.e_aux[p] = aux;
      }
      pos += __popcll(mask);
    }
  }
  // ---- group 3: rec->lig (flipped cross edges), sorted by residue then ligand atom
  // (one wave per residue, lane = ligand atom: the residue's edges leave as one contiguous store)
  for (int j = wave + (GT / 64) * slice; j < n_rec; j += (GT / 64) * FILL_SLICES) {
    const float rx = rp[3 * j], ry = rp[3 * j + 1], rz = rp[3 * j + 2];
    int pos = g3 + off1 + c_rl[j];
    for (int i0 = 0; i0 < n_lig; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < n_lig && cross_within4(lps[min(i, n_lig - 1)], rx, ry, rz);
      const unsigned long long mask = __ballot(in);
      if (in) {
        const int p = pos + __popcll(mask & ((1ull << lane) - 1ull));
        G.e_src[p] = rec0 + j; G.e_dst[p] = lig0 + i; G.e_aux[p] = -1;
      }
      pos += __popcll(mask);
    }
  }
  // ---- group 2: the static receptor edges of this sample in four segments [A | B | C | rest] by the level of the receiving
  // residue; inside a segment by (sample, residue, static order), i.e. still sorted by edge_src.  Position of static edge k of
  // residue j: segment start + edges of this level in the samples before + in the residues before + rank among j's edges.
  __syncthreads();                                  // group 3 has consumed c_rl: reuse it for the per-level prefix
  level_scan(lvl, G.rr_outdeg, n_rec, c_rl, lvl_tmp, lvl_tot);
  {
    constexpr int U = 2;                              // the index loads of U edges in flight (two dependent L2 round trips per batch instead of per edge)
    for (int k0 = tid + GT * slice; k0 < G.E_rr; k0 += GT * FILL_SLICES * U) {
      int j[U], d[U], st[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + GT * FILL_SLICES * u;
        j[u] = k < G.E_rr ? G.rr_src[k] : -1;
        d[u] = k < G.E_rr ? G.rr_dst[k] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) st[u] = j[u] >= 0 ? G.rr_start[j[u]] : 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j[u] < 0) continue;
        const int k = k0 + GT * FILL_SLICES * u, l = lvl[j[u]];
        const int pos = (l == 0 ? segoff[0] : l == 1 ? segoff[1] : l == 2 ? segoff[2] : segoff[3]) + c_rl[j[u]] + (k - st[u]);
        G.e_src[pos] = rec0 + j[u]; G.e_dst[pos] = rec0 + d[u]; G.e_aux[pos] = k;
      }
    }
  }
  // ---- the shared copy of the receptor edges (sample-0 numbering) behind the four groups: layer 0 evaluates the rec-rec
  // messages once for the whole batch (see model.hip)
  if (G.shared_rr && b == 0) {
    for (int k = tid + GT * slice; k < G.E_rr; k += GT * FILL_SLICES) {
      G.e_src[g4 + k] = rec_base + G.rr_src[k]; G.e_dst[g4 + k] = rec_base + G.rr_dst[k]; G.e_aux[g4 + k] = k;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Edge features: spherical harmonics (component normalised, lmax=1: [1, sqrt3*v^]) and the edge-embedding MLPs
// lig/rec/cross_edge_embedding (score_model.py:51-56,193,199,207) on [bond one-hot | sigma_emb | gaussians].
// The sigma_emb columns are the same for every edge of a forward -> folded into the first-layer bias on the host.
// ---------------------------------------------------------------------------------------------------

__device__ __forceinline__ void edge_features_body(const EdgeFeatArgs& A, const int blk, const int tid) {
  const int bs1 = A.info[I_FB + 1], bs2 = A.info[I_FB + 2], bs3 = A.info[I_FB + 3], bs4 = A.info[I_FB + 4], bs5 = A.info[I_FB + 5];
  const int bs6 = A.patch_off >= 0 ? A.info[I_FBX] : bs5;
  if (blk >= bs6) return;
  // feature group: the four edge groups + the shared rec-rec copy + the DisCo patch group (rec-rec edges with per-sample latents)
  const int fg = (blk >= bs1) + (blk >= bs2) + (blk >= bs3) + (blk >= bs4) + (blk >= bs5);
  const int bstart = fg == 0 ? 0 : (fg == 1 ? bs1 : (fg == 2 ? bs2 : (fg == 3 ? bs3 : (fg == 4 ? bs4 : bs5))));
  const int first = fg < 4 ? A.info[I_GO + fg] : (fg == 4 ? A.info[I_SHARED] : (int)A.patch_off);
  const int last = fg < 4 ? A.info[I_GO + 1 + fg] : (fg == 4 ? A.info[I_SHARED] + A.n_shared : (int)A.patch_off + A.info[I_PATCH]);
  const int e = first + 256 * (blk - bstart) + tid;
  if (e >= last) return;
  if (fg == 2 && A.g2_live_only && e >= A.info[I_SEG + 3]) return;      // behind the level-C segment: no layer evaluates these messages
  const int g = fg >= 4 ? 2 : fg;
  const int sn = A.e_src[e], dn = A.e_dst[e], aux = A.e_aux[e];
  const EdgeMlpDev& M = g == 0 ? A.lig : (g == 2 ? A.rec : A.cross);
  const float* sigb = g == 0 ? A.sp.lig_edge_sigb : (g == 2 ? A.sp.rec_edge_sigb : A.sp.cross_edge_sigb);
  float h[NS];
  float4 shv;
  if (g == 2) {
    shv = *reinterpret_cast<const float4*>(A.rr_sh + 4 * (size_t)aux);
#pragma unroll
    for (int o = 0; o < NS; ++o) h[o] = A.rr_pre1[(size_t)aux * NS + o] + sigb[o];
  } else {
    float vx, vy, vz;
    if (g == 0) {
      const float *ps = A.lig_pos + 3 * (size_t)sn, *pd = A.lig_pos + 3 * (size_t)dn;
      vx = pd[0] - ps[0]; vy = pd[1] - ps[1]; vz = pd[2] - ps[2];
    } else {   // cross edges: vec = rec - lig for BOTH directions (score_model.py:387, 220-223)
      const int ln = g == 1 ? sn : dn, rn = g == 1 ? dn : sn;
      const float* pl = A.lig_pos + 3 * (size_t)ln;
      const float* pr = A.rec_pos + 3 * (size_t)((rn - (A.rec_node_base >= 0 ? A.rec_node_base : A.n_lig_total)) % A.n_rec);
      vx = pr[0] - pl[0]; vy = pr[1] - pl[1]; vz = pr[2] - pl[2];
    }
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
    shv = make_float4(1.0f, vx * inv, vy * inv, vz * inv);
    float gs[DE];
#pragma unroll
    for (int k = 0; k < DE; ++k) {
      const float t = d - M.offset[k];
      gs[k] = expf(M.coeff * (t * t));
    }
    // input-major: h[o] += w1d_t[k][o] * gs[k], k ascending (the same summation order per output as the row-major form, 24 independent chains)
#pragma unroll
    for (int o = 0; o < NS; ++o) h[o] = sigb[o];
#pragma unroll
    for (int k = 0; k < DE; ++k) {
#pragma unroll
      for (int o = 0; o < NS; ++o) h[o] = fmaf(M.w1d_t[k * NS + o], gs[k], h[o]);
    }
    if (g == 0 && aux >= 0) {
      const float4 ba = *reinterpret_cast<const float4*>(A.bond_attr + 4 * (size_t)aux);
#pragma unroll
      for (int o = 0; o < NS; ++o)
        h[o] += M.w1b[o * 4] * ba.x + M.w1b[o * 4 + 1] * ba.y + M.w1b[o * 4 + 2] * ba.z + M.w1b[o * 4 + 3] * ba.w;
    }
  }
  // DisCo latent conditioning (score_model.py:329-337,358-366): [latent[src] | latent[dst]] columns of the first layer for
  // lig-lig and rec-rec edges; the cross edges' latent columns multiply zeros (score_model.py:401)
  if (A.latent_dim > 0 && (g == 0 || g == 2)) {
    const float* lat = g == 0 ? A.lig_latent : A.rec_latent;
    const int off = g == 0 ? 0 : A.n_lig_total;
    const int ld = A.latent_dim;
    for (int j = 0; j < ld; ++j) {
      const float ls = lat[(size_t)(sn - off) * ld + j], ldv = lat[(size_t)(dn - off) * ld + j];
#pragma unroll
      for (int o = 0; o < NS; ++o) h[o] += M.w1l[o * 2 * ld + j] * ls + M.w1l[o * 2 * ld + ld + j] * ldv;
    }
  }
#pragma unroll
  for (int o = 0; o < NS; ++o) h[o] = fmaxf(h[o], 0.0f);
  float* out = A.e_emb + (size_t)e * NS;
  float* out2 = (g == 1 && A.cross_mirror) ? A.e_emb + (size_t)aux * NS : nullptr;      // the flipped copy carries the SAME embedding and sh (score_model.py:220-223)
  const float uncw = (A.latent_dim > 0 && M.unc != nullptr) ? A.unconditional : 0.0f;
  float y[NS];
#pragma unroll
  for (int o = 0; o < NS; ++o) y[o] = M.b2[o];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
#pragma unroll
    for (int o = 0; o < NS; ++o) y[o] = fmaf(M.w2_t[k * NS + o], h[k], y[o]);
  }
  if (uncw != 0.0f) {      // + unconditional[src] * *_edge_unconditional_embedding (score_model.py:213-215)
#pragma unroll
    for (int o = 0; o < NS; ++o) y[o] += uncw * M.unc[o];
  }
#pragma unroll
  for (int o4 = 0; o4 < NS / 4; ++o4) {
    const float4 rv = make_float4(y[4 * o4], y[4 * o4 + 1], y[4 * o4 + 2], y[4 * o4 + 3]);
    *reinterpret_cast<float4*>(out + 4 * o4) = rv;
    if (out2) *reinterpret_cast<float4*>(out2 + 4 * o4) = rv;
  }
  *reinterpret_cast<float4*>(A.e_sh + 4 * (size_t)e) = shv;
  if (out2) *reinterpret_cast<float4*>(A.e_sh + 4 * (size_t)aux) = shv;
}

__global__ __launch_bounds__(256) void edge_features_kernel(EdgeFeatArgs A) { edge_features_body(A, (int)blockIdx.x, (int)threadIdx.x); }

// The edge features and, beside them, the node embedding + layer 0's node terms (node_finalize_pre_body<2>): the latter depend on the diffusion time and
// the latents only - not on the poses, not on the graph - so their blocks run in the same launch instead of in one of their own behind it (round 6:
// -1 launch and ~10 us per reverse step).  Blocks [0, node_blocks) are node tiles (PRE_W threads each), the rest edge blocks (256 of the PRE_W threads).
__global__ __launch_bounds__(PRE_W) void edge_features_node_kernel(EdgeFeatArgs A, NodePreArgs P, NodeEmbedArgs E, int node_blocks) {
  if ((int)blockIdx.x < node_blocks) { node_finalize_pre_body<2>(P, E, (int)blockIdx.x, node_blocks); return; }
  if (threadIdx.x < 256) edge_features_body(A, (int)blockIdx.x - node_blocks, (int)threadIdx.x);
}

// node embeddings: static part (categorical embeddings, ESM projection, bias) + the per-step sigma part
// (AtomEncoder, models/layers.py:140-149), written zero-padded to XW floats per node
__global__ void node_embed_kernel(NodeEmbedArgs A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_l = (int64_t)A.B * A.n_lig, n = n_l + (int64_t)A.B * A.n_rec;
  if (i >= n * XW) return;
  const int64_t node = i / XW;
  const int c = (int)(i % XW);
  float v = 0.0f;
  if (c < NS) {
    const bool lig = node < n_l;
    if (lig) v = A.lig_static[(node % A.n_lig) * NS + c] + A.sp.lig_node_sig[c];
    else v = A.rec_static[((node - n_l) % A.n_rec) * NS + c] + A.sp.rec_node_sig[c];
    if (A.latent_dim > 0) {   // latent columns of the AtomEncoder Linear + unconditional embedding (score_model.py:211-212)
      const float* lat = lig ? A.lig_latent + node * A.latent_dim : A.rec_latent + (node - n_l) * A.latent_dim;
      const float* w = (lig ? A.lig_w_lat : A.rec_w_lat) + c * A.latent_dim;
      for (int j = 0; j < A.latent_dim; ++j) v += w[j] * lat[j];
      const float* u = lig ? A.lig_unc : A.rec_unc;
      if (u != nullptr) v += A.unconditional * u[c];
    }
  }
  A.x[i] = v;
}

// Static per-complex precompute on the upload stream (ddk_complex_create): the receptor's node embedding without its sigma columns,
//   out[j] = W[:, :ns] . table[residue_j] + W[:, ns:ns+lm] . lm_features_j + b     (AtomEncoder, models/layers.py:140-149; 1336-wide at lm = 1280)
// with fp64 accumulators, rounded to fp32 once.  One workgroup per residue: 10 K slices x 24 outputs (round 3 ran the 1304-long sum as ONE
// serial chain per output on 38 workgroups: 600 us for 9.6 MFLOP); w_lm_t is [lm][ns], so the 24 threads of a slice read contiguous weights
// and the same feature.
constexpr int RNS_SLICES = 10;
__global__ __launch_bounds__(RNS_SLICES * NS) void rec_node_static_kernel(RecStaticArgs A) {
  __shared__ double part[RNS_SLICES][NS];
  const int o = threadIdx.x % NS, sl = threadIdx.x / NS, j = blockIdx.x;
  const float* xr = A.rec_x + (size_t)j * A.feat_dim;
  double a = 0.0;
  if (sl == 0) {
    const float* emb = A.rec_table + (size_t)(int)xr[0] * NS;
    a = A.b[o];
    for (int k = 0; k < NS; ++k) a += (double)A.w_emb[o * NS + k] * emb[k];
  }
  const int per = (A.lm + RNS_SLICES - 1) / RNS_SLICES, k0 = sl * per, k1 = min(A.lm, k0 + per);
  for (int k = k0; k < k1; ++k) a += (double)A.w_lm_t[(size_t)k * NS + o] * (double)xr[1 + k];
  part[sl][o] = a;
  __syncthreads();
  if (sl == 0) {
#pragma unroll
    for (int q = 1; q < RNS_SLICES; ++q) a += part[q][o];
    A.out[(size_t)j * NS + o] = (float)a;
  }
}

// ... and the distance half of rec_edge_embedding.0 on the static receptor edges: pre1[k] = W1[:, dist] . gauss(|pos_b - pos_a|)
// (score_model.py:327-344 + GaussianSmearing, tensor_layers.py:171-181): six threads per edge, four outputs each (a thread per edge left the
// 7 200 edges of a 300-residue receptor on 57 workgroups with an 800-FMA chain each)
__global__ __launch_bounds__(256) void rec_edge_static_kernel(const int32_t* rr_src, const int32_t* rr_dst, const float* rec_pos, int E, EdgeMlpDev m,
                                                              float* pre1) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = gid / (NS / 4), o4 = gid % (NS / 4);
  if (k >= E) return;
  const int a = rr_src[k], b = rr_dst[k];
  const float vx = rec_pos[3 * b] - rec_pos[3 * a], vy = rec_pos[3 * b + 1] - rec_pos[3 * a + 1], vz = rec_pos[3 * b + 2] - rec_pos[3 * a + 2];
  const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
  float gs[DE];
#pragma unroll
  for (int q = 0; q < DE; ++q) { const float t = dist - m.offset[q]; gs[q] = expf(m.coeff * (t * t)); }
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = 4 * o4 + i;
    float a2 = 0.0f;
#pragma unroll
    for (int q = 0; q < DE; ++q) a2 += m.w1d[o * DE + q] * gs[q];
    r[i] = a2;
  }
  *reinterpret_cast<float4*>(pre1 + (size_t)k * NS + 4 * o4) = make_float4(r[0], r[1], r[2], r[3]);
}

// zero fill in 16-B stores (hipMemsetAsync's fill kernel takes 80 us for the 4.4 MB of a complex' accumulators); p 16-B aligned
__global__ void zero_fill_kernel(float4* p, int64_t n4, float* tail, int n_tail) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n_tail) tail[i] = 0.0f;
}
hipError_t launch_zero_fill(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(p) & 15) || (bytes & 3)) return hipMemsetAsync(p, 0, bytes, s);
  const int64_t n4 = (int64_t)(bytes / 16);
  const int n_tail = (int)((bytes % 16) / 4);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n4 + 255) / 256 + 1)), dim3(256), 0, s, reinterpret_cast<float4*>(p), n4,
                     reinterpret_cast<float*>(p) + 4 * n4, n_tail);
  return hipGetLastError();
}

hipError_t launch_complex_static(const RecStaticArgs& R, const int32_t* rr_src, const int32_t* rr_dst, const float* rec_pos, int E, const EdgeMlpDev& m,
                                 float* pre1, hipStream_t s) {
  if (R.n_rec > 0) hipLaunchKernelGGL(rec_node_static_kernel, dim3(R.n_rec), dim3(RNS_SLICES * NS), 0, s, R);
  if (E > 0) hipLaunchKernelGGL(rec_edge_static_kernel, dim3((E * (NS / 4) + 255) / 256), dim3(256), 0, s, rr_src, rr_dst, rec_pos, E, m, pre1);
  return hipGetLastError();
}

// dynamic LDS of the two graph kernels for n_rec residues
static inline size_t graph_lds_bytes(int n_rec) {
  return (size_t)(MAX_LIG * 3 + n_rec * 3) * 4 + (size_t)n_rec * 4 + (((size_t)n_rec + 15) & ~(size_t)15);   // + the level bytes
}

// Per DEVICE (called by ddk_create, like conv_prepare_device): opt both graph kernels in to the largest dynamic LDS their static LDS leaves
// of the 160 KB (the attribute is per device: a process-wide "granted so far" would leave every device but the first without it), and
// report the largest receptor a complex on this device may have - ddk_complex_create refuses more with a message instead of letting every
// forward fail with a raw launch error.
static size_t g_graph_dyn[64] = {};      // dynamic LDS the fill kernel may use, per device (graph_prepare_device)
static inline size_t graph_lds_bytes_mirror(int n_rec, int n_lig) { return graph_lds_bytes(n_rec) + (size_t)n_rec * ((n_lig + 63) / 64) * 8; }
int graph_cross_mirror_fits(int n_lig, int n_rec) {
  static const bool off = getenv("DDK_NO_CROSS_MIRROR") != nullptr;      // debugging aid: both directions evaluate their own features
  if (off) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  return g_graph_dyn[dev] > 0 && graph_lds_bytes_mirror(n_rec, n_lig) <= g_graph_dyn[dev];
}

hipError_t graph_prepare_device(int* max_rec) {
  hipFuncAttributes a1, a2;
  hipError_t e = hipFuncGetAttributes(&a1, reinterpret_cast<const void*>(&graph_count_kernel));
  if (e == hipSuccess) e = hipFuncGetAttributes(&a2, reinterpret_cast<const void*>(&graph_fill_kernel));
  if (e != hipSuccess) return e;
  const size_t st = a1.sharedSizeBytes > a2.sharedSizeBytes ? a1.sharedSizeBytes : a2.sharedSizeBytes;
  const size_t dyn = 160 * 1024 - st;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&graph_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - a1.sharedSizeBytes));
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&graph_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - a2.sharedSizeBytes));
  if (e != hipSuccess) return e;
  int n = MAX_REC;
  while (n > 1 && graph_lds_bytes(n) > dyn) --n;
  *max_rec = n;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) g_graph_dyn[dev] = 160 * 1024 - a2.sharedSizeBytes;
  return hipSuccess;
}

hipError_t launch_graph(const GraphArgs& G, int64_t edge_cap, hipStream_t s) {
  const size_t lds = graph_lds_bytes(G.n_rec);
  GraphArgs Gc = G;
  Gc.edge_cap = edge_cap;
  hipLaunchKernelGGL(graph_count_kernel, dim3(G.B), dim3(GT), lds, s, Gc);
  hipLaunchKernelGGL(graph_fill_kernel, dim3(G.B, FILL_SLICES), dim3(GT), G.cross_mirror ? graph_lds_bytes_mirror(G.n_rec, G.n_lig) : lds, s, Gc);
  return hipGetLastError();
}

// ---- deterministic mode: the sample-aligned ranges of a conv launch (ConvKArgs::det_rng, k_conv_common.h) ---------------------------------------
// One range per (edge group [, level segment of group 2], sample), in the launch's group order; the offsets are graph_fill_kernel's own
// (sample_prefix: the same counts, the same arithmetic).  kind: 0 = groups ll, lr only (the last layer); 1 / 2 / 3 = + the level segments A / A, B / A, B, C
// of group 2 + rl; 4 = all four segments (no pruning) + rl; 5 = [ll | lr | the shared rec-rec copy (ONE range) | rl] (layer 0);
// 6 = n_uniform consecutive ranges of len_uniform edges from edge 0 (the heads' centre edges: one range per sample).
// out = [pb[0 .. nr] | beg[nr] | end[nr] | group[nr]].  One workgroup; wave w takes samples w, w + 16, ...
__global__ __launch_bounds__(GT) void det_ranges_kernel(GraphArgs G, int kind, int len_uniform, int32_t* out, int nr) {
  extern __shared__ int nb_s[];                       // [nr] blocks per range -> exclusive prefix
  __shared__ int scan_tmp[GT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t* beg_o = out + nr + 1;
  int32_t* end_o = beg_o + nr;
  int32_t* grp_o = end_o + nr;
  auto put = [&](int r, int beg, int end, int g) {
    if (lane == 0) { beg_o[r] = beg; end_o[r] = end; grp_o[r] = g; nb_s[r] = (end - beg + 255) / 256; }
  };
  if (kind == 6) {
    for (int r = tid; r < nr; r += GT) { beg_o[r] = r * len_uniform; end_o[r] = (r + 1) * len_uniform; grp_o[r] = 0; nb_s[r] = (len_uniform + 255) / 256; }
  } else {
    const int nseg = kind == 5 ? 0 : kind;            // level segments of group 2 that the launch evaluates
    for (int b = wave; b < G.B; b += GT / 64) {
      int ex[6], tot[5];
      sample_prefix(G, b, ex, tot);
      const int g1 = tot[0], g2 = tot[0] + tot[1], g3 = g2 + G.B * G.E_rr;
      const int cnt0 = G.counts[CNT_STRIDE * b] + G.M, cnt1 = G.counts[CNT_STRIDE * b + 1];
      const int ca = G.counts[CNT_STRIDE * b + 2], cb = G.counts[CNT_STRIDE * b + 3], cc = G.counts[CNT_STRIDE * b + 4];
      put(b, ex[0], ex[0] + cnt0, 0);
      put(G.B + b, g1 + ex[1], g1 + ex[1] + cnt1, 1);
      if (kind == 0) continue;
      int r = 2 * G.B;
      const int segbeg[4] = {g2 + ex[2], g2 + tot[2] + ex[3], g2 + tot[2] + tot[3] + ex[4], g2 + tot[2] + tot[3] + tot[4] + ex[5]};
      const int seglen[4] = {ca, cb, cc, G.E_rr - ca - cb - cc};
      for (int l = 0; l < nseg; ++l) { put(r + b, segbeg[l], segbeg[l] + seglen[l], 2); r += G.B; }
      if (kind == 5) {
        const int g4 = g3 + tot[1];
        if (b == 0) put(r, g4, g4 + G.E_rr, 2);
        r += 1;
      }
      put(r + b, g3 + ex[1], g3 + ex[1] + cnt1, 3);
    }
  }
  __syncthreads();
  block_exclusive_scan(nb_s, nr, scan_tmp);           // (in place; ends with a barrier)
  for (int r = tid; r < nr; r += GT) out[r] = nb_s[r];
  if (tid == 0) out[nr] = nb_s[nr - 1] + (end_o[nr - 1] - beg_o[nr - 1] + 255) / 256;
}

int det_ranges_count(int kind, int B) { return kind == 0 ? 2 * B : (kind == 5 ? 3 * B + 1 : (kind == 6 ? B : (3 + kind) * B)); }

hipError_t launch_det_ranges(const GraphArgs& G, int kind, int len_uniform, int32_t* out, hipStream_t s) {
  const int nr = det_ranges_count(kind, G.B);
  hipLaunchKernelGGL(det_ranges_kernel, dim3(1), dim3(GT), (size_t)(nr + 1) * sizeof(int), s, G, kind, len_uniform, out, nr);
  return hipGetLastError();
}

hipError_t launch_edge_features(const EdgeFeatArgs& A, int64_t edge_cap, hipStream_t s) {
  const unsigned blocks = (unsigned)((edge_cap + 255) / 256 + 5);
  hipLaunchKernelGGL(edge_features_kernel, dim3(blocks), dim3(256), 0, s, A);
  return hipGetLastError();
}

hipError_t launch_edge_features_node(const EdgeFeatArgs& A, int64_t edge_cap, const NodePreArgs& P, const NodeEmbedArgs& E, hipStream_t s) {
  const unsigned blocks = (unsigned)((edge_cap + 255) / 256 + 5);
  const int nb = node_pre_tiles(P);
  hipLaunchKernelGGL(edge_features_node_kernel, dim3(blocks + nb), dim3(PRE_W), 0, s, A, P, E, nb);
  return hipGetLastError();
}

hipError_t launch_node_embed(const NodeEmbedArgs& a, hipStream_t s) {
  const int64_t tot = (int64_t)a.B * (a.n_lig + a.n_rec) * XW;
  hipLaunchKernelGGL(node_embed_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace ddk
