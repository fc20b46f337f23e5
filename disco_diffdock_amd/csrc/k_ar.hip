// AR latent model on the device: the two predictor MLPs of PretrainedScoreEncoder
// (reference models/pretrained_score_encoder.py:24-45,76-88: Linear(2*ns_ar, H) - BatchNorm1d - ReLU - Linear(H, H) - BatchNorm1d -
// ReLU - Linear(H, 1) on the scalar channels [x[:, :ns_ar] | x[:, -ns_ar:]] of the conv stack's output, one MLP for ligand atoms and
// one for residues) and the per-graph pick of GenericEncoder.encode_ar (models/model_classes.py:21-47: temperature-scaled logits,
// argmax at temperature >= 100, else a draw with probabilities exp(T logit) / sum; one-hot into the latent arrays).
// The embed() pass that produces the node features is the ordinary score-model forward (ddk_score_forward with
// ddk_set_keep_receptor_features(on)); these two kernels are launch-latency sized (13 k nodes x 41 kFLOP).
#include <float.h>

#include "model.h"

namespace ddk {

constexpr int AR_TILE = 32;     // nodes per workgroup

// one workgroup = AR_TILE nodes of ONE node type; thread j = hidden unit j (weights of its rows live in registers)
__global__ __launch_bounds__(AR_H) void ar_logits_kernel(ArArgs A) {
  __shared__ __attribute__((aligned(16))) float xs[AR_TILE][2 * AR_NS_MAX];
  __shared__ __attribute__((aligned(16))) float h1[AR_TILE][AR_H];
  __shared__ float part[AR_TILE][AR_H / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lig_tiles = (A.n_lig_total + AR_TILE - 1) / AR_TILE;
  const bool lig = (int)blockIdx.x < lig_tiles;
  const int tile = lig ? blockIdx.x : blockIdx.x - lig_tiles;
  const int n_type = lig ? A.n_lig_total : A.n_rec_total;
  const int node0 = tile * AR_TILE, cnt = min(AR_TILE, n_type - node0);
  const ArMlpDev& M = lig ? A.s : A.r;
  const int ns = A.ar_ns, nin = 2 * ns;
  const float* xbase = A.x + (size_t)((lig ? 0 : A.n_lig_total) + node0) * XW;
  for (int i = tid; i < cnt * nin; i += AR_H) {
    const int n = i / nin, k = i - n * nin;
    xs[n][k] = xbase[(size_t)n * XW + (k < ns ? k : XW - 2 * ns + k)];     // [x[:, :ns] | x[:, -ns:]]
  }
  float w0[2 * AR_NS_MAX], w4[AR_H];
  const bool on = tid < A.H;
  for (int k = 0; k < nin; ++k) w0[k] = on ? M.w0[(size_t)tid * nin + k] : 0.0f;
#pragma unroll
  for (int k = 0; k < AR_H; ++k) w4[k] = (on && k < A.H) ? M.w4[(size_t)tid * A.H + k] : 0.0f;
  const float b0 = on ? M.b0[tid] : 0.0f, b4 = on ? M.b4[tid] : 0.0f, w8 = on ? M.w8[tid] : 0.0f;
  __syncthreads();
  for (int n = 0; n < cnt; ++n) {
    float a = b0;
    for (int k = 0; k < nin; ++k) a += w0[k] * xs[n][k];
    h1[n][tid] = fmaxf(a, 0.0f);
  }
  __syncthreads();
  for (int n = 0; n < cnt; ++n) {
    float a = b4, a1 = 0.0f;      // (the hidden row is read as broadcast 16-B words: a quarter of the LDS instructions)
#pragma unroll
    for (int k4 = 0; k4 < AR_H / 4; ++k4) {
      const float4 hv = *reinterpret_cast<const float4*>(&h1[n][4 * k4]);
      a = fmaf(w4[4 * k4], hv.x, a); a1 = fmaf(w4[4 * k4 + 1], hv.y, a1);
      a = fmaf(w4[4 * k4 + 2], hv.z, a); a1 = fmaf(w4[4 * k4 + 3], hv.w, a1);
    }
    a += a1;
    float v = w8 * fmaxf(a, 0.0f);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) part[n][wave] = v;
  }
  __syncthreads();
  if (tid < cnt) {
    float v = M.b8;
    for (int w = 0; w < AR_H / 64; ++w) v += part[tid][w];
    // logits[b, :] = [ligand atoms of graph b | residues of graph b]   (pretrained_score_encoder.py:84-88)
    const int node = node0 + tid, per = lig ? A.n_lig : A.n_rec;
    const int b = node / per, i = node - b * per;
    A.logits[(size_t)b * (A.n_lig + A.n_rec) + (lig ? 0 : A.n_lig) + i] = v;
  }
}

// one workgroup per graph: p_i = exp(T logit_i) (NaN -> 0, inf -> FLT_MAX as torch.nan_to_num), pick = first index whose running
// sum exceeds u * total (u: caller-supplied uniform in [0, 1)), or the first maximum for T >= 100; writes the one-hot
__global__ __launch_bounds__(256) void ar_decode_kernel(ArDecodeArgs A) {
  __shared__ double wsum[4];
  __shared__ float bestv[256];
  __shared__ int besti[256];
  __shared__ int pick;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = A.n_lig + A.n_rec;
  const float* lg = A.logits + (size_t)b * n;
  const int per = (n + 255) / 256, beg = min(tid * per, n), end = min(beg + per, n);
  if (tid == 0) pick = n - 1;
  if (A.temperature >= 100.0f) {
    // torch.argmax order: NaN counts as the largest value, ties go to the lowest index (a graph whose logits are all -inf picks node 0)
    auto gt = [](float a, float b) { return a > b || (isnan(a) && !isnan(b)); };
    auto eq = [](float a, float b) { return a == b || (isnan(a) && isnan(b)); };
    float bv = -INFINITY; int bi = beg < end ? beg : 0x7fffffff;
    for (int i = beg; i < end; ++i) { const float v = lg[i] * A.temperature; if (gt(v, bv)) { bv = v; bi = i; } }
    bestv[tid] = bv; besti[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o && (gt(bestv[tid + o], bestv[tid]) || (eq(bestv[tid + o], bestv[tid]) && besti[tid + o] < besti[tid]))) {
        bestv[tid] = bestv[tid + o]; besti[tid] = besti[tid + o];
      }
      __syncthreads();
    }
    if (tid == 0) pick = besti[0];
    __syncthreads();
  } else {
    auto prob = [&](int i) {
      float p = expf(lg[i] * A.temperature);
      if (isnan(p)) p = 0.0f;
      if (isinf(p)) p = FLT_MAX;
      return (double)p;
    };
    double s = 0.0;
    for (int i = beg; i < end; ++i) s += prob(i);
    double v = s;
#pragma unroll
    for (int d = 1; d < 64; d *= 2) {
      const double t = __shfl_up(v, d, 64);
      if (lane >= d) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    double base = v - s, total = 0.0;
    for (int w = 0; w < 4; ++w) { if (w < wave) base += wsum[w]; total += wsum[w]; }
    const double target = (double)A.uniforms[b] * total;
    // the crossing lies in exactly one thread's chunk (running sums are monotone); ties / u*total == total fall to the last index
    double run = base;
    int found = -1;
    for (int i = beg; i < end; ++i) {
      run += prob(i);
      if (found < 0 && run > target && base <= target) found = i;
    }
    if (found >= 0) atomicMin(&pick, found);
    __syncthreads();
  }
  if (tid == 0) {
    const int c = min(max(pick, 0), n - 1);      // (never outside the graph, whatever the logits held)
    if (A.choices) A.choices[(size_t)b * A.latent_dim + A.idx] = c;
    if (c < A.n_lig) A.lig_latent[((size_t)b * A.n_lig + c) * A.latent_dim + A.idx] = 1.0f;
    else A.rec_latent[((size_t)b * A.n_rec + (c - A.n_lig)) * A.latent_dim + A.idx] = 1.0f;
  }
}

hipError_t launch_ar_logits(const ArArgs& A, hipStream_t s) {
  const int tiles = (A.n_lig_total + AR_TILE - 1) / AR_TILE + (A.n_rec_total + AR_TILE - 1) / AR_TILE;
  hipLaunchKernelGGL(ar_logits_kernel, dim3(tiles), dim3(AR_H), 0, s, A);
  return hipGetLastError();
}

hipError_t launch_ar_decode(const ArDecodeArgs& A, int B, hipStream_t s) {
  hipLaunchKernelGGL(ar_decode_kernel, dim3(B), dim3(256), 0, s, A);
  return hipGetLastError();
}

}  // namespace ddk
