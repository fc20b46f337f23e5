// Internal declarations shared by the ddk translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/ddk.h"

namespace ddk {

// ------------------------------------------------------------------------------------------
// Node-feature layout: every node-feature buffer is [N, XW] fp32, zero padded past the layer's
// irreps ("ns x0e + nv x1o + nv x1e + ns x0o" -> 84 floats for ns=24, nv=6).
// ------------------------------------------------------------------------------------------
constexpr int NS = 24;           // scalar multiplicity (yml: ns)
constexpr int NV = 6;            // vector multiplicity (yml: nv)
constexpr int XW = 2 * NS + 6 * NV;  // 84
constexpr int NE = 3 * NS;       // radial-MLP input / hidden width (72)
constexpr int OFF_P = NS;            // 1o block of a node row
constexpr int OFF_Q = NS + 3 * NV;   // 1e block
constexpr int OFF_C = NS + 6 * NV;   // 0o block

// LDS "F row" of one edge in the fused conv kernel (floats); see k_conv.hip
constexpr int F_A = 0;                 // a[ns]
constexpr int F_PV = NS;               // (p.v)/sqrt3 [nv] (+pad to 8)
constexpr int F_T1O = NS + 8;          // [p*s0 (nv x3) ; (q x v)/sqrt2 (nv x3)]
constexpr int F_T1E = F_T1O + 6 * NV;  // [(p x v)/sqrt2 (nv x3) ; q*s0 (nv x3)]
constexpr int F_QV = F_T1E + 6 * NV;   // (q.v)/sqrt3 [nv] (+pad to 8)
constexpr int F_C = F_QV + 8;          // c[ns]
constexpr int F_SH = F_C + NS;         // s0, vx, vy, vz
constexpr int F_STRIDE = F_SH + 4;     // 140 floats: 16-B aligned rows, bank-conflict free for ds_read_b128
static_assert(F_STRIDE == 140, "F row layout");

enum UnitKind : int { U_R1_S0 = 0, U_R1_V = 1, U_T_S = 2, U_T_V = 3, U_PAD = 4 };

// One "unit" = 4 consecutive rows (i) of one weight block for the output-channel pair (2k, 2k+1):
// lanes 0-31 of the wave hold channel 2k, lanes 32-63 channel 2k+1 (MFMA 32x32 D layout).
struct Unit {
  int32_t w0;         // kind | flags<<4 | ncomp<<8 | f_off<<16
                      //   kind: UnitKind; flags bit0: first unit of its (block, kpair), bit1: last;
                      //   ncomp: 1 (scalar output) or 3 (vector output); f_off: float offset into the F row
  int32_t w1;         // chan0 | chan_step<<16  (output channel of lane-half 0 / delta for lane-half 1)
  float scale;        // 1/sqrt(n_in of the block)
  int32_t pad;
};
static inline Unit make_unit(int kind, int f_off, int ncomp, int chan0, int chan_step, float scale) {
  Unit u;
  u.w0 = kind | (ncomp << 8) | (f_off << 16);
  u.w1 = chan0 | (chan_step << 16);
  u.scale = scale;
  u.pad = 0;
  return u;
}

struct ConvLayerDev {          // device copies for one TensorProductConvLayer with FasterTensorProduct
  int n_tiles = 0;             // W2 tiles of 32 rows
  bool has_weights = false;    // radial-MLP / BatchNorm weights were present at finalize time
  int W = 0;                   // weight_numel
  int din = 0, dout = 0;
  float* w1p[4] = {};          // [3][9][64][4]
  float* b1p[4] = {};          // [3][2][16]
  float* w2p[4] = {};          // [n_tiles][9][64][4]
  float* b2p[4] = {};          // [n_tiles][2][16]
  Unit* units = nullptr;       // [n_tiles*4]
  float* bn_mean = nullptr;    // [XW]  running_mean on 0e channels, 0 elsewhere
  float* bn_scale = nullptr;   // [XW]  weight/sqrt(var+eps)   (1 when batch_norm is off)
  float* bn_bias = nullptr;    // [XW]  bias on 0e channels, 0 elsewhere
  // host copies kept for tests (ddk_debug_export)
  std::vector<float> h_w1p[4], h_b1p[4], h_w2p[4], h_b2p[4], h_bn_mean, h_bn_scale, h_bn_bias;
  std::vector<Unit> h_units;
  // block shapes of the FasterTensorProduct (tensor_layers.py:58-63), order 0e,1o,1e,0o
  int n_in[4] = {}, n_out[4] = {}, blk_off[4] = {};
  int in_mul[4] = {}, out_mul[4] = {};   // 0e,1o,1e,0o multiplicities of the layer's in/out irreps
};

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

struct Workspace {     // per-launch scratch of the fused conv entry point
  int32_t* tile_info = nullptr;   // [16]: tile_start[5], group_off[5], counter
  float* xpad = nullptr;          // padded input copy for the explicit-boundary API
  float* sum = nullptr;
  int32_t* deg = nullptr;
  size_t xpad_cap = 0, sum_cap = 0, deg_cap = 0;
};

}  // namespace ddk

struct ddk_ctx {
  ddk_config cfg;
  std::string err;
  bool finalized = false;
  bool host_only = false;
  int n_cu = 256;
  std::map<std::string, ddk::HostTensor> weights;
  std::vector<ddk::ConvLayerDev> conv;
  std::vector<double> so3_table, torus_table;
  ddk::Workspace ws;
  std::vector<void*> dev_allocs;
  // packed small weights for the non-conv kernels live in model.hip (opaque here)
  void* model = nullptr;
  // profiling (ddk_profile_enable / ddk_profile_read)
  bool prof = false;
  struct ProfRec { hipEvent_t a, b; int layer; int slot; int64_t skipped = 0; bool lig_only = false; };   // skipped: edges not evaluated (layer-0 rec-rec dedup)
  std::vector<ProfRec> prof_recs;
  int32_t* prof_edges = nullptr;   // pinned host: total edges of forward #slot
  int prof_slots = 0, prof_cap = 0;
};

namespace ddk {
int fail(ddk_ctx* ctx, int code, const std::string& msg);
int hip_fail(ddk_ctx* ctx, hipError_t e, const char* what);
void* dev_alloc(ddk_ctx* ctx, size_t bytes);                      // tracked, freed in ddk_destroy
float* dev_upload(ddk_ctx* ctx, const std::vector<float>& v);
int ensure(ddk_ctx* ctx, void** p, size_t* cap, size_t bytes);    // grow-only workspace

// k_conv.hip
struct ConvLaunch {
  const float* x;            // [N, XW] padded node features
  const int32_t* src;        // [E]
  const int32_t* dst;        // [E]
  const float* edge_attr;    // explicit mode: [E, 72]; gather mode: edge_emb [E, 24]
  const float* sh;           // [E, 4]
  float* sum;                // [N_out, XW] fp32 accumulators (zeroed by the caller)
  const int32_t* tile_info;  // device: tile_start[5], group_off[5]
  int32_t* counter;          // device tile counter (zeroed by the caller)
  int gather;                // 1: edge_attr is edge_emb[E,24] and x[src][:24], x[dst][:24] are gathered
  // layer-0 receptor-receptor de-duplication (all samples of a batch share the receptor and, before the first conv,
  // its node/edge features): only the first g2_limit edges of group 2 (sample 0) are evaluated, their messages go to
  // sum_g2[(src - g2_node_off)] and node_finalize adds that row to every sample's copy.  g2_limit < 0: off.
  int g2_limit = -1;
  int lig_side_only = 0;     // 1: evaluate only groups 0 and 1 (messages into ligand nodes); the last layer's receptor rows are dead
  float* sum_g2 = nullptr;
  int g2_node_off = 0;
};
hipError_t launch_conv_fused(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s);
hipError_t launch_conv_setup(int32_t* tile_info, const int64_t* group_offsets_host, hipStream_t s);
hipError_t launch_pad_rows(const float* x, int64_t n, int din, float* xpad, hipStream_t s);
hipError_t launch_count_deg(const int32_t* src, int64_t E, int32_t* deg, hipStream_t s);
hipError_t launch_node_finalize(const float* sum, const int32_t* deg, const float* x_in /*[N,XW] or null*/,
                                const float* bn_mean, const float* bn_scale, const float* bn_bias, int64_t n, int dout,
                                int out_stride, float* out, hipStream_t s, const float* sum_rr0 = nullptr,
                                int64_t n_lig_total = 0, int n_rec = 1);
// k_tp.hip
hipError_t launch_tp_forward(const ConvLayerDev& L, const float* x_dst, const float* sh, const float* w, int64_t E,
                             float* out, hipStream_t s);
}  // namespace ddk
