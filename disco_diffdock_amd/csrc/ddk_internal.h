// Internal declarations shared by the ddk translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/ddk.h"
#include "../../include/ddk_debug.h"

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
constexpr int F_C = NS;                // c[ns]
constexpr int F_T1O = 2 * NS;          // 12 rows x xyz: [p*s0 (nv) ; (q x v)/sqrt2 (nv)], component-major inside each quad of rows
constexpr int F_T1E = F_T1O + 6 * NV;  // 12 rows x xyz: [(p x v)/sqrt2 (nv) ; q*s0 (nv)]
constexpr int F_PQ = F_T1E + 6 * NV;   // [pv0..3 | qv0..3 | pv4 pv5 qv4 qv5]   pv = (p.v)/sqrt3, qv = (q.v)/sqrt3
constexpr int F_STRIDE = F_PQ + 12;    // 132 floats: 16-B aligned rows, 16-B slot = row mod 16 -> conflict free for ds_read_b128
static_assert(F_STRIDE == 132 && NV == 6, "F row layout");

// F-row extension of the confidence model's l<=2 tensor product (e3nn FullyConnectedTensorProduct with sh 0e+1o+2e on the
// same 0e/1o/1e/0o node irreps): the 1o(x)2e->1o and 1e(x)2e->1e paths contract p / q with the symmetric traceless
// v^ v^T - |v^|^2 I/3 (|v^| = 1, or 0 for a zero-length edge), i.e. 6 more vector rows per block (two quads, component-major like T1O/T1E).
constexpr int F_T2O = F_STRIDE;            // 6 rows x xyz (+2 pad rows): v^ (v^.p) - p/3
constexpr int F_T2E = F_T2O + 24;          // same with q
constexpr int F_STRIDE2 = F_T2E + 24;      // 180 floats: 16-B slot = 13*row mod 16 -> conflict free
static_assert(F_STRIDE2 == 180, "F row layout (l<=2)");

// Workgroup shape of the fused conv kernel: 8 waves (2 per SIMD) x 32 edges, one workgroup per CU (6 waves with the wider
// F rows of the confidence model); the W2 tile records are fetched once per workgroup into a 2-stage LDS ring.
constexpr int CONV_WAVES = 8;
constexpr int CONV_BLOCK_EDGES = 32 * CONV_WAVES;
constexpr int W2_TILE_FLOATS = 9 * 64 * 4 + 32 + 4;   // MFMA fragments [9][64][4] + bias [2][16] + tile descriptor (w0, chan0, 0, 0)
constexpr int CONV_LDS_FLOATS = CONV_WAVES * 32 * F_STRIDE + 2 * W2_TILE_FLOATS + 16;
constexpr size_t CONV_LDS_BYTES = (size_t)CONV_LDS_FLOATS * 4;   // 153,952 B of the 160 KiB
static_assert(CONV_LDS_BYTES <= 160 * 1024, "LDS budget");
template <int MODE> struct ConvTraits;      // MODE 0: FasterTensorProduct (score model), MODE 1: l<=2 FCTP (confidence model)
template <> struct ConvTraits<0> { static constexpr int WAVES = CONV_WAVES, FS = F_STRIDE; };
template <> struct ConvTraits<1> { static constexpr int WAVES = 6, FS = F_STRIDE2; };
template <int MODE> constexpr size_t conv_lds_bytes() { return (size_t)(ConvTraits<MODE>::WAVES * 32 * ConvTraits<MODE>::FS + 2 * W2_TILE_FLOATS + 16) * 4; }
static_assert(conv_lds_bytes<1>() <= 160 * 1024, "LDS budget (l<=2)");
constexpr int CONV_MAX_GROUPS = 9;
// Exact three-limb f16 kernel (k_conv_x.hip, ddk_config.conv_kernel = 0): W2 tile record = three limbs (hi | mid | lo, fp16, each at its own weight) x
// [4 fragments [64 lanes][8] of K steps 0..3 | tail fragment [64][4] of the last 8 K values] | bias [2][16] f32 | pad ; element (s, lane, i) of
// a fragment = weight of tile row lane&31 for the hidden unit held by register 8*s+i of lane half lane>>5 (K = 72 = 4 x 16 + 8)
constexpr int W2X_LIMB_BYTES = 4 * 1024 + 512;                      // 4,608
constexpr int W2X_BIAS_OFF = 3 * W2X_LIMB_BYTES;                    // 13,824
constexpr int W2X_DESC_OFF = W2X_BIAS_OFF + 128;                    // the tile's two descriptor words (x_tile_word(w0), chan0) | 8 B pad
constexpr int W2X_TILE_BYTES = W2X_DESC_OFF + 16;                   // 13,968 = 873 x 16 B
constexpr int W1X_TILE_BYTES = 3 * W2X_LIMB_BYTES;                  // GEMM1: the three limbs of one 32-row tile
constexpr int CONV_TRACE_TILES = 1024;                              // tiles per wave the TRACE instantiation of the kernel records
// F row of the three-limb kernel (100 floats instead of 132): the vector blocks keep the RAW p / q rows once (12 rows x xyz, component-major inside
// each quad of rows like T1O / T1E) instead of the four products p*s0, (q x v)/sqrt2, (p x v)/sqrt2, q*s0 - multiplying by s0 and crossing with v
// are linear, so they are applied ONCE per output channel when a vector column is flushed (two accumulator sets: "times s0" and "cross v").
// 16-B slot of row r = 25 r mod 16 = 9 r mod 16: conflict free for ds_read_b128.  The 36,864 B this frees hold two more ring stages.
constexpr int FX_A = 0, FX_C = NS, FX_R = 2 * NS, FX_PQ = FX_R + 6 * NV, FX_STRIDE = FX_PQ + 12 + 4;     // 0, 24, 48, 84, 100
static_assert(FX_STRIDE == 100 && FX_A == F_A && FX_C == F_C, "F row layout (three-limb f16)");
constexpr int W2X_STAGES = 4;                                       // LDS ring stages of the three-limb kernel
constexpr size_t CONV_X_LDS_BYTES = (size_t)CONV_WAVES * 32 * FX_STRIDE * 4 + W2X_STAGES * W2X_TILE_BYTES + 16;
static_assert(CONV_X_LDS_BYTES <= 160 * 1024 && W2X_TILE_BYTES % 16 == 0, "LDS budget (three-limb f16)");

// One W2 "tile" = 32 weight rows x 72 hidden units = one burst of 36 v_mfma_f32_32x32x2_f32 per 32 edges.
// Tile row rho = 8*rq + 4*hh + j (rq = accumulator quad 0..3, hh = lane half, j = 0..3) holds the weight that multiplies
// input row (row0 + j) of one FasterTensorProduct block for output channel k = 8*col + 2*rq + hh: every lane owns four
// output channels (rq) of its edge and the whole tile consumes the SAME four feature rows -> one F read per tile and a
// kind-specialised epilogue (VALU work steals issue cycles from the fp32 MFMA pipe).
enum TileKind : int {
  T_RA = 0,   // scalar features F[f_off..+4); accumulate into accA (multiplied by s0 or v when the column is flushed)
  T_RT = 1,   // scalar features F[f_off..+4); accumulate into accV[0] (already complete: (p.v)/sqrt3, (q.v)/sqrt3)
  T_TV = 2,   // vector features F[f_off..+12) (4 rows x xyz); accumulate into accV[0..2]
  T_RTS = 3   // the shared tail of the two dot-product parts, F[F_PQ+8..+4) = [pv4 pv5 | qv4 qv5]: rows j = 0,1 finish the 0e column
              // (accumulated, the column is flushed), rows j = 2,3 START the 0o column of the same channel slots behind the flush
};
enum FlushMode : int { FL_NONE = 0, FL_S = 1 /* out = accA*s0 + accV0 */, FL_V = 2 /* out_c = accA*v_c + accV_c */ };

struct TileDesc {
  int32_t w0;     // kind | flush_mode<<2 | nrq<<4 (valid accumulator quads of the column, 1..4) | f_off<<16
  int32_t chan0;  // flush: output column of (rq=0, hh=0); slot (rq,hh) writes chan0 + cstep*(2*rq+hh) (+c), cstep = 1 (FL_S) / 3 (FL_V)
  int32_t pad0, pad1;
};
static inline TileDesc make_tile(int kind, int f_off, int flush, int nrq, int chan0) {
  TileDesc t;
  t.w0 = kind | (flush << 2) | (nrq << 4) | (f_off << 16);
  t.chan0 = chan0; t.pad0 = 0; t.pad1 = 0;
  return t;
}

// Descriptor word of a tile for the three-limb kernel: the tile table is written for the F row of k_conv.hip; this moves the feature offset to
// the kernel's own row (FX_*) and tells a vector tile which of its rows are "times s0" rows and which are crossed with v (bit 14: rows j = 0,1
// are cross rows, bit 15: rows j = 2,3; T1O = [p s0 (nv) ; q x v (nv)], T1E = [p x v (nv) ; q s0 (nv)]).  The confidence model's l = 2 row groups
// (T2O: six p rows, T2E: six q rows, contracted with v^ v^T - |v^|^2 I/3) read the same raw rows and accumulate into a third set (bit 24): T2O quad q
// = raw quad q (rows 2,3 of its second quad are q0, q1: zero weights), T2E's two tiles = raw quads 1 and 2, i.e. [. . q0 q1] and [q2 q3 q4 q5] -
// pack_x3 moves the weight rows accordingly.  -1: not representable.
constexpr int32_t X_TILE_L2 = 1 << 24;
static inline int32_t x_tile_word(int32_t w0) {
  const int kind = w0 & 3, f_off = w0 >> 16;
  int nf = f_off;
  if (kind == T_RT || kind == T_RTS) nf = FX_PQ + (f_off - F_PQ);
  else if (kind == T_TV && f_off >= F_T2O) {
    const bool odd = f_off < F_T2E;
    const int rel = f_off - (odd ? F_T2O : F_T2E), q = rel / 12;
    if (f_off >= F_STRIDE2 || rel % 12 || q > 1 || (w0 & 0xc000)) return -1;
    nf = FX_R + 12 * (odd ? q : q + 1);
    return (w0 & 0xffff) | (nf << 16) | X_TILE_L2;
  } else if (kind == T_TV) {
    const bool odd = f_off < F_T1E;
    const int rel = f_off - (odd ? F_T1O : F_T1E), q = rel / 12;
    if (f_off < F_T1O || f_off >= F_PQ || rel % 12 || (w0 & 0xc000)) return -1;
    nf = FX_R + 12 * q;
    for (int half = 0; half < 2; ++half) {
      const bool first = 4 * q + 2 * half < NV;                 // rows of the block's first part (p): s0 rows in T1O, cross rows in T1E
      if (first != odd) w0 |= 0x4000 << half;
    }
  }
  return (w0 & 0xffff) | (nf << 16);
}

struct ConvLayerDev {          // device copies for one TensorProductConvLayer with FasterTensorProduct
  int n_tiles = 0;             // W2 tiles of 32 rows
  bool has_weights = false;    // radial-MLP / BatchNorm weights were present at finalize time
  int W = 0;                   // weight_numel
  int din = 0, dout = 0;
  float* w1p[4] = {};          // [3][9][64][4]
  float* b1p[4] = {};          // [3][2][16]
  float* w2r[4] = {};          // [n_tiles][W2_TILE_FLOATS]: per tile the fragments [9][64][4], the bias [2][16], the TileDesc words
  uint8_t* w1x = nullptr;      // three-limb f16 kernel: [groups][3][W1X_TILE_BYTES]
  uint8_t* w2x = nullptr;      // three-limb f16 kernel: [groups][n_tiles][W2X_TILE_BYTES]
  float w1s[CONV_MAX_GROUPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1}, w2s[CONV_MAX_GROUPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1};   // three-limb f16 kernel: power-of-two range scale of the packed W1 / W2 of each weight set
  int limbs = 2;               // fp16 limbs per fp32 operand in the f16-limb kernel: 2 (three products, k_conv_x2.hip: the default) or 3 (six products, k_conv_x.hip: conv_kernel = 3); set by pack_x3
  bool epi_ok = false;         // the tile table has the column shapes the generated asm epilogue hard-codes (conv_epilogue_shapes_ok): set by pack_x3
  int n_cols = 0;              // flush columns (8 output channels each); col_start[c] = first tile of column c, col_start[n_cols] = n_tiles
  int col_start[17] = {};
  // GEMM1 split: W1 [edge_emb | x_src[:ns] | x_dst[:ns]] = W1a edge_emb + (W1b x[src][:ns] + b1) + W1c x[dst][:ns]; the
  // two node terms are formed once per NODE and layer (node_pre / node_finalize_pre kernels) instead of once per edge.  A node has four
  // roles: ligand atoms  receive in groups 0,1 (slots 0,1) and send in groups 0,3 (slots 2,3); residues receive in 2,3 and send in 1,2.
  // wn [2 node types][4 slots][72 pos][ns], bn [2][4][72]; pos = position of the hidden unit in GEMM1's accumulator layout
  // (lane half hh, tile T, register r): hh*36 + T*16 + r  (pre_pos below)
  float* wn = nullptr;
  float* bnp = nullptr;
  std::vector<float> h_wn, h_bnp;
  float* bn_mean = nullptr;    // [XW]  running_mean on 0e channels, 0 elsewhere
  float* bn_scale = nullptr;   // [XW]  weight/sqrt(var+eps)   (1 when batch_norm is off)
  float* bn_bias = nullptr;    // [XW]  bias on 0e channels, 0 elsewhere
  // host copies kept for tests (ddk_debug_export)
  std::vector<std::vector<float>> h_w1p, h_b1p, h_w2p, h_b2p;   // [n_groups]
  std::vector<float> h_bn_mean, h_bn_scale, h_bn_bias;          // [n_bn][XW]: one BatchNorm per layer (score) or per conv (confidence)
  int n_groups = 4;
  std::vector<TileDesc> h_tiles;
  std::vector<uint8_t> h_w1x, h_w2x;     // host copies of the three-limb records (ddk_debug_export, CPU tests)
  // block shapes of the FasterTensorProduct (tensor_layers.py:58-63), order 0e,1o,1e,0o
  int n_in[4] = {}, n_out[4] = {}, blk_off[4] = {};
  int in_mul[4] = {}, out_mul[4] = {};   // 0e,1o,1e,0o multiplicities of the layer's in/out irreps
};

constexpr int PRE_W = 4 * NE;   // floats of node pre-activations per node: 4 roles x 72 hidden units
// position of hidden unit o (0..71) inside a role's 72 floats: the order in which lane half hh reads its GEMM1 accumulator init
static inline int pre_pos(int o) {
  const int r32 = o < 64 ? o % 32 : o - 64, hh = (r32 >> 2) & 1;
  return o < 64 ? hh * 36 + (o / 32) * 16 + (r32 & 3) + 4 * (r32 >> 3) : hh * 36 + 32 + (r32 & 3);
}

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

namespace ddk { constexpr int PROF_INTS = 8; }

struct ddk_ctx {
  ddk_config cfg;
  std::string err;
  bool finalized = false;
  bool host_only = false;
  int n_cu = 256;
  int max_rec = 0;                // residues per sample the graph kernels' LDS holds on this device (graph_prepare_device)
  std::map<std::string, ddk::HostTensor> weights;
  std::vector<ddk::ConvLayerDev> conv;
  ddk::ConvLayerDev head[2];      // [0] tor_bond_conv, [1] final_conv as layouts of the fused conv kernel (build_head_layer)
  std::vector<double> so3_table, torus_table;
  ddk::Workspace ws;
  std::vector<void*> dev_allocs;
  // every hipMalloc of the context goes through ctx_malloc / ctx_free: dev_bytes = device memory this context holds right now (weights, workspaces,
  // chunks of live complexes AND chunks parked in the pool); alloc_limit > 0 (ddk_debug_set_alloc_limit) makes a request beyond it fail like a real
  // hipErrorOutOfMemory - the hook behind the retry-with-half-the-batch test (evaluate.py:228-231,394-398 of the reference)
  int64_t dev_bytes = 0, alloc_limit = 0, alloc_refusals = 0;
  std::map<void*, size_t> dev_sizes;
  // packed small weights for the non-conv kernels live in model.hip (opaque here)
  void* model = nullptr;
  void* conf_model = nullptr;   // conf.hip
  // asynchronous complex upload (model.hip): a non-blocking upload stream, pinned staging buffers and a pool of device chunks, so
  // that ddk_complex_create / ddk_complex_destroy never synchronise with a sampling loop in flight on the compute stream
  hipStream_t up_stream = nullptr;
  hipStream_t head_stream = nullptr;   // final_conv runs beside tor_bond_conv (two short launches that fill a fraction of the CUs each)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  struct StageBuf { char* p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool in_flight = false; };
  std::vector<StageBuf> stage_pool;
  struct PoolChunk { void* p = nullptr; size_t cap = 0; hipEvent_t free_after = nullptr; };   // free_after: last use by the previous owner
  std::vector<PoolChunk> chunk_pool;
  size_t chunk_pool_bytes = 0;
  int64_t pool_mallocs = 0, pool_reuses = 0, pool_frees = 0, pool_bytes_out = 0, pool_bytes_out_peak = 0;   // ddk_debug_pool_stats
  // profiling (ddk_profile_enable / ddk_profile_read)
  bool prof = false;
  bool prune = true;                // backward receptive-field pruning of the rec-rec messages (ddk_set_receptive_field_pruning)
  uint32_t* conv_trace = nullptr; int conv_trace_layer = -1; int conv_trace_coarse = 0;   // ddk_debug_conv_trace: the next forward's launch of this layer runs the TRACE kernel
  bool layer0_dedup = true;         // layer-0 rec-rec messages once per batch (+ per-sample patches for the latent-conditioned model); ddk_debug_set_layer0_dedup
  struct ProfRec { hipEvent_t a, b; int layer; int slot; int tab = 0; int64_t r01_skipped = 0; bool lig_only = false; };   // tab: group table of the launch
  std::vector<ProfRec> prof_recs;
  int32_t* prof_edges = nullptr;   // pinned host: PROF_INTS edge counts of forward #slot (InfoSlot I_EXEC block)
  int prof_slots = 0, prof_cap = 0;
};

namespace ddk {
int fail(ddk_ctx* ctx, int code, const std::string& msg);
int hip_fail(ddk_ctx* ctx, hipError_t e, const char* what);
hipError_t ctx_malloc(ddk_ctx* ctx, void** p, size_t bytes);      // hipMalloc under the context's accounting (and debug limit)
void ctx_free(ddk_ctx* ctx, void* p);
size_t pool_evict_idle(ddk_ctx* ctx, bool wait);                  // model.hip: give parked chunks back to the driver (memory pressure)
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
  float* part = nullptr;        // deterministic mode (ddk_config.deterministic): [edge_bound / 32 + 8][2][XW] partial rows; null: fp32 atomics
  int64_t edge_bound = 0;       // host upper bound of the last edge index of the launch (sizes the fix-up grid)
  const int32_t* det_rng = nullptr;   // deterministic mode: sample-aligned ranges of the launch (ConvKArgs::det_rng, det_ranges_kernel) ...
  int det_nr = 0;                     // ... and their number (0: blocks run through the groups)
  const float* pre = nullptr;   // [N, PRE_W] node terms of GEMM1 (gather mode, score model) or null: GEMM1 over all 72 inputs
  int gather;                // 1: edge_attr is edge_emb[E,24] and x[src][:24], x[dst][:24] are gathered
  // layer-0 receptor-receptor de-duplication (all samples of a batch share the receptor and, before the first conv,
  // its node/edge features): group 2 of the launch's group table is the shared copy of the receptor edges (sample-0 numbering,
  // graph_fill_kernel); its messages go to sum_g2[(src - g2_node_off)] and node_finalize adds that row to every sample's copy.
  float* sum_g2 = nullptr;
  int g2_node_off = 0;
  // n_groups edge groups, group g occupies edges [gbeg[g], gend[g]) (device arrays; null: tile_info[5 + g] .. tile_info[6 + g]), uses the
  // g-th radial MLP of the layer and accumulates into sum[(node * n_slots + slot(g)) * XW]; only the first n_active groups run
  // (score model, last layer: 2 = the messages into ligand nodes; its receptor rows are dead)
  int mode = 0;              // ConvTraits MODE
  int n_groups = 4, n_active = 4, n_slots = 1;
  uint32_t slots = 0;        // 2 bits per group
  uint64_t wmap = 0x876543210ull;   // 4 bits per group: weight set of group g (the DisCo patch group 4 uses the rec-rec weights: 0x23210)
  const int32_t* gbeg = nullptr;
  const int32_t* gend = nullptr;
  uint32_t* trace = nullptr;   // != null (three-limb kernel, split gather path): workgroup 0 records its half-phase time stamps here
  int trace_coarse = 0;        // one record per unit instead of per tile (no stamps inside the tile loop)
  bool use_y = false;          // variant builds only (DDK_VARIANT_CONV_Y): tools/variants/k_conv_y.hip for the gather launches with node terms
};
hipError_t launch_conv_fused(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s);
hipError_t launch_conv_fused_x(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s);   // k_conv_x.hip (three f16 limbs per operand, six products)
hipError_t launch_conv_fused_x2(const ConvLayerDev& L, const ConvLaunch& a, int n_cu, hipStream_t s);  // k_conv_x2.hip (two limbs, three products: the default)
hipError_t launch_split3_probe(const float* x, int64_t n, int group, float* hi, float* mid, float* lo, float* scale, hipStream_t s);
int build_head_layer(ddk_ctx* ctx, int mode, ConvLayerDev& L);      // ddk_capi.hip: mode 2 = tor_bond_conv, 3 = final_conv
hipError_t conv_prepare_device();     // per-device kernel attributes (dynamic LDS opt-in), called by ddk_create
hipError_t conv_prepare_device_x();   // k_conv_x.hip
hipError_t conv_prepare_device_x2();  // k_conv_x2.hip
hipError_t graph_prepare_device(int* max_rec);   // k_graph.hip, per device: dynamic-LDS opt-in of the graph kernels, largest receptor their LDS holds
hipError_t launch_conv_setup(int32_t* tile_info, const int64_t* group_offsets_host, hipStream_t s);
hipError_t launch_conv_one_group(int32_t* gt, int n_groups, int k, int64_t E, hipStream_t s);
hipError_t launch_pad_rows(const float* x, int64_t n, int din, float* xpad, hipStream_t s);
hipError_t launch_count_deg(const int32_t* src, int64_t E, int32_t* deg, hipStream_t s);
// node_finalize of layer l fused with the node terms of layer l+1's GEMM1 (x_in == null, sum == null: the node terms of `x_out` alone)
struct NodePreArgs {
  float* sum; const int32_t* deg; const float* x_in; const float* bn_mean; const float* bn_scale; const float* bn_bias;
  int dout; float* x_out; const float* sum_rr0; int n_lig_total, n_rec_total, n_rec; float* zero_extra; int64_t n_extra; int n_slots;
  const float* wn; const float* bnp; float* pre;
  int lig_roles, rec_roles;                  // bit r: role slot r (ConvLayerDev::wn) is needed by the layer the terms are for
  const uint8_t* rr0_mask = nullptr;         // [n_rec_total] 1: this residue row takes no shared layer-0 rec-rec row (its messages came per sample)
  const uint8_t* levels; int max_level;      // [B * n_rec] receptive-field level of the residues (k_graph.hip) and the deepest one this layer still needs; null: all
};
struct NodeEmbedArgs;      // model.h
hipError_t launch_node_finalize_pre(const NodePreArgs& a, bool finalize, hipStream_t s, const NodeEmbedArgs* embed = nullptr);      // embed != null (finalize false): the node embedding first
hipError_t launch_node_finalize(float* sum, const int32_t* deg, const float* x_in /*[N,XW] or null*/,
                                const float* bn_mean, const float* bn_scale, const float* bn_bias, int64_t n, int dout,
                                int out_stride, float* out, hipStream_t s, const float* sum_rr0 = nullptr,
                                int64_t n_lig_total = 0, int n_rec = 1, int clear_sum = 0, float* zero_extra = nullptr,
                                int64_t n_extra = 0, int n_slots = 1, const uint8_t* rr0_mask = nullptr);
// k_tp.hip
hipError_t launch_tp_forward(const ConvLayerDev& L, const float* x_dst, const float* sh, const float* w, int64_t E,
                             float* out, hipStream_t s);
}  // namespace ddk
