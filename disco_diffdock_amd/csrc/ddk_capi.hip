// C-ABI entry points (include/ddk.h): context, checkpoint loading, weight packing into MFMA fragment
// order, and the operator-level entry points (ddk_tp_forward, ddk_conv_forward).
#include <math.h>
#include <string.h>

#include "ddk_internal.h"

using namespace ddk;

namespace ddk {

int fail(ddk_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int hip_fail(ddk_ctx* ctx, hipError_t e, const char* what) {
  return fail(ctx, DDK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

void* dev_alloc(ddk_ctx* ctx, size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  ctx->dev_allocs.push_back(p);
  return p;
}

float* dev_upload(ddk_ctx* ctx, const std::vector<float>& v) {
  float* p = (float*)dev_alloc(ctx, v.size() * sizeof(float));
  if (p && !v.empty()) {
    if (hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  }
  return p;
}

int ensure(ddk_ctx* ctx, void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes && *p) return DDK_OK;
  if (*p) {
    hipFree(*p);
    *p = nullptr;
  }
  size_t want = bytes + bytes / 4 + 256;
  if (hipMalloc(p, want) != hipSuccess) return fail(ctx, DDK_ERR_NOMEM, "hipMalloc failed for workspace");
  *cap = want;
  return DDK_OK;
}

static const HostTensor* find_w(ddk_ctx* ctx, const std::string& name, std::initializer_list<int64_t> shape) {
  auto it = ctx->weights.find(name);
  if (it == ctx->weights.end()) {
    ctx->err = "missing state_dict key: " + name;
    return nullptr;
  }
  std::vector<int64_t> want(shape);
  if (it->second.shape != want) {
    std::string s = "shape mismatch for " + name + ": got [";
    for (auto d : it->second.shape) s += std::to_string(d) + ",";
    s += "] want [";
    for (auto d : want) s += std::to_string(d) + ",";
    ctx->err = s + "]";
    return nullptr;
  }
  return &it->second;
}

// row of the 32x32 MFMA D tile held by accumulator register r of lane-half hh
static inline int d_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
// hidden-unit index consumed by GEMM2 step s in lane-half hh (== D layout of GEMM1, see k_conv.hip)
static inline int hid_of(int s, int hh) { return s < 32 ? 32 * (s / 16) + d_row(s % 16, hh) : 64 + (s - 32) + 4 * hh; }
// input index consumed by GEMM1 step s in lane-half hh
static inline int kin_of(int s, int hh) { return 24 * (s / 12) + 12 * hh + (s % 12); }

struct Quad { int kind, f_off, row0, jlo, n; };   // tile rows j in [jlo, jlo+n) <-> block rows row0 .. row0+n-1

static int build_conv_layer(ddk_ctx* ctx, int l, ConvLayerDev& L) {
  const ddk_config& c = ctx->cfg;
  const int ns = c.ns, nv = c.nv;
  const int seq[4][4] = {{ns, 0, 0, 0}, {ns, nv, 0, 0}, {ns, nv, nv, 0}, {ns, nv, nv, ns}};
  const int* in = seq[l < 3 ? l : 3];
  const int* out = seq[l + 1 < 3 ? l + 1 : 3];
  for (int b = 0; b < 4; ++b) { L.in_mul[b] = in[b]; L.out_mul[b] = out[b]; }
  L.n_in[0] = in[0] + in[1];          L.n_out[0] = out[0];
  L.n_in[1] = in[0] + in[1] + in[2];  L.n_out[1] = out[1];
  L.n_in[2] = in[1] + in[2] + in[3];  L.n_out[2] = out[2];
  L.n_in[3] = in[2] + in[3];          L.n_out[3] = out[3];
  int off = 0;
  for (int b = 0; b < 4; ++b) { L.blk_off[b] = off; off += L.n_in[b] * L.n_out[b]; }
  L.W = off;
  L.din = in[0] + 3 * in[1] + 3 * in[2] + in[3];
  L.dout = out[0] + 3 * out[1] + 3 * out[2] + out[3];
  if (in[2] > 0 && in[1] == 0) return fail(ctx, DDK_ERR_INVALID, "unsupported irreps sequence");

  // row quads of each block in the reference's row order (tensor_layers.py:72-83)
  std::vector<Quad> quads[4];
  auto add_rows = [&](int b, int kind, int f_off, int row_start, int rows) {
    for (int q = 0; 4 * q < rows; ++q)
      quads[b].push_back({kind, f_off + (kind == T_TV ? 12 * q : 4 * q), row_start + 4 * q, 0, rows - 4 * q < 4 ? rows - 4 * q : 4});
  };
  auto add_dot_rows = [&](int b, int which /*0: p.v, 1: q.v*/, int row_start, int rows) {   // F_PQ = [pv0..3 | qv0..3 | pv4 pv5 qv4 qv5]
    quads[b].push_back({T_RT, F_PQ + 4 * which, row_start, 0, rows < 4 ? rows : 4});
    if (rows > 4) quads[b].push_back({T_RT, F_PQ + 8, row_start + 4, 2 * which, rows - 4});
  };
  add_rows(0, T_RA, F_A, 0, in[0]);                                        // a * s0
  if (in[1]) add_dot_rows(0, 0, in[0], in[1]);                             // (p.v)/sqrt3
  add_rows(1, T_RA, F_A, 0, in[0]);                                        // a (x) v
  if (in[1] + in[2]) add_rows(1, T_TV, F_T1O, in[0], in[1] + in[2]);       // p*s0 ; (q x v)/sqrt2
  if (in[1] + in[2]) add_rows(2, T_TV, F_T1E, 0, in[1] + in[2]);           // (p x v)/sqrt2 ; q*s0
  if (in[3]) add_rows(2, T_RA, F_C, in[1] + in[2], in[3]);                 // c (x) v
  if (in[2]) add_dot_rows(3, 1, 0, in[2]);                                 // (q.v)/sqrt3
  if (in[3]) add_rows(3, T_RA, F_C, in[2], in[3]);                         // c * s0

  const int oc[4] = {0, out[0], out[0] + 3 * out[1], out[0] + 3 * out[1] + 3 * out[2]};
  struct TRow { int blk, col, row0, jlo, n; };
  std::vector<TileDesc> tiles;
  std::vector<TRow> trows;
  for (int b = 0; b < 4; ++b) {
    if (L.n_in[b] == 0 || L.n_out[b] == 0) continue;
    if (L.n_out[b] % 2) return fail(ctx, DDK_ERR_INVALID, "odd output multiplicity unsupported");
    const bool vec = (b == 1 || b == 2);
    for (int col = 0; 8 * col < L.n_out[b]; ++col) {
      if (L.n_cols >= 16) return fail(ctx, DDK_ERR_INVALID, "too many output columns");
      L.col_start[L.n_cols++] = (int)tiles.size();
      const int nch = L.n_out[b] - 8 * col < 8 ? L.n_out[b] - 8 * col : 8;
      for (const Quad& q : quads[b]) {
        tiles.push_back(make_tile(q.kind, q.f_off, FL_NONE, nch / 2, oc[b] + (vec ? 3 : 1) * 8 * col));
        trows.push_back({b, col, q.row0, q.jlo, q.n});
      }
      tiles.back().w0 |= (vec ? FL_V : FL_S) << 2;
    }
  }
  L.n_tiles = (int)tiles.size();
  L.col_start[L.n_cols] = L.n_tiles;
  L.h_tiles = tiles;

  // row map: tile row rho = 8*rq + 4*hh + j  ->  row of the reference weight vector (or -1 = zero row)
  std::vector<int> rowmap((size_t)L.n_tiles * 32, -1);
  std::vector<float> rowscale((size_t)L.n_tiles * 32, 0.f);
  for (int t = 0; t < L.n_tiles; ++t) {
    const TRow& tr = trows[t];
    for (int rq = 0; rq < 4; ++rq)
      for (int hh = 0; hh < 2; ++hh)
        for (int j = tr.jlo; j < tr.jlo + tr.n; ++j) {
          const int i = tr.row0 + j - tr.jlo, k = 8 * tr.col + 2 * rq + hh;
          if (k >= L.n_out[tr.blk]) continue;
          rowmap[(size_t)t * 32 + 8 * rq + 4 * hh + j] = L.blk_off[tr.blk] + i * L.n_out[tr.blk] + k;
          rowscale[(size_t)t * 32 + 8 * rq + 4 * hh + j] = 1.0f / sqrtf((float)L.n_in[tr.blk]);   // tensor_layers.py:89-92, folded
        }
  }
  // every weight row must be used exactly once
  {
    std::vector<int> cnt(L.W, 0);
    for (int r : rowmap) if (r >= 0) cnt[r]++;
    for (int r = 0; r < L.W; ++r) if (cnt[r] != 1) return fail(ctx, DDK_ERR_INVALID, "internal: weight row map is not a bijection");
  }

  const int ne = 3 * ns;
  const std::string pre = "conv_layers." + std::to_string(l);
  if (ctx->weights.find(pre + ".fc.0.0.weight") == ctx->weights.end()) {
    L.has_weights = false;     // shape-only layer: ddk_tp_forward works, ddk_conv_forward refuses
    return DDK_OK;
  }
  L.has_weights = true;
  const size_t w1sz = 3 * 9 * 64 * 4, b1sz = 3 * 2 * 16, w2sz = (size_t)L.n_tiles * 9 * 64 * 4, b2sz = (size_t)L.n_tiles * 32;
  std::vector<float> w1all(4 * w1sz, 0.f), b1all(4 * b1sz, 0.f), w2all(4 * w2sz, 0.f), b2all(4 * b2sz, 0.f);
  for (int g = 0; g < 4; ++g) {
    const std::string f = pre + ".fc." + std::to_string(g);
    const HostTensor* W1 = find_w(ctx, f + ".0.weight", {ne, ne});
    const HostTensor* B1 = find_w(ctx, f + ".0.bias", {ne});
    const HostTensor* W2 = find_w(ctx, f + ".4.weight", {L.W, ne});
    const HostTensor* B2 = find_w(ctx, f + ".4.bias", {L.W});
    if (!W1 || !B1 || !W2 || !B2) return DDK_ERR_INVALID;
    float* w1 = w1all.data() + g * w1sz;
    float* b1 = b1all.data() + g * b1sz;
    float* w2 = w2all.data() + g * w2sz;
    float* b2 = b2all.data() + g * b2sz;
    for (int T = 0; T < 3; ++T) {
      for (int s = 0; s < 36; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int hidden = 32 * T + (lane & 31), hh = lane >> 5;
          w1[(((size_t)T * 9 + s / 4) * 64 + lane) * 4 + (s & 3)] = hidden < ne ? W1->data[(size_t)hidden * ne + kin_of(s, hh)] : 0.f;
        }
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) {
          const int hidden = 32 * T + d_row(r, hh);
          b1[(T * 2 + hh) * 16 + r] = hidden < ne ? B1->data[hidden] : 0.f;
        }
    }
    for (int t = 0; t < L.n_tiles; ++t) {
      for (int s = 0; s < 36; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = rowmap[(size_t)t * 32 + (lane & 31)], hh = lane >> 5;
          w2[(((size_t)t * 9 + s / 4) * 64 + lane) * 4 + (s & 3)] =
              row >= 0 ? W2->data[(size_t)row * ne + hid_of(s, hh)] * rowscale[(size_t)t * 32 + (lane & 31)] : 0.f;
        }
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) {
          const int row = rowmap[(size_t)t * 32 + d_row(r, hh)];
          b2[((size_t)t * 2 + hh) * 16 + r] = row >= 0 ? B2->data[row] * rowscale[(size_t)t * 32 + d_row(r, hh)] : 0.f;
        }
    }
  }
  // BatchNorm (e3nn, eval): per multiplicity channel
  L.h_bn_mean.assign(XW, 0.f);
  L.h_bn_scale.assign(XW, 1.f);
  L.h_bn_bias.assign(XW, 0.f);
  if (c.batch_norm) {
    const int nf = out[0] + out[1] + out[2] + out[3];
    const HostTensor* bw = find_w(ctx, pre + ".batch_norm.weight", {nf});
    const HostTensor* bb = find_w(ctx, pre + ".batch_norm.bias", {out[0]});
    const HostTensor* bm = find_w(ctx, pre + ".batch_norm.running_mean", {out[0]});
    const HostTensor* bv = find_w(ctx, pre + ".batch_norm.running_var", {nf});
    if (!bw || !bb || !bm || !bv) return DDK_ERR_INVALID;
    int ch = 0, f = 0;
    const int dims[4] = {1, 3, 3, 1};
    for (int b = 0; b < 4; ++b)
      for (int m = 0; m < out[b]; ++m, ++f) {
        const float sc = powf(bv->data[f] + 1e-5f, -0.5f) * bw->data[f];
        for (int d = 0; d < dims[b]; ++d, ++ch) {
          L.h_bn_scale[ch] = sc;
          if (b == 0) { L.h_bn_mean[ch] = bm->data[m]; L.h_bn_bias[ch] = bb->data[m]; }
        }
      }
  }
  for (int g = 0; g < 4; ++g) {
    L.h_w1p[g].assign(w1all.begin() + g * w1sz, w1all.begin() + (g + 1) * w1sz);
    L.h_b1p[g].assign(b1all.begin() + g * b1sz, b1all.begin() + (g + 1) * b1sz);
    L.h_w2p[g].assign(w2all.begin() + g * w2sz, w2all.begin() + (g + 1) * w2sz);
    L.h_b2p[g].assign(b2all.begin() + g * b2sz, b2all.begin() + (g + 1) * b2sz);
  }
  if (!ctx->host_only) {
    float* d1 = dev_upload(ctx, w1all);
    float* db1 = dev_upload(ctx, b1all);
    std::vector<float> w2rec((size_t)4 * L.n_tiles * W2_TILE_FLOATS);
    for (int g = 0; g < 4; ++g)
      for (int t = 0; t < L.n_tiles; ++t) {
        float* rec = w2rec.data() + ((size_t)g * L.n_tiles + t) * W2_TILE_FLOATS;
        memcpy(rec, w2all.data() + g * w2sz + (size_t)t * 2304, 2304 * sizeof(float));
        memcpy(rec + 2304, b2all.data() + g * b2sz + (size_t)t * 32, 32 * sizeof(float));
        memcpy(rec + 2336, &tiles[t], 2 * sizeof(int32_t));   // descriptor rides with the record (bit pattern)
      }
    float* d2 = dev_upload(ctx, w2rec);
    L.tiles = (TileDesc*)dev_alloc(ctx, tiles.size() * sizeof(TileDesc));
    L.bn_mean = dev_upload(ctx, L.h_bn_mean);
    L.bn_scale = dev_upload(ctx, L.h_bn_scale);
    L.bn_bias = dev_upload(ctx, L.h_bn_bias);
    if (!d1 || !db1 || !d2 || !L.tiles || !L.bn_mean || !L.bn_scale || !L.bn_bias)
      return fail(ctx, DDK_ERR_NOMEM, "device allocation failed while packing conv weights");
    if (hipMemcpy(L.tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice) != hipSuccess)
      return fail(ctx, DDK_ERR_HIP, "tile table upload failed");
    for (int g = 0; g < 4; ++g) {
      L.w1p[g] = d1 + g * w1sz;
      L.b1p[g] = db1 + g * b1sz;
      L.w2r[g] = d2 + (size_t)g * L.n_tiles * W2_TILE_FLOATS;
    }
  }
  return DDK_OK;
}

int model_finalize(ddk_ctx* ctx);   // model.hip
void model_destroy(ddk_ctx* ctx);   // model.hip

}  // namespace ddk

extern "C" {

const char* ddk_version(void) { return "ddk 0.1 (gfx950)"; }

int ddk_create(const ddk_config* cfg, ddk_ctx** out) {
  if (!cfg || !out) return DDK_ERR_INVALID;
  ddk_ctx* ctx = new ddk_ctx();
  ctx->cfg = *cfg;
  *out = ctx;
  if (cfg->ns != NS || cfg->nv != NV)
    return fail(ctx, DDK_ERR_INVALID, "only ns=24, nv=6 (DiffDock-S / DisCo-DiffDock-S) is compiled in");
  if (cfg->num_conv_layers < 4 || cfg->num_conv_layers > 16)
    return fail(ctx, DDK_ERR_INVALID, "num_conv_layers must be in [4,16] (heads assume the full 0e+1o+1e+0o irreps)");
  if (cfg->sigma_embed_dim != 32 || cfg->distance_embed_dim != 32 || cfg->cross_distance_embed_dim != 32)
    return fail(ctx, DDK_ERR_INVALID, "only 32-wide sigma / distance embeddings are compiled in");
  if (cfg->device < 0) {
    ctx->host_only = true;   // packing-only context (CPU tests); every launch entry point refuses to run
    return DDK_OK;
  }
  hipError_t e = hipSetDevice(cfg->device);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, cfg->device);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipGetDeviceProperties");
  ctx->n_cu = prop.multiProcessorCount;
  ctx->ws.tile_info = (int32_t*)dev_alloc(ctx, 64 * sizeof(int32_t));
  if (!ctx->ws.tile_info) return fail(ctx, DDK_ERR_NOMEM, "hipMalloc failed");
  return DDK_OK;
}

void ddk_destroy(ddk_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->host_only) {
    hipSetDevice(ctx->cfg.device);
    model_destroy(ctx);
    for (auto& r : ctx->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    if (ctx->prof_edges) hipHostFree(ctx->prof_edges);
    for (void* p : ctx->dev_allocs) hipFree(p);
    if (ctx->ws.xpad) hipFree(ctx->ws.xpad);
    if (ctx->ws.sum) hipFree(ctx->ws.sum);
    if (ctx->ws.deg) hipFree(ctx->ws.deg);
  }
  delete ctx;
}

const char* ddk_last_error(ddk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ddk_load_weights(ddk_ctx* ctx, const char* name, const float* host_ptr, const int64_t* shape, int32_t ndim) {
  if (!ctx || !name || (!host_ptr && ndim > 0 && shape[0] != 0)) return fail(ctx, DDK_ERR_INVALID, "ddk_load_weights: null argument");
  const std::string n(name);
  if (n.find(".tp.") != std::string::npos) return DDK_OK;   // e3nn-internal buffers of real checkpoints
  HostTensor t;
  int64_t numel = 1;
  for (int i = 0; i < ndim; ++i) {
    t.shape.push_back(shape[i]);
    numel *= shape[i];
  }
  t.data.assign(host_ptr, host_ptr + numel);
  ctx->weights[n] = std::move(t);
  ctx->finalized = false;
  return DDK_OK;
}

int ddk_finalize_weights(ddk_ctx* ctx) {
  if (!ctx) return DDK_ERR_INVALID;
  if (!ctx->host_only) hipSetDevice(ctx->cfg.device);
  ctx->conv.clear();
  ctx->conv.resize(ctx->cfg.num_conv_layers);
  for (int l = 0; l < ctx->cfg.num_conv_layers; ++l) {
    int rc = build_conv_layer(ctx, l, ctx->conv[l]);
    if (rc != DDK_OK) return rc;
  }
  int rc = model_finalize(ctx);
  if (rc != DDK_OK) return rc;
  ctx->finalized = true;
  return DDK_OK;
}

int ddk_set_score_norm_tables(ddk_ctx* ctx, const double* so3, int32_t n_so3, const double* torus, int32_t n_torus) {
  if (!ctx || !so3 || !torus) return fail(ctx, DDK_ERR_INVALID, "null table");
  if (n_so3 != 1000 || n_torus != 5001) return fail(ctx, DDK_ERR_INVALID, "expected 1000 so3 and 5001 torus entries");
  ctx->so3_table.assign(so3, so3 + n_so3);
  ctx->torus_table.assign(torus, torus + n_torus);
  return DDK_OK;
}

static int check_launchable(ddk_ctx* ctx, int32_t layer) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  if (!ctx->finalized) return fail(ctx, DDK_ERR_STATE, "weights not finalised");
  if (layer < 0 || layer >= (int)ctx->conv.size()) return fail(ctx, DDK_ERR_INVALID, "layer out of range");
  return DDK_OK;
}

int ddk_tp_forward(ddk_ctx* ctx, int32_t layer, const float* x_dst, const float* sh, const float* w, int64_t E,
                   float* out, void* stream) {
  int rc = check_launchable(ctx, layer);
  if (rc) return rc;
  hipError_t e = launch_tp_forward(ctx->conv[layer], x_dst, sh, w, E, out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "tp_forward launch");
  return DDK_OK;
}

int ddk_conv_forward(ddk_ctx* ctx, int32_t layer, const float* x, int64_t N, const int32_t* edge_src,
                     const int32_t* edge_dst, const int64_t* go, const float* edge_attr, const float* sh, float* out,
                     void* stream) {
  int rc = check_launchable(ctx, layer);
  if (rc) return rc;
  if (!go || go[0] != 0 || go[1] < go[0] || go[2] < go[1] || go[3] < go[2] || go[4] < go[3])
    return fail(ctx, DDK_ERR_INVALID, "group_offsets must be a non-decreasing prefix starting at 0");
  if (go[4] >= (int64_t)1 << 31 || N * XW >= (int64_t)1 << 40) return fail(ctx, DDK_ERR_INVALID, "graph too large");
  hipStream_t s = (hipStream_t)stream;
  const ConvLayerDev& L = ctx->conv[layer];
  if (!L.has_weights) return fail(ctx, DDK_ERR_STATE, "conv layer has no weights loaded");
  Workspace& ws = ctx->ws;
  if ((rc = ensure(ctx, (void**)&ws.xpad, &ws.xpad_cap, (size_t)N * XW * 4))) return rc;
  if ((rc = ensure(ctx, (void**)&ws.sum, &ws.sum_cap, (size_t)N * XW * 4))) return rc;
  if ((rc = ensure(ctx, (void**)&ws.deg, &ws.deg_cap, (size_t)N * 4))) return rc;
  hipError_t e;
  const int64_t E = go[4];
#define CK(x, what) do { e = (x); if (e != hipSuccess) return hip_fail(ctx, e, what); } while (0)
  CK(launch_pad_rows(x, N, L.din, ws.xpad, s), "pad_rows");
  CK(hipMemsetAsync(ws.sum, 0, (size_t)N * XW * 4, s), "memset sum");
  CK(hipMemsetAsync(ws.deg, 0, (size_t)N * 4, s), "memset deg");
  if (E > 0) {
    CK(launch_conv_setup(ws.tile_info, go, s), "conv_setup");
    CK(launch_count_deg(edge_src, E, ws.deg, s), "count_deg");
    ConvLaunch a;
    a.x = ws.xpad; a.src = edge_src; a.dst = edge_dst; a.edge_attr = edge_attr; a.sh = sh; a.sum = ws.sum;
    a.tile_info = ws.tile_info; a.counter = ws.tile_info + 10; a.gather = 0;
    CK(launch_conv_fused(L, a, ctx->n_cu, s), "conv_fused");
  }
  // with no edges the reference returns zeros + residual (tensor_layers.py:149-151): BatchNorm is skipped
  if (E > 0)
    CK(launch_node_finalize(ws.sum, ws.deg, ws.xpad, L.bn_mean, L.bn_scale, L.bn_bias, N, L.dout, L.dout, out, s), "node_finalize");
  else
    CK(launch_node_finalize(ws.sum, ws.deg, ws.xpad, nullptr, nullptr, nullptr, N, 0, L.dout, out, s), "node_finalize");
#undef CK
  return DDK_OK;
}

// Test hook: copy a packed host-side array out of the context ("conv.<l>.w2p.<g>", "conv.<l>.units", ...).
// Returns the number of floats (or int32 words) of the item, or a negative status.  buf may be NULL to query.
int64_t ddk_debug_export(ddk_ctx* ctx, const char* what, void* buf, int64_t cap_words) {
  if (!ctx || !what) return DDK_ERR_INVALID;
  int l = 0, g = 0;
  char item[32] = {0};
  const void* src = nullptr;
  int64_t n = 0;
  if (sscanf(what, "conv.%d.%31[a-z0-9_].%d", &l, item, &g) >= 2) {
    if (l < 0 || l >= (int)ctx->conv.size() || g < 0 || g > 3) return fail(ctx, DDK_ERR_INVALID, "bad export index");
    ConvLayerDev& L = ctx->conv[l];
    const std::string it(item);
    if (it == "w1p") { src = L.h_w1p[g].data(); n = L.h_w1p[g].size(); }
    else if (it == "b1p") { src = L.h_b1p[g].data(); n = L.h_b1p[g].size(); }
    else if (it == "w2p") { src = L.h_w2p[g].data(); n = L.h_w2p[g].size(); }
    else if (it == "b2p") { src = L.h_b2p[g].data(); n = L.h_b2p[g].size(); }
    else if (it == "tiles") { src = L.h_tiles.data(); n = L.h_tiles.size() * (sizeof(TileDesc) / 4); }
    else if (it == "bn_mean") { src = L.h_bn_mean.data(); n = XW; }
    else if (it == "bn_scale") { src = L.h_bn_scale.data(); n = XW; }
    else if (it == "bn_bias") { src = L.h_bn_bias.data(); n = XW; }
    else return fail(ctx, DDK_ERR_INVALID, "unknown export item");
  } else {
    return fail(ctx, DDK_ERR_INVALID, "unknown export name");
  }
  if (buf) {
    if (cap_words < n) return fail(ctx, DDK_ERR_INVALID, "export buffer too small");
    memcpy(buf, src, (size_t)n * 4);
  }
  return n;
}

}  // extern "C"
