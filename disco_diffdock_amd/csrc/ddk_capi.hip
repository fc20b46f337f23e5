// C-ABI entry points (include/ddk.h): context, checkpoint loading, weight packing into MFMA fragment
// order, and the operator-level entry points (ddk_tp_forward, ddk_conv_forward).
#include <math.h>
#include <algorithm>
#include <cmath>
#include <string.h>

#include "ddk_internal.h"
#include "k_conv_common.h"

using namespace ddk;

namespace ddk {


int fail(ddk_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

int hip_fail(ddk_ctx* ctx, hipError_t e, const char* what) {
  return fail(ctx, DDK_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

hipError_t ctx_malloc(ddk_ctx* ctx, void** p, size_t bytes) {
  *p = nullptr;
  if (ctx->alloc_limit > 0 && ctx->dev_bytes + (int64_t)bytes > ctx->alloc_limit) { ctx->alloc_refusals++; return hipErrorOutOfMemory; }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) { *p = nullptr; (void)hipGetLastError(); return e; }      // (the sticky error is consumed: the context stays usable)
  ctx->dev_bytes += (int64_t)bytes;
  ctx->dev_sizes[*p] = bytes;
  return hipSuccess;
}

void ctx_free(ddk_ctx* ctx, void* p) {
  if (!p) return;
  auto it = ctx->dev_sizes.find(p);
  if (it != ctx->dev_sizes.end()) { ctx->dev_bytes -= (int64_t)it->second; ctx->dev_sizes.erase(it); }
  hipFree(p);
}

void* dev_alloc(ddk_ctx* ctx, size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  if (ctx_malloc(ctx, &p, bytes) != hipSuccess) {
    // memory pressure: the chunks parked in the pool are the context's own slack - hand them back and try once more
    if (pool_evict_idle(ctx, true) == 0 || ctx_malloc(ctx, &p, bytes) != hipSuccess) return nullptr;
  }
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
    ctx_free(ctx, *p);
    *p = nullptr;
    *cap = 0;
  }
  size_t want = bytes + bytes / 4 + 256;
  if (ctx_malloc(ctx, p, want) != hipSuccess) {
    if (pool_evict_idle(ctx, true) == 0 || ctx_malloc(ctx, p, want) != hipSuccess)
      return fail(ctx, DDK_ERR_NOMEM, "out of device memory: workspace of " + std::to_string(want) + " B (context holds " + std::to_string(ctx->dev_bytes) + " B" +
                  (ctx->alloc_limit > 0 ? ", debug limit " + std::to_string(ctx->alloc_limit) + " B" : std::string()) + ")");
  }
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

struct RowSrc { int wbase; float scale; };           // weight of (row, channel k) = W2[wbase + k] * scale
struct Part { int kind, f_off, dot_which; std::vector<RowSrc> rows; };   // dot_which >= 0: rows live in the F_PQ layout

// Tile table + weight-row map of one conv layer.  mode 0: FasterTensorProduct (tensor_layers.py:58-63,72-92: blocks 0e,1o,1e,0o,
// weights [in_,out] row-major, 1/sqrt(in_)); mode 1: e3nn FullyConnectedTensorProduct(in, 0e+1o+2e, out, shared_weights=False)
// as all_atom_score_model.py:25 builds it: instructions in (in1, sh, out) loop order with 'uvw' weights [mul_in, 1, mul_out],
// path coefficient sqrt(dim_out / sum of mul_in over the paths into that output irrep), real wigner-3j with Frobenius norm 1
// (w3j(0,1,1)=w3j(1,0,1)=w3j(1,1,0)=delta/sqrt3, w3j(1,1,1)=eps/sqrt6, w3j(1,2,1).Y2 = sqrt(3/2) (v^ v^T - I/3)).
static int build_layout(ddk_ctx* ctx, int mode, int l, ConvLayerDev& L, std::vector<int>& rowmap, std::vector<float>& rowscale,
                        std::vector<TileDesc>& tiles) {
  const ddk_config& c = ctx->cfg;
  const int ns = c.ns, nv = c.nv;
  const int seq[4][4] = {{ns, 0, 0, 0}, {ns, nv, 0, 0}, {ns, nv, nv, 0}, {ns, nv, nv, ns}};
  // modes 2 / 3: the output heads as layouts of the same kernel (score_model.py:132-161: tor_bond_conv / final_conv read the full irreps);
  // their "out" multiplicities are per FasterTensorProduct-style block (0e, 1o, 1e, 0o), the output columns are set below
  const int head_out[2][4] = {{ns, 0, 0, ns}, {0, 2, 2, 0}};
  const int* in = seq[(mode >= 2 || l >= 3) ? 3 : l];
  const int* out = mode >= 2 ? head_out[mode - 2] : seq[l + 1 < 3 ? l + 1 : 3];
  for (int b = 0; b < 4; ++b) { L.in_mul[b] = in[b]; L.out_mul[b] = out[b]; }
  L.n_out[0] = out[0]; L.n_out[1] = out[1]; L.n_out[2] = out[2]; L.n_out[3] = out[3];
  L.din = in[0] + 3 * in[1] + 3 * in[2] + in[3];
  L.dout = out[0] + 3 * out[1] + 3 * out[2] + out[3];
  if (in[2] > 0 && in[1] == 0) return fail(ctx, DDK_ERR_INVALID, "unsupported irreps sequence");

  std::vector<Part> parts[4];
  auto rows_of = [](int wbase0, int stride, int n, float scale) {
    std::vector<RowSrc> r;
    for (int i = 0; i < n; ++i) r.push_back({wbase0 + i * stride, scale});
    return r;
  };
  auto cat = [](std::vector<RowSrc> a, const std::vector<RowSrc>& b) { a.insert(a.end(), b.begin(), b.end()); return a; };
  if (mode == 0) {
    L.n_in[0] = in[0] + in[1];
    L.n_in[1] = in[0] + in[1] + in[2];
    L.n_in[2] = in[1] + in[2] + in[3];
    L.n_in[3] = in[2] + in[3];
    int off = 0;
    for (int b = 0; b < 4; ++b) { L.blk_off[b] = off; off += L.n_in[b] * L.n_out[b]; }
    L.W = off;
    auto blk = [&](int b, int row0, int n) { return rows_of(L.blk_off[b] + row0 * L.n_out[b], L.n_out[b], n, 1.0f / sqrtf((float)L.n_in[b])); };
    parts[0].push_back({T_RA, F_A, -1, blk(0, 0, in[0])});                                        // a * s0
    if (in[1]) parts[0].push_back({T_RT, 0, 0, blk(0, in[0], in[1])});                            // (p.v)/sqrt3
    parts[1].push_back({T_RA, F_A, -1, blk(1, 0, in[0])});                                        // a (x) v
    if (in[1] + in[2]) parts[1].push_back({T_TV, F_T1O, -1, blk(1, in[0], in[1] + in[2])});       // p*s0 ; (q x v)/sqrt2
    if (in[1] + in[2]) parts[2].push_back({T_TV, F_T1E, -1, blk(2, 0, in[1] + in[2])});           // (p x v)/sqrt2 ; q*s0
    if (in[3]) parts[2].push_back({T_RA, F_C, -1, blk(2, in[1] + in[2], in[3])});                 // c (x) v
    if (in[2]) parts[3].push_back({T_RT, 0, 1, blk(3, 0, in[2])});                                // (q.v)/sqrt3
    if (in[3]) parts[3].push_back({T_RA, F_C, -1, blk(3, in[2], in[3])});                         // c * s0
  } else if (mode == 2) {
    // tor_bond_conv: e3nn FullyConnectedTensorProduct(84, sh (x) sh_2e, '24x0o + 24x0e') keeps two paths (score_model.py:152-156): 1o (x) T -> 0e
    // (weights [nv][ns] at 0) and 1e (x) T -> 0o (at nv*ns), T = the 1o block of the full tensor product, path coefficient sqrt(1/nv) and
    // w3j(1,1,0) = delta/sqrt3.  The kernel forms (p.v)/sqrt3 and (q.v)/sqrt3 with v := T (heads_pre_kernel puts T into the edge's sh), which
    // leaves 1/sqrt(nv) for the packed weights.  Output columns: [0o | 0e].
    for (int b = 0; b < 4; ++b) { L.n_in[b] = (b == 0 || b == 3) ? nv : 0; L.blk_off[b] = 0; }
    L.W = 2 * nv * ns;
    const float sc = 1.0f / sqrtf((float)nv);
    parts[0].push_back({T_RT, 0, 0, rows_of(0, ns, nv, sc)});
    parts[3].push_back({T_RT, 0, 1, rows_of(nv * ns, ns, nv, sc)});
  } else if (mode == 3) {
    // final_conv: FullyConnectedTensorProduct(84, 0e+1o, '2x1o + 2x1e'), weight blocks in instruction order (score_model.py:132-139):
    //   A 0e(x)1o->1o [ns][2] | B 1o(x)0e->1o [nv][2] | C 1o(x)1o->1e [nv][2] | D 1e(x)0e->1e [nv][2] | E 1e(x)1o->1o [nv][2] | F 0o(x)1o->1e [ns][2]
    // path coefficient sqrt(3 / (ns + 2 nv)) = 1/sqrt12 for both outputs; w3j(0,1,1) = w3j(1,0,1) = delta/sqrt3, w3j(1,1,1) = eps/sqrt6;
    // the kernel's rows are a (x) v, p*s0, (q x v)/sqrt2 | (p x v)/sqrt2, q*s0, c (x) v  with s0 = 1, v = sh[1:4]
    for (int b = 0; b < 4; ++b) { L.n_in[b] = (b == 1 || b == 2) ? ns + 2 * nv : 0; L.blk_off[b] = 0; }
    L.W = 2 * 2 * (ns + 2 * nv);
    const float pc = sqrtf(3.0f / (float)(ns + 2 * nv));
    const float cS = pc * 0.57735026918962576451f, cX = pc * 0.40824829046386301637f * 1.41421356237309504880f;
    const int oA = 0, oB = 2 * ns, oC = oB + 2 * nv, oD = oC + 2 * nv, oE = oD + 2 * nv, oF = oE + 2 * nv;
    parts[1].push_back({T_RA, F_A, -1, rows_of(oA, 2, ns, cS)});
    parts[1].push_back({T_TV, F_T1O, -1, cat(rows_of(oB, 2, nv, cS), rows_of(oE, 2, nv, cX))});
    parts[2].push_back({T_TV, F_T1E, -1, cat(rows_of(oC, 2, nv, cX), rows_of(oD, 2, nv, cS))});
    parts[2].push_back({T_RA, F_C, -1, rows_of(oF, 2, ns, cS)});
  } else {
    // irreps as (l, parity): node 0e,1o,1e,0o ; sh 0e,1o,2e
    const int nl[4] = {0, 1, 1, 0}, np_[4] = {+1, -1, +1, -1}, sl[3] = {0, 1, 2}, sp[3] = {+1, -1, +1};
    int inst_off[4][3][4];
    int fan[4] = {0, 0, 0, 0}, off = 0;
    for (int i1 = 0; i1 < 4; ++i1)
      for (int i2 = 0; i2 < 3; ++i2)
        for (int io = 0; io < 4; ++io) {
          inst_off[i1][i2][io] = -1;
          if (!in[i1] || !out[io]) continue;
          const bool tri = nl[io] >= abs(nl[i1] - sl[i2]) && nl[io] <= nl[i1] + sl[i2];
          if (!tri || np_[i1] * sp[i2] != np_[io]) continue;
          inst_off[i1][i2][io] = off;
          off += in[i1] * out[io];
          fan[io] += in[i1];
        }
    L.W = off;
    for (int b = 0; b < 4; ++b) { L.n_in[b] = fan[b]; L.blk_off[b] = 0; }
    const float is3 = 0.57735026918962576451f, kappa = 1.22474487139158904910f;
    auto coeff = [&](int io) { return sqrtf((float)(2 * nl[io] + 1) / (float)fan[io]); };
    auto inst = [&](int i1, int i2, int io, float k) { return rows_of(inst_off[i1][i2][io], out[io], in[i1], coeff(io) * k); };
    if (out[0]) {
      parts[0].push_back({T_RA, F_A, -1, inst(0, 0, 0, 1.0f)});                                   // 0e x Y0 -> 0e : a*s0
      if (in[1]) parts[0].push_back({T_RT, 0, 0, inst(1, 1, 0, 1.0f)});                           // 1o x Y1 -> 0e : (p.v)/sqrt3
    }
    if (out[1]) {
      parts[1].push_back({T_RA, F_A, -1, inst(0, 1, 1, is3)});                                    // 0e x Y1 -> 1o : a v / sqrt3
      if (in[1]) {
        std::vector<RowSrc> r = inst(1, 0, 1, is3);                                               // 1o x Y0 -> 1o : p*s0 / sqrt3
        if (in[2]) r = cat(r, inst(2, 1, 1, is3));                                                // 1e x Y1 -> 1o : (q x v)/sqrt6
        parts[1].push_back({T_TV, F_T1O, -1, r});
        parts[1].push_back({T_TV, F_T2O, -1, inst(1, 2, 1, kappa)});                              // 1o x Y2 -> 1o
      }
    }
    if (out[2]) {
      if (in[1]) {
        std::vector<RowSrc> r = inst(1, 1, 2, is3);                                               // 1o x Y1 -> 1e : (p x v)/sqrt6
        if (in[2]) r = cat(r, inst(2, 0, 2, is3));                                                // 1e x Y0 -> 1e : q*s0 / sqrt3
        parts[2].push_back({T_TV, F_T1E, -1, r});
      }
      if (in[3]) parts[2].push_back({T_RA, F_C, -1, inst(3, 1, 2, is3)});                         // 0o x Y1 -> 1e : c v / sqrt3
      if (in[2]) parts[2].push_back({T_TV, F_T2E, -1, inst(2, 2, 2, kappa)});                     // 1e x Y2 -> 1e
    }
    if (out[3]) {
      if (in[2]) parts[3].push_back({T_RT, 0, 1, inst(2, 1, 3, 1.0f)});                           // 1e x Y1 -> 0o : (q.v)/sqrt3
      if (in[3]) parts[3].push_back({T_RA, F_C, -1, inst(3, 0, 3, 1.0f)});                        // 0o x Y0 -> 0o : c*s0
    }
  }

  int oc[4] = {0, out[0], out[0] + 3 * out[1], out[0] + 3 * out[1] + 3 * out[2]};
  if (mode == 2) { oc[0] = ns; oc[3] = 0; L.dout = 2 * ns; }       // '24x0o + 24x0e': the 0o channels come first
  if (mode == 3) { oc[1] = 0; oc[2] = 6; L.dout = 12; }
  // per tile row j: the block it belongs to (the shared tail tile holds two); has_x: accumulator quad rq = 3 of a 6-channel column carries
  // the rows xr[] of ANOTHER row quad for the channel pair xpair (see pack_quads below)
  struct TRow { int blk[4], col; RowSrc r[4]; bool ok[4]; bool has_x = false; int xpair = 0; RowSrc xr[4]; };
  std::vector<TRow> trows;
  tiles.clear();
  L.n_cols = 0;
  auto chan0_of = [&](int b, int col) { return oc[b] + ((b == 1 || b == 2) ? 3 : 1) * 8 * col; };
  auto nch_of = [&](int b, int col) { return L.n_out[b] - 8 * col < 8 ? L.n_out[b] - 8 * col : 8; };
  auto emit = [&](int b, int col, const Part& p, int f_off, int row0, int jlo, int cnt) {
    tiles.push_back(make_tile(p.kind, f_off, FL_NONE, nch_of(b, col) / 2, chan0_of(b, col)));
    TRow t; t.col = col;
    for (int j = 0; j < 4; ++j) { t.blk[j] = b; t.ok[j] = j >= jlo && j < jlo + cnt; if (t.ok[j]) t.r[j] = p.rows[row0 + j - jlo]; }
    trows.push_back(t);
  };
  // all tiles of one part of one output column; a dot-product part's tail (rows 4, 5: the [pv4 pv5 qv4 qv5] quad) can be left to the caller
  auto emit_part = [&](int b, int col, const Part& p, bool with_tail) -> int {
    const int n = (int)p.rows.size();
    if (p.dot_which >= 0) {     // F_PQ = [pv0..3 | qv0..3 | pv4 pv5 qv4 qv5]
      if (n > 6) return fail(ctx, DDK_ERR_INVALID, "dot-product parts hold at most 6 rows");
      emit(b, col, p, F_PQ + 4 * p.dot_which, 0, 0, n < 4 ? n : 4);
      if (n > 4 && with_tail) emit(b, col, p, F_PQ + 8, 4, 2 * p.dot_which, n - 4);
    } else {
      for (int q = 0; 4 * q < n; ++q) emit(b, col, p, p.f_off + (p.kind == T_TV ? 12 * q : 4 * q), 4 * q, 0, n - 4 * q < 4 ? n - 4 * q : 4);
    }
    return DDK_OK;
  };
  auto dot_part = [&](int b) -> const Part* {
    for (const Part& p : parts[b]) if (p.dot_which >= 0 && p.rows.size() > 4) return &p;
    return nullptr;
  };
  // The 0e and 0o blocks' dot-product parts (p.v, q.v: 6 rows each) would each end in a half-empty tile (pv4 pv5 . . / . . qv4 qv5).  When both
  // exist with the same output width they share ONE tile (kind T_RTS): the two columns of the same channel slots are laid out back to back,
  // [0e column ... | shared tail: flushes the 0e column, opens the 0o column | 0o column ...]  (3 tiles less for W = 1872 and W = 1152).
  const Part *dp0 = dot_part(0), *dp3 = dot_part(3);
  const bool share = mode == 0 && dp0 && dp3 && dp0->rows.size() == 6 && dp3->rows.size() == 6 && L.n_out[0] == L.n_out[3] &&
                     L.n_out[0] % 2 == 0;
  bool done[4] = {false, false, false, false};
  if (share) {
    for (int col = 0; 8 * col < L.n_out[0]; ++col) {
      if (L.n_cols >= 16) return fail(ctx, DDK_ERR_INVALID, "too many output columns");
      L.col_start[L.n_cols++] = (int)tiles.size();            // (one split point per PAIR of columns: the shared tile binds them)
      for (const Part& p : parts[0])
        if (int rc = emit_part(0, col, p, false)) return rc;
      tiles.push_back(make_tile(T_RTS, F_PQ + 8, FL_S, nch_of(0, col) / 2, chan0_of(0, col)));
      TRow t; t.col = col;
      for (int j = 0; j < 4; ++j) { t.blk[j] = j < 2 ? 0 : 3; t.ok[j] = true; t.r[j] = j < 2 ? dp0->rows[4 + j] : dp3->rows[4 + j - 2]; }
      trows.push_back(t);
      for (const Part& p : parts[3])
        if (int rc = emit_part(3, col, p, false)) return rc;
      tiles.back().w0 |= FL_S << 2;
    }
    done[0] = done[3] = true;
  }
  // A column of 6 output channels (the vector blocks: nv = 6) fills only 3 of a tile's 4 accumulator quads.  The 4th quad of the column's
  // first tiles carries the (row quad, channel pair) units of the column's LAST a (x) v / c (x) v row quads instead, whose own tiles disappear:
  // Q row quads -> ceil(3Q/4) tiles (9 -> 7 for in = 36 rows, 8 -> 6, 6 -> 5).  Tile word: bit 7 = extra unit present, bits 8-9 = its channel
  // pair, bits 10-13 = its F offset / 4 (an a / c quad).
  const bool pack_quads = mode == 0 || mode == 1;
  for (int b = 0; b < 4; ++b) {
    if (done[b] || parts[b].empty() || L.n_out[b] == 0) continue;
    if (L.n_out[b] % 2) return fail(ctx, DDK_ERR_INVALID, "odd output multiplicity unsupported");
    const bool vec = (b == 1 || b == 2);
    for (int col = 0; 8 * col < L.n_out[b]; ++col) {
      if (L.n_cols >= 16) return fail(ctx, DDK_ERR_INVALID, "too many output columns");
      L.col_start[L.n_cols++] = (int)tiles.size();
      const int t_first = (int)tiles.size();
      for (const Part& p : parts[b])
        if (int rc = emit_part(b, col, p, true)) return rc;
      if (pack_quads && vec && nch_of(b, col) == 6) {
        const int Q = (int)tiles.size() - t_first, T = (3 * Q + 3) / 4, n_x = Q - T;
        // the extras: the last n_x full scalar-row quads (kind T_RA, rows a or c) of the column
        std::vector<int> xs;
        for (int t = (int)tiles.size() - 1; t >= t_first && (int)xs.size() < n_x; --t) {
          const TRow& tr = trows[t];
          if ((tiles[t].w0 & 3) == T_RA && tr.ok[0] && tr.ok[1] && tr.ok[2] && tr.ok[3] && (tiles[t].w0 >> 16) % 4 == 0 && (tiles[t].w0 >> 16) < 64) xs.push_back(t);
        }
        if (n_x > 0 && (int)xs.size() == n_x) {
          std::vector<TileDesc> keep_t; std::vector<TRow> keep_r, x_r; std::vector<int> x_off;
          for (int t = t_first; t < (int)tiles.size(); ++t) {
            if (std::find(xs.begin(), xs.end(), t) != xs.end()) { x_r.push_back(trows[t]); x_off.push_back(tiles[t].w0 >> 16); }
            else { keep_t.push_back(tiles[t]); keep_r.push_back(trows[t]); }
          }
          for (int u = 0; u < 3 * n_x; ++u) {       // unit u = (extra quad u / 3, channel pair u % 3) rides on the column's tile u
            keep_r[u].has_x = true; keep_r[u].xpair = u % 3;
            for (int j = 0; j < 4; ++j) keep_r[u].xr[j] = x_r[u / 3].r[j];
            keep_t[u].w0 |= 0x80 | ((u % 3) << 8) | ((x_off[u / 3] / 4) << 10);
          }
          tiles.resize(t_first); trows.resize(t_first);
          tiles.insert(tiles.end(), keep_t.begin(), keep_t.end());
          trows.insert(trows.end(), keep_r.begin(), keep_r.end());
        }
      }
      tiles.back().w0 |= (vec ? FL_V : FL_S) << 2;
    }
  }
  if (mode == 3 && !ctx->cfg.deterministic) {      // (the deterministic scatter STORES: every output channel must be flushed exactly once)
    // final_conv has a handful of edge blocks (B * n_lig edges): the kernel's short-queue split hands out COLUMNS, so cut its two 9-tile columns
    // into flush columns of two tiles (a flush adds the partial sums to the same output channels: the sum is what counts)
    L.n_cols = 0;
    for (int t = 0; t < (int)tiles.size(); ++t) {
      const bool prev_flushes = t > 0 && ((tiles[t - 1].w0 >> 2) & 3) != FL_NONE;
      if (t == 0 || prev_flushes || t - L.col_start[L.n_cols - 1] == 2) {
        if (L.n_cols >= 16) return fail(ctx, DDK_ERR_INVALID, "too many output columns");
        if (t > 0 && !prev_flushes) tiles[t - 1].w0 |= FL_V << 2;
        L.col_start[L.n_cols++] = t;
      }
    }
  }
  L.n_tiles = (int)tiles.size();
  L.col_start[L.n_cols] = L.n_tiles;
  L.h_tiles = tiles;

  // row map: tile row rho = 8*rq + 4*hh + j  ->  index into the reference weight vector (or -1 = zero row) and its scale
  rowmap.assign((size_t)L.n_tiles * 32, -1);
  rowscale.assign((size_t)L.n_tiles * 32, 0.f);
  for (int t = 0; t < L.n_tiles; ++t) {
    const TRow& tr = trows[t];
    for (int rq = 0; rq < 4; ++rq)
      for (int hh = 0; hh < 2; ++hh)
        for (int j = 0; j < 4; ++j) {
          if (tr.has_x && rq == 3) {      // the extra unit: rows of another quad, channel pair xpair of the same column
            rowmap[(size_t)t * 32 + 8 * rq + 4 * hh + j] = tr.xr[j].wbase + 8 * tr.col + 2 * tr.xpair + hh;
            rowscale[(size_t)t * 32 + 8 * rq + 4 * hh + j] = tr.xr[j].scale;
            continue;
          }
          const int k = 8 * tr.col + 2 * rq + hh;
          if (!tr.ok[j] || k >= L.n_out[tr.blk[j]]) continue;
          rowmap[(size_t)t * 32 + 8 * rq + 4 * hh + j] = tr.r[j].wbase + k;
          rowscale[(size_t)t * 32 + 8 * rq + 4 * hh + j] = tr.r[j].scale;
        }
  }
  // every weight row must be used exactly once
  std::vector<int> cnt(L.W, 0);
  for (int r : rowmap) if (r >= 0) cnt[r]++;
  for (int r = 0; r < L.W; ++r) if (cnt[r] != 1) return fail(ctx, DDK_ERR_INVALID, "internal: weight row map is not a bijection");
  return DDK_OK;
}

// e3nn BatchNorm (eval) folded to per-channel mean / scale / bias over the padded XW columns
static int fold_batch_norm(ddk_ctx* ctx, const std::string& pre, const int* out, float* mean, float* scale, float* bias) {
  for (int i = 0; i < XW; ++i) { mean[i] = 0.f; scale[i] = 1.f; bias[i] = 0.f; }
  const int nf = out[0] + out[1] + out[2] + out[3];
  const HostTensor* bw = find_w(ctx, pre + ".weight", {nf});
  const HostTensor* bb = find_w(ctx, pre + ".bias", {out[0]});
  const HostTensor* bm = find_w(ctx, pre + ".running_mean", {out[0]});
  const HostTensor* bv = find_w(ctx, pre + ".running_var", {nf});
  if (!bw || !bb || !bm || !bv) return DDK_ERR_INVALID;
  int ch = 0, f = 0;
  const int dims[4] = {1, 3, 3, 1};
  for (int b = 0; b < 4; ++b)
    for (int m = 0; m < out[b]; ++m, ++f) {
      const float sc = powf(bv->data[f] + 1e-5f, -0.5f) * bw->data[f];
      for (int d = 0; d < dims[b]; ++d, ++ch) {
        scale[ch] = sc;
        if (b == 0) { mean[ch] = bm->data[m]; bias[ch] = bb->data[m]; }
      }
    }
  return DDK_OK;
}

// Three-limb fp16 records of a layer's radial-MLP weights for k_conv_x.hip: every (range-scaled) fp32 weight v becomes
// hi + mid + lo with hi = fp16(v), mid = fp16(v - hi), lo = fp16(v - hi - mid), each limb carrying its own weight (fp16 subnormals included).
// Exact whenever the last bit of v is a multiple of the fp16 subnormal step 2^-24, i.e. for |v| >= 0.5 after scaling (the group's maximum is
// scaled into [2^14, 2^15)); smaller values are off by at most 2^-25 = 2^-39 of the maximum.  Checked here for every value.
// w1all / w2all: the fp32 fragment arrays [.][s/4][lane][s&3] (s = register of the lane half), b2all [t][2][16].
static int pack_x3(ddk_ctx* ctx, ConvLayerDev& L, int NG, const std::vector<float>& w1all, const std::vector<float>& w2all,
                   const std::vector<float>& b2all) {
  const size_t w1sz = 3 * 9 * 64 * 4, w2sz = (size_t)L.n_tiles * 9 * 64 * 4, b2sz = (size_t)L.n_tiles * 32;
  bool exact = true;
  auto split = [&exact](float v, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    const _Float16 h = (_Float16)v;
    const float r1 = v - (float)h;
    const _Float16 m = (_Float16)r1;
    const _Float16 l = (_Float16)(r1 - (float)m);
    memcpy(&hi, &h, 2); memcpy(&mid, &m, 2); memcpy(&lo, &l, 2);
    // the limbs must reproduce v bit for bit (values below 0.5 - more than 2^-15 under the group's maximum - within 2^-25)
    const double back = (double)(float)h + (double)(float)m + (double)(float)l;
    if (std::fabs(v) >= 0.5f ? back != (double)v : std::fabs(back - (double)v) > 0x1p-25) exact = false;
  };
  // exact power-of-two range scaling: max|w| of a group is brought into [2^14, 2^15) so that no limb leaves the fp16 range whatever the
  // scale of the checkpoint (the kernel scales the activations per edge the same way)
  auto range_scale = [](const float* v, size_t n, const float* v2, size_t n2) {
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(v[i]));
    for (size_t i = 0; i < n2; ++i) m = std::max(m, std::fabs(v2[i]));
    int e = 0;
    std::frexp(std::max(m, 0x1p-40f), &e);      // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.0f, 15 - e);
  };
  // (+ three zero records behind the last group: the kernel requests record t+3 without clamping at a group's last tile)
  std::vector<uint8_t> w2x((size_t)(NG * L.n_tiles + 4) * W2X_TILE_BYTES, 0), w1x((size_t)NG * 3 * W1X_TILE_BYTES, 0);
  // one tile of fp32 fragments [9][64][4] -> three limbs x [4 x [64][8] | [64][4]]
  auto frags = [&](const float* src, float sc, uint8_t* dst) {
    for (int r = 0; r < 36; ++r)
      for (int lane = 0; lane < 64; ++lane) {
        const float v = src[((size_t)(r / 4) * 64 + lane) * 4 + (r & 3)] * sc;
        const int s_ = r / 8, i = r % 8;
        const size_t off = s_ < 4 ? (size_t)s_ * 1024 + lane * 16 + 2 * i : (size_t)4096 + lane * 8 + 2 * i;
        uint16_t h, m, l;
        split(v, h, m, l);
        memcpy(dst + off, &h, 2); memcpy(dst + W2X_LIMB_BYTES + off, &m, 2); memcpy(dst + 2 * W2X_LIMB_BYTES + off, &l, 2);
      }
  };
  for (int g = 0; g < NG; ++g) {
    const float* w2 = w2all.data() + g * w2sz;
    const float* w1 = w1all.data() + g * w1sz;
    const float* b2 = b2all.data() + g * b2sz;
    const float sc1 = range_scale(w1, w1sz, nullptr, 0), sc2 = range_scale(w2, w2sz, nullptr, 0);
    L.w1s[g] = sc1; L.w2s[g] = sc2;
    for (int t = 0; t < L.n_tiles; ++t) {
      uint8_t* rec = w2x.data() + ((size_t)g * L.n_tiles + t) * W2X_TILE_BYTES;
      frags(w2 + (size_t)t * 2304, sc2, rec);
      memcpy(rec + W2X_BIAS_OFF, b2 + (size_t)t * 32, 128);       // fp32 as is: the kernel scales it like the products
      const int32_t dq[2] = {x_tile_word(L.h_tiles[t].w0), L.h_tiles[t].chan0};
      if (dq[0] < 0) return fail(ctx, DDK_ERR_INVALID, "internal: a vector tile of the conv layout does not sit on a T1O / T1E row quad");
      memcpy(rec + W2X_DESC_OFF, dq, 8);
    }
    for (int T = 0; T < 3; ++T) frags(w1 + (size_t)T * 2304, sc1, w1x.data() + ((size_t)g * 3 + T) * W1X_TILE_BYTES);
  }
  if (!exact) return fail(ctx, DDK_ERR_INVALID, "internal: the three-limb fp16 split of a conv weight is not exact");
  L.h_w2x = w2x; L.h_w1x = w1x;
  L.epi_ok = conv_epilogue_shapes_ok(L.h_tiles);      // launch_conv_fused_x refuses the asm-epilogue instantiation otherwise
  L.limbs = ctx->cfg.conv_kernel == 3 ? 3 : 2;        // the default two-limb form reads the first two limbs of the same records (k_conv_x2.hip); conv_kernel = 3: all three, six products
  if (ctx->host_only) return DDK_OK;
  L.w2x = (uint8_t*)dev_alloc(ctx, w2x.size());
  L.w1x = (uint8_t*)dev_alloc(ctx, w1x.size());
  if (!L.w2x || !L.w1x) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed while packing conv weights (three-limb f16)");
  if (hipMemcpy(L.w2x, w2x.data(), w2x.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(L.w1x, w1x.data(), w1x.size(), hipMemcpyHostToDevice) != hipSuccess)
    return fail(ctx, DDK_ERR_HIP, "three-limb weight upload failed");
  return DDK_OK;
}

// mode 0: score-model layer l = conv_layers.{l} with 4 edge groups (fc.{g}.{0,4}) and one BatchNorm;
// mode 1: confidence-model layer l = the 9 convs conv_layers.{9l+g} (fc.{0,3}), each with its own BatchNorm.
static int build_conv_layer(ddk_ctx* ctx, int mode, int l, ConvLayerDev& L) {
  const ddk_config& c = ctx->cfg;
  const int ns = c.ns;
  std::vector<int> rowmap;
  std::vector<float> rowscale;
  std::vector<TileDesc> tiles;
  int rc = build_layout(ctx, mode, l, L, rowmap, rowscale, tiles);
  if (rc) return rc;
  const int NG = mode == 0 ? 4 : 9;
  L.n_groups = NG;
  const int ne = 3 * ns;
  auto fc_name = [&](int g) { return mode == 0 ? "conv_layers." + std::to_string(l) + ".fc." + std::to_string(g) : "conv_layers." + std::to_string(9 * l + g) + ".fc"; };
  const std::string lin2 = mode == 0 ? ".4" : ".3";
  if (ctx->weights.find(fc_name(0) + ".0.weight") == ctx->weights.end()) {
    L.has_weights = false;     // shape-only layer: ddk_tp_forward works, ddk_conv_forward refuses
    return DDK_OK;
  }
  L.has_weights = true;
  const size_t w1sz = 3 * 9 * 64 * 4, b1sz = 3 * 2 * 16, w2sz = (size_t)L.n_tiles * 9 * 64 * 4, b2sz = (size_t)L.n_tiles * 32;
  std::vector<float> w1all(NG * w1sz, 0.f), b1all(NG * b1sz, 0.f), w2all(NG * w2sz, 0.f), b2all(NG * b2sz, 0.f);
  for (int g = 0; g < NG; ++g) {
    const std::string f = fc_name(g);
    const HostTensor* W1 = find_w(ctx, f + ".0.weight", {ne, ne});
    const HostTensor* B1 = find_w(ctx, f + ".0.bias", {ne});
    const HostTensor* W2 = find_w(ctx, f + lin2 + ".weight", {L.W, ne});
    const HostTensor* B2 = find_w(ctx, f + lin2 + ".bias", {L.W});
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
  // node terms of GEMM1 (score model): see ConvLayerDev::wn
  if (mode == 0) {
    L.h_wn.assign((size_t)2 * 4 * NE * NS, 0.f);
    L.h_bnp.assign((size_t)2 * 4 * NE, 0.f);
    const int recv_g[2][2] = {{0, 1}, {2, 3}}, send_g[2][2] = {{0, 3}, {1, 2}};     // [node type][slot]: ligand atom / residue
    for (int type = 0; type < 2; ++type)
      for (int slot = 0; slot < 4; ++slot) {
        const int g = slot < 2 ? recv_g[type][slot] : send_g[type][slot - 2];
        const HostTensor* W1 = find_w(ctx, fc_name(g) + ".0.weight", {ne, ne});
        const HostTensor* B1 = find_w(ctx, fc_name(g) + ".0.bias", {ne});
        if (!W1 || !B1) return DDK_ERR_INVALID;
        const int col0 = slot < 2 ? NS : 2 * NS;       // x[edge_src][:ns] columns (receiver) / x[edge_dst][:ns] columns (sender)
        for (int o = 0; o < ne; ++o) {
          const size_t row = ((size_t)(type * 4 + slot) * NE + pre_pos(o));
          for (int k = 0; k < NS; ++k) L.h_wn[row * NS + k] = W1->data[(size_t)o * ne + col0 + k];
          L.h_bnp[row] = slot < 2 ? B1->data[o] : 0.f;       // the bias rides with the receiver's term
        }
      }
  }
  // BatchNorm (e3nn, eval): per multiplicity channel; one per layer (mode 0) or one per conv (mode 1)
  const int n_bn = mode == 0 ? 1 : NG;
  L.h_bn_mean.assign((size_t)n_bn * XW, 0.f);
  L.h_bn_scale.assign((size_t)n_bn * XW, 1.f);
  L.h_bn_bias.assign((size_t)n_bn * XW, 0.f);
  if (c.batch_norm)
    for (int g = 0; g < n_bn; ++g) {
      const std::string pre = "conv_layers." + std::to_string(mode == 0 ? l : 9 * l + g) + ".batch_norm";
      if ((rc = fold_batch_norm(ctx, pre, L.out_mul, L.h_bn_mean.data() + (size_t)g * XW, L.h_bn_scale.data() + (size_t)g * XW,
                                L.h_bn_bias.data() + (size_t)g * XW)))
        return rc;
    }
  L.h_w1p.resize(NG); L.h_b1p.resize(NG); L.h_w2p.resize(NG); L.h_b2p.resize(NG);
  for (int g = 0; g < NG; ++g) {
    L.h_w1p[g].assign(w1all.begin() + g * w1sz, w1all.begin() + (g + 1) * w1sz);
    L.h_b1p[g].assign(b1all.begin() + g * b1sz, b1all.begin() + (g + 1) * b1sz);
    L.h_w2p[g].assign(w2all.begin() + g * w2sz, w2all.begin() + (g + 1) * w2sz);
    L.h_b2p[g].assign(b2all.begin() + g * b2sz, b2all.begin() + (g + 1) * b2sz);
  }
  if (mode == 0 && c.conv_kernel != 1 && (rc = pack_x3(ctx, L, NG, w1all, w2all, b2all))) return rc;
  if (mode == 1 && c.conv_kernel != 1) {
    // The three-limb kernel keeps the raw p / q rows once ([p0..p5 | q0..q5], quads of four): the l = 2 group of the q rows (T2E), whose two
    // tiles are [q0 q1 q2 q3], [q4 q5 . .] in the table, reads raw quads 1 and 2 = [. . q0 q1], [q2 q3 q4 q5]: move its weight rows accordingly
    // (accumulator quads 0..2 only: quad 3 of a 6-channel column is empty or carries a packed extra unit).  x_tile_word() gives the tiles their offsets.
    std::vector<int> rmx = rowmap;
    std::vector<float> rsx = rowscale;
    for (int t = 0; t + 1 < L.n_tiles; ++t) {
      if ((tiles[t].w0 & 3) != T_TV || (tiles[t].w0 >> 16) != F_T2E) continue;
      if ((tiles[t + 1].w0 & 3) != T_TV || (tiles[t + 1].w0 >> 16) != F_T2E + 12) return fail(ctx, DDK_ERR_INVALID, "internal: the two tiles of a T2E row group are not adjacent");
      for (int rq = 0; rq < 3; ++rq)
        for (int hh = 0; hh < 2; ++hh) {
          const size_t a = (size_t)t * 32 + 8 * rq + 4 * hh, b = (size_t)(t + 1) * 32 + 8 * rq + 4 * hh;
          if (rowmap[b + 2] >= 0 || rowmap[b + 3] >= 0) return fail(ctx, DDK_ERR_INVALID, "internal: a T2E row group holds more than six rows");
          for (int j = 0; j < 2; ++j) {
            rmx[a + j] = -1; rsx[a + j] = 0.f;
            rmx[a + 2 + j] = rowmap[a + j]; rsx[a + 2 + j] = rowscale[a + j];
            rmx[b + j] = rowmap[a + 2 + j]; rsx[b + j] = rowscale[a + 2 + j];
            rmx[b + 2 + j] = rowmap[b + j]; rsx[b + 2 + j] = rowscale[b + j];
          }
        }
    }
    std::vector<float> w2x_all(NG * w2sz, 0.f), b2x_all(NG * b2sz, 0.f);
    for (int g = 0; g < NG; ++g) {
      const HostTensor* W2 = find_w(ctx, fc_name(g) + lin2 + ".weight", {L.W, ne});
      const HostTensor* B2 = find_w(ctx, fc_name(g) + lin2 + ".bias", {L.W});
      if (!W2 || !B2) return DDK_ERR_INVALID;
      float* w2 = w2x_all.data() + g * w2sz;
      float* b2 = b2x_all.data() + g * b2sz;
      for (int t = 0; t < L.n_tiles; ++t) {
        for (int s_ = 0; s_ < 36; ++s_)
          for (int lane = 0; lane < 64; ++lane) {
            const int row = rmx[(size_t)t * 32 + (lane & 31)], hh = lane >> 5;
            w2[(((size_t)t * 9 + s_ / 4) * 64 + lane) * 4 + (s_ & 3)] = row >= 0 ? W2->data[(size_t)row * ne + hid_of(s_, hh)] * rsx[(size_t)t * 32 + (lane & 31)] : 0.f;
          }
        for (int hh = 0; hh < 2; ++hh)
          for (int r = 0; r < 16; ++r) {
            const int row = rmx[(size_t)t * 32 + d_row(r, hh)];
            b2[((size_t)t * 2 + hh) * 16 + r] = row >= 0 ? B2->data[row] * rsx[(size_t)t * 32 + d_row(r, hh)] : 0.f;
          }
      }
    }
    if ((rc = pack_x3(ctx, L, NG, w1all, w2x_all, b2x_all))) return rc;
  }
  if (!ctx->host_only) {
    std::vector<float> w2rec((size_t)NG * L.n_tiles * W2_TILE_FLOATS);
    for (int g = 0; g < NG; ++g)
      for (int t = 0; t < L.n_tiles; ++t) {
        float* rec = w2rec.data() + ((size_t)g * L.n_tiles + t) * W2_TILE_FLOATS;
        memcpy(rec, w2all.data() + g * w2sz + (size_t)t * 2304, 2304 * sizeof(float));
        memcpy(rec + 2304, b2all.data() + g * b2sz + (size_t)t * 32, 32 * sizeof(float));
        memcpy(rec + 2336, &tiles[t], 2 * sizeof(int32_t));   // descriptor rides with the record (bit pattern)
      }
    float* d1 = dev_upload(ctx, w1all);
    float* db1 = dev_upload(ctx, b1all);
    float* d2 = dev_upload(ctx, w2rec);
    if (mode == 0) { L.wn = dev_upload(ctx, L.h_wn); L.bnp = dev_upload(ctx, L.h_bnp); if (!L.wn || !L.bnp) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed while packing conv weights"); }
    L.bn_mean = dev_upload(ctx, L.h_bn_mean);
    L.bn_scale = dev_upload(ctx, L.h_bn_scale);
    L.bn_bias = dev_upload(ctx, L.h_bn_bias);
    if (!d1 || !db1 || !d2 || !L.bn_mean || !L.bn_scale || !L.bn_bias)
      return fail(ctx, DDK_ERR_NOMEM, "device allocation failed while packing conv weights");
    L.w1p[0] = d1; L.b1p[0] = db1; L.w2r[0] = d2;    // group-major contiguous: group g at + g * stride
    for (int g = 1; g < 4; ++g) {
      L.w1p[g] = d1 + g * w1sz;
      L.b1p[g] = db1 + g * b1sz;
      L.w2r[g] = d2 + (size_t)g * L.n_tiles * W2_TILE_FLOATS;
    }
  }
  return DDK_OK;
}

// tor_bond_conv (mode 2, radial MLP 72 -> 72 -> 288) and final_conv (mode 3, 48 -> 48 -> 144, zero padded to the kernel's 72-wide GEMMs)
// packed like one edge group of a conv layer: the heads run through conv_fused_kernel<GATHER = false> on explicit edge attributes
int build_head_layer(ddk_ctx* ctx, int mode, ConvLayerDev& L) {
  std::vector<int> rowmap;
  std::vector<float> rowscale;
  std::vector<TileDesc> tiles;
  int rc = build_layout(ctx, mode, 3, L, rowmap, rowscale, tiles);
  if (rc) return rc;
  L.n_groups = 1;
  const std::string pre = mode == 2 ? "tor_bond_conv.fc" : "final_conv.fc";
  const int kin = mode == 2 ? NE : 2 * NS;          // true width of the MLP's input and hidden layer
  const HostTensor* W1 = find_w(ctx, pre + ".0.weight", {kin, kin});
  const HostTensor* B1 = find_w(ctx, pre + ".0.bias", {kin});
  const HostTensor* W2 = find_w(ctx, pre + ".4.weight", {L.W, kin});
  const HostTensor* B2 = find_w(ctx, pre + ".4.bias", {L.W});
  if (!W1 || !B1 || !W2 || !B2) return DDK_ERR_INVALID;
  L.has_weights = true;
  std::vector<float> w1(3 * 9 * 64 * 4, 0.f), b1(3 * 2 * 16, 0.f), w2((size_t)L.n_tiles * 9 * 64 * 4, 0.f), b2((size_t)L.n_tiles * 32, 0.f);
  for (int T = 0; T < 3; ++T) {
    for (int s = 0; s < 36; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int hidden = 32 * T + (lane & 31), k = kin_of(s, lane >> 5);
        w1[(((size_t)T * 9 + s / 4) * 64 + lane) * 4 + (s & 3)] = (hidden < kin && k < kin) ? W1->data[(size_t)hidden * kin + k] : 0.f;
      }
    for (int hh = 0; hh < 2; ++hh)
      for (int r = 0; r < 16; ++r) {
        const int hidden = 32 * T + d_row(r, hh);
        b1[(T * 2 + hh) * 16 + r] = hidden < kin ? B1->data[hidden] : 0.f;
      }
  }
  for (int t = 0; t < L.n_tiles; ++t) {
    for (int s = 0; s < 36; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = rowmap[(size_t)t * 32 + (lane & 31)], hd = hid_of(s, lane >> 5);
        w2[(((size_t)t * 9 + s / 4) * 64 + lane) * 4 + (s & 3)] =
            (row >= 0 && hd < kin) ? W2->data[(size_t)row * kin + hd] * rowscale[(size_t)t * 32 + (lane & 31)] : 0.f;
      }
    for (int hh = 0; hh < 2; ++hh)
      for (int r = 0; r < 16; ++r) {
        const int row = rowmap[(size_t)t * 32 + d_row(r, hh)];
        b2[((size_t)t * 2 + hh) * 16 + r] = row >= 0 ? B2->data[row] * rowscale[(size_t)t * 32 + d_row(r, hh)] : 0.f;
      }
  }
  L.h_w1p.assign(1, w1); L.h_b1p.assign(1, b1); L.h_w2p.assign(1, w2); L.h_b2p.assign(1, b2);
  L.h_bn_mean.assign(XW, 0.f); L.h_bn_scale.assign(XW, 1.f); L.h_bn_bias.assign(XW, 0.f);
  if (ctx->cfg.conv_kernel != 1 && (rc = pack_x3(ctx, L, 1, w1, w2, b2))) return rc;
  if (ctx->host_only) return DDK_OK;
  std::vector<float> w2rec((size_t)L.n_tiles * W2_TILE_FLOATS);
  for (int t = 0; t < L.n_tiles; ++t) {
    float* rec = w2rec.data() + (size_t)t * W2_TILE_FLOATS;
    memcpy(rec, w2.data() + (size_t)t * 2304, 2304 * sizeof(float));
    memcpy(rec + 2304, b2.data() + (size_t)t * 32, 32 * sizeof(float));
    memcpy(rec + 2336, &tiles[t], 2 * sizeof(int32_t));
  }
  L.w1p[0] = dev_upload(ctx, w1); L.b1p[0] = dev_upload(ctx, b1); L.w2r[0] = dev_upload(ctx, w2rec);
  if (!L.w1p[0] || !L.b1p[0] || !L.w2r[0]) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed while packing the head weights");
  return DDK_OK;
}

int model_finalize(ddk_ctx* ctx);   // model.hip
int conf_model_finalize(ddk_ctx* ctx);   // conf.hip
void conf_model_destroy(ddk_ctx* ctx);
void model_destroy(ddk_ctx* ctx);   // model.hip

}  // namespace ddk

extern "C" {

const char* ddk_version(void) { return "ddk 0.8 (gfx950)"; }      // 0.8: conv_kernel = 0 is the two-limb / three-product form of the f16-limb kernel (k_conv_x2.hip), the three-limb / six-product form (the default of 0.4 - 0.7) is conv_kernel = 3; 0.7: conv_kernel = 2 left the library (tools/variants/), ddk_tp_forward walks columns (flat 16-B reads), ddk_debug_set_alloc_limit; 0.6: conv_kernel = 2 (k_conv_y.hip), streaming ddk_tp_forward, confidence edge capacity from geometry; 0.5: ddk_config.confidence_mode + ddk_score_confidence; 0.4: three-limb records carry limbs at their own weight + tile descriptors; conv_f16x3 removed (INTEGRATION.md "ABI notes")

int ddk_create(const ddk_config* cfg, ddk_ctx** out) {
  if (!cfg || !out) return DDK_ERR_INVALID;
  ddk_ctx* ctx = new ddk_ctx();
  ctx->cfg = *cfg;
  *out = ctx;
  if (cfg->ns != NS || cfg->nv != NV)
    return fail(ctx, DDK_ERR_INVALID, "only ns=24, nv=6 (DiffDock-S / DisCo-DiffDock-S) is compiled in");
  if (cfg->num_conv_layers < 4 || cfg->num_conv_layers > 16)
    return fail(ctx, DDK_ERR_INVALID, "num_conv_layers must be in [4,16] (the heads and the confidence predictor assume the full 0e+1o+1e+0o irreps)");
  if (cfg->confidence_mode && cfg->all_atoms)
    return fail(ctx, DDK_ERR_INVALID, "confidence_mode selects the COARSE-GRAINED score model's confidence head; the all-atom confidence model is all_atoms = 1 alone (include/ddk.h)");
  if (cfg->sigma_embed_dim != 32 || cfg->distance_embed_dim != 32 || cfg->cross_distance_embed_dim != 32)
    return fail(ctx, DDK_ERR_INVALID, "only 32-wide sigma / distance embeddings are compiled in");
  if (cfg->deterministic && cfg->all_atoms)
    return fail(ctx, DDK_ERR_INVALID, "deterministic scatter is implemented for the score model (not with all_atoms)");
#ifdef DDK_VARIANT_CONV_Y
  if (cfg->conv_kernel < 0 || cfg->conv_kernel > 3) return fail(ctx, DDK_ERR_INVALID, "conv_kernel must be 0, 1, 2 (variant build) or 3");
#else
  if (cfg->conv_kernel < 0 || cfg->conv_kernel > 3 || cfg->conv_kernel == 2)
    return fail(ctx, DDK_ERR_INVALID, "conv_kernel must be 0 (two f16 limbs per operand, three products), 1 (fp32 MFMA) or 3 (three f16 limbs, six products); 2 (round 5's software-pipelined form) lives under tools/variants/ and is not part of libddk.so");
#endif
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
  e = hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamCreate (upload stream)");
  e = hipStreamCreateWithFlags(&ctx->head_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamCreate (head stream)");
  e = conv_prepare_device();
  if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(dynamic LDS)");
  e = graph_prepare_device(&ctx->max_rec);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipFuncSetAttribute(dynamic LDS, graph kernels)");
  ctx->ws.tile_info = (int32_t*)dev_alloc(ctx, 64 * sizeof(int32_t));
  if (!ctx->ws.tile_info) return fail(ctx, DDK_ERR_NOMEM, "hipMalloc failed");
  return DDK_OK;
}

void ddk_destroy(ddk_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->host_only) {
    hipSetDevice(ctx->cfg.device);
    model_destroy(ctx);
    conf_model_destroy(ctx);
    for (auto& r : ctx->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    if (ctx->prof_edges) hipHostFree(ctx->prof_edges);
    for (auto& c : ctx->chunk_pool) { if (c.free_after) hipEventDestroy(c.free_after); ctx_free(ctx, c.p); }
    for (auto& b : ctx->stage_pool) { if (b.done) hipEventDestroy(b.done); hipHostFree(b.p); }
    if (ctx->up_stream) hipStreamDestroy(ctx->up_stream);
    if (ctx->head_stream) hipStreamDestroy(ctx->head_stream);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    for (void* p : ctx->dev_allocs) ctx_free(ctx, p);
    if (ctx->ws.xpad) ctx_free(ctx, ctx->ws.xpad);
    if (ctx->ws.sum) ctx_free(ctx, ctx->ws.sum);
    if (ctx->ws.deg) ctx_free(ctx, ctx->ws.deg);
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
    int rc = build_conv_layer(ctx, ctx->cfg.all_atoms ? 1 : 0, l, ctx->conv[l]);
    if (rc != DDK_OK) return rc;
  }
  // the score model's two heads as layouts of the fused conv kernel (also in host-only contexts: the CPU emulation tests read the packing)
  ctx->head[0] = ConvLayerDev(); ctx->head[1] = ConvLayerDev();
  if (!ctx->cfg.all_atoms && ctx->weights.find("final_conv.fc.0.weight") != ctx->weights.end()) {
    int rch = build_head_layer(ctx, 3, ctx->head[1]);
    if (rch == DDK_OK && !ctx->cfg.no_torsion) rch = build_head_layer(ctx, 2, ctx->head[0]);
    if (rch != DDK_OK) return rch;
  }
  int rc = ctx->cfg.all_atoms ? conf_model_finalize(ctx) : model_finalize(ctx);
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
  if (e == hipErrorInvalidValue)      // (say what is wrong instead of a bare HIP error)
    return fail(ctx, DDK_ERR_INVALID, "ddk_tp_forward: the layer's irreps are not one of the four conv-layer shapes of this model family (24x0e [+ 6x1o [+ 6x1e [+ 24x0o]]] -> the next entry)");
  if (e != hipSuccess) return hip_fail(ctx, e, "tp_forward launch");
  return DDK_OK;
}

int ddk_conv_forward(ddk_ctx* ctx, int32_t layer, const float* x, int64_t N, const int32_t* edge_src,
                     const int32_t* edge_dst, const int64_t* go, const float* edge_attr, const float* sh, float* out,
                     void* stream) {
  // all-atom (confidence) contexts: `layer` is the index of ONE reference conv (conv_layers.{layer}, all_atom_score_model.py:37-50,
  // residual=False): every edge belongs to it, out = BatchNorm(scatter_mean(...))
  const bool aa = ctx && ctx->cfg.all_atoms;
  const int conv_k = aa ? layer % 9 : 0;
  if (aa) layer /= 9;
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
    if (aa) {
      CK(launch_conv_one_group(ws.tile_info + 32, 9, conv_k, E, s), "group table");
      a.mode = 1; a.n_groups = 9; a.n_active = 9; a.n_slots = 1; a.slots = 0; a.gbeg = ws.tile_info + 32; a.gend = ws.tile_info + 41;
    }
    CK(launch_conv_fused(L, a, ctx->n_cu, s), "conv_fused");
  }
  if (aa) {
    CK(launch_node_finalize(ws.sum, ws.deg, nullptr, L.bn_mean + conv_k * XW, L.bn_scale + conv_k * XW, L.bn_bias + conv_k * XW, N, L.dout, L.dout, out, s),
       "node_finalize");
    return DDK_OK;
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
    const bool head = (l == 100 || l == 101) && ctx->head[l - 100].n_tiles > 0;      // conv.100 = tor_bond_conv, conv.101 = final_conv layouts
    if (!head && (l < 0 || l >= (int)ctx->conv.size())) return fail(ctx, DDK_ERR_INVALID, "bad export index");
    ConvLayerDev& L = head ? ctx->head[l - 100] : ctx->conv[l];
    if (g < 0 || g >= L.n_groups) return fail(ctx, DDK_ERR_INVALID, "bad export index");
    const std::string it(item);
    if (it == "w1p") { src = L.h_w1p[g].data(); n = L.h_w1p[g].size(); }
    else if (it == "b1p") { src = L.h_b1p[g].data(); n = L.h_b1p[g].size(); }
    else if (it == "w2p") { src = L.h_w2p[g].data(); n = L.h_w2p[g].size(); }
    else if (it == "b2p") { src = L.h_b2p[g].data(); n = L.h_b2p[g].size(); }
    else if (it == "wn") { src = L.h_wn.data(); n = L.h_wn.size(); }
    else if (it == "bnp") { src = L.h_bnp.data(); n = L.h_bnp.size(); }
    else if (it == "tiles") { src = L.h_tiles.data(); n = L.h_tiles.size() * (sizeof(TileDesc) / 4); }
    else if (it == "bn_mean") { src = L.h_bn_mean.data(); n = L.h_bn_mean.size(); }
    else if (it == "bn_scale") { src = L.h_bn_scale.data(); n = L.h_bn_scale.size(); }
    else if (it == "bn_bias") { src = L.h_bn_bias.data(); n = L.h_bn_bias.size(); }
    // three-limb f16 kernel: all groups' records (bytes, padded to words) and the power-of-two range scales [w1s[4] | w2s[4]]
    else if (it == "w1x") { src = L.h_w1x.data(); n = L.h_w1x.size() / 4; }
    else if (it == "w2x") { src = L.h_w2x.data(); n = L.h_w2x.size() / 4; }
    else if (it == "xscale") { static thread_local float sc[8]; for (int k = 0; k < 4; ++k) { sc[k] = L.w1s[k]; sc[4 + k] = L.w2s[k]; } src = sc; n = 8; }
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

// Test hook: the conv kernel's in-register limb split on a DEVICE array: x [n] is cut into groups of `group` consecutive values, each group is
// range-scaled by its own power of two (like the per-edge scaling of the activations) and split; hi / mid / lo [n] come back as fp32 values
// of the fp16 limbs, scale [n] is the group's power of two:  x * scale == hi + mid 2^-11 + lo 2^-22 exactly (tests/test_gpu_round3.py).
int ddk_debug_split3(ddk_ctx* ctx, const float* x, int64_t n, int32_t group, float* hi, float* mid, float* lo, float* scale, void* stream) {
  if (!ctx || ctx->host_only) return DDK_ERR_STATE;
  if (n < 0 || group < 1 || !x || !hi || !mid || !lo || !scale) return fail(ctx, DDK_ERR_INVALID, "ddk_debug_split3: bad argument");
  hipError_t e = launch_split3_probe(x, n, group, hi, mid, lo, scale, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "split3 probe");
}

}  // extern "C"
