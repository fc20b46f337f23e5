// All-atom confidence model: models/all_atom_score_model.py in confidence_mode as get_model builds it
// from workdir/paper_confidence_model/model_parameters.yml (utils/model_utils.py:25-68): three node types (ligand atoms,
// receptor atoms, residues), nine tensor-product convolutions per layer with the l<=2 e3nn FullyConnectedTensorProduct,
// OldAtomEncoder node embeddings, BatchNorm per conv, scatter-mean pooled ligand scalars -> confidence_predictor MLP.
//
// The forward always runs at t = 0 (utils/sampling.py:236 set_time(..., 0, 0, 0); in confidence_mode complex_t is used as sigma
// directly, all_atom_score_model.py:205-207), so every sigma-embedding product is a constant of the model and everything that
// does not depend on the ligand pose is a constant of the complex:
//   * host, once per model:   sigma-embedding halves of all first layers, folded BatchNorm1d of the predictor;
//   * host, once per complex: the three node embeddings, and edge embedding + spherical harmonics of the static edge sets
//                             (atom-atom, atom->residue and its flip), replicated for max_batch samples;
//   * device, per forward:    ligand radius graph + bonds, ligand-residue edges (cutoff 3*0+20 A) and their flip, receptor edges
//                             (the score model's graph / edge-feature kernels with this model's weights), ligand-atom edges
//                             within 5 A and their flip (conf_la_kernel), then num_conv_layers launches of the fused conv
//                             kernel in its l<=2 mode over the nine edge groups
//                                 [ll | lr | la | aa | al | ar | rr | rl | ra]  =  conv_layers.{9l + 0..8}
//                             with three accumulator slots per node (one per conv feeding that node type), conf_finalize
//                             (mean, BatchNorm of each conv, sum, residual) and the pooled head.
// Node numbering: [ligand b*n_lig+i | atom Bm*n_lig + b*n_atom + a | residue Bm*n_lig + (Bm+1)*n_atom + b*n_rec + r] with
// Bm = max_batch, so the replicated static edge sets are valid for every B <= Bm.  Atom / residue sample b = Bm is the VIRTUAL
// ligand-free sample: the pose-independent work of the first two layers is evaluated on it once per forward and shared by every
// real sample (layer 0: all static groups; layer 1: the receivers none of whose messages changed; see ddk_confidence_forward).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "model.h"

namespace ddk {

static const int LIG_DIMS_C[16] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};   // process_mols.py:62-79
static const int ATOM_DIMS_C[4] = {38, 119, 23, 38};                                        // process_mols.py:81-86
static const int REC_DIM_C = 38;
constexpr int CONF_MAX_OUT = 8;

struct HostMlp {   // Linear(in, NS) -> ReLU -> Linear(NS, NS) with the sigma-embedding columns folded into the first bias
  std::vector<float> w1d, w1b, b1s, w2, b2;   // [NS][DE], [NS][4] or empty, [NS], [NS][NS], [NS]
  std::vector<float> offset; float coeff = 0.f;
};

struct ConfModel {
  EdgeMlpDev lig_edge, rec_edge, lr_edge, la_edge;   // device copies for the dynamic edge sets
  StepParams sp;                                     // t = 0 constants in the layout the shared edge-feature kernel expects
  float la_sigb[NS];
  HostMlp h_rec, h_atom, h_ar;                       // host copies for the per-complex static precompute
  // OldAtomEncoder pieces (host)
  std::vector<float> lig_tables, atom_tables, rec_table;
  std::vector<int> lig_off, atom_off;
  std::vector<float> lig_const, atom_const;          // linear(sigma_emb(0)) + bias  [NS]
  std::vector<float> rec_lin_w, rec_lin_b;           // linear over ESM[:32]
  std::vector<float> rec_lm_w, rec_lm_b;             // lm_embedding_layer [NS][lm + NS]
  float emb0[SIG];
  ConfPredictorDev pred;      // confidence_predictor (device): W0 [NS][2NS], affine0 [NS] x2, W4 [NS][NS], affine4, W8 [n_out][NS], b8
  bool ready = false;
};

struct ConfComplex {
  int n_atom = 0, E_aa = 0;
  float* atom_pos = nullptr;
  float *lig_x0 = nullptr, *atom_x0 = nullptr, *rec_x0 = nullptr;   // [n, NS]
  int64_t cap4 = 0, cap_la = 0, cap_total = 0, off_la = 0, off_al = 0, off_aa = 0, off_ar = 0, off_ra = 0, off_vrr = 0;
  int32_t *e_src = nullptr, *e_dst = nullptr, *e_aux = nullptr;
  float *e_emb = nullptr, *e_sh = nullptr;
  int32_t *st_a = nullptr, *st_b = nullptr;      // one copy of the static sets: aa (atom, atom) then ar (atom, residue) local endpoints
  float *st_emb = nullptr, *st_sh = nullptr;
  int32_t* gtab = nullptr;     // [0..8] gbeg, [9..17] gend, [18] la counter, [19] overflow flag, [32..49] layer-0 table, [64..81] level-A table, [82..85] cursors,
                               // [96..113] level-B table, [114..117] its cursors, [128..145] layer-1 table, [146..149] its cursors
  // backward receptive field of the pooled ligand rows: the second-to-last layer evaluates the static groups (aa, ar, rr, ra) only into the
  // atoms / residues that SEND to a ligand atom in the last layer (level A); their edge records are compacted into a scratch region per forward
  int64_t off_scr = 0, cap_scr = 0;
  uint8_t *flag_a = nullptr, *flag_r = nullptr;      // [2][Bm * n_atom], [2][Bm * n_rec]: level A, level B (= A + the senders of the edges into A)
  // layer-1 sharing: need[y][node of sample b] = 1 when the receiver's messages of static group y (aa, ar | rr, ra) differ from the ligand-free
  // receptor's in sample b (the receiver or one of its senders received a ligand message in layer 0)
  uint8_t* need[4] = {nullptr, nullptr, nullptr, nullptr};      // [Bm * n_atom] x 2, [Bm * n_rec] x 2
  int32_t* deg_static = nullptr;                     // [(n_atom + n_rec)][3]: in-degrees of the static groups (slot 1, the ligand's, is 0)
  int32_t* deg_scratch = nullptr;
  float *xa = nullptr, *xb = nullptr, *sum3 = nullptr;
  int32_t* deg3 = nullptr;
  int64_t n_nodes = 0;         // Bm * n_lig + (Bm + 1) * (n_atom + n_rec)
};

static const HostTensor* getw(ddk_ctx* ctx, const std::string& name, std::initializer_list<int64_t> shape) {
  auto it = ctx->weights.find(name);
  if (it == ctx->weights.end()) { ctx->err = "missing state_dict key: " + name; return nullptr; }
  if (it->second.shape != std::vector<int64_t>(shape)) { ctx->err = "shape mismatch for " + name; return nullptr; }
  return &it->second;
}
static std::vector<float> colsc(const HostTensor* t, int c0, int c1) {
  const int rows = (int)t->shape[0], nc = (int)t->shape[1];
  std::vector<float> o((size_t)rows * (c1 - c0));
  for (int r = 0; r < rows; ++r)
    for (int c = c0; c < c1; ++c) o[(size_t)r * (c1 - c0) + (c - c0)] = t->data[(size_t)r * nc + c];
  return o;
}

// [bond(n_bond) | sigma(32) | dist(32)] -> NS first layer; sigma part folded with emb0 into the bias
static bool load_mlp(ddk_ctx* ctx, const char* name, int n_bond, const char* expansion, float stop, const float* emb0, HostMlp& h,
                     EdgeMlpDev* dev) {
  const HostTensor* w0 = getw(ctx, std::string(name) + ".0.weight", {NS, n_bond + SIG + DE});
  const HostTensor* b0 = getw(ctx, std::string(name) + ".0.bias", {NS});
  const HostTensor* w3 = getw(ctx, std::string(name) + ".3.weight", {NS, NS});
  const HostTensor* b3 = getw(ctx, std::string(name) + ".3.bias", {NS});
  if (!w0 || !b0 || !w3 || !b3) return false;
  h.w1b = n_bond ? colsc(w0, 0, n_bond) : std::vector<float>();
  h.w1d = colsc(w0, n_bond + SIG, n_bond + SIG + DE);
  const std::vector<float> ws = colsc(w0, n_bond, n_bond + SIG);
  h.b1s.assign(NS, 0.f);
  for (int o = 0; o < NS; ++o) {
    float a = b0->data[o];
    for (int k = 0; k < SIG; ++k) a += ws[(size_t)o * SIG + k] * emb0[k];
    h.b1s[o] = a;
  }
  h.w2 = w3->data; h.b2 = b3->data;
  h.offset.resize(DE);
  auto it = ctx->weights.find(std::string(expansion) + "_distance_expansion.offset");
  if (it != ctx->weights.end() && it->second.data.size() == (size_t)DE) h.offset = it->second.data;
  else for (int k = 0; k < DE; ++k) h.offset[k] = stop * (float)k / (float)(DE - 1);
  const double d = (double)(h.offset[1] - h.offset[0]);
  h.coeff = (float)(-0.5 / (d * d));
  if (dev) {
    dev->w1d = dev_upload(ctx, h.w1d);
    {
      std::vector<float> t1((size_t)NS * DE), t2((size_t)NS * NS);
      for (int o = 0; o < NS; ++o) {
        for (int k = 0; k < DE; ++k) t1[(size_t)k * NS + o] = h.w1d[(size_t)o * DE + k];
        for (int k = 0; k < NS; ++k) t2[(size_t)k * NS + o] = w3->data[(size_t)o * NS + k];
      }
      dev->w1d_t = dev_upload(ctx, t1); dev->w2_t = dev_upload(ctx, t2);
      if (!dev->w1d_t || !dev->w2_t) return false;
    }
    dev->w1b = n_bond ? dev_upload(ctx, h.w1b) : nullptr;
    dev->w2 = dev_upload(ctx, h.w2);
    dev->b2 = dev_upload(ctx, h.b2);
    dev->w1l = nullptr; dev->unc = nullptr;
    dev->coeff = h.coeff; dev->step = h.offset[1] - h.offset[0];
    dev->offset = dev_upload(ctx, h.offset);
    if (!dev->w1d || !dev->w2 || !dev->b2 || !dev->offset) return false;
  }
  return true;
}

// host evaluation of an edge MLP on one edge vector (static edge sets): emb[NS], sh[4]
static void host_edge(const HostMlp& m, float vx, float vy, float vz, float* emb, float* sh) {
  const float d = sqrtf(vx * vx + vy * vy + vz * vz);
  const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
  sh[0] = 1.0f; sh[1] = vx * inv; sh[2] = vy * inv; sh[3] = vz * inv;
  float gs[DE], h[NS];
  for (int k = 0; k < DE; ++k) { const float t = d - m.offset[k]; gs[k] = expf(m.coeff * (t * t)); }
  for (int o = 0; o < NS; ++o) {
    float a = m.b1s[o];
    for (int k = 0; k < DE; ++k) a += m.w1d[(size_t)o * DE + k] * gs[k];
    h[o] = fmaxf(a, 0.0f);
  }
  for (int o = 0; o < NS; ++o) {
    float a = m.b2[o];
    for (int k = 0; k < NS; ++k) a += m.w2[(size_t)o * NS + k] * h[k];
    emb[o] = a;
  }
}

int conf_model_finalize(ddk_ctx* ctx) {
  conf_model_destroy(ctx);
  if (ctx->host_only || ctx->weights.find("lig_node_embedding.linear.weight") == ctx->weights.end()) return DDK_OK;
  const ddk_config& c = ctx->cfg;
  if (c.num_confidence_outputs < 1 || c.num_confidence_outputs > CONF_MAX_OUT) return fail(ctx, DDK_ERR_INVALID, "num_confidence_outputs out of range");
  if (c.num_conv_layers < 3) return fail(ctx, DDK_ERR_INVALID, "confidence model: num_conv_layers >= 3 is implemented (2*ns pooled scalars)");
  ConfModel* M = new ConfModel();
  ctx->conf_model = M;
  const int lm = c.lm_embedding_dim;
#define GET(var, name, ...) const HostTensor* var = getw(ctx, name, {__VA_ARGS__}); if (!var) return DDK_ERR_INVALID
  // sinusoidal_embedding(embedding_scale * 0, 32) = [sin 0 ... | cos 0 ...]
  for (int k = 0; k < SIG / 2; ++k) { M->emb0[k] = 0.0f; M->emb0[SIG / 2 + k] = 1.0f; }
  // ---- OldAtomEncoder (models/layers.py:81-116) -------------------------------------------------
  int off = 0;
  for (int i = 0; i < 16; ++i) {
    GET(t, "lig_node_embedding.atom_embedding_list." + std::to_string(i) + ".weight", LIG_DIMS_C[i], NS);
    M->lig_off.push_back(off); M->lig_tables.insert(M->lig_tables.end(), t->data.begin(), t->data.end()); off += LIG_DIMS_C[i];
  }
  off = 0;
  for (int i = 0; i < 4; ++i) {
    GET(t, "atom_node_embedding.atom_embedding_list." + std::to_string(i) + ".weight", ATOM_DIMS_C[i], NS);
    M->atom_off.push_back(off); M->atom_tables.insert(M->atom_tables.end(), t->data.begin(), t->data.end()); off += ATOM_DIMS_C[i];
  }
  auto lin_const = [&](const char* pre, std::vector<float>& out) -> bool {   // linear(sigma_emb(0)) + bias
    const HostTensor* w = getw(ctx, std::string(pre) + ".linear.weight", {NS, SIG});
    const HostTensor* b = getw(ctx, std::string(pre) + ".linear.bias", {NS});
    if (!w || !b) return false;
    out.assign(NS, 0.f);
    for (int o = 0; o < NS; ++o) {
      float a = b->data[o];
      for (int k = 0; k < SIG; ++k) a += w->data[(size_t)o * SIG + k] * M->emb0[k];
      out[o] = a;
    }
    return true;
  };
  if (!lin_const("lig_node_embedding", M->lig_const) || !lin_const("atom_node_embedding", M->atom_const)) return DDK_ERR_INVALID;
  {
    GET(rt, "rec_node_embedding.atom_embedding_list.0.weight", REC_DIM_C, NS);
    GET(rw, "rec_node_embedding.linear.weight", NS, SIG);
    GET(rb, "rec_node_embedding.linear.bias", NS);
    M->rec_table = rt->data; M->rec_lin_w = rw->data; M->rec_lin_b = rb->data;
    if (lm > 0) {
      if (lm < SIG) return fail(ctx, DDK_ERR_INVALID, "lm_embedding_dim < sigma_embed_dim");
      GET(lw, "rec_node_embedding.lm_embedding_layer.weight", NS, lm + NS);
      GET(lb, "rec_node_embedding.lm_embedding_layer.bias", NS);
      M->rec_lm_w = lw->data; M->rec_lm_b = lb->data;
    }
  }
  // ---- edge embedding MLPs ----------------------------------------------------------------------
  HostMlp tmp;
  if (!load_mlp(ctx, "lig_edge_embedding", 4, "lig", c.lig_max_radius, M->emb0, tmp, &M->lig_edge)) return DDK_ERR_INVALID;
  memcpy(M->sp.lig_edge_sigb, tmp.b1s.data(), NS * sizeof(float));
  if (!load_mlp(ctx, "rec_edge_embedding", 0, "rec", c.rec_max_radius, M->emb0, M->h_rec, &M->rec_edge)) return DDK_ERR_INVALID;
  memcpy(M->sp.rec_edge_sigb, M->h_rec.b1s.data(), NS * sizeof(float));
  if (!load_mlp(ctx, "lr_edge_embedding", 0, "cross", c.cross_max_distance, M->emb0, tmp, &M->lr_edge)) return DDK_ERR_INVALID;
  memcpy(M->sp.cross_edge_sigb, tmp.b1s.data(), NS * sizeof(float));
  if (!load_mlp(ctx, "la_edge_embedding", 0, "cross", c.cross_max_distance, M->emb0, tmp, &M->la_edge)) return DDK_ERR_INVALID;
  memcpy(M->la_sigb, tmp.b1s.data(), NS * sizeof(float));
  if (!load_mlp(ctx, "atom_edge_embedding", 0, "lig", c.lig_max_radius, M->emb0, M->h_atom, nullptr)) return DDK_ERR_INVALID;   // :382 lig expansion
  if (!load_mlp(ctx, "ar_edge_embedding", 0, "rec", c.rec_max_radius, M->emb0, M->h_ar, nullptr)) return DDK_ERR_INVALID;
  M->sp.tr_sigma = 0.0f; M->sp.rot_sigma = 0.0f; M->sp.tor_sigma = 0.0f;
  M->sp.cross_cutoff = c.dynamic_max_cross ? 20.0f : c.cross_max_distance;       // 3 * complex_t['tr'] + 20 with complex_t = 0
  { int rcp = conf_predictor_load(ctx, M->pred); if (rcp) return rcp; }
#undef GET
  for (int l = 0; l < c.num_conv_layers; ++l)
    if (!ctx->conv[l].has_weights) return fail(ctx, DDK_ERR_INVALID, "confidence checkpoint lacks conv_layers." + std::to_string(9 * l));
  M->ready = true;
  return DDK_OK;
}

// ---- confidence_predictor: Linear, BN1d, ReLU, Dropout, Linear, BN1d, ReLU, Dropout, Linear (:143-153; score_model.py:110-121) ----
int conf_predictor_load(ddk_ctx* ctx, ConfPredictorDev& P) {
  const ddk_config& c = ctx->cfg;
  if (c.num_confidence_outputs < 1 || c.num_confidence_outputs > CONF_MAX_OUT) return fail(ctx, DDK_ERR_INVALID, "num_confidence_outputs out of range");
#define GET(var, name, ...) const HostTensor* var = getw(ctx, name, {__VA_ARGS__}); if (!var) return DDK_ERR_INVALID
  P.n_out = c.num_confidence_outputs;
  GET(w0, "confidence_predictor.0.weight", NS, 2 * NS);
  GET(b0, "confidence_predictor.0.bias", NS);
  GET(w4, "confidence_predictor.4.weight", NS, NS);
  GET(b4, "confidence_predictor.4.bias", NS);
  GET(w8, "confidence_predictor.8.weight", P.n_out, NS);
  GET(b8, "confidence_predictor.8.bias", P.n_out);
#undef GET
  auto affine = [&](int idx, const HostTensor* lin_b, std::vector<float>& s, std::vector<float>& t) -> bool {
    s.assign(NS, 1.f); t = lin_b->data;     // y = s * (W x) + t
    if (c.confidence_no_batchnorm) return true;
    const std::string p = "confidence_predictor." + std::to_string(idx);
    const HostTensor *g = getw(ctx, p + ".weight", {NS}), *be = getw(ctx, p + ".bias", {NS}), *mu = getw(ctx, p + ".running_mean", {NS}),
                     *var = getw(ctx, p + ".running_var", {NS});
    if (!g || !be || !mu || !var) return false;
    for (int o = 0; o < NS; ++o) {
      s[o] = g->data[o] / sqrtf(var->data[o] + 1e-5f);
      t[o] = (lin_b->data[o] - mu->data[o]) * s[o] + be->data[o];
    }
    return true;
  };
  std::vector<float> s0, t0, s4, t4;
  if (!affine(1, b0, s0, t0) || !affine(5, b4, s4, t4)) return DDK_ERR_INVALID;
  P.w0 = dev_upload(ctx, w0->data); P.s0 = dev_upload(ctx, s0); P.t0 = dev_upload(ctx, t0);
  P.w4 = dev_upload(ctx, w4->data); P.s4 = dev_upload(ctx, s4); P.t4 = dev_upload(ctx, t4);
  P.w8 = dev_upload(ctx, w8->data); P.b8 = dev_upload(ctx, b8->data);
  if (!P.w0 || !P.s0 || !P.t0 || !P.w4 || !P.s4 || !P.t4 || !P.w8 || !P.b8) return fail(ctx, DDK_ERR_NOMEM, "alloc");
  return DDK_OK;
}

void conf_model_destroy(ddk_ctx* ctx) {
  if (ctx->conf_model) { delete (ConfModel*)ctx->conf_model; ctx->conf_model = nullptr; }
}

// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------
struct LaArgs {
  const float* lig_pos;    // [B, n_lig, 3]
  const float* atom_pos;   // [n_atom, 3]
  int B, n_lig, n_atom, atom_node_base;
  float r2;
  EdgeMlpDev mlp;
  float sigb[NS];
  int32_t* gtab;           // [18] counter, [19] overflow
  int64_t off_la, off_al, cap;
  int32_t *e_src, *e_dst;
  float *e_emb, *e_sh;
};

constexpr int COMPACT_SLICES = 32;      // z blocks of conf_level_compact_kernel
// max over the points a of the number of points within `r` of a (a itself included): uniform cell grid, cells of edge >= r searched by coordinate range.
// Returns -1 for non-finite coordinates (the float -> int casts below are undefined for NaN / inf, and a far-flung atom must not buy a
// 256^3-cell grid: the cell edge doubles until the grid has at most max(4096, 8 n) cells - wider cells only enlarge the superset that is searched).
static int max_neighbours_within(const float* pos, int n, float r) {
  if (n <= 0) return 0;
  for (int i = 0; i < 3 * n; ++i)
    if (!std::isfinite(pos[i])) return -1;
  float lo[3] = {pos[0], pos[1], pos[2]}, hi[3] = {pos[0], pos[1], pos[2]};
  for (int i = 1; i < n; ++i)
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], pos[3 * i + k]); hi[k] = std::max(hi[k], pos[3 * i + k]); }
  int dim[3];
  float cell = r;
  const size_t max_cells = std::max<size_t>(4096, 8 * (size_t)n);
  for (;;) {
    size_t total = 1;
    for (int k = 0; k < 3; ++k) { dim[k] = (int)std::min(255.0f, (hi[k] - lo[k]) / cell) + 1; total *= (size_t)dim[k]; }      // (clamped in float: the cast is always defined)
    if (total <= max_cells) break;
    cell *= 2.0f;
  }
  auto cell_at = [&](float x, int k) { return std::max(0, std::min(dim[k] - 1, (int)std::min(255.0f, std::max(0.0f, (x - lo[k]) / cell)))); };      // (clamped grids only merge cells: still a superset search)
  std::vector<int> start((size_t)dim[0] * dim[1] * dim[2] + 1, 0), item(n);
  auto cid = [&](int cx, int cy, int cz) { return ((size_t)cx * dim[1] + cy) * dim[2] + cz; };
  for (int i = 0; i < n; ++i) ++start[cid(cell_at(pos[3 * i], 0), cell_at(pos[3 * i + 1], 1), cell_at(pos[3 * i + 2], 2)) + 1];
  for (size_t q = 1; q < start.size(); ++q) start[q] += start[q - 1];
  std::vector<int> fill(start.begin(), start.end() - 1);
  for (int i = 0; i < n; ++i) item[fill[cid(cell_at(pos[3 * i], 0), cell_at(pos[3 * i + 1], 1), cell_at(pos[3 * i + 2], 2))]++] = i;
  const float r2 = r * r;
  int best = 0;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    int c0[3], c1[3];
    for (int k = 0; k < 3; ++k) { c0[k] = cell_at(pos[3 * i + k] - r, k); c1[k] = cell_at(pos[3 * i + k] + r, k); }
    for (int cx = c0[0]; cx <= c1[0]; ++cx)
      for (int cy = c0[1]; cy <= c1[1]; ++cy)
        for (int cz = c0[2]; cz <= c1[2]; ++cz)
          for (int q = start[cid(cx, cy, cz)]; q < start[cid(cx, cy, cz) + 1]; ++q) {
            const int j = item[q];
            const float dx = pos[3 * j] - pos[3 * i], dy = pos[3 * j + 1] - pos[3 * i + 1], dz = pos[3 * j + 2] - pos[3 * i + 2];
            cnt += dx * dx + dy * dy + dz * dz < r2;
          }
    best = std::max(best, cnt);
  }
  return best;
}

constexpr int LA_LIST = 256;      // per-wave list of the receptor atoms found around one ligand atom (flushed when fewer than 64 slots are left)

// ligand-atom edges: radius(atom.pos, lig.pos, lig_max_radius) (all_atom_score_model.py:413-420) -> group la (src ligand atom,
// dst receptor atom) and its flip al (src receptor atom, dst ligand atom) with the SAME edge embedding and spherical harmonics
// (vector atom - ligand for both, :232-238).  One wave per ligand atom, two phases: the wave first collects the atoms in range into an
// LDS list (64 distance tests per trip), then evaluates the edge MLP with one edge per lane - all lanes busy - and appends ONE
// contiguous run per ligand atom to the group (round 3 ran the 1.5 k-FMA MLP inside the search loop with the two or three lanes of
// a 64-atom chunk that were in range: 276 us for 37 k edges).
__global__ __launch_bounds__(256) void conf_la_kernel(LaArgs A) {
  __shared__ int list[4][LA_LIST];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int* mine = list[wave];
  // grid (B, LA_SLICES): the ligand atoms of a sample are dealt out over LA_SLICES workgroups (edge order inside the group is free)
  for (int i = wave + 4 * blockIdx.y; i < A.n_lig; i += 4 * gridDim.y) {
    const float* lp = A.lig_pos + ((size_t)b * A.n_lig + i) * 3;
    const float lx = lp[0], ly = lp[1], lz = lp[2];
    const int ln = b * A.n_lig + i;
    int n_list = 0;
    for (int j0 = 0; j0 < A.n_atom; j0 += 64) {
      {
        const int j = j0 + lane;
        bool in = false;
        if (j < A.n_atom) {
          const float vx = A.atom_pos[3 * j] - lx, vy = A.atom_pos[3 * j + 1] - ly, vz = A.atom_pos[3 * j + 2] - lz;
          in = vx * vx + vy * vy + vz * vz < A.r2;
        }
        const unsigned long long mask = __ballot(in);
        if (in) mine[n_list + __popcll(mask & ((1ull << lane) - 1ull))] = j;
        n_list += __popcll(mask);
      }
      const bool last = j0 + 64 >= A.n_atom;
      if (n_list == 0 || (!last && n_list <= LA_LIST - 64)) continue;      // (wave-uniform)
      // ---- flush: one edge per lane ----
      int base = 0;
      if (lane == 0) base = atomicAdd(A.gtab + 18, n_list);
      base = __shfl(base, 0, 64);
      for (int q0 = 0; q0 < n_list; q0 += 64) {
        const int q = q0 + lane;
        if (q >= n_list) continue;
        const int64_t p = (int64_t)base + q;
        if (p >= A.cap) { A.gtab[19] = 1; continue; }
        const int j = mine[q];
        const float vx = A.atom_pos[3 * j] - lx, vy = A.atom_pos[3 * j + 1] - ly, vz = A.atom_pos[3 * j + 2] - lz;
        const float d = sqrtf(vx * vx + vy * vy + vz * vz);
        const float inv = 1.7320508075688772f / fmaxf(d, 1e-12f);
        const float4 shv = make_float4(1.0f, vx * inv, vy * inv, vz * inv);
        float gs[DE], h[NS];
#pragma unroll
        for (int k = 0; k < DE; ++k) { const float t = d - A.mlp.offset[k]; gs[k] = expf(A.mlp.coeff * (t * t)); }
        // input-major weights (EdgeMlpDev::w1d_t / w2_t): one uniform load of an input's 24 weights feeds 24 independent accumulators; the
        // summation order per output is the row-major form's
#pragma unroll
        for (int o = 0; o < NS; ++o) h[o] = A.sigb[o];
#pragma unroll
        for (int k = 0; k < DE; ++k) {
#pragma unroll
          for (int o = 0; o < NS; ++o) h[o] = fmaf(A.mlp.w1d_t[k * NS + o], gs[k], h[o]);
        }
#pragma unroll
        for (int o = 0; o < NS; ++o) h[o] = fmaxf(h[o], 0.0f);
        const int an = A.atom_node_base + b * A.n_atom + j;
        const int64_t e1 = A.off_la + p, e2 = A.off_al + p;
        A.e_src[e1] = ln; A.e_dst[e1] = an;
        A.e_src[e2] = an; A.e_dst[e2] = ln;
        *reinterpret_cast<float4*>(A.e_sh + 4 * e1) = shv;
        *reinterpret_cast<float4*>(A.e_sh + 4 * e2) = shv;
        float y[NS];
#pragma unroll
        for (int o = 0; o < NS; ++o) y[o] = A.mlp.b2[o];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
          for (int o = 0; o < NS; ++o) y[o] = fmaf(A.mlp.w2_t[k * NS + o], h[k], y[o]);
        }
#pragma unroll
        for (int o4 = 0; o4 < NS / 4; ++o4) {
          const float4 rv = make_float4(y[4 * o4], y[4 * o4 + 1], y[4 * o4 + 2], y[4 * o4 + 3]);
          *reinterpret_cast<float4*>(A.e_emb + e1 * NS + 4 * o4) = rv;
          *reinterpret_cast<float4*>(A.e_emb + e2 * NS + 4 * o4) = rv;
        }
      }
      n_list = 0;
    }
  }
}

// group table of one forward: [0..8] gbeg, [9..17] gend for [ll lr la aa al ar rr rl ra] from the shared graph kernel's info
// table (go[0..4] of its [ll | lr | rr | rl] list), the la counter and the static set sizes
__global__ void conf_gtab_kernel(int32_t* gtab, const int32_t* info, int B, int Bm, int E_aa, int E_rr, int n_atom, int off_la, int off_al, int off_aa,
                                 int off_ar, int off_ra, int off_vrr, int cap_la) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int go0 = info[I_GO], go1 = info[I_GO + 1], go2 = info[I_GO + 2], go3 = info[I_GO + 3], go4 = info[I_GO + 4];
  const int n_la = min(gtab[18], cap_la);
  const int beg[9] = {go0, go1, off_la, off_aa, off_al, off_ar, go2, go3, off_ra};
  const int end[9] = {go1, go2, off_la + n_la, off_aa + B * E_aa, off_al + n_la, off_ar + B * n_atom, go3, go4, off_ra + B * n_atom};
  for (int g = 0; g < 9; ++g) { gtab[g] = beg[g]; gtab[9 + g] = end[g]; }
  // layer 0: before the first conv the atom and residue rows (and the static edge sets' features) are the same in every sample, so the
  // pose-independent groups aa, ar, rr, ra are evaluated ONCE, on the virtual ligand-free sample Bm (their edges are stored sample-major;
  // its rec-rec records are conf_vrr_kernel's copy of sample 0's), and conf_finalize_kernel reads its accumulators for every sample
  int32_t* t0 = gtab + 32;
  for (int g = 0; g < 9; ++g) { t0[g] = beg[g]; t0[9 + g] = end[g]; }
  t0[3] = off_aa + Bm * E_aa; t0[9 + 3] = t0[3] + E_aa;
  t0[5] = off_ar + Bm * n_atom; t0[9 + 5] = t0[5] + n_atom;
  t0[6] = off_vrr; t0[9 + 6] = off_vrr + E_rr;
  t0[8] = off_ra + Bm * n_atom; t0[9 + 8] = t0[8] + n_atom;
}

// rec-rec records of the virtual sample: sample 0's (the first E_rr of the group: the graph kernel stores them sample-major, and their features do
// not depend on the sample) with the node ids moved to sample Bm
__global__ void conf_vrr_kernel(const int32_t* gtab, int E_rr, int id_shift, int64_t off_vrr, int32_t* e_src, int32_t* e_dst, float* e_emb, float* e_sh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = i / 8, q = i % 8;        // 8 threads per edge: six embedding quads, the SH quad, the ids
  if (k >= E_rr) return;
  const int64_t e = (int64_t)gtab[6] + k, p = off_vrr + k;
  if (q < 6) reinterpret_cast<float4*>(e_emb + (size_t)p * NS)[q] = reinterpret_cast<const float4*>(e_emb + (size_t)e * NS)[q];
  else if (q == 6) *reinterpret_cast<float4*>(e_sh + (size_t)p * 4) = *reinterpret_cast<const float4*>(e_sh + (size_t)e * 4);
  else { e_src[p] = e_src[e] + id_shift; e_dst[p] = e_dst[e] + id_shift; }
}

// ---- level A of the backward receptive field (second-to-last layer) ---------------------------------------------------------
// The last layer updates ligand rows only (groups ll, lr, la), so the second-to-last layer has to PRODUCE only the ligand rows and the
// rows of the atoms / residues that send in la / lr: exactly the receivers of the al / rl edges.  Of the static groups it therefore
// evaluates only the edges received by those nodes.  Exact: a dropped message never reaches the pooled ligand rows.
struct ConfLevelArgs {
  const int32_t* gtab;
  int32_t *e_src, *e_dst;
  float *e_emb, *e_sh;
  uint8_t *flag_a, *flag_r;      // level-A flags (conf_level_flags_kernel)
  const uint8_t* fl[4];          // compaction: the receiver flags of the four static groups aa, ar | rr, ra
  int with_virtual;              // compaction: sample index B = the virtual sample Bm, every edge kept (layer-1 table)
  int B, n_atom, n_rec, E_aa, E_rr;
  int64_t atom_base, rec_base, off_aa, off_ar, off_ra, off_vrr, off_scr, Bm;
  int32_t* cursors;      // [4] (zeroed by the caller)
  int32_t* tabA;         // [18]
};

__global__ void conf_level_flags_kernel(ConfLevelArgs A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int al0 = A.gtab[4], al1 = A.gtab[9 + 4], rl0 = A.gtab[7], rl1 = A.gtab[9 + 7];
  if (i < al1 - al0) A.flag_a[A.e_src[al0 + i] - A.atom_base] = 1;
  else if (i - (al1 - al0) < rl1 - rl0) A.flag_r[A.e_src[rl0 + (i - (al1 - al0))] - A.rec_base] = 1;
}

// level B = level A + every atom / residue that sends along an edge the level-A layer evaluates (its row is read there): the senders of the
// compacted level-A copies of aa, ra (atoms) and ar, rr (residues); flags B start as a copy of flags A (the caller copies)
struct ConfLevelBArgs {
  const int32_t* tabA;
  const int32_t* e_dst;
  uint8_t *flag_a, *flag_r;      // level-B flags
  int64_t atom_base, rec_base;
};
__global__ void conf_level_b_flags_kernel(ConfLevelBArgs A, int64_t cap_scr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap_scr) return;
  const int gid[4] = {3, 5, 6, 8};
  int64_t k = i;
  for (int y = 0; y < 4; ++y) {
    const int64_t n = A.tabA[9 + gid[y]] - A.tabA[gid[y]];
    if (k < n) {
      const int dn = A.e_dst[A.tabA[gid[y]] + k];
      if (y == 0 || y == 3) A.flag_a[dn - A.atom_base] = 1;      // aa, ra: the sender is an atom
      else A.flag_r[dn - A.rec_base] = 1;                         // ar, rr: the sender is a residue
      return;
    }
    k -= n;
  }
}

// Layer-1 sharing.  After layer 0 the row of an atom / residue differs from the ligand-free receptor's (the virtual sample's) only if it received
// a ligand message: the receivers of the al / rl edges = the level-A flags.  A receiver's layer-1 messages of static group y are the virtual
// sample's unless the receiver or one of its senders in y is such a node: need[y][receiver] = flagA[receiver] (the caller copies) | OR over its
// edges of flagA[sender].  One thread per static edge of the real samples.
struct ConfNeedArgs {
  const int32_t* gtab;
  const int32_t *e_src, *e_dst;
  const uint8_t *flag_a, *flag_r;
  uint8_t* need[4];
  int B, n_atom, n_rec, E_aa, E_rr;
  int64_t atom_base, rec_base, off_aa, off_ar, off_ra;
};
__global__ void conf_need_flags_kernel(ConfNeedArgs A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_aa = (int64_t)A.B * A.E_aa, n_at = (int64_t)A.B * A.n_atom, n_rr = (int64_t)A.B * A.E_rr;
  int y;
  int64_t e;
  if (i < n_aa) { y = 0; e = A.off_aa + i; }
  else if (i < n_aa + n_at) { y = 1; e = A.off_ar + (i - n_aa); }
  else if (i < n_aa + n_at + n_rr) { y = 2; e = (int64_t)A.gtab[6] + (i - n_aa - n_at); }
  else if (i < n_aa + 2 * n_at + n_rr) { y = 3; e = A.off_ra + (i - n_aa - n_at - n_rr); }
  else return;
  const int sn = A.e_src[e], dn = A.e_dst[e];
  const bool sender_atom = y == 0 || y == 3;       // aa, ra: the sender (edge_dst) is an atom
  const bool hit = sender_atom ? A.flag_a[dn - A.atom_base] != 0 : A.flag_r[dn - A.rec_base] != 0;
  if (hit) A.need[y][sn - (y < 2 ? A.atom_base : A.rec_base)] = 1;
}

// grid (samples, 4, slices): sample b, static group y in {aa, ar, rr, ra}: stable compaction of the edges whose receiver is flagged into the
// scratch region (a slice claims one contiguous output range: receivers stay grouped inside a slice).  Two coalesced passes over the slice:
// count, then per 256-edge chunk a ballot scan for the positions and a cooperative copy of the kept records (seven 16-B words per edge).
__global__ __launch_bounds__(256) void conf_level_compact_kernel(ConfLevelArgs A) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  __shared__ int keep_e[256], keep_p[256];
  const int b = blockIdx.x, y = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool virt = A.with_virtual && b == A.B;
  const int64_t bs = virt ? A.Bm : b;              // segment index of the static sets
  const int n = y == 0 ? A.E_aa : (y == 2 ? A.E_rr : A.n_atom);
  const int64_t first = y == 0 ? A.off_aa + bs * A.E_aa : (y == 1 ? A.off_ar + bs * A.n_atom
                      : (y == 2 ? (virt ? A.off_vrr : (int64_t)A.gtab[6] + (int64_t)b * A.E_rr) : A.off_ra + bs * A.n_atom));
  const uint8_t* fl = A.fl[y];
  const int64_t nb = y < 2 ? A.atom_base : A.rec_base;
  // slices of at least one 256-edge chunk (the kernel is latency bound - a chunk is four dependent memory round trips -, so the large atom-atom set
  // is cut into many short slices; the small sets leave most of their z blocks idle)
  const int n_sl = max((n + (int)gridDim.z - 1) / (int)gridDim.z, 256), s0 = min((int)blockIdx.z * n_sl, n), s1 = min(s0 + n_sl, n);
  if (s0 >= s1) return;
  int cnt = 0;
  if (virt) cnt = (s1 - s0 + 255 - tid) / 256;
  else for (int k = s0 + tid; k < s1; k += 256) cnt += fl[A.e_src[first + k] - nb] ? 1 : 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int64_t region = A.off_scr + (y == 0 ? 0 : (y == 1 ? (A.Bm + 1) * A.E_aa : (y == 2 ? (A.Bm + 1) * ((int64_t)A.E_aa + A.n_atom)
                                                                                              : (A.Bm + 1) * ((int64_t)A.E_aa + A.n_atom) + (A.Bm + 1) * (int64_t)A.E_rr)));
    base_s = (int)region + atomicAdd(A.cursors + y, total);
  }
  __syncthreads();
  int run = base_s;
  for (int c0 = s0; c0 < s1; c0 += 256) {
    const int k = c0 + tid;
    int sn = 0;
    bool keep = false;
    if (k < s1) { sn = A.e_src[first + k]; keep = virt || fl[sn - nb] != 0; }
    const unsigned long long mask = __ballot(keep);
    __syncthreads();                       // the previous chunk's lists have been consumed
    if (lane == 0) wsum[wave] = __popcll(mask);
    __syncthreads();
    int pre = __popcll(mask & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (keep) {
      const int pos = run + pre;
      A.e_src[pos] = sn; A.e_dst[pos] = A.e_dst[first + k];
      keep_e[pre] = k; keep_p[pre] = pos;
    }
    __syncthreads();
    for (int it = tid; it < tot * 7; it += 256) {
      const int m = it / 7, q = it - 7 * m;
      const int64_t e = first + keep_e[m], pos = keep_p[m];
      if (q < 6) reinterpret_cast<float4*>(A.e_emb + (size_t)pos * NS)[q] = reinterpret_cast<const float4*>(A.e_emb + (size_t)e * NS)[q];
      else *reinterpret_cast<float4*>(A.e_sh + (size_t)pos * 4) = *reinterpret_cast<const float4*>(A.e_sh + (size_t)e * 4);
    }
    run += tot;
  }
}

// the group table of the pruned layer: the full table with the four static groups replaced by their compacted copies
__global__ void conf_level_table_kernel(ConfLevelArgs A) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int g = 0; g < 18; ++g) A.tabA[g] = A.gtab[g];
  const int64_t r[4] = {A.off_scr, A.off_scr + (A.Bm + 1) * A.E_aa, A.off_scr + (A.Bm + 1) * ((int64_t)A.E_aa + A.n_atom),
                        A.off_scr + (A.Bm + 1) * ((int64_t)A.E_aa + A.n_atom) + (A.Bm + 1) * (int64_t)A.E_rr};
  const int gid[4] = {3, 5, 6, 8};
  for (int y = 0; y < 4; ++y) { A.tabA[gid[y]] = (int)r[y]; A.tabA[9 + gid[y]] = (int)r[y] + A.cursors[y]; }
}

// in-degree of every (node, slot), slot = group % 3: the static groups' (aa, ar | rr, ra) are constants of the complex (deg_static: the same in
// every sample, the virtual one included), the ligand-dependent groups ll, lr, la, al, rl are counted per forward
__global__ void conf_deg_init_kernel(const int32_t* deg_static, int64_t atom0, int64_t rec0, int64_t n_nodes, int n_atom, int n_rec, int32_t* deg3) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes * 3) return;
  const int64_t node = i / 3;
  const int s = (int)(i - 3 * node);
  int v = 0;
  if (node >= rec0) v = deg_static[((int64_t)n_atom + (node - rec0) % n_rec) * 3 + s];
  else if (node >= atom0) v = deg_static[((node - atom0) % n_atom) * 3 + s];
  deg3[i] = v;
}
__global__ void conf_deg_kernel(const int32_t* gtab, const int32_t* e_src, int32_t* deg3, int64_t n_dyn) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int gid[5] = {0, 1, 2, 4, 7};
  int key = -1 - lane;                     // (distinct negative keys: lanes outside every group never join a run)
  if (e < n_dyn)
    for (int q = 0; q < 5; ++q) {
      const int g = gid[q];
      if (e >= gtab[g] && e < gtab[9 + g]) { key = e_src[e] * 3 + (g % 3); break; }
    }
  // the groups are sorted by receiver (or made of runs): one atomic per run of equal (receiver, slot) inside the wave instead of one per edge
  const int prev = __shfl_up(key, 1, 64);
  const bool head = lane == 0 || key != prev;
  const unsigned long long heads = __ballot(head);
  if (head && key >= 0) {
    const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int len = later ? __ffsll((long long)later) : 64 - lane;
    atomicAdd(deg3 + key, len);
  }
}

// initial node features: static embedding broadcast over the samples (Bm ligand samples, Bm + 1 atom / residue samples), zero padded to XW
__global__ void conf_node_init_kernel(const float* lig_x0, const float* atom_x0, const float* rec_x0, int64_t a0, int64_t r0, int64_t n, int n_lig, int n_atom,
                                      int n_rec, float* x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * XW) return;
  const int64_t node = i / XW;
  const int c = (int)(i % XW);
  float v = 0.0f;
  if (c < NS) {
    if (node < a0) v = lig_x0[(node % n_lig) * NS + c];
    else if (node < r0) v = atom_x0[((node - a0) % n_atom) * NS + c];
    else v = rec_x0[((node - r0) % n_rec) * NS + c];
  }
  x[i] = v;
}

// x_out = pad(x_in) + sum over the 3 convs feeding the node type of BN_conv(sum / max(deg,1))   (all_atom_score_model.py:37-50,272-279)
// share 1 (layer 0): slots 0 and 2 of the atom / residue rows (groups aa, ar / rr, ra) were accumulated for the virtual sample only: every
// sample reads the virtual sample's row; share 2 (layer 1): ... unless need[y] marks the (receiver, group) as evaluated in its own sample.
// vatom0 / vrec0 = first atom / residue row of the virtual sample; n_atom, n_rec = nodes per sample
struct ConfFinalizeArgs {
  const float* sum3; const int32_t* deg3; const float* x_in; const float *bn_mean, *bn_scale, *bn_bias;
  int64_t n_nodes, n_update, atom0, rec0, vatom0, vrec0;
  int dout, share, n_atom, n_rec;
  const uint8_t* need[4];
  float* x_out;
};
__global__ void conf_finalize_kernel(ConfFinalizeArgs A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n_nodes * XW) return;
  const int64_t node = i / XW;
  const int c = (int)(i % XW);
  float v = A.x_in[i];
  if (node < A.n_update && c < A.dout) {
    const int type = node < A.atom0 ? 0 : (node < A.rec0 ? 1 : 2);
    int64_t node_sh = node;        // the virtual sample's copy of this atom / residue
    if (A.share && type == 1) node_sh = A.vatom0 + (node - A.atom0) % A.n_atom;
    if (A.share && type == 2) node_sh = A.vrec0 + (node - A.rec0) % A.n_rec;
    const bool real = node_sh != node;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int g = 3 * type + s;
      const int d = A.deg3[node * 3 + s];
      int64_t nd = (s != 1) ? node_sh : node;
      if (A.share == 2 && s != 1 && real) {
        const uint8_t* nf = A.need[(type == 1 ? 0 : 2) + (s == 2 ? 1 : 0)];
        if (nf[node - (type == 1 ? A.atom0 : A.rec0)]) nd = node;
      }
      const float m = A.sum3[(nd * 3 + s) * XW + c] / (float)(d > 1 ? d : 1);
      v += (m - A.bn_mean[g * XW + c]) * A.bn_scale[g * XW + c] + A.bn_bias[g * XW + c];
    }
  }
  A.x_out[i] = v;
}

struct HeadCArgs {
  const float* x; int B, n_lig, n_out;
  const float *w0, *s0, *t0, *w4, *s4, *t4, *w8, *b8;
  float* out;
  const int32_t* ovf;      // ligand-atom edge capacity overflow flag of this forward (gtab[19]): the batch's confidences become NaN
};
// scatter_mean of [x[:, :ns] | x[:, -ns:]] over each graph's ligand atoms, then the predictor MLP (:281-284)
__global__ __launch_bounds__(64) void conf_head_kernel(HeadCArgs A) {
  __shared__ float pooled[2 * NS], h1[NS], h2[NS];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < 2 * NS) {
    const int col = t < NS ? t : (XW - NS) + (t - NS);
    float s = 0.0f;
    for (int i = 0; i < A.n_lig; ++i) s += A.x[((size_t)b * A.n_lig + i) * XW + col];
    pooled[t] = s / (float)A.n_lig;
  }
  __syncthreads();
  if (t < NS) {
    float a = 0.0f;
    for (int k = 0; k < 2 * NS; ++k) a += A.w0[t * 2 * NS + k] * pooled[k];
    h1[t] = fmaxf(a * A.s0[t] + A.t0[t], 0.0f);
  }
  __syncthreads();
  if (t < NS) {
    float a = 0.0f;
    for (int k = 0; k < NS; ++k) a += A.w4[t * NS + k] * h1[k];
    h2[t] = fmaxf(a * A.s4[t] + A.t4[t], 0.0f);
  }
  __syncthreads();
  if (t < A.n_out) {
    float a = A.b8[t];
    for (int k = 0; k < NS; ++k) a += A.w8[t * NS + k] * h2[k];
    // an overflowed edge list means a truncated graph: the result must not look like a confidence (sampling() maps NaN to -1000 like the
    // reference's nan_to_num, utils/sampling.py:246)
    A.out[(size_t)b * A.n_out + t] = (A.ovf != nullptr && *A.ovf) ? __builtin_nanf("") : a;
  }
}

hipError_t launch_conf_head(const ConfPredictorDev& P, const float* x, int B, int n_lig, float* out, const int32_t* ovf, hipStream_t s) {
  HeadCArgs H;
  H.x = x; H.B = B; H.n_lig = n_lig; H.n_out = P.n_out;
  H.w0 = P.w0; H.s0 = P.s0; H.t0 = P.t0; H.w4 = P.w4; H.s4 = P.s4; H.t4 = P.t4; H.w8 = P.w8; H.b8 = P.b8; H.out = out; H.ovf = ovf;
  hipLaunchKernelGGL(conf_head_kernel, dim3(B), dim3(64), 0, s, H);
  return hipGetLastError();
}

// per-sample replicas of the static edge sets: aa (src = atom a, dst = atom b), ar (src = atom, dst = its residue) and ra (the flip with
// the same features), node ids offset by the sample
__global__ void conf_static_replicate_kernel(const int32_t* a1, const int32_t* b1, const float* emb1, const float* sh1, int E_aa, int n_atom, int n_rec,
                                             int Bm, int64_t atom_base, int64_t rec_base, int64_t off_aa, int64_t off_ar, int64_t off_ra,
                                             int32_t* e_src, int32_t* e_dst, float* e_emb, float* e_sh) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_aa = (int64_t)Bm * E_aa;
  if (i >= n_aa + (int64_t)Bm * n_atom) return;
  int64_t p, p2 = -1;
  int k, sn, dn;
  if (i < n_aa) {
    const int b = (int)(i / E_aa);
    k = (int)(i - (int64_t)b * E_aa);
    p = off_aa + i;
    sn = (int)(atom_base + (int64_t)b * n_atom + a1[k]); dn = (int)(atom_base + (int64_t)b * n_atom + b1[k]);
  } else {
    const int64_t j = i - n_aa;
    const int b = (int)(j / n_atom), a = (int)(j - (int64_t)b * n_atom);
    k = E_aa + a;
    p = off_ar + j; p2 = off_ra + j;
    sn = (int)(atom_base + (int64_t)b * n_atom + a); dn = (int)(rec_base + (int64_t)b * n_rec + b1[k]);
  }
  e_src[p] = sn; e_dst[p] = dn;
  const float4* es = reinterpret_cast<const float4*>(emb1 + (size_t)k * NS);
  float4* ed = reinterpret_cast<float4*>(e_emb + (size_t)p * NS);
#pragma unroll
  for (int q = 0; q < NS / 4; ++q) ed[q] = es[q];
  const float4 shv = *reinterpret_cast<const float4*>(sh1 + (size_t)k * 4);
  *reinterpret_cast<float4*>(e_sh + (size_t)p * 4) = shv;
  if (p2 >= 0) {
    e_src[p2] = dn; e_dst[p2] = sn;
    float4* ed2 = reinterpret_cast<float4*>(e_emb + (size_t)p2 * NS);
#pragma unroll
    for (int q = 0; q < NS / 4; ++q) ed2[q] = es[q];
    *reinterpret_cast<float4*>(e_sh + (size_t)p2 * 4) = shv;
  }
}

void conf_complex_free(ddk_complex* cx) {
  if (cx && cx->conf) { delete cx->conf; cx->conf = nullptr; }   // device arrays live in cx->allocs
}

template <typename T>
static T* cxu(ddk_complex* cx, const T* src, size_t n) {
  T* p = (T*)cx_alloc(cx, n * sizeof(T));
  if (!p) return nullptr;
  if (n && src && !cx_put(cx, p, src, n * sizeof(T))) return nullptr;
  return p;
}

}  // namespace ddk

using namespace ddk;

extern "C" {

int ddk_complex_set_atoms(ddk_ctx* ctx, ddk_complex* cx, const ddk_atoms_desc* d, const int32_t* lig_x, const float* rec_x, int32_t rec_feat_dim) {
  if (!ctx || !cx || !d || !lig_x || !rec_x) return DDK_ERR_INVALID;
  ConfModel* M = (ConfModel*)ctx->conf_model;
  if (!ctx->cfg.all_atoms || !M || !M->ready) return fail(ctx, DDK_ERR_STATE, "ddk_complex_set_atoms needs a finalised all-atom confidence model context");
  if (cx->conf) return fail(ctx, DDK_ERR_STATE, "atoms already set for this complex");
  const ddk_config& c = ctx->cfg;
  const int n_lig = cx->n_lig, n_rec = cx->n_rec, n_atom = d->n_atom, E_aa = d->n_atom_edges, lm = c.lm_embedding_dim;
  if (n_atom < 1 || n_atom > (1 << 20)) return fail(ctx, DDK_ERR_INVALID, "n_atom out of range");
  if (rec_feat_dim != 1 + lm) return fail(ctx, DDK_ERR_INVALID, "receptor feature width != 1 + lm_embedding_dim");
  hipSetDevice(c.device);
  ConfComplex* K = new ConfComplex();
  cx->conf = K;
  K->n_atom = n_atom; K->E_aa = E_aa;
  const int64_t Bm = cx->max_batch, Bv = Bm + 1;
  {   // staged, asynchronous upload of everything below (model.h: cx_put / cx_stage_flush)
    const size_t n_st = (size_t)E_aa + (size_t)n_atom;
    int rc0 = cx_stage_begin(ctx, cx, (size_t)(n_lig + n_atom + n_rec) * NS * 4 + (size_t)n_atom * 12 + (size_t)cx->E_rr * NS * 4 +
                                          n_st * (8 + NS * 4 + 16) + (size_t)(n_atom + n_rec) * 12 + 16 * 256);
    if (rc0) return rc0;
  }
  // ---- node embeddings (OldAtomEncoder at t = 0) --------------------------------------------------
  std::vector<float> lx((size_t)n_lig * NS), ax((size_t)n_atom * NS), rx((size_t)n_rec * NS);
  for (int i = 0; i < n_lig; ++i)
    for (int o = 0; o < NS; ++o) {
      float a = M->lig_const[o];
      for (int f = 0; f < 16; ++f) {
        const int v = lig_x[(size_t)i * 16 + f];
        if (v < 0 || v >= LIG_DIMS_C[f]) return fail(ctx, DDK_ERR_INVALID, "ligand categorical feature out of range");
        a += M->lig_tables[(size_t)(M->lig_off[f] + v) * NS + o];
      }
      lx[(size_t)i * NS + o] = a;
    }
  for (int i = 0; i < n_atom; ++i)
    for (int o = 0; o < NS; ++o) {
      float a = M->atom_const[o];
      for (int f = 0; f < 4; ++f) {
        const int v = d->atom_x[(size_t)i * 4 + f];
        if (v < 0 || v >= ATOM_DIMS_C[f]) return fail(ctx, DDK_ERR_INVALID, "receptor-atom categorical feature out of range");
        a += M->atom_tables[(size_t)(M->atom_off[f] + v) * NS + o];
      }
      ax[(size_t)i * NS + o] = a;
    }
  for (int j = 0; j < n_rec; ++j) {
    const int res = (int)rec_x[(size_t)j * rec_feat_dim];
    if (res < 0 || res >= REC_DIM_C) return fail(ctx, DDK_ERR_INVALID, "residue id out of range");
  }
  host_parallel_for(n_rec, [&](int j) {
    const float* xr = rec_x + (size_t)j * rec_feat_dim;      // [res id | ESM(lm)]; node_attr = [x | sigma_emb]
    const int res = (int)xr[0];
    float emb[NS];
    for (int o = 0; o < NS; ++o) {
      // OldAtomEncoder quirk (layers.py:112): the "scalar feature" slice x[:, 1:1+32] is ESM[:32] when lm features are present
      double a = M->rec_lin_b[o] + M->rec_table[(size_t)res * NS + o];
      for (int k = 0; k < SIG; ++k) a += (double)M->rec_lin_w[(size_t)o * SIG + k] * (lm > 0 ? xr[1 + k] : M->emb0[k]);
      emb[o] = (float)a;
    }
    if (lm > 0) {   // lm_embedding_layer(cat([emb, x[:, -lm:]])) with x[:, -lm:] = [ESM[32:] | sigma_emb(0)]
      for (int o = 0; o < NS; ++o) {
        const float* w = M->rec_lm_w.data() + (size_t)o * (lm + NS);
        double a = M->rec_lm_b[o];
        for (int k = 0; k < NS; ++k) a += (double)w[k] * emb[k];
        for (int k = 0; k < lm - SIG; ++k) a += (double)w[NS + k] * xr[1 + SIG + k];
        for (int k = 0; k < SIG; ++k) a += (double)w[NS + lm - SIG + k] * M->emb0[k];
        rx[(size_t)j * NS + o] = (float)a;
      }
    } else {
      for (int o = 0; o < NS; ++o) rx[(size_t)j * NS + o] = emb[o];
    }
  });
  K->lig_x0 = cxu(cx, lx.data(), lx.size()); K->atom_x0 = cxu(cx, ax.data(), ax.size()); K->rec_x0 = cxu(cx, rx.data(), rx.size());
  K->atom_pos = cxu(cx, d->atom_pos, (size_t)n_atom * 3);
  // ---- receptor-edge first layer with THIS model's rec_edge_embedding (the shared edge-feature kernel reads cx->rr_pre1) ----
  {
    const std::vector<int32_t>& ei = cx->h_rr;          // host copies kept by ddk_complex_create (no read-back: the upload may be in flight)
    const std::vector<float>& rp = cx->h_rec_pos;
    std::vector<float> pre1((size_t)cx->E_rr * NS);
    host_parallel_for(cx->E_rr, [&](int k) {
      const int a = ei[k], b = ei[cx->E_rr + k];
      const float vx = rp[3 * b] - rp[3 * a], vy = rp[3 * b + 1] - rp[3 * a + 1], vz = rp[3 * b + 2] - rp[3 * a + 2];
      const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
      float gs[DE];
      for (int q = 0; q < DE; ++q) { const float t = dist - M->h_rec.offset[q]; gs[q] = expf(M->h_rec.coeff * (t * t)); }
      for (int o = 0; o < NS; ++o) {
        float a2 = 0.0f;
        for (int q = 0; q < DE; ++q) a2 += M->h_rec.w1d[(size_t)o * DE + q] * gs[q];
        pre1[(size_t)k * NS + o] = a2;
      }
    });
    if (!cx_put(cx, cx->rr_pre1, pre1.data(), pre1.size() * 4)) return fail(ctx, DDK_ERR_HIP, "rr_pre1 upload failed");
    // static sets need rec_pos on the host below
    // ---- edge arrays: [4-group region of the shared graph kernel | la | al | aa | ar | ra] -------------
    K->cap4 = cx->edge_cap;
    // ligand-atom edges (radius(atom.pos, ligand.pos, lig_max_radius, max_num_neighbors = 10000), all_atom_score_model.py:409-410): a capacity that CANNOT
    // overflow, from the receptor's own geometry.  If any atom a lies within r of a point x, every atom within r of x lies within 2r of a: no ligand atom,
    // wherever a pose puts it, collects more than max_a |{b : |b - a| < 2r}| neighbours (round 4 assumed 96 per ligand atom on average and turned the whole
    // batch into NaN / -1000 when a dense pocket exceeded it).  ~250 for protein heavy atoms at r = 5 A; a cell grid makes the count O(n_atom)
    const int la_raw = max_neighbours_within(d->atom_pos, n_atom, 2.0f * c.lig_max_radius);
    if (la_raw < 0) return fail(ctx, DDK_ERR_INVALID, "ddk_complex_set_atoms: non-finite receptor atom coordinates");
    const int la_bound = std::min(n_atom, la_raw);
    K->cap_la = Bm * (int64_t)n_lig * std::max(la_bound, 1) + 64;
    K->off_la = K->cap4; K->off_al = K->off_la + K->cap_la; K->off_aa = K->off_al + K->cap_la;
    // (Bv = Bm + 1 segments of every static set: the last one belongs to the virtual ligand-free sample; vrr = its rec-rec records)
    K->off_ar = K->off_aa + Bv * E_aa; K->off_ra = K->off_ar + Bv * n_atom; K->off_vrr = K->off_ra + Bv * n_atom; K->cap_total = K->off_vrr + cx->E_rr;
    K->off_scr = K->cap_total; K->cap_scr = Bv * ((int64_t)E_aa + 2 * (int64_t)n_atom + cx->E_rr);     // compacted copies of aa | ar | rr | ra (worst case: all)
    const int64_t e_all = K->cap_total + 3 * K->cap_scr;       // (level A, level B = the third-to-last layer, and layer 1's table)
    if (e_all >= ((int64_t)1 << 31)) return fail(ctx, DDK_ERR_INVALID, "edge capacity exceeds int32 (reduce max_batch)");
    K->e_src = cxu<int32_t>(cx, nullptr, e_all); K->e_dst = cxu<int32_t>(cx, nullptr, e_all);
    K->e_aux = cxu<int32_t>(cx, nullptr, K->cap4);
    K->e_emb = cxu<float>(cx, nullptr, e_all * NS); K->e_sh = cxu<float>(cx, nullptr, e_all * 4);
    K->flag_a = cxu<uint8_t>(cx, nullptr, 2 * Bm * n_atom); K->flag_r = cxu<uint8_t>(cx, nullptr, 2 * Bm * n_rec);
    K->need[0] = cxu<uint8_t>(cx, nullptr, 2 * Bm * n_atom); K->need[2] = cxu<uint8_t>(cx, nullptr, 2 * Bm * n_rec);
    if (!K->flag_a || !K->flag_r || !K->need[0] || !K->need[2]) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed (confidence level flags)");
    K->need[1] = K->need[0] + Bm * n_atom; K->need[3] = K->need[2] + Bm * n_rec;
    if (!K->e_src || !K->e_dst || !K->e_aux || !K->e_emb || !K->e_sh) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed (confidence edge arrays)");
    // static sets (atom-atom; atom->residue and its flip)
    // ONE copy of the static sets goes up (E_aa + n_atom edges: local endpoints, embedding, SH); a kernel writes the Bm per-sample
    // replicas with their node-id offsets (the host used to assemble and upload all Bm copies: ~115 MB for 2400 atoms x 40 samples)
    const int n1 = E_aa + n_atom;
    std::vector<int32_t> a1((size_t)n1), b1((size_t)n1);
    std::vector<float> emb1((size_t)n1 * NS), sh1((size_t)n1 * 4);
    for (int k = 0; k < E_aa; ++k) {
      const int a = d->atom_edge_index[k], b = d->atom_edge_index[E_aa + k];
      if (a < 0 || a >= n_atom || b < 0 || b >= n_atom) return fail(ctx, DDK_ERR_INVALID, "atom edge index out of range");
      a1[k] = a; b1[k] = b;
    }
    for (int i = 0; i < n_atom; ++i) {
      if (d->atom_rec_index[i] != i) return fail(ctx, DDK_ERR_INVALID, "atom_rec_index row 0 must be arange(n_atom) (process_mols.py:472)");
      const int r = d->atom_rec_index[n_atom + i];
      if (r < 0 || r >= n_rec) return fail(ctx, DDK_ERR_INVALID, "atom residue index out of range");
      a1[E_aa + i] = i; b1[E_aa + i] = r;
    }
    host_parallel_for(n1, [&](int k) {       // edge embedding + sh of the static sets: 32 exp + 1.3 k MAC per edge on a few host threads
      if (k < E_aa) {
        const int a = a1[k], b = b1[k];
        host_edge(M->h_atom, d->atom_pos[3 * b] - d->atom_pos[3 * a], d->atom_pos[3 * b + 1] - d->atom_pos[3 * a + 1],
                  d->atom_pos[3 * b + 2] - d->atom_pos[3 * a + 2], emb1.data() + (size_t)k * NS, sh1.data() + (size_t)k * 4);
      } else {
        const int i = k - E_aa, r = b1[k];
        host_edge(M->h_ar, rp[3 * r] - d->atom_pos[3 * i], rp[3 * r + 1] - d->atom_pos[3 * i + 1], rp[3 * r + 2] - d->atom_pos[3 * i + 2],
                  emb1.data() + (size_t)k * NS, sh1.data() + (size_t)k * 4);
      }
    });
    int32_t* d_a1 = cxu(cx, a1.data(), a1.size());
    int32_t* d_b1 = cxu(cx, b1.data(), b1.size());
    float* d_emb1 = cxu(cx, emb1.data(), emb1.size());
    float* d_sh1 = cxu(cx, sh1.data(), sh1.size());
    if (!d_a1 || !d_b1 || !d_emb1 || !d_sh1) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed (confidence static sets)");
    K->st_a = d_a1; K->st_b = d_b1; K->st_emb = d_emb1; K->st_sh = d_sh1;
    // in-degrees of the static groups (messages are received at edge_src): atoms [aa, 0, ar = 1], residues [rr, 0, ra = its atoms]
    std::vector<int32_t> dst((size_t)(n_atom + n_rec) * 3, 0);
    for (int k = 0; k < E_aa; ++k) dst[(size_t)a1[k] * 3]++;
    for (int i = 0; i < n_atom; ++i) { dst[(size_t)i * 3 + 2] = 1; dst[(size_t)(n_atom + b1[E_aa + i]) * 3 + 2]++; }
    for (int k = 0; k < cx->E_rr; ++k) dst[(size_t)(n_atom + ei[k]) * 3]++;
    K->deg_static = cxu(cx, dst.data(), dst.size());
    if (!K->deg_static) return fail(ctx, DDK_ERR_NOMEM, "device allocation failed (confidence static degrees)");
  }
  K->n_nodes = Bm * (int64_t)n_lig + Bv * ((int64_t)n_atom + n_rec);
  K->gtab = cxu<int32_t>(cx, nullptr, 160);       // see ConfComplex::gtab
  K->deg_scratch = cxu<int32_t>(cx, nullptr, K->n_nodes);
  K->xa = cxu<float>(cx, nullptr, K->n_nodes * XW); K->xb = cxu<float>(cx, nullptr, K->n_nodes * XW);
  K->sum3 = cxu<float>(cx, nullptr, K->n_nodes * 3 * XW);
  K->deg3 = cxu<int32_t>(cx, nullptr, K->n_nodes * 3);
  if (!K->lig_x0 || !K->atom_x0 || !K->rec_x0 || !K->atom_pos || !K->gtab || !K->deg_scratch || !K->xa || !K->xb || !K->sum3 || !K->deg3)
    return fail(ctx, DDK_ERR_NOMEM, "device allocation failed in ddk_complex_set_atoms");
  int rcf = cx_stage_flush(ctx, cx);
  if (rcf) return rcf;
  {   // replicate the static sets for the Bm + 1 samples on the device, behind the upload on the upload stream
    const int64_t atom_base = Bm * n_lig, rec_base = atom_base + Bv * n_atom;
    const int64_t tot = Bv * ((int64_t)E_aa + n_atom);
    hipLaunchKernelGGL(conf_static_replicate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->up_stream, K->st_a, K->st_b, K->st_emb,
                       K->st_sh, E_aa, n_atom, n_rec, (int)Bv, atom_base, rec_base, K->off_aa, K->off_ar, K->off_ra, K->e_src, K->e_dst, K->e_emb, K->e_sh);
    if (hipGetLastError() != hipSuccess) return fail(ctx, DDK_ERR_HIP, "static set replication launch failed");
    hipError_t e = hipEventRecord(cx->ready, ctx->up_stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "event record");
  }
  return DDK_OK;
}

int ddk_confidence_forward(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float* out, void* stream) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  ConfModel* M = (ConfModel*)ctx->conf_model;
  if (!ctx->finalized || !M || !M->ready) return fail(ctx, DDK_ERR_STATE, "confidence model weights not loaded / finalised");
  if (!cx || !cx->conf) return fail(ctx, DDK_ERR_STATE, "ddk_complex_set_atoms has not run for this complex");
  if (B < 1 || B > cx->max_batch || !lig_pos || !out) return fail(ctx, DDK_ERR_INVALID, "bad batch / null argument");
  const ddk_config& c = ctx->cfg;
  ConfComplex* K = cx->conf;
  hipStream_t s = (hipStream_t)stream;
  { hipError_t we = cx_wait_ready(cx, s); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  const int n_lig = cx->n_lig, n_rec = cx->n_rec, n_atom = K->n_atom;
  const int64_t Bm = cx->max_batch;
  const int64_t atom_base = Bm * n_lig, rec_base = atom_base + (Bm + 1) * (int64_t)n_atom;      // (+ 1: the virtual ligand-free sample)
  hipError_t e;
#define CK(x, what) do { e = (x); if (e != hipSuccess) return hip_fail(ctx, e, what); } while (0)
  // ---- dynamic graphs ------------------------------------------------------------------------------
  GraphArgs G;
  G.lig_pos = lig_pos; G.rec_pos = cx->rec_pos; G.bond_src = cx->bond_src; G.bond_dst = cx->bond_dst;
  G.rr_src = cx->rr_src; G.rr_dst = cx->rr_dst; G.rr_outdeg = cx->rr_outdeg; G.rr_start = cx->rr_start;
  G.B = B; G.n_lig = n_lig; G.n_rec = n_rec; G.M = cx->M; G.E_rr = cx->E_rr;
  G.lig_r2 = c.lig_max_radius * c.lig_max_radius; G.cross_cutoff = M->sp.cross_cutoff;
  G.counts = cx->counts; G.offs = cx->offs; G.info = cx->info; G.levels = cx->levels; G.e_src = K->e_src; G.e_dst = K->e_dst; G.e_aux = K->e_aux;
  G.deg = K->deg_scratch; G.rec_node_base = (int)rec_base;
  G.cross_mirror = graph_cross_mirror_fits(n_lig, n_rec);      // lr / rl: one evaluation of the edge MLP per pair
  CK(launch_graph(G, K->cap4, s), "graph");
  EdgeFeatArgs EF;
  EF.cross_mirror = G.cross_mirror;
  EF.lig_pos = lig_pos; EF.rec_pos = cx->rec_pos; EF.bond_attr = cx->bond_attr; EF.rr_pre1 = cx->rr_pre1; EF.rr_sh = cx->rr_sh;
  EF.e_src = K->e_src; EF.e_dst = K->e_dst; EF.e_aux = K->e_aux; EF.info = cx->info; EF.e_emb = K->e_emb; EF.e_sh = K->e_sh;
  EF.lig = M->lig_edge; EF.rec = M->rec_edge; EF.cross = M->lr_edge; EF.sp = M->sp;
  EF.n_lig_total = B * n_lig; EF.rec_node_base = (int)rec_base; EF.n_rec = n_rec;
  EF.lig_latent = nullptr; EF.rec_latent = nullptr; EF.unconditional = 0.0f; EF.latent_dim = 0;
  CK(launch_edge_features(EF, K->cap4, s), "edge features");
  CK(hipMemsetAsync(K->gtab + 18, 0, 2 * sizeof(int32_t), s), "la counter");
  LaArgs LA;
  LA.lig_pos = lig_pos; LA.atom_pos = K->atom_pos; LA.B = B; LA.n_lig = n_lig; LA.n_atom = n_atom; LA.atom_node_base = (int)atom_base;
  LA.r2 = c.lig_max_radius * c.lig_max_radius; LA.mlp = M->la_edge; memcpy(LA.sigb, M->la_sigb, sizeof(LA.sigb));
  LA.gtab = K->gtab; LA.off_la = K->off_la; LA.off_al = K->off_al; LA.cap = K->cap_la;
  LA.e_src = K->e_src; LA.e_dst = K->e_dst; LA.e_emb = K->e_emb; LA.e_sh = K->e_sh;
  hipLaunchKernelGGL(conf_la_kernel, dim3(B, 8), dim3(256), 0, s, LA);
  CK(hipGetLastError(), "la graph");
  hipLaunchKernelGGL(conf_gtab_kernel, dim3(1), dim3(64), 0, s, K->gtab, cx->info, B, (int)Bm, K->E_aa, cx->E_rr, n_atom, (int)K->off_la, (int)K->off_al,
                     (int)K->off_aa, (int)K->off_ar, (int)K->off_ra, (int)K->off_vrr, (int)K->cap_la);
  CK(hipGetLastError(), "group table");
  // degrees: the static groups' constants, then the ligand-dependent groups (all of them live in front of the static sets)
  hipLaunchKernelGGL(conf_deg_init_kernel, dim3((unsigned)((K->n_nodes * 3 + 255) / 256)), dim3(256), 0, s, K->deg_static, atom_base, rec_base, K->n_nodes, n_atom,
                     n_rec, K->deg3);
  hipLaunchKernelGGL(conf_deg_kernel, dim3((unsigned)((K->off_aa + 255) / 256)), dim3(256), 0, s, K->gtab, K->e_src, K->deg3, K->off_aa);
  CK(hipGetLastError(), "degrees");
  // the virtual ligand-free sample (atom / residue sample Bm) carries the pose-independent work of the first layers
  const bool share0 = B > 1 && c.num_conv_layers >= 2 && ctx->layer0_dedup;
  if (share0 && cx->E_rr > 0) {
    hipLaunchKernelGGL(conf_vrr_kernel, dim3((unsigned)((cx->E_rr * 8 + 255) / 256)), dim3(256), 0, s, K->gtab, cx->E_rr, (int)(Bm * n_rec), K->off_vrr, K->e_src,
                       K->e_dst, K->e_emb, K->e_sh);
    CK(hipGetLastError(), "virtual rec-rec records");
  }
  // level-A pruning of the second-to-last layer (ddk_set_receptive_field_pruning; needs a layer between the shared layer 0 and the last one)
  const bool pruneA = ctx->prune && c.num_conv_layers >= 3 && cx->E_rr > 0;
  // layer 1 evaluates a static group only into the receivers whose messages differ from the virtual sample's (needs layer 1 to be a full layer:
  // not the last one and not one of the two level layers)
  const bool share1 = share0 && pruneA && c.num_conv_layers >= 5;
  if (pruneA) {
    ConfLevelArgs LV;
    LV.gtab = K->gtab; LV.e_src = K->e_src; LV.e_dst = K->e_dst; LV.e_emb = K->e_emb; LV.e_sh = K->e_sh; LV.flag_a = K->flag_a; LV.flag_r = K->flag_r;
    LV.fl[0] = K->flag_a; LV.fl[1] = K->flag_a; LV.fl[2] = K->flag_r; LV.fl[3] = K->flag_r; LV.with_virtual = 0;
    LV.B = B; LV.n_atom = n_atom; LV.n_rec = n_rec; LV.E_aa = K->E_aa; LV.E_rr = cx->E_rr; LV.atom_base = atom_base; LV.rec_base = rec_base;
    LV.off_aa = K->off_aa; LV.off_ar = K->off_ar; LV.off_ra = K->off_ra; LV.off_vrr = K->off_vrr; LV.off_scr = K->off_scr; LV.Bm = Bm;
    LV.cursors = K->gtab + 82; LV.tabA = K->gtab + 64;
    CK(hipMemsetAsync(K->flag_a, 0, (size_t)Bm * n_atom, s), "level flags");
    CK(hipMemsetAsync(K->flag_r, 0, (size_t)Bm * n_rec, s), "level flags");
    CK(hipMemsetAsync(K->gtab + 82, 0, 4 * sizeof(int32_t), s), "level cursors");
    const int64_t n_mark = K->cap_la + (int64_t)B * n_lig * n_rec;      // upper bounds of the al and rl edge counts
    hipLaunchKernelGGL(conf_level_flags_kernel, dim3((unsigned)((n_mark + 255) / 256)), dim3(256), 0, s, LV);
    hipLaunchKernelGGL(conf_level_compact_kernel, dim3(B, 4, COMPACT_SLICES), dim3(256), 0, s, LV);
    hipLaunchKernelGGL(conf_level_table_kernel, dim3(1), dim3(64), 0, s, LV);
    CK(hipGetLastError(), "level-A compaction");
    if (c.num_conv_layers >= 4) {      // level B for the third-to-last layer: the same compaction on the wider flag set, second scratch region
      uint8_t* fb_a = K->flag_a + (size_t)Bm * n_atom;
      uint8_t* fb_r = K->flag_r + (size_t)Bm * n_rec;
      CK(hipMemcpyAsync(fb_a, K->flag_a, (size_t)Bm * n_atom, hipMemcpyDeviceToDevice, s), "level-B flags");
      CK(hipMemcpyAsync(fb_r, K->flag_r, (size_t)Bm * n_rec, hipMemcpyDeviceToDevice, s), "level-B flags");
      CK(hipMemsetAsync(K->gtab + 114, 0, 4 * sizeof(int32_t), s), "level cursors");
      ConfLevelBArgs LB;
      LB.tabA = K->gtab + 64; LB.e_dst = K->e_dst; LB.flag_a = fb_a; LB.flag_r = fb_r; LB.atom_base = atom_base; LB.rec_base = rec_base;
      hipLaunchKernelGGL(conf_level_b_flags_kernel, dim3((unsigned)((K->cap_scr + 255) / 256)), dim3(256), 0, s, LB, K->cap_scr);
      ConfLevelArgs L2 = LV;
      L2.flag_a = fb_a; L2.flag_r = fb_r; L2.fl[0] = fb_a; L2.fl[1] = fb_a; L2.fl[2] = fb_r; L2.fl[3] = fb_r;
      L2.off_scr = K->off_scr + K->cap_scr; L2.cursors = K->gtab + 114; L2.tabA = K->gtab + 96;
      hipLaunchKernelGGL(conf_level_compact_kernel, dim3(B, 4, COMPACT_SLICES), dim3(256), 0, s, L2);
      hipLaunchKernelGGL(conf_level_table_kernel, dim3(1), dim3(64), 0, s, L2);
      CK(hipGetLastError(), "level-B compaction");
    }
    if (share1) {      // layer 1: need flags from the level-A flags (= the nodes that received a ligand message in layer 0), third scratch region
      CK(hipMemcpyAsync(K->need[0], K->flag_a, (size_t)Bm * n_atom, hipMemcpyDeviceToDevice, s), "need flags");
      CK(hipMemcpyAsync(K->need[1], K->flag_a, (size_t)Bm * n_atom, hipMemcpyDeviceToDevice, s), "need flags");
      CK(hipMemcpyAsync(K->need[2], K->flag_r, (size_t)Bm * n_rec, hipMemcpyDeviceToDevice, s), "need flags");
      CK(hipMemcpyAsync(K->need[3], K->flag_r, (size_t)Bm * n_rec, hipMemcpyDeviceToDevice, s), "need flags");
      CK(hipMemsetAsync(K->gtab + 146, 0, 4 * sizeof(int32_t), s), "layer-1 cursors");
      ConfNeedArgs NA;
      NA.gtab = K->gtab; NA.e_src = K->e_src; NA.e_dst = K->e_dst; NA.flag_a = K->flag_a; NA.flag_r = K->flag_r;
      for (int y = 0; y < 4; ++y) NA.need[y] = K->need[y];
      NA.B = B; NA.n_atom = n_atom; NA.n_rec = n_rec; NA.E_aa = K->E_aa; NA.E_rr = cx->E_rr; NA.atom_base = atom_base; NA.rec_base = rec_base;
      NA.off_aa = K->off_aa; NA.off_ar = K->off_ar; NA.off_ra = K->off_ra;
      const int64_t n_st = (int64_t)B * ((int64_t)K->E_aa + 2 * (int64_t)n_atom + cx->E_rr);
      hipLaunchKernelGGL(conf_need_flags_kernel, dim3((unsigned)((n_st + 255) / 256)), dim3(256), 0, s, NA);
      ConfLevelArgs L1 = LV;
      for (int y = 0; y < 4; ++y) L1.fl[y] = K->need[y];
      L1.with_virtual = 1; L1.off_scr = K->off_scr + 2 * K->cap_scr; L1.cursors = K->gtab + 146; L1.tabA = K->gtab + 128;
      hipLaunchKernelGGL(conf_level_compact_kernel, dim3(B + 1, 4, COMPACT_SLICES), dim3(256), 0, s, L1);
      hipLaunchKernelGGL(conf_level_table_kernel, dim3(1), dim3(64), 0, s, L1);
      CK(hipGetLastError(), "layer-1 compaction");
    }
  }
  // ---- node features and the conv stack ----------------------------------------------------------
  float *xin = K->xa, *xout = K->xb;
  const int64_t tot = K->n_nodes * XW;
  hipLaunchKernelGGL(conf_node_init_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, K->lig_x0, K->atom_x0, K->rec_x0, atom_base, rec_base, K->n_nodes,
                     n_lig, n_atom, n_rec, xin);
  CK(hipGetLastError(), "node init");
  for (int l = 0; l < c.num_conv_layers; ++l) {
    const ConvLayerDev& L = ctx->conv[l];
    const bool last = l == c.num_conv_layers - 1;      // all_atom_score_model.py:241 "last layer optimisation": ligand updates only
    CK(hipMemsetAsync(K->sum3, 0, (size_t)(last ? atom_base : K->n_nodes) * 3 * XW * sizeof(float), s), "memset sum3");
    CK(hipMemsetAsync(cx->info + I_CNT + (l % 8), 0, sizeof(int32_t), s), "counter reset");
    ConvLaunch a;
    a.x = xin; a.src = K->e_src; a.dst = K->e_dst; a.edge_attr = K->e_emb; a.sh = K->e_sh; a.sum = K->sum3;
    a.tile_info = cx->info; a.counter = cx->info + I_CNT + (l % 8); a.gather = 1;
    a.mode = 1; a.n_groups = 9; a.n_active = last ? 3 : 9; a.n_slots = 3;
    a.slots = 0;
    for (int g = 0; g < 9; ++g) a.slots |= (uint32_t)(g % 3) << (2 * g);
    const bool sh0 = share0 && l == 0 && !last;                               // pose-independent groups once per batch, on the virtual sample (conf_gtab_kernel)
    const bool sh1 = share1 && l == 1;                                        // ... and in layer 1 for the receivers that see no change
    const bool levelA = pruneA && l == c.num_conv_layers - 2 && !sh0 && !sh1;     // only level-A receivers of the static groups
    const bool levelB = pruneA && c.num_conv_layers >= 4 && l == c.num_conv_layers - 3 && !sh0 && !sh1;
    a.gbeg = K->gtab + (sh0 ? 32 : (sh1 ? 128 : (levelA ? 64 : (levelB ? 96 : 0)))); a.gend = a.gbeg + 9;
    CK(launch_conv_fused(L, a, ctx->n_cu, s), "conv_fused (confidence)");
    ConfFinalizeArgs FA;
    FA.sum3 = K->sum3; FA.deg3 = K->deg3; FA.x_in = xin; FA.bn_mean = L.bn_mean; FA.bn_scale = L.bn_scale; FA.bn_bias = L.bn_bias;
    FA.n_nodes = K->n_nodes; FA.n_update = last ? atom_base : K->n_nodes; FA.atom0 = atom_base; FA.rec0 = rec_base;
    FA.vatom0 = atom_base + Bm * n_atom; FA.vrec0 = rec_base + Bm * n_rec;
    FA.dout = L.dout; FA.share = sh0 ? 1 : (sh1 ? 2 : 0); FA.n_atom = n_atom; FA.n_rec = n_rec; FA.x_out = xout;
    for (int y = 0; y < 4; ++y) FA.need[y] = K->need[y];
    hipLaunchKernelGGL(conf_finalize_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, FA);
    CK(hipGetLastError(), "conf finalize");
    float* t = xin; xin = xout; xout = t;
  }
  cx->x_last = xin;
  cx->last_B = B;
  CK(launch_conf_head(M->pred, xin, B, n_lig, out, K->gtab + 19, s), "confidence head");
#undef CK
  return DDK_OK;
}

// group table of the last confidence forward, copied asynchronously (include/ddk.h)
int ddk_confidence_status(ddk_ctx* ctx, ddk_complex* cx, int32_t* host_out20, void* stream) {
  if (!ctx || !cx || !cx->conf || !host_out20) return DDK_ERR_INVALID;
  hipError_t e = hipMemcpyAsync(host_out20, cx->conf->gtab, 20 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "confidence status copy");
}

// Test hook: edge counts of the nine groups of the last confidence forward ([ll lr la aa al ar rr rl ra]) + la overflow flag.
int ddk_debug_conf_counts(ddk_ctx* ctx, ddk_complex* cx, int32_t* out10) {
  if (!ctx || !cx || !cx->conf || !out10) return DDK_ERR_INVALID;
  int32_t g[32];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(g, cx->conf->gtab, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(ctx, DDK_ERR_HIP, "gtab read-back failed");
  for (int k = 0; k < 9; ++k) out10[k] = g[9 + k] - g[k];
  out10[9] = g[19];
  return DDK_OK;
}

int ddk_debug_conf_table(ddk_ctx* ctx, ddk_complex* cx, int32_t which, int32_t* out9) {
  if (!ctx || !cx || !cx->conf || !out9 || which < 0 || which > 4) return DDK_ERR_INVALID;
  int32_t g[18];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(g, cx->conf->gtab + 32 * which, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(ctx, DDK_ERR_HIP, "gtab read-back failed");
  for (int k = 0; k < 9; ++k) out9[k] = g[9 + k] - g[k];
  return DDK_OK;
}

// Test hook: all node features [n_nodes, XW] after the last confidence forward (device numbering) and the per-slot degrees.
int ddk_debug_conf_nodes(ddk_ctx* ctx, ddk_complex* cx, float* x, int32_t* deg3, int64_t n_nodes) {
  if (!ctx || !cx || !cx->conf || !cx->x_last) return DDK_ERR_INVALID;
  if (n_nodes != cx->conf->n_nodes) return fail(ctx, DDK_ERR_INVALID, "node count");
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(x, cx->x_last, (size_t)n_nodes * XW * 4, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(deg3, cx->conf->deg3, (size_t)n_nodes * 12, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(ctx, DDK_ERR_HIP, "copy failed");
  return DDK_OK;
}

// Test hook: raw edge arrays [off, off+n) of the last confidence forward and the group table (18 ints).
int ddk_debug_conf_edges(ddk_ctx* ctx, ddk_complex* cx, int64_t off, int64_t n, int32_t* src, int32_t* dst, float* emb, float* sh, int32_t* gtab18) {
  if (!ctx || !cx || !cx->conf) return DDK_ERR_INVALID;
  ConfComplex* K = cx->conf;
  if (hipDeviceSynchronize() != hipSuccess) return fail(ctx, DDK_ERR_HIP, "sync failed");
  if (gtab18 && hipMemcpy(gtab18, K->gtab, 18 * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(ctx, DDK_ERR_HIP, "copy failed");
  if (n > 0 && (off < 0 || off + n > K->cap_total)) return fail(ctx, DDK_ERR_INVALID, "range");
  if (n > 0 && (hipMemcpy(src, K->e_src + off, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(dst, K->e_dst + off, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(emb, K->e_emb + off * NS, n * NS * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sh, K->e_sh + off * 4, n * 16, hipMemcpyDeviceToHost) != hipSuccess))
    return fail(ctx, DDK_ERR_HIP, "copy failed");
  return DDK_OK;
}

}  // extern "C"
