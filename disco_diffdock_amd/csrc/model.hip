// Score-model level host code: small-weight upload, per-complex static precompute, per-forward host scalars,
// kernel orchestration for ddk_score_forward / ddk_se3_update / ddk_sample (include/ddk.h).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>

#include "model.h"

namespace ddk {

static const int LIG_DIMS[16] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};   // process_mols.py:62-79
static const int REC_DIM = 38;                                                             // process_mols.py:88-90

static const HostTensor* getw(ddk_ctx* ctx, const std::string& name, std::initializer_list<int64_t> shape) {
  auto it = ctx->weights.find(name);
  if (it == ctx->weights.end()) { ctx->err = "missing state_dict key: " + name; return nullptr; }
  if (it->second.shape != std::vector<int64_t>(shape)) { ctx->err = "shape mismatch for " + name; return nullptr; }
  return &it->second;
}

// columns [c0, c1) of a row-major [rows, cols] matrix
static std::vector<float> cols(const HostTensor* t, int c0, int c1) {
  const int rows = (int)t->shape[0], nc = (int)t->shape[1];
  std::vector<float> o((size_t)rows * (c1 - c0));
  for (int r = 0; r < rows; ++r)
    for (int c = c0; c < c1; ++c) o[(size_t)r * (c1 - c0) + (c - c0)] = t->data[(size_t)r * nc + c];
  return o;
}

static bool smearing(ddk_ctx* ctx, const char* name, float stop, EdgeMlpDev& m, std::vector<float>* host_off, float* host_coeff) {
  // GaussianSmearing (tensor_layers.py:171-181): offset buffer from the checkpoint, coeff = -0.5/(offset[1]-offset[0])^2
  std::vector<float> off(DE);
  auto it = ctx->weights.find(std::string(name) + "_distance_expansion.offset");
  if (it != ctx->weights.end() && it->second.data.size() == (size_t)DE) off = it->second.data;
  else for (int k = 0; k < DE; ++k) off[k] = stop * (float)k / (float)(DE - 1);
  const double d = (double)(off[1] - off[0]);
  m.coeff = (float)(-0.5 / (d * d));
  m.step = off[1] - off[0];
  m.offset = dev_upload(ctx, off);
  if (host_off) *host_off = off;
  if (host_coeff) *host_coeff = m.coeff;
  return m.offset != nullptr;
}

static std::vector<float> transpose_rm(const std::vector<float>& w, int rows, int cols_) {      // [rows][cols] -> [cols][rows]
  std::vector<float> t((size_t)rows * cols_);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols_; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols_ + c];
  return t;
}

int model_finalize(ddk_ctx* ctx) {
  if (ctx->model) { delete (Model*)ctx->model; ctx->model = nullptr; }
  if (ctx->host_only || ctx->weights.find("lig_node_embedding.additional_features_embedder.weight") == ctx->weights.end())
    return DDK_OK;   // packing-only / operator-only context (no score model on the device)
  const ddk_config& c = ctx->cfg;
  if (c.latent_dim != 0 && c.latent_vocab != 1)
    return fail(ctx, DDK_ERR_INVALID, "latent conditioning is implemented for the equivariant-latent models only (latent_vocab == 1)");
  if (c.latent_dim < 0 || c.latent_dim > 8) return fail(ctx, DDK_ERR_INVALID, "latent_dim out of range");
  const int LD = c.latent_dim, LE = 2 * c.latent_dim;   // node / edge latent columns (latent_dim * max(latent_vocab, 2))
  Model* M = new Model();
  ctx->model = M;
  ModelHost& H = M->host;
  ModelDev& D = M->dev;
  const int lm = c.lm_embedding_dim;
#define GET(var, name, ...) const HostTensor* var = getw(ctx, name, {__VA_ARGS__}); if (!var) return DDK_ERR_INVALID
  // ---- node encoders -----------------------------------------------------------------------
  int off = 0;
  for (int i = 0; i < 16; ++i) {
    GET(t, "lig_node_embedding.atom_embedding_list." + std::to_string(i) + ".weight", LIG_DIMS[i], NS);
    H.lig_table_off.push_back(off);
    H.lig_tables.insert(H.lig_tables.end(), t->data.begin(), t->data.end());
    off += LIG_DIMS[i];
  }
  GET(lw, "lig_node_embedding.additional_features_embedder.weight", NS, NS + SIG + LD);
  GET(lb, "lig_node_embedding.additional_features_embedder.bias", NS);
  H.lig_w_emb = cols(lw, 0, NS); H.lig_w_sig = cols(lw, NS, NS + SIG); H.lig_b = lb->data;
  GET(rt, "rec_node_embedding.atom_embedding_list.0.weight", REC_DIM, NS);
  GET(rw, "rec_node_embedding.additional_features_embedder.weight", NS, NS + lm + SIG + LD);
  GET(rb, "rec_node_embedding.additional_features_embedder.bias", NS);
  H.rec_table = rt->data; H.rec_w_emb = cols(rw, 0, NS); H.rec_w_esm = cols(rw, NS, NS + lm);
  H.rec_w_sig = cols(rw, NS + lm, NS + lm + SIG); H.rec_b = rb->data;
  {
    std::vector<float> w_lm_t((size_t)lm * NS);      // [lm][NS]: the 24 outputs of one feature are contiguous
    for (int o = 0; o < NS; ++o)
      for (int k = 0; k < lm; ++k) w_lm_t[(size_t)k * NS + o] = H.rec_w_esm[(size_t)o * lm + k];
    D.rec_table = dev_upload(ctx, H.rec_table); D.rec_w_emb = dev_upload(ctx, H.rec_w_emb); D.rec_w_lm_t = dev_upload(ctx, w_lm_t); D.rec_b = dev_upload(ctx, H.rec_b);
    if (!D.rec_table || !D.rec_w_emb || !D.rec_w_lm_t || !D.rec_b) return fail(ctx, DDK_ERR_NOMEM, "alloc");
  }
  D.latent_dim = LD;
  D.lig_w_lat = D.rec_w_lat = D.lig_node_unc = D.rec_node_unc = nullptr;
  if (LD > 0) {
    D.lig_w_lat = dev_upload(ctx, cols(lw, NS + SIG, NS + SIG + LD));
    D.rec_w_lat = dev_upload(ctx, cols(rw, NS + lm + SIG, NS + lm + SIG + LD));
    if (c.latent_droprate > 0) {
      GET(lu, "lig_node_unconditional_embedding", 1, NS);
      GET(ru, "rec_node_unconditional_embedding", 1, NS);
      D.lig_node_unc = dev_upload(ctx, lu->data);
      D.rec_node_unc = dev_upload(ctx, ru->data);
    }
  }
  // ---- edge embedding MLPs -------------------------------------------------------------------
  auto edge_mlp = [&](const char* name, int n_bond, bool sigma_first, EdgeMlpDev& m, std::vector<float>* w1s,
                      std::vector<float>* b1, std::vector<float>* w1d_host, int lat_cols = 0, const char* unc_name = nullptr) -> bool {
    const int in = n_bond + (w1s ? SIG : 0) + DE + lat_cols;
    const HostTensor* w0 = getw(ctx, std::string(name) + ".0.weight", {NS, in});
    const HostTensor* b0 = getw(ctx, std::string(name) + ".0.bias", {NS});
    const HostTensor* w3 = getw(ctx, std::string(name) + ".3.weight", {NS, NS});
    const HostTensor* b3 = getw(ctx, std::string(name) + ".3.bias", {NS});
    if (!w0 || !b0 || !w3 || !b3) return false;
    int c_sig, c_d;
    if (!w1s) { c_sig = -1; c_d = n_bond; }
    else if (sigma_first) { c_sig = n_bond; c_d = n_bond + SIG; }       // [bond | sigma | dist]
    else { c_d = n_bond; c_sig = n_bond + DE; }                           // [dist | sigma]
    std::vector<float> w1d = cols(w0, c_d, c_d + DE);
    if (w1d_host) *w1d_host = w1d;
    m.w1d = dev_upload(ctx, w1d);
    m.w1d_t = dev_upload(ctx, transpose_rm(w1d, NS, DE));
    m.w2_t = dev_upload(ctx, transpose_rm(w3->data, NS, NS));
    m.w1b = n_bond ? dev_upload(ctx, cols(w0, 0, n_bond)) : nullptr;
    m.w2 = dev_upload(ctx, w3->data);
    m.b2 = dev_upload(ctx, b3->data);
    m.w1l = lat_cols ? dev_upload(ctx, cols(w0, in - lat_cols, in)) : nullptr;   // latent columns are the last ones
    m.unc = nullptr;
    if (unc_name && c.latent_droprate > 0 && LD > 0) {
      const HostTensor* u = getw(ctx, unc_name, {1, NS});
      if (!u) return false;
      m.unc = dev_upload(ctx, u->data);
    }
    if (w1s) *w1s = cols(w0, c_sig, c_sig + SIG);
    *b1 = b0->data;
    return m.w1d && m.w2 && m.b2 && m.w1d_t && m.w2_t;
  };
  std::vector<float> fe_b1;
  if (!edge_mlp("lig_edge_embedding", 4, true, D.lig_edge, &H.le_w1s, &H.le_b1, nullptr, LE, "lig_edge_unconditional_embedding")) return DDK_ERR_INVALID;
  if (!edge_mlp("rec_edge_embedding", 0, true, D.rec_edge, &H.re_w1s, &H.re_b1, &H.re_w1d, LE, "rec_edge_unconditional_embedding")) return DDK_ERR_INVALID;
  if (!edge_mlp("cross_edge_embedding", 0, true, D.cross_edge, &H.ce_w1s, &H.ce_b1, nullptr, LE, "cross_edge_unconditional_embedding")) return DDK_ERR_INVALID;
  if (!smearing(ctx, "lig", c.lig_max_radius, D.lig_edge, nullptr, nullptr)) return fail(ctx, DDK_ERR_NOMEM, "alloc");
  if (!smearing(ctx, "rec", c.rec_max_radius, D.rec_edge, &H.rec_offset, &H.rec_coeff)) return fail(ctx, DDK_ERR_NOMEM, "alloc");
  if (!smearing(ctx, "cross", c.cross_max_distance, D.cross_edge, nullptr, nullptr)) return fail(ctx, DDK_ERR_NOMEM, "alloc");
  M->confidence_mode = c.confidence_mode != 0;
  if (M->confidence_mode) {
    // score_model.py:110-121: the confidence_predictor replaces the centre / torsion heads; it reads [x[:, :ns] | x[:, -ns:]] of the ligand rows,
    // the full-width layout of num_conv_layers >= 3 (:264)
    if (c.num_conv_layers < 3) return fail(ctx, DDK_ERR_INVALID, "confidence_mode needs num_conv_layers >= 3 (the predictor reads the 0e and 0o scalars)");
    if (c.latent_dim > 0) return fail(ctx, DDK_ERR_INVALID, "confidence_mode with latent conditioning is not implemented");
    int rcp = conf_predictor_load(ctx, M->pred);
    if (rcp) return rcp;
  }
  if (!M->confidence_mode) {
  if (!edge_mlp("center_edge_embedding", 0, false, D.center_edge, &H.cen_w1s, &H.cen_b1, nullptr)) return DDK_ERR_INVALID;
  if (!smearing(ctx, "center", c.center_max_distance, D.center_edge, nullptr, nullptr)) return fail(ctx, DDK_ERR_NOMEM, "alloc");
  // ---- tr / rot head -------------------------------------------------------------------------
  {
    GET(w0, "final_conv.fc.0.weight", 2 * NS, 2 * NS);
    GET(b0, "final_conv.fc.0.bias", 2 * NS);
    GET(w4, "final_conv.fc.4.weight", 144, 2 * NS);
    GET(b4, "final_conv.fc.4.bias", 144);
    D.fc_w0 = dev_upload(ctx, w0->data); D.fc_b0 = dev_upload(ctx, b0->data);
    D.fc_w4 = dev_upload(ctx, w4->data); D.fc_b4 = dev_upload(ctx, b4->data);
    for (int k = 0; k < 4; ++k) D.fc_bn_scale[k] = 1.0f;
    if (c.batch_norm) {
      GET(bw, "final_conv.batch_norm.weight", 4);
      GET(bv, "final_conv.batch_norm.running_var", 4);
      for (int k = 0; k < 4; ++k) D.fc_bn_scale[k] = powf(bv->data[k] + 1e-5f, -0.5f) * bw->data[k];
    }
    GET(t0, "tr_final_layer.0.weight", NS, 1 + SIG);
    GET(tb0, "tr_final_layer.0.bias", NS);
    GET(t3, "tr_final_layer.3.weight", 1, NS);
    GET(tb3, "tr_final_layer.3.bias", 1);
    GET(r0, "rot_final_layer.0.weight", NS, 1 + SIG);
    GET(rb0, "rot_final_layer.0.bias", NS);
    GET(r3, "rot_final_layer.3.weight", 1, NS);
    GET(rb3, "rot_final_layer.3.bias", 1);
    D.tr_w0n = dev_upload(ctx, cols(t0, 0, 1)); D.tr_w3 = dev_upload(ctx, t3->data); D.tr_b3 = tb3->data[0];
    D.rot_w0n = dev_upload(ctx, cols(r0, 0, 1)); D.rot_w3 = dev_upload(ctx, r3->data); D.rot_b3 = rb3->data[0];
    H.tr_w0s = cols(t0, 1, 1 + SIG); H.tr_b0 = tb0->data; H.rot_w0s = cols(r0, 1, 1 + SIG); H.rot_b0 = rb0->data;
  }
  // ---- torsion head --------------------------------------------------------------------------
  if (!c.no_torsion) {
    if (!edge_mlp("final_edge_embedding", 0, false, D.final_edge, nullptr, &fe_b1, nullptr)) return DDK_ERR_INVALID;
    D.final_edge.coeff = D.lig_edge.coeff; D.final_edge.step = D.lig_edge.step; D.final_edge.offset = D.lig_edge.offset;
    D.final_edge_b1 = dev_upload(ctx, fe_b1);
    GET(w0, "tor_bond_conv.fc.0.weight", NE, NE);
    GET(b0, "tor_bond_conv.fc.0.bias", NE);
    GET(w4, "tor_bond_conv.fc.4.weight", 2 * NV * NS, NE);
    GET(b4, "tor_bond_conv.fc.4.bias", 2 * NV * NS);
    D.tb_w0 = dev_upload(ctx, w0->data); D.tb_b0 = dev_upload(ctx, b0->data);
    D.tb_w4 = dev_upload(ctx, w4->data); D.tb_b4 = dev_upload(ctx, b4->data);
    std::vector<float> sc(2 * NS, 1.f), mn(2 * NS, 0.f), bi(2 * NS, 0.f);
    if (c.batch_norm) {   // irreps 24x0o + 24x0e: only the 0e half (channels 24..47) has mean / bias
      GET(bw, "tor_bond_conv.batch_norm.weight", 2 * NS);
      GET(bb, "tor_bond_conv.batch_norm.bias", NS);
      GET(bm, "tor_bond_conv.batch_norm.running_mean", NS);
      GET(bv, "tor_bond_conv.batch_norm.running_var", 2 * NS);
      for (int k = 0; k < 2 * NS; ++k) sc[k] = powf(bv->data[k] + 1e-5f, -0.5f) * bw->data[k];
      for (int k = 0; k < NS; ++k) { mn[NS + k] = bm->data[k]; bi[NS + k] = bb->data[k]; }
    }
    D.tb_bn_scale = dev_upload(ctx, sc); D.tb_bn_mean = dev_upload(ctx, mn); D.tb_bn_bias = dev_upload(ctx, bi);
    GET(f0, "tor_final_layer.0.weight", NS, 2 * NS);
    GET(f3, "tor_final_layer.3.weight", 1, NS);
    D.tf_w0 = dev_upload(ctx, f0->data); D.tf_w3 = dev_upload(ctx, f3->data);
  }
  }      // (!confidence_mode)
#undef GET
  for (int l = 0; l < c.num_conv_layers; ++l)
    if (!ctx->conv[l].has_weights) return fail(ctx, DDK_ERR_INVALID, "score model checkpoint lacks conv_layers." + std::to_string(l));
  // ---- AR latent model predictors (models/pretrained_score_encoder.py:24-45), present when this context holds the AR checkpoint's
  //      own copy of the score model: Linear - BatchNorm1d - ReLU - Linear - BatchNorm1d - ReLU - Linear, BatchNorm (eval) folded ----
  if (ctx->weights.find("latent_s_predictor.0.weight") != ctx->weights.end()) {
    auto w0it = ctx->weights.find("latent_s_predictor.0.weight");
    if (w0it->second.shape.size() != 2) return fail(ctx, DDK_ERR_INVALID, "latent_s_predictor.0.weight must be 2-d");
    const int Hd = (int)w0it->second.shape[0], nin = (int)w0it->second.shape[1];
    if (Hd < 1 || Hd > AR_H || nin < 2 || nin % 2 || nin / 2 > AR_NS_MAX || nin > XW)
      return fail(ctx, DDK_ERR_INVALID, "AR predictor: hidden width must be <= 128 and the input 2*ns with ns <= 24 (num_conv_layers >= 3 layout)");
    auto pack = [&](const char* pre, ArMlpDev& D) -> bool {
      const std::string p(pre);
      const HostTensor* w0 = getw(ctx, p + ".0.weight", {Hd, nin});
      const HostTensor* b0 = getw(ctx, p + ".0.bias", {Hd});
      const HostTensor* w4 = getw(ctx, p + ".4.weight", {Hd, Hd});
      const HostTensor* b4 = getw(ctx, p + ".4.bias", {Hd});
      const HostTensor* w8 = getw(ctx, p + ".8.weight", {1, Hd});
      const HostTensor* b8 = getw(ctx, p + ".8.bias", {1});
      if (!w0 || !b0 || !w4 || !b4 || !w8 || !b8)
        return false;     // (latent_dim of the predictors is 1: model_utils.py:133-139 builds PretrainedScoreEncoder(latent_dim=1))
      std::vector<float> W0 = w0->data, B0 = b0->data, W4 = w4->data, B4 = b4->data;
      auto fold = [&](const char* idx, std::vector<float>& W, std::vector<float>& Bv, int in) -> bool {
        if (ctx->weights.find(p + "." + idx + ".running_var") == ctx->weights.end()) return true;    // latent_no_batchnorm
        const HostTensor* g = getw(ctx, p + "." + idx + ".weight", {Hd});
        const HostTensor* be = getw(ctx, p + "." + idx + ".bias", {Hd});
        const HostTensor* rm = getw(ctx, p + "." + idx + ".running_mean", {Hd});
        const HostTensor* rv = getw(ctx, p + "." + idx + ".running_var", {Hd});
        if (!g || !be || !rm || !rv) return false;
        for (int j = 0; j < Hd; ++j) {
          const float sc = g->data[j] / sqrtf(rv->data[j] + 1e-5f);      // nn.BatchNorm1d default eps
          for (int k = 0; k < in; ++k) W[(size_t)j * in + k] *= sc;
          Bv[j] = (Bv[j] - rm->data[j]) * sc + be->data[j];
        }
        return true;
      };
      if (!fold("1", W0, B0, nin) || !fold("5", W4, B4, Hd)) return false;
      D.w0 = dev_upload(ctx, W0); D.b0 = dev_upload(ctx, B0); D.w4 = dev_upload(ctx, W4); D.b4 = dev_upload(ctx, B4);
      D.w8 = dev_upload(ctx, w8->data); D.b8 = b8->data[0];
      return D.w0 && D.b0 && D.w4 && D.b4 && D.w8;
    };
    if (!pack("latent_s_predictor", M->ar_s) || !pack("latent_r_predictor", M->ar_r)) return DDK_ERR_INVALID;
    M->has_ar = true; M->ar_ns = nin / 2; M->ar_H = Hd;
  }
  H.ready = true;
  return DDK_OK;
}

void model_destroy(ddk_ctx* ctx) {
  if (ctx->model) { delete (Model*)ctx->model; ctx->model = nullptr; }
}

static void matvec(const std::vector<float>& W, const float* x, int rows, int colsn, const float* bias, float* out) {
  for (int r = 0; r < rows; ++r) {
    float a = bias ? bias[r] : 0.0f;
    for (int k = 0; k < colsn; ++k) a += W[(size_t)r * colsn + k] * x[k];
    out[r] = a;
  }
}

// host scalars of one forward at diffusion time (t_tr, t_rot, t_tor); fp32 like the reference's tensors
static int make_step_params(ddk_ctx* ctx, float t_tr, float t_rot, float t_tor, StepParams& sp) {
  const ddk_config& c = ctx->cfg;
  const ModelHost& H = ((Model*)ctx->model)->host;
  const bool conf_mode = ((Model*)ctx->model)->confidence_mode;
  // sinusoidal_embedding(embedding_scale * t, 32)  (utils/diffusion_utils.py:58-69)
  float emb[SIG];
  const int half = SIG / 2;
  const double e = log(10000.0) / (double)(half - 1);
  const float ts = c.embedding_scale * t_tr;
  for (int k = 0; k < half; ++k) {
    const float f = expf((float)k * (float)(-e));
    const float a = ts * f;
    emb[k] = sinf(a);
    emb[half + k] = cosf(a);
  }
  matvec(H.lig_w_sig, emb, NS, SIG, nullptr, sp.lig_node_sig);
  matvec(H.rec_w_sig, emb, NS, SIG, nullptr, sp.rec_node_sig);
  matvec(H.le_w1s, emb, NS, SIG, H.le_b1.data(), sp.lig_edge_sigb);
  matvec(H.re_w1s, emb, NS, SIG, H.re_b1.data(), sp.rec_edge_sigb);
  matvec(H.ce_w1s, emb, NS, SIG, H.ce_b1.data(), sp.cross_edge_sigb);
  if (conf_mode) {
    // score_model.py:186-189: in confidence_mode complex_t is used as sigma directly (the cross cutoff is 3 t_tr + 20); no heads
    sp.tr_sigma = t_tr; sp.rot_sigma = t_rot; sp.tor_sigma = t_tor;
    sp.cross_cutoff = c.dynamic_max_cross ? sp.tr_sigma * 3.0f + 20.0f : c.cross_max_distance;
    sp.so3_norm = 1.0f; sp.torus_norm_sqrt = 1.0f;
    return DDK_OK;
  }
  matvec(H.cen_w1s, emb, NS, SIG, H.cen_b1.data(), sp.center_edge_sigb);
  matvec(H.tr_w0s, emb, NS, SIG, H.tr_b0.data(), sp.tr_sigb);
  matvec(H.rot_w0s, emb, NS, SIG, H.rot_b0.data(), sp.rot_sigb);
  // t_to_sigma on fp32 tensors (utils/diffusion_utils.py:12-16)
  sp.tr_sigma = powf(c.tr_sigma_min, 1.0f - t_tr) * powf(c.tr_sigma_max, t_tr);
  sp.rot_sigma = powf(c.rot_sigma_min, 1.0f - t_rot) * powf(c.rot_sigma_max, t_rot);
  sp.tor_sigma = powf(c.tor_sigma_min, 1.0f - t_tor) * powf(c.tor_sigma_max, t_tor);
  sp.cross_cutoff = c.dynamic_max_cross ? sp.tr_sigma * 3.0f + 20.0f : c.cross_max_distance;
  if (ctx->so3_table.size() != 1000 || ctx->torus_table.size() != 5001)
    return fail(ctx, DDK_ERR_STATE, "score-norm tables not set (ddk_set_score_norm_tables)");
  {   // so3.score_norm (utils/so3.py:91-95), float32 index arithmetic like numpy on a float32 array
    float idx = (log10f(sp.rot_sigma) - (float)log10(0.01)) / (float)(log10(2.0) - log10(0.01)) * 1000.0f;
    long i = lrintf(idx);
    i = i < 0 ? 0 : (i > 999 ? 999 : i);
    sp.so3_norm = (float)ctx->so3_table[i];
  }
  {   // torus.score_norm (utils/torus.py:79-83)
    float s = logf(sp.tor_sigma / (float)M_PI);
    s = (s - (float)log(3e-3)) / (float)(log(2.0) - log(3e-3)) * 5000.0f;
    s = s < 0.f ? 0.f : (s > 5000.f ? 5000.f : s);
    const long i = lrintf(s);
    sp.torus_norm_sqrt = sqrtf((float)ctx->torus_table[i]);
  }
  return DDK_OK;
}

// ---- asynchronous upload machinery (see ddk_ctx / ddk_complex) -----------------------------------------------------------------
constexpr size_t CHUNK_POOL_MAX_CHUNKS = 1024;
constexpr size_t CHUNK_POOL_MAX_BYTES = (size_t)96 << 30;   // device memory parked in the pool (288 GB of HBM per GPU); beyond it hipFree (device sync)

size_t pool_evict_idle(ddk_ctx* ctx, bool wait) {
  size_t freed = 0;
  for (size_t i = 0; i < ctx->chunk_pool.size();) {
    ddk_ctx::PoolChunk& c = ctx->chunk_pool[i];
    const bool idle = !c.free_after || hipEventQuery(c.free_after) == hipSuccess;
    if (!idle && !wait) { ++i; continue; }
    if (!idle) hipEventSynchronize(c.free_after);      // (hipFree does not wait for work on the context's non-blocking streams)
    if (c.free_after) hipEventDestroy(c.free_after);
    ctx_free(ctx, c.p);
    freed += c.cap;
    ctx->chunk_pool_bytes -= c.cap;
    ctx->pool_frees++;
    ctx->chunk_pool.erase(ctx->chunk_pool.begin() + i);
  }
  return freed;
}

void* cx_new_chunk(ddk_complex* cx, size_t cap) {
  ddk_ctx* ctx = cx->owner;
  {   // size classes {2^k, 1.5 * 2^k}: complexes of similar size take each other's chunks (ligands of 20-40 atoms differ by +-30 %)
    size_t c2 = (size_t)1 << 20;
    while (c2 < cap) c2 = (c2 & (c2 - 1)) ? (c2 / 3) * 4 : c2 + c2 / 2;
    cap = c2;
  }
  int best = -1;
  for (int i = 0; i < (int)ctx->chunk_pool.size(); ++i) {
    const auto& c = ctx->chunk_pool[i];
    // only chunks whose previous owner has finished: a chunk that is still read by a loop in flight would make the staged copies (and with
    // them the pinned staging buffers) wait for that loop - a fresh hipMalloc is cheap and HBM is plentiful
    if (c.free_after && hipEventQuery(c.free_after) != hipSuccess) continue;
    if (c.cap >= cap && c.cap <= 2 * cap && (best < 0 || c.cap < ctx->chunk_pool[best].cap)) best = i;
  }
  void* p = nullptr;
  if (best >= 0) {
    ddk_ctx::PoolChunk c = ctx->chunk_pool[best];
    ctx->chunk_pool.erase(ctx->chunk_pool.begin() + best);
    ctx->chunk_pool_bytes -= c.cap;
    if (c.free_after) hipEventDestroy(c.free_after);      // (complete: checked above)
    p = c.p; cap = c.cap;
    ctx->pool_reuses++;
  } else {
    if (ctx_malloc(ctx, &p, cap) != hipSuccess) {
      // memory pressure (or the debug limit): what is parked in the pool is this context's own slack - hand the idle chunks back to the driver and try
      // once more, waiting for chunks whose last user is still running only if that is what it takes
      if (pool_evict_idle(ctx, false) == 0 || ctx_malloc(ctx, &p, cap) != hipSuccess)
        if (pool_evict_idle(ctx, true) == 0 || ctx_malloc(ctx, &p, cap) != hipSuccess) {
          ctx->err = "out of device memory: a complex asked for a chunk of " + std::to_string(cap) + " B (context holds " + std::to_string(ctx->dev_bytes) + " B" +
                     (ctx->alloc_limit > 0 ? ", debug limit " + std::to_string(ctx->alloc_limit) + " B" : std::string()) + "); retry with a smaller batch";
          return nullptr;
        }
    }
    ctx->pool_mallocs++;
  }
  ctx->pool_bytes_out += (int64_t)cap;
  if (ctx->pool_bytes_out > ctx->pool_bytes_out_peak) ctx->pool_bytes_out_peak = ctx->pool_bytes_out;
  cx->allocs.push_back({p, cap});
  cx->chunk_cap = cap;
  return p;
}

int cx_stage_begin(ddk_ctx* ctx, ddk_complex* cx, size_t bytes) {
  bytes = (bytes + 4095) & ~(size_t)4095;
  int idx = -1;
  for (int i = 0; i < (int)ctx->stage_pool.size(); ++i) {
    auto& b = ctx->stage_pool[i];
    if (b.in_flight && hipEventQuery(b.done) == hipSuccess) b.in_flight = false;
    if (!b.in_flight && b.cap >= bytes && (idx < 0 || b.cap < ctx->stage_pool[idx].cap)) idx = i;
  }
  if (idx < 0) {
    ddk_ctx::StageBuf b;
    b.cap = bytes < ((size_t)8 << 20) ? ((size_t)8 << 20) : bytes;      // (pinned allocations are slow while the GPU is busy: few, large, reused)
    if (hipHostMalloc((void**)&b.p, b.cap) != hipSuccess) return fail(ctx, DDK_ERR_NOMEM, "hipHostMalloc failed (staging buffer)");
    if (hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) return fail(ctx, DDK_ERR_HIP, "event create failed");
    ctx->stage_pool.push_back(b);
    idx = (int)ctx->stage_pool.size() - 1;
  }
  ctx->stage_pool[idx].in_flight = true;       // reserved for this session (released by the event of cx_stage_flush)
  cx->stage_idx = idx; cx->stage_off = 0; cx->pending.clear();
  return DDK_OK;
}

bool cx_put(ddk_complex* cx, void* dst, const void* src, size_t bytes) {
  if (!dst || !bytes) return dst != nullptr;
  ddk_ctx* ctx = cx->owner;
  if (cx->stage_idx >= 0) {
    auto& b = ctx->stage_pool[cx->stage_idx];
    const size_t off = (cx->stage_off + 255) & ~(size_t)255;
    if (off + bytes <= b.cap) {
      memcpy(b.p + off, src, bytes);
      cx->stage_off = off + bytes;
      if (!cx->pending.empty()) {      // bump allocation on both sides: neighbours stay neighbours -> one DMA for a run of arrays
        auto& l = cx->pending.back();
        const size_t gap = off - l.off;
        if ((char*)l.dst + gap == (char*)dst && gap >= l.bytes && gap - l.bytes < 256) { l.bytes = gap + bytes; return true; }
      }
      cx->pending.push_back({dst, off, bytes});
      return true;
    }
  }
  // no session / staging buffer full: ordered behind the staged copies on the upload stream, synchronous for the caller's memory
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->up_stream) != hipSuccess) return false;
  return hipStreamSynchronize(ctx->up_stream) == hipSuccess;
}

int cx_stage_flush(ddk_ctx* ctx, ddk_complex* cx) {
  hipError_t e = hipSuccess;
  if (cx->stage_idx >= 0) {
    auto& b = ctx->stage_pool[cx->stage_idx];
    for (const auto& r : cx->pending)
      if (e == hipSuccess) e = hipMemcpyAsync(r.dst, b.p + r.off, r.bytes, hipMemcpyHostToDevice, ctx->up_stream);
    if (e == hipSuccess) e = hipEventRecord(b.done, ctx->up_stream);
    cx->pending.clear();
    cx->stage_idx = -1;
  }
  if (e == hipSuccess && !cx->ready) e = hipEventCreateWithFlags(&cx->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventRecord(cx->ready, ctx->up_stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "complex upload");
  cx->ready_pending = true;
  return DDK_OK;
}

hipError_t cx_wait_ready(ddk_complex* cx, hipStream_t s) {
  cx->last_stream = s; cx->used = true;
  if (std::find(cx->streams.begin(), cx->streams.end(), s) == cx->streams.end()) cx->streams.push_back(s);
  if (!cx->ready_pending) return hipSuccess;
  if (hipEventQuery(cx->ready) == hipSuccess) { cx->ready_pending = false; return hipSuccess; }
  return hipStreamWaitEvent(s, cx->ready, 0);
}

template <typename T>
static T* cx_upload(ddk_complex* cx, const T* src, size_t n) {
  T* p = (T*)cx_alloc(cx, n * sizeof(T));
  if (!p) return nullptr;
  if (n && src && !cx_put(cx, p, src, n * sizeof(T))) return nullptr;
  return p;
}

// a6-a8 + the merge of score_model.py:218-225: counts, offsets and the sorted edge list of B poses into the complex' workspace
static hipError_t build_graph(ddk_ctx* ctx, ddk_complex* cx, int B, const float* lig_pos, float cross_cutoff, bool prune, bool shared_rr,
                              hipStream_t s, int64_t patch_off = -1, int cross_mirror = 0) {
  const ddk_config& c = ctx->cfg;
  GraphArgs G;
  G.cross_mirror = cross_mirror;
  G.lig_pos = lig_pos; G.rec_pos = cx->rec_pos; G.bond_src = cx->bond_src; G.bond_dst = cx->bond_dst;
  G.rr_src = cx->rr_src; G.rr_dst = cx->rr_dst; G.rr_outdeg = cx->rr_outdeg; G.rr_start = cx->rr_start;
  G.prune = prune ? 1 : 0; G.shared_rr = shared_rr ? 1 : 0; G.patch_off = patch_off;
  G.B = B; G.n_lig = cx->n_lig; G.n_rec = cx->n_rec; G.M = cx->M; G.E_rr = cx->E_rr;
  G.lig_r2 = c.lig_max_radius * c.lig_max_radius; G.cross_cutoff = cross_cutoff;
  G.counts = cx->counts; G.offs = cx->offs; G.info = cx->info; G.levels = cx->levels; G.e_src = cx->e_src; G.e_dst = cx->e_dst; G.e_aux = cx->e_aux;
  G.deg = cx->deg;
  return launch_graph(G, cx->edge_cap, s);
}

// defer_post != null: the launch that finishes the scores (heads_post) is left to the caller, who hands *defer_post to the SE(3) update's launch
static int score_forward_impl(ddk_ctx* ctx, ddk_complex* cx, int B, const float* lig_pos, const StepParams& sp,
                              float* tr_out, float* rot_out, float* tor_out, hipStream_t s, HeadArgs* defer_post = nullptr) {
  const ddk_config& c = ctx->cfg;
  Model* M = (Model*)ctx->model;
  const int n_lig = cx->n_lig, n_rec = cx->n_rec;
  const int64_t N = (int64_t)B * (n_lig + n_rec);
  hipError_t e;
#define CK(x, what) do { e = (x); if (e != hipSuccess) return hip_fail(ctx, e, what); } while (0)
  // layer 0: the receptor's node features and rec-rec edge features are the same for every sample of the batch
  // (no latents) -> evaluate the rec-rec messages once, exact in real arithmetic
  // Latent-conditioned (DisCo) model: the latents are one-hot at a few nodes per sample, so the shared pass (on sample 0's rows) is right
  // for every receiver whose senders and itself carry zero latents in sample s and in sample 0; all the other receivers get their
  // rec-rec messages per sample from a fifth "patch" edge group (disco_patch kernels, rebuilt when the latents change)
  const bool patched = c.latent_dim > 0 && ctx->layer0_dedup && !c.deterministic && cx->patch_off >= 0 && cx->pre != nullptr &&
                       ctx->conv[0].wn != nullptr && c.num_conv_layers > 1;
  const bool dedup = ((c.latent_dim == 0 && ctx->layer0_dedup) || patched) && B > 1 && cx->E_rr > 0;
  if (c.latent_dim > 0 && (!cx->lig_latent || !cx->rec_latent))
    return fail(ctx, DDK_ERR_STATE, "latent-conditioned model: call ddk_set_latents before the forward");
  if (dedup && patched && (cx->latent_dirty || cx->patch_B != B)) {
    PatchArgs P;
    P.rec_latent = cx->rec_latent; P.rr_start = cx->rr_start; P.rr_outdeg = cx->rr_outdeg; P.rr_dst = cx->rr_dst;
    P.B = B; P.n_lig = n_lig; P.n_rec = n_rec; P.E_rr = cx->E_rr; P.latent_dim = c.latent_dim; P.patch_off = cx->patch_off;
    P.rr_mask = cx->rr_mask; P.patch_cnt = cx->patch_cnt; P.info = cx->info; P.e_src = cx->e_src; P.e_dst = cx->e_dst; P.e_aux = cx->e_aux;
    CK(launch_disco_patch(P, s), "layer-0 patch group");
    cx->latent_dirty = false; cx->patch_B = B;
  }
  // backward receptive-field pruning of the rec-rec messages (k_graph.hip): off when the caller wants the receptor rows of the last layer
  const bool prune = ctx->prune && !cx->keep_rec && cx->E_rr > 0;
  // the flipped cross edges carry the same embedding and sh as the lig->rec edges: evaluated once per pair when the fill kernel's LDS holds the pair matrix
  const int mirror = graph_cross_mirror_fits(n_lig, n_rec);
  CK(build_graph(ctx, cx, B, lig_pos, sp.cross_cutoff, prune, dedup, s, (dedup && patched) ? cx->patch_off : -1, mirror), "graph build");
  EdgeFeatArgs F;
  F.cross_mirror = mirror;
  F.lig_pos = lig_pos; F.rec_pos = cx->rec_pos; F.bond_attr = cx->bond_attr; F.rr_pre1 = cx->rr_pre1; F.rr_sh = cx->rr_sh;
  F.e_src = cx->e_src; F.e_dst = cx->e_dst; F.e_aux = cx->e_aux; F.info = cx->info; F.e_emb = cx->e_emb; F.e_sh = cx->e_sh;
  F.lig = M->dev.lig_edge; F.rec = M->dev.rec_edge; F.cross = M->dev.cross_edge; F.sp = sp;
  F.n_lig_total = B * n_lig; F.n_rec = n_rec; F.n_shared = dedup ? cx->E_rr : 0;
  // 5-layer model with the layer-0 de-duplication and the pruning on: layers 1..3 evaluate the rec-rec messages of levels C, B, A only, layer 0 the
  // shared copy, layer 4 none -> the per-sample rec-rec edges behind the level-C segment never need their embedding / sh
  F.g2_live_only = (dedup && prune && c.num_conv_layers == 5) ? 1 : 0;
  F.patch_off = (dedup && patched) ? cx->patch_off : -1;
  F.latent_dim = c.latent_dim; F.lig_latent = cx->lig_latent; F.rec_latent = cx->rec_latent; F.unconditional = cx->unconditional;
  // worst-case edge count of THIS batch size bounds the launch
  const int64_t cap_b = (int64_t)B * ((int64_t)cx->M + (int64_t)n_lig * LIG_CAP + 2LL * n_lig * n_rec + cx->E_rr) + cx->E_rr;
  const int64_t feat_cap = (cap_b < cx->edge_cap ? cap_b : cx->edge_cap) + (F.patch_off >= 0 ? (int64_t)B * cx->E_rr + 256 : 0);
  float* xin = cx->xa;
  float* xout = cx->xb;
  NodeEmbedArgs NE_;
  NE_.lig_static = cx->lig_node_static; NE_.rec_static = cx->rec_node_static; NE_.sp = sp; NE_.B = B; NE_.n_lig = n_lig; NE_.n_rec = n_rec;
  NE_.x = xin; NE_.lig_latent = cx->lig_latent; NE_.rec_latent = cx->rec_latent; NE_.lig_w_lat = M->dev.lig_w_lat; NE_.rec_w_lat = M->dev.rec_w_lat;
  NE_.lig_unc = M->dev.lig_node_unc; NE_.rec_unc = M->dev.rec_node_unc; NE_.unconditional = cx->unconditional; NE_.latent_dim = c.latent_dim;
  // per-node terms of layer 0's GEMM1 (ConvLayerDev::wn): one launch with the node embedding itself
  const bool split = ctx->conv[0].wn != nullptr && cx->pre != nullptr;
  if (split) {
    NodePreArgs PA = {};
    PA.x_out = xin; PA.n_lig_total = B * n_lig; PA.n_rec_total = B * n_rec; PA.n_rec = n_rec;
    PA.wn = ctx->conv[0].wn; PA.bnp = ctx->conv[0].bnp; PA.pre = cx->pre; PA.n_slots = 1; PA.lig_roles = 15; PA.rec_roles = 15;
    // ONE launch: the edge features and, beside them, the node embedding + layer 0's node terms (they depend on t and the latents only)
    CK(launch_edge_features_node(F, feat_cap, PA, NE_, s), "edge features + node embed + node_pre");
  } else {
    CK(launch_edge_features(F, feat_cap, s), "edge features");
    CK(launch_node_embed(NE_, s), "node embed");
  }
  // accumulators: node_finalize zeroes what it reads, so a forward that ran to its end leaves them clean for the next one
  if (!cx->sum_clean) {
    CK(hipMemsetAsync(cx->sum, 0, (size_t)cx->max_batch * (n_lig + n_rec) * XW * sizeof(float) * (c.deterministic ? 2 : 1), s), "memset sum");
    CK(hipMemsetAsync(cx->sum_rr0, 0, (size_t)n_rec * XW * sizeof(float), s), "memset sum_rr0");
  }
  cx->sum_clean = false;
  bool rr0_dirty = false;
  const int NL = c.num_conv_layers;
  int32_t* prof_slot = nullptr;
  if (ctx->prof && ctx->prof_slots + 1 <= ctx->prof_cap) prof_slot = ctx->prof_edges + (size_t)PROF_INTS * ctx->prof_slots;
  for (int l = 0; l < NL; ++l) {
    const ConvLayerDev& L = ctx->conv[l];
    ConvLaunch a;
    a.x = xin; a.src = cx->e_src; a.dst = cx->e_dst; a.edge_attr = cx->e_emb; a.sh = cx->e_sh; a.sum = cx->sum;
    a.tile_info = cx->info; a.counter = cx->info + I_CNT + (l % 8); a.gather = 1;
    a.pre = split ? cx->pre : nullptr;
    // which rec-rec messages this layer evaluates: layer 0 the shared copy (de-duplication); layers L-2, L-3, L-4 only those received by
    // the residues inside the backward receptive field of the heads (levels A, B, C of k_graph.hip); the last layer none (and no
    // rec->lig... group 3 either): only ligand rows are read downstream unless the caller asked for the receptor rows
    const bool lig_only = (l == NL - 1 && !cx->keep_rec);
    const bool shared0 = dedup && l == 0;
    int tab = TAB_ALL;
    if (shared0) tab = TAB_SHARED;
    else if (prune && l == NL - 2) tab = TAB_A;
    else if (prune && l == NL - 3) tab = TAB_B;
    else if (prune && l == NL - 4) tab = TAB_C;
    a.gbeg = cx->info + I_TAB + 8 * tab; a.gend = a.gbeg + 4;
    a.n_groups = 4; a.n_active = lig_only ? 2 : 4; a.n_slots = 1; a.slots = 0;
    if (shared0 && patched) {      // [ll | lr | shared rr | rl | per-sample patches (rec-rec weights and node-term roles)]
      a.gbeg = cx->info + I_TABX; a.gend = a.gbeg + 5; a.n_groups = 5; a.n_active = 5; a.wmap = 0x23210ull;
    }
    if (c.deterministic) {     // ligand atoms receive in groups 0,1, residues in 2,3: slot = g & 1 -> one writer per (node, slot, channel)
      a.n_slots = 2; a.slots = (0u) | (1u << 2) | (0u << 4) | (1u << 6);
      a.part = cx->part; a.edge_bound = cap_b;
      // sample-aligned work units: where the 32-edge tiles cut a node's run depends on the sample's own edges only (not on the batch around it)
      const int kind = lig_only ? 0 : (shared0 ? 5 : (tab == TAB_A ? 1 : (tab == TAB_B ? 2 : (tab == TAB_C ? 3 : 4))));
      GraphArgs Gd = {};
      Gd.counts = cx->counts; Gd.B = B; Gd.M = cx->M; Gd.E_rr = cx->E_rr;
      CK(launch_det_ranges(Gd, kind, 0, cx->det_rng, s), "deterministic ranges");
      a.det_rng = cx->det_rng; a.det_nr = det_ranges_count(kind, B);
    }
    if (shared0) {
      rr0_dirty = true;
      a.sum_g2 = cx->sum_rr0; a.g2_node_off = B * n_lig;
    }
    if (l >= 8) CK(hipMemsetAsync(cx->info + I_CNT + (l % 8), 0, sizeof(int32_t), s), "counter reset");
    ddk_ctx::ProfRec pr;
    if (prof_slot) {
      CK(hipEventCreate(&pr.a), "event"); CK(hipEventCreate(&pr.b), "event");
      pr.layer = l; pr.slot = ctx->prof_slots; pr.tab = tab; pr.lig_only = lig_only; pr.r01_skipped = shared0 ? (patched ? -1 : (int64_t)(B - 1) * cx->E_rr) : 0;      // (-1: E - executed edges, the patch count lives on the device)
      CK(hipEventRecord(pr.a, s), "event record");
    }
    a.use_y = ctx->cfg.conv_kernel == 2;
    if (ctx->conv_trace != nullptr && ctx->conv_trace_layer == l) { a.trace = ctx->conv_trace; a.trace_coarse = ctx->conv_trace_coarse; }
#if defined(DDK_ABL_CONCURRENT_LAYERS)
#ifndef DDK_TIMING_ONLY_BUILD
#error "DDK_ABL_CONCURRENT_LAYERS gives WRONG RESULTS (timing-only ablation): it needs -DDDK_TIMING_ONLY_BUILD as well (tools/build_variant_model.sh adds it)"
#endif
    {   // TIMING ONLY (results invalid): the five conv launches of a forward on five streams, none waiting for the finalize before it - how long the forward's
        // conv work takes when the launches fill each other's ramps and tails: the upper bound of what ONE persistent launch over the layers could save
      static hipStream_t abl_s[8] = {};
      static hipEvent_t abl_fork = nullptr, abl_join[8] = {};
      if (!abl_fork) {
        hipEventCreateWithFlags(&abl_fork, hipEventDisableTiming);
        for (int k = 0; k < 8; ++k) { hipStreamCreateWithFlags(&abl_s[k], hipStreamNonBlocking); hipEventCreateWithFlags(&abl_join[k], hipEventDisableTiming); }
      }
      if (l == 0) {
        CK(hipEventRecord(abl_fork, s), "abl fork");
        for (int k = 0; k < NL; ++k) CK(hipStreamWaitEvent(abl_s[k], abl_fork, 0), "abl fork");
      }
      CK(launch_conv_fused(L, a, ctx->n_cu, abl_s[l]), "conv_fused (ablation)");
      CK(hipEventRecord(abl_join[l], abl_s[l]), "abl join");
      if (l == NL - 1)
        for (int k = 0; k < NL; ++k) CK(hipStreamWaitEvent(s, abl_join[k], 0), "abl join");
    }
#else
    CK(launch_conv_fused(L, a, ctx->n_cu, s), "conv_fused");
#endif
    if (prof_slot) {
      CK(hipEventRecord(pr.b, s), "event record");
      ctx->prof_recs.push_back(pr);
    }
    const bool clear_rr0 = rr0_dirty && !shared0;     // one launch after the layer whose finalize read the shared rows
    if (split && l + 1 < NL) {     // finalize fused with the node terms of the next layer's GEMM1
      NodePreArgs PA = {};
      PA.sum = cx->sum; PA.deg = cx->deg; PA.x_in = xin; PA.bn_mean = L.bn_mean; PA.bn_scale = L.bn_scale; PA.bn_bias = L.bn_bias;
      PA.dout = L.dout; PA.x_out = xout; PA.sum_rr0 = shared0 ? cx->sum_rr0 : nullptr; PA.n_lig_total = B * n_lig; PA.n_rec_total = B * n_rec;
      PA.rr0_mask = (shared0 && patched) ? cx->rr_mask : nullptr;
      PA.n_rec = n_rec; PA.zero_extra = clear_rr0 ? cx->sum_rr0 : nullptr; PA.n_extra = clear_rr0 ? (int64_t)n_rec * XW : 0;
      PA.wn = ctx->conv[l + 1].wn; PA.bnp = ctx->conv[l + 1].bnp; PA.pre = cx->pre; PA.n_slots = c.deterministic ? 2 : 1;
      // rows nothing downstream reads (the same receptive-field argument as for the rec-rec messages) are neither finalised nor given node terms
      // the last layer evaluates groups 0 and 1 only (unless the receptor rows were asked for): ligand atoms receive in both and send in group 0,
      // residues only send in group 1 -> 3 of 4 resp. 1 of 4 role slots
      const bool next_lig_only = (l + 1 == NL - 1) && !cx->keep_rec;
      PA.lig_roles = next_lig_only ? 0x7 : 0xF; PA.rec_roles = next_lig_only ? 0x4 : 0xF;
      PA.levels = prune ? cx->levels : nullptr; PA.max_level = l == NL - 2 ? 0 : (l == NL - 3 ? 1 : (l == NL - 4 ? 2 : 3));
      CK(launch_node_finalize_pre(PA, true, s), "node_finalize_pre");
    } else
    CK(launch_node_finalize(cx->sum, cx->deg, xin, L.bn_mean, L.bn_scale, L.bn_bias, lig_only ? (int64_t)B * n_lig : N, L.dout, XW, xout, s,
                            shared0 ? cx->sum_rr0 : nullptr, (int64_t)B * n_lig, n_rec, 1, clear_rr0 ? cx->sum_rr0 : nullptr,
                            clear_rr0 ? (int64_t)n_rec * XW : 0, c.deterministic ? 2 : 1, (shared0 && patched) ? cx->rr_mask : nullptr), "node_finalize");
    if (clear_rr0) rr0_dirty = false;
    float* t = xin; xin = xout; xout = t;
  }
  cx->x_last = xin;
  cx->last_B = B;
  cx->last_full = cx->keep_rec;
  cx->sum_clean = !rr0_dirty;       // (a one-layer model leaves the shared rows to the memset of the next forward)
  if (M->confidence_mode) return DDK_OK;      // no score heads: ddk_score_confidence pools the ligand rows
  // heads: both are tensor-product convolutions -> the fused conv kernel on their own small edge lists (k_heads.hip)
  const bool torsion = !c.no_torsion && tor_out != nullptr && cx->R > 0;
  HeadArgs Hd;
  Hd.lig_pos = lig_pos; Hd.x = xin; Hd.md = M->dev; Hd.sp = sp; Hd.B = B; Hd.n_lig = n_lig; Hd.R = cx->R;
  Hd.scale_by_sigma = c.scale_by_sigma; Hd.rot_u = cx->rot_u; Hd.rot_v = cx->rot_v; Hd.lig_r2 = c.lig_max_radius * c.lig_max_radius;
  Hd.tr_out = tr_out; Hd.rot_out = rot_out; Hd.tor_out = tor_out;
  Hd.h_src = cx->h_src; Hd.h_dst = cx->h_dst; Hd.h_deg = cx->h_deg; Hd.h_info = cx->info + I_HEAD; Hd.h_attr = cx->h_attr; Hd.h_sh = cx->h_sh;
  Hd.h_sum = cx->h_sum; Hd.deterministic = c.deterministic;
  if (prof_slot) {    // E, edges of groups 0+1 and the edges each group table evaluates, of THIS forward: written by heads_post_kernel into the
    Hd.prof_out = prof_slot; Hd.exec_info = cx->info + I_EXEC;      // pinned slot (read at ddk_profile_read), no copy launch of its own
    ctx->prof_slots += 1;
  }
  CK(launch_heads_pre(Hd, torsion, s), "heads_pre");
  // final_conv on the context's side stream beside tor_bond_conv on the caller's stream (disjoint accumulator rows and work queues)
  // deterministic mode, final_conv: the centre edges of a sample are ONE sample-aligned range (the bond edges of tor_bond_conv sit in fixed 32-edge slots
  // per bond already); built on the caller's stream before the fork
  int32_t* head_rng = nullptr;
  if (c.deterministic) {
    head_rng = cx->det_rng + (4 * (7 * (size_t)cx->max_batch + 1) + 8);
    GraphArgs Gd = {};
    Gd.B = B;
    CK(launch_det_ranges(Gd, 6, n_lig, head_rng, s), "deterministic ranges (head)");
  }
  if (torsion) {
    CK(hipEventRecord(ctx->ev_fork, s), "head fork");
    CK(hipStreamWaitEvent(ctx->head_stream, ctx->ev_fork, 0), "head fork");
  }
  for (int hd = 1; hd >= (torsion ? 0 : 1); --hd) {       // final_conv (centre), then tor_bond_conv
    ConvLaunch a;
    a.x = xin; a.src = cx->h_src; a.dst = cx->h_dst; a.edge_attr = cx->h_attr; a.sh = cx->h_sh; a.sum = cx->h_sum;
    a.tile_info = cx->info; a.counter = cx->info + I_HEAD + 4 + hd; a.gather = 0; a.pre = nullptr;
    a.n_groups = 1; a.n_active = 1; a.n_slots = 1; a.slots = 0;
    a.gbeg = cx->info + I_HEAD + (hd == 1 ? 0 : 2); a.gend = a.gbeg + 1;
    if (c.deterministic) {     // (the two head launches run side by side: each its own part of the partial-row buffer)
      const int64_t eh = (int64_t)B * (n_lig + (int64_t)cx->R * BOND_CAP);
      a.part = cx->part + (hd == 1 ? 0 : ((size_t)B + 2) * CONV_WAVES * 2 * XW); a.edge_bound = eh;
      if (hd == 1) { a.det_rng = head_rng; a.det_nr = B; }
    }
    CK(launch_conv_fused(ctx->head[hd], a, ctx->n_cu, (hd == 1 && torsion) ? ctx->head_stream : s), "conv_fused (head)");
  }
  if (torsion) {
    CK(hipEventRecord(ctx->ev_join, ctx->head_stream), "head join");
    CK(hipStreamWaitEvent(s, ctx->ev_join, 0), "head join");
  }
  if (defer_post) *defer_post = Hd;
  else CK(launch_heads_post(Hd, torsion, s), "heads_post");
#undef CK
  return DDK_OK;
}

}  // namespace ddk

using namespace ddk;

static int check_model(ddk_ctx* ctx, ddk_complex* cx, int B) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  if (!ctx->finalized || !ctx->model || !((Model*)ctx->model)->host.ready)
    return fail(ctx, DDK_ERR_STATE, "score model weights not loaded / finalised");
  if (!cx) return fail(ctx, DDK_ERR_INVALID, "null complex");
  if (B < 1 || B > cx->max_batch) return fail(ctx, DDK_ERR_INVALID, "batch size exceeds the complex's max_batch");
  return DDK_OK;
}

extern "C" {

int ddk_complex_create(ddk_ctx* ctx, const ddk_complex_desc* d, int32_t max_batch, ddk_complex** out) {
  if (!ctx || !d || !out) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot hold device data");
  if (!ctx->finalized) return fail(ctx, DDK_ERR_STATE, "ddk_finalize_weights has not run");
  const ddk_config& c = ctx->cfg;
  // without a loaded score model the complex carries topology only (enough for ddk_se3_update)
  const bool has_model = ctx->model != nullptr;
  static const ModelHost no_model;
  const ModelHost& H = has_model ? ((Model*)ctx->model)->host : no_model;
  if (d->n_lig < 1 || d->n_lig > MAX_LIG) return fail(ctx, DDK_ERR_INVALID, "n_lig must be in [1, 256]");
  if (d->n_rec < 1 || d->n_rec > ctx->max_rec)
    return fail(ctx, DDK_ERR_INVALID, "n_rec must be in [1, " + std::to_string(ctx->max_rec) + "]: the graph kernels keep a sample's receptor (17 B per residue) in LDS");
  if (has_model && d->rec_feat_dim != 1 + c.lm_embedding_dim) return fail(ctx, DDK_ERR_INVALID, "receptor feature width != 1 + lm_embedding_dim");
  if (max_batch < 1) return fail(ctx, DDK_ERR_INVALID, "max_batch < 1");
  hipSetDevice(c.device);
  ddk_complex* cx = new ddk_complex();
  cx->owner = ctx;
  *out = cx;
  cx->n_lig = d->n_lig; cx->n_rec = d->n_rec; cx->M = d->n_bond_edges; cx->R = d->n_rot; cx->E_rr = d->n_rec_edges;
  cx->max_batch = max_batch;
  const int n_lig = d->n_lig, n_rec = d->n_rec, M = d->n_bond_edges, lm = c.lm_embedding_dim;
  {   // one chunk for everything this function allocates (sizes below mirror the uploads / workspaces; 256 B of slack per array)
    const size_t E0 = (size_t)d->n_rec_edges, Bm0 = (size_t)max_batch, R0 = (size_t)(d->n_rot > 0 ? d->n_rot : 1);
    const size_t cap0 = Bm0 * ((size_t)M + (size_t)n_lig * LIG_CAP + 2 * (size_t)n_lig * n_rec + E0) + E0 + 64, N0 = Bm0 * (size_t)(n_lig + n_rec);
    size_t need = (size_t)M * 24 + R0 * 8 + R0 * n_lig + (size_t)n_rec * 12 + (size_t)(n_lig + n_rec) * NS * 4 + E0 * (8 + NS * 4 + 16) + (size_t)n_rec * 4 +
                  (size_t)n_rec * d->rec_feat_dim * 4;
    need += cap0 * (12 + NS * 4 + 16) + N0 * 4 + N0 * PRE_W * 4 + (c.deterministic ? N0 * XW * 4 + (cap0 / 32 + 128 + 8 * (7 * Bm0 + 8)) * 2 * XW * 4 + 64 * (7 * Bm0 + 8) : 0) + Bm0 * ((size_t)n_lig + R0 * BOND_CAP + 2) * (8 + NE * 4 + 16) + Bm0 * (1 + R0) * (XW * 4 + 4) + 8 * 256 + Bm0 * 2 * CNT_STRIDE * 4 + INFO_INTS * 4 + (size_t)n_rec * 4 + Bm0 * n_rec + 3 * N0 * XW * 4 + (size_t)n_rec * XW * 4 + Bm0 * n_lig * 12 + 2 * Bm0 * (6 + R0) * 4;
    if (c.latent_dim > 0) need += N0 * c.latent_dim * 4 + Bm0 * E0 * (12 + NS * 4 + 16) + Bm0 * n_rec + (Bm0 + 1) * 4 + 3 * 256;
    cx_reserve(cx, need + 64 * 256);
    // everything uploaded below (topology, static embeddings, receptor-edge geometry) goes through one pinned staging buffer
    const size_t staged = (size_t)M * 24 + R0 * 8 + R0 * n_lig + (size_t)n_rec * 12 + (size_t)n_lig * NS * 4 + E0 * (8 + 16) +
                          (size_t)n_rec * 8 + (size_t)n_rec * d->rec_feat_dim * 4 + 24 * 256;
    int rc0 = cx_stage_begin(ctx, cx, staged);
    if (rc0) return rc0;
  }
  // ---- topology ------------------------------------------------------------------------------
  std::vector<int32_t> ru, rv;
  for (int m = 0; m < M; ++m) {
    const int a = d->bond_index[m], b = d->bond_index[M + m];
    if (a < 0 || a >= n_lig || b < 0 || b >= n_lig) return fail(ctx, DDK_ERR_INVALID, "bond index out of range");
    if (d->edge_mask[m]) { ru.push_back(a); rv.push_back(b); }
  }
  if ((int)ru.size() != d->n_rot) return fail(ctx, DDK_ERR_INVALID, "edge_mask.sum() != n_rot");
  for (int r = 0; r < d->n_rot; ++r)   // orientation asserted by utils/torsion.py:77-78
    if (d->mask_rotate[(size_t)r * n_lig + ru[r]] || !d->mask_rotate[(size_t)r * n_lig + rv[r]])
      return fail(ctx, DDK_ERR_INVALID, "mask_rotate orientation: u must be fixed and v rotating for every rotatable bond");
  cx->bond_src = cx_upload(cx, d->bond_index, (size_t)M);
  cx->bond_dst = cx_upload(cx, d->bond_index + M, (size_t)M);
  cx->bond_attr = cx_upload(cx, d->bond_attr, (size_t)M * 4);
  cx->rot_u = cx_upload(cx, ru.data(), ru.size());
  cx->rot_v = cx_upload(cx, rv.data(), rv.size());
  cx->mask_rotate = cx_upload(cx, d->mask_rotate, (size_t)d->n_rot * n_lig);
  cx->rec_pos = cx_upload(cx, d->rec_pos, (size_t)n_rec * 3);
  // ---- static node embeddings (AtomEncoder without its sigma columns, models/layers.py:140-149) ----
  std::vector<float> ls((size_t)n_lig * NS);
  for (int i = 0; has_model && i < n_lig; ++i) {
    float emb[NS] = {0};
    for (int f = 0; f < 16; ++f) {
      const int v = d->lig_x[(size_t)i * 16 + f];
      if (v < 0 || v >= LIG_DIMS[f]) return fail(ctx, DDK_ERR_INVALID, "ligand categorical feature out of range");
      const float* row = H.lig_tables.data() + (size_t)(H.lig_table_off[f] + v) * NS;
      for (int k = 0; k < NS; ++k) emb[k] += row[k];
    }
    for (int o = 0; o < NS; ++o) {
      double a = H.lig_b[o];
      for (int k = 0; k < NS; ++k) a += (double)H.lig_w_emb[(size_t)o * NS + k] * emb[k];
      ls[(size_t)i * NS + o] = (float)a;
    }
  }
  for (int j = 0; has_model && j < n_rec; ++j) {
    const int res = (int)d->rec_x[(size_t)j * d->rec_feat_dim];
    if (res < 0 || res >= REC_DIM) return fail(ctx, DDK_ERR_INVALID, "residue id out of range");
  }
  // the receptor's language-model features go up as they are; the 1336-wide projection runs on the upload stream (rec_node_static_kernel)
  float* rec_x_dev = has_model ? cx_upload(cx, d->rec_x, (size_t)n_rec * d->rec_feat_dim) : nullptr;
  cx->lig_node_static = cx_upload(cx, ls.data(), ls.size());
  cx->rec_node_static = cx_upload<float>(cx, nullptr, (size_t)n_rec * NS);
  // ---- static receptor edges: geometry, SH and the distance half of rec_edge_embedding.0 --------
  const int E = d->n_rec_edges;
  std::vector<int32_t> outdeg(n_rec, 0);
  std::vector<float> sh((size_t)E * 4);
  for (int k = 0; k < E; ++k) {
    const int a = d->rec_edge_index[k], b = d->rec_edge_index[E + k];
    if (a < 0 || a >= n_rec || b < 0 || b >= n_rec) return fail(ctx, DDK_ERR_INVALID, "receptor edge index out of range");
    if (k > 0 && a < d->rec_edge_index[k - 1]) return fail(ctx, DDK_ERR_INVALID, "receptor edges must be grouped by source (process_mols.py:337-353 order)");
    outdeg[a]++;
    const float vx = d->rec_pos[3 * b] - d->rec_pos[3 * a], vy = d->rec_pos[3 * b + 1] - d->rec_pos[3 * a + 1],
                vz = d->rec_pos[3 * b + 2] - d->rec_pos[3 * a + 2];
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.7320508075688772f / fmaxf(dist, 1e-12f);
    sh[4 * (size_t)k] = 1.0f; sh[4 * (size_t)k + 1] = vx * inv; sh[4 * (size_t)k + 2] = vy * inv; sh[4 * (size_t)k + 3] = vz * inv;
  }
  cx->h_rr.assign(d->rec_edge_index, d->rec_edge_index + 2 * (size_t)E);
  cx->h_rec_pos.assign(d->rec_pos, d->rec_pos + 3 * (size_t)n_rec);
  cx->rr_src = cx_upload(cx, d->rec_edge_index, (size_t)E);
  cx->rr_dst = cx_upload(cx, d->rec_edge_index + E, (size_t)E);
  cx->rr_outdeg = cx_upload(cx, outdeg.data(), outdeg.size());
  std::vector<int32_t> rstart(n_rec, 0);
  for (int j = 1; j < n_rec; ++j) rstart[j] = rstart[j - 1] + outdeg[j - 1];
  cx->rr_start = cx_upload(cx, rstart.data(), rstart.size());
  cx->rr_pre1 = cx_upload<float>(cx, nullptr, (size_t)E * NS);      // filled by rec_edge_static_kernel behind the uploads
  cx->rr_sh = cx_upload(cx, sh.data(), sh.size());
  // ---- workspaces -------------------------------------------------------------------------------
  const int64_t Bm = max_batch;
  cx->edge_cap = Bm * ((int64_t)M + (int64_t)n_lig * LIG_CAP + 2LL * n_lig * n_rec + E) + E + 64;   // + the shared rec-rec copy
  if (cx->edge_cap >= ((int64_t)1 << 31)) return fail(ctx, DDK_ERR_INVALID, "edge capacity exceeds int32 (reduce max_batch)");
  const int64_t N = Bm * (n_lig + n_rec);
  // latent-conditioned model: room for the layer-0 patch group (at most every rec-rec edge of every sample) behind the regular edges
  const int64_t patch_cap = (has_model && c.latent_dim > 0 && E > 0) ? Bm * (int64_t)E : 0;
  if (cx->edge_cap + patch_cap >= ((int64_t)1 << 31)) return fail(ctx, DDK_ERR_INVALID, "edge capacity exceeds int32 (reduce max_batch)");
  const int64_t e_alloc = cx->edge_cap + patch_cap;
  cx->e_src = cx_upload<int32_t>(cx, nullptr, e_alloc);
  cx->e_dst = cx_upload<int32_t>(cx, nullptr, e_alloc);
  cx->e_aux = cx_upload<int32_t>(cx, nullptr, e_alloc);
  cx->e_emb = cx_upload<float>(cx, nullptr, e_alloc * NS);
  cx->e_sh = cx_upload<float>(cx, nullptr, e_alloc * 4);
  if (patch_cap > 0) {
    cx->patch_off = cx->edge_cap;
    cx->rr_mask = cx_upload<uint8_t>(cx, nullptr, Bm * n_rec);
    cx->patch_cnt = cx_upload<int32_t>(cx, nullptr, Bm + 1);
  }
  cx->deg = cx_upload<int32_t>(cx, nullptr, N);
  cx->counts = cx_upload<int32_t>(cx, nullptr, Bm * CNT_STRIDE);
  cx->offs = cx_upload<int32_t>(cx, nullptr, Bm * CNT_STRIDE);
  cx->info = cx_upload<int32_t>(cx, nullptr, INFO_INTS);
  cx->levels = cx_upload<uint8_t>(cx, nullptr, Bm * n_rec);
  cx->xa = cx_upload<float>(cx, nullptr, N * XW);
  cx->xb = cx_upload<float>(cx, nullptr, N * XW);
  const int det = c.deterministic ? 1 : 0;
  cx->sum = cx_upload<float>(cx, nullptr, N * XW * (det ? 2 : 1));      // deterministic: one accumulator per (node, receiving group)
  if (det) {      // partial rows of the runs that straddle 32-edge tiles (one block more per sample-aligned range: 7 per sample at most) + the ranges themselves
    cx->part = cx_upload<float>(cx, nullptr, (cx->edge_cap / CONV_BLOCK_EDGES + 16 + 7 * Bm + 8) * CONV_WAVES * 2 * XW);
    cx->det_rng = cx_upload<int32_t>(cx, nullptr, 2 * (4 * (7 * Bm + 1) + 8));      // [conv layers | final_conv head]
  }
  cx->sum_rr0 = cx_upload<float>(cx, nullptr, (int64_t)n_rec * XW);
  if (has_model) cx->pre = cx_upload<float>(cx, nullptr, N * PRE_W);
  if (has_model) {      // heads (k_heads.hip): edge list [B*n_lig centre edges | <= B*R*BOND_CAP bond-neighbour edges], accumulators [B | B*R] rows
    const int64_t Eh = Bm * ((int64_t)n_lig + (int64_t)(d->n_rot > 0 ? d->n_rot : 0) * BOND_CAP) + 64, Nh = Bm * (1 + (int64_t)(d->n_rot > 0 ? d->n_rot : 0)) + 1;   // (+ a scratch row for the deterministic mode's null edges)
    cx->h_src = cx_upload<int32_t>(cx, nullptr, Eh);
    cx->h_dst = cx_upload<int32_t>(cx, nullptr, Eh);
    cx->h_attr = cx_upload<float>(cx, nullptr, Eh * NE);
    cx->h_sh = cx_upload<float>(cx, nullptr, Eh * 4);
    cx->h_deg = cx_upload<int32_t>(cx, nullptr, Nh);
    cx->h_sum = cx_upload<float>(cx, nullptr, Nh * XW);
    if (cx->h_sum) launch_zero_fill(cx->h_sum, (size_t)Nh * XW * sizeof(float), ctx->up_stream);
  }
  cx->scores = cx_upload<float>(cx, nullptr, Bm * (6 + (d->n_rot > 0 ? d->n_rot : 1)));
  cx->scores2 = cx_upload<float>(cx, nullptr, Bm * (6 + (d->n_rot > 0 ? d->n_rot : 1)));
  if (c.latent_dim > 0) {
    cx->zero_lat = cx_upload<float>(cx, nullptr, N * c.latent_dim);
    if (cx->zero_lat) launch_zero_fill(cx->zero_lat, (size_t)N * c.latent_dim * sizeof(float), ctx->up_stream);
  }
  if (cx->oom || !cx->bond_src || !cx->rr_sh || !cx->scores)
    return fail(ctx, DDK_ERR_NOMEM, "ddk_complex_create (max_batch " + std::to_string(max_batch) + ", " + std::to_string(n_rec) + " residues): " +
                (ctx->err.rfind("out of device memory", 0) == 0 ? ctx->err : std::string("device allocation failed")));
  hipMemsetAsync(cx->info, 0, INFO_INTS * sizeof(int32_t), ctx->up_stream);
  // the accumulators start clean (node_finalize clears behind itself): cleared here, on the upload stream, beside the previous complex' loop
  if (cx->sum && cx->sum_rr0) {
    launch_zero_fill(cx->sum, (size_t)N * XW * sizeof(float) * (det ? 2 : 1), ctx->up_stream);
    launch_zero_fill(cx->sum_rr0, (size_t)n_rec * XW * sizeof(float), ctx->up_stream);
    cx->sum_clean = true;
  }
  int rcf = cx_stage_flush(ctx, cx);   // the copies are in flight on the upload stream; every launch entry point waits for cx->ready
  if (rcf || !has_model) return rcf;
  {   // static precompute behind the copies, on the upload stream; `ready` moves behind it
    const ModelDev& MD = ((Model*)ctx->model)->dev;
    RecStaticArgs RS;
    RS.rec_x = rec_x_dev; RS.n_rec = n_rec; RS.feat_dim = d->rec_feat_dim; RS.lm = lm; RS.rec_table = MD.rec_table; RS.w_emb = MD.rec_w_emb; RS.w_lm_t = MD.rec_w_lm_t;
    RS.b = MD.rec_b; RS.out = cx->rec_node_static;
    hipError_t e = launch_complex_static(RS, cx->rr_src, cx->rr_dst, cx->rec_pos, E, MD.rec_edge, cx->rr_pre1, ctx->up_stream);
    if (e == hipSuccess) e = hipEventRecord(cx->ready, ctx->up_stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "static precompute");
  }
  return DDK_OK;
}

void ddk_complex_destroy(ddk_ctx* ctx, ddk_complex* cx) {
  if (!cx) return;
  if (!ctx) ctx = cx->owner;
  hipSetDevice(ctx->cfg.device);
  if (!cx->canaries.empty()) {      // DDK_CANARY: who wrote behind its array?
    hipDeviceSynchronize();
    std::vector<unsigned char> h(4096);
    for (size_t i = 0; i < cx->canaries.size(); ++i) {
      if (hipMemcpy(h.data(), cx->canaries[i].tail, 4096, hipMemcpyDeviceToHost) != hipSuccess) continue;
      size_t bad = 0, first = 4096;
      for (size_t k = 0; k < 4096; ++k) if (h[k] != 0xA5) { ++bad; if (first == 4096) first = k; }
      if (bad) fprintf(stderr, "[ddk canary] cx %p (n_lig %d n_rec %d all_atoms %d): allocation #%zu of %zu B overrun: %zu bytes changed, first at +%zu\n", (void*)cx, cx->n_lig, cx->n_rec,
                       ctx->cfg.all_atoms, i, cx->canaries[i].bytes, bad, first);
    }
  }
  if (cx->stage_idx >= 0) { ctx->stage_pool[cx->stage_idx].in_flight = false; cx->stage_idx = -1; }   // a create that failed half way
  // the chunks go back to the context's pool (hipFree would synchronise the device); whoever takes one waits for this complex' last launch.
  // A complex that was driven from several streams (the API takes a stream per call): the last one first waits for the work still in
  // flight on the others, so that ONE event on it covers every launch that touched the chunks
  bool joined = true;
  for (hipStream_t o : cx->streams) {
    if (o == cx->last_stream) continue;
    hipEvent_t ev = nullptr;
    bool ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    if (ok) ok = hipEventRecord(ev, o) == hipSuccess && hipStreamWaitEvent(cx->last_stream, ev, 0) == hipSuccess;
    if (ev) hipEventDestroy(ev);      // (destroying a recorded event is deferred by the runtime until it has completed)
    joined = joined && ok;
  }
  if (!joined) hipDeviceSynchronize();      // could not order the streams: give the memory back only when everything has drained
  for (auto& a : cx->allocs) {
    if (!a.p) continue;
    ddk_ctx::PoolChunk c;
    c.p = a.p; c.cap = a.cap;
    ctx->pool_bytes_out -= (int64_t)a.cap;
    bool ok = hipEventCreateWithFlags(&c.free_after, hipEventDisableTiming) == hipSuccess;
    // (a complex that was never launched: its copies may still be in flight on the upload stream; otherwise the launch stream, which
    // waited for them in cx_wait_ready)
    if (ok) ok = hipEventRecord(c.free_after, cx->used ? cx->last_stream : ctx->up_stream) == hipSuccess;
    if (!ok || ctx->chunk_pool_bytes + c.cap > CHUNK_POOL_MAX_BYTES || ctx->chunk_pool.size() >= CHUNK_POOL_MAX_CHUNKS) {
      // the chunk leaves the process: wait for its last use first (hipFree does not wait for work on the context's NON-BLOCKING streams; a
      // 363-complex stream with ligands of 10-80 atoms filled round 3's 64-chunk pool and freed memory under kernels still in flight)
      if (ok) hipEventSynchronize(c.free_after); else hipDeviceSynchronize();
      if (c.free_after) hipEventDestroy(c.free_after);
      ctx_free(ctx, a.p);
      ctx->pool_frees++;
      continue;
    }
    ctx->chunk_pool.push_back(c);
    ctx->chunk_pool_bytes += c.cap;
  }
  if (cx->ready) hipEventDestroy(cx->ready);
  conf_complex_free(cx);
  delete cx;
}

int ddk_score_confidence(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float t_tr, float t_rot, float t_tor, float* out,
                         void* stream) {
  int rc = check_model(ctx, cx, B);
  if (rc) return rc;
  Model* M = (Model*)ctx->model;
  if (!M->confidence_mode) return fail(ctx, DDK_ERR_STATE, "ddk_score_confidence needs a context created with confidence_mode = 1");
  if (!lig_pos || !out) return fail(ctx, DDK_ERR_INVALID, "ddk_score_confidence: null argument");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  StepParams sp;
  if ((rc = make_step_params(ctx, t_tr, t_rot, t_tor, sp))) return rc;
  if ((rc = score_forward_impl(ctx, cx, B, lig_pos, sp, nullptr, nullptr, nullptr, (hipStream_t)stream))) return rc;
  hipError_t e = launch_conf_head(M->pred, cx->x_last, B, cx->n_lig, out, nullptr, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "confidence head");
}

int ddk_score_forward(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float t_tr, float t_rot, float t_tor,
                      float* tr_out, float* rot_out, float* tor_out, void* stream) {
  int rc = check_model(ctx, cx, B);
  if (rc) return rc;
  if (((Model*)ctx->model)->confidence_mode) return fail(ctx, DDK_ERR_STATE, "a confidence_mode context has no score heads (ddk_score_confidence)");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  StepParams sp;
  if ((rc = make_step_params(ctx, t_tr, t_rot, t_tor, sp))) return rc;
  return score_forward_impl(ctx, cx, B, lig_pos, sp, tr_out, rot_out, tor_out, (hipStream_t)stream);
}

int ddk_build_graph(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float t_tr, int32_t* edge_src_out,
                    int32_t* edge_dst_out, int64_t cap, int32_t* group_offsets_out, void* stream) {
  int rc = check_model(ctx, cx, B);
  if (rc) return rc;
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  if (!lig_pos || !edge_src_out || !edge_dst_out || !group_offsets_out) return fail(ctx, DDK_ERR_INVALID, "ddk_build_graph: null argument");
  const int64_t cap_b = (int64_t)B * ((int64_t)cx->M + (int64_t)cx->n_lig * LIG_CAP + 2LL * cx->n_lig * cx->n_rec + cx->E_rr);
  const int64_t need = cap_b < cx->edge_cap ? cap_b : cx->edge_cap;
  if (cap < need) return fail(ctx, DDK_ERR_INVALID, "ddk_build_graph: cap " + std::to_string(cap) + " is below the worst case " + std::to_string(need) + " of this batch");
  StepParams sp;
  if ((rc = make_step_params(ctx, t_tr, t_tr, t_tr, sp))) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = build_graph(ctx, cx, B, lig_pos, sp.cross_cutoff, /*prune=*/false, /*shared_rr=*/false, s);
  if (e == hipSuccess) e = hipMemcpyAsync(edge_src_out, cx->e_src, (size_t)need * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(edge_dst_out, cx->e_dst, (size_t)need * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(group_offsets_out, cx->info + I_GO, 5 * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return hip_fail(ctx, e, "ddk_build_graph");
  return DDK_OK;
}

int ddk_se3_update(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* pos, const float* tr, const float* rot,
                   const float* tor, float* pos_out, void* stream) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  if (!cx || B < 1) return fail(ctx, DDK_ERR_INVALID, "bad complex / batch");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  Se3Args A;
  A.pos = pos; A.tr = tr; A.rot = rot; A.tor = tor; A.noise = nullptr;
  for (int k = 0; k < 3; ++k) { A.sc[k] = 1.0f; A.nc[k] = 0.0f; }
  A.rot_u = cx->rot_u; A.rot_v = cx->rot_v; A.mask_rotate = cx->mask_rotate; A.B = B; A.n_lig = cx->n_lig; A.R = cx->R;
  A.pos_out = pos_out;
  hipError_t e = launch_se3(A, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "se3_update launch");
  return DDK_OK;
}

int ddk_randomize_position(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* pos0, const float* tor, const float* rot,
                           const float* tr, float* pos_out, void* stream) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  if (!cx || B < 1 || !pos0 || !rot || !pos_out) return fail(ctx, DDK_ERR_INVALID, "ddk_randomize_position: bad complex / batch / null argument");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  RandPosArgs A;
  A.pos0 = pos0; A.tor = tor; A.rot = rot; A.tr = tr;
  A.rot_u = cx->rot_u; A.rot_v = cx->rot_v; A.mask_rotate = cx->mask_rotate; A.B = B; A.n_lig = cx->n_lig; A.R = cx->R;
  A.pos_out = pos_out;
  hipError_t e = launch_randomize(A, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "randomize_position launch");
  return DDK_OK;
}

int ddk_pose_metrics(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* pos, const float* ref_pos, const uint8_t* atom_mask,
                     const int32_t* perms, int32_t n_perms, const float* rec_atom_pos, int32_t n_rec_atoms, float* out, void* stream) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  if (!cx || B < 1 || !pos || !ref_pos || !out) return fail(ctx, DDK_ERR_INVALID, "ddk_pose_metrics: bad complex / batch / null argument");
  if ((perms != nullptr) != (n_perms > 0) || (rec_atom_pos != nullptr) != (n_rec_atoms > 0))
    return fail(ctx, DDK_ERR_INVALID, "ddk_pose_metrics: perms / n_perms and rec_atom_pos / n_rec_atoms come in pairs");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  hipError_t e = launch_pose_metrics(pos, ref_pos, atom_mask, perms, n_perms, rec_atom_pos ? rec_atom_pos : cx->rec_pos, B, cx->n_lig,
                                     rec_atom_pos ? n_rec_atoms : cx->n_rec, out, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "pose_metrics launch");
  return DDK_OK;
}

int ddk_sample(ddk_ctx* ctx, ddk_complex* cx, int32_t B, int32_t steps, const float* t, const float* score_coeff,
               const float* noise_coeff, const float* noise, float* pos, void* stream) {
  int rc = check_model(ctx, cx, B);
  if (rc) return rc;
  if (((Model*)ctx->model)->confidence_mode) return fail(ctx, DDK_ERR_STATE, "a confidence_mode context has no score heads (ddk_score_confidence)");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  if (steps < 1 || !t || !score_coeff || !noise_coeff || !pos) return fail(ctx, DDK_ERR_INVALID, "ddk_sample: null argument");
  hipStream_t s = (hipStream_t)stream;
  const int R = cx->R;
  float* tr = cx->scores;
  float* rot = tr + (size_t)B * 3;
  float* tor = rot + (size_t)B * 3;
  const bool torsion = !ctx->cfg.no_torsion && R > 0;
  std::vector<StepParams> sps(steps);
  for (int k = 0; k < steps; ++k)
    if ((rc = make_step_params(ctx, t[3 * k], t[3 * k + 1], t[3 * k + 2], sps[k]))) return rc;
  const size_t n_sc = (size_t)B * (6 + (torsion ? R : 0));
  for (int k = 0; k < steps; ++k) {
    const bool guided = cx->cfg_weight != 0.0f && ctx->cfg.latent_dim > 0 && t[3 * k] <= cx->cfg_start && t[3 * k] >= cx->cfg_end;
    HeadArgs post;
    if ((rc = score_forward_impl(ctx, cx, B, pos, sps[k], tr, rot, torsion ? tor : nullptr, s, guided ? nullptr : &post))) return rc;
    if (guided) {
      // second, unconditional forward: unconditional = 1, latents zeroed (sampling.py:119-129)
      const float* ll = cx->lig_latent; const float* rl = cx->rec_latent; const float un = cx->unconditional;
      cx->lig_latent = cx->zero_lat; cx->rec_latent = cx->zero_lat + (size_t)B * cx->n_lig * ctx->cfg.latent_dim; cx->unconditional = 1.0f;
      float* tr2 = cx->scores2;
      float* rot2 = tr2 + (size_t)B * 3;
      float* tor2 = rot2 + (size_t)B * 3;
      rc = score_forward_impl(ctx, cx, B, pos, sps[k], tr2, rot2, torsion ? tor2 : nullptr, s);
      cx->lig_latent = ll; cx->rec_latent = rl; cx->unconditional = un;
      if (rc) return rc;
      hipError_t e2 = launch_cfg_combine(cx->scores, cx->scores2, cx->cfg_weight, (int64_t)n_sc, s);
      if (e2 != hipSuccess) return hip_fail(ctx, e2, "cfg_combine launch");
    }
    Se3Args A;
    A.pos = pos; A.tr = tr; A.rot = rot; A.tor = torsion ? tor : nullptr;
    A.noise = noise ? noise + (size_t)k * B * (6 + R) : nullptr;
    for (int j = 0; j < 3; ++j) { A.sc[j] = score_coeff[3 * k + j]; A.nc[j] = noise_coeff[3 * k + j]; }
    A.rot_u = cx->rot_u; A.rot_v = cx->rot_v; A.mask_rotate = cx->mask_rotate; A.B = B; A.n_lig = cx->n_lig; A.R = R;
    A.pos_out = pos;       // in place: the workgroup of sample b stages pos[b] in LDS before it writes pos[b] (k_se3.hip)
    hipError_t e = launch_se3(A, s, guided ? nullptr : &post);
    if (e != hipSuccess) return hip_fail(ctx, e, "se3_update launch");
  }
  return DDK_OK;
}

int ddk_last_graph_stats(ddk_ctx* ctx, ddk_complex* cx, int64_t* out, void* stream) {
  if (!ctx || !cx || !out) return DDK_ERR_INVALID;
  int32_t info[INFO_INTS];
  hipError_t e = hipMemcpyAsync(info, cx->info, sizeof(info), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(ctx, e, "graph stats readback");
  for (int g = 0; g < 4; ++g) out[g] = info[I_GO + 1 + g] - info[I_GO + g];
  out[4] = info[I_SHARED] >= 0 ? cx->E_rr : 0;   // edges of the shared rec-rec copy (layer-0 de-duplication)
  out[5] = info[I_E];      // total edges of the reference graph
  out[6] = info[I_OVF];    // overflow flag
  out[7] = cx->edge_cap;
  // rec-rec edges inside the backward receptive field of the heads: levels A, A+B, A+B+C (k_graph.hip); = E_rr * B when pruning is off
  out[8] = info[I_SEG + 1] - info[I_SEG]; out[9] = info[I_SEG + 2] - info[I_SEG]; out[10] = info[I_SEG + 3] - info[I_SEG];
  out[11] = info[I_MISMATCH];      // != 0: count / fill kernels disagreed about a sample's edges in some forward since the complex was created
  return DDK_OK;
}

int ddk_last_node_features(ddk_ctx* ctx, ddk_complex* cx, int32_t B, float* lig_out, float* rec_out, void* stream) {
  if (!ctx || !cx || !cx->x_last || B != cx->last_B) return fail(ctx, DDK_ERR_STATE, "no forward with this batch size has run");
  if (rec_out && !cx->last_full)
    return fail(ctx, DDK_ERR_STATE, "receptor rows of the last conv layer were not evaluated: call ddk_set_keep_receptor_features(on) before the forward");
  hipStream_t s = (hipStream_t)stream;
  const size_t nl = (size_t)B * cx->n_lig * XW, nr = (size_t)B * cx->n_rec * XW;
  hipError_t e = hipSuccess;
  if (lig_out) e = hipMemcpyAsync(lig_out, cx->x_last, nl * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess && rec_out) e = hipMemcpyAsync(rec_out, cx->x_last + nl, nr * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return hip_fail(ctx, e, "node feature copy");
  return DDK_OK;
}

int ddk_set_latents(ddk_ctx* ctx, ddk_complex* cx, const float* lig_latent, const float* rec_latent, float unconditional) {
  if (!ctx || !cx) return DDK_ERR_INVALID;
  if ((lig_latent == nullptr) != (rec_latent == nullptr)) return fail(ctx, DDK_ERR_INVALID, "ddk_set_latents: pass both latent arrays or neither");
  cx->lig_latent = lig_latent; cx->rec_latent = rec_latent; cx->unconditional = unconditional;
  cx->latent_dirty = true;      // (the arrays are caller-owned: every call may carry new values)
  return DDK_OK;
}

int ddk_set_keep_receptor_features(ddk_ctx* ctx, ddk_complex* cx, int32_t on) {
  if (!ctx || !cx) return DDK_ERR_INVALID;
  cx->keep_rec = on != 0;
  return DDK_OK;
}

int ddk_ar_logits(ddk_ctx* ctx, ddk_complex* cx, int32_t B, float* logits_out, void* stream) {
  int rc = check_model(ctx, cx, B);
  if (rc) return rc;
  Model* M = (Model*)ctx->model;
  if (!M->has_ar) return fail(ctx, DDK_ERR_STATE, "no AR predictor weights in this context (latent_s_predictor.* / latent_r_predictor.*)");
  if (!logits_out) return fail(ctx, DDK_ERR_INVALID, "ddk_ar_logits: null argument");
  if (!cx->x_last || cx->last_B != B || !cx->last_full)
    return fail(ctx, DDK_ERR_STATE, "ddk_ar_logits needs a forward of this batch size with ddk_set_keep_receptor_features(on) first");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  ArArgs A;
  A.x = cx->x_last; A.s = M->ar_s; A.r = M->ar_r; A.ar_ns = M->ar_ns; A.H = M->ar_H;
  A.n_lig_total = B * cx->n_lig; A.n_rec_total = B * cx->n_rec; A.n_lig = cx->n_lig; A.n_rec = cx->n_rec; A.logits = logits_out;
  hipError_t e = launch_ar_logits(A, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "ar_logits launch");
}

int ddk_ar_decode(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* logits, float temperature, const float* uniforms,
                  int32_t decoding_idx, int32_t latent_dim, float* lig_latent, float* rec_latent, int32_t* choices, void* stream) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context (device < 0) cannot launch kernels");
  if (!cx || B < 1 || !logits || !lig_latent || !rec_latent || latent_dim < 1 || decoding_idx < 0 || decoding_idx >= latent_dim)
    return fail(ctx, DDK_ERR_INVALID, "ddk_ar_decode: bad argument");
  if (temperature < 100.0f && !uniforms) return fail(ctx, DDK_ERR_INVALID, "ddk_ar_decode: uniforms are required below temperature 100");
  { hipError_t we = cx_wait_ready(cx, (hipStream_t)stream); if (we != hipSuccess) return hip_fail(ctx, we, "wait for the complex upload"); }
  ArDecodeArgs A;
  A.logits = logits; A.uniforms = uniforms; A.temperature = temperature; A.n_lig = cx->n_lig; A.n_rec = cx->n_rec; A.idx = decoding_idx;
  A.latent_dim = latent_dim; A.lig_latent = lig_latent; A.rec_latent = rec_latent; A.choices = choices;
  cx->latent_dirty = true;      // the decode writes into latent arrays that may be the ones ddk_set_latents registered
  hipError_t e = launch_ar_decode(A, B, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "ar_decode launch");
}

int ddk_set_receptive_field_pruning(ddk_ctx* ctx, int32_t on) {
  if (!ctx) return DDK_ERR_INVALID;
  ctx->prune = on != 0;
  return DDK_OK;
}

int ddk_set_guidance(ddk_ctx* ctx, ddk_complex* cx, float weight, float cfg_start, float cfg_end) {
  if (!ctx || !cx) return DDK_ERR_INVALID;
  if (weight != 0.0f && ctx->cfg.latent_dim <= 0) return fail(ctx, DDK_ERR_INVALID, "classifier-free guidance needs a latent-conditioned model");
  cx->cfg_weight = weight; cx->cfg_start = cfg_start; cx->cfg_end = cfg_end;
  return DDK_OK;
}

int ddk_profile_enable(ddk_ctx* ctx, int32_t on) {
  if (!ctx) return DDK_ERR_INVALID;
  if (ctx->host_only) return fail(ctx, DDK_ERR_STATE, "host-only context");
  for (auto& r : ctx->prof_recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  ctx->prof_recs.clear();
  ctx->prof_slots = 0;
  if (on && !ctx->prof_edges) {
    ctx->prof_cap = 32768;     // forwards (a 363-complex stream of config 4 is ~8 700; round 4's 4096 silently stopped recording after 205 complexes)
    if (hipHostMalloc((void**)&ctx->prof_edges, (size_t)ctx->prof_cap * PROF_INTS * sizeof(int32_t)) != hipSuccess)
      return fail(ctx, DDK_ERR_NOMEM, "hipHostMalloc failed");
  }
  ctx->prof = on != 0;
  return DDK_OK;
}

int ddk_profile_read(ddk_ctx* ctx, double* out, int32_t n) {
  if (!ctx || !out) return DDK_ERR_INVALID;
  const int L = ctx->cfg.num_conv_layers;
  if (n < 5 * L) return fail(ctx, DDK_ERR_INVALID, "profile buffer too small");
  for (int i = 0; i < 5 * L; ++i) out[i] = 0.0;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail(ctx, e, "profile sync");
  for (auto& r : ctx->prof_recs) {
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipEventElapsedTime");
    const int32_t* pe = ctx->prof_edges + (size_t)PROF_INTS * r.slot;     // [0] = E, [1] = edges of groups 0+1, [2 + k] = edges of table k
    const double E = pe[0], E01 = pe[1];
    out[5 * r.layer] += ms;
    out[5 * r.layer + 1] += 1.0;
    out[5 * r.layer + 2] += r.lig_only ? E01 : (double)pe[2 + r.tab];            // edges the launch evaluated
    out[5 * r.layer + 3] += r.lig_only ? E01 : (r.r01_skipped < 0 ? (double)pe[2 + r.tab] : E - (double)r.r01_skipped);   // without the receptive-field pruning (round-1 accounting; de-duplicated layer-0 messages are not work)
    out[5 * r.layer + 4] += E;                                                     // edges the reference evaluates in this layer
  }
  return DDK_OK;
}

int ddk_profile_read_forwards(ddk_ctx* ctx, double* out, int32_t max_forwards) {
  if (!ctx || !out || max_forwards < 0) return DDK_ERR_INVALID;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail(ctx, e, "profile sync");
  const int nf = ctx->prof_slots < max_forwards ? ctx->prof_slots : max_forwards;
  for (int i = 0; i < 4 * nf; ++i) out[i] = 0.0;
  for (auto& r : ctx->prof_recs) {
    if (r.slot >= nf) continue;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipEventElapsedTime");
    const int32_t* pe = ctx->prof_edges + (size_t)PROF_INTS * r.slot;
    const double E = pe[0], E01 = pe[1];
    out[4 * r.slot] += ms;
    out[4 * r.slot + 1] += r.lig_only ? E01 : (double)pe[2 + r.tab];
    out[4 * r.slot + 2] += r.lig_only ? E01 : (r.r01_skipped < 0 ? (double)pe[2 + r.tab] : E - (double)r.r01_skipped);
    out[4 * r.slot + 3] = (double)pe[7];
  }
  return ctx->prof_slots;
}

// Test hook: conv layer `layer` of the following forwards runs the three-limb kernel's TRACE instantiation, whose workgroup 0 writes the
// s_memtime stamps of its half phases to trace (DEVICE, [8][1024][8] uint32: burst start, burst end, epilogue start, epilogue end, before K steps 0 / 1 / 2 / 3); null: off
int ddk_debug_conv_trace(ddk_ctx* ctx, int32_t layer, uint32_t* trace) {
  if (!ctx) return DDK_ERR_INVALID;
  ctx->conv_trace = trace; ctx->conv_trace_layer = layer % 100; ctx->conv_trace_coarse = layer / 100;      // 100 + l: one record per unit; 200 + l: four more stamps inside every epilogue
  return DDK_OK;
}

// Test hook: layer-0 de-duplication of the rec-rec messages on / off (on by default; off = every sample evaluates all its rec-rec messages)
int ddk_debug_set_conv_workgroups(ddk_ctx* ctx, int32_t n) {
  if (!ctx || n < 1 || n > 1024) return DDK_ERR_INVALID;
  ctx->n_cu = n;      // persistent workgroups of a conv launch (default: one per CU); two contexts with half the CUs each can run side by side
  return DDK_OK;
}

int ddk_debug_pool_stats(ddk_ctx* ctx, int64_t* out) {
  if (!ctx || !out) return DDK_ERR_INVALID;
  out[0] = ctx->pool_mallocs; out[1] = ctx->pool_reuses; out[2] = ctx->pool_frees; out[3] = (int64_t)ctx->chunk_pool_bytes;
  out[4] = (int64_t)ctx->chunk_pool.size(); out[5] = ctx->pool_bytes_out; out[6] = ctx->pool_bytes_out_peak; out[7] = ctx->dev_bytes;
  return DDK_OK;
}

int ddk_debug_set_alloc_limit(ddk_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return DDK_ERR_INVALID;
  ctx->alloc_limit = bytes;
  return DDK_OK;
}

int ddk_debug_set_layer0_dedup(ddk_ctx* ctx, int32_t on) {
  if (!ctx) return DDK_ERR_INVALID;
  ctx->layer0_dedup = on != 0;
  return DDK_OK;
}

// Test hook: the layer-0 patch group of the last forward of the latent-conditioned model: counts[B + 1] = exclusive prefix of the patch edges per
// sample (counts[B] = total), mask[B * n_rec] = receivers that take their rec-rec sum from the patch group (HOST pointers; synchronises)
int ddk_debug_read_patch(ddk_ctx* ctx, ddk_complex* cx, int32_t B, int32_t* counts, uint8_t* mask) {
  if (!ctx || !cx || !counts || !mask) return DDK_ERR_INVALID;
  if (cx->patch_off < 0 || B < 1 || B > cx->max_batch) return fail(ctx, DDK_ERR_STATE, "no patch group on this complex (latent-conditioned models only)");
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(counts, cx->patch_cnt, (size_t)(B + 1) * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(mask, cx->rr_mask, (size_t)B * cx->n_rec, hipMemcpyDeviceToHost);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "ddk_debug_read_patch");
}

// Test hooks: the device Kabsch / axis-angle routines of k_se3.hip on caller-supplied DEVICE arrays (A, B [nb, n, 3] -> R [nb,3,3], t [nb,3];
// aa [n,3] -> R [n,3,3])
int ddk_debug_kabsch(ddk_ctx* ctx, int32_t nb, int32_t n, const float* A, const float* B, float* R_out, float* t_out, void* stream) {
  if (!ctx || ctx->host_only) return DDK_ERR_STATE;
  if (nb < 1 || n < 1 || n > MAX_LIG || !A || !B || !R_out || !t_out) return fail(ctx, DDK_ERR_INVALID, "ddk_debug_kabsch: bad argument");
  hipError_t e = launch_debug_kabsch(A, B, nb, n, R_out, t_out, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "debug kabsch");
}

int ddk_debug_axis_angle(ddk_ctx* ctx, int32_t n, const float* aa, float* R_out, void* stream) {
  if (!ctx || ctx->host_only) return DDK_ERR_STATE;
  if (n < 1 || !aa || !R_out) return fail(ctx, DDK_ERR_INVALID, "ddk_debug_axis_angle: bad argument");
  hipError_t e = launch_debug_axis_angle(aa, n, R_out, (hipStream_t)stream);
  return e == hipSuccess ? DDK_OK : hip_fail(ctx, e, "debug axis angle");
}

// Test hook: copy raw device-side edge arrays of the last forward to host buffers (counts via ddk_last_graph_stats).
int ddk_debug_read_edges(ddk_ctx* ctx, ddk_complex* cx, int64_t n, int32_t* src, int32_t* dst, float* emb, float* sh, int32_t* deg,
                         int64_t n_nodes) {
  if (!ctx || !cx) return DDK_ERR_INVALID;
  hipDeviceSynchronize();
  if (src) hipMemcpy(src, cx->e_src, n * 4, hipMemcpyDeviceToHost);
  if (dst) hipMemcpy(dst, cx->e_dst, n * 4, hipMemcpyDeviceToHost);
  if (emb) hipMemcpy(emb, cx->e_emb, n * NS * 4, hipMemcpyDeviceToHost);
  if (sh) hipMemcpy(sh, cx->e_sh, n * 16, hipMemcpyDeviceToHost);
  if (deg) hipMemcpy(deg, cx->deg, n_nodes * 4, hipMemcpyDeviceToHost);
  return DDK_OK;
}

}  // extern "C"
