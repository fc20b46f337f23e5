// Score-model level device structures (internal; see include/ddk.h for the ABI).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <thread>

#include "ddk_internal.h"

namespace ddk {

constexpr int SIG = 32;        // sigma_embed_dim
constexpr int DE = 32;         // distance_embed_dim
constexpr int MAX_LIG = 256;   // ligand atoms per sample the graph kernels support
constexpr int MAX_REC = 8192;  // residues per sample
constexpr int LIG_CAP = 33;    // radius_graph(max_num_neighbors=32) -> radius(..., 33) including self; an atom with 33 lower-index atoms in range keeps all 33 (self is not among them)
constexpr int BOND_CAP = 32;   // radius(..., max_num_neighbors=32) of the bond-centre graph

// Layout of the per-complex int32 `info` table written by graph_fill_kernel (write_group_tables; device side; no launch depends on a host read-back).
// Group 2 (rec-rec) of the merged edge list is stored in four segments ordered by the BACKWARD RECEPTIVE-FIELD LEVEL of the edge's
// receiving residue (see graph_fill_kernel): [A | B | C | rest], so that every conv layer evaluates a PREFIX of the group.
enum InfoSlot : int {
  I_GO = 5,       // [5..9]   go[0..4]: offsets of the reference's four edge groups [ll | lr | rr | rl], go[4] = E
  I_CNT = 10,     // [10..17] per-layer work-queue counters of the fused conv kernel
  I_FB = 18,      // [18..23] edge-feature block prefix over 5 feature groups (the 4 groups + the shared rec-rec copy), 256 edges per block
  I_E = 24,       // total edges of the reference graph (go[4])
  I_OVF = 25,     // capacity overflow flag
  I_MISMATCH = 31,   // sticky: graph_fill_kernel's recount of a sample's lig-lig / cross edges disagreed with graph_count_kernel's (the offsets come from one, the
                     // writes from the other: a slot would keep what the reused chunk held - ADVICE r04); reported by ddk_last_graph_stats out[11]
  I_SEG = 27,     // [27..30] first edge of the four level segments [A | B | C | rest] of group 2
  I_SHARED = 26,  // first edge of the shared rec-rec copy (= go[4]; E_rr edges in sample-0 numbering, layer-0 de-duplication) or -1
  I_TAB = 32,     // group tables: [32 + 8k + g] = gbeg, [36 + 8k + g] = gend of table k (N_TAB tables)
  I_EXEC = 80,    // [80] = E, [81] = go[2] (edges of groups 0+1), [82 + k] = edges table k evaluates over all four groups (k < 5), [87] = cross edges lig->rec
  I_HEAD = 96,    // heads' edge list (k_heads.hip): [96] = 0, [97] = B*n_lig (centre edges), [98] = B*n_lig, [99] = end of the bond edges (atomic
                  // cursor of heads_pre_kernel), [100], [101] = work-queue counters of the two head launches
  // latent-conditioned (DisCo) model, layer-0 de-duplication with per-sample patches (model.hip): a fifth edge group = ALL rec-rec edges of the
  // receivers whose layer-0 sum depends on this sample's (or sample 0's) non-zero receptor latents, at a fixed offset behind the edge arrays
  I_TABX = 104,   // [104..108] gbeg, [109..113] gend of the 5-group table [ll | lr | shared rr | rl | patch]
  I_PATCH = 114,  // number of patch edges (disco_patch kernels; stays until the latents change)
  I_FBX = 115,    // edge-feature block prefix end including the patch group
  INFO_INTS = 128
};
// group tables: which rec-rec edges a layer evaluates
enum GroupTable : int { TAB_ALL = 0, TAB_C = 1, TAB_B = 2, TAB_A = 3, TAB_SHARED = 4, N_TAB = 5 };
constexpr int CNT_STRIDE = 6;   // counts / offs per sample: ll radius edges, lr edges, rec-rec edges of level A, B, C, (spare)

// Everything that depends only on the diffusion time of a forward (computed on the HOST from t, which the
// sampler knows without reading the device: utils/sampling.py:106-113, diffusion_utils.py:12-16,58-69).
struct StepParams {
  float lig_node_sig[NS], rec_node_sig[NS];                   // W[:, sigma cols] . sigma_emb(t)  (+ nothing: bias is in the static part)
  float lig_edge_sigb[NS], rec_edge_sigb[NS], cross_edge_sigb[NS], center_edge_sigb[NS];   // first-layer bias incl. the sigma part
  float tr_sigb[NS], rot_sigb[NS];                            // tr/rot_final_layer.0: W[:,1:33].sigma_emb + b
  float tr_sigma, rot_sigma, tor_sigma;
  float so3_norm;          // so3.score_norm(rot_sigma)
  float torus_norm_sqrt;   // sqrt(torus.score_norm(tor_sigma))
  float cross_cutoff;      // 3*tr_sigma + 20 (dynamic_max_cross) or cross_max_distance
};

// 2-layer edge-embedding MLP: h = relu(W1d.gauss(+W1b.bond) + sigb) ; out = W2.h + b2
struct EdgeMlpDev {
  const float* w1d;   // [NS][DE]  columns that multiply the Gaussian distance expansion
  const float* w1b;   // [NS][4]   bond one-hot columns (ligand edges) or null
  const float* w2;    // [NS][NS]
  const float* b2;    // [NS]
  const float* w1l;   // [NS][2*latent_dim] latent columns (DisCo latent conditioning) or null
  const float* unc;   // [NS] *_edge_unconditional_embedding (latent_droprate > 0) or null
  float coeff;        // GaussianSmearing coeff
  float step;         // offset spacing (offset_k = k*step)
  const float* offset;  // [DE]
  // the same two weight matrices input-major ([DE][NS], [NS in][NS out]) for edge_features_kernel: a uniform (scalar) load of one input's 24 weights feeds 24
  // INDEPENDENT accumulators, and consecutive inputs are contiguous (k_graph.hip)
  const float* w1d_t = nullptr;
  const float* w2_t = nullptr;
};

struct ModelDev {
  EdgeMlpDev lig_edge, rec_edge, cross_edge, center_edge, final_edge;
  const float* final_edge_b1;   // [NS] (final_edge_embedding has no sigma part)
  // latent conditioning (latent_dim > 0, latent_vocab == 1): node-level weights
  const float *lig_w_lat, *rec_w_lat;       // [NS][latent_dim]
  const float *lig_node_unc, *rec_node_unc; // [NS] *_node_unconditional_embedding
  int latent_dim;
  // final_conv (tr/rot head)
  const float *fc_w0, *fc_b0, *fc_w4, *fc_b4;   // [2ns][2ns],[2ns],[144][2ns],[144]
  float fc_bn_scale[4];
  const float *tr_w0n, *tr_w3, *rot_w0n, *rot_w3;  // [NS] each
  float tr_b3, rot_b3;
  // torsion head
  const float *tb_w0, *tb_b0, *tb_w4, *tb_b4;   // [72][72],[72],[288][72],[288]
  const float *tb_bn_scale, *tb_bn_mean, *tb_bn_bias;  // [48] (mean/bias non-zero only on the 0e half)
  const float *tf_w0, *tf_w3;                    // [NS][2NS], [NS]
  // receptor AtomEncoder without its sigma columns (rec_node_static_kernel): table [38][ns], W[:, :ns], W[:, ns:ns+lm] transposed to [lm][ns], bias
  const float *rec_table, *rec_w_emb, *rec_w_lm_t, *rec_b;
};

// AR latent model predictors (k_ar.hip); BatchNorm1d (eval) folded into the Linear in front of it
constexpr int AR_H = 128;        // latent_hidden_dim the kernel is compiled for (smaller values run zero padded)
constexpr int AR_NS_MAX = 24;    // ns of the AR yml (scalar channels taken from each end of the node row)
struct ArMlpDev { const float *w0, *b0, *w4, *b4, *w8; float b8; };
struct ArArgs {
  const float* x;          // [N, XW] node features after the conv stack (ligand nodes first), receptor rows of the last layer included
  ArMlpDev s, r;           // latent_s_predictor (ligand atoms), latent_r_predictor (residues)
  int ar_ns, H;
  int n_lig_total, n_rec_total, n_lig, n_rec;
  float* logits;           // [B, n_lig + n_rec]
};
struct ArDecodeArgs {
  const float* logits;     // [B, n_lig + n_rec]
  const float* uniforms;   // [B] in [0, 1) (unused for temperature >= 100)
  float temperature;
  int n_lig, n_rec, idx, latent_dim;
  float* lig_latent;       // [B * n_lig, latent_dim]
  float* rec_latent;       // [B * n_rec, latent_dim]
  int32_t* choices;        // [B, latent_dim] or null
};
hipError_t launch_ar_logits(const ArArgs& A, hipStream_t s);
hipError_t launch_ar_decode(const ArDecodeArgs& A, int B, hipStream_t s);

struct ModelHost {   // host copies needed per forward / per complex
  std::vector<float> lig_tables;        // concatenated embedding tables [sum(dims)][NS]
  std::vector<int> lig_table_off;       // row offset of each categorical feature
  std::vector<float> lig_w_emb, lig_w_sig, lig_b;      // [NS][NS], [NS][SIG], [NS]
  std::vector<float> rec_table, rec_w_emb, rec_w_esm, rec_w_sig, rec_b;
  std::vector<float> le_w1s, le_b1, re_w1s, re_b1, ce_w1s, ce_b1, cen_w1s, cen_b1;   // sigma columns + bias of the edge MLP first layers
  std::vector<float> re_w1d;            // rec edge distance columns (static precompute per complex)
  std::vector<float> tr_w0s, tr_b0, rot_w0s, rot_b0;   // [NS][SIG], [NS]
  float rec_coeff = 0.f;
  std::vector<float> rec_offset;
  bool ready = false;
};

// confidence_predictor of a model in confidence_mode (score_model.py:110-121 / all_atom_score_model.py:143-153):
// Linear - BatchNorm1d - ReLU - Dropout - Linear - BatchNorm1d - ReLU - Dropout - Linear, the eval-mode BatchNorm folded into y = s (W x) + t
struct ConfPredictorDev {
  float *w0 = nullptr, *s0 = nullptr, *t0 = nullptr, *w4 = nullptr, *s4 = nullptr, *t4 = nullptr, *w8 = nullptr, *b8 = nullptr;
  int n_out = 1;
};
int conf_predictor_load(ddk_ctx* ctx, ConfPredictorDev& P);      // conf.hip
// scatter_mean of [x[:, :ns] | x[:, -ns:]] over every graph's ligand atoms + the predictor; ovf != null: a non-zero word turns the batch into NaN
hipError_t launch_conf_head(const ConfPredictorDev& P, const float* x, int B, int n_lig, float* out, const int32_t* ovf, hipStream_t s);

struct Model {
  ModelDev dev;
  ModelHost host;
  bool confidence_mode = false;      // TensorProductScoreModel(confidence_mode=True): no score heads, a confidence_predictor (ddk_score_confidence)
  ConfPredictorDev pred;
  bool has_ar = false;     // latent_{s,r}_predictor.* tensors were loaded into this context (the AR checkpoint's score-model copy)
  ArMlpDev ar_s, ar_r;
  int ar_ns = 0, ar_H = 0;
};


// ---- kernel argument blocks + launchers (k_graph.hip / k_heads.hip / k_se3.hip) ---------------------------
struct GraphArgs {
  const float* lig_pos;     // [B, n_lig, 3]
  const float* rec_pos;     // [n_rec, 3]
  const int32_t* bond_src;  // [M]
  const int32_t* bond_dst;
  const int32_t* rr_src;    // [E_rr]
  const int32_t* rr_dst;
  const int32_t* rr_outdeg; // [n_rec]
  int B, n_lig, n_rec, M, E_rr;
  float lig_r2;             // lig_max_radius^2
  float cross_cutoff;
  const int32_t* rr_start;  // [n_rec] exclusive prefix of rr_outdeg (first static edge of each residue)
  int32_t* counts;          // [B, CNT_STRIDE]: ll radius edges, lr edges, rec-rec edges of level A / B / C
  int32_t* offs;            // (unused since round 3: every wave of graph_fill_kernel prefixes the counts itself)
  int32_t* info;            // InfoSlot table
  uint8_t* levels;          // [B, n_rec] receptive-field level of every residue (graph_count_kernel -> graph_fill_kernel)
  int prune = 0;            // 1: order group 2 by receptive-field level (0: every residue is level A -> the reference order)
  int shared_rr = 0;        // 1: append the shared rec-rec copy (layer-0 de-duplication) behind the four groups
  int32_t* e_src;
  int32_t* e_dst;
  int32_t* e_aux;
  int32_t* deg;             // [B*(n_lig+n_rec)]
  int rec_node_base = -1;   // node id of sample 0's first residue (-1: B*n_lig, the score model's [lig | rec] numbering)
  int64_t patch_off = -1;   // >= 0: first edge of the DisCo patch group (I_TABX / I_PATCH / I_FBX are maintained)
  int64_t edge_cap = 0;     // capacity of the edge arrays (set by launch_graph)
  int cross_mirror = 0;     // 1: e_aux of a lig->rec edge = the slot of its flipped copy in the rec->lig group (edge_features_kernel evaluates the pair once); needs graph_cross_mirror_fits
};

// DisCo layer-0 patches: receivers whose rec-rec messages differ from the shared (sample-0) evaluation
// deterministic mode: the sample-aligned ranges of a conv launch (k_graph.hip: det_ranges_kernel; kinds documented there)
int det_ranges_count(int kind, int B);
hipError_t launch_det_ranges(const GraphArgs& G, int kind, int len_uniform, int32_t* out, hipStream_t s);

struct PatchArgs {
  const float* rec_latent;   // [B * n_rec, latent_dim]
  const int32_t* rr_start;   // [n_rec]
  const int32_t* rr_outdeg;  // [n_rec]
  const int32_t* rr_dst;     // [E_rr]
  int B, n_lig, n_rec, E_rr, latent_dim;
  int64_t patch_off;
  uint8_t* rr_mask;          // [B * n_rec]
  int32_t* patch_cnt;        // [B + 1]
  int32_t* info;
  int32_t *e_src, *e_dst, *e_aux;
};
hipError_t launch_disco_patch(const PatchArgs& a, hipStream_t s);

struct EdgeFeatArgs {
  const float* lig_pos;    // [B*n_lig,3]
  const float* rec_pos;    // [n_rec,3]
  const float* bond_attr;  // [M,4]
  const float* rr_pre1;    // [E_rr,NS]  W1d.gauss for the static receptor edges
  const float* rr_sh;      // [E_rr,4]
  const int32_t* e_src;
  const int32_t* e_dst;
  const int32_t* e_aux;
  const int32_t* info;
  float* e_emb;            // [E,NS]
  float* e_sh;             // [E,4]
  EdgeMlpDev lig, rec, cross;
  StepParams sp;
  int n_lig_total;         // B*n_lig
  int rec_node_base = -1;  // -1: n_lig_total
  int n_rec;
  int n_shared = 0;        // edges of the shared rec-rec copy (E_rr or 0)
  int cross_mirror = 0;    // 1: the lig->rec edges write their features into the flipped copies too (GraphArgs::cross_mirror); the rec->lig group has no feature blocks
  int g2_live_only = 0;    // 1: only the rec-rec edges of the level segments A, B, C get features (no layer evaluates the rest)
  int64_t patch_off = -1;  // >= 0: the DisCo patch group's edges start here (count in info[I_PATCH], feature blocks behind the shared copy's)
  const float* lig_latent; // [B*n_lig, latent_dim] or null
  const float* rec_latent; // [B*n_rec, latent_dim] or null
  float unconditional;     // data[...].unconditional (same value on every node of a forward, sampling.py:114-115,121-122)
  int latent_dim;
};

struct HeadArgs {
  const float* lig_pos;   // [B, n_lig, 3]
  const float* x;         // [N, XW] node features after the conv stack (ligand nodes first)
  ModelDev md;
  StepParams sp;
  int B, n_lig, R;
  int scale_by_sigma;
  const int32_t* rot_u;   // [R]
  const int32_t* rot_v;
  float lig_r2;
  float* tr_out;          // [B,3]
  float* rot_out;         // [B,3]
  float* tor_out;         // [B*R]
  // the heads' own edge list [centre edges (B*n_lig) | bond-neighbour edges (<= B*R*32)] and accumulators [B graphs | B*R bonds] (ddk_complex)
  int32_t *h_src, *h_dst, *h_deg, *h_info;
  float *h_attr, *h_sh, *h_sum;
  int32_t* prof_out = nullptr;          // profile mode: pinned host slot that receives exec_info[0..PROF_INTS) (heads_post_kernel), else null
  const int32_t* exec_info = nullptr;
  int deterministic;      // 1: fixed-stride bond edge ranges (BOND_CAP per bond, padded with null edges into a scratch row) instead of an atomic cursor
};

struct Se3Args {
  const float* pos;      // [B, n_lig, 3]
  const float* tr;       // [B,3] scores (or updates when coeff = (1,0))
  const float* rot;      // [B,3]
  const float* tor;      // [B*R] or null
  const float* noise;    // [B, 6+R] or null
  float sc[3], nc[3];
  const int32_t* rot_u;
  const int32_t* rot_v;
  const uint8_t* mask_rotate;   // [R, n_lig]
  int B, n_lig, R;
  float* pos_out;
};

struct RandPosArgs {
  const float* pos0;     // [n_lig, 3]
  const float* tor;      // [B, R] or null
  const float* rot;      // [B, 3, 3]
  const float* tr;       // [B, 3] or null
  const int32_t* rot_u;
  const int32_t* rot_v;
  const uint8_t* mask_rotate;   // [R, n_lig]
  int B, n_lig, R;
  float* pos_out;
};
hipError_t launch_randomize(const RandPosArgs& A, hipStream_t s);
hipError_t launch_debug_kabsch(const float* A, const float* Bp, int B, int n, float* R_out, float* t_out, hipStream_t s);
hipError_t launch_debug_axis_angle(const float* aa, int n, float* R_out, hipStream_t s);
hipError_t launch_pose_metrics(const float* pos, const float* ref, const uint8_t* mask, const int32_t* perms, int n_perms, const float* rec_pos,
                               int B, int n_lig, int n_rec, float* out, hipStream_t s);

int conf_model_finalize(ddk_ctx* ctx);   // conf.hip (all-atom confidence model)
void conf_complex_free(ddk_complex* cx);
void conf_model_destroy(ddk_ctx* ctx);
struct ConfComplex;

hipError_t launch_graph(const GraphArgs& G, int64_t edge_cap, hipStream_t s);
int graph_cross_mirror_fits(int n_lig, int n_rec);      // k_graph.hip: does the current device's LDS hold the residue x ligand-atom bit matrix of GraphArgs::cross_mirror?
hipError_t launch_edge_features(const EdgeFeatArgs& A, int64_t edge_cap, hipStream_t s);
hipError_t launch_edge_features_node(const EdgeFeatArgs& A, int64_t edge_cap, const NodePreArgs& P, const NodeEmbedArgs& E, hipStream_t s);   // + node embedding and layer-0 node terms
struct NodeEmbedArgs {
  const float* lig_static; const float* rec_static; StepParams sp; int B, n_lig, n_rec; float* x;
  const float *lig_latent, *rec_latent, *lig_w_lat, *rec_w_lat, *lig_unc, *rec_unc; float unconditional; int latent_dim;
};
hipError_t launch_node_embed(const NodeEmbedArgs& a, hipStream_t s);
// static per-complex precompute on the device (ddk_complex_create, upload stream)
struct RecStaticArgs { const float* rec_x; int n_rec, feat_dim, lm; const float *rec_table, *w_emb, *w_lm_t, *b; float* out; };
hipError_t launch_complex_static(const RecStaticArgs& R, const int32_t* rr_src, const int32_t* rr_dst, const float* rec_pos, int E, const EdgeMlpDev& m,
                                 float* pre1, hipStream_t s);
hipError_t launch_zero_fill(void* p, size_t bytes, hipStream_t s);      // k_graph.hip: 16-B stores (hipMemsetAsync's fill kernel is 50x slower at a few MB)
hipError_t launch_heads_pre(const HeadArgs& A, bool torsion, hipStream_t s);
hipError_t launch_heads_post(const HeadArgs& A, bool torsion, hipStream_t s);
hipError_t launch_se3(const Se3Args& A, hipStream_t s, const HeadArgs* post = nullptr);      // post != null: heads_post_kernel's work first, in the same launch
hipError_t launch_cfg_combine(float* score, const float* uncond, float weight, int64_t n, hipStream_t s);

}  // namespace ddk

struct ddk_complex {
  int n_lig = 0, n_rec = 0, M = 0, R = 0, E_rr = 0, max_batch = 0;
  // static device data
  int32_t *bond_src = nullptr, *bond_dst = nullptr, *rot_u = nullptr, *rot_v = nullptr, *rot_bond = nullptr;
  float* bond_attr = nullptr;
  uint8_t* mask_rotate = nullptr;
  float *rec_pos = nullptr, *lig_node_static = nullptr, *rec_node_static = nullptr;
  int32_t *rr_src = nullptr, *rr_dst = nullptr, *rr_outdeg = nullptr, *rr_start = nullptr;
  float *rr_pre1 = nullptr, *rr_sh = nullptr;
  // per-forward workspaces (sized for max_batch)
  int64_t edge_cap = 0;
  int32_t *e_src = nullptr, *e_dst = nullptr, *e_aux = nullptr, *deg = nullptr, *counts = nullptr, *offs = nullptr, *info = nullptr;
  uint8_t* levels = nullptr;
  float *e_emb = nullptr, *e_sh = nullptr, *xa = nullptr, *xb = nullptr, *sum = nullptr;
  float* scores = nullptr;
  const float *lig_latent = nullptr, *rec_latent = nullptr;   // caller-owned device arrays set by ddk_set_latents
  float unconditional = 0.0f;
  // layer-0 de-duplication of the latent-conditioned model (score_forward_impl): per-sample patch group behind the edge arrays
  int64_t patch_off = -1;          // first edge of the patch region (capacity max_batch * E_rr) or -1: not allocated
  uint8_t* rr_mask = nullptr;      // [max_batch * n_rec] 1: this receiver's rec-rec sum comes from the patch group, not from the shared rows
  int32_t* patch_cnt = nullptr;    // [max_batch + 1] patch edges per sample -> exclusive prefix
  bool latent_dirty = true;        // the latent arrays may have changed since the patch group was built
  int patch_B = 0;                 // batch size the patch group was built for (node numbering depends on it)
  float cfg_weight = 0.0f, cfg_start = 1.0f, cfg_end = 0.0f;   // ddk_set_guidance
  float *zero_lat = nullptr, *scores2 = nullptr;
  float* part = nullptr;      // deterministic mode: partial rows of the conv launches (ConvLaunch::part)
  int32_t* det_rng = nullptr; // deterministic mode: sample-aligned ranges of the current conv launch | of the final_conv head (det_ranges_kernel)
  float* sum_rr0 = nullptr;   // [n_rec, XW] layer-0 rec-rec messages shared by all samples
  int32_t *h_src = nullptr, *h_dst = nullptr, *h_deg = nullptr;      // heads' edge list and accumulators (k_heads.hip)
  float *h_attr = nullptr, *h_sh = nullptr, *h_sum = nullptr;
  float* pre = nullptr;       // [N, PRE_W] per-node terms of the upcoming layer's GEMM1 (ddk_internal.h: ConvLayerDev::wn)
  float* x_last = nullptr;    // node features after the conv stack of the last forward
  int last_B = 0;
  bool sum_clean = false;             // the accumulators are all zero (the last forward completed; node_finalize clears behind itself)
  bool keep_rec = false, last_full = false;   // last conv layer: all groups (true) or ligand-side groups only
  ddk::ConfComplex* conf = nullptr;   // all-atom level (ddk_complex_set_atoms), owned
  ddk_ctx* owner = nullptr;
  struct Chunk { void* p; size_t cap; };
  std::vector<Chunk> allocs;          // arena chunks (returned to the context's pool by ddk_complex_destroy)
  char* chunk = nullptr;              // current chunk: bump allocation, 256-B aligned
  size_t chunk_cap = 0, chunk_off = 0, reserve_hint = 0;
  bool oom = false;                   // a chunk allocation failed (checked once at the end of the creating call)
  // staged upload: host arrays are copied into a pinned buffer and leave for the device on the context's upload stream
  int stage_idx = -1;                 // index into owner->stage_pool while a staging session is open
  size_t stage_off = 0;
  struct CopyRec { void* dst; size_t off, bytes; };
  std::vector<CopyRec> pending;
  hipEvent_t ready = nullptr;         // recorded on the upload stream behind the last staged copy
  bool ready_pending = false;         // compute streams still have to wait for `ready`
  hipStream_t last_stream = nullptr;  // stream of the last launch that used this complex (ordering of the chunks' reuse)
  std::vector<hipStream_t> streams;   // every stream an entry point was given for this complex (ddk_complex_destroy joins them)
  bool used = false;
  struct Canary { char* tail; size_t bytes; };
  std::vector<Canary> canaries;       // DDK_CANARY debugging aid
  std::vector<int32_t> h_rr;          // host copies the confidence level needs again (ddk_complex_set_atoms)
  std::vector<float> h_rec_pos;
};

namespace ddk {
// static per-complex precompute on a few host threads (independent rows; results do not depend on the thread count)
template <typename F>
inline void host_parallel_for(int n, F&& body) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 8 ? 8 : nt);
  if (n < 512 || nt == 1) {
    for (int i = 0; i < n; ++i) body(i);
    return;
  }
  std::vector<std::thread> th;
  const int per = (n + (int)nt - 1) / (int)nt;
  for (unsigned t = 0; t < nt; ++t) {
    const int lo = (int)t * per, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) break;
    th.emplace_back([lo, hi, &body]() { for (int i = lo; i < hi; ++i) body(i); });
  }
  for (auto& x : th) x.join();
}


// Device memory of a complex comes from a few large chunks instead of one hipMalloc per array (34 + of them per complex, each a
// driver call), and the chunks come from / return to a pool of the context: hipFree synchronises the device, which would stall
// a caller that drops complexes while the GPU is busy.
void* cx_new_chunk(ddk_complex* cx, size_t cap);                       // model.hip: pool or hipMalloc
int cx_stage_begin(ddk_ctx* ctx, ddk_complex* cx, size_t bytes);        // open a staging session with room for `bytes`
bool cx_put(ddk_complex* cx, void* dst, const void* src, size_t bytes); // host -> staging (or a synchronous copy without a session)
int cx_stage_flush(ddk_ctx* ctx, ddk_complex* cx);                      // enqueue the staged copies, record cx->ready
hipError_t cx_wait_ready(ddk_complex* cx, hipStream_t s);               // first thing every launch entry point does with a complex
inline void cx_reserve(ddk_complex* cx, size_t bytes) { cx->reserve_hint = bytes; }
inline bool cx_canary_on() { static const bool on = getenv("DDK_CANARY") != nullptr; return on; }      // debugging aid: 4 KB of 0xA5 behind every array, checked at destroy
inline void* cx_alloc(ddk_complex* cx, size_t bytes) {
  const size_t need = (((bytes ? bytes : 4) + 255) & ~(size_t)255) + (cx_canary_on() ? 4096 : 0);
  if (!cx->chunk || cx->chunk_off + need > cx->chunk_cap) {
    size_t cap = need > cx->reserve_hint ? need : cx->reserve_hint;
    if (cap < ((size_t)1 << 20)) cap = (size_t)1 << 20;
    void* p = cx_new_chunk(cx, cap);
    if (!p) { cx->oom = true; return nullptr; }
    cx->chunk = (char*)p; cx->chunk_off = 0; cx->reserve_hint = 0;
  }
  void* r = cx->chunk + cx->chunk_off;
  cx->chunk_off += need;
  if (cx_canary_on()) {
    char* tail = (char*)r + (need - 4096);
    hipMemset(tail, 0xA5, 4096);
    cx->canaries.push_back({tail, bytes});
  }
  {      // debugging aid (DDK_TRACE_ALLOC): every array of a complex with its address range, to match a GPU fault address against
    static const bool trace = getenv("DDK_TRACE_ALLOC") != nullptr;
    if (trace && bytes >= 4096) fprintf(stderr, "[ddk alloc] cx %p ctx %p  %p .. %p  (%zu B, chunk %p + %zu of %zu)\n", (void*)cx, (void*)cx->owner, r, (void*)((char*)r + bytes), bytes,
                                        (void*)cx->chunk, cx->chunk_off - need, cx->chunk_cap);
  }
  return r;
}
}  // namespace ddk
