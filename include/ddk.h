/* ddk — C ABI of the MI355X-native DiffDock-S / DisCo-DiffDock-S score-model + sampler hot path.
 *
 * This is the drop-in boundary.  Every entry point names the reference
 * interface it replaces (paths relative to the reference checkout).  Conventions:
 *   - plain C, no torch types; all array arguments are CALLER-OWNED DEVICE pointers to
 *     contiguous row-major fp32 / int32 arrays unless the parameter is documented "host";
 *   - functions enqueue work on the given hipStream_t (passed as void*) and do not synchronise;
 *   - return 0 on success, a negative ddk_status otherwise; never throw across the ABI;
 *     ddk_last_error(ctx) returns a message for the last failure on that context;
 *   - a context is bound to one device, owns only its weight copies and workspaces, is not
 *     thread-safe, and holds no global state (one context per GPU / process).
 */
#ifndef DDK_H
#define DDK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ddk_ctx ddk_ctx;
typedef struct ddk_complex ddk_complex;

enum ddk_status {
  DDK_OK = 0,
  DDK_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  DDK_ERR_HIP = -2,       /* a HIP runtime call failed */
  DDK_ERR_STATE = -3,     /* call order (e.g. weights not finalised) */
  DDK_ERR_NOMEM = -4
};

/* Constructor arguments of TensorProductScoreModel as utils/model_utils.py:25-68 (get_model) maps
 * them from model_parameters.yml, plus the ctor defaults it leaves alone (models/score_model.py:15-24)
 * and the diffusion constants t_to_sigma needs (utils/diffusion_utils.py:12-16). */
typedef struct ddk_config {
  int32_t ns, nv, num_conv_layers;
  int32_t sigma_embed_dim, distance_embed_dim, cross_distance_embed_dim;
  float lig_max_radius, rec_max_radius, cross_max_distance, center_max_distance;
  int32_t dynamic_max_cross;
  float embedding_scale;
  int32_t scale_by_sigma, no_torsion, batch_norm;
  int32_t latent_dim, latent_vocab;
  float latent_droprate;
  int32_t lm_embedding_dim;            /* 1280 when esm_embeddings_path is set, else 0 */
  float tr_sigma_min, tr_sigma_max, rot_sigma_min, rot_sigma_max, tor_sigma_min, tor_sigma_max;
  int32_t device;                      /* HIP device ordinal */
  /* all-atom confidence model (models/all_atom_score_model.py through get_model(..., confidence_mode=True)): */
  int32_t all_atoms;                   /* 1: AAScoreModel in confidence_mode (sh_lmax=2 FCTP, OldAtomEncoder, 9 convs/layer) */
  int32_t num_confidence_outputs;      /* len(rmsd_classification_cutoff)+1 when it is a list, else 1 */
  int32_t confidence_no_batchnorm;
  /* how the radial-MLP GEMMs (Linear(72,72) + ReLU + Linear(72,W), tensor_layers.py:140-143) of the fused conv kernel are multiplied.  Inputs, weights,
   * accumulators and outputs are fp32 in every mode; 0 and 3 run on the f16 matrix pipe with every fp32 operand carried as fp16 LIMBS (after an exact
   * power-of-two range scaling per weight group / per edge), 1 on the fp32 matrix pipe.
   * 0 (default, since ddk 0.8): TWO limbs x = hi + mid (hi = fp16(x), mid = fp16(x - hi), both rounded to nearest: |x - hi - mid| <= 2^-22 |x|) and the
   *    three limb products hi.hi + hi.mid + mid.hi in one fp32 accumulator (k_conv_x2.hip); mid.mid, <= 2^-22 relative like the operands' own truncation, is dropped.
   *    Error per product <= 3 * 2^-22 relative, of a K = 72 dot product below the classical fp32 bound 72 * 2^-24 and, restated bit for bit on the host, not above an fp32 FMA
   *    chain's over the same operands (tests/test_limb_bound.py); measured against the fp64 oracle it is level with mode 1
   *    and mode 3 (6 - 10e-8 relative on every layer shape, tests/test_gpu_round6.py::test_two_limb_kernel_is_fp32_grade), 14 instead of 27 MFMAs per 32-edge weight tile
   *    (DESIGN.md 3.3).
   * 3: THREE limbs x = hi + mid + lo (exact for every value within 2^-15 of its range-scaling group's maximum, off by <= 2^-39 of that maximum below), six of the
   *    nine limb products kept (the dropped ones are <= 3 * 2^-33 relative), two fp32 accumulators (k_conv_x.hip): products exact to 2^-33 - the default of
   *    ddk 0.4 - 0.7, ~35 % more conv time than 0.
   * 1: v_mfma_f32_32x32x2_f32, plain fp32 FMA chains (k_conv.hip) - the stated fallback.  All three apply to the score model, its heads and the all-atom
   *    confidence model's conv layers.
   *    (Round 5's value 2 - a software-pipelined one-wave-per-SIMD form, measured 9 % slower - is refused: the kernel lives under tools/variants/.) */
  int32_t conv_kernel;
  /* 1: fixed summation order per node in the score model's conv layers and heads (scatter_mean of tensor_layers.py:159): edges are
   *    sorted by the receiving node, so run tails STORE and the runs that straddle 32-edge tiles are folded in tile order by a second
   *    small kernel; one accumulator per (node, receiving edge group); no float atomics -> bit-identical outputs run to run.
   *    0 (default): wave-level segmented sums + fp32 atomics on the run tails (~1e-7 relative run-to-run noise).  Applies to
   *    ddk_score_forward / ddk_sample; ddk_conv_forward (caller-ordered edges) keeps the atomics; not with all_atoms. */
  int32_t deterministic;
  /* 1 (with all_atoms = 0): the coarse-grained model in confidence_mode - TensorProductScoreModel(confidence_mode=True) as
   *    get_model(args, ..., confidence_mode=True) builds it for a checkpoint without all_atoms (models/score_model.py:110-121, 186-189, 263-266):
   *    the score model's graph, embeddings and conv stack, complex_t used as sigma directly, no centre / torsion heads, a confidence_predictor
   *    on the pooled ligand scalars.  Evaluated by ddk_score_confidence; ddk_score_forward / ddk_sample refuse such a context. */
  int32_t confidence_mode;
} ddk_config;

/* ---- lifetime ---------------------------------------------------------------------------- */
int ddk_create(const ddk_config* cfg, ddk_ctx** out);
void ddk_destroy(ddk_ctx* ctx);
const char* ddk_last_error(ddk_ctx* ctx);
const char* ddk_version(void);

/* ---- checkpoint: replaces model.score_model.load_state_dict(state_dict, strict=True)
 *      (evaluate.py:169-171).  One call per state_dict tensor, HOST pointer, reference key name
 *      (e.g. "conv_layers.3.fc.2.4.weight").  Unknown "*.tp.*" buffer keys are ignored.
 *      ddk_finalize_weights checks that every required key arrived with the right shape and packs
 *      the radial-MLP weights into the MFMA fragment order the fused kernel streams. */
int ddk_load_weights(ddk_ctx* ctx, const char* name, const float* host_ptr, const int64_t* shape, int32_t ndim);
int ddk_finalize_weights(ddk_ctx* ctx);

/* Host tables of utils/so3.py:91-95 (_exp_score_norms, 1000 doubles) and utils/torus.py:79-83
 * (score_norm_, 5001 doubles; Monte-Carlo in the reference, shipped as data here). */
int ddk_set_score_norm_tables(ddk_ctx* ctx, const double* so3_exp_score_norms, int32_t n_so3,
                              const double* torus_score_norm, int32_t n_torus);

/* ---- a12: FasterTensorProduct.forward(in_, sh, weight)  models/tensor_layers.py:65-116.
 *      x_dst [E, Din] (already gathered node_attr[edge_dst]), sh [E,4], w [E, W] -> out [E, Dout]
 *      with the irreps of conv layer `layer` (0..num_conv_layers-1).  HBM-bound (streams w). */
int ddk_tp_forward(ddk_ctx* ctx, int32_t layer, const float* x_dst, const float* sh, const float* w,
                   int64_t E, float* out, void* stream);

/* ---- a11: TensorProductConvLayer.forward(node_attr, edge_index, edge_attr, edge_sh)
 *      models/tensor_layers.py:147-168 for conv layer `layer`, fused: radial MLP (fp32 MFMA) +
 *      tensor product + segmented scatter-sum + mean + BatchNorm(eval) + residual, without ever
 *      materialising the [E, W] weight tensor.
 *      x [N, Din]; edge_src/edge_dst [E] int32 (edge_index rows 0/1); group_offsets[5] HOST array
 *      (edge ranges of the 4 edge groups, group g uses fc[g]); edge_attr [E, 3*ns] (the
 *      concatenated per-group edge_attr list); sh [E,4]; out [N, Dout]. */
int ddk_conv_forward(ddk_ctx* ctx, int32_t layer, const float* x, int64_t N, const int32_t* edge_src,
                     const int32_t* edge_dst, const int64_t* group_offsets, const float* edge_attr,
                     const float* sh, float* out, void* stream);

/* ---- one complex: the graph tensors datasets_utils/process_mols.py emits.
 *      All pointers are HOST pointers; the library uploads them and precomputes everything that is
 *      constant over the reverse-diffusion steps and over the samples (receptor embedding without
 *      its sigma part, receptor-receptor geometry). */
typedef struct ddk_complex_desc {
  int32_t n_lig, n_rec, n_bond_edges /* directed, = 2*bonds */, n_rot, n_rec_edges, rec_feat_dim /* 1 + lm dim */;
  const int32_t* lig_x;          /* [n_lig, 16]   data['ligand'].x */
  const int32_t* bond_index;     /* [2, n_bond_edges]   data['ligand','ligand'].edge_index */
  const float* bond_attr;        /* [n_bond_edges, 4]   .edge_attr */
  const uint8_t* edge_mask;      /* [n_bond_edges]      data['ligand'].edge_mask */
  const uint8_t* mask_rotate;    /* [n_rot, n_lig]      data['ligand'].mask_rotate */
  const float* rec_x;            /* [n_rec, rec_feat_dim] data['receptor'].x */
  const float* rec_pos;          /* [n_rec, 3] */
  const int32_t* rec_edge_index; /* [2, n_rec_edges]    data['receptor','receptor'].edge_index */
} ddk_complex_desc;

int ddk_complex_create(ddk_ctx* ctx, const ddk_complex_desc* desc, int32_t max_batch, ddk_complex** out);
void ddk_complex_destroy(ddk_ctx* ctx, ddk_complex* cx);

/* ---- all-atom confidence model (models/all_atom_score_model.py in confidence_mode; context created with all_atoms = 1 and
 *      the reference's confidence checkpoint keys): the receptor-atom level of the graph (datasets_utils/process_mols.py:383-477),
 *      HOST pointers; ddk_complex_set_atoms also needs the complex's ligand ids and receptor features again because the node
 *      embeddings of this model (OldAtomEncoder, models/layers.py:81-116) are computed here. */
typedef struct ddk_atoms_desc {
  int32_t n_atom, n_atom_edges;
  const int32_t* atom_x;          /* [n_atom, 4]   data['atom'].x */
  const float* atom_pos;          /* [n_atom, 3] */
  const int32_t* atom_edge_index; /* [2, n_atom_edges]  data['atom','atom'].edge_index */
  const int32_t* atom_rec_index;  /* [2, n_atom]        data['atom','receptor'].edge_index (row 0 = arange) */
} ddk_atoms_desc;
int ddk_complex_set_atoms(ddk_ctx* ctx, ddk_complex* cx, const ddk_atoms_desc* atoms, const int32_t* lig_x, const float* rec_x,
                          int32_t rec_feat_dim);

/* confidence_model(batch) -> [B, num_confidence_outputs]  (utils/sampling.py:230-243 with set_time(..., 0, 0, 0);
 * models/all_atom_score_model.py:203-284): lig_pos [B, n_lig, 3] DEVICE, out [B, num_confidence_outputs] DEVICE. */
int ddk_confidence_forward(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float* out, void* stream);

/* confidence_model(complex_graph_batch) of utils/sampling.py:239-240 - the branch WITHOUT confidence_data_list: a coarse-grained confidence
 * model (ddk_config.confidence_mode) evaluated on the score model's own graphs at the final poses.  The reference does not reset the times in that
 * branch: (t_tr, t_rot, t_tor) are the LAST step's schedule values, which confidence_mode uses as sigmas (models/score_model.py:186-189).
 * cx: a complex created in THIS context from the same ddk_complex_desc the score model's was; lig_pos [B, n_lig, 3] DEVICE,
 * out [B, num_confidence_outputs] DEVICE. */
int ddk_score_confidence(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float t_tr, float t_rot, float t_tor, float* out,
                         void* stream);

/* Status words of the last ddk_confidence_forward of `cx`, copied WITHOUT synchronising: enqueues an asynchronous copy of 20 int32 into
 * host_out (HOST, pinned memory if the copy is to overlap): [0..8] first edge and [9..17] end of the nine edge groups [ll lr la aa al ar
 * rr rl ra], [18] ligand-atom edge cursor, [19] != 0: the ligand-atom edge capacity overflowed (the confidences of that batch are invalid;
 * the Python shim raises).  Valid once the stream has passed the copy. */
int ddk_confidence_status(ddk_ctx* ctx, ddk_complex* cx, int32_t* host_out, void* stream);

/* ---- DisCo latent conditioning (models/score_model.py:170-184, 209-215, 329-337, 358-366, 392-402; latent_vocab == 1):
 *      lig_latent [B*n_lig, latent_dim], rec_latent [B*n_rec, latent_dim] (data['ligand'|'receptor'].latent_h, DEVICE,
 *      caller-owned, must stay valid for the following forwards) and the value of data[...].unconditional.
 *      Applies to the subsequent ddk_score_forward / ddk_sample calls on this complex; NULL, NULL clears.
 *      The library derives a per-sample edge group from WHERE the receptor latents are non-zero (layer-0 de-duplication)
 *      when the first forward after this call runs: call ddk_set_latents again after changing the arrays in place (ddk_ar_decode does
 *      this bookkeeping itself). */
int ddk_set_latents(ddk_ctx* ctx, ddk_complex* cx, const float* lig_latent, const float* rec_latent, float unconditional);

/* ---- a22: the AR latent model (config 3).  A context that was given the AR checkpoint (its own score-model copy under the plain
 *      key names + latent_s_predictor.* / latent_r_predictor.*) evaluates, after a forward at t = 1 with unconditional = 1, the
 *      partially decoded latents bound by ddk_set_latents and ddk_set_keep_receptor_features(on)  (= score_model.embed(),
 *      models/pretrained_score_encoder.py:58-75):
 *      ddk_ar_logits: the two predictor MLPs (Linear-BatchNorm1d-ReLU-Linear-BatchNorm1d-ReLU-Linear, :24-45) on the scalar
 *        channels [x[:, :ns] | x[:, -ns:]] of every node -> logits [B, n_lig + n_rec] (ligand atoms first, :84-88), DEVICE;
 *      ddk_ar_decode: GenericEncoder.encode_ar's pick for latent dimension decoding_idx (models/model_classes.py:21-47):
 *        logits * temperature; temperature >= 100: argmax; else index i with probability exp(.)_i / sum (NaN -> 0, inf -> FLT_MAX
 *        like torch.nan_to_num), drawn by inverse CDF from the caller's uniforms [B] in [0,1) (DEVICE; the reference calls
 *        torch.multinomial: same distribution, the draws stay with the caller); sets lig_latent[b*n_lig + c, decoding_idx] = 1 or
 *        rec_latent[b*n_rec + c - n_lig, decoding_idx] = 1 ([B*n, latent_dim] DEVICE arrays the caller zeroed) and, when not NULL,
 *        choices[b, decoding_idx] = c ([B, latent_dim] int32 DEVICE). */
int ddk_ar_logits(ddk_ctx* ctx, ddk_complex* cx, int32_t B, float* logits_out, void* stream);
int ddk_ar_decode(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* logits, float temperature, const float* uniforms,
                  int32_t decoding_idx, int32_t latent_dim, float* lig_latent, float* rec_latent, int32_t* choices, void* stream);

/* ---- classifier-free guidance of the sampler (utils/sampling.py:119-135): while cfg_end <= t_tr <= cfg_start every step of
 *      ddk_sample runs a second forward with unconditional = 1 and zeroed latents and uses
 *      score + weight * (score - score_unconditional).  weight = 0 (default) disables it. */
int ddk_set_guidance(ddk_ctx* ctx, ddk_complex* cx, float weight, float cfg_start, float cfg_end);

/* ---- The heads read ligand rows only (models/score_model.py:286-308), so by default the LAST conv layer evaluates only
 *      the messages into ligand nodes (edge groups lig-lig and lig->rec) and the receptor rows after it are not produced.
 *      Turn this on before a forward whose receptor rows will be read with ddk_last_node_features (the embed() path of
 *      the AR latent model, models/pretrained_score_encoder.py:66-75): the last layer then evaluates all four groups as
 *      the reference does.  tr/rot/tor are identical either way. */
int ddk_set_keep_receptor_features(ddk_ctx* ctx, ddk_complex* cx, int32_t on);

/* ---- Backward receptive-field pruning (default ON; exact in real arithmetic): because the heads read ligand rows only, conv
 *      layer L-2 has to produce receptor rows only at the residues that carry a cross edge, layer L-3 only at those plus the
 *      senders of their receptor-receptor messages, and so on (csrc/k_graph.hip).  The receptor-receptor messages outside that
 *      set are not evaluated.  tr/rot/tor are unchanged (tests: test_pruned_layers_equal_full); 0 switches it off (every layer
 *      evaluates every receptor-receptor message, as the reference does).  Forwards with ddk_set_keep_receptor_features(on)
 *      never prune.  In a confidence-model context the same switch governs the level-A / level-B pruning of the static edge groups in its
 *      second- and third-to-last layers (the predictor pools ligand rows only; DESIGN.md 3.2, test_confidence_level_a_pruning_equal_full). */
int ddk_set_receptive_field_pruning(ddk_ctx* ctx, int32_t on);

/* ---- a5-a17: model.score_model(batch) -> (tr[B,3], rot[B,3], tor[B*R])  models/score_model.py:259-308
 *      for B copies of one complex at a common time (utils/sampling.py:113-117).
 *      lig_pos [B, n_lig, 3]; outputs tr [B,3], rot [B,3], tor [B*n_rot]. */
int ddk_score_forward(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float t_tr, float t_rot,
                      float t_tor, float* tr_out, float* rot_out, float* tor_out, void* stream);

/* ---- a18-a21: modify_conformer_batch(pos, data, tr_update, rot_update, torsion_updates, mask_rotate)
 *      utils/diffusion_utils.py:37-55 (axis-angle rotation, sequential torsion updates, Kabsch re-alignment).
 *      pos [B,n_lig,3], tr [B,3], rot [B,3], tor [B*n_rot] (may be NULL: rigid only) -> pos_out [B,n_lig,3]. */
int ddk_se3_update(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* pos, const float* tr, const float* rot,
                   const float* tor, float* pos_out, void* stream);

/* ---- randomize_position(data_list, no_torsion, no_random, tr_sigma_max)  utils/sampling.py:12-34 for B copies of one
 *      complex in one launch: per sample  pos = torsions(pos0, tor[b])  (bond order, utils/torsion.py:48-68,
 *      zero updates skipped);  pos = (pos - mean(pos)) @ rot[b]^T + tr[b].
 *      pos0 [n_lig,3] the conformer (DEVICE); tor [B, n_rot] = np.random.uniform(-pi, pi) draws or NULL (no_torsion);
 *      rot [B,3,3] row-major rotation matrices (scipy Rotation.random().as_matrix()); tr [B,3] = N(0, tr_sigma_max) draws or
 *      NULL (no_random); pos_out [B,n_lig,3].  The draws stay with the caller: the reference's host RNG streams or device RNG. */
int ddk_randomize_position(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* pos0, const float* tor, const float* rot,
                           const float* tr, float* pos_out, void* stream);

/* ---- pose metrics of evaluate.py:297-338 for B poses of one complex, one launch:
 *      out[b] = { rmsd, centroid_distance, min_cross_distance, min_self_distance } with
 *        rmsd = min over k < n_perms of sqrt(mean_i |pos[perms[k][i]] - ref[i]|^2)
 *               = the symmetry-corrected RMSD of evaluate.py:308-310 (spyrmsd.rmsd.symmrmsd, no minimisation: the minimum over the
 *               graph automorphisms of the ligand).  perms [n_perms, n_lig] int32 DEVICE: row k maps every ligand atom i to its image
 *               (computed once per ligand on the caller's side, INTEGRATION.md shows the spyrmsd / networkx recipe; row 0 should be
 *               the identity; entries of masked-out atoms are ignored; a row with an entry outside [0, n_lig) is checked on the device and
 *               never wins the minimum - rmsd = inf if no row is valid).  perms = NULL, n_perms = 0: identity only = the uncorrected
 *               fallback of evaluate.py:313.
 *        centroid_distance = |mean_i pos_i - mean_i ref_i|         (:315)
 *        min_cross_distance = min over receptor points r, atoms i of |rec_r - pos_i|   (:331-332): rec_atom_pos [n_rec_atoms, 3] DEVICE,
 *               the receptor ATOM coordinates the reference reads from the PDB file minus original_center; NULL, 0: the C-alpha
 *               coordinates of the complex
 *        min_self_distance  = min over atom pairs i != j of |pos_i - pos_j|          (:333-335)
 *      all over the atoms with atom_mask[i] != 0 (filterHs, evaluate.py:297).  pos [B,n_lig,3], ref_pos [n_lig,3] (already minus
 *      original_center), atom_mask [n_lig] uint8 (NULL = all atoms), out [B,4]; all DEVICE pointers. */
int ddk_pose_metrics(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* pos, const float* ref_pos, const uint8_t* atom_mask,
                     const int32_t* perms, int32_t n_perms, const float* rec_atom_pos, int32_t n_rec_atoms, float* out, void* stream);

/* ---- a1-a2: the reverse-diffusion loop of sampling()  utils/sampling.py:105-198 for one batch:
 *      per step  perturb = score_coeff*score + noise_coeff*z  (coefficients are the host scalars of
 *      sampling.py:137-192, including the low-temperature variant), then ddk_se3_update.
 *      t [steps,3]; score_coeff / noise_coeff [steps,3] HOST arrays (tr,rot,tor);
 *      noise: DEVICE array [steps, B, 6 + n_rot] of N(0,1) draws (tr xyz, rot xyz, tor...), or NULL = zeros.
 *      pos [B,n_lig,3] is updated in place.  No host synchronisation inside. */
int ddk_sample(ddk_ctx* ctx, ddk_complex* cx, int32_t B, int32_t steps, const float* t, const float* score_coeff,
               const float* noise_coeff, const float* noise, float* pos, void* stream);

/* ---- graph construction alone (score_model.py:310-344 build_lig_conv_graph, :346-373 build_rec_conv_graph, :375-408
 *      build_cross_conv_graph, merged as in :218-225): for B poses of the complex at diffusion time t_tr, the ONE edge list the
 *      conv layers consume, in the reference's group order [lig-lig | lig->rec | rec-rec | rec->lig(flipped)], every group sorted
 *      by edge_src.  Row convention of tensor_layers.py:147-159: edge_src = node that RECEIVES the message (scatter index),
 *      edge_dst = node whose features enter the tensor product; node numbering [ligand atoms of all samples | residues of all
 *      samples].  lig-lig = covalent bonds + radius_graph(lig_max_radius, max_num_neighbors = 32: up to 33 kept per atom, see csrc/model.h LIG_CAP); cross cutoff = 3 sigma_tr(t_tr) + 20
 *      (dynamic_max_cross) or cross_max_distance.  edge_src_out / edge_dst_out: device [cap] int32; group_offsets_out: device [5]
 *      int32 (offsets of the four groups, [4] = E).  DDK_ERR_INVALID when cap is below the complex' worst case
 *      (ddk_last_graph_stats out[7]); no host synchronisation inside. */
int ddk_build_graph(ddk_ctx* ctx, ddk_complex* cx, int32_t B, const float* lig_pos, float t_tr, int32_t* edge_src_out,
                    int32_t* edge_dst_out, int64_t cap, int32_t* group_offsets_out, void* stream);

/* ---- introspection for tests / benches ------------------------------------------------------ */
/* Copies the last forward's edge counts into out[12] (HOST, synchronises the stream): [0..3] = E_ll, E_lr, E_rr, E_rl of the reference
 * graph, [4] = edges of the shared receptor-receptor copy (layer-0 de-duplication), [5] = E, [6] = capacity overflow flag,
 * [7] = edge capacity, [8..10] = receptor-receptor edges inside the heads' backward receptive field one / two / three layers below
 * the last conv layer (= E_rr when the pruning is off), [11] != 0: the graph's count and fill kernels disagreed about a sample's edge count in some
 * forward since the complex was created (an internal consistency guard: the edge list of that forward is not to be trusted). */
int ddk_last_graph_stats(ddk_ctx* ctx, ddk_complex* cx, int64_t* out, void* stream);
/* Node features after the conv stack of the last forward: lig [B*n_lig, 84], rec [B*n_rec, 84] (device ptrs, may be NULL).
 * rec_out != NULL requires ddk_set_keep_receptor_features(on) before that forward (DDK_ERR_STATE otherwise). */
int ddk_last_node_features(ddk_ctx* ctx, ddk_complex* cx, int32_t B, float* lig_out, float* rec_out, void* stream);

/* ---- measurement: HIP-event timing of every fused TP-conv launch on the stream it is launched on (bench.py's
 *      roofline leg).  ddk_profile_read synchronises, then fills per conv layer l (n >= 5 * num_conv_layers doubles, HOST):
 *      out[5l] = total kernel ms, out[5l+1] = launches, out[5l+2] = edges the launches evaluated, out[5l+3] = edges they would
 *      have evaluated without the receptive-field pruning (layer-0 de-duplication and the last layer's ligand-only evaluation
 *      still applied: the round-1 accounting), out[5l+4] = edges the reference evaluates in that layer (all of E every layer). */
int ddk_profile_enable(ddk_ctx* ctx, int32_t on);
int ddk_profile_read(ddk_ctx* ctx, double* out, int32_t n);
/*      The same records per FORWARD, in launch order (a sampling loop of K steps = K consecutive forwards): out[4f] = conv kernel ms
 *      of forward f (its conv launches together), out[4f+1] = edges they evaluated, out[4f+2] = edges without the receptive-field
 *      pruning, out[4f+3] = cross edges lig->rec of the forward's graph.  Returns the number of forwards recorded since
 *      ddk_profile_enable(on) (at most max_forwards are written), or a negative error code. */
int ddk_profile_read_forwards(ddk_ctx* ctx, double* out, int32_t max_forwards);

#ifdef __cplusplus
}
#endif
#endif /* DDK_H */
