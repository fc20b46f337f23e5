/* ddk_debug.h — test hooks of libddk.so.  NOT part of the drop-in boundary (include/ddk.h): these entry points copy internal
 * device arrays to HOST buffers for the parity tests, synchronise the device, and may change between rounds.  Nothing in the
 * product path (sampling(), bench.py's timed region) calls them. */
#ifndef DDK_DEBUG_H
#define DDK_DEBUG_H

#include "ddk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Packed host-side arrays of a context ("conv.<l>.w1p.<g>", "conv.<l>.w2p.<g>", "conv.<l>.tiles", "conv.<l>.bn_scale", "conv.<l>.w2x", ...):
 * returns the number of 32-bit words of the item (buf may be NULL to query) or a negative ddk_status. */
int64_t ddk_debug_export(ddk_ctx* ctx, const char* what, void* buf, int64_t cap_words);

/* Edge arrays of the last score-model forward of `cx` (counts via ddk_last_graph_stats): src/dst [n] int32, emb [n,24],
 * sh [n,4], deg [n_nodes] int32; HOST pointers, any may be NULL. */
int ddk_debug_read_edges(ddk_ctx* ctx, ddk_complex* cx, int64_t n, int32_t* src, int32_t* dst, float* emb, float* sh, int32_t* deg,
                         int64_t n_nodes);

/* Confidence model (conf.hip), last ddk_confidence_forward of `cx`:
 *   counts: out[0..8] = edges of the nine groups [ll lr la aa al ar rr rl ra], out[9] = ligand-atom edge capacity overflow flag;
 *   nodes:  x [n, 84] features after the conv stack and deg [n, 3] per-slot in-degrees, n = max_batch * n_lig + (max_batch + 1) * (n_atom + n_rec) (atom / residue sample max_batch = the virtual ligand-free sample of conf.hip);
 *   edges:  group table gt[18] (begin[9], end[9]) when gt != NULL, else n edges from `first`: src, dst, emb [n,24], sh [n,4]. */
int ddk_debug_conf_counts(ddk_ctx* ctx, ddk_complex* cx, int32_t* out);
/* ... and the group table one of its layers ran on: which = 0 the full table, 1 layer 0 (static groups on the virtual sample), 2 / 3 the level-A / level-B
 * tables of the second- / third-to-last layer, 4 layer 1 (static groups: the receivers whose messages differ from the virtual sample's + the virtual sample);
 * out[0..8] = edges per group.  Tables the last forward did not build hold stale or zero counts. */
int ddk_debug_conf_table(ddk_ctx* ctx, ddk_complex* cx, int32_t which, int32_t* out);
int ddk_debug_conf_nodes(ddk_ctx* ctx, ddk_complex* cx, float* x, int32_t* deg, int64_t n);
int ddk_debug_conf_edges(ddk_ctx* ctx, ddk_complex* cx, int64_t first, int64_t n, int32_t* src, int32_t* dst, float* emb, float* sh,
                         int32_t* gt);

/* Layer-0 de-duplication of the receptor-receptor messages (one evaluation per batch; for the latent-conditioned model plus the per-sample
 * patch group of the receivers that see a non-zero latent): on by default, `on = 0` makes every sample evaluate all its messages.  Exists
 * for the equality test of the two paths. */
int ddk_debug_set_layer0_dedup(ddk_ctx* ctx, int32_t on);

/* Device-chunk pool of the complexes (ddk_complex_create / destroy never synchronise: chunks whose previous owner has finished are reused,
 * otherwise hipMalloc): out[8] = hipMalloc calls, reuses, hipFree calls, bytes parked in the pool, chunks parked, bytes owned by live
 * complexes, the peak of that, device bytes the context holds in all (weights + workspaces + live and parked chunks). */
int ddk_debug_pool_stats(ddk_ctx* ctx, int64_t* out);
/* Cap on the device memory this context may hold (0 = none): a hipMalloc that would take it beyond `bytes` fails exactly like hipErrorOutOfMemory
 * (after the pool has handed its parked chunks back, as under real memory pressure): ddk_complex_create / ddk_conv_forward return DDK_ERR_NOMEM with
 * ddk_last_error naming the request, the context and its pool stay usable, a smaller batch then succeeds.  The reference's recovery path
 * (evaluate.py:228-231, 394-398: halve the batch and retry) is tested through it. */
int ddk_debug_set_alloc_limit(ddk_ctx* ctx, int64_t bytes);

/* Persistent workgroups of this context's conv launches (default: one per CU).  Experiments with two contexts / streams side by side. */
int ddk_debug_set_conv_workgroups(ddk_ctx* ctx, int32_t n);
/* The patch group of the last forward of a latent-conditioned model: counts[B + 1] = exclusive prefix of the patch edges per sample
 * (counts[B] = total), mask[B * n_rec] = 1 for receivers whose rec-rec sum comes from the patch group.  HOST pointers; synchronises. */
int ddk_debug_read_patch(ddk_ctx* ctx, ddk_complex* cx, int32_t B, int32_t* counts, uint8_t* mask);

/* The device routines of csrc/k_se3.hip on caller-supplied DEVICE arrays (enqueue on `stream`, no synchronisation):
 *   kabsch:     A, B [nb, n, 3] -> R [nb, 3, 3], t [nb, 3] with R a + t ~ b   (utils/geometry.py:126-156, reflection case included)
 *   axis_angle: aa [n, 3] -> R [n, 3, 3]                                     (utils/geometry.py:71-85, small-angle branch included) */
int ddk_debug_kabsch(ddk_ctx* ctx, int32_t nb, int32_t n, const float* A, const float* B, float* R_out, float* t_out, void* stream);
int ddk_debug_axis_angle(ddk_ctx* ctx, int32_t n, const float* aa, float* R_out, void* stream);

/* Timeline of the default conv kernel (k_conv_x.hip): conv layer `layer` of the following score-model forwards runs the kernel's TRACE
 * instantiation, whose workgroup 0 stamps s_memtime at the four edges of every tile's two half phases into trace (DEVICE,
 * [8 waves][1024 tiles][8] uint32: burst start, burst end, epilogue start, epilogue end, then four stamps inside the burst: before K step 0, 1, 2, 3; tools/conv_trace.py).  trace = NULL: off. */
/* layer = 100 + l: one record per UNIT instead of per tile (slots 4-7 as above, 0 = tile loop done, 1 = tiles of the unit, 2 = unit handed over):
 * no stamp inside the tile loop, undisturbed cycles per tile. */
int ddk_debug_conv_trace(ddk_ctx* ctx, int32_t layer, uint32_t* trace);

/* The default conv kernel's limb split (k_conv_x.hip) on a DEVICE array x [n], cut into groups of `group` consecutive values that share one
 * power-of-two range scale (the kernel scales per edge): hi / mid / lo [n] = the fp16 limbs as fp32, scale [n] = the group's scale;
 * x * scale == hi + mid + lo bit for bit for every value within 2^-15 of its group's maximum (|x * scale| >= 0.5), within 2^-25 below. */
int ddk_debug_split3(ddk_ctx* ctx, const float* x, int64_t n, int32_t group, float* hi, float* mid, float* lo, float* scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDK_DEBUG_H */
