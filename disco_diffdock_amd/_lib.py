"""ctypes binding of libddk.so (include/ddk.h).  There is NO fallback: if the HIP library is missing
or fails to load, importing the operators raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DDK_LIB') or os.path.join(_HERE, 'libddk.so')   # DDK_LIB: an alternative build of the library (kernel experiments)


class ddk_config(C.Structure):
    _fields_ = [('ns', C.c_int32), ('nv', C.c_int32), ('num_conv_layers', C.c_int32),
                ('sigma_embed_dim', C.c_int32), ('distance_embed_dim', C.c_int32), ('cross_distance_embed_dim', C.c_int32),
                ('lig_max_radius', C.c_float), ('rec_max_radius', C.c_float), ('cross_max_distance', C.c_float),
                ('center_max_distance', C.c_float), ('dynamic_max_cross', C.c_int32), ('embedding_scale', C.c_float),
                ('scale_by_sigma', C.c_int32), ('no_torsion', C.c_int32), ('batch_norm', C.c_int32),
                ('latent_dim', C.c_int32), ('latent_vocab', C.c_int32), ('latent_droprate', C.c_float),
                ('lm_embedding_dim', C.c_int32),
                ('tr_sigma_min', C.c_float), ('tr_sigma_max', C.c_float), ('rot_sigma_min', C.c_float),
                ('rot_sigma_max', C.c_float), ('tor_sigma_min', C.c_float), ('tor_sigma_max', C.c_float),
                ('device', C.c_int32), ('all_atoms', C.c_int32), ('num_confidence_outputs', C.c_int32),
                ('confidence_no_batchnorm', C.c_int32), ('conv_kernel', C.c_int32), ('deterministic', C.c_int32), ('confidence_mode', C.c_int32)]


class ddk_complex_desc(C.Structure):
    _fields_ = [('n_lig', C.c_int32), ('n_rec', C.c_int32), ('n_bond_edges', C.c_int32), ('n_rot', C.c_int32),
                ('n_rec_edges', C.c_int32), ('rec_feat_dim', C.c_int32),
                ('lig_x', C.c_void_p), ('bond_index', C.c_void_p), ('bond_attr', C.c_void_p), ('edge_mask', C.c_void_p),
                ('mask_rotate', C.c_void_p), ('rec_x', C.c_void_p), ('rec_pos', C.c_void_p), ('rec_edge_index', C.c_void_p)]


class ddk_atoms_desc(C.Structure):
    _fields_ = [('n_atom', C.c_int32), ('n_atom_edges', C.c_int32), ('atom_x', C.c_void_p), ('atom_pos', C.c_void_p),
                ('atom_edge_index', C.c_void_p), ('atom_rec_index', C.c_void_p)]


_lib = None

# every symbol include/ddk.h declares (tests check that the library exports all of them)
SYMBOLS = ['ddk_create', 'ddk_destroy', 'ddk_last_error', 'ddk_version', 'ddk_load_weights', 'ddk_finalize_weights',
           'ddk_set_score_norm_tables', 'ddk_tp_forward', 'ddk_conv_forward', 'ddk_complex_create', 'ddk_complex_destroy',
           'ddk_score_forward', 'ddk_se3_update', 'ddk_sample', 'ddk_last_graph_stats', 'ddk_last_node_features',
           'ddk_profile_enable', 'ddk_profile_read', 'ddk_profile_read_forwards', 'ddk_set_latents', 'ddk_set_guidance',
           'ddk_set_keep_receptor_features', 'ddk_randomize_position', 'ddk_complex_set_atoms',
           'ddk_confidence_forward', 'ddk_score_confidence', 'ddk_pose_metrics', 'ddk_build_graph', 'ddk_set_receptive_field_pruning', 'ddk_ar_logits', 'ddk_ar_decode', 'ddk_confidence_status']

# test hooks (include/ddk_debug.h): not part of the drop-in boundary
DEBUG_SYMBOLS = ['ddk_debug_export', 'ddk_debug_read_edges', 'ddk_debug_conf_counts', 'ddk_debug_conf_table', 'ddk_debug_conf_nodes', 'ddk_debug_conf_edges', 'ddk_debug_kabsch', 'ddk_debug_axis_angle', 'ddk_debug_set_layer0_dedup', 'ddk_debug_read_patch', 'ddk_debug_split3', 'ddk_debug_conv_trace', 'ddk_debug_pool_stats', 'ddk_debug_set_conv_workgroups', 'ddk_debug_set_alloc_limit']


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: build the HIP extension first (python -m disco_diffdock_amd.build). '
                           'disco_diffdock_amd has no CPU fallback.')
    import torch  # noqa: F401  (first: libddk.so must bind to the HIP runtime PyTorch-ROCm ships; loaded before torch it binds to /opt/rocm's copy,
    #                and two HIP runtimes in one process leave the second without a device: "no ROCm-capable device is detected")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.ddk_create.argtypes = [C.POINTER(ddk_config), C.POINTER(vp)]
    L.ddk_destroy.argtypes = [vp]
    L.ddk_destroy.restype = None
    L.ddk_last_error.argtypes = [vp]
    L.ddk_last_error.restype = C.c_char_p
    L.ddk_version.restype = C.c_char_p
    L.ddk_load_weights.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32]
    L.ddk_finalize_weights.argtypes = [vp]
    L.ddk_set_score_norm_tables.argtypes = [vp, vp, i32, vp, i32]
    L.ddk_tp_forward.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp]
    L.ddk_conv_forward.argtypes = [vp, i32, vp, i64, vp, vp, C.POINTER(i64), vp, vp, vp, vp]
    L.ddk_complex_create.argtypes = [vp, C.POINTER(ddk_complex_desc), i32, C.POINTER(vp)]
    L.ddk_complex_destroy.argtypes = [vp, vp]
    L.ddk_complex_destroy.restype = None
    L.ddk_score_forward.argtypes = [vp, vp, i32, vp, f32, f32, f32, vp, vp, vp, vp]
    L.ddk_se3_update.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.ddk_randomize_position.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.ddk_complex_set_atoms.argtypes = [vp, vp, vp, vp, vp, i32]
    L.ddk_confidence_forward.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ddk_score_confidence.argtypes = [vp, vp, i32, vp, C.c_float, C.c_float, C.c_float, vp, vp]
    L.ddk_pose_metrics.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp]
    L.ddk_sample.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.ddk_last_graph_stats.argtypes = [vp, vp, vp, vp]
    L.ddk_build_graph.argtypes = [vp, vp, C.c_int32, vp, C.c_float, vp, vp, C.c_int64, vp, vp]
    L.ddk_last_node_features.argtypes = [vp, vp, i32, vp, vp, vp]
    L.ddk_set_latents.argtypes = [vp, vp, vp, vp, f32]
    L.ddk_set_guidance.argtypes = [vp, vp, f32, f32, f32]
    L.ddk_set_keep_receptor_features.argtypes = [vp, vp, i32]
    L.ddk_profile_enable.argtypes = [vp, i32]
    L.ddk_set_receptive_field_pruning.argtypes = [vp, i32]
    L.ddk_ar_logits.argtypes = [vp, vp, i32, vp, vp]
    L.ddk_confidence_status.argtypes = [vp, vp, vp, vp]
    L.ddk_ar_decode.argtypes = [vp, vp, i32, vp, f32, vp, i32, i32, vp, vp, vp, vp]
    L.ddk_profile_read.argtypes = [vp, vp, i32]
    L.ddk_profile_read_forwards.argtypes = [vp, vp, i32]
    L.ddk_debug_export.argtypes = [vp, C.c_char_p, vp, i64]
    L.ddk_debug_export.restype = i64
    _declare_debug(L)
    _lib = L
    return L


def _declare_debug(L):
    import ctypes as C
    L.ddk_debug_read_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64]
    L.ddk_debug_kabsch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddk_debug_axis_angle.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddk_debug_set_layer0_dedup.argtypes = [C.c_void_p, C.c_int32]
    L.ddk_debug_read_patch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.ddk_debug_split3.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddk_debug_conv_trace.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.ddk_debug_pool_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ddk_debug_set_conv_workgroups.argtypes = [C.c_void_p, C.c_int32]
    L.ddk_debug_set_alloc_limit.argtypes = [C.c_void_p, C.c_int64]
    L.ddk_debug_conf_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ddk_debug_conf_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.ddk_debug_conf_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.ddk_debug_conf_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
