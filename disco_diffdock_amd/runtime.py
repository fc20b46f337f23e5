"""Thin Python owner of a ddk context (include/ddk.h): config mapping, checkpoint upload, operator calls.
PyTorch is used only for device memory and streams; all compute happens in libddk.so."""
import ctypes as C
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')

DEFAULTS = dict(ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32, distance_embed_dim=32, cross_distance_embed_dim=32,
                lig_max_radius=5.0, rec_max_radius=30.0, cross_max_distance=80.0, center_max_distance=30.0,
                dynamic_max_cross=1, embedding_scale=1000.0, scale_by_sigma=1, no_torsion=0, batch_norm=1,
                latent_dim=0, latent_vocab=0, latent_droprate=0.0, lm_embedding_dim=1280,
                tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55,
                tor_sigma_min=0.03, tor_sigma_max=3.14, device=0, all_atoms=0, num_confidence_outputs=1, confidence_no_batchnorm=0,
                conv_kernel=0, deterministic=0, confidence_mode=0)


def config_from_args(args, device=0):
    """model_parameters.yml Namespace -> ddk_config fields, the mapping of get_model
    (reference utils/model_utils.py:39-68) plus the constructor defaults it leaves alone."""
    g = lambda k, d: getattr(args, k, d)
    d = dict(DEFAULTS)
    d.update(ns=args.ns, nv=args.nv, num_conv_layers=args.num_conv_layers, sigma_embed_dim=args.sigma_embed_dim,
             distance_embed_dim=args.distance_embed_dim, cross_distance_embed_dim=args.cross_distance_embed_dim,
             lig_max_radius=float(args.max_radius), cross_max_distance=float(args.cross_max_distance),
             dynamic_max_cross=int(bool(args.dynamic_max_cross)), embedding_scale=float(args.embedding_scale),
             scale_by_sigma=int(bool(args.scale_by_sigma)), no_torsion=int(bool(args.no_torsion)),
             batch_norm=int(not args.no_batch_norm), latent_dim=int(g('latent_dim', 0)),
             latent_vocab=int(g('latent_vocab', 0)), latent_droprate=float(g('latent_droprate', 0.0)),
             lm_embedding_dim=1280 if g('esm_embeddings_path', None) is not None else 0,
             tr_sigma_min=args.tr_sigma_min, tr_sigma_max=args.tr_sigma_max, rot_sigma_min=args.rot_sigma_min,
             rot_sigma_max=args.rot_sigma_max, tor_sigma_min=args.tor_sigma_min, tor_sigma_max=args.tor_sigma_max,
             device=device)
    if g('sh_lmax', 2) != 1 or g('use_second_order_repr', False):
        raise RuntimeError('ddk implements the sh_lmax=1, first-order (FasterTensorProduct) score model only')
    return d


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_side_streams = {}


def h2d_async(t, device):
    """Host tensor -> device WITHOUT waiting for the work already queued on the current stream: the copy runs from pinned memory on
    a side stream and the current stream waits for it (a plain ``t.to(device)`` of pageable memory blocks the host until the copy
    has run, i.e. until the sampling loop of the previous complex has drained).  Device tensors pass through."""
    device = torch.device(device)
    if t.is_cuda:
        return t if t.device == device else t.to(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _side_streams.get(idx)
    if side is None:
        side = _side_streams[idx] = torch.cuda.Stream(device=idx)
    pinned = t.contiguous().pin_memory()
    cur = torch.cuda.current_stream(idx)
    with torch.cuda.stream(side):
        d = pinned.to(device, non_blocking=True)
    cur.wait_stream(side)
    d.record_stream(cur)
    return d


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError('ddk: device tensors only (no CPU path exists); got a tensor on ' + str(t.device))
    return t


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """One ddk_ctx (one device).  device=-1 gives a host-only context (weight packing only, for CPU tests)."""

    def __init__(self, device=0, **cfg):
        self.L = _lib.lib()
        d = dict(DEFAULTS)
        if os.environ.get('DDK_CONV_KERNEL'):     # 1: the fp32-MFMA conv kernel (the fallback), 3: the three-limb / six-product form, for every context of this process
            d['conv_kernel'] = int(os.environ['DDK_CONV_KERNEL'])
        if os.environ.get('DDK_DETERMINISTIC'):   # select the deterministic scatter for every context of this process
            d['deterministic'] = int(os.environ['DDK_DETERMINISTIC'])
        d.update(cfg)
        if d.get('all_atoms'):
            d['deterministic'] = int(cfg.get('deterministic', 0))      # the env switch applies to score-model contexts only
        d['device'] = device
        self.cfg = SimpleNamespace(**d)
        c = _lib.ddk_config(**d)
        self.h = C.c_void_p()
        rc = self.L.ddk_create(C.byref(c), C.byref(self.h))
        if rc != 0:
            msg = self.L.ddk_last_error(self.h).decode() if self.h else 'ddk_create failed'
            raise RuntimeError(f'ddk_create: {msg}')
        self.device = device
        self._tables_set = False

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f'{what}: {self.L.ddk_last_error(self.h).decode()} (rc={rc})')

    def close(self):
        if getattr(self, 'h', None):
            self.L.ddk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- checkpoint --------------------------------------------------------------------------
    def load_state_dict(self, state_dict, prefix='', finalize=True):
        """state_dict with the reference's key names (score_model.state_dict(), evaluate.py:169-171)."""
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.is_floating_point():
                continue
            a = np.ascontiguousarray(v.detach().cpu().float().numpy())
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(self.L.ddk_load_weights(self.h, (prefix + k).encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                        f'ddk_load_weights({k})')
        if finalize:
            self.finalize()

    def finalize(self):
        if not self._tables_set:
            self.set_tables(np.load(os.path.join(_DATA, 'so3_exp_score_norms.npy')),
                            np.load(os.path.join(_DATA, 'torus_score_norm_seed0.npy')))
        self._check(self.L.ddk_finalize_weights(self.h), 'ddk_finalize_weights')

    def set_tables(self, so3, torus):
        so3 = np.ascontiguousarray(so3, dtype=np.float64)
        torus = np.ascontiguousarray(torus, dtype=np.float64)
        self._check(self.L.ddk_set_score_norm_tables(self.h, so3.ctypes.data_as(C.c_void_p), len(so3),
                                                     torus.ctypes.data_as(C.c_void_p), len(torus)), 'ddk_set_score_norm_tables')
        self._tables_set = True

    def export(self, what, dtype=np.float32):
        n = self.L.ddk_debug_export(self.h, what.encode(), None, 0)
        if n < 0:
            raise RuntimeError(f'ddk_debug_export({what}): {self.L.ddk_last_error(self.h).decode()}')
        buf = np.zeros(n, dtype=dtype)
        self.L.ddk_debug_export(self.h, what.encode(), buf.ctypes.data_as(C.c_void_p), n)
        return buf

    # ---- measurement -------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.L.ddk_profile_enable(self.h, int(on)), 'ddk_profile_enable')

    def profile_read(self):
        """per conv layer: kernel ms, launches, edges evaluated, edges without the receptive-field pruning, reference edges"""
        n = 5 * self.cfg.num_conv_layers
        buf = (C.c_double * n)()
        self._check(self.L.ddk_profile_read(self.h, buf, n), 'ddk_profile_read')
        a = np.array(list(buf)).reshape(-1, 5)
        return [dict(ms=float(r[0]), launches=int(r[1]), edges=int(r[2]), edges_unpruned=int(r[3]), edges_reference=int(r[4])) for r in a]

    def profile_read_forwards(self, max_forwards=32768):
        """per forward since profile_enable(True), in launch order: array [n, 4] = conv kernel ms, edges evaluated, edges without the
        receptive-field pruning, cross edges of the forward's graph"""
        buf = (C.c_double * (4 * max_forwards))()
        n = self.L.ddk_profile_read_forwards(self.h, buf, max_forwards)
        if n < 0:
            raise RuntimeError(f'ddk_profile_read_forwards: {self.L.ddk_last_error(self.h).decode()}')
        if n >= max_forwards:
            raise RuntimeError(f'ddk_profile_read_forwards: {n} forwards recorded, the profile buffer holds {max_forwards}: later launches were not timed')
        return np.array(buf[:4 * min(n, max_forwards)]).reshape(-1, 4)

    def set_pruning(self, on=True):
        """backward receptive-field pruning of the receptor-receptor messages (default on; exact)"""
        self._check(self.L.ddk_set_receptive_field_pruning(self.h, int(bool(on))), 'ddk_set_receptive_field_pruning')

    def pool_stats(self):
        """device-chunk pool of the complexes (include/ddk_debug.h): hipMalloc calls, reuses, hipFree calls, bytes / chunks parked, bytes owned now / at peak"""
        buf = (C.c_int64 * 8)()
        self._check(self.L.ddk_debug_pool_stats(self.h, buf), 'ddk_debug_pool_stats')
        keys = ('hipMalloc_calls', 'reuses', 'hipFree_calls', 'bytes_parked', 'chunks_parked', 'bytes_owned', 'bytes_owned_peak', 'device_bytes_held')
        return dict(zip(keys, [int(v) for v in buf[:8]]))

    def debug_set_alloc_limit(self, nbytes=0):
        """test hook (include/ddk_debug.h): cap on the device memory this context may hold (0 = none); a request beyond it fails like a real out-of-memory"""
        self._check(self.L.ddk_debug_set_alloc_limit(self.h, int(nbytes)), 'ddk_debug_set_alloc_limit')

    def debug_set_layer0_dedup(self, on=True):
        """test hook (include/ddk_debug.h): layer-0 de-duplication of the rec-rec messages on / off"""
        self._check(self.L.ddk_debug_set_layer0_dedup(self.h, int(bool(on))), 'ddk_debug_set_layer0_dedup')

    # ---- operators ---------------------------------------------------------------------------
    def tp_forward(self, layer, x_dst, sh, w, dout):
        x_dst, sh, w = x_dst.contiguous().float(), sh.contiguous().float(), w.contiguous().float()
        E = x_dst.shape[0]
        out = torch.empty((E, dout), dtype=torch.float32, device=x_dst.device)
        self._check(self.L.ddk_tp_forward(self.h, layer, _ptr(x_dst), _ptr(sh), _ptr(w), E, _ptr(out), _stream()), 'ddk_tp_forward')
        return out

    def conv_forward(self, layer, x, edge_src, edge_dst, group_offsets, edge_attr, sh, dout):
        x, edge_attr, sh = x.contiguous().float(), edge_attr.contiguous().float(), sh.contiguous().float()
        edge_src, edge_dst = edge_src.contiguous().int(), edge_dst.contiguous().int()
        N = x.shape[0]
        out = torch.empty((N, dout), dtype=torch.float32, device=x.device)
        go = (C.c_int64 * 5)(*[int(v) for v in group_offsets])
        self._check(self.L.ddk_conv_forward(self.h, layer, _ptr(x), N, _ptr(edge_src), _ptr(edge_dst), go, _ptr(edge_attr),
                                            _ptr(sh), _ptr(out), _stream()), 'ddk_conv_forward')
        return out


class Complex:
    """Device-resident static data of one complex (ddk_complex): topology, receptor embedding without its
    sigma part, receptor-receptor geometry, and the per-forward workspaces for up to max_batch samples."""

    def __init__(self, ctx, c, max_batch):
        """c: dict with the arrays of SURVEY.md Appendix B.1 (numpy or torch): lig_x [n,16], bond_index [2,M],
        bond_attr [M,4], edge_mask [M], mask_rotate [R,n], rec_x [n_rec,1+lm], rec_pos [n_rec,3], rec_edge_index [2,E]."""
        self.ctx = ctx
        f = lambda a, dt: np.ascontiguousarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=dt)
        self.arr = dict(lig_x=f(c['lig_x'], np.int32), bond_index=f(c['bond_index'], np.int32), bond_attr=f(c['bond_attr'], np.float32),
                        edge_mask=f(c['edge_mask'], np.uint8), mask_rotate=f(c['mask_rotate'], np.uint8).reshape(-1, len(c['lig_x'])),
                        rec_x=f(c['rec_x'], np.float32), rec_pos=f(c['rec_pos'], np.float32),
                        rec_edge_index=f(c['rec_edge_index'], np.int32))
        a = self.arr
        self.n_lig, self.n_rec = a['lig_x'].shape[0], a['rec_pos'].shape[0]
        self.M, self.R, self.E_rr = a['bond_index'].shape[1], a['mask_rotate'].shape[0], a['rec_edge_index'].shape[1]
        self.max_batch = max_batch
        d = _lib.ddk_complex_desc(n_lig=self.n_lig, n_rec=self.n_rec, n_bond_edges=self.M, n_rot=self.R, n_rec_edges=self.E_rr,
                                  rec_feat_dim=a['rec_x'].shape[1],
                                  **{k: v.ctypes.data_as(C.c_void_p) for k, v in a.items()})
        self.h = C.c_void_p()
        rc = ctx.L.ddk_complex_create(ctx.h, C.byref(d), max_batch, C.byref(self.h))
        if rc != 0:
            msg = ctx.L.ddk_last_error(ctx.h).decode()
            if self.h:
                ctx.L.ddk_complex_destroy(ctx.h, self.h)
                self.h = None
            raise RuntimeError(f'ddk_complex_create: {msg}')

    def close(self):
        if getattr(self, 'h', None) and getattr(self.ctx, 'h', None):
            self.ctx.L.ddk_complex_destroy(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_read_patch(self, B):
        """test hook: (exclusive prefix of the layer-0 patch edges per sample [B + 1], receiver mask [B, n_rec]) of the last forward"""
        cnt, mask = np.zeros(B + 1, np.int32), np.zeros((B, self.n_rec), np.uint8)
        self.ctx._check(self.ctx.L.ddk_debug_read_patch(self.ctx.h, self.h, B, cnt.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p)),
                        'ddk_debug_read_patch')
        return cnt, mask

    def set_latents(self, lig_latent=None, rec_latent=None, unconditional=0.0):
        """data['ligand'|'receptor'].latent_h of the batch ([B*n, latent_dim], device) for the following forwards."""
        if lig_latent is not None:
            lig_latent, rec_latent = lig_latent.contiguous().float(), rec_latent.contiguous().float()
            ld = int(self.ctx.cfg.latent_dim)
            nb = lig_latent.shape[0] // max(self.n_lig, 1)
            if (lig_latent.dim() != 2 or rec_latent.dim() != 2 or lig_latent.shape[1] != ld or rec_latent.shape[1] != ld or nb < 1
                    or lig_latent.shape[0] != nb * self.n_lig or rec_latent.shape[0] != nb * self.n_rec):
                raise RuntimeError(f'ddk: latent arrays must be [B*{self.n_lig}, {ld}] and [B*{self.n_rec}, {ld}] for one batch size B, got '
                                   f'{tuple(lig_latent.shape)} and {tuple(rec_latent.shape)} (the library reads them through raw pointers)')
        self._latents = (lig_latent, rec_latent)      # keep the device arrays alive
        self.ctx._check(self.ctx.L.ddk_set_latents(self.ctx.h, self.h, _ptr(lig_latent), _ptr(rec_latent), float(unconditional)),
                        'ddk_set_latents')

    def set_guidance(self, weight=0.0, cfg_start=1.0, cfg_end=0.0):
        self.ctx._check(self.ctx.L.ddk_set_guidance(self.ctx.h, self.h, float(weight), float(cfg_start), float(cfg_end)), 'ddk_set_guidance')

    # ---- model.score_model(batch) ------------------------------------------------------------------
    def score_forward(self, pos, t_tr, t_rot, t_tor):
        ctx = self.ctx
        pos = _need_cuda(pos).contiguous().float().reshape(-1, self.n_lig, 3)
        B = pos.shape[0]
        dev = pos.device
        tr = torch.empty((B, 3), dtype=torch.float32, device=dev)
        rot = torch.empty((B, 3), dtype=torch.float32, device=dev)
        tor = torch.empty((B * self.R,), dtype=torch.float32, device=dev)
        ctx._check(ctx.L.ddk_score_forward(ctx.h, self.h, B, _ptr(pos), float(t_tr), float(t_rot), float(t_tor),
                                           _ptr(tr), _ptr(rot), _ptr(tor), _stream()), 'ddk_score_forward')
        return tr, rot, tor

    def se3_update(self, pos, tr, rot, tor):
        ctx = self.ctx
        pos = pos.contiguous().float().reshape(-1, self.n_lig, 3)
        B = pos.shape[0]
        out = torch.empty_like(pos)
        tr, rot = tr.contiguous().float(), rot.contiguous().float()
        tor = tor.contiguous().float() if tor is not None else None
        ctx._check(ctx.L.ddk_se3_update(ctx.h, self.h, B, _ptr(pos), _ptr(tr), _ptr(rot), _ptr(tor), _ptr(out), _stream()),
                   'ddk_se3_update')
        return out

    # ---- AR latent model (ddk_ar_logits / ddk_ar_decode) --------------------------------------------------------------------
    def ar_logits(self, B):
        """predictor logits [B, n_lig + n_rec] on the node features of the last forward (keep_receptor_features(True) before it)"""
        dev = torch.device('cuda', self.ctx.device)
        out = torch.empty((B, self.n_lig + self.n_rec), dtype=torch.float32, device=dev)
        self.ctx._check(self.ctx.L.ddk_ar_logits(self.ctx.h, self.h, B, _ptr(out), _stream()), 'ddk_ar_logits')
        return out

    def ar_decode(self, logits, temperature, uniforms, idx, latent_l, latent_r, choices=None):
        """in place: one-hot of the picked node of every graph into column idx of latent_l / latent_r ([B*n, D], contiguous fp32)"""
        B, D = logits.shape[0], latent_l.shape[1]
        assert logits.is_contiguous() and latent_l.is_contiguous() and latent_r.is_contiguous() and latent_l.dtype == torch.float32
        assert choices is None or (choices.dtype == torch.int32 and choices.is_contiguous() and tuple(choices.shape) == (B, D))
        self.ctx._check(self.ctx.L.ddk_ar_decode(self.ctx.h, self.h, B, _ptr(logits), float(temperature), _ptr(uniforms), int(idx), D,
                                                 _ptr(latent_l), _ptr(latent_r), _ptr(choices), _stream()), 'ddk_ar_decode')

    def set_atoms(self, atom_x, atom_pos, atom_edge_index, atom_rec_index, lig_x=None, rec_x=None):
        """Receptor-atom level of the confidence model's graph (data['atom'] of datasets_utils/process_mols.py:474-477)."""
        f = lambda a, dt: np.ascontiguousarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, dtype=dt)
        at = dict(atom_x=f(atom_x, np.int32), atom_pos=f(atom_pos, np.float32), atom_edge_index=f(atom_edge_index, np.int32),
                  atom_rec_index=f(atom_rec_index, np.int32))
        self.n_atom = at['atom_x'].shape[0]
        d = _lib.ddk_atoms_desc(n_atom=self.n_atom, n_atom_edges=at['atom_edge_index'].shape[1],
                                **{k: v.ctypes.data_as(C.c_void_p) for k, v in at.items()})
        lig_x = self.arr['lig_x'] if lig_x is None else f(lig_x, np.int32)
        rec_x = self.arr['rec_x'] if rec_x is None else f(rec_x, np.float32)
        self.ctx._check(self.ctx.L.ddk_complex_set_atoms(self.ctx.h, self.h, C.byref(d), lig_x.ctypes.data_as(C.c_void_p),
                                                         rec_x.ctypes.data_as(C.c_void_p), rec_x.shape[1]), 'ddk_complex_set_atoms')

    def confidence_forward(self, pos, check=True):
        """confidence_model(batch) for B poses of this complex -> [B, num_confidence_outputs] (device).  check=False skips the host
        read-back of the edge-capacity flag (the caller runs ``confidence_counts()`` at its own synchronisation point)."""
        pos = pos.contiguous().float().reshape(-1, self.n_lig, 3)
        B = pos.shape[0]
        out = torch.empty((B, int(self.ctx.cfg.num_confidence_outputs)), dtype=torch.float32, device=pos.device)
        self.ctx._check(self.ctx.L.ddk_confidence_forward(self.ctx.h, self.h, B, _ptr(pos), _ptr(out), _stream()), 'ddk_confidence_forward')
        if check:
            self.confidence_counts()      # one host sync per confidence batch: fails loudly if the ligand-atom edge capacity overflowed
        return out

    def score_confidence(self, pos, t_tr, t_rot, t_tor):
        """coarse-grained confidence model (ddk_config.confidence_mode) on B poses of this complex at the given complex_t -> [B, num_confidence_outputs]
        (device); utils/sampling.py:239-240."""
        pos = pos.contiguous().float().reshape(-1, self.n_lig, 3)
        B = pos.shape[0]
        out = torch.empty((B, int(self.ctx.cfg.num_confidence_outputs)), dtype=torch.float32, device=pos.device)
        self.ctx._check(self.ctx.L.ddk_score_confidence(self.ctx.h, self.h, B, _ptr(pos), float(t_tr), float(t_rot), float(t_tor), _ptr(out), _stream()),
                        'ddk_score_confidence')
        return out

    def confidence_status_async(self):
        """-> pinned int32[20] that holds the group table / overflow flag of the last confidence forward once the current stream has
        passed this point (ddk_confidence_status); no synchronisation here"""
        out = torch.empty(20, dtype=torch.int32, pin_memory=True)
        self.ctx._check(self.ctx.L.ddk_confidence_status(self.ctx.h, self.h, C.c_void_p(out.data_ptr()), _stream()), 'ddk_confidence_status')
        return out

    def confidence_counts(self):
        out = (C.c_int32 * 10)()
        self.ctx._check(self.ctx.L.ddk_debug_conf_counts(self.ctx.h, self.h, out), 'ddk_debug_conf_counts')
        v = list(out)
        if v[9]:
            raise RuntimeError('ddk: ligand-atom edge capacity overflow')
        return dict(zip(('ll', 'lr', 'la', 'aa', 'al', 'ar', 'rr', 'rl', 'ra'), v[:9]))

    def confidence_table(self, which):
        """test hook: edges per group of the table layer `which` ran on (0 full, 1 layer 0, 2 level A, 3 level B, 4 layer 1; include/ddk_debug.h)."""
        out = (C.c_int32 * 9)()
        self.ctx._check(self.ctx.L.ddk_debug_conf_table(self.ctx.h, self.h, which, out), 'ddk_debug_conf_table')
        return dict(zip(('ll', 'lr', 'la', 'aa', 'al', 'ar', 'rr', 'rl', 'ra'), list(out)))

    def confidence_nodes(self):
        n = self.max_batch * self.n_lig + (self.max_batch + 1) * (self.n_atom + self.n_rec)      # (+ the virtual ligand-free sample)
        x, deg = np.zeros((n, 84), np.float32), np.zeros((n, 3), np.int32)
        self.ctx._check(self.ctx.L.ddk_debug_conf_nodes(self.ctx.h, self.h, x.ctypes.data_as(C.c_void_p), deg.ctypes.data_as(C.c_void_p), n), 'ddk_debug_conf_nodes')
        return x, deg

    def confidence_edges(self):
        """test hook: {group: (src, dst, emb, sh)} of the last confidence forward (node ids in the device numbering)."""
        gt = (C.c_int32 * 18)()
        self.ctx._check(self.ctx.L.ddk_debug_conf_edges(self.ctx.h, self.h, 0, 0, None, None, None, None, gt), 'ddk_debug_conf_edges')
        out = {}
        for k, name in enumerate(('ll', 'lr', 'la', 'aa', 'al', 'ar', 'rr', 'rl', 'ra')):
            n = gt[9 + k] - gt[k]
            src, dst = np.zeros(n, np.int32), np.zeros(n, np.int32)
            emb, sh = np.zeros((n, 24), np.float32), np.zeros((n, 4), np.float32)
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            self.ctx._check(self.ctx.L.ddk_debug_conf_edges(self.ctx.h, self.h, gt[k], n, p(src), p(dst), p(emb), p(sh), None), 'ddk_debug_conf_edges')
            out[name] = (src, dst, emb, sh)
        return out

    def lig_node_features(self, B, device):
        lig = torch.empty((B * self.n_lig, 84), dtype=torch.float32, device=device)
        self.ctx._check(self.ctx.L.ddk_last_node_features(self.ctx.h, self.h, B, _ptr(lig), None, _stream()), 'ddk_last_node_features')
        return lig

    def pose_metrics(self, pos, ref_pos, atom_mask=None, perms=None, rec_atom_pos=None):
        """evaluate.py:297-338 for B poses: tensor [B,4] = rmsd, centroid distance, min cross distance, min self distance.
        perms [K, n_lig] (int): graph automorphisms of the ligand -> symmetry-corrected RMSD (evaluate.py:308-310); None: uncorrected
        (:313).  rec_atom_pos [n, 3]: receptor atom coordinates for the cross distance (default: the C-alpha coordinates)."""
        pos = pos.contiguous().float().reshape(-1, self.n_lig, 3)
        ref = ref_pos.contiguous().float().reshape(self.n_lig, 3).to(pos.device)
        m = None if atom_mask is None else atom_mask.to(pos.device).to(torch.uint8).contiguous()
        pm = None if perms is None else torch.as_tensor(perms).to(pos.device).to(torch.int32).reshape(-1, self.n_lig).contiguous()
        if pm is not None and (int(pm.min()) < 0 or int(pm.max()) >= self.n_lig):
            raise RuntimeError('ddk: permutation table entries must be ligand atom indices')
        ra = None if rec_atom_pos is None else torch.as_tensor(rec_atom_pos).to(pos.device).float().reshape(-1, 3).contiguous()
        out = torch.empty((pos.shape[0], 4), dtype=torch.float32, device=pos.device)
        self.ctx._check(self.ctx.L.ddk_pose_metrics(self.ctx.h, self.h, pos.shape[0], _ptr(pos), _ptr(ref), _ptr(m), _ptr(pm),
                                                    0 if pm is None else pm.shape[0], _ptr(ra), 0 if ra is None else ra.shape[0], _ptr(out),
                                                    _stream()), 'ddk_pose_metrics')
        return out

    def randomize_position(self, pos0, rot, tor=None, tr=None):
        """utils/sampling.py:12-34 for B = rot.shape[0] copies of the conformer pos0 [n_lig,3]; returns [B,n_lig,3]."""
        ctx = self.ctx
        pos0 = pos0.contiguous().float().reshape(self.n_lig, 3)
        rot = rot.contiguous().float().reshape(-1, 3, 3)
        B = rot.shape[0]
        tor = tor.contiguous().float().reshape(B, self.R) if tor is not None and self.R > 0 else None
        tr = tr.contiguous().float().reshape(B, 3) if tr is not None else None
        out = torch.empty((B, self.n_lig, 3), dtype=torch.float32, device=pos0.device)
        ctx._check(ctx.L.ddk_randomize_position(ctx.h, self.h, B, _ptr(pos0), _ptr(tor), _ptr(rot), _ptr(tr), _ptr(out), _stream()),
                   'ddk_randomize_position')
        return out

    def sample(self, pos, t, score_coeff, noise_coeff, noise=None):
        """in-place reverse diffusion of pos [B,n_lig,3]; t/score_coeff/noise_coeff: [steps,3] host arrays."""
        ctx = self.ctx
        assert pos.is_contiguous() and pos.dtype == torch.float32
        B = pos.numel() // (self.n_lig * 3)
        t = np.ascontiguousarray(t, dtype=np.float32)
        sc = np.ascontiguousarray(score_coeff, dtype=np.float32)
        nc = np.ascontiguousarray(noise_coeff, dtype=np.float32)
        steps = t.shape[0]
        self.keep_receptor_features(False)   # the sampler reads ligand rows only
        if noise is not None:
            noise = noise.contiguous().float()
            assert tuple(noise.shape) == (steps, B, 6 + self.R)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        ctx._check(ctx.L.ddk_sample(ctx.h, self.h, B, steps, p(t), p(sc), p(nc), _ptr(noise), _ptr(pos), _stream()), 'ddk_sample')
        return pos

    def build_graph(self, pos, t_tr):
        """score_model.py:310-408 + :218-225 alone: ``(edge_index [2, E] int32, group_offsets [5])`` of the merged graph of the B poses
        ``pos`` [B, n_lig, 3] at diffusion time t_tr; groups [lig-lig | lig->rec | rec-rec | rec->lig], each sorted by row 0 (the
        receiving node, tensor_layers.py:159); nodes numbered [all ligand atoms | all residues]."""
        pos = _need_cuda(pos).contiguous().float().reshape(-1, self.n_lig, 3)
        B = pos.shape[0]
        cap = B * (self.M + self.n_lig * 33 + 2 * self.n_lig * self.n_rec + self.E_rr)      # 33: radius(..., 32 + 1) minus a self loop that may not be among them
        src = torch.empty(cap, dtype=torch.int32, device=pos.device)
        dst = torch.empty(cap, dtype=torch.int32, device=pos.device)
        off = torch.empty(5, dtype=torch.int32, device=pos.device)
        self.ctx._check(self.ctx.L.ddk_build_graph(self.ctx.h, self.h, B, _ptr(pos), float(t_tr), _ptr(src), _ptr(dst), cap, _ptr(off),
                                                   _stream()), 'ddk_build_graph')
        off = off.cpu()
        E = int(off[4])
        return torch.stack([src[:E], dst[:E]]), off

    def graph_stats(self):
        out = (C.c_int64 * 12)()
        self.ctx._check(self.ctx.L.ddk_last_graph_stats(self.ctx.h, self.h, out, _stream()), 'ddk_last_graph_stats')
        v = list(out)
        if v[6]:
            raise RuntimeError('ddk: edge capacity overflow')
        if v[11]:
            raise RuntimeError('ddk: the graph count and fill kernels disagreed about an edge count (internal consistency guard)')
        return dict(E_ll=v[0], E_lr=v[1], E_rr=v[2], E_rl=v[3], E_shared=v[4], E=v[5], cap=v[7], E_rr_live=(v[8], v[9], v[10]))

    def keep_receptor_features(self, on=True):
        """Evaluate the receptor rows of the last conv layer too (needed before ``node_features``' receptor output)."""
        self.ctx._check(self.ctx.L.ddk_set_keep_receptor_features(self.ctx.h, self.h, int(bool(on))), 'ddk_set_keep_receptor_features')

    def node_features(self, B, device):
        lig = torch.empty((B * self.n_lig, 84), dtype=torch.float32, device=device)
        rec = torch.empty((B * self.n_rec, 84), dtype=torch.float32, device=device)
        self.ctx._check(self.ctx.L.ddk_last_node_features(self.ctx.h, self.h, B, _ptr(lig), _ptr(rec), _stream()), 'ddk_last_node_features')
        return lig, rec

    def read_edges(self, B):
        st = self.graph_stats()
        E, N = st['E'], B * (self.n_lig + self.n_rec)
        src, dst = np.zeros(E, np.int32), np.zeros(E, np.int32)
        emb, sh, deg = np.zeros((E, 24), np.float32), np.zeros((E, 4), np.float32), np.zeros(N, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.ctx.L.ddk_debug_read_edges(self.ctx.h, self.h, E, p(src), p(dst), p(emb), p(sh), p(deg), N)
        return st, src, dst, emb, sh, deg
