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
                tor_sigma_min=0.03, tor_sigma_max=3.14, device=0)


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


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """One ddk_ctx (one device).  device=-1 gives a host-only context (weight packing only, for CPU tests)."""

    def __init__(self, device=0, **cfg):
        self.L = _lib.lib()
        d = dict(DEFAULTS)
        d.update(cfg)
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
