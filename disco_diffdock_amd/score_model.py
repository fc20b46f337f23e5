"""Host mirror of the reference's models/score_model.py call surface: ``TensorProductScoreModel`` whose
``forward(batch) -> (tr_pred [B,3], rot_pred [B,3], tor_pred [sum R])`` (score_model.py:259-308) runs entirely in
libddk.so.  The checkpoint format is the reference's ``score_model.state_dict()`` (evaluate.py:169-171)."""
import numpy as np
import os
import torch
from torch import nn

from .runtime import Context, Complex, config_from_args, DEFAULTS

import collections
_complex_cache = collections.OrderedDict()     # (id(ctx), content key) -> Complex, least recently used first
_CACHE_PER_CTX = 3       # complexes kept per context: evicted ones hand their device chunks back to the context's pool (no hipMalloc in steady state)
_default_ctx = {}


def _first_view(batch, B):
    """(graph, 1) for batches that know their first graph (our collate), else (batch, B): slices below take 1/B of every array"""
    first = getattr(batch, 'first', None)
    return (first, 1) if first is not None else (batch, B)


def _fingerprint(batch, B, mask_rotate=None):
    batch, B0 = _first_view(batch, B)
    return (B,) + _fingerprint_of(batch, B0, mask_rotate)[1:]


def _bytes_of(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a).tobytes()


def _fingerprint_of(batch, B, mask_rotate=None):
    """Content key of ONE complex: a digest over everything ``Complex`` uploads (atom features in atom order, bond topology and
    types, rotatable-bond masks, receptor geometry / residue ids / edges and a slice + checksum of the ESM features).  Sums of
    features are not enough: two ligands of the same composition (atom order reversed, regio-isomers) share every sum."""
    import hashlib
    lig, rec = batch['ligand'], batch['receptor']
    n_l, n_r = lig.num_nodes // B, rec.num_nodes // B
    M, E = batch['ligand', 'ligand'].num_edges // B, batch['receptor', 'receptor'].num_edges // B
    if mask_rotate is None:
        mask_rotate = lig.mask_rotate
        while isinstance(mask_rotate, (list, tuple)):
            mask_rotate = mask_rotate[0]
    h = hashlib.blake2b(digest_size=16)
    for a in (lig.x[:n_l], batch['ligand', 'ligand'].edge_index[:, :M], batch['ligand', 'ligand'].edge_attr[:M], lig.edge_mask[:M],
              np.asarray(mask_rotate.cpu() if torch.is_tensor(mask_rotate) else mask_rotate, dtype=np.uint8).reshape(-1),
              rec.pos[:n_r], rec.x[:n_r, :17], batch['receptor', 'receptor'].edge_index[:, :E]):
        h.update(_bytes_of(a))
        h.update(b'|')
    # (numpy, not torch: an OpenMP-parallel torch CPU op was measured to stall for a whole GPU loop (~80 ms) on a many-core host while the
    # HIP runtime is busy; the sums below are single-threaded and need no temporary)
    xr = rec.x[:n_r]
    xr = xr.detach().cpu().numpy() if torch.is_tensor(xr) else np.asarray(xr)
    h.update(np.float64(xr.sum(dtype=np.float64)).tobytes())
    h.update(np.ascontiguousarray(xr[:, 17::97]).tobytes())       # + a strided sample of the language-model features
    return (B, n_l, n_r, M, E, h.hexdigest())


def arrays_from_batch(batch, B, mask_rotate=None):
    """Slice the first graph out of a batch of B copies of one complex (utils/sampling.py:57, Appendix A.10)."""
    batch, B = _first_view(batch, B)
    lig, rec = batch['ligand'], batch['receptor']
    n_l, n_r = lig.num_nodes // B, rec.num_nodes // B
    M = batch['ligand', 'ligand'].num_edges // B
    E = batch['receptor', 'receptor'].num_edges // B
    if mask_rotate is None:
        mr = lig.mask_rotate
        while isinstance(mr, (list, tuple)):
            mr = mr[0]
        mask_rotate = mr
    mask_rotate = mask_rotate.cpu().numpy() if torch.is_tensor(mask_rotate) else np.asarray(mask_rotate)
    return dict(lig_x=lig.x[:n_l].cpu(), bond_index=batch['ligand', 'ligand'].edge_index[:, :M].cpu(),
                bond_attr=batch['ligand', 'ligand'].edge_attr[:M].cpu(), edge_mask=lig.edge_mask[:M].cpu(),
                mask_rotate=mask_rotate.reshape(-1, n_l), rec_x=rec.x[:n_r].cpu(), rec_pos=rec.pos[:n_r].cpu(),
                rec_edge_index=batch['receptor', 'receptor'].edge_index[:, :E].cpu())


def complex_for_batch(batch, device, ctx=None, mask_rotate=None, need_model=True):
    B = batch.num_graphs
    if ctx is None:
        if need_model:
            raise RuntimeError('ddk: no model context bound to this call')
        from .tensor_layers import _shape_context
        ctx = _shape_context(device.index or 0)
    key = (id(ctx),) + _fingerprint(batch, B, mask_rotate)
    cx = _complex_cache.get(key)
    if cx is None or cx.max_batch < B:
        mine = [k for k in _complex_cache if k[0] == key[0] and k != key]
        for k in mine[:max(0, len(mine) - (_CACHE_PER_CTX - 1))]:      # oldest first
            del _complex_cache[k]
        cx = Complex(ctx, arrays_from_batch(batch, B, mask_rotate), max_batch=B)
        _complex_cache[key] = cx
    _complex_cache.move_to_end(key)
    return cx, B


def _is_e3nn_internal(key):
    """Keys a real checkpoint carries for e3nn's own modules and that hold no model parameter: the tensor-product objects inside
    every conv layer (``*.tp.weight`` is empty with shared_weights=False, ``*.tp.output_mask``), the weightless
    ``final_tp_tor = o3.FullTensorProduct(...)`` of the torsion head (models/score_model.py:152: ``final_tp_tor.weight`` [0],
    ``final_tp_tor.output_mask``) and the TorchScript sub-modules of e3nn's code generator (``_compiled_*``)."""
    parts = key.split('.')
    return ('tp' in parts[:-1] or parts[0] == 'final_tp_tor' or parts[-1] == 'output_mask'
            or any(p.startswith('_compiled_') for p in parts))


def check_state_dict(spec, state_dict, strict=True, what='ddk score model'):
    """``nn.Module.load_state_dict`` key / shape validation of ``state_dict`` against ``spec`` (name -> shape) after dropping e3nn's internal
    keys: returns (tensors of the spec'd keys, missing, unexpected); raises like torch on mis-shaped tensors, on missing / unexpected keys
    with ``strict``, and on missing keys always (nothing on the device can run with a partial checkpoint)."""
    have = {k: v for k, v in state_dict.items() if k in spec or not _is_e3nn_internal(k)}
    missing = [k for k in spec if k not in have]
    unexpected = [k for k in have if k not in spec]
    bad = [f'{k}: checkpoint {tuple(have[k].shape)} vs model {tuple(spec[k])}' for k in spec
           if k in have and tuple(have[k].shape) != tuple(spec[k])]
    if bad:
        raise RuntimeError(f'{what}: size mismatch for ' + '; '.join(bad))
    if strict and (missing or unexpected):
        raise RuntimeError(f'{what}: error(s) in loading state_dict: missing keys {missing}, unexpected keys {unexpected}')
    if missing:
        raise RuntimeError(f'{what}: the device path needs the complete checkpoint; missing {missing}')
    return have, missing, unexpected


class TensorProductScoreModel(nn.Module):
    """Constructor keywords follow models/score_model.py:15-24; only the coarse-grained sh_lmax=1 score model
    (DiffDock-S) is implemented on the device - anything else raises instead of silently falling back."""

    def __init__(self, t_to_sigma=None, device=None, timestep_emb_func=None, in_lig_edge_features=4, sigma_embed_dim=32,
                 sh_lmax=2, ns=16, nv=4, num_conv_layers=2, lig_max_radius=5, rec_max_radius=30, cross_max_distance=250,
                 center_max_distance=30, distance_embed_dim=32, cross_distance_embed_dim=32, no_torsion=False,
                 scale_by_sigma=True, use_second_order_repr=False, batch_norm=True, dynamic_max_cross=False, dropout=0.0,
                 lm_embedding_type=None, confidence_mode=False, use_old_atom_encoder=False, latent_dim=0, latent_vocab=32,
                 latent_cross_attention=False, latent_droprate=0.0, embedding_scale=1000.0, sigma_limits=None, conv_kernel=None,
                 confidence_dropout=0, confidence_no_batchnorm=False, num_confidence_outputs=1, **unused):
        super().__init__()
        if 'conv_f16x3' in unused or os.environ.get('DDK_CONV_F16X3') is not None:
            # round 2's switch had the opposite sense (1 = the f16 kernel); round 3 replaced it by conv_kernel (0 = the f16-limb kernel, the
            # default; 1 = fp32 MFMA; since ddk 0.8: 0 = two limbs / three products, 3 = three limbs / six products).  Swallowing the old spelling would switch kernels silently.
            raise RuntimeError("ddk: the conv_f16x3 option / DDK_CONV_F16X3 variable was replaced by the conv_kernel option "
                               "(0 = f16-limb product, default; 1 = fp32 MFMA; 3 = three-limb / six-product form) - see INTEGRATION.md")
        if sh_lmax != 1 or use_second_order_repr or use_old_atom_encoder or latent_cross_attention:
            raise RuntimeError('ddk implements the sh_lmax=1 first-order score model with the new AtomEncoder only')
        if confidence_mode and (num_conv_layers < 4 or latent_dim):      # (ddk_create: num_conv_layers in [4, 16]; the predictor reads the full 0e+1o+1e+0o rows)
            raise RuntimeError('ddk: confidence_mode of the coarse-grained model needs num_conv_layers >= 4 and no latents')
        if in_lig_edge_features != 4:
            raise RuntimeError('ddk: in_lig_edge_features must be 4')
        lim = sigma_limits or {k: DEFAULTS[k] for k in ('tr_sigma_min', 'tr_sigma_max', 'rot_sigma_min', 'rot_sigma_max',
                                                        'tor_sigma_min', 'tor_sigma_max')}
        dev_index = (device.index or 0) if isinstance(device, torch.device) and device.type == 'cuda' else 0
        self.device = torch.device('cuda', dev_index)
        self.cfg = dict(ns=ns, nv=nv, num_conv_layers=num_conv_layers, sigma_embed_dim=sigma_embed_dim,
                        distance_embed_dim=distance_embed_dim, cross_distance_embed_dim=cross_distance_embed_dim,
                        lig_max_radius=float(lig_max_radius), rec_max_radius=float(rec_max_radius),
                        cross_max_distance=float(cross_max_distance), center_max_distance=float(center_max_distance),
                        dynamic_max_cross=int(bool(dynamic_max_cross)), embedding_scale=float(embedding_scale),
                        scale_by_sigma=int(bool(scale_by_sigma)), no_torsion=int(bool(no_torsion)), batch_norm=int(bool(batch_norm)),
                        latent_dim=int(latent_dim), latent_vocab=int(latent_vocab), latent_droprate=float(latent_droprate),
                        lm_embedding_dim=1280 if lm_embedding_type == 'esm' else 0, **lim)
        self.confidence_mode = bool(confidence_mode)
        if confidence_mode:      # models/score_model.py:110-121: no score heads, a confidence_predictor on the pooled ligand scalars (ddk_score_confidence)
            self.cfg.update(confidence_mode=1, num_confidence_outputs=int(num_confidence_outputs), confidence_no_batchnorm=int(bool(confidence_no_batchnorm)))
        if conv_kernel is not None:      # extra (not in the reference ctor): 1 selects the fp32-MFMA conv kernel, 3 the three-limb / six-product form (ddk_config.conv_kernel)
            self.cfg['conv_kernel'] = int(conv_kernel)
        self.ctx = Context(device=dev_index, **self.cfg)
        self.no_torsion = no_torsion
        self._loaded = False

    def expected_state_dict_spec(self):
        """name -> shape of the reference ``score_model.state_dict()`` for this configuration: the tensors
        ``ddk_finalize_weights`` consumes."""
        from .synthetic import score_model_state_dict_spec
        c = self.cfg
        ns, sig, dist, lm = c['ns'], c['sigma_embed_dim'], c['distance_embed_dim'], c['lm_embedding_dim']
        spec = dict(score_model_state_dict_spec(ns=ns, nv=c['nv'], num_conv_layers=c['num_conv_layers'], sigma=sig, dist=dist, lm=lm,
                                                latent_dim=c['latent_dim'], latent_droprate=c['latent_droprate'],
                                                confidence_mode=getattr(self, 'confidence_mode', False), num_confidence_outputs=c.get('num_confidence_outputs', 1),
                                                confidence_no_batchnorm=bool(c.get('confidence_no_batchnorm', 0))))
        if not c['batch_norm']:
            spec = {k: v for k, v in spec.items() if '.batch_norm.' not in k}
        if c['no_torsion'] and not getattr(self, 'confidence_mode', False):
            spec = {k: v for k, v in spec.items() if not k.startswith(('final_edge_embedding', 'tor_bond_conv', 'tor_final_layer'))}
        return spec

    # the parameters live in the ddk context (packed for the kernels), not in nn.Parameters
    def load_state_dict(self, state_dict, strict=True, extra=None):
        """``nn.Module.load_state_dict`` semantics on the reference key set: with ``strict`` a missing, unexpected or mis-shaped
        key raises; e3nn's internal buffers of real checkpoints (:func:`_is_e3nn_internal`) are ignored.  ``extra``: further tensors
        for the same ddk context (the AR model's predictor weights, already validated by the caller)."""
        # (BatchNorm1d's num_batches_tracked counters of the confidence_predictor carry nothing the eval-mode forward reads: optional)
        spec = {k: v for k, v in self.expected_state_dict_spec().items() if not k.endswith('num_batches_tracked')}
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith('num_batches_tracked')}
        have, missing, unexpected = check_state_dict(spec, state_dict, strict)
        if extra:
            self.ctx.load_state_dict(extra, finalize=False)
        self.ctx.load_state_dict({k: have[k] for k in spec})
        self._loaded = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def eval(self):
        return self

    def to(self, device):
        return self

    def _bind_latents(self, cx, data):
        """latent_h / unconditional inputs of the DisCo models (score_model.py:170-184,209-215)."""
        if self.cfg['latent_dim'] <= 0:
            return
        lig, rec = data['ligand'], data['receptor']
        if 'latent_h' not in lig or 'latent_h' not in rec:
            raise RuntimeError('ddk: latent-conditioned score model needs data[...].latent_h (sampling.py:87-88)')
        unc = float(lig.unconditional.reshape(-1)[0]) if 'unconditional' in lig else 0.0
        cx.set_latents(lig.latent_h.to(self.device), rec.latent_h.to(self.device), unc)

    def confidence(self, data, pos, t):
        """confidence_mode: [B] or [B, k] for the poses ``pos`` of the batch's complex at complex_t = ``t`` (three floats, used as sigmas:
        models/score_model.py:186-189, 263-266).  No host synchronisation (sampling() calls this behind its device loop)."""
        if not self._loaded:
            raise RuntimeError('ddk score model: load_state_dict() first')
        if not pos.is_cuda:
            raise RuntimeError('ddk score model runs on the GPU only (no CPU fallback)')
        cx, B = complex_for_batch(data, pos.device, ctx=self.ctx)
        self.last_complex = cx
        return cx.score_confidence(pos.reshape(B, -1, 3), *t).squeeze(dim=-1)

    def forward(self, data, keep_receptor_features=False):
        if not self._loaded:
            raise RuntimeError('ddk score model: load_state_dict() first')
        pos = data['ligand'].pos
        if not pos.is_cuda:
            raise RuntimeError('ddk score model runs on the GPU only (no CPU fallback)')
        if self.confidence_mode:
            return self.confidence(data, pos, [float(data.complex_t[k][0]) for k in ('tr', 'rot', 'tor')])
        cx, B = complex_for_batch(data, pos.device, ctx=self.ctx)
        self._bind_latents(cx, data)
        cx.keep_receptor_features(keep_receptor_features)
        t = [float(data.complex_t[k][0]) for k in ('tr', 'rot', 'tor')]
        tr, rot, tor = cx.score_forward(pos.reshape(B, -1, 3), *t)
        self.last_complex = cx
        if self.no_torsion or cx.R == 0:
            tor = torch.empty(0, device=pos.device)
        return tr, rot, tor

    def embed(self, data):
        """models/score_model.py:169-257: (lig_node_attr, rec_node_attr, tr_sigma, rot_sigma, tor_sigma) after the conv stack."""
        from .diffusion_utils import t_to_sigma
        from types import SimpleNamespace
        self.forward(data, keep_receptor_features=True)
        B = data.num_graphs
        lig, rec = self.last_complex.node_features(B, data['ligand'].pos.device)
        sig = t_to_sigma(*[data.complex_t[k] for k in ('tr', 'rot', 'tor')], SimpleNamespace(**self.cfg))
        return (lig, rec) + tuple(sig)


class ModelWrapper(nn.Module):
    """models/model_classes.py:53-85 as far as inference needs it: ``model.score_model`` and ``model.encoder``."""

    def __init__(self, encoder, score_model):
        super().__init__()
        self.encoder = encoder
        self.score_model = score_model

    def forward(self, data):
        return self.score_model(data)

    def load_state_dict(self, state_dict, strict=True):
        """evaluate.py:164-171 loads either the wrapper's state_dict (keys ``score_model.*`` / ``encoder.*``) or the bare
        score model's; the (oracle) latent encoder is outside the hot path, so its keys are reported as unexpected."""
        wrapped = any(k.startswith('score_model.') for k in state_dict)
        inner = {k[len('score_model.'):]: v for k, v in state_dict.items() if k.startswith('score_model.')} if wrapped else dict(state_dict)
        extra = [k for k in state_dict if wrapped and not k.startswith('score_model.')]
        if strict and extra and self.encoder is None:
            enc = [k for k in extra if not k.startswith('encoder.')]
            if enc:
                raise RuntimeError(f'ddk: unexpected keys {enc}')
        res = self.score_model.load_state_dict(inner, strict=strict)
        return torch.nn.modules.module._IncompatibleKeys(list(res.missing_keys), list(res.unexpected_keys) + extra)
