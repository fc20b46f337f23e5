"""Host mirror of the reference's confidence model interface: what ``get_model(confidence_args, device,
t_to_sigma, no_parallel=True, confidence_mode=True)`` returns for ``all_atoms: true`` checkpoints (utils/model_utils.py:25-68 ->
models/all_atom_score_model.py), i.e. an object with ``load_state_dict(state_dict, strict=True)``, ``eval()`` and
``__call__(batch) -> confidence [B] or [B, k]``.  The forward runs in libddk.so (``ddk_confidence_forward``); there is no CPU
fallback.  Supported: the paper_confidence_model configuration family (ns=24, nv=6, sh_lmax=2, first-order irreps,
OldAtomEncoder, ESM features, BatchNorm); anything else raises."""
import numpy as np
import torch
from torch import nn

from .runtime import Context, Complex, config_from_args

import collections
_conf_cache = collections.OrderedDict()        # content key -> Complex (+ atoms), least recently used first


def confidence_config(args, device_index=0):
    g = lambda k, d: getattr(args, k, d)
    if not g('all_atoms', False):
        raise RuntimeError('ddk: the confidence model on the device is the all-atom model (all_atoms: true)')
    if g('sh_lmax', 2) != 2 or g('use_second_order_repr', False) or not g('use_old_atom_encoder', True) or g('latent_dim', 0):
        raise RuntimeError('ddk: confidence model: only sh_lmax=2, first-order irreps, OldAtomEncoder, no latents are implemented')
    import copy
    a = copy.copy(args)
    a.sh_lmax = 1                      # config_from_args validates the score-model family; the all-atom flag selects the l<=2 product
    d = config_from_args(a, device_index)
    cut = g('rmsd_classification_cutoff', None)
    d.update(all_atoms=1, num_confidence_outputs=len(cut) + 1 if isinstance(cut, list) else 1,
             confidence_no_batchnorm=int(bool(g('confidence_no_batchnorm', False))))
    return d


class ConfidenceModel(nn.Module):
    def __init__(self, args, device):
        super().__init__()
        dev_index = (device.index or 0) if isinstance(device, torch.device) and device.type == 'cuda' else 0
        self.device = torch.device('cuda', dev_index)
        self.cfg = confidence_config(args, dev_index)
        self.ctx = None
        self._loaded = False
        self.last_complex = None

    def load_state_dict(self, state_dict, strict=True):
        if self.ctx is not None:
            self.ctx.close()
        self.ctx = Context(**self.cfg)
        if strict:
            need = [k for k in ('lig_node_embedding.linear.weight', 'confidence_predictor.8.weight', 'conv_layers.0.fc.3.weight') if k not in state_dict]
            if need:
                raise RuntimeError(f'ddk confidence model: missing keys {need}')
        self.ctx.load_state_dict(state_dict)
        self._loaded = True
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def to(self, *a, **k):
        return self

    def complex_for(self, batch):
        """Complex (+ atoms) in this model's context for a batch of B copies of one all-atom complex graph."""
        from .score_model import arrays_from_batch, _fingerprint, _first_view
        B = batch.num_graphs
        g0, B00 = _first_view(batch, B)
        n_a0, E_aa0 = g0['atom'].num_nodes // B00, g0['atom', 'atom'].num_edges // B00
        import hashlib
        from .score_model import _bytes_of
        ha = hashlib.blake2b(digest_size=16)
        for a in (g0['atom'].x[:n_a0], g0['atom'].pos[:n_a0], g0['atom', 'atom'].edge_index[:, :E_aa0], g0['atom', 'receptor'].edge_index[:, :n_a0]):
            ha.update(_bytes_of(a))
            ha.update(b'|')
        key = (id(self.ctx),) + _fingerprint(batch, B) + (n_a0, ha.hexdigest())
        cx = _conf_cache.get(key)
        if cx is None or cx.max_batch < B:
            while len(_conf_cache) > 2:            # evicted complexes hand their device chunks back to the context's pool
                _conf_cache.popitem(last=False)
            arr = arrays_from_batch(batch, B)
            g, B0 = _first_view(batch, B)          # the first graph itself when the batch knows it (no concatenation of the 40 copies)
            n_a = g['atom'].num_nodes // B0
            E_aa = g['atom', 'atom'].num_edges // B0
            cx = Complex(self.ctx, arr, max_batch=B)
            cx.set_atoms(g['atom'].x[:n_a].cpu(), g['atom'].pos[:n_a].cpu(), g['atom', 'atom'].edge_index[:, :E_aa].cpu(),
                         g['atom', 'receptor'].edge_index[:, :n_a].cpu())
            _conf_cache[key] = cx
        _conf_cache.move_to_end(key)
        return cx, B

    def forward(self, data, check=True):
        """``check=False``: no host synchronisation here; the caller runs ``self.last_complex.confidence_counts()`` later (sampling()
        does it once per call, behind its own read-back)."""
        if not self._loaded:
            raise RuntimeError('ddk confidence model: load_state_dict() first')
        pos = data['ligand'].pos
        if not pos.is_cuda:
            raise RuntimeError('ddk confidence model runs on the GPU only (no CPU fallback)')
        t = data.complex_t['tr'] if hasattr(data, 'complex_t') else None
        # (check=False is sampling()'s own call, which has just set the times to 0 itself: reading a device tensor here would wait for the
        # whole reverse-diffusion loop queued in front of it)
        if check and t is not None and float(torch.as_tensor(t).abs().max()) != 0.0:
            raise RuntimeError('ddk confidence model: implemented for complex_t = 0 (utils/sampling.py:236)')
        cx, B = self.complex_for(data)
        self.last_complex = cx
        out = cx.confidence_forward(pos.reshape(B, -1, 3), check=check)
        return out.squeeze(dim=-1)
