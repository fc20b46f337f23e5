"""CPU restatement of the DisCo-DiffDock AR latent model at inference.  TEST INFRASTRUCTURE.

* ``ar_logits``   models/pretrained_score_encoder.py:46-89 (PretrainedScoreEncoder.forward, apply_gumbel_softmax=False):
                  score_model.embed() at t=1 with unconditional=1 and the partially decoded input latents, then two 3-layer
                  MLPs with BatchNorm1d(eval) on the scalar channels [x[:, :ns] | x[:, -ns:]] (ns = the AR yml's ns).
* ``encode_ar``   models/model_classes.py:9-49 (GenericEncoder.encode_ar, latent_vocab == 1): one node of the complex is picked
                  per latent dimension (argmax for temperature >= 100, else multinomial over exp(T * logit)).
Depends on the *_lite restatements through score_model_ref.embed -> PARITY UNPINNED (oracle/__init__.py)."""
import copy
import math

import torch
import torch.nn.functional as F

from . import score_model_ref as smr
from . import sampler_ref as spr


def _predictor(x, P, prefix):
    def bn(h, i):
        return (h - P[f'{prefix}.{i}.running_mean']) / torch.sqrt(P[f'{prefix}.{i}.running_var'] + 1e-5) * P[f'{prefix}.{i}.weight'] + P[f'{prefix}.{i}.bias']
    h = torch.relu(bn(F.linear(x, P[f'{prefix}.0.weight'], P[f'{prefix}.0.bias']), 1))
    h = torch.relu(bn(F.linear(h, P[f'{prefix}.4.weight'], P[f'{prefix}.4.bias']), 5))
    return F.linear(h, P[f'{prefix}.8.weight'], P[f'{prefix}.8.bias'])


def ar_logits(P, cfg, ar_ns, data, dtype=torch.float32):
    """returns [B, latent_out_dim, n_lig + n_rec] logits (graphs are copies of one complex)."""
    Ps = {k[len('pretrained_score_model.'):]: v for k, v in P.items() if k.startswith('pretrained_score_model.')}
    Ps = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in Ps.items()}
    data['ligand'].latent_h, data['receptor'].latent_h = data['ligand'].input_latent, data['receptor'].input_latent
    B = data.num_graphs
    spr.set_time(data, 1, 1, 1, B)
    data['ligand'].unconditional = torch.ones(data['ligand'].num_nodes, 1)
    data['receptor'].unconditional = torch.ones(data['receptor'].num_nodes, 1)
    for nt in ('ligand', 'receptor'):
        data[nt].pos = data[nt].pos.to(dtype)
    lig, rec = smr.embed(Ps, cfg, data, dtype)[:2]
    sl = _predictor(torch.cat([lig[:, :ar_ns], lig[:, -ar_ns:]], 1), P, 'latent_s_predictor')
    sr = _predictor(torch.cat([rec[:, :ar_ns], rec[:, -ar_ns:]], 1), P, 'latent_r_predictor')
    n_l, n_r = lig.shape[0] // B, rec.shape[0] // B
    return torch.stack([torch.cat([sl[i * n_l:(i + 1) * n_l], sr[i * n_r:(i + 1) * n_r]], 0).T for i in range(B)])


def encode_ar(P, cfg, ar_ns, data, sampling_temperature=1.0, choice_fn=None):
    B = data.num_graphs
    n_l, n_r = data['ligand'].num_nodes // B, data['receptor'].num_nodes // B
    latent_l = torch.zeros(B * n_l, cfg.latent_dim)
    latent_r = torch.zeros(B * n_r, cfg.latent_dim)
    for idx in range(cfg.latent_dim):
        d = copy.deepcopy(data)
        d['ligand'].input_latent, d['receptor'].input_latent = latent_l.clone(), latent_r.clone()
        lat = ar_logits(P, cfg, ar_ns, d)[:, 0, :] * sampling_temperature
        if sampling_temperature >= 100:
            choice = torch.argmax(lat, 1, keepdim=True)
        elif choice_fn is not None:
            choice = choice_fn(idx, lat)
        else:
            choice = torch.multinomial(torch.nan_to_num(torch.exp(lat)), 1)
        for i in range(B):
            c = int(choice[i, 0])
            if c < n_l:
                latent_l[i * n_l + c, idx] = 1
            else:
                latent_r[i * n_r + c - n_l, idx] = 1
    return latent_l, latent_r


def random_ar_state_dict(cfg, ar_ns=16, hidden=128, seed=0):
    """AR checkpoint layout: pretrained_score_model.* (its own copy of the DisCo score model) + the two predictors."""
    g = torch.Generator().manual_seed(seed)
    P = {'pretrained_score_model.' + k: v for k, v in smr.random_state_dict(cfg, seed=seed + 1).items()}
    for name in ('latent_s_predictor', 'latent_r_predictor'):
        for i, (o, n_in) in ((0, (hidden, 2 * ar_ns)), (4, (hidden, hidden)), (8, (1, hidden))):
            P[f'{name}.{i}.weight'] = (torch.rand(o, n_in, generator=g) * 2 - 1) / math.sqrt(n_in)
            P[f'{name}.{i}.bias'] = (torch.rand(o, generator=g) * 2 - 1) / math.sqrt(n_in)
        for i in (1, 5):
            P[f'{name}.{i}.weight'] = torch.rand(hidden, generator=g) + 0.5
            P[f'{name}.{i}.bias'] = torch.randn(hidden, generator=g) * 0.1
            P[f'{name}.{i}.running_mean'] = torch.randn(hidden, generator=g) * 0.1
            P[f'{name}.{i}.running_var'] = torch.rand(hidden, generator=g) + 0.5
            P[f'{name}.{i}.num_batches_tracked'] = torch.tensor(7)
    return P
