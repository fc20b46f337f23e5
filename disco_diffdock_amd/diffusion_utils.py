"""Host mirror of the reference's utils/diffusion_utils.py entries on the hot path (same names and argument
meaning): t_to_sigma (:12-16), get_t_schedule (:97-98), set_time (:101-117), sinusoidal_embedding (:58-69),
get_timestep_embedding (:87-94) and modify_conformer_batch (:37-55, executed by ddk_se3_update on the GPU)."""
import math

import numpy as np
import torch


def t_to_sigma(t_tr, t_rot, t_tor, args):
    tr_sigma = args.tr_sigma_min ** (1 - t_tr) * args.tr_sigma_max ** t_tr
    rot_sigma = args.rot_sigma_min ** (1 - t_rot) * args.rot_sigma_max ** t_rot
    tor_sigma = args.tor_sigma_min ** (1 - t_tor) * args.tor_sigma_max ** t_tor
    return tr_sigma, rot_sigma, tor_sigma


def get_t_schedule(inference_steps):
    return np.linspace(1, 0, inference_steps + 1)[:-1]


def sinusoidal_embedding(timesteps, embedding_dim, max_positions=10000):
    half_dim = embedding_dim // 2
    emb = math.log(max_positions) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim, dtype=torch.float32, device=timesteps.device) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def get_timestep_embedding(embedding_type, embedding_dim, embedding_scale=10000):
    if embedding_type != 'sinusoidal':
        raise NotImplementedError('ddk implements the sinusoidal timestep embedding of the shipped models')
    return lambda x: sinusoidal_embedding(embedding_scale * x, embedding_dim)


def set_time(complex_graphs, t_tr, t_rot, t_tor, batchsize, all_atoms, device):
    types = ('ligand', 'receptor', 'atom') if all_atoms else ('ligand', 'receptor')
    for nt in types:
        n = complex_graphs[nt].num_nodes
        complex_graphs[nt].node_t = {'tr': t_tr * torch.ones(n, device=device), 'rot': t_rot * torch.ones(n, device=device),
                                     'tor': t_tor * torch.ones(n, device=device)}
    complex_graphs.complex_t = {'tr': t_tr * torch.ones(batchsize, device=device), 'rot': t_rot * torch.ones(batchsize, device=device),
                                'tor': t_tor * torch.ones(batchsize, device=device)}


def modify_conformer_batch(orig_pos, data, tr_update, rot_update, torsion_updates, mask_rotate):
    """Same signature as the reference; ``data`` is a batch of B copies of one complex.  Runs on the GPU."""
    from .score_model import complex_for_batch
    if not orig_pos.is_cuda:
        raise RuntimeError('ddk modify_conformer_batch runs on the GPU only (no CPU fallback)')
    cx, B = complex_for_batch(data, orig_pos.device, mask_rotate=mask_rotate, need_model=False)
    out = cx.se3_update(orig_pos.reshape(B, -1, 3), tr_update, rot_update,
                        torsion_updates.reshape(-1) if torsion_updates is not None else None)
    return out.reshape(-1, 3)
