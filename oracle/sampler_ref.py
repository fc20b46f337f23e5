"""CPU restatement of the reverse-diffusion sampler and the SE(3)/torus conformer update.
TEST INFRASTRUCTURE.  Plain PyTorch-CPU in the reference's op order; cites reference lines
(paths relative to /root/reference).  Noise is injected by the caller (the reference never
seeds its generators) through ``noise_fn``."""
import copy
import math

import numpy as np
import torch

from . import score_model_ref as smr
from .graph_lite import DataLoader


def get_t_schedule(inference_steps):
    """utils/diffusion_utils.py:97-98"""
    return np.linspace(1, 0, inference_steps + 1)[:-1]


def set_time(batch, t_tr, t_rot, t_tor, batchsize):
    """utils/diffusion_utils.py:101-117 (coarse-grained graphs: ligand + receptor node types)"""
    for nt in ('ligand', 'receptor'):
        n = batch[nt].num_nodes
        batch[nt].node_t = {'tr': t_tr * torch.ones(n), 'rot': t_rot * torch.ones(n), 'tor': t_tor * torch.ones(n)}
    batch.complex_t = {'tr': t_tr * torch.ones(batchsize), 'rot': t_rot * torch.ones(batchsize),
                       'tor': t_tor * torch.ones(batchsize)}


# ---- utils/geometry.py ---------------------------------------------------------------------
def axis_angle_to_quaternion(axis_angle):
    """utils/geometry.py:38-68"""
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    small = angles.abs() < 1e-6
    s = torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    return torch.cat([torch.cos(half), axis_angle * s], dim=-1)


def quaternion_to_matrix(q):
    """utils/geometry.py:6-35"""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_matrix(axis_angle):
    """utils/geometry.py:71-85"""
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def kabsch_batch(A, B):
    """utils/geometry.py:126-156 (rigid_transform_Kabsch_3D_torch_batch): R,t with R A + t ~ B."""
    A, B = A.permute(0, 2, 1), B.permute(0, 2, 1)
    cA, cB = A.mean(dim=2, keepdim=True), B.mean(dim=2, keepdim=True)
    H = torch.bmm(A - cA, (B - cB).transpose(1, 2))
    U, S, Vt = torch.linalg.svd(H)
    R = torch.bmm(Vt.transpose(1, 2), U.transpose(1, 2))
    SS = torch.diag(torch.tensor([1., 1., -1.], dtype=A.dtype))
    Rm = torch.bmm(Vt.transpose(1, 2) @ SS, U.transpose(1, 2))
    R = torch.where(torch.linalg.det(R)[:, None, None] < 0, Rm, R)
    t = torch.bmm(-R, cA) + cB
    return R, t


# ---- utils/torsion.py ----------------------------------------------------------------------
def modify_conformer_torsion_angles_batch(pos, edge_index, mask_rotate, torsion_updates):
    """utils/torsion.py:71-86: sequential, in bond order, on already-updated coordinates."""
    pos = pos + 0
    for idx_edge, e in enumerate(edge_index):
        u, v = int(e[0]), int(e[1])
        assert not mask_rotate[idx_edge, u] and mask_rotate[idx_edge, v]
        rot_vec = pos[:, u] - pos[:, v]
        rot_mat = axis_angle_to_matrix(rot_vec / torch.linalg.norm(rot_vec, dim=-1, keepdims=True)
                                       * torsion_updates[:, idx_edge:idx_edge + 1])
        m = mask_rotate[idx_edge]
        pos[:, m] = torch.bmm(pos[:, m] - pos[:, v:v + 1], rot_mat.transpose(1, 2)) + pos[:, v:v + 1]
    return pos


def modify_conformer_torsion_angles_np(pos, edge_index, mask_rotate, torsion_updates):
    """utils/torsion.py:48-68 (numpy variant used by randomize_position; skips zero updates)."""
    from scipy.spatial.transform import Rotation as R
    pos = copy.deepcopy(pos)
    if type(pos) != np.ndarray:
        pos = pos.cpu().numpy()
    for idx_edge, e in enumerate(np.asarray(edge_index)):
        if torsion_updates[idx_edge] == 0:
            continue
        u, v = e[0], e[1]
        rot_vec = pos[u] - pos[v]
        rot_vec = rot_vec * torsion_updates[idx_edge] / np.linalg.norm(rot_vec)
        rot_mat = R.from_rotvec(rot_vec).as_matrix()
        pos[mask_rotate[idx_edge]] = (pos[mask_rotate[idx_edge]] - pos[v]) @ rot_mat.T + pos[v]
    return torch.from_numpy(pos.astype(np.float32))


# ---- utils/diffusion_utils.py --------------------------------------------------------------
def modify_conformer_batch(orig_pos, batch, tr_update, rot_update, torsion_updates, mask_rotate):
    """utils/diffusion_utils.py:37-55"""
    B = batch.num_graphs
    N = batch['ligand'].num_nodes // B
    M = batch['ligand', 'ligand'].num_edges // B
    pos = orig_pos.reshape(B, N, 3) + 0
    edge_index = batch['ligand', 'ligand'].edge_index[:, :M]
    edge_mask = batch['ligand'].edge_mask[:M]
    torsion_updates = torsion_updates.reshape(B, -1) if torsion_updates is not None else None
    lig_center = torch.mean(pos, dim=1, keepdim=True)
    rot_mat = axis_angle_to_matrix(rot_update)
    rigid = torch.bmm(pos - lig_center, rot_mat.permute(0, 2, 1)) + tr_update.unsqueeze(1) + lig_center
    if torsion_updates is None:
        return rigid.reshape(-1, 3)
    flex = modify_conformer_torsion_angles_batch(rigid, edge_index.T[edge_mask], mask_rotate, torsion_updates)
    R, t = kabsch_batch(flex, rigid)
    return (torch.bmm(flex, R.transpose(1, 2)) + t.transpose(1, 2)).reshape(-1, 3)


# ---- utils/sampling.py ---------------------------------------------------------------------
def _as3(v):
    try:
        iter(v)
        return list(v)
    except TypeError:
        return [v] * 3


def sde_step_coefficients(t_idx, inference_steps, schedules, cfg, ode=False, temp_sampling=1.0, temp_psi=0.0,
                          temp_sigma_data=0.5):
    """Host scalars of one reverse step (utils/sampling.py:106-111,137-192): for each of tr/rot/tor
    returns (sigma, score_coeff, noise_coeff) with  perturb = score_coeff*score + noise_coeff*z."""
    temp_sampling, temp_psi, temp_sigma_data = _as3(temp_sampling), _as3(temp_psi), _as3(temp_sigma_data)
    out = []
    ts = [schedules[k][t_idx] for k in range(3)]
    sig = smr.t_to_sigma(ts[0], ts[1], ts[2], cfg)
    lims = [(cfg.tr_sigma_min, cfg.tr_sigma_max), (cfg.rot_sigma_min, cfg.rot_sigma_max), (cfg.tor_sigma_min, cfg.tor_sigma_max)]
    for k in range(3):
        sch = schedules[k]
        dt = sch[t_idx] - sch[t_idx + 1] if t_idx < inference_steps - 1 else sch[t_idx]
        lo, hi = lims[k]
        # reference: sigma * torch.sqrt(torch.tensor(2*np.log(hi/lo)))  -> fp32 sqrt of an fp64->fp32 cast, times numpy float64
        g = sig[k] * torch.sqrt(torch.tensor(2 * np.log(hi / lo)))
        if ode:
            sc, nc = 0.5 * g ** 2 * dt, 0.0 * g
        else:
            sc, nc = g ** 2 * dt, g * np.sqrt(dt)
        if temp_sampling[k] != 1.0:
            sd = np.exp(temp_sigma_data[k] * np.log(hi) + (1 - temp_sigma_data[k]) * np.log(lo))
            lam = (sd + sig[k]) / (sd + sig[k] / temp_sampling[k])
            sc = g ** 2 * dt * (lam + temp_sampling[k] * temp_psi[k] / 2)
            nc = g * np.sqrt(dt * (1 + temp_psi[k]))
        out.append((float(sig[k]), sc, nc))
    return out


def sampling(data_list, P, cfg, so3_table, torus_table, inference_steps, tr_schedule, rot_schedule, tor_schedule,
             noise_fn=None, no_random=False, ode=False, batch_size=32, no_final_step_noise=False,
             temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5, dtype=torch.float32, trace=None,
             classifier_free_guidance_weight=0.0, cfg_start=1.0, cfg_end=0.0):
    """utils/sampling.py:49-249 without CFG / confidence model; latent-conditioned models read latent_h / unconditional from the graphs (set by the caller after AR decoding, sampling.py:69-103).

    ``noise_fn(batch_id, t_idx, name, shape)`` supplies z ~ N(0,1) (name in 'tr','rot','tor');
    default: torch.normal under the global generator, like the reference."""
    N = len(data_list)
    loader = DataLoader(data_list, batch_size=batch_size)
    mask_rotate = torch.from_numpy(data_list[0]['ligand'].mask_rotate[0])
    if noise_fn is None:
        noise_fn = lambda b, t, name, shape: torch.normal(mean=0, std=1, size=shape)
    schedules = (tr_schedule, rot_schedule, tor_schedule)
    with torch.no_grad():
        for batch_id, batch in enumerate(loader):
            b = batch.num_graphs
            for t_idx in range(inference_steps):
                t_tr, t_rot, t_tor = tr_schedule[t_idx], rot_schedule[t_idx], tor_schedule[t_idx]
                set_time(batch, t_tr, t_rot, t_tor, b)
                tr_score, rot_score, tor_score = smr.score_model_forward(P, cfg, batch, so3_table, torus_table, dtype)
                if classifier_free_guidance_weight != 0.0 and t_tr <= cfg_start and t_tr >= cfg_end:   # sampling.py:119-135
                    w = classifier_free_guidance_weight
                    keep = (batch['ligand'].latent_h, batch['receptor'].latent_h, batch['ligand'].unconditional, batch['receptor'].unconditional)
                    batch['ligand'].unconditional, batch['receptor'].unconditional = torch.ones_like(keep[2]), torch.ones_like(keep[3])
                    batch['ligand'].latent_h, batch['receptor'].latent_h = 0 * keep[0], 0 * keep[1]
                    u_tr, u_rot, u_tor = smr.score_model_forward(P, cfg, batch, so3_table, torus_table, dtype)
                    tr_score, rot_score, tor_score = tr_score + w * (tr_score - u_tr), rot_score + w * (rot_score - u_rot), tor_score + w * (tor_score - u_tor)
                    batch['ligand'].latent_h, batch['receptor'].latent_h, batch['ligand'].unconditional, batch['receptor'].unconditional = keep
                coef = sde_step_coefficients(t_idx, inference_steps, schedules, cfg, ode, temp_sampling, temp_psi, temp_sigma_data)
                zero = no_random or (no_final_step_noise and t_idx == inference_steps - 1)
                nb = min(batch_size, N)

                def z(name, shape):
                    return torch.zeros(shape, dtype=dtype) if (zero or ode) else noise_fn(batch_id, t_idx, name, shape).to(dtype)

                tr_perturb = coef[0][1] * tr_score + coef[0][2] * z('tr', (nb, 3))
                rot_perturb = coef[1][1] * rot_score + coef[1][2] * z('rot', (nb, 3))
                if not cfg.no_torsion:
                    tor_perturb = coef[2][1] * tor_score + coef[2][2] * z('tor', tuple(tor_score.shape))
                else:
                    tor_perturb = None
                if trace is not None:
                    trace.append(dict(t_idx=t_idx, pos=batch['ligand'].pos.clone(), tr_score=tr_score.clone(),
                                      rot_score=rot_score.clone(), tor_score=tor_score.clone(),
                                      tr_perturb=tr_perturb.clone(), rot_perturb=rot_perturb.clone(),
                                      tor_perturb=None if tor_perturb is None else tor_perturb.clone()))
                batch['ligand'].pos = modify_conformer_batch(batch['ligand'].pos, batch, tr_perturb.to(dtype),
                                                             rot_perturb.to(dtype),
                                                             tor_perturb.to(dtype) if tor_perturb is not None else None,
                                                             mask_rotate)
            len_lig = len(batch['ligand'].pos) // b
            for i in range(b):
                data_list[batch_id * batch_size + i]['ligand'].pos = batch['ligand'].pos[i * len_lig:len_lig * (i + 1)]
    return data_list, None


def randomize_position(data_list, no_torsion, no_random, tr_sigma_max, rng=None):
    """utils/sampling.py:12-34 with an explicit numpy Generator instead of the global RNGs."""
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(0) if rng is None else rng
    for g in data_list:
        lig = g['ligand']
        if not no_torsion:
            upd = rng.uniform(low=-np.pi, high=np.pi, size=int(lig.edge_mask.sum()))
            lig.pos = modify_conformer_torsion_angles_np(lig.pos, g['ligand', 'ligand'].edge_index.T[lig.edge_mask],
                                                         lig.mask_rotate[0], upd)
    for g in data_list:
        lig = g['ligand']
        center = torch.mean(lig.pos, dim=0, keepdim=True)
        rot = torch.from_numpy(R.random(random_state=rng).as_matrix()).float()
        lig.pos = (lig.pos - center) @ rot.T
        if not no_random:
            lig.pos = lig.pos + torch.from_numpy(rng.normal(0, tr_sigma_max, size=(1, 3))).float()
