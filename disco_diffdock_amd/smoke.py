"""One tiny end-to-end hot-path invocation on cuda:0 checked against the oracle (used by __graft_entry__.smoke):
a 3-step, 3-sample reverse diffusion of one small synthetic complex through libddk.so vs the CPU oracle."""
import os
import sys

import numpy as np
import torch


def run():
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    if root not in sys.path:
        sys.path.insert(0, root)
    from argparse import Namespace
    from functools import partial
    from oracle import score_model_ref as smr, sampler_ref as spr, graph_lite     # the checker only
    from . import synthetic
    from .runtime import Context, Complex
    from .sampling import step_coefficients
    from .diffusion_utils import t_to_sigma, get_t_schedule
    dev = torch.device('cuda:0')
    args = Namespace(tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03,
                     tor_sigma_max=3.14, no_torsion=False)
    cfg = smr.ScoreModelConfig(latent_vocab=64)
    P = synthetic.random_score_model_state_dict(seed=1)
    c = synthetic.make_complex(4, n_res=40, n_lig=21)
    B, steps = 3, 3
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, B)
    sched = get_t_schedule(steps)
    t_arr, sc, nc = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=args), args, False, False, True, 1.0, 0.0, 0.5)
    rng = np.random.default_rng(0)
    pos0 = np.stack([c['lig_pos'] + rng.normal(0, 5.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    z = torch.randn(steps, B, 6 + cx.R, generator=torch.Generator().manual_seed(0))
    pos = torch.from_numpy(pos0).to(dev)
    cx.sample(pos, t_arr, sc, nc, z.to(dev))
    torch.cuda.synchronize()
    d = os.path.join(root, 'disco_diffdock_amd', 'data')
    tables = (np.load(os.path.join(d, 'so3_exp_score_norms.npy')), np.load(os.path.join(d, 'torus_score_norm_seed0.npy')))
    dl = []
    for p in pos0:
        g = graph_lite.make_complex(c['lig_x'], p, c['bond_index'], c['bond_attr'], c['edge_mask'], c['mask_rotate'], c['rec_x'],
                                    c['rec_pos'], c['rec_edge_index'])
        g['ligand'].mask_rotate = [g['ligand'].mask_rotate]
        dl.append(g)
    nf = lambda b, t, name, shape: {'tr': z[t, :, 0:3], 'rot': z[t, :, 3:6], 'tor': z[t, :, 6:].reshape(-1)}[name]
    ref, _ = spr.sampling(dl, P, cfg, tables[0], tables[1], steps, sched, sched, sched, noise_fn=nf, batch_size=B, no_final_step_noise=True)
    ref = torch.cat([g['ligand'].pos for g in ref])
    err = float((pos.cpu().reshape(-1, 3) - ref).abs().max() / ref.abs().max())
    print(f'smoke: 3-step reverse diffusion on cuda:0 vs oracle, rel err {err:.2e}; graph {cx.graph_stats()}')
    assert err < 1e-4
