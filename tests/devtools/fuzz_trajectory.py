"""Development aid: randomised 2-step reverse-diffusion trajectories (ddk_sample: score model + SDE step + rigid / torsion / Kabsch
update) against the CPU oracle sampler over small complexes of varied shape and ligand topology."""
import os, sys
from functools import partial
from argparse import Namespace
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import score_model_ref as smr, sampler_ref as spr
from helpers import to_graph, rel_err
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex
from disco_diffdock_amd.sampling import step_coefficients
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device('cuda:0')
d = os.path.join(ROOT, 'disco_diffdock_amd', 'data')
tables = (np.load(os.path.join(d, 'so3_exp_score_norms.npy')), np.load(os.path.join(d, 'torus_score_norm_seed0.npy')))
CFG = smr.ScoreModelConfig(latent_vocab=64)
args = Namespace(tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03, tor_sigma_max=3.14, no_torsion=False)
README = dict(temp_sampling=[1.886430780895051, 5.659562317960644, 2.8888668488630156],
              temp_psi=[0.07085125444659945, 2.686505606141324, 4.089493860493927],
              temp_sigma_data=[0.3617563913086843, 0.7437588205919711, 0.08897393057297842])
rng = np.random.default_rng(77)
worst = (0.0, '')
for case in range(N):
    seed = int(rng.integers(100000))
    n_res, n_lig = int(rng.choice([12, 30, 64])), int(rng.choice([12, 20, 31, 44]))
    B, steps = int(rng.choice([1, 2, 4])), int(rng.choice([2, 3]))
    lowtemp = bool(rng.integers(2))
    c = synthetic.make_complex(seed, n_res=n_res, n_lig=n_lig)
    P = smr.random_state_dict(CFG, seed=seed % 997)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, B)
    sched = get_t_schedule(steps)
    kw = README if lowtemp else dict(temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5)
    t_arr, sc, nc = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=args), args, False, False, True,
                                      kw['temp_sampling'], kw['temp_psi'], kw['temp_sigma_data'])
    pos0 = np.stack([c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    z = 0.3 * torch.randn(steps, B, 6 + cx.R, generator=torch.Generator().manual_seed(seed))
    pos = torch.from_numpy(pos0).to(dev)
    cx.sample(pos, t_arr, sc, nc, z.to(dev))
    dl = []
    for p in pos0:
        g = to_graph(c)
        g['ligand'].pos = torch.from_numpy(p).float()
        dl.append(g)
    nf = lambda b, t, name, shape: {'tr': z[t, :, 0:3], 'rot': z[t, :, 3:6], 'tor': z[t, :, 6:].reshape(-1)}[name]
    ref, _ = spr.sampling(dl, P, CFG, tables[0], tables[1], steps, sched, sched, sched, noise_fn=nf, batch_size=B, no_final_step_noise=True, **kw)
    ref = torch.cat([g['ligand'].pos for g in ref])
    err = rel_err(pos.cpu().reshape(-1, 3), ref)
    desc = f'case {case}: seed={seed} n_res={n_res} n_lig={len(c["lig_pos"])} R={cx.R} B={B} steps={steps} lowtemp={lowtemp}'
    print(desc, f'err={err:.2e}', '' if err < 1e-4 else '   <-- ABOVE 1e-4', flush=True)
    if err > worst[0]:
        worst = (err, desc)
    cx.close(); ctx.close()
print('worst:', worst)
