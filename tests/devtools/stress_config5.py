"""BASELINE config 5 shape (large pocket: ~2000 C-alpha, 24-NN receptor graph, 40 samples): parity of a 2-sample forward
against the oracle and timing / graph statistics of the 40-sample forward."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import score_model_ref as smr, sampler_ref as spr
from helpers import batch_of, rel_err
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex

dev = torch.device('cuda:0')
cfg = smr.ScoreModelConfig(latent_vocab=64)
P = synthetic.random_score_model_state_dict(seed=2)
c = synthetic.make_complex(5, n_res=2000)
ctx = Context(device=0)
ctx.load_state_dict(P)
d = os.path.join(ROOT, 'disco_diffdock_amd', 'data')
tables = (np.load(os.path.join(d, 'so3_exp_score_norms.npy')), np.load(os.path.join(d, 'torus_score_norm_seed0.npy')))
rng = np.random.default_rng(0)
t = 0.6
B = 2
pos = np.stack([c['lig_pos'] + rng.normal(0, 6.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
cx = Complex(ctx, c, max_batch=40)
tr, rot, tor = cx.score_forward(torch.from_numpy(pos).to(dev), t, t, t)
print('B=2 graph', cx.graph_stats())
b = batch_of(c, B, pos)
spr.set_time(b, t, t, t, B)
t0 = time.time()
tr_r, rot_r, tor_r = smr.score_model_forward(P, cfg, b, tables[0], tables[1])
print(f'oracle forward {time.time() - t0:.1f} s; rel err tr {rel_err(tr.cpu(), tr_r):.2e} rot {rel_err(rot.cpu(), rot_r):.2e} tor {rel_err(tor.cpu(), tor_r):.2e}')
B = 40
pos = np.stack([c['lig_pos'] + rng.normal(0, 6.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
p = torch.from_numpy(pos).to(dev)
for tt in (1.0, 0.6, 0.05):
    cx.score_forward(p, tt, tt, tt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        cx.score_forward(p, tt, tt, tt)
    torch.cuda.synchronize()
    print(f't={tt}: B=40 forward {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms  graph {cx.graph_stats()}  mem {torch.cuda.memory_allocated() / 1e9:.2f} GB torch')
