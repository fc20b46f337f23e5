"""Development aid: cProfile of sampling() on the DisCo path (AR latent model + latent-conditioned score model), one complex."""
import os, sys, time, cProfile, pstats
from argparse import Namespace
from functools import partial
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import score_model_ref as smr, ar_ref
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.model_utils import get_model, get_ar_model
from disco_diffdock_amd.sampling import sampling
from disco_diffdock_amd.data import from_arrays
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
import test_gpu_model as tg
dev = torch.device('cuda:0')
S, STEPS = 40, 20
cs = [synthetic.make_complex(100 + i, n_res=300) for i in range(3)]
sched = get_t_schedule(STEPS)
score_args = Namespace(**dict(vars(tg.ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
ar_args = Namespace(use_pretrained_score=True, ns=16, latent_no_batchnorm=False, latent_dropout=0.0, latent_hidden_dim=128,
                    esm_embeddings_path='x', no_randomness=False)
cfg = smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
m3 = get_model(score_args, dev, partial(t_to_sigma, args=score_args), no_parallel=True)
m3.score_model.load_state_dict(smr.random_state_dict(cfg, seed=13))
ar = get_ar_model(ar_args, score_args, dev, training=False)
ar.load_state_dict(ar_ref.random_ar_state_dict(cfg, seed=14))
ar.eval()
rng = np.random.default_rng(0)


def once(c):
    dl = [from_arrays(c) for _ in range(S)]
    for d in dl:
        p = torch.from_numpy(c['lig_pos'] + rng.normal(0, 5.0, size=(1, 3)).astype(np.float32)).float()
        d['ligand'].pos = p
        d['ligand'].ar_pos = p.clone()
    t1 = time.perf_counter()
    sampling(dl, m3, STEPS, sched, sched, sched, dev, partial(t_to_sigma, args=score_args), score_args, batch_size=S,
             no_final_step_noise=True, ar_model=ar, ar_args=ar_args, softmax_latent_temperature=float(np.exp(-1.5)), **tg.README_S)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    return t2 - t1, time.perf_counter() - t2


once(cs[0])
print('new complex: sampling() returns after %.1f ms, final sync %.1f ms' % tuple(1e3 * v for v in once(cs[1])))
pr = cProfile.Profile()
pr.enable()
once(cs[2])
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
