"""Development aid: cProfile of the host side of sampling() (reference call surface) for one synthetic complex, 40 samples x 20 steps."""
import os, sys, time, cProfile, pstats
from functools import partial
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.model_utils import get_model
from disco_diffdock_amd.sampling import sampling
from disco_diffdock_amd.data import from_arrays
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
import test_gpu_model as tg
dev = torch.device('cuda:0')
S, STEPS = 40, 20
cs = [synthetic.make_complex(100 + i, n_res=300) for i in range(3)]
sched = get_t_schedule(STEPS)
m2 = get_model(tg.ARGS_S, dev, partial(t_to_sigma, args=tg.ARGS_S), no_parallel=True)
m2.score_model.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
rng = np.random.default_rng(0)


def once(c):
    t0 = time.perf_counter()
    dl = [from_arrays(c) for _ in range(S)]
    for d in dl:
        d['ligand'].pos = torch.from_numpy(c['lig_pos'] + rng.normal(0, 5.0, size=(1, 3)).astype(np.float32)).float()
    t1 = time.perf_counter()
    out, conf = sampling(dl, m2, STEPS, sched, sched, sched, dev, partial(t_to_sigma, args=tg.ARGS_S), tg.ARGS_S, batch_size=S,
                         no_final_step_noise=True, use_latent=False, **tg.README_S)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


once(cs[0])
print('same complex again: data_list %.1f ms, sampling() returns after %.1f ms, final sync %.1f ms' % tuple(1e3 * v for v in once(cs[0])))
print('new complex:        data_list %.1f ms, sampling() returns after %.1f ms, final sync %.1f ms' % tuple(1e3 * v for v in once(cs[1])))
pr = cProfile.Profile()
pr.enable()
once(cs[2])
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
