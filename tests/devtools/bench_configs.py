"""Development aid (not the judged bench): wall time per complex of BASELINE.json configs 2, 3 and 4 on ONE GPU through the
reference's call surface ``sampling(data_list, model, ...)`` (host RNG draws, Python collate, per-call Complex creation included):
  config 2: DiffDock-S score model                                  40 samples x 20 steps
  config 3: DisCo-DiffDock-S score model + AR latent model           (latent_dim = 2: + 2 encoder forwards per batch; README CFG off)
  config 4: config 3 + all-atom confidence model on the final poses
Synthetic complexes (300 residues, ~2400 receptor atoms), random-init weights in the reference layouts.  Lives under tests/devtools
because the AR model's random weights come from the oracle's layout helper."""
import os, sys, time
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
N_C, S, STEPS = 4, 40, 20
rng = np.random.default_rng(0)
cs = []
for i in range(N_C):
    c = synthetic.make_complex(100 + i, n_res=300)
    synthetic.add_receptor_atoms(c, np.random.default_rng(i))
    cs.append(c)
sched = get_t_schedule(STEPS)


def data_lists(c):
    score_only = {k: v for k, v in c.items() if not k.startswith('atom_')}
    dl, cdl = [from_arrays(score_only) for _ in range(S)], [from_arrays(c) for _ in range(S)]
    for d in dl:
        p = torch.from_numpy(c['lig_pos'] + rng.normal(0, 5.0, size=(1, 3)).astype(np.float32)).float()
        d['ligand'].pos = p
        d['ligand'].ar_pos = p.clone()
    return dl, cdl


def run(name, model, args, **kw):
    times = []
    for rep in range(2):                       # rep 0 = warm-up
        t0 = time.perf_counter()
        for c in cs:
            dl, cdl = data_lists(c)
            extra = dict(kw)
            if 'confidence_model' in extra:
                extra['confidence_data_list'] = cdl
            out, conf = sampling(dl, model, STEPS, sched, sched, sched, dev, partial(t_to_sigma, args=args), args, batch_size=S,
                                 no_final_step_noise=True, **tg.README_S, **extra)
            torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / N_C)
    print(f'{name}: {1e3 * times[1]:.1f} ms per complex = {1 / times[1]:.2f} complexes/s (reference call surface, {S} samples x {STEPS} steps)', flush=True)


# config 2
m2 = get_model(tg.ARGS_S, dev, partial(t_to_sigma, args=tg.ARGS_S), no_parallel=True)
m2.score_model.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
run('config 2 (DiffDock-S)', m2, tg.ARGS_S, use_latent=False)
# config 3
score_args = Namespace(**dict(vars(tg.ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
ar_args = Namespace(use_pretrained_score=True, ns=16, latent_no_batchnorm=False, latent_dropout=0.0, latent_hidden_dim=128,
                    esm_embeddings_path='x', no_randomness=False)
cfg = smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
m3 = get_model(score_args, dev, partial(t_to_sigma, args=score_args), no_parallel=True)
m3.score_model.load_state_dict(smr.random_state_dict(cfg, seed=13))
ar = get_ar_model(ar_args, score_args, dev, training=False)
ar.load_state_dict(ar_ref.random_ar_state_dict(cfg, seed=14))
ar.eval()
run('config 3 (DisCo-DiffDock-S + AR latents)', m3, score_args, ar_model=ar, ar_args=ar_args, softmax_latent_temperature=float(np.exp(-1.5)))
# config 4
cm = get_model(tg.CONF_ARGS, dev, partial(t_to_sigma, args=tg.CONF_ARGS), no_parallel=True, confidence_mode=True)
cm.load_state_dict(synthetic.random_confidence_state_dict(seed=1), strict=True)
cm.eval()
run('config 4 (config 3 + confidence model)', m3, score_args, ar_model=ar, ar_args=ar_args, softmax_latent_temperature=float(np.exp(-1.5)),
    confidence_model=cm, confidence_model_args=tg.CONF_ARGS)
