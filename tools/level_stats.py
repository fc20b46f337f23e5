"""Receptive-field level sizes along the bench's trajectories (not the judged bench): per reverse step the share of the rec-rec edges received by level-A / A+B /
A+B+C residues (ddk_last_graph_stats), for the default workload (randomize_position start) and the pocket-bound one.  Sizes the layer-1 sharing idea of DESIGN.md 8."""
import os, sys
from functools import partial
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench as b
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Complex
from disco_diffdock_amd.model_utils import get_model
from disco_diffdock_amd.sampling import step_coefficients, draw_noise
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule

dev = torch.device('cuda:0')
margs, temps = b.ARGS_S, b.README_S
tsig = partial(t_to_sigma, args=margs)
model = get_model(margs, dev, tsig, no_parallel=True)
sm = getattr(model, 'score_model', model)
sm.load_state_dict(synthetic.random_score_model_state_dict(seed=0), strict=True)
ctx = sm.ctx
STEPS, B = 20, 40
sched = get_t_schedule(STEPS)
t_arr, sc, nc = step_coefficients(STEPS, sched, sched, sched, tsig, margs, False, False, True, temps['temp_sampling'], temps['temp_psi'], temps['temp_sigma_data'])
for name in ('default', 'pocket'):
    acc = np.zeros((STEPS, 4))
    for seed in range(3):
        c = synthetic.make_complex(seed, n_res=300)
        cx = Complex(ctx, c, B)
        rng = np.random.default_rng(seed)
        if name == 'default':
            pos = b.start_poses(c, rng, B, ctx, dev)
        else:
            pos = b.pocket_poses(c, rng, B)
        pos = torch.as_tensor(pos).to(dev).float().reshape(B, -1, 3).contiguous().clone()
        z = draw_noise(STEPS, B, cx.R, cx.R, nc, dev)
        if name == 'pocket':
            z = z * 0.2
        for k in range(STEPS):
            cx.sample(pos, t_arr[k:k + 1], sc[k:k + 1], nc[k:k + 1], z[k:k + 1].contiguous())
            g = cx.graph_stats()
            tot = B * g['E_rr'] // B if False else g['E_rr']
            a, ab, abc = g['E_rr_live']
            acc[k] += np.array([a / tot, ab / tot, abc / tot, g['E_lr'] / B])
    acc /= 3
    print(name)
    for k in range(STEPS):
        print('  step %2d t %.2f   A %.3f  A+B %.3f  A+B+C %.3f   cross/sample %6.0f' % (k, t_arr[k][0], acc[k][0], acc[k][1], acc[k][2], acc[k][3]))
    print('  mean over steps: C - B share of the rec-rec edges (what a layer-1 sharing pass could drop) %.3f' % float(np.mean(acc[:, 2] - acc[:, 1])))
