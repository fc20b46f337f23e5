"""Experiment: does it pay to run TWO reverse-diffusion loops side by side, each on half of the CUs?  The 18 % of a step that is not the conv
kernel (20 small launches, latency bound) would then run beside the other loop's conv launches.  Two contexts (a ddk_ctx is not meant for
concurrent launches), two streams, `ddk_debug_set_conv_workgroups` = 128 each.
    (a) one loop, 40 samples, 256 workgroups (what sampling() does)
    (b) the 40 samples of a complex as two 20-sample loops side by side
    (c) two different complexes side by side, 40 samples each"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench as B
from functools import partial
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex
from disco_diffdock_amd.sampling import step_coefficients
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule

dev = torch.device('cuda:0')
margs = B.ARGS_S
sched = get_t_schedule(B.STEPS)
t_arr, sc, nc = step_coefficients(B.STEPS, sched, sched, sched, partial(t_to_sigma, args=margs), margs, False, False, True,
                                  B.README_S['temp_sampling'], B.README_S['temp_psi'], B.README_S['temp_sigma_data'])
P = synthetic.random_score_model_state_dict(seed=0)
ctxs = []
for _ in range(2):
    c = Context(device=0)
    c.load_state_dict(P)
    ctxs.append(c)
cs = [synthetic.make_complex(i, n_res=300) for i in range(8)]
S = B.SAMPLES
pos0 = [torch.from_numpy(B.start_poses(c, np.random.default_rng(i), S)).to(dev) for i, c in enumerate(cs)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(mode, wg):
    for c in ctxs:
        c._check(c.L.ddk_debug_set_conv_workgroups(c.h, wg), 'wg')
    if mode == 'a':
        cxs = [Complex(ctxs[0], c, S) for c in cs]
        noise = [torch.randn((B.STEPS, S, 6 + cx.R), device=dev) for cx in cxs]
    elif mode == 'b':
        cxs = [(Complex(ctxs[0], c, S // 2), Complex(ctxs[1], c, S // 2)) for c in cs]
        noise = [torch.randn((B.STEPS, S, 6 + cx[0].R), device=dev) for cx in cxs]
    else:
        cxs = [Complex(ctxs[k % 2], c, S) for k, c in enumerate(cs)]
        noise = [torch.randn((B.STEPS, S, 6 + cx.R), device=dev) for cx in cxs]
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(8):
            if mode == 'a':
                p = pos0[k].clone()
                cxs[k].sample(p, t_arr, sc, nc, noise[k])
            elif mode == 'b':
                for h in range(2):
                    with torch.cuda.stream(streams[h]):
                        p = pos0[k][h * (S // 2):(h + 1) * (S // 2)].clone()
                        cxs[k][h].sample(p, t_arr, sc, nc, noise[k][:, h * (S // 2):(h + 1) * (S // 2)].contiguous())
            else:
                with torch.cuda.stream(streams[k % 2]):
                    p = pos0[k].clone()
                    cxs[k].sample(p, t_arr, sc, nc, noise[k])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    for cx in cxs:
        for x in (cx if isinstance(cx, tuple) else (cx,)):
            x.close()
    return 8 / best


for mode, wg, name in (('a', 256, 'one loop, 40 samples, 256 workgroups'), ('b', 128, 'two 20-sample loops side by side, 128 workgroups each'),
                       ('b', 256, 'two 20-sample loops side by side, 256 workgroups each'), ('c', 128, 'two complexes side by side, 128 workgroups each'),
                       ('c', 256, 'two complexes side by side, 256 workgroups each'), ('a', 256, 'one loop again')):
    print(f'{name:62s} {run(mode, wg):6.2f} complexes/s', flush=True)
