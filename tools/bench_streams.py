"""Development aid: the bench workload (8 complexes, 40 samples, 20 steps) with 1, 2 or 3 complexes in flight on separate HIP
streams of ONE context.  Prints complexes/s per setting."""
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
ctx = Context(device=0)
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
cs = [synthetic.make_complex(i, n_res=300) for i in range(8)]
cxs = [Complex(ctx, c, B.SAMPLES) for c in cs]
pos0 = [torch.from_numpy(B.start_poses(c, np.random.default_rng(i), B.SAMPLES)).to(dev) for i, c in enumerate(cs)]
noise = [torch.randn((B.STEPS, B.SAMPLES, 6 + cx.R), device=dev) for cx in cxs]
torch.cuda.synchronize()
ref = None
for ns in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(2):        # rep 0 = warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for k in range(8):
            with torch.cuda.stream(streams[k % ns]):
                p = pos0[k].clone()
                cxs[k].sample(p, t_arr, sc, nc, noise[k])
                outs.append(p)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    chk = torch.stack([o.sum() for o in outs]).cpu()
    if ref is None:
        ref = chk
    print(f'{ns} stream(s): {8 / dt:.2f} complexes/s ({1e3 * dt / 8:.2f} ms per complex); max checksum deviation vs 1 stream {float((chk - ref).abs().max()):.3e}')
