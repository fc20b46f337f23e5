"""Development aid: a few reverse-diffusion trajectories of one synthetic complex, meant to be run under
`rocprofv3 --kernel-trace --stats` (optionally with DDK_LIB=<variant>) to read the small kernels' durations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench as B
from functools import partial
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex
from disco_diffdock_amd.sampling import step_coefficients
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule

dev = torch.device('cuda:0')
c = synthetic.make_complex(0, n_res=B.N_RES)
margs = B.model_args()
sched = get_t_schedule(B.STEPS)
t_arr, sc, nc = step_coefficients(B.STEPS, sched, sched, sched, partial(t_to_sigma, args=margs), margs, False, False, True,
                                  B.README_S['temp_sampling'], B.README_S['temp_psi'], B.README_S['temp_sigma_data'])
ctx = Context(device=0)
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
cx = Complex(ctx, c, B.SAMPLES)
pos0 = torch.from_numpy(B.start_poses(c, np.random.default_rng(0), B.SAMPLES)).to(dev)
noise = torch.randn((B.STEPS, B.SAMPLES, 6 + cx.R), device=dev)
for _ in range(3):
    pos = pos0.clone()
    cx.sample(pos, t_arr, sc, nc, noise)
torch.cuda.synchronize()
print('R', cx.R, 'n_lig', cx.n_lig)
