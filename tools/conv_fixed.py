"""Fixed-input timing of the conv kernel for kernel experiments: the SAME 300-residue complex and the SAME 40 poses at three diffusion times,
forwards repeated, conv launches timed by the library's HIP events (ddk_profile_read).  Unlike bench.py's trajectories the edge counts do not
depend on what the kernel computes, so builds whose results differ (ablations) can be compared too.  DDK_LIB selects the library build.

    python tools/conv_fixed.py [--reps 20]          prints ms per conv launch by layer and the mean over the three times"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--B', type=int, default=40, help='samples in the batch (5: the node rows of a forward fit one XCD-sized L2)')
a = ap.parse_args()
dev = torch.device('cuda:0')
ctx = Context(device=0)
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
c = synthetic.make_complex(0, n_res=300)
B = a.B
cx = Complex(ctx, c, B)
rng = np.random.default_rng(0)
pos = torch.from_numpy(np.stack([c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)).to(dev)
tot = 0.0
for t in (1.0, 0.6, 0.2):
    for _ in range(3):
        cx.score_forward(pos, t, t, t)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    for _ in range(a.reps):
        cx.score_forward(pos, t, t, t)
    torch.cuda.synchronize()
    p = ctx.profile_read()
    ctx.profile_enable(False)
    ms = [q['ms'] / max(q['launches'], 1) for q in p]
    tot += sum(ms)
    print('t=%.1f  ' % t + '  '.join('L%d %.4f' % (l, m) for l, m in enumerate(ms)) + '   sum %.4f ms   edges %s' % (sum(ms), [q['edges'] // max(q['launches'], 1) for q in p]))
print('mean conv ms per forward over the three times: %.4f' % (tot / 3))
print('ns per edge and layer (t = 0.2): ' + '  '.join('L%d %.3f' % (l, 1e6 * q['ms'] / max(q['edges'], 1)) for l, q in enumerate(p)))
