"""Wall time of whole score-model forwards on FIXED inputs (the SAME 300-residue complex, the SAME 40 poses, four diffusion times), events on the launch
stream: for A/Bs of the launch structure (tools/build_variant_model.sh), where the conv kernels' own event times say nothing.  DDK_LIB selects the library.

    python tools/forward_time.py [--reps 30] [--pocket]        ms per forward at t = 1.0 / 0.6 / 0.2 / 0.05 and their mean"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=30)
ap.add_argument('--spread', type=float, default=12.0, help='sigma (A) of the poses around the pocket: 12 ~ the late steps of the default workload, 1 = pocket-bound')
a = ap.parse_args()
dev = torch.device('cuda:0')
ctx = Context(device=0)
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
c = synthetic.make_complex(0, n_res=300)
B = 40
cx = Complex(ctx, c, B)
rng = np.random.default_rng(0)
pos = torch.from_numpy(np.stack([c['lig_pos'] + rng.normal(0, a.spread, size=(1, 3)) for _ in range(B)]).astype(np.float32)).to(dev)
tot = []
for t in (1.0, 0.6, 0.2, 0.05):
    for _ in range(5):
        cx.score_forward(pos, t, t, t)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(a.reps):
        cx.score_forward(pos, t, t, t)
    en.record()
    torch.cuda.synchronize()
    ms = st.elapsed_time(en) / a.reps
    tot.append(ms)
    print('t=%.2f  %.4f ms per forward   cross edges per sample %d' % (t, ms, cx.graph_stats()['E_lr'] // B))
print('mean ms per forward: %.4f' % (sum(tot) / len(tot)))
