"""Host-side cost of ddk_complex_create (static precompute + staged uploads) and, for the confidence model, ddk_complex_set_atoms,
per complex: the time the CALL takes on the host (the uploads themselves run on the context's upload stream beside the compute stream)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex
ctx = Context(device=0)
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
for n_res in (300, 2000):
    c = synthetic.make_complex(1, n_res=n_res)
    Complex(ctx, c, 40).close()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        cx = Complex(ctx, c, 40)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        cx.close()
    print(f'n_res={n_res}: ddk_complex_create (max_batch 40) host call {1e3 * np.median(ts):.2f} ms per complex')
cctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
cctx.load_state_dict(synthetic.random_confidence_state_dict(seed=1))
c = synthetic.make_complex(1, n_res=300)
synthetic.add_receptor_atoms(c, np.random.default_rng(1))
t_c, t_a = [], []
for rep in range(6):
    t0 = time.perf_counter()
    cx = Complex(cctx, c, 40)
    t1 = time.perf_counter()
    cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    cx.close()
    if rep:
        t_c.append(t1 - t0); t_a.append(t2 - t1)
print(f'confidence complex ({len(c["atom_x"])} atoms): create {1e3 * np.median(t_c):.2f} ms, set_atoms {1e3 * np.median(t_a):.2f} ms (host calls)')
