"""Timing of the all-atom confidence forward (not the judged bench): 40 poses of one 300-residue complex (~2400 receptor atoms)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex
dev = torch.device('cuda:0')
B = 40
c = synthetic.make_complex(0, n_res=300)
synthetic.add_receptor_atoms(c, np.random.default_rng(0))
ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
ctx.load_state_dict(synthetic.random_confidence_state_dict(seed=1))
t0 = time.time()
cx = Complex(ctx, c, max_batch=B)
cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
t_create = time.time() - t0
rng = np.random.default_rng(1)
pos = torch.as_tensor(np.stack([c['lig_pos'] + rng.normal(0, 1.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)).to(dev)
for _ in range(2):
    out = cx.confidence_forward(pos)
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(10):
    out = cx.confidence_forward(pos)
en.record(); torch.cuda.synchronize()
print(f'n_atom={len(c["atom_x"])} E_aa={c["atom_edge_index"].shape[1]} counts={cx.confidence_counts()}')
print(f'confidence forward: {st.elapsed_time(en) / 10:.3f} ms per batch of {B} poses; complex_create + set_atoms {t_create * 1e3:.0f} ms; out[0]={out[0].tolist()}')
