"""Host-side cost around the timed region of bench.py: ddk_complex_create (static precompute + uploads) per complex."""
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
    t0 = time.perf_counter()
    for _ in range(5):
        cx = Complex(ctx, c, 40)
        torch.cuda.synchronize()
        cx.close()
    print(f'n_res={n_res}: ddk_complex_create (max_batch 40) {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per complex')
