"""Per-call host durations of the expensive pieces of sampling() (which calls block, and for how long) while bench.py runs.
Run on the GPU box:  python tools/host_calls.py --config 4 --no-cpu-baseline --no-alt"""
import sys, os, time, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench
from disco_diffdock_amd import runtime, score_model, confidence, sampling, diffusion_utils, pretrained_score_encoder as pse
T = collections.defaultdict(list)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label].append(round(1e3 * (time.perf_counter() - t0), 2))
    setattr(obj, name, g)
wrap(runtime.Complex, 'set_atoms'); wrap(runtime.Complex, '__init__', 'Complex()'); wrap(runtime.Complex, 'sample')
wrap(runtime.Complex, 'confidence_forward'); wrap(score_model, '_fingerprint'); wrap(sampling, 'complex_for_batch', 'complex_for_batch(score)')
wrap(confidence.ConfidenceModel, 'complex_for', 'conf.complex_for'); wrap(sampling, 'set_time'); wrap(pse.GenericEncoder, 'encode_ar')
wrap(sampling, 'sampling'); wrap(sampling, 'draw_noise'); wrap(sampling, 'h2d_async')
import torch
wrap(score_model, '_bytes_of'); wrap(score_model, '_first_view'); wrap(torch.Tensor, 'double', 'Tensor.double'); wrap(torch.Tensor, 'item', 'Tensor.item'); wrap(torch.Tensor, 'sum', 'Tensor.sum')
bench.main()
for k, v in T.items():
    print(f'{k:28s} n={len(v):3d} sum={sum(v):8.1f} ms  {v[-16:]}', file=sys.stderr)
