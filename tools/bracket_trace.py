"""Where the sampling() bracket spends its time beyond the resident loop: host time stamps of every call (enter, Complex ready, sampler
enqueued, exit) next to the device time stamps of the call's first and last kernel (events, aligned to the host clock at a sync).
Run on the GPU box:  python tools/bracket_trace.py [n_calls]"""
import sys, os, time, copy
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import bench
from functools import partial
from disco_diffdock_amd import synthetic, score_model as sm_mod, sampling as smp
from disco_diffdock_amd.data import from_arrays
from disco_diffdock_amd.model_utils import get_model
from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
margs, temps = bench.ARGS_S, bench.README_S
tsig = partial(t_to_sigma, args=margs)
model = get_model(margs, dev, tsig, no_parallel=True)
model.load_state_dict(synthetic.random_score_model_state_dict(seed=0), strict=True)
sched = get_t_schedule(bench.STEPS)
calls = []
for i in range(n_calls + 2):
    c = synthetic.make_complex(i, n_res=300)
    g0 = from_arrays(c)
    poses = bench.start_poses(c, np.random.default_rng(i), bench.SAMPLES)
    dl = [copy.copy(g0) for _ in range(bench.SAMPLES)]
    for d, p in zip(dl, poses):
        d['ligand'].pos = torch.from_numpy(p)
    calls.append(dl)

marks = []
orig_cfb, orig_draw = smp.complex_for_batch, smp.draw_noise
def cfb(*a, **k):
    marks[-1]['pre_cx'] = time.perf_counter()
    r = orig_cfb(*a, **k)
    marks[-1]['cx'] = time.perf_counter()
    return r
def draw(*a, **k):
    marks[-1]['pre_noise'] = time.perf_counter()
    r = orig_draw(*a, **k)
    marks[-1]['noise'] = time.perf_counter()
    return r
smp.complex_for_batch, smp.draw_noise = cfb, draw

def run(k):
    m = {'enter': time.perf_counter(), 'ev0': torch.cuda.Event(enable_timing=True), 'ev1': torch.cuda.Event(enable_timing=True)}
    marks.append(m)
    m['ev0'].record()
    smp.sampling(calls[k], model, bench.STEPS, sched, sched, sched, dev, tsig, margs, batch_size=bench.SAMPLES, no_final_step_noise=True,
                 use_latent=False, **temps)
    m['ev1'].record()
    m['exit'] = time.perf_counter()

for k in range(2):
    run(k)
torch.cuda.synchronize()
sm_mod._complex_cache.clear()
marks.clear()
base = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
base.record(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(2, 2 + n_calls):
    run(k)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f'{n_calls} calls: {1e3 * (t1 - t0) / n_calls:.2f} ms per call')
print('call | host: enter  +collate  +complex  +to noise  +noise  +sample..exit | device: begin  end   (ms from t0) | idle before begin')
prev_end = 0.0
for k, m in enumerate(marks):
    h = lambda x: 1e3 * (m[x] - t0)
    d0, d1 = base.elapsed_time(m['ev0']), base.elapsed_time(m['ev1'])
    print(f'{k:3d}  | {h("enter"):8.2f} {1e3 * (m["pre_cx"] - m["enter"]):8.2f} {1e3 * (m["cx"] - m["pre_cx"]):8.2f} '
          f'{1e3 * (m["pre_noise"] - m["cx"]):8.2f} {1e3 * (m["noise"] - m["pre_noise"]):8.2f} {1e3 * (m["exit"] - m["noise"]):8.2f} | '
          f'{d0:8.2f} {d1:8.2f} | len {d1 - d0:6.2f}  gap {d0 - prev_end:6.2f}')
    prev_end = d1
