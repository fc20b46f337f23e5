"""BASELINE metric, second clause, at the REFERENCE's op boundary: FasterTensorProduct.forward (models/tensor_layers.py:65-116) with the per-edge
weights [E, W] resident in HBM (SURVEY.md 8(d) boundary (A): 8 184 B per edge at layers 3 / 4, 0.7 FLOP/B - HBM-bound).  Times ddk_tp_forward
(tp_stream_kernel, k_tp.hip) with events on the launch stream and checks a slice against the fp64 restatement below.

    python tools/bench_tp.py [--layer 3] [--edges 800000] [--json out.json]        (run it under rocprofv3 --kernel-trace / --pmc for the profile)"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd.tensor_layers import FasterTensorProduct

p = argparse.ArgumentParser()
p.add_argument('--layer', type=int, default=3)
p.add_argument('--edges', type=int, default=800000)
p.add_argument('--iters', type=int, default=10)
p.add_argument('--json', default=None)
a = p.parse_args()
dev = torch.device('cuda:0')
seq = ['24x0e', '24x0e+6x1o', '24x0e+6x1o+6x1e', '24x0e+6x1o+6x1e+24x0o']
i_irr, o_irr = seq[min(a.layer, 3)], seq[min(a.layer + 1, 3)]
tp = FasterTensorProduct(i_irr, '1x0e+1x1o', o_irr)
E, W = a.edges, tp.weight_numel
din = {0: 24, 1: 42, 2: 60, 3: 84, 4: 84}[a.layer]
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(E, din, device=dev, generator=g)
sh = torch.randn(E, 4, device=dev, generator=g)
w = torch.randn(E, W, device=dev, generator=g)
out = tp(x, sh, w)
torch.cuda.synchronize()
# fp64 restatement of tensor_layers.py:65-116 on a slice (the committed goldens of the unmodified class are checked by tests/test_gpu_ops.py)
n = 4096
xs, ss, ws = x[:n].double().cpu(), sh[:n].double().cpu(), w[:n].double().cpu()
import re
mul = {k: 0 for k in ('0e', '1o', '1e', '0o')}
for c in i_irr.split('+'):
    m, ir = c.split('x'); mul[ir] = int(m)
omul = {k: 0 for k in ('0e', '1o', '1e', '0o')}
for c in o_irr.split('+'):
    m, ir = c.split('x'); omul[ir] = int(m)
o = 0
A_ = xs[:, o:o + mul['0e']]; o += mul['0e']
P_ = xs[:, o:o + 3 * mul['1o']].reshape(n, -1, 3); o += 3 * mul['1o']
Q_ = xs[:, o:o + 3 * mul['1e']].reshape(n, -1, 3); o += 3 * mul['1e']
C_ = xs[:, o:o + mul['0o']]
s0, v = ss[:, :1], ss[:, 1:]
rows = {
    '0e': torch.cat([A_ * s0, (P_ * v[:, None]).sum(-1) / 3 ** 0.5], 1)[..., None],
    '1o': torch.cat([A_[..., None] * v[:, None], P_ * s0[..., None], torch.cross(Q_, v[:, None].expand_as(Q_), dim=-1) / 2 ** 0.5], 1),
    '1e': torch.cat([torch.cross(P_, v[:, None].expand_as(P_), dim=-1) / 2 ** 0.5, Q_ * s0[..., None], C_[..., None] * v[:, None]], 1),
    '0o': torch.cat([(Q_ * v[:, None]).sum(-1) / 3 ** 0.5, C_ * s0], 1)[..., None],
}
ref, off = [], 0
for k in ('0e', '1o', '1e', '0o'):
    n_in, n_out = rows[k].shape[1], omul[k]
    if n_out == 0:
        continue
    wk = ws[:, off:off + n_in * n_out].reshape(n, n_in, n_out); off += n_in * n_out
    if n_in:
        ref.append((torch.einsum('eic,eio->eoc', rows[k], wk) / n_in ** 0.5).reshape(n, -1))
ref = torch.cat(ref, 1)
err = float((out[:n].double().cpu() - ref).abs().max() / ref.abs().max())
assert off == W and err < 1e-5, (off, W, err)
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    tp(x, sh, w)
torch.cuda.synchronize()
st.record()
for _ in range(a.iters):
    tp(x, sh, w)
en.record()
torch.cuda.synchronize()
ms = st.elapsed_time(en) / a.iters
bytes_per_edge = 4 * (W + din + 4) + 8 + 4 * out.shape[1]      # SURVEY.md 8(d) boundary (A) (incl. the two int32 indices of the gather the reference does first)
moved = 4 * (W + din + 4) + 4 * out.shape[1]                   # what this entry point really reads / writes (x_dst arrives gathered)
res = {'layer': a.layer, 'edges': E, 'W': W, 'ms_per_call': ms, 'algorithmic_bytes_per_edge': bytes_per_edge, 'bytes_moved_per_edge': moved,
       'achieved_GBps_algorithmic': E * bytes_per_edge / ms / 1e6, 'frac_of_8000': E * bytes_per_edge / ms / 1e6 / 8000, 'frac_of_6300': E * bytes_per_edge / ms / 1e6 / 6300,
       'Medges_per_s': E / ms / 1e3, 'max_rel_err_vs_fp64': err}
print(json.dumps(res))
if a.json:
    json.dump(res, open(a.json, 'w'), indent=1)
