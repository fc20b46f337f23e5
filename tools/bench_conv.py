"""Kernel-level timing of the fused TP-conv kernel (not the judged bench): E edges of one conv layer."""
import argparse
import sys, os
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from disco_diffdock_amd.runtime import Context
from disco_diffdock_amd import runtime

p = argparse.ArgumentParser()
p.add_argument('--layer', type=int, default=3)
p.add_argument('--edges', type=int, default=800000)
p.add_argument('--nodes', type=int, default=13200)
p.add_argument('--iters', type=int, default=10)
a = p.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
W = [720, 936, 1152, 1872, 1872][a.layer]
din = [24, 42, 60, 84, 84][a.layer]
P = {}
for grp in range(4):
    P[f'conv_layers.{a.layer}.fc.{grp}.0.weight'] = torch.randn(72, 72, generator=g) / 8.5
    P[f'conv_layers.{a.layer}.fc.{grp}.0.bias'] = torch.randn(72, generator=g) * 0.1
    P[f'conv_layers.{a.layer}.fc.{grp}.4.weight'] = torch.randn(W, 72, generator=g) / 8.5
    P[f'conv_layers.{a.layer}.fc.{grp}.4.bias'] = torch.randn(W, generator=g) * 0.1
ctx = Context(device=0, batch_norm=0)
ctx.load_state_dict(P)
E, N = a.edges, a.nodes
x = torch.randn(N, din, device=dev)
src = torch.sort(torch.randint(0, N, (E,), device=dev)).values.int()
dst = torch.randint(0, N, (E,), device=dev).int()
ea = torch.randn(E, 72, device=dev)
sh = torch.randn(E, 4, device=dev)
go = [0, E // 40, E // 2, 3 * E // 4, E]
dout = [42, 60, 84, 84, 84][a.layer]
for _ in range(2):
    ctx.conv_forward(a.layer, x, src, dst, go, ea, sh, dout)
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(a.iters):
    ctx.conv_forward(a.layer, x, src, dst, go, ea, sh, dout)
en.record()
torch.cuda.synchronize()
ms = st.elapsed_time(en) / a.iters
flop = E * (2 * 72 * (72 + W) + 2 * W * 1.0)
print(f'layer {a.layer} E={E}: {ms:.3f} ms/call (incl. pad/finalize)  {flop / ms / 1e9:.1f} TFLOP/s  {E / ms / 1e3:.2f} Medges/s')
