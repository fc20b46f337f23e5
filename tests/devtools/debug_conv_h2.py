import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from oracle import score_model_ref as smr
from disco_diffdock_amd.runtime import Context
CFG = smr.ScoreModelConfig()
dev = torch.device('cuda:0')
l, per = int(sys.argv[1]), int(sys.argv[2])
Pl = smr.random_conv_layer_params(CFG, l, 40 + l, True)
g = torch.Generator().manual_seed(5)
i_irr, o_irr = CFG.conv_irreps(l)
din, dout = smr.irreps_dim(i_irr), smr.irreps_dim(o_irr)
splits = [0, per, per, per, per]
E = per
N = E + 10
node = torch.randn(N, din, generator=g)
src = torch.arange(E)
dst = torch.randint(0, N, (E,), generator=g)
ea = torch.randn(E, 72, generator=g); sh = torch.randn(E, 4, generator=g)
outs = {}
for mode in (0, 1):
    ctx = Context(device=0, conv_f16x3=mode)
    ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in Pl.items()})
    outs[mode] = ctx.conv_forward(l, node.to(dev), src.to(dev), dst.to(dev), splits, ea.to(dev), sh.to(dev), dout).cpu().numpy()
e = np.abs(outs[1] - outs[0])[:E]
per_edge = e.max(1)
for w in range((E + 31) // 32):
    seg = per_edge[32 * w:32 * w + 32]
    print('edges', 32 * w, '..', 'max err', float(seg.max()), 'bad lanes', np.where(seg > 1e-3)[0][:40])
print('cols bad', np.where(e.max(0) > 1e-3)[0])
