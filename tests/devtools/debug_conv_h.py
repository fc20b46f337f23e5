import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from oracle import score_model_ref as smr
from helpers import rel_err
from disco_diffdock_amd.runtime import Context
CFG = smr.ScoreModelConfig()
dev = torch.device('cuda:0')
l = int(sys.argv[1]); N = int(sys.argv[2]); per = int(sys.argv[3])
Pl = smr.random_conv_layer_params(CFG, l, 40 + l, True)
g = torch.Generator().manual_seed(5)
i_irr, o_irr = CFG.conv_irreps(l)
din, dout = smr.irreps_dim(i_irr), smr.irreps_dim(o_irr)
splits = [0, per, 2 * per, 3 * per, 4 * per]
E = splits[-1]
node = torch.randn(N, din, generator=g)
src = torch.cat([torch.sort(torch.randint(0, N, (per,), generator=g)).values for _ in range(4)])
dst = torch.randint(0, N, (E,), generator=g)
ea = torch.randn(E, 72, generator=g); sh = torch.randn(E, 4, generator=g)
outs = {}
for mode in (0, 1):
    ctx = Context(device=0, conv_f16x3=mode)
    ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in Pl.items()})
    outs[mode] = ctx.conv_forward(l, node.to(dev), src.to(dev), dst.to(dev), splits, ea.to(dev), sh.to(dev), dout).cpu().numpy()
e = np.abs(outs[1] - outs[0])
print('l', l, 'N', N, 'per', per, 'rel', e.max() / np.abs(outs[0]).max())
print('col err', np.round(e.max(0), 3))
bad = np.where(e.max(1) > 1e-3)[0]
print('bad nodes', len(bad), bad[:20])
# which edges feed the bad nodes: positions within their group
for gi in range(4):
    s_g = src[splits[gi]:splits[gi + 1]].numpy()
    pos = [int(np.where(s_g == b)[0][0]) for b in bad[:10] if (s_g == b).any()]
    print('group', gi, 'first positions of bad nodes', pos)
