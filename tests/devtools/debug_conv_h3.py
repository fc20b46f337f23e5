import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from oracle import score_model_ref as smr
from disco_diffdock_amd.runtime import Context
CFG = smr.ScoreModelConfig()
dev = torch.device('cuda:0')
l, N, per, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
Pl = smr.random_conv_layer_params(CFG, l, 40 + l, True)
g = torch.Generator().manual_seed(5)
i_irr, o_irr = CFG.conv_irreps(l)
din, dout = smr.irreps_dim(i_irr), smr.irreps_dim(o_irr)
splits = [0, per, 2 * per, 3 * per, 4 * per]
E = splits[-1]
node = torch.randn(N, din, generator=g).to(dev)
src = torch.cat([torch.sort(torch.randint(0, N, (per,), generator=g)).values for _ in range(4)]).to(dev)
dst = torch.randint(0, N, (E,), generator=g).to(dev)
ea = torch.randn(E, 72, generator=g).to(dev); sh = torch.randn(E, 4, generator=g).to(dev)
ref = None
for mode in (0, 1):
    ctx = Context(device=0, conv_f16x3=mode)
    ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in Pl.items()})
    bad = 0
    for r in range(reps):
        out = ctx.conv_forward(l, node, src, dst, splits, ea, sh, dout).cpu().numpy()
        if ref is None:
            ref = out
        e = np.abs(out - ref).max() / np.abs(ref).max()
        if e > 1e-4:
            bad += 1
            rows = np.where(np.abs(out - ref).max(1) > 1e-3)[0]
            cols = np.where(np.abs(out - ref).max(0) > 1e-3)[0]
            print('  mode', mode, 'rep', r, 'err', e, 'rows', rows[:12], 'cols', cols[:16])
    print('mode', mode, 'bad runs', bad, 'of', reps)
