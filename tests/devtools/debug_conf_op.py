import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from oracle import confidence_ref as cr, e3nn_lite as o3, score_model_ref as smr
from helpers import rel_err
from disco_diffdock_amd.runtime import Context
dev = torch.device('cuda:0')
cfg = cr.ConfidenceModelConfig()
l = 1
P = {k: v for k, v in cr.random_state_dict(cfg, seed=61).items() if k.startswith('conv_layers.')}
ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
ctx.load_state_dict(P)
i_irr, o_irr = cfg.conv_irreps(l)
din, dout = smr.irreps_dim(i_irr), smr.irreps_dim(o_irr)
g = torch.Generator().manual_seed(3)
for N, E, sort in ((300, 5000, True), (300, 5000, False), (3000, 6564, False), (3000, 20000, False)):
    node = torch.randn(N, din, generator=g)
    src = torch.randint(0, N, (E,), generator=g)
    if sort:
        src = torch.sort(src).values
    dst = torch.randint(0, N, (E,), generator=g)
    ea = torch.randn(E, 72, generator=g)
    sh9 = o3.spherical_harmonics(cfg.sh_irreps, torch.randn(E, 3, generator=g), normalize=True, normalization='component')
    for k in (3,):
        want = cr.conv_layer(P, f'conv_layers.{9 * l + k}', cfg, l, node, torch.stack([src, dst]), ea, sh9, out_nodes=N)
        got = ctx.conv_forward(9 * l + k, node.to(dev), src.to(dev), dst.to(dev), [0, E, E, E, E], ea.to(dev), sh9[:, :4].contiguous().to(dev), dout)
        e = (got.cpu() - want).abs()
        print(N, E, sort, 'rel err', rel_err(got.cpu(), want), 'worst col', int(e.max(0).values.argmax()))
