"""One tiny hot-path invocation on cuda:0 checked against the oracle (used by __graft_entry__.smoke)."""
import os
import sys

import torch


def run():
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import score_model_ref as smr          # the checker only
    from .tensor_layers import TensorProductConvLayer
    dev = torch.device('cuda:0')
    cfg = smr.ScoreModelConfig()
    l, N, E = 3, 96, 1500
    i_irr, o_irr = cfg.conv_irreps(l)
    P = smr.random_conv_layer_params(cfg, l, 5, True)
    g = torch.Generator().manual_seed(0)
    node = torch.randn(N, 84, generator=g)
    ei = torch.stack([torch.sort(torch.randint(0, N, (E,), generator=g)).values, torch.randint(0, N, (E,), generator=g)])
    ea, sh = torch.randn(E, 72, generator=g), torch.randn(E, 4, generator=g)
    splits = [0, 200, 700, 1100, E]
    layer = TensorProductConvLayer(i_irr, '1x0e+1x1o', o_irr, 72, hidden_features=72, batch_norm=True, faster=True, edge_groups=4).eval()
    layer.load_state_dict(P, strict=True)
    ea_d = ea.to(dev)
    out = layer(node.to(dev), ei.to(dev), [ea_d[splits[i]:splits[i + 1]] for i in range(4)], sh.to(dev)).cpu()
    ref = smr.tp_conv_layer({'L.' + k: v for k, v in P.items()}, 'L', node, ei, [ea[splits[i]:splits[i + 1]] for i in range(4)], sh,
                            i_irr, '1x0e+1x1o', o_irr, residual=True, batch_norm=True, faster=True, edge_groups=4)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'smoke: fused conv layer vs oracle rel err {err:.2e}')
    assert err < 1e-4
