"""development aid: ddk_config.conv_kernel 0 (k_conv_x.hip) against 2 (tools/variants/k_conv_y.hip) on the same inputs - run with DDK_LIB pointing at a variant library built by
tools/build_variant_y.sh (the product library refuses conv_kernel = 2 since round 6)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', 'tests')))
from oracle import score_model_ref as smr
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex
CFG = smr.ScoreModelConfig(latent_vocab=64)
dev = torch.device('cuda:0')
c = synthetic.make_complex(3, n_res=300)
P = smr.random_state_dict(CFG, seed=9)
for B in (2, 8, 40):
    rng = np.random.default_rng(5)
    pos = torch.from_numpy(np.stack([c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)).to(dev)
    for t in (1.0, 0.3):
        res = {}
        for kernel in (0, 2, 1):
            ctx = Context(device=0, conv_kernel=kernel)
            ctx.load_state_dict(P)
            if os.environ.get('NOPRUNE'):
                ctx.set_pruning(False)
            cx = Complex(ctx, c, B)
            tr, rot, tor = cx.score_forward(pos, t, t, t, keep_receptor_features=True) if False else cx.score_forward(pos, t, t, t)
            lig = cx.lig_node_features(B, dev).cpu()
            res[kernel] = (tr.cpu(), rot.cpu(), tor.cpu(), lig)
            cx.close(); ctx.close()
        def rel(a, b): return float((a - b).abs().max() / b.abs().max())
        print(f'B={B} t={t}: y vs x ' + ' '.join(f'{n} {rel(res[2][i], res[0][i]):.1e}' for i, n in enumerate(('tr', 'rot', 'tor', 'lig'))) +
              '   | fp32 kernel vs x ' + ' '.join(f'{n} {rel(res[1][i], res[0][i]):.1e}' for i, n in enumerate(('tr', 'rot', 'tor', 'lig'))))
        d = (res[2][3] - res[0][3]).abs()
        i = int(d.argmax()); print('   worst lig element', i // d.shape[1], i % d.shape[1], float(d.max()), 'of', float(res[0][3].abs().max()))
