"""Development aid: confidence model on the device vs the oracle, stage by stage."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from oracle import confidence_ref as cr, graph_lite
from helpers import complex_from_npz, to_graph, rel_err
from disco_diffdock_amd.runtime import Context, Complex
G = os.path.join(os.path.dirname(__file__), '..', 'golden')
z, c = np.load(os.path.join(G, 'confidence_paper_model.npz')), complex_from_npz(np.load(os.path.join(G, 'complex_confidence.npz')))
cfg = cr.ConfidenceModelConfig()
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg.num_conv_layers = nl
P = cr.random_state_dict(cr.ConfidenceModelConfig(), seed=int(z['seed']))
B = int(z['B'])
b = graph_lite.collate([graph_lite.add_atoms(to_graph(c), c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index']) for _ in range(B)])
b['ligand'].pos = torch.as_tensor(z['pos']).float()
for nt in ('ligand', 'receptor', 'atom'):
    b[nt].node_t = {k: torch.zeros(b[nt].num_nodes) for k in ('tr', 'rot', 'tor')}
b.complex_t = {k: torch.zeros(B) for k in ('tr', 'rot', 'tor')}
stop = int(os.environ.get('DDK_CONF_MAX_LAYERS', '1000'))
mask = int(os.environ.get('DDK_CONF_GROUP_MASK', str(0x1ff)))
conf, inter = cr.confidence_forward(P, cfg, b, return_intermediates=True, stop_after=stop, group_mask=mask, mask_from=int(os.environ.get('DDK_CONF_MASK_FROM_LAYER', '0')))
dev = torch.device('cuda:0')
ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2, num_conv_layers=nl)
ctx.load_state_dict(P)
cx = Complex(ctx, c, max_batch=B)
cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
out = cx.confidence_forward(torch.as_tensor(z['pos']).float().to(dev))
lig = cx.lig_node_features(B, dev).cpu().numpy()
print('layers', nl, 'oracle counts', inter['counts'], 'device', cx.confidence_counts())
want = inter['lig_node_attr'].numpy()
err = np.abs(lig[:, :want.shape[1]] - want)
print('lig rel err', rel_err(lig[:, :want.shape[1]], want), 'col blocks max err: 0e', err[:, :24].max(), '1o', err[:, 24:42].max(), '1e', err[:, 42:60].max() if want.shape[1] > 42 else None,
      '0o', err[:, 60:].max() if want.shape[1] > 60 else None)
print('worst rows', np.argsort(-err.max(1))[:5], np.sort(-err.max(1))[:5])
print('conf', out.cpu().numpy().ravel(), conf.numpy().ravel())
# ---- edge-level comparison
n_lig, n_atom, n_rec = len(c['lig_x']), len(c['atom_x']), len(c['rec_pos'])
ed = cx.confidence_edges()
base = dict(l=0, a=B * n_lig, r=B * (n_lig + n_atom))
for name, (ei, attr, sh) in inter['edge_sets'].items():
    s_t, d_t = name[0], name[1]
    key_o = {}
    for k in range(ei.shape[1]):
        key_o.setdefault((int(ei[0, k]) + base[s_t], int(ei[1, k]) + base[d_t]), []).append(k)
    src, dst, emb, shd = ed[name]
    miss, e_emb, e_sh = 0, 0.0, 0.0
    for k in range(len(src)):
        js = key_o.get((int(src[k]), int(dst[k])))
        if not js:
            miss += 1; continue
        e_emb = max(e_emb, min(float(np.abs(emb[k] - attr[j].numpy()).max()) for j in js))
        e_sh = max(e_sh, min(float(np.abs(shd[k] - sh[j, :4].numpy()).max()) for j in js))
    print(name, 'n', len(src), ei.shape[1], 'unmatched', miss, 'emb err', e_emb, 'sh err', e_sh, 'emb scale', float(attr.abs().max()))

# ---- node embeddings at layer 0 and degree check through a 4-layer... (x0)
print('x0 lig scale', float(inter['x0']['lig'].abs().max()))
x, deg = cx.confidence_nodes()
a0, r0 = B * n_lig, B * (n_lig + n_atom)
for name, sl, want in (('atom', slice(a0, r0), inter['atom_node_attr'].numpy()), ('rec', slice(r0, None), inter['rec_node_attr'].numpy())):
    got = x[sl][:, :want.shape[1]]
    e = np.abs(got - want)
    print(name, 'rel err', rel_err(got, want), 'worst rows', np.argsort(-e.max(1))[:4], 'col of worst', np.argmax(e.max(0)))
# degrees
for name, ei in (('ll', inter['edge_sets']['ll'][0]), ('lr', inter['edge_sets']['lr'][0]), ('la', inter['edge_sets']['la'][0])):
    d_o = np.bincount(ei[0].numpy(), minlength=B * n_lig)
    print('deg', name, 'match', np.array_equal(d_o, deg[:B * n_lig, ('ll', 'lr', 'la').index(name)]))
aa, ar, rr = (inter['edge_sets'][k][0] for k in ('aa', 'ar', 'rr'))
la, lr = inter['edge_sets']['la'][0], inter['edge_sets']['lr'][0]
print('deg aa', np.array_equal(np.bincount(aa[0].numpy(), minlength=B * n_atom), deg[a0:r0, 0]), 'al', np.array_equal(np.bincount(la[1].numpy(), minlength=B * n_atom), deg[a0:r0, 1]),
      'ar', np.array_equal(np.bincount(ar[0].numpy(), minlength=B * n_atom), deg[a0:r0, 2]))
print('deg rr', np.array_equal(np.bincount(rr[0].numpy(), minlength=B * n_rec), deg[r0:, 0]), 'rl', np.array_equal(np.bincount(lr[1].numpy(), minlength=B * n_rec), deg[r0:, 1]),
      'ra', np.array_equal(np.bincount(ar[1].numpy(), minlength=B * n_rec), deg[r0:, 2]))
