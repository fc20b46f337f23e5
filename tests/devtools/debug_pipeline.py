"""Stage-by-stage comparison of the HIP pipeline against the oracle on one small synthetic complex (GPU box)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import score_model_ref as smr, sampler_ref as spr
from helpers import batch_of, rel_err
from disco_diffdock_amd import synthetic
from disco_diffdock_amd.runtime import Context, Complex

dev = torch.device('cuda:0')
cfg = smr.ScoreModelConfig(latent_vocab=64)
P = smr.random_state_dict(cfg, seed=7)
n_res, n_lig, B = int(os.environ.get('NRES', 60)), int(os.environ.get('NLIG', 22)), int(os.environ.get('B', 3))
c = synthetic.make_complex(3, n_res=n_res, n_lig=n_lig)
ctx = Context(device=0)
ctx.load_state_dict(P)
cx = Complex(ctx, c, max_batch=B)
d = os.path.join(ROOT, 'disco_diffdock_amd', 'data')
tables = (np.load(os.path.join(d, 'so3_exp_score_norms.npy')), np.load(os.path.join(d, 'torus_score_norm_seed0.npy')))
rng = np.random.default_rng(0)
for t in (1.0, 0.5, 0.05):
    pos = np.stack([c['lig_pos'] + rng.normal(0, 3 * t + 0.3, size=(1, 3)) + rng.normal(0, 0.3, size=c['lig_pos'].shape) for _ in range(B)]).astype(np.float32)
    b = batch_of(c, B, pos)
    spr.set_time(b, t, t, t, B)
    tr_r, rot_r, tor_r, inter = smr.score_model_forward(P, cfg, b, tables[0], tables[1], return_intermediates=True)
    g = inter['graph']
    tr, rot, tor = cx.score_forward(torch.from_numpy(pos).to(dev), t, t, t)
    torch.cuda.synchronize()
    st, src, dst, emb, sh, deg = cx.read_edges(B)
    s1, s2, s3 = g['splits']
    Eo = g['edge_index'].shape[1]
    print(f't={t}: edges ours {st}  oracle ll={s1} lr={s2 - s1} rr={s3 - s2} rl={Eo - s3}')
    ok_counts = (st['E_ll'], st['E_lr'], st['E_rr'], st['E_rl']) == (s1, s2 - s1, s3 - s2, Eo - s3)
    if ok_counts:
        grp_o = np.concatenate([np.full(s1, 0), np.full(s2 - s1, 1), np.full(s3 - s2, 2), np.full(Eo - s3, 3)])
        grp_m = np.concatenate([np.full(st['E_ll'], 0), np.full(st['E_lr'], 1), np.full(st['E_rr'], 2), np.full(st['E_rl'], 3)])
        eo = g['edge_index'].numpy()
        ko = np.lexsort((eo[1], eo[0], grp_o))
        km = np.lexsort((dst, src, grp_m))
        same = np.array_equal(eo[0][ko], src[km]) and np.array_equal(eo[1][ko], dst[km])
        print('   edge sets equal:', same, ' src sorted within groups:', all(np.all(np.diff(src[grp_m == q]) >= 0) for q in range(4)))
        if same:
            print('   edge_emb rel err', rel_err(emb[km], g['edge_emb'].numpy()[ko]), ' sh rel err', rel_err(sh[km], g['edge_sh'].numpy()[ko]))
            dego = np.bincount(eo[0], minlength=len(deg))
            print('   deg equal:', np.array_equal(dego, deg))
    lig, rec = cx.node_features(B, dev)
    print('   lig_node_attr rel err', rel_err(lig.cpu(), inter['lig_node_attr']), ' rec', rel_err(rec.cpu(), inter['rec_node_attr']))
    print('   tr', rel_err(tr.cpu(), tr_r), ' rot', rel_err(rot.cpu(), rot_r), ' tor', rel_err(tor.cpu(), tor_r))
    if not ok_counts:
        print('   tr ours', tr.cpu().numpy()[0], 'ref', tr_r.numpy()[0])

# se3 update vs oracle
posb = torch.from_numpy(pos)
tr_u, rot_u, tor_u = torch.randn(B, 3), 0.4 * torch.randn(B, 3), torch.randn(B * cx.R)
b = batch_of(c, B, pos)
ref = spr.modify_conformer_batch(posb.reshape(-1, 3), b, tr_u, rot_u, tor_u, torch.from_numpy(c['mask_rotate']))
out = cx.se3_update(posb.to(dev), tr_u.to(dev), rot_u.to(dev), tor_u.to(dev))
print('se3_update rel err', rel_err(out.cpu().reshape(-1, 3), ref))
ref = spr.modify_conformer_batch(posb.reshape(-1, 3), b, tr_u, rot_u, None, torch.from_numpy(c['mask_rotate']))
out = cx.se3_update(posb.to(dev), tr_u.to(dev), rot_u.to(dev), None)
print('se3_update rigid-only rel err', rel_err(out.cpu().reshape(-1, 3), ref))
