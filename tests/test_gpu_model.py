"""GPU parity tests of the score-model / sampler hot path through the C ABI (drop-in call surface of the
reference: get_model -> model.score_model(batch), modify_conformer_batch, sampling()).

* golden vectors produced by the reference (tests/golden/make_golden.py): 1e-4 relative on tr/rot/tor (north star)
* oracle comparisons on seeded synthetic complexes
* size-independent properties at the BASELINE.json config-2 size (40 samples, ~300 residues): SE(3) equivariance,
  batch-permutation consistency, rigid-update isometry."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr
from helpers import complex_from_npz, batch_of, rel_err, to_graph

pytestmark = pytest.mark.gpu
T = torch.from_numpy

ARGS_S = Namespace(ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32, distance_embed_dim=32, cross_distance_embed_dim=32,
                   max_radius=5.0, cross_max_distance=80, dynamic_max_cross=True, embedding_scale=1000, embedding_type='sinusoidal',
                   scale_by_sigma=True, no_torsion=False, no_batch_norm=False, dropout=0.1, sh_lmax=1, use_second_order_repr=False,
                   use_old_atom_encoder=False, esm_embeddings_path='data/esm2_3billion_embeddings.pt', latent_dim=0, latent_vocab=64,
                   latent_cross_attention=False, tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55,
                   tor_sigma_min=0.03, tor_sigma_max=3.14)
README_S = dict(temp_sampling=[1.886430780895051, 5.659562317960644, 2.8888668488630156],
                temp_psi=[0.07085125444659945, 2.686505606141324, 4.089493860493927],
                temp_sigma_data=[0.3617563913086843, 0.7437588205919711, 0.08897393057297842])
CFG = smr.ScoreModelConfig(latent_vocab=64)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    from disco_diffdock_amd import build
    build.build(verbose=False)
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def model7(dev):
    """the reference call surface: get_model(args, device, t_to_sigma) + load_state_dict of the score model"""
    from functools import partial
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.diffusion_utils import t_to_sigma
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    model.score_model.load_state_dict(smr.random_state_dict(CFG, seed=7), strict=True)
    return model


def _dev_batch(c, B, pos, dev, t):
    from disco_diffdock_amd.data import from_arrays, collate
    from disco_diffdock_amd.diffusion_utils import set_time
    b = collate([from_arrays(c) for _ in range(B)])
    b['ligand'].pos = torch.as_tensor(pos).float().reshape(-1, 3)
    b = b.to(dev)
    set_time(b, t, t, t, B, False, dev)
    return b


@pytest.mark.parametrize('t', [1.0, 0.55, 0.05])
def test_score_model_golden(dev, model7, golden, t):
    tag = 'diffdockS_score_model'
    z = golden(f'score_{tag}_t{t}')
    c = complex_from_npz(golden(f'complex_{tag}'))
    B = int(z['B'])
    b = _dev_batch(c, B, z['pos'], dev, t)
    tr, rot, tor = model7.score_model(b, keep_receptor_features=True)
    cx = model7.score_model.last_complex
    lig, rec = cx.node_features(B, dev)
    assert rel_err(lig.cpu(), z['lig_node_attr']) < 1e-4
    assert rel_err(rec.cpu(), z['rec_node_attr']) < 1e-4
    for name, a in (('tr', tr), ('rot', rot), ('tor', tor)):
        assert rel_err(a.cpu(), z[name]) < 1e-4, name


def test_conformer_update_golden(dev, golden):
    from disco_diffdock_amd.diffusion_utils import modify_conformer_batch
    z = golden('conformer_update')
    c = complex_from_npz(golden('toy_complex'))
    B = int(z['B'])
    b = _dev_batch(c, B, z['pos'], dev, 0.5)
    mr = T(z['mask_rotate'])
    new = modify_conformer_batch(T(z['pos']).to(dev), b, T(z['tr']).to(dev), T(z['rot']).to(dev), T(z['tor']).to(dev), mr)
    assert rel_err(new.cpu(), z['new_pos']) < 2e-5
    rigid = modify_conformer_batch(T(z['pos']).to(dev), b, T(z['tr']).to(dev), T(z['rot']).to(dev), None, mr)
    assert rel_err(rigid.cpu(), z['rigid_only']) < 1e-5


@pytest.mark.parametrize('tag,kw', [('plain', {}), ('lowtemp', README_S), ('ode', dict(ode=True))])
def test_sde_steps_golden(dev, golden, tag, kw):
    """reference sampling() arithmetic with fixed scores: host step coefficients + GPU conformer update,
    noise from the reference's torch.manual_seed CPU stream (draw order tr, rot, tor per step)."""
    from functools import partial
    from disco_diffdock_amd.sampling import step_coefficients
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule, modify_conformer_batch
    z = golden(f'sde_steps_{tag}')
    c = complex_from_npz(golden('toy_complex'))
    B, steps = z['tr'].shape[0], int(z['steps'])
    sched = get_t_schedule(steps)
    kw = dict(kw)
    ode = kw.pop('ode', False)
    t_arr, sc, nc = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=ARGS_S), ARGS_S, ode, False, True,
                                      kw.get('temp_sampling', 1.0), kw.get('temp_psi', 0.0), kw.get('temp_sigma_data', 0.5))
    torch.manual_seed(int(z['seed']))
    pos = T(z['pos0']).to(dev)
    b = _dev_batch(c, B, z['pos0'], dev, 1.0)
    R = int(c['edge_mask'].sum())
    for k in range(steps):
        last = k == steps - 1
        zt = torch.zeros(B, 3) if (last or ode) else torch.normal(mean=0, std=1, size=(B, 3))
        zr = torch.zeros(B, 3) if (last or ode) else torch.normal(mean=0, std=1, size=(B, 3))
        zo = torch.zeros(B * R) if (last or ode) else torch.normal(mean=0, std=1, size=(B * R,))
        tr = sc[k, 0] * T(z['tr']) + nc[k, 0] * zt
        rot = sc[k, 1] * T(z['rot']) + nc[k, 1] * zr
        tor = sc[k, 2] * T(z['tor']) + nc[k, 2] * zo
        pos = modify_conformer_batch(pos, b, tr.to(dev), rot.to(dev), tor.to(dev), T(c['mask_rotate']))
    assert rel_err(pos.cpu(), z['pos_out']) < 5e-5


def _ref_noise(seed, steps, B, R, no_final_step_noise=True):
    torch.manual_seed(seed)
    z = torch.zeros(steps, B, 6 + R)
    for k in range(steps):
        if no_final_step_noise and k == steps - 1:
            continue
        z[k, :, 0:3] = torch.normal(mean=0, std=1, size=(B, 3))
        z[k, :, 3:6] = torch.normal(mean=0, std=1, size=(B, 3))
        z[k, :, 6:] = torch.normal(mean=0, std=1, size=(B * R,)).reshape(B, R)
    return z


def test_trajectory_golden(dev, model7, golden):
    """3-step sampling() trajectory of the reference (its own sampler + score model on the stand-ins)."""
    from functools import partial
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    tag = 'diffdockS_score_model'
    z = golden(f'trajectory_{tag}')
    c = complex_from_npz(golden(f'complex_{tag}'))
    B, steps, n = 2, int(z['steps']), len(c['lig_pos'])
    dl = [from_arrays(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
    sched = get_t_schedule(steps)
    noise = [_ref_noise(int(z['seed']), steps, B, int(c['edge_mask'].sum()))]
    out, conf = sampling(dl, model7, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B,
                         no_final_step_noise=True, use_latent=False, noise=noise, **README_S)
    assert conf is None
    pos = torch.cat([d['ligand'].pos for d in out]).cpu()
    assert rel_err(pos, z['pos_out']) < 1e-4


def test_sampling_vs_oracle(dev, tables):
    """5 reverse steps, 4 samples, README DiffDock-S temperatures, against the CPU oracle with the same noise."""
    from functools import partial
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from helpers import to_graph
    c = synthetic.make_complex(21, n_res=50, n_lig=24)
    P = smr.random_state_dict(CFG, seed=11)
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    model.score_model.load_state_dict(P)
    B, steps, R, n = 4, 5, int(c['edge_mask'].sum()), len(c['lig_pos'])
    rng = np.random.default_rng(1)
    start = [c['lig_pos'] + rng.normal(0, 6.0, size=(1, 3)).astype(np.float32) for _ in range(B)]
    z = _ref_noise(77, steps, B, R)
    sched = get_t_schedule(steps)
    dl = [from_arrays(c) for _ in range(B)]
    for d, p in zip(dl, start):
        d['ligand'].pos = T(p).float()
    out, _ = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B,
                      no_final_step_noise=True, noise=[z], **README_S)
    ol = [to_graph(c) for _ in range(B)]
    for d, p in zip(ol, start):
        d['ligand'].pos = T(p).float()
    nf = lambda b, t, name, shape: {'tr': z[t, :, 0:3], 'rot': z[t, :, 3:6], 'tor': z[t, :, 6:].reshape(-1)}[name]
    ref, _ = spr.sampling(ol, P, CFG, tables[0], tables[1], steps, sched, sched, sched, noise_fn=nf, batch_size=B,
                          no_final_step_noise=True, **README_S)
    a = torch.cat([d['ligand'].pos for d in out]).cpu()
    r = torch.cat([d['ligand'].pos for d in ref])
    assert rel_err(a, r) < 1e-4
    # the README DiffDock-S command samples 40 poses in batches of 10 (--batch_size 10): batches are independent, so two batches of 2
    # with the noise split the same way reproduce the single batch of 4
    dl2 = [from_arrays(c) for _ in range(B)]
    for d, p in zip(dl2, start):
        d['ligand'].pos = T(p).float()
    out2, _ = sampling(dl2, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=2,
                       no_final_step_noise=True, noise=[z[:, :2].contiguous(), z[:, 2:].contiguous()], **README_S)
    a2 = torch.cat([d['ligand'].pos for d in out2]).cpu()
    assert rel_err(a2, a) < 1e-5


@pytest.mark.parametrize('n_res,t,min_edges', [(300, 0.4, 500000), (2000, 0.4, 2000000), (2000, 1.0, 5000000)])
def test_equivariance_and_batch_consistency_full_size(dev, n_res, t, min_edges):
    """BASELINE config-2 shape (40 samples, 300 residues) and config-5 shape (2000 residues; at t = 1 every ligand-residue pair is a
    cross edge): rotating + translating the whole complex rotates tr/rot and leaves tor unchanged; permuting the samples of the batch
    permutes the outputs."""
    from scipy.spatial.transform import Rotation
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    c = synthetic.make_complex(2, n_res=n_res)
    P = smr.random_state_dict(CFG, seed=3)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    B = 40
    rng = np.random.default_rng(0)
    pos = np.stack([c['lig_pos'] + rng.normal(0, 4.0, size=(1, 3)) + rng.normal(0, 0.3, size=c['lig_pos'].shape) for _ in range(B)]).astype(np.float32)
    cx = Complex(ctx, c, B)
    tr0, rot0, tor0 = [x.double().cpu() for x in cx.score_forward(T(pos).to(dev), t, t, t)]
    st = cx.graph_stats()
    assert st['E_rr'] == B * c['rec_edge_index'].shape[1] and st['E_lr'] == st['E_rl'] and st['E'] > min_edges
    if t == 1.0:       # cutoff 3 sigma_tr + 20 = 77 A: every ligand atom sees every residue of the 40 A ball
        assert st['E_lr'] == B * len(c['lig_pos']) * n_res
    perm = rng.permutation(B)
    tr1, rot1, tor1 = [x.double().cpu() for x in cx.score_forward(T(pos[perm]).to(dev), t, t, t)]
    R = cx.R
    assert rel_err(tr1, tr0[perm]) < 1e-5 and rel_err(rot1, rot0[perm]) < 1e-5
    assert rel_err(tor1.reshape(B, R), tor0.reshape(B, R)[perm]) < 1e-5
    Rm = Rotation.random(random_state=4).as_matrix()
    shift = np.array([[2.0, -3.0, 1.5]])
    c2 = dict(c)
    c2['rec_pos'] = (c['rec_pos'].astype(np.float64) @ Rm.T + shift).astype(np.float32)
    pos2 = (pos.astype(np.float64) @ Rm.T + shift).astype(np.float32)
    cx2 = Complex(ctx, c2, B)
    tr2, rot2, tor2 = [x.double().cpu() for x in cx2.score_forward(T(pos2).to(dev), t, t, t)]
    Rt = torch.from_numpy(Rm)
    assert rel_err(tr0 @ Rt.T, tr2) < 2e-4
    assert rel_err(rot0 @ Rt.T, rot2) < 2e-4
    assert rel_err(tor0, tor2) < 2e-4
    # rigid update is an isometry: pairwise distances preserved
    out = cx.se3_update(T(pos).to(dev), torch.randn(B, 3, device=dev), torch.randn(B, 3, device=dev), None).cpu()
    pd = lambda x: (x[:, :, None, :] - x[:, None, :, :]).norm(dim=-1)
    assert float((pd(T(pos)) - pd(out)).abs().max()) < 2e-4


def test_rigid_ligand_and_single_sample(dev, tables):
    """ligand without rotatable bonds (tor is empty, reference returns torch.empty(0)) and B = 1"""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    c = synthetic.make_complex(8, n_res=40, n_lig=20)
    c['edge_mask'] = np.zeros_like(c['edge_mask'])
    c['mask_rotate'] = np.zeros((0, len(c['lig_pos'])), dtype=bool)
    P = smr.random_state_dict(CFG, seed=5)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, 1)
    pos = c['lig_pos'][None] + 1.0
    tr, rot, tor = cx.score_forward(T(pos).to(dev), 0.3, 0.3, 0.3)
    assert tor.numel() == 0
    b = batch_of(c, 1, pos)
    spr.set_time(b, 0.3, 0.3, 0.3, 1)
    tr_r, rot_r, tor_r = smr.score_model_forward(P, CFG, b, tables[0], tables[1])
    assert rel_err(tr.cpu(), tr_r) < 1e-4 and rel_err(rot.cpu(), rot_r) < 1e-4 and tor_r.numel() == 0
    out = cx.se3_update(T(pos).to(dev), tr, rot, None)
    ref = spr.modify_conformer_batch(T(pos).reshape(-1, 3), b, tr.cpu(), rot.cpu(), None, T(c['mask_rotate']))
    assert rel_err(out.cpu().reshape(-1, 3), ref) < 1e-5


def test_unsupported_options_are_loud(dev, model7):
    from functools import partial
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.diffusion_utils import t_to_sigma
    with pytest.raises(RuntimeError, match='confidence'):
        sampling([], model7, 2, [1, .5], [1, .5], [1, .5], dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, confidence_model=object())
    with pytest.raises(RuntimeError, match='GPU only'):
        sampling([], model7, 2, [1, .5], [1, .5], [1, .5], 'cpu', partial(t_to_sigma, args=ARGS_S), ARGS_S)


def test_disco_latent_score_model_golden(dev, golden):
    """DisCo-DiffDock-S score model (latent_dim=2, latent_vocab=1, latent_droprate=0.1) with one-hot node latents."""
    from functools import partial
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.diffusion_utils import t_to_sigma
    tag = 'disco_diffdockS_score_model'
    args = Namespace(**dict(vars(ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
    cfg = smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
    model = get_model(args, dev, partial(t_to_sigma, args=args), no_parallel=True)
    model.score_model.load_state_dict(smr.random_state_dict(cfg, seed=7), strict=True)
    c = complex_from_npz(golden(f'complex_{tag}'))
    for t in (1.0, 0.55, 0.05):
        z = golden(f'score_{tag}_t{t}')
        B = int(z['B'])
        b = _dev_batch(c, B, z['pos'], dev, t)
        b['ligand'].latent_h, b['receptor'].latent_h = T(z['latent_l']).to(dev), T(z['latent_r']).to(dev)
        b['ligand'].unconditional = torch.zeros(b['ligand'].num_nodes, 1, device=dev)
        b['receptor'].unconditional = torch.zeros(b['receptor'].num_nodes, 1, device=dev)
        tr, rot, tor = model.score_model(b, keep_receptor_features=True)
        lig, rec = model.score_model.last_complex.node_features(B, dev)
        assert rel_err(lig.cpu(), z['lig_node_attr']) < 1e-4 and rel_err(rec.cpu(), z['rec_node_attr']) < 1e-4
        for name, a in (('tr', tr), ('rot', rot), ('tor', tor)):
            assert rel_err(a.cpu(), z[name]) < 1e-4, name
    # unconditional = 1 and zeroed latents (the classifier-free-guidance / AR-encoder input) against the oracle
    from oracle import sampler_ref
    from helpers import batch_of
    B = 2
    pos = golden(f'score_{tag}_t0.55')['pos']
    b = _dev_batch(c, B, pos, dev, 1.0)
    b['ligand'].latent_h = torch.zeros(b['ligand'].num_nodes, 2, device=dev)
    b['receptor'].latent_h = torch.zeros(b['receptor'].num_nodes, 2, device=dev)
    b['ligand'].unconditional = torch.ones(b['ligand'].num_nodes, 1, device=dev)
    b['receptor'].unconditional = torch.ones(b['receptor'].num_nodes, 1, device=dev)
    tr, rot, tor = model.score_model(b)
    ob = batch_of(c, B, pos)
    sampler_ref.set_time(ob, 1.0, 1.0, 1.0, B)
    ob['ligand'].latent_h, ob['receptor'].latent_h = torch.zeros(ob['ligand'].num_nodes, 2), torch.zeros(ob['receptor'].num_nodes, 2)
    ob['ligand'].unconditional, ob['receptor'].unconditional = torch.ones(ob['ligand'].num_nodes, 1), torch.ones(ob['receptor'].num_nodes, 1)
    import numpy as _np, os as _os
    d = _os.path.join(_os.path.dirname(__file__), '..', 'disco_diffdock_amd', 'data')
    tab = (_np.load(_os.path.join(d, 'so3_exp_score_norms.npy')), _np.load(_os.path.join(d, 'torus_score_norm_seed0.npy')))
    tr_r, rot_r, tor_r = smr.score_model_forward(smr.random_state_dict(cfg, seed=7), cfg, ob, tab[0], tab[1])
    assert rel_err(tr.cpu(), tr_r) < 1e-4 and rel_err(rot.cpu(), rot_r) < 1e-4 and rel_err(tor.cpu(), tor_r) < 1e-4


def test_ar_latent_model_golden(dev, golden):
    """AR latent model (a22): logits of PretrainedScoreEncoder and the argmax-decoded latents of encode_ar vs the reference."""
    from disco_diffdock_amd.model_utils import get_ar_model
    from oracle import ar_ref
    tag = 'disco_diffdockS_score_model'
    z = golden(f'ar_{tag}')
    score_args = Namespace(**dict(vars(ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
    ar_args = Namespace(use_pretrained_score=True, ns=16, latent_no_batchnorm=False, latent_dropout=0.0, latent_hidden_dim=128,
                        esm_embeddings_path='x', original_model_dir='unused', ckpt='unused')
    cfg = smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
    ar = get_ar_model(ar_args, score_args, dev, training=False)
    ar.load_state_dict(ar_ref.random_ar_state_dict(cfg, ar_ns=16, hidden=128, seed=int(z['seed'])), strict=True)
    ar.eval()
    c = complex_from_npz(golden(f'complex_{tag}'))
    B = int(z['B'])
    b = _dev_batch(c, B, z['pos'], dev, 1.0)
    b['ligand'].input_latent = torch.zeros(b['ligand'].num_nodes, 2, device=dev)
    b['receptor'].input_latent = torch.zeros(b['receptor'].num_nodes, 2, device=dev)
    with torch.no_grad():
        logits = ar.logits(b)
        assert rel_err(logits.cpu(), z['logits0']) < 1e-4
        assert 'latent_h' not in b['ligand']          # the input batch is left untouched
        lat_l, lat_r = ar.encode_ar(b, 100.0)
    assert torch.equal(lat_l.cpu(), T(z['latent_l'])) and torch.equal(lat_r.cpu(), T(z['latent_r']))


def test_disco_sampling_with_ar_model_vs_oracle(dev, tables):
    """config-3 path end to end on a small complex: AR decoding (injected choices) -> latent-conditioned 3-step sampling."""
    from functools import partial
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.model_utils import get_model, get_ar_model
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from oracle import ar_ref
    from helpers import to_graph
    score_args = Namespace(**dict(vars(ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
    ar_args = Namespace(use_pretrained_score=True, ns=16, latent_no_batchnorm=False, latent_dropout=0.0, latent_hidden_dim=128,
                        esm_embeddings_path='x', no_randomness=False)
    cfg = smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
    P, P_ar = smr.random_state_dict(cfg, seed=13), ar_ref.random_ar_state_dict(cfg, seed=14)
    model = get_model(score_args, dev, partial(t_to_sigma, args=score_args), no_parallel=True)
    model.score_model.load_state_dict(P)
    ar = get_ar_model(ar_args, score_args, dev, training=False)
    ar.load_state_dict(P_ar)
    ar.eval()
    c = synthetic.make_complex(31, n_res=40, n_lig=22)
    B, steps, R = 3, 3, int(c['edge_mask'].sum())
    rng = np.random.default_rng(2)
    start = [c['lig_pos'] + rng.normal(0, 4.0, size=(1, 3)).astype(np.float32) for _ in range(B)]
    z = _ref_noise(5, steps, B, R)
    sched = get_t_schedule(steps)
    dl = [from_arrays(c) for _ in range(B)]
    for d, p in zip(dl, start):
        d['ligand'].pos = T(p).float()
        d['ligand'].ar_pos = T(p).float()
    CFG = dict(classifier_free_guidance_weight=0.7, cfg_start=0.9, cfg_end=0.2)     # active on the middle step only (t = 1, 2/3, 1/3)
    out, _ = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=score_args), score_args, batch_size=B,
                      no_final_step_noise=True, ar_model=ar, ar_args=ar_args, softmax_latent_temperature=100.0, noise=[z], **README_S, **CFG)
    # oracle: same AR decoding (argmax) and latent-conditioned sampling
    ol = [to_graph(c) for _ in range(B)]
    for d, p in zip(ol, start):
        d['ligand'].pos = T(p).float()
    from oracle import graph_lite
    ob = graph_lite.collate(ol)
    lat_l, lat_r = ar_ref.encode_ar(P_ar, cfg, 16, ob, sampling_temperature=100.0)
    n_l, n_r = len(c['lig_pos']), len(c['rec_pos'])
    for i, d in enumerate(ol):
        d['ligand'].latent_h, d['receptor'].latent_h = lat_l[i * n_l:(i + 1) * n_l], lat_r[i * n_r:(i + 1) * n_r]
        d['ligand'].unconditional, d['receptor'].unconditional = torch.zeros(n_l, 1), torch.zeros(n_r, 1)
    nf = lambda b, t, name, shape: {'tr': z[t, :, 0:3], 'rot': z[t, :, 3:6], 'tor': z[t, :, 6:].reshape(-1)}[name]
    ref, _ = spr.sampling(ol, P, cfg, tables[0], tables[1], steps, sched, sched, sched, noise_fn=nf, batch_size=B,
                          no_final_step_noise=True, **README_S, **CFG)
    a = torch.cat([d['ligand'].pos for d in out]).cpu()
    r = torch.cat([d['ligand'].pos for d in ref])
    assert rel_err(a, r) < 1e-4
    assert all(hasattr(d, 'latent_str') and len(d.latent_pos) == 2 for d in out)


def test_disco_trajectory_golden(dev, golden):
    """the reference's sampling() on the DisCo path (AR argmax decoding, latents, CFG, README DisCo temperatures) through ddk."""
    from functools import partial
    from disco_diffdock_amd.model_utils import get_model, get_ar_model
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from oracle import ar_ref
    tag = 'disco_diffdockS_score_model'
    z = golden(f'trajectory_{tag}')
    score_args = Namespace(**dict(vars(ARGS_S), latent_dim=2, latent_vocab=1, latent_droprate=0.1))
    ar_args = Namespace(use_pretrained_score=True, ns=16, latent_no_batchnorm=False, latent_dropout=0.0, latent_hidden_dim=128,
                        esm_embeddings_path='x', no_randomness=False)
    cfg = smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
    model = get_model(score_args, dev, partial(t_to_sigma, args=score_args), no_parallel=True)
    model.score_model.load_state_dict(smr.random_state_dict(cfg, seed=7))
    ar = get_ar_model(ar_args, score_args, dev, training=False)
    ar.load_state_dict(ar_ref.random_ar_state_dict(cfg, ar_ns=16, hidden=128, seed=int(z['ar_seed'])))
    ar.eval()
    c = complex_from_npz(golden(f'complex_{tag}'))
    B, steps, n = 2, int(z['steps']), len(c['lig_pos'])
    dl = [from_arrays(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
        d['ligand'].ar_pos = d['ligand'].pos.clone()
    sched = get_t_schedule(steps)
    README_D = dict(temp_sampling=[1.546842681537956, 4.005218254154881, 3.6499018519649384],
                    temp_psi=[1.2685697872473618, 1.2760150490206228, 2.0625243924678136],
                    temp_sigma_data=[0.8456140350087653, 0.453446580767075, 0.3292199987743284])
    noise = [_ref_noise(int(z['seed']), steps, B, int(c['edge_mask'].sum()))]
    out, _ = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=score_args), score_args, batch_size=B,
                      no_final_step_noise=True, use_latent=True, ar_model=ar, ar_args=ar_args, softmax_latent_temperature=100.0,
                      classifier_free_guidance_weight=0.7, cfg_start=0.9, cfg_end=0.2, noise=noise, **README_D)
    assert [d.latent_str for d in out] == [str(x) for x in z['latent_str']]
    assert rel_err(torch.cat([d['ligand'].pos for d in out]).cpu(), z['pos_out']) < 1e-4


def test_neighbour_caps_bind_dense_ligand(dev, tables):
    """Edge case of the radius graphs: a compact 70-atom ligand where radius_graph's max_num_neighbors=32 (score_model.py:315)
    and the bond-centre radius cap of 32 (score_model.py:430) both bind; ligand far from the pocket at small t (empty cross
    graph).  Edge sets and scores must agree with the oracle's restatement of the cap semantics (first-k by index)."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    rng = np.random.default_rng(3)
    c = synthetic.make_complex(12, n_res=40, n_lig=40)
    n = 70
    # compact blob: points on a jittered grid of 1.6 A spacing -> ~45 atoms within 5 A of an interior atom
    g = np.stack(np.meshgrid(np.arange(5), np.arange(4), np.arange(4), indexing='ij'), -1).reshape(-1, 3)[:n] * 1.6
    pos = (g + rng.normal(0, 0.1, size=g.shape)).astype(np.float32)
    bonds = [(i, i + 1) for i in range(n - 1)]
    edge_mask, mask_rotate = synthetic.transformation_mask(n, bonds)
    ei = np.zeros((2, 2 * len(bonds)), np.int64)
    for bi, (a, b) in enumerate(bonds):
        ei[:, 2 * bi], ei[:, 2 * bi + 1] = (a, b), (b, a)
    ea = np.zeros((2 * len(bonds), 4), np.float32)
    ea[:, 0] = 1
    c.update(lig_x=np.stack([rng.integers(0, d, size=n) for d in synthetic.LIG_FEATURE_DIMS], 1), lig_pos=pos, bond_index=ei,
             bond_attr=ea, edge_mask=edge_mask, mask_rotate=mask_rotate)
    P = smr.random_state_dict(CFG, seed=2)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, 2)
    for shift, t in ((np.zeros(3), 0.7), (np.array([200.0, 0, 0]), 0.05)):
        p2 = np.stack([pos + shift, pos + shift + 0.3]).astype(np.float32)
        tr, rot, tor = cx.score_forward(T(p2).to(dev), t, t, t)
        st = cx.graph_stats()
        b = batch_of(c, 2, p2)
        spr.set_time(b, t, t, t, 2)
        tr_r, rot_r, tor_r, inter = smr.score_model_forward(P, CFG, b, tables[0], tables[1], return_intermediates=True)
        s1, s2, s3 = inter['graph']['splits']
        assert (st['E_ll'], st['E_lr'], st['E_rr']) == (s1, s2 - s1, s3 - s2)
        assert st['E_ll'] < 2 * (2 * len(bonds) + n * 32) + 1 and st['E_ll'] > 2 * (2 * len(bonds) + n * 20)   # the cap binds
        if shift[0] > 0:
            assert st['E_lr'] == 0
        for name, a, r in (('tr', tr, tr_r), ('rot', rot, rot_r), ('tor', tor, tor_r)):
            assert rel_err(a.cpu(), r) < 1e-4, name


def test_size_limits_are_loud(dev):
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    ctx = Context(device=0)
    ctx.load_state_dict(smr.random_state_dict(CFG, seed=2))
    c = synthetic.make_complex(1, n_res=20, n_lig=20)
    big = dict(c)
    n = 257
    big.update(lig_x=np.zeros((n, 16), np.int64), lig_pos=np.zeros((n, 3), np.float32), mask_rotate=np.zeros((0, n), bool),
               edge_mask=np.zeros(c['bond_index'].shape[1], bool))
    with pytest.raises(RuntimeError, match='n_lig'):
        Complex(ctx, big, 1)
    cx = Complex(ctx, c, 2)
    with pytest.raises(RuntimeError, match='max_batch'):
        cx.score_forward(torch.zeros(3, 20, 3, device=dev), 0.5, 0.5, 0.5)
    bad = dict(c)
    bad['rec_x'] = c['rec_x'][:, :100]
    with pytest.raises(RuntimeError, match='feature width'):
        Complex(ctx, bad, 1)


def test_last_layer_receptor_rows_on_request(dev, model7, golden):
    """The last conv layer skips the messages into receptor nodes unless asked (ddk_set_keep_receptor_features);
    scores are the same either way and reading the missing rows fails loudly."""
    tag = 'diffdockS_score_model'
    z = golden(f'score_{tag}_t0.55')
    c = complex_from_npz(golden(f'complex_{tag}'))
    B = int(z['B'])
    b = _dev_batch(c, B, z['pos'], dev, 0.55)
    out_lean = [a.clone() for a in model7.score_model(b)]
    cx = model7.score_model.last_complex
    with pytest.raises(RuntimeError, match='receptor rows'):
        cx.node_features(B, dev)
    out_full = model7.score_model(b, keep_receptor_features=True)
    lig, rec = cx.node_features(B, dev)
    assert rel_err(rec.cpu(), z['rec_node_attr']) < 1e-4
    for a, f_, name in zip(out_lean, out_full, ('tr', 'rot', 'tor')):
        assert rel_err(a.cpu(), f_.cpu().numpy()) < 5e-6, name          # (two evaluations with different atomic-add orders: ~1e-6 noise)
        assert rel_err(a.cpu(), z[name]) < 1e-4, name


@pytest.mark.parametrize('no_torsion,no_random', [(False, False), (True, False), (False, True)])
def test_randomize_position_device(dev, golden, no_torsion, no_random):
    """ddk_randomize_position == the oracle's randomize_position (utils/sampling.py:12-34) on the same draws."""
    from scipy.spatial.transform import Rotation as R
    from disco_diffdock_amd.runtime import Context, Complex
    c = complex_from_npz(golden('complex_diffdockS_score_model'))
    B, sig = 6, 19.0
    gl = [to_graph(c) for _ in range(B)]
    spr.randomize_position(gl, no_torsion, no_random, sig, rng=np.random.default_rng(5))
    want = np.stack([g['ligand'].pos.numpy() for g in gl])
    rng = np.random.default_rng(5)                       # replay the oracle's draw order
    n_rot = int(np.asarray(c['edge_mask']).sum())
    tor = None if no_torsion else np.stack([rng.uniform(low=-np.pi, high=np.pi, size=n_rot) for _ in range(B)]).astype(np.float32)
    rot, tr = np.empty((B, 3, 3), np.float32), np.empty((B, 3), np.float32)
    for b in range(B):
        rot[b] = R.random(random_state=rng).as_matrix()
        if not no_random:
            tr[b] = rng.normal(0, sig, size=(1, 3))[0]
    from disco_diffdock_amd.tensor_layers import _shape_context
    cx = Complex(_shape_context(0), c, max_batch=B)
    got = cx.randomize_position(T(np.asarray(c['lig_pos'], np.float32)).to(dev), T(rot).to(dev),
                                None if tor is None else T(tor).to(dev), None if no_random else T(tr).to(dev))
    assert rel_err(got.cpu(), want) < 2e-6


def test_randomize_position_device_wrapper(dev, golden):
    """The list-level wrapper draws from the reference's host RNG streams and writes device views back into the graphs."""
    from disco_diffdock_amd.sampling import randomize_position_device
    from disco_diffdock_amd.data import from_arrays
    c = complex_from_npz(golden('complex_diffdockS_score_model'))
    gl = [from_arrays(c) for _ in range(4)]
    np.random.seed(0); torch.manual_seed(0)
    pos = randomize_position_device(gl, False, False, 19.0, dev)
    assert pos.shape == (4, c['lig_pos'].shape[0], 3) and all(g['ligand'].pos.is_cuda for g in gl)
    d0 = torch.cdist(T(np.asarray(c['lig_pos'], np.float32)), T(np.asarray(c['lig_pos'], np.float32)))
    bonds = np.asarray(c['bond_index'])
    for g in gl:                                          # torsions + rigid motion keep every bond length
        p = g['ligand'].pos.cpu()
        assert torch.allclose(torch.linalg.norm(p[bonds[0]] - p[bonds[1]], dim=1), d0[bonds[0], bonds[1]], atol=1e-4)
    with pytest.raises(RuntimeError, match='cuda'):
        randomize_position_device(gl, False, False, 19.0, torch.device('cpu'))


@pytest.mark.parametrize('kernel', [0, 1, 3])
def test_confidence_model_golden(dev, golden, kernel):
    """ddk_confidence_forward == the reference's all-atom confidence model (golden produced by
    models/all_atom_score_model.py through get_model on the stand-ins) on the same poses; ligand features after the conv stack too.
    kernel 0: the default two-limb / three-product f16 form (k_conv_x2.hip, l = 2 row groups included), 1: the fp32-MFMA fallback, 3: the three-limb / six-product form (k_conv_x.hip)."""
    from oracle import confidence_ref as cr
    from disco_diffdock_amd.runtime import Context, Complex
    z, c = golden('confidence_paper_model'), complex_from_npz(golden('complex_confidence'))
    cfg = cr.ConfidenceModelConfig()
    ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2, conv_kernel=kernel)
    ctx.load_state_dict(cr.random_state_dict(cfg, seed=int(z['seed'])))
    B = int(z['B'])
    cx = Complex(ctx, c, max_batch=B)
    cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
    conf = cx.confidence_forward(T(z['pos']).to(dev))
    lig = cx.lig_node_features(B, dev)
    assert rel_err(lig.cpu(), z['lig_node_attr']) < 1e-4
    assert rel_err(conf.cpu(), z['confidence']) < 1e-4


CONF_ARGS = Namespace(all_atoms=True, ns=24, nv=6, num_conv_layers=5, sigma_embed_dim=32, distance_embed_dim=32, cross_distance_embed_dim=32,
                      max_radius=5.0, cross_max_distance=80, dynamic_max_cross=True, embedding_type='sinusoidal', embedding_scale=10000,
                      scale_by_sigma=True, no_torsion=False, no_batch_norm=False, dropout=0.1, use_second_order_repr=False,
                      esm_embeddings_path='data/esm2_3billion_embeddings.pt', rmsd_classification_cutoff=[2.0], confidence_no_batchnorm=False,
                      tr_sigma_min=0.1, tr_sigma_max=34.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.0314, tor_sigma_max=3.14)


def test_sampling_with_confidence_golden(dev, model7, golden):
    """utils/sampling.py:59-62,230-249: the reference's sampling(confidence_model=..., confidence_data_list=...) - DiffDock-S reverse
    diffusion followed by the all-atom confidence model on the final poses - reproduced through get_model(confidence_mode=True)."""
    from functools import partial
    from oracle import confidence_ref as cr
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    z, c = golden('trajectory_confidence'), complex_from_npz(golden('complex_confidence'))
    cm = get_model(CONF_ARGS, dev, partial(t_to_sigma, args=CONF_ARGS), no_parallel=True, confidence_mode=True)
    cm.load_state_dict(cr.random_state_dict(cr.ConfidenceModelConfig(), seed=int(z['conf_seed'])), strict=True)
    cm.eval()
    n = len(c['lig_pos'])
    B, steps = len(z['pos0']) // n, int(z['steps'])
    score_only = {k: v for k, v in c.items() if not k.startswith('atom_')}
    dl, cdl = [from_arrays(score_only) for _ in range(B)], [from_arrays(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
    sched = get_t_schedule(steps)
    noise = [_ref_noise(int(z['seed']), steps, B, int(c['edge_mask'].sum()))]
    out, conf = sampling(dl, model7, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B,
                         no_final_step_noise=True, use_latent=False, noise=noise, confidence_model=cm, confidence_data_list=cdl,
                         confidence_model_args=CONF_ARGS, temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5)
    assert rel_err(torch.cat([d['ligand'].pos for d in out]).cpu(), z['pos_out']) < 1e-4
    assert tuple(conf.shape) == z['confidence'].shape and rel_err(conf.cpu(), z['confidence']) < 1e-4
    with pytest.raises(RuntimeError, match='confidence_data_list'):
        sampling(dl, model7, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B, confidence_model=cm)


@pytest.mark.parametrize('t', [1.0, 0.05])
def test_score_model_golden_fp32_kernel(dev, golden, t):
    """The fallback fp32-MFMA conv kernel (ddk_config.conv_kernel = 1) against the reference-produced score goldens, same 1e-4 bar."""
    from functools import partial
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.diffusion_utils import t_to_sigma
    tag = 'diffdockS_score_model'
    z = golden(f'score_{tag}_t{t}')
    c = complex_from_npz(golden(f'complex_{tag}'))
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    sm = model.score_model
    sm.ctx.close()
    from disco_diffdock_amd.runtime import Context
    sm.cfg['conv_kernel'] = 1
    sm.ctx = Context(device=0, **sm.cfg)
    sm.load_state_dict(smr.random_state_dict(CFG, seed=7), strict=True)
    B = int(z['B'])
    b = _dev_batch(c, B, z['pos'], dev, t)
    tr, rot, tor = sm(b, keep_receptor_features=True)
    lig, rec = sm.last_complex.node_features(B, dev)
    assert rel_err(lig.cpu(), z['lig_node_attr']) < 1e-4 and rel_err(rec.cpu(), z['rec_node_attr']) < 1e-4
    errs = {name: rel_err(a.cpu(), z[name]) for name, a in (('tr', tr), ('rot', rot), ('tor', tor))}
    assert max(errs.values()) < 1e-4, errs


@pytest.mark.parametrize('shift,B,max_batch', [(0.0, 1, 1), (0.0, 2, 5), (60.0, 3, 3)])
def test_confidence_model_vs_oracle_edge_cases(dev, golden, shift, B, max_batch):
    """Confidence model against the CPU oracle: a single pose, fewer poses than max_batch (node numbering uses max_batch strides), and a
    ligand far outside the receptor (no ligand-atom and no ligand-residue edges: empty edge groups, BatchNorm of zeros)."""
    from oracle import confidence_ref as cr, graph_lite
    from disco_diffdock_amd.runtime import Context, Complex
    c = complex_from_npz(golden('complex_confidence'))
    cfg = cr.ConfidenceModelConfig()
    P = cr.random_state_dict(cfg, seed=5)
    rng = np.random.default_rng(3)
    n = len(c['lig_pos'])
    pocket = c['atom_pos'][17]
    lig0 = c['lig_pos'] - c['lig_pos'].mean(0, keepdims=True)
    pos = np.stack([lig0 + pocket + shift + rng.normal(0, 1.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    b = graph_lite.collate([graph_lite.add_atoms(to_graph(c), c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index']) for _ in range(B)])
    b['ligand'].pos = T(pos.reshape(-1, 3))
    for nt in ('ligand', 'receptor', 'atom'):
        b[nt].node_t = {k: torch.zeros(b[nt].num_nodes) for k in ('tr', 'rot', 'tor')}
    b.complex_t = {k: torch.zeros(B) for k in ('tr', 'rot', 'tor')}
    want, inter = cr.confidence_forward(P, cfg, b, return_intermediates=True)
    if shift:
        assert inter['counts']['la'] == 0 and inter['counts']['lr'] == 0
    ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, max_batch=max_batch)
    cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
    got = cx.confidence_forward(T(pos).to(dev))
    cnt = cx.confidence_counts()
    assert all(cnt[k] == inter['counts'][k] for k in ('ll', 'lr', 'la', 'rr'))
    assert rel_err(got.cpu(), want.reshape(B, -1)) < 1e-4
    with pytest.raises(RuntimeError, match='batch'):
        cx.confidence_forward(T(np.repeat(pos[:1], max_batch + 1, 0)).to(dev))


def test_pose_metrics(dev, golden):
    """evaluate.py:297-338 (uncorrected RMSD :313, centroid distance :315, min cross distance :331-332, min self distance :333-335,
    heavy-atom filter :297) restated in numpy here vs ddk_pose_metrics."""
    from disco_diffdock_amd.runtime import Complex
    from disco_diffdock_amd.tensor_layers import _shape_context
    c = complex_from_npz(golden('complex_diffdockS_score_model'))
    B, n = 5, len(c['lig_pos'])
    rng = np.random.default_rng(2)
    pos = np.stack([c['lig_pos'] + rng.normal(0, 2.0, size=(1, 3)) + rng.normal(0, 0.3, size=(n, 3)) for _ in range(B)]).astype(np.float32)
    ref = c['lig_pos'].astype(np.float32)
    for mask in (None, rng.random(n) > 0.3):
        f = np.ones(n, bool) if mask is None else mask
        ligand_pos, orig = pos[:, f], ref[None, f]
        rmsd = np.sqrt(((ligand_pos - orig) ** 2).sum(axis=2).mean(axis=1))
        cen = np.linalg.norm(ligand_pos.mean(axis=1) - orig.mean(axis=1), axis=1)
        cross = np.linalg.norm(c['rec_pos'][None, :, None, :] - ligand_pos[:, None, :, :], axis=-1).min(axis=(1, 2))
        sd = np.linalg.norm(ligand_pos[:, :, None, :] - ligand_pos[:, None, :, :], axis=-1)
        sd = np.where(np.eye(sd.shape[2]), np.inf, sd).min(axis=(1, 2))
        cx = Complex(_shape_context(0), c, max_batch=B)
        got = cx.pose_metrics(T(pos).to(dev), T(ref), None if mask is None else T(mask)).cpu().numpy()
        assert np.allclose(got, np.stack([rmsd, cen, cross, sd], 1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('t', [1.0, 0.3, 0.0])
def test_build_graph_vs_oracle(dev, t):
    """ddk_build_graph (score_model.py:310-408, 218-225 alone) against the oracle's graph builders: same edge multisets per group (the
    reference order inside a group is torch_cluster's, ours is sorted by the receiving node), group order and node numbering of the
    merged graph, every group sorted by edge_src."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    c = synthetic.make_complex(12, n_res=60, n_lig=23)
    ctx = Context(device=0)
    ctx.load_state_dict(smr.random_state_dict(CFG, seed=2))
    B = 3
    rng = np.random.default_rng(3)
    pos = np.stack([c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    cx = Complex(ctx, c, B)
    ei, off = cx.build_graph(T(pos).to(dev), t)
    ei, off = ei.cpu().long(), [int(v) for v in off]
    b = batch_of(c, B, pos)
    spr.set_time(b, t, t, t, B)
    g = smr.embed(smr.random_state_dict(CFG, seed=2), CFG, b, return_graph=True)[-1]
    s1, s2, s3 = g['splits']
    ref_ei = g['edge_index']
    bounds = [0, s1, s2, s3, ref_ei.shape[1]]
    from collections import Counter     # lig-lig is a MULTISET: a covalent bond inside the radius appears as bond edge and as radius edge
    want = {k: Counter(zip(ref_ei[0, bounds[k]:bounds[k + 1]].tolist(), ref_ei[1, bounds[k]:bounds[k + 1]].tolist())) for k in range(4)}
    assert off[0] == 0 and off[4] == ei.shape[1] == ref_ei.shape[1]
    for k in range(4):
        sl = ei[:, off[k]:off[k + 1]]
        assert Counter(zip(sl[0].tolist(), sl[1].tolist())) == want[k], k
        assert bool((sl[0, 1:] >= sl[0, :-1]).all()), k
    with pytest.raises(RuntimeError):
        cx.build_graph(T(pos), t)          # CPU tensor: no CPU path
    # a too small caller buffer is refused before anything is written (C ABI level)
    import ctypes as C
    small = torch.zeros(16, dtype=torch.int32, device=dev)
    off5 = torch.zeros(5, dtype=torch.int32, device=dev)
    p = T(pos).to(dev)
    rc = ctx.L.ddk_build_graph(ctx.h, cx.h, B, C.c_void_p(p.data_ptr()), C.c_float(t), C.c_void_p(small.data_ptr()), C.c_void_p(small.data_ptr()),
                               C.c_int64(16), C.c_void_p(off5.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b'worst case' in ctx.L.ddk_last_error(ctx.h) and int(small.abs().sum()) == 0


@pytest.mark.parametrize('case', range(10))
def test_randomised_shapes_vs_oracle(dev, tables, case):
    """Randomised sweep over shapes the fixed cases do not hit together (5-97 residues, 12-48 ligand atoms, batch 1-5, t in [0, 1],
    ligand in the pocket / far from the receptor (no cross edges) / compressed so that the neighbour caps bind / spread out):
    tr, rot, tor against the oracle at the north-star bar.  (tests/devtools/fuzz_parity.py runs longer sweeps of the same kind.)"""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    rng = np.random.default_rng(9000 + case)
    seed = int(rng.integers(100000))
    n_res, n_lig = int(rng.choice([5, 17, 40, 64, 97])), int(rng.choice([12, 18, 25, 33, 48]))
    B, t = int(rng.choice([1, 2, 3, 5])), float(rng.choice([1.0, 0.9, 0.5, 0.2, 0.03, 0.0]))
    place = ['pocket', 'far', 'compressed', 'spread'][case % 4]
    c = synthetic.make_complex(seed, n_res=n_res, n_lig=n_lig)
    P = smr.random_state_dict(CFG, seed=seed % 1000)
    base = c['lig_pos'].astype(np.float64)
    cen = base.mean(0, keepdims=True)
    pos = np.stack([{'far': base + np.array([[150.0, -80.0, 60.0]]), 'compressed': cen + 0.35 * (base - cen),
                     'spread': base + rng.normal(0, 12.0, size=(1, 3)),
                     'pocket': base + rng.normal(0, 2.0, size=(1, 3)) + rng.normal(0, 0.2, size=base.shape)}[place] for _ in range(B)]).astype(np.float32)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, B)
    tr, rot, tor = cx.score_forward(T(pos).to(dev), t, t, t)
    if place == 'far':
        assert cx.graph_stats()['E_lr'] == 0
    b = batch_of(c, B, pos)
    spr.set_time(b, t, t, t, B)
    tr_r, rot_r, tor_r = smr.score_model_forward(P, CFG, b, tables[0], tables[1])
    assert rel_err(tr.cpu(), tr_r) < 1e-4 and rel_err(rot.cpu(), rot_r) < 1e-4
    assert tor.numel() == tor_r.numel() and (tor_r.numel() == 0 or rel_err(tor.cpu(), tor_r) < 1e-4)
