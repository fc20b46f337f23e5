"""CPU tests: the oracle restatement against the golden vectors produced by the reference
(tests/golden/make_golden.py).  Tolerances: fp32 round-off class (1e-5 relative to the tensor max)
unless stated; the north-star tolerance for scores (1e-4 relative) is used for the full model."""
import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr
from oracle import e3nn_lite as o3
from helpers import complex_from_npz, to_graph, batch_of, rel_err

CFG = smr.ScoreModelConfig()
T = torch.from_numpy


@pytest.mark.parametrize('l', range(5))
def test_faster_tp(golden, l):
    z = golden(f'faster_tp_l{l}')
    i_irr, o_irr = CFG.conv_irreps(l)
    assert smr.faster_tp_weight_numel(i_irr, o_irr) == int(z['weight_numel']) == [720, 936, 1152, 1872, 1872][l]
    out = smr.faster_tensor_product(T(z['x']), T(z['sh']), T(z['w']), i_irr, o_irr)
    assert rel_err(out, z['out']) < 2e-6


@pytest.mark.parametrize('l', range(5))
@pytest.mark.parametrize('bn', [0, 1])
def test_conv_layer(golden, l, bn):
    z = golden(f'conv_layer_l{l}_bn{bn}')
    P = {'L.' + k: v for k, v in smr.random_conv_layer_params(CFG, l, int(z['param_seed']), bool(bn)).items()}
    i_irr, o_irr = CFG.conv_irreps(l)
    s = z['splits']
    ea = T(z['edge_attr'])
    out = smr.tp_conv_layer(P, 'L', T(z['node']), T(z['edge_index']), [ea[s[i]:s[i + 1]] for i in range(4)], T(z['sh']),
                            i_irr, '1x0e+1x1o', o_irr, residual=True, batch_norm=bool(bn), faster=True, edge_groups=4)
    assert rel_err(out, z['out']) < 5e-6


def test_faster_tp_equals_fctp():
    """Self-consistency of the e3nn restatement: FasterTensorProduct == FullyConnectedTensorProduct after
    re-laying the weights (same algebra: 1/sqrt(fan-in), dot/sqrt3, cross/sqrt2)."""
    i_irr, o_irr = CFG.conv_irreps(3)
    g = torch.Generator().manual_seed(0)
    E = 8
    x, sh = torch.randn(E, 84, generator=g, dtype=torch.float64), torch.randn(E, 4, generator=g, dtype=torch.float64)
    fctp = o3.FullyConnectedTensorProduct(i_irr, '1x0e+1x1o', o_irr)
    assert fctp.weight_numel == 1872
    w_f = torch.randn(E, 1872, generator=g, dtype=torch.float64)
    shapes = smr.faster_tp_weight_shapes(i_irr, o_irr)
    names = {0: '0e', 1: '1o', 2: '1e', 3: '0o'}
    # FasterTP row blocks per output irrep, in order of the in1 irreps that feed it
    row_off = {'0e': {0: 0, 1: 24}, '1o': {0: 0, 1: 24, 2: 30}, '1e': {1: 0, 2: 6, 3: 12}, '0o': {2: 0, 3: 6}}
    blk_off, o = {}, 0
    for k in ('0e', '1o', '1e', '0o'):
        blk_off[k] = o
        o += shapes[k][0] * shapes[k][1]
    w_fast = torch.zeros(E, 1872, dtype=torch.float64)
    for (i1, i2, io, off, shape) in fctp.instructions:
        key = names[io]
        n_in, n_out = shapes[key]
        blk = w_f[:, off:off + shape[0] * shape[2]].reshape(E, shape[0], shape[2])
        dst = w_fast[:, blk_off[key]:blk_off[key] + n_in * n_out].view(E, n_in, n_out)
        dst[:, row_off[key][i1]:row_off[key][i1] + shape[0], :] = blk
    a = smr.faster_tensor_product(x, sh, w_fast, i_irr, o_irr)
    b = fctp(x, sh, w_f)
    assert rel_err(a, b) < 1e-12


def test_smearing_and_time_embedding(golden):
    z = golden('gaussian_smearing')
    for stop in (5, 30, 80):
        out = smr.gaussian_smearing(T(z['d']), float(stop), 32, torch.float32)
        assert rel_err(out, z[f'out_{stop}']) < 1e-6
    z = golden('time_embedding')
    emb = smr.sinusoidal_embedding(1000 * T(z['t']), 32)
    assert rel_err(emb, z['emb']) < 1e-6
    sched = spr.get_t_schedule(20)
    sig = np.asarray([smr.t_to_sigma(t, t, t, CFG) for t in sched])
    assert rel_err(sig, z['sigmas']) < 1e-12


def test_atom_encoder(golden):
    z = golden('atom_encoder')
    P = {'E.' + k[2:]: T(z[k]) for k in z.files if k.startswith('P.')}
    out = smr.atom_encoder(T(z['x']), P, 'E', 16)
    assert rel_err(out, z['out']) < 2e-6


def test_geometry(golden):
    z = golden('axis_angle')
    assert rel_err(spr.axis_angle_to_matrix(T(z['aa'])), z['R']) < 1e-6
    z = golden('kabsch')
    R, t = spr.kabsch_batch(T(z['A']), T(z['B']))
    assert rel_err(R, z['R']) < 1e-5 and rel_err(t, z['t']) < 1e-5
    assert float(torch.linalg.det(R)[5]) > 0.99        # reflection case fixed


def test_conformer_update(golden):
    z = golden('conformer_update')
    c = complex_from_npz(golden('toy_complex'))
    B = int(z['B'])
    b = batch_of(c, B, z['pos'])
    mr = T(z['mask_rotate'])
    flex = spr.modify_conformer_torsion_angles_batch(T(z['pos']).reshape(B, -1, 3), T(z['rot_bonds']), mr, T(z['tor']).reshape(B, -1))
    assert rel_err(flex, z['flex']) < 1e-5
    new = spr.modify_conformer_batch(T(z['pos']), b, T(z['tr']), T(z['rot']), T(z['tor']), mr)
    assert rel_err(new, z['new_pos']) < 2e-5
    rigid = spr.modify_conformer_batch(T(z['pos']), b, T(z['tr']), T(z['rot']), None, mr)
    assert rel_err(rigid, z['rigid_only']) < 1e-5


README_S = dict(temp_sampling=[1.886430780895051, 5.659562317960644, 2.8888668488630156],
                temp_psi=[0.07085125444659945, 2.686505606141324, 4.089493860493927],
                temp_sigma_data=[0.3617563913086843, 0.7437588205919711, 0.08897393057297842])


@pytest.mark.parametrize('tag,kw', [('plain', {}), ('lowtemp', README_S), ('ode', dict(ode=True))])
def test_sde_steps(golden, tables, tag, kw, monkeypatch):
    """sampling() arithmetic with a fixed-score model and the reference's torch.manual_seed noise stream."""
    z = golden(f'sde_steps_{tag}')
    c = complex_from_npz(golden('toy_complex'))
    B = z['tr'].shape[0]
    fixed = (T(z['tr']), T(z['rot']), T(z['tor']))
    monkeypatch.setattr(smr, 'score_model_forward', lambda *a, **k: tuple(f.clone() for f in fixed))
    dl = [to_graph(c) for _ in range(B)]
    steps = int(z['steps'])
    sched = spr.get_t_schedule(steps)
    torch.manual_seed(int(z['seed']))
    out, _ = spr.sampling(dl, {}, CFG, tables[0], tables[1], steps, sched, sched, sched, batch_size=B,
                          no_final_step_noise=True, **kw)
    pos = torch.cat([d['ligand'].pos for d in out])
    assert rel_err(pos, z['pos_out']) < 2e-5


def _cfg_for(tag):
    if tag.startswith('disco'):
        return smr.ScoreModelConfig(latent_dim=2, latent_vocab=1, latent_droprate=0.1)
    return smr.ScoreModelConfig(latent_dim=0, latent_vocab=64)


@pytest.mark.parametrize('tag', ['diffdockS_score_model', 'disco_diffdockS_score_model'])
def test_state_dict_layout(golden, tag):
    z = golden(f'weights_probe_{tag}')
    P = smr.random_state_dict(_cfg_for(tag), seed=int(z['seed']))
    assert len(P) == int(z['n_tensors']) == {'diffdockS_score_model': 171, 'disco_diffdockS_score_model': 176}[tag]
    assert sum(v.numel() for v in P.values()) == int(z['n_elements'])
    assert abs(sum(float(v.double().sum()) for v in P.values()) - float(z['checksum'])) < 1e-6


@pytest.mark.parametrize('tag', ['diffdockS_score_model', 'disco_diffdockS_score_model'])
@pytest.mark.parametrize('t', [1.0, 0.55, 0.05])
def test_score_model_forward(golden, tables, tag, t):
    """Tier B: the restatement vs the reference's own score_model.py run on the *_lite stand-ins."""
    z = golden(f'score_{tag}_t{t}')
    cfg = _cfg_for(tag)
    P = smr.random_state_dict(cfg, seed=7)
    c = complex_from_npz(golden(f'complex_{tag}'))
    B = int(z['B'])
    b = batch_of(c, B, z['pos'])
    spr.set_time(b, t, t, t, B)
    if cfg.latent_dim > 0:
        b['ligand'].latent_h, b['receptor'].latent_h = T(z['latent_l']), T(z['latent_r'])
        b['ligand'].unconditional = torch.zeros(b['ligand'].num_nodes, 1)
        b['receptor'].unconditional = torch.zeros(b['receptor'].num_nodes, 1)
    tr, rot, tor, inter = smr.score_model_forward(P, cfg, b, tables[0], tables[1], return_intermediates=True)
    assert rel_err(inter['lig_node_attr'], z['lig_node_attr']) < 1e-4
    assert rel_err(inter['rec_node_attr'], z['rec_node_attr']) < 1e-4
    for name, a in (('tr', tr), ('rot', rot), ('tor', tor)):
        assert rel_err(a, z[name]) < 1e-4, name


def test_trajectory(golden, tables):
    tag = 'diffdockS_score_model'
    z = golden(f'trajectory_{tag}')
    cfg = _cfg_for(tag)
    P = smr.random_state_dict(cfg, seed=7)
    c = complex_from_npz(golden(f'complex_{tag}'))
    B = 2
    dl = [to_graph(c) for _ in range(B)]
    n = len(c['lig_pos'])
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
    steps = int(z['steps'])
    sched = spr.get_t_schedule(steps)
    torch.manual_seed(int(z['seed']))
    out, _ = spr.sampling(dl, P, cfg, tables[0], tables[1], steps, sched, sched, sched, batch_size=B,
                          no_final_step_noise=True, **README_S)
    pos = torch.cat([d['ligand'].pos for d in out])
    assert rel_err(pos, z['pos_out']) < 1e-4


def test_equivariance(tables):
    """Self-check the reference lacks: tr/rot rotate as vectors, tor is invariant, under a
    random proper rotation + translation of the whole complex."""
    from scipy.spatial.transform import Rotation
    from disco_diffdock_amd import synthetic
    cfg = _cfg_for('diffdockS_score_model')
    P = smr.random_state_dict(cfg, seed=3)
    c = synthetic.make_complex(5, n_res=30, n_lig=20)
    Rm = torch.from_numpy(Rotation.random(random_state=2).as_matrix()).double()
    shift = torch.tensor([[3.0, -2.0, 5.0]], dtype=torch.float64)
    outs = []
    for rot in (False, True):
        b = batch_of(c, 2)
        b['ligand'].pos = b['ligand'].pos.double() + torch.tensor([[1.0, 2.0, -1.0]], dtype=torch.float64)
        b['receptor'].pos = b['receptor'].pos.double()
        if rot:
            b['ligand'].pos = b['ligand'].pos @ Rm.T + shift
            b['receptor'].pos = b['receptor'].pos @ Rm.T + shift
        spr.set_time(b, 0.4, 0.4, 0.4, 2)
        outs.append(smr.score_model_forward(P, cfg, b, tables[0], tables[1], dtype=torch.float64))
    (tr0, rot0, tor0), (tr1, rot1, tor1) = outs
    assert rel_err(tr0 @ Rm.T, tr1) < 1e-9
    assert rel_err(rot0 @ Rm.T, rot1) < 1e-9
    assert rel_err(tor0, tor1) < 1e-9


def test_ar_latent_model(golden):
    """Tier B: AR latent model logits and the argmax-decoded latents vs the reference's PretrainedScoreEncoder / encode_ar."""
    from oracle import ar_ref
    tag = 'disco_diffdockS_score_model'
    z = golden(f'ar_{tag}')
    cfg = _cfg_for(tag)
    P = ar_ref.random_ar_state_dict(cfg, ar_ns=16, hidden=128, seed=int(z['seed']))
    assert len(P) == int(z['n_tensors'])
    c = complex_from_npz(golden(f'complex_{tag}'))
    B = int(z['B'])
    b = batch_of(c, B, z['pos'])
    b['ligand'].input_latent = torch.zeros(b['ligand'].num_nodes, cfg.latent_dim)
    b['receptor'].input_latent = torch.zeros(b['receptor'].num_nodes, cfg.latent_dim)
    logits = ar_ref.ar_logits(P, cfg, 16, b)
    assert rel_err(logits, z['logits0']) < 1e-4
    b = batch_of(c, B, z['pos'])
    lat_l, lat_r = ar_ref.encode_ar(P, cfg, 16, b, sampling_temperature=100.0)
    assert torch.equal(lat_l, T(z['latent_l'])) and torch.equal(lat_r, T(z['latent_r']))
    assert float(lat_l.sum() + lat_r.sum()) == B * cfg.latent_dim


README_D = dict(temp_sampling=[1.546842681537956, 4.005218254154881, 3.6499018519649384],
                temp_psi=[1.2685697872473618, 1.2760150490206228, 2.0625243924678136],
                temp_sigma_data=[0.8456140350087653, 0.453446580767075, 0.3292199987743284])


def test_disco_trajectory_with_ar_and_cfg(golden, tables):
    """Tier B: the reference's own sampling() on the DisCo path (AR decoding -> latents -> CFG on the middle step)."""
    from oracle import ar_ref, graph_lite
    tag = 'disco_diffdockS_score_model'
    z = golden(f'trajectory_{tag}')
    cfg = _cfg_for(tag)
    P = smr.random_state_dict(cfg, seed=7)
    P_ar = ar_ref.random_ar_state_dict(cfg, ar_ns=16, hidden=128, seed=int(z['ar_seed']))
    c = complex_from_npz(golden(f'complex_{tag}'))
    B, n_l, n_r = 2, len(c['lig_pos']), len(c['rec_pos'])
    dl = [to_graph(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n_l:(i + 1) * n_l])
    lat_l, lat_r = ar_ref.encode_ar(P_ar, cfg, 16, graph_lite.collate(dl), sampling_temperature=100.0)
    strs = []
    for i, d in enumerate(dl):
        d['ligand'].latent_h, d['receptor'].latent_h = lat_l[i * n_l:(i + 1) * n_l], lat_r[i * n_r:(i + 1) * n_r]
        d['ligand'].unconditional, d['receptor'].unconditional = torch.zeros(n_l, 1), torch.zeros(n_r, 1)
        s = ''
        for j in range(cfg.latent_dim):
            s += ('L' + str(int(d['ligand'].latent_h[:, j].argmax()))) if d['ligand'].latent_h[:, j].sum() == 1 else \
                 ('R' + str(int(d['receptor'].latent_h[:, j].argmax())))
        strs.append(s)
    assert strs == [str(x) for x in z['latent_str']]
    steps = int(z['steps'])
    sched = spr.get_t_schedule(steps)
    torch.manual_seed(int(z['seed']))
    out, _ = spr.sampling(dl, P, cfg, tables[0], tables[1], steps, sched, sched, sched, batch_size=B, no_final_step_noise=True,
                          classifier_free_guidance_weight=0.7, cfg_start=0.9, cfg_end=0.2, **README_D)
    assert rel_err(torch.cat([d['ligand'].pos for d in out]), z['pos_out']) < 1e-4


def test_confidence_model_golden(golden):
    """the oracle restatement of the all-atom confidence model reproduces the output of the reference's own
    models/all_atom_score_model.py (confidence_mode, paper_confidence_model yml through get_model) on the same stand-ins."""
    from oracle import confidence_ref as cr, graph_lite
    z, c = golden('confidence_paper_model'), complex_from_npz(golden('complex_confidence'))
    cfg = cr.ConfidenceModelConfig()
    P = cr.random_state_dict(cfg, seed=int(z['seed']))
    assert len(P) == int(z['n_tensors']) and sum(v.numel() for v in P.values()) == int(z['n_elements'])
    B = int(z['B'])

    def graph():
        return graph_lite.add_atoms(to_graph(c), c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])

    b = graph_lite.collate([graph() for _ in range(B)])
    b['ligand'].pos = torch.as_tensor(z['pos']).float()
    for nt in ('ligand', 'receptor', 'atom'):
        b[nt].node_t = {k: torch.zeros(b[nt].num_nodes) for k in ('tr', 'rot', 'tor')}
    b.complex_t = {k: torch.zeros(B) for k in ('tr', 'rot', 'tor')}
    conf, inter = cr.confidence_forward(P, cfg, b, return_intermediates=True)
    assert [inter['counts'][k] for k in ('ll', 'lr', 'la', 'aa', 'ar', 'rr')] == list(z['counts'])
    assert rel_err(inter['lig_node_attr'], z['lig_node_attr']) < 1e-5
    assert rel_err(conf, z['confidence']) < 1e-5


def test_sampling_with_confidence_golden(golden, tables):
    """Reference sampling(confidence_model=..., confidence_data_list=...) == oracle sampler followed by the oracle confidence model."""
    from oracle import confidence_ref as cr, graph_lite
    z, c = golden('trajectory_confidence'), complex_from_npz(golden('complex_confidence'))
    cfg = _cfg_for('diffdockS_score_model')
    P = smr.random_state_dict(cfg, seed=int(z['score_seed']))
    n = len(c['lig_pos'])
    B, steps = len(z['pos0']) // n, int(z['steps'])
    dl = [to_graph(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
    sched = spr.get_t_schedule(steps)
    torch.manual_seed(int(z['seed']))
    out, _ = spr.sampling(dl, P, cfg, tables[0], tables[1], steps, sched, sched, sched, batch_size=B, no_final_step_noise=True,
                          temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5)
    pos = torch.cat([d['ligand'].pos for d in out])
    assert rel_err(pos, z['pos_out']) < 1e-4
    ccfg = cr.ConfidenceModelConfig()
    b = graph_lite.collate([graph_lite.add_atoms(to_graph(c), c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index']) for _ in range(B)])
    b['ligand'].pos = pos.float()
    for nt in ('ligand', 'receptor', 'atom'):
        b[nt].node_t = {k: torch.zeros(b[nt].num_nodes) for k in ('tr', 'rot', 'tor')}
    b.complex_t = {k: torch.zeros(B) for k in ('tr', 'rot', 'tor')}
    conf = cr.confidence_forward(cr.random_state_dict(ccfg, seed=int(z['conf_seed'])), ccfg, b)
    assert rel_err(conf, z['confidence']) < 1e-4


def _cg_conf_cfg():
    cfg = _cfg_for('diffdockS_score_model')
    cfg.confidence_mode, cfg.num_confidence_outputs = True, 3
    return cfg


def test_cg_confidence_model_golden(golden):
    """The oracle's coarse-grained confidence model (score_model_ref.confidence_forward: models/score_model.py in confidence_mode, complex_t used as
    sigma) reproduces the reference's own model built by get_model(DiffDock-S yml + rmsd_classification_cutoff, confidence_mode=True)."""
    from oracle import graph_lite
    z, c = golden('cg_confidence_model'), complex_from_npz(golden('complex_cg_confidence'))
    cfg = _cg_conf_cfg()
    P = smr.random_state_dict(cfg, seed=int(z['seed']))
    assert 'center_edge_embedding.0.weight' not in P and 'final_conv.fc.0.weight' not in P and P['confidence_predictor.8.weight'].shape == (3, 24)
    B = int(z['B'])
    b = graph_lite.collate([to_graph(c) for _ in range(B)])
    b['ligand'].pos = T(z['pos']).float()
    spr.set_time(b, *[float(x) for x in z['t']], B)
    assert rel_err(smr.embed(P, cfg, b)[0], z['lig_node_attr']) < 1e-5
    conf = smr.confidence_forward(P, cfg, b)
    assert tuple(conf.shape) == (B, 3) and rel_err(conf, z['confidence']) < 1e-5


def test_sampling_with_cg_confidence_golden(golden, tables):
    """Reference sampling(confidence_model=<coarse-grained model>, confidence_data_list=None) (utils/sampling.py:239-240) == the oracle sampler followed
    by the oracle's confidence model on the score batch at the LAST step's times (the reference does not reset them in this branch)."""
    from oracle import graph_lite
    z, c = golden('trajectory_cg_confidence'), complex_from_npz(golden('complex_cg_confidence'))
    cfg = _cfg_for('diffdockS_score_model')
    P = smr.random_state_dict(cfg, seed=int(z['score_seed']))
    n = len(c['lig_pos'])
    B, steps = len(z['pos0']) // n, int(z['steps'])
    dl = [to_graph(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
    sched = spr.get_t_schedule(steps)
    torch.manual_seed(int(z['seed']))
    out, _ = spr.sampling(dl, P, cfg, tables[0], tables[1], steps, sched, sched, sched, batch_size=B, no_final_step_noise=True,
                          temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5)
    pos = torch.cat([d['ligand'].pos for d in out])
    assert rel_err(pos, z['pos_out']) < 1e-4
    ccfg = _cg_conf_cfg()
    b = graph_lite.collate([to_graph(c) for _ in range(B)])
    b['ligand'].pos = pos.float()
    spr.set_time(b, float(sched[-1]), float(sched[-1]), float(sched[-1]), B)
    conf = smr.confidence_forward(smr.random_state_dict(ccfg, seed=int(z['conf_seed'])), ccfg, b)
    assert rel_err(conf, z['confidence']) < 1e-4
