"""Round-5 GPU tests (VERDICT r04 "Next" #1, #2, #4, #5d, #9), through the C ABI / the shipped entry points:

* the streaming FasterTensorProduct kernel (k_tp.hip, the BASELINE metric's HBM-bound form) at E = 200 000 against the fp64 oracle, every conv layer's
  shape, and on a weight tensor that is only 4-byte aligned;
* ddk_config.conv_kernel = 2 (round 5's k_conv_y.hip, now under tools/variants/) is refused by the product library;
* the pocket-bound bracket of bench.py as a 20-step oracle trajectory;
* ddk_create on a device ordinal the box does not have."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr
from helpers import batch_of, chan_err, elem_err, rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CFG = smr.ScoreModelConfig(latent_vocab=64)
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a MI355X'
    from disco_diffdock_amd import build
    build.build(verbose=False)
    return torch.device('cuda:0')


@pytest.mark.parametrize('l', range(5))
def test_tp_stream_kernel_vs_oracle_200k(dev, l):
    """FasterTensorProduct.forward (models/tensor_layers.py:65-116) on 200 000 edges against the fp64 restatement, in chunks the CPU finishes in seconds:
    every workgroup of tp_stream_kernel walks many edges (grid = 8192 waves), both register sets and the loop's tail are exercised."""
    from disco_diffdock_amd.tensor_layers import FasterTensorProduct
    i_irr, o_irr = CFG.conv_irreps(l)
    tp = FasterTensorProduct(i_irr, '1x0e+1x1o', o_irr)
    E = 200_000 + 37                      # (not a multiple of the grid)
    g = torch.Generator().manual_seed(100 + l)
    x = torch.randn(E, smr.irreps_dim(i_irr), generator=g)
    sh = torch.randn(E, 4, generator=g)
    w = torch.randn(E, tp.weight_numel, generator=g)
    out = tp(x.to(dev), sh.to(dev), w.to(dev)).cpu()
    worst = 0.0
    for a in list(range(0, E, 50_000)) + [E - 3000]:
        b = min(a + 3000, E)
        ref = smr.faster_tensor_product(x[a:b].double(), sh[a:b].double(), w[a:b].double(), i_irr, o_irr)
        worst = max(worst, rel_err(out[a:b], ref))
    assert worst < 1e-5, worst


def test_tp_stream_kernel_dword_aligned_weights(dev):
    """A weight tensor that starts 4 bytes behind a 16-byte boundary (a view into a larger buffer): the kernel's float4 / float2 reads need dword alignment
    only.  Same results."""
    from disco_diffdock_amd.tensor_layers import FasterTensorProduct
    i_irr, o_irr = CFG.conv_irreps(3)
    tp = FasterTensorProduct(i_irr, '1x0e+1x1o', o_irr)
    E = 5000
    g = torch.Generator().manual_seed(7)
    x, sh, w = torch.randn(E, 84, generator=g), torch.randn(E, 4, generator=g), torch.randn(E, tp.weight_numel, generator=g)
    from disco_diffdock_amd.tensor_layers import _shape_context
    ctx = _shape_context(0)
    buf = torch.empty(E * tp.weight_numel + 1, device=dev)
    wv = buf[1:].view(E, tp.weight_numel)
    wv.copy_(w)
    assert wv.data_ptr() % 16 == 4 and wv.is_contiguous()
    xd, sd = x.to(dev), sh.to(dev)
    out = torch.empty((E, 84), device=dev)
    ctx._check(ctx.L.ddk_tp_forward(ctx.h, 3, C.c_void_p(xd.data_ptr()), C.c_void_p(sd.data_ptr()), C.c_void_p(wv.data_ptr()), E, C.c_void_p(out.data_ptr()), None),
               'ddk_tp_forward')
    torch.cuda.synchronize()
    ref = smr.faster_tensor_product(x.double(), sh.double(), w.double(), i_irr, o_irr)
    assert rel_err(out.cpu(), ref) < 1e-5
    assert rel_err(out.cpu(), tp(xd, sd, w.to(dev)).cpu()) < 1e-6


def test_conv_kernel_2_is_refused_by_the_product_library(dev):
    """Round 5's ddk_config.conv_kernel = 2 (the one-wave-per-SIMD form, 9 % slower) left libddk.so in round 6 (tools/variants/k_conv_y.hip,
    tools/build_variant_y.sh): the product library refuses the value with a message instead of silently running kernel 0."""
    from disco_diffdock_amd.runtime import Context
    with pytest.raises(RuntimeError) as ei:
        Context(device=0, conv_kernel=2)
    assert 'conv_kernel' in str(ei.value) and 'variants' in str(ei.value), str(ei.value)


def test_create_on_a_missing_device_fails_cleanly(dev):
    """VERDICT r04 #5d: a context on a device ordinal the box does not have is refused with an error string, not a crash (one-GPU boxes)."""
    from disco_diffdock_amd import _lib
    from disco_diffdock_amd.runtime import Context
    n = torch.cuda.device_count()
    with pytest.raises(RuntimeError) as ei:
        Context(device=n)
    assert 'device' in str(ei.value).lower() or 'hip' in str(ei.value).lower(), str(ei.value)
    ctx = Context(device=0)          # ... and the process is still usable
    assert ctx.h is not None


def test_pocket_bound_trajectory_vs_oracle(dev, tables):
    """the pocket-bound bracket of bench.py (`value_pocket_bound`: start poses inside the pocket = rotation about the centroid + N(0, 1 A), the
    N(0,1) draws scaled by 0.2, README low-temperature coefficients) as a 20-step trajectory of the WHOLE 40-sample batch on the device - stepped one reverse
    step per call so that every step's scores and cross-edge count can be read - against oracle.sampler_ref on four of the samples (the samples of a batch are
    independent; same noise).  Every step keeps >= 2 500 cross edges per sample (what a trained checkpoint's trajectories look like; every other 20-step oracle
    test uses wandering ligands).  Bars: final poses 1e-3 of the receptor scale, per-step scores 1e-4 (relative to each score vector's largest element)."""
    from functools import partial
    from argparse import Namespace
    from scipy.spatial.transform import Rotation
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    from disco_diffdock_amd.sampling import step_coefficients
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from helpers import to_graph
    from test_gpu_model import README_S
    from test_gpu_round3 import _record_drift
    args = Namespace(tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03, tor_sigma_max=3.14, no_torsion=False)
    c = synthetic.make_complex(0, n_res=300)          # (the bench's first complex)
    P = smr.random_state_dict(CFG, seed=21)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    B, steps = 40, 20
    cx = Complex(ctx, c, B)
    sched = get_t_schedule(steps)
    t_arr, sc, nc = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=args), args, False, False, True, README_S['temp_sampling'],
                                      README_S['temp_psi'], README_S['temp_sigma_data'])
    rng = np.random.default_rng(1000)
    lp = c['lig_pos'].astype(np.float64)
    ctr = lp.mean(0, keepdims=True)
    pos0 = np.stack([(lp - ctr) @ Rotation.random(random_state=rng).as_matrix().T + ctr + rng.normal(0, 1.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    z = 0.2 * torch.randn(steps, B, 6 + cx.R, generator=torch.Generator().manual_seed(19))
    sub = [0, 13, 26, 39]
    pos = T(pos0.copy()).to(dev)
    dev_scores, cross = [], []
    for s in range(steps):
        t = float(sched[s])
        tr, rot, tor = cx.score_forward(pos, t, t, t)
        cross.append(cx.graph_stats()['E_lr'] / B)
        dev_scores.append((tr.cpu()[sub], rot.cpu()[sub], tor.cpu().reshape(B, -1)[sub].reshape(-1)))
        cx.sample(pos, t_arr[s:s + 1], sc[s:s + 1], nc[s:s + 1], z[s:s + 1].to(dev))
    assert min(cross) >= 2500, cross
    dl = []
    for i in sub:
        g = to_graph(c)
        g['ligand'].pos = T(pos0[i])
        dl.append(g)
    zs = z[:, sub]
    nf = lambda b, t, name, shape: {'tr': zs[t, :, 0:3], 'rot': zs[t, :, 3:6], 'tor': zs[t, :, 6:].reshape(-1)}[name]
    trace = []
    ref, _ = spr.sampling(dl, P, CFG, tables[0], tables[1], steps, sched, sched, sched, noise_fn=nf, batch_size=len(dl), no_final_step_noise=True, trace=trace, **README_S)
    r = torch.cat([g['ligand'].pos for g in ref])
    err_pos = rel_err(pos.cpu()[sub].reshape(-1, 3), r)
    per_step = []
    for s in range(steps):
        e = 0.0
        for a, b in zip(dev_scores[s], (trace[s]['tr_score'], trace[s]['rot_score'], trace[s]['tor_score'])):
            if b.numel():
                e = max(e, rel_err(a, b))
        per_step.append(e)
    print(f'pocket-bound 20-step trajectory, B = 40 (oracle on samples {sub}): poses {err_pos:.2e}, per-step scores max {max(per_step):.2e} '
          f'(first {per_step[0]:.1e}, last {per_step[-1]:.1e}); cross edges per sample {min(cross):.0f} .. {max(cross):.0f}')
    _record_drift('pocket_bound_trajectory_20_steps_300_residues_B40_poses_vs_oracle', err_pos, bar=1e-3, min_cross_edges_per_sample=float(min(cross)))
    _record_drift('pocket_bound_trajectory_20_steps_300_residues_B40_per_step_scores_vs_oracle', max(per_step), bar=1e-4, per_step=[float(v) for v in per_step])
    assert err_pos < 1e-3 and max(per_step) < 1e-4


def test_confidence_ligand_atom_capacity_cannot_overflow(dev):
    """the ligand-atom edge list of the all-atom confidence model (radius(atom.pos, ligand.pos, 5 A, max_num_neighbors = 10000),
    all_atom_score_model.py:409-410; the reference has no capacity) is sized from the receptor's own geometry (no ligand atom, wherever a pose puts it, has more
    atoms within r than the densest atom has within 2r).  A receptor whose atoms are packed 2.2 x closer than a protein's, with compressed ligands dropped on its
    densest spot: round 4's capacity (96 per ligand atom on average) overflowed here and returned -1000 for the whole batch; now the edge count matches the
    oracle's, the status flag stays clear and the confidences agree."""
    from oracle import confidence_ref as cr, graph_lite
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    from helpers import to_graph
    c = synthetic.make_complex(43, n_res=120, n_lig=30)
    synthetic.add_receptor_atoms(c, np.random.default_rng(43))
    c['atom_pos'] = (0.45 * np.asarray(c['atom_pos'], np.float64)).astype(np.float32)          # ~11 x the heavy-atom density of a protein
    ap = c['atom_pos'].astype(np.float64)
    d2 = ((ap[:, None] - ap[None]) ** 2).sum(-1)
    dense = int((d2 < 25.0).sum(1).argmax())
    assert int((d2[dense] < 25.0).sum()) > 150                                                   # far beyond 96 atoms within 5 A
    cfg = cr.ConfidenceModelConfig()
    P = cr.random_state_dict(cfg, seed=7)
    B = 3
    base = c['lig_pos'].astype(np.float64)
    rng = np.random.default_rng(2)
    pos = np.stack([ap[dense] + 0.3 * (base - base.mean(0)) + rng.normal(0, 0.3, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    b = graph_lite.collate([graph_lite.add_atoms(to_graph(c), c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index']) for _ in range(B)])
    b['ligand'].pos = T(pos.reshape(-1, 3))
    for nt in ('ligand', 'receptor', 'atom'):
        b[nt].node_t = {k: torch.zeros(b[nt].num_nodes) for k in ('tr', 'rot', 'tor')}
    b.complex_t = {k: torch.zeros(B) for k in ('tr', 'rot', 'tor')}
    want, inter = cr.confidence_forward(P, cfg, b, return_intermediates=True)
    n_lig = base.shape[0]
    assert inter['counts']['la'] > 96 * n_lig * B                                                # round 4's capacity
    ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, max_batch=B)
    cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
    got = cx.confidence_forward(T(pos).to(dev))
    cnt = cx.confidence_counts()
    assert cnt['la'] == inter['counts']['la']
    st = cx.confidence_status_async()
    torch.cuda.synchronize()
    assert int(st[19]) == 0
    assert bool(torch.isfinite(got).all()) and rel_err(got.cpu(), want.reshape(B, -1)) < 1e-4


def test_cg_confidence_reads_the_last_executed_steps_time(dev, golden):
    """evaluate.py:269 passes the FULL schedule with inference_steps = actual_steps; the reference leaves complex_t at the last EXECUTED step
    (utils/sampling.py:105-111), which a coarse-grained confidence model (confidence_data_list = None, :239-240) reads as its sigmas.  A 25-entry schedule
    run for 20 steps: the confidences are those of t = schedule[19], not schedule[24]."""
    from functools import partial
    from helpers import complex_from_npz
    from test_gpu_model import ARGS_S, _ref_noise
    from test_gpu_round4 import _cg_conf_model
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.data import from_arrays, DataLoader
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    z, c = golden('trajectory_cg_confidence'), complex_from_npz(golden('complex_cg_confidence'))
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    model.score_model.load_state_dict(smr.random_state_dict(smr.ScoreModelConfig(latent_vocab=64), seed=int(z['score_seed'])), strict=True)
    cm, cargs = _cg_conf_model(dev, int(z['conf_seed']))
    n = len(c['lig_pos'])
    B, steps = len(z['pos0']) // n, 20
    dl = [from_arrays(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = torch.from_numpy(z['pos0'][i * n:(i + 1) * n])
    sched = get_t_schedule(25)
    noise = [_ref_noise(int(z['seed']), steps, B, int(c['edge_mask'].sum()))]
    out, conf = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B, no_final_step_noise=True, use_latent=False,
                         noise=noise, confidence_model=cm, confidence_data_list=None, confidence_model_args=cargs, temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5)
    pos = torch.stack([d['ligand'].pos for d in out]).to(dev)
    cg = getattr(cm, 'score_model', cm)
    batch = next(iter(DataLoader([from_arrays(c) for _ in range(B)], batch_size=B)))
    at = lambda k: cg.confidence(batch, pos.clone(), (float(sched[k]),) * 3).cpu()
    assert rel_err(conf.cpu(), at(steps - 1)) < 1e-6
    assert rel_err(conf.cpu(), at(24)) > 1e-4          # (the two times give different confidences: the test can see the difference)
