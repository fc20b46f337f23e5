"""Round-4 GPU tests (VERDICT r03 "Next" #4, #5), through the C ABI / the shipped entry points:

* parity at the BENCH's batch size: a 40-sample forward of the 300-residue workload (and 8 samples at 2000 residues) against the oracle,
  which runs in chunks of 4 samples (the samples of a batch are independent);
* RCCL on the one GPU of the box: a world_size-1 ``nccl`` process group beside a live ddk context, the three gathers of
  disco_diffdock_amd/distributed.py on device tensors, and ``bench.py --gpus 1 --force-dist`` (every barrier / all_reduce / all_gather of the
  N > 1 path on RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr
from helpers import batch_of, chan_err, elem_err, rel_err
from test_gpu_round2 import _poses
from test_gpu_round3 import _record_drift

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


@pytest.mark.parametrize('n_res,B,t', [(300, 40, 0.5), (2000, 8, 0.5)])
def test_bench_batch_size_oracle_parity(dev, tables, n_res, B, t):
    """The bench runs 40 samples per forward: more blocks in the conv kernel's work queue (column-split tail), more run tails per node and more
    32-edge tiles per receiving node than any B = 2 / 3 case.  tr / rot / tor and the ligand rows of a B-sample forward at t = 0.5 (pruning
    active, cross cutoff ~ 20 A) against oracle.score_model_ref run in chunks of 4 samples; per-element AND per-channel measures."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    c = synthetic.make_complex(5, n_res=n_res)
    P = smr.random_state_dict(CFG, seed=8)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    pos = _poses(c, B, np.random.default_rng(1), spread=5.0)
    cx = Complex(ctx, c, B)
    tr, rot, tor = [x.cpu() for x in cx.score_forward(T(pos).to(dev), t, t, t)]
    lig = cx.lig_node_features(B, dev).cpu()
    st = cx.graph_stats()
    R, n_lig = int(c['edge_mask'].sum()), c['lig_pos'].shape[0]
    tr_r, rot_r, tor_r, lig_r = [], [], [], []
    for lo in range(0, B, 4):
        b = batch_of(c, 4, pos[lo:lo + 4])
        spr.set_time(b, t, t, t, 4)
        a, bb, cc, inter = smr.score_model_forward(P, CFG, b, tables[0], tables[1], return_intermediates=True)
        tr_r.append(a); rot_r.append(bb); tor_r.append(cc); lig_r.append(inter['lig_node_attr'])
    tr_r, rot_r, tor_r, lig_r = [torch.cat(x) for x in (tr_r, rot_r, tor_r, lig_r)]
    assert tor.shape[0] == B * R and lig.shape[0] == B * n_lig
    def vec_err(a, b):      # tr / rot are one 3-vector per sample: |a - b| / |b| per SAMPLE (a component that vanishes by cancellation has no scale of its own)
        a, b = a.double().reshape(-1, 3), b.double().reshape(-1, 3)
        nb = b.norm(dim=1)
        return float(((a - b).norm(dim=1) / torch.clamp(nb, min=1e-3 * float(nb.max()))).max())
    errs = {'tr': vec_err(tr, tr_r), 'rot': vec_err(rot, rot_r), 'tor': elem_err(tor, tor_r), 'lig_chan': chan_err(lig, lig_r),
            'tr_global': rel_err(tr, tr_r), 'rot_global': rel_err(rot, rot_r)}
    info = {'tr_elem': elem_err(tr, tr_r), 'rot_elem': elem_err(rot, rot_r), 'lig_elem': elem_err(lig, lig_r)}      # per-element figures: recorded, not gated
    print(f'parity at the bench batch size: n_res={n_res} B={B} t={t}: {errs}  per element: {info}  graph: {st}')
    _record_drift(f'forward_B{B}_{n_res}_residues_t{t}_vs_oracle', max(errs.values()), bar=1e-4, **{k: float(v) for k, v in {**errs, **info}.items()})
    for k, v in errs.items():
        assert v < 1e-4, (k, errs)


_RCCL_SCRIPT = r'''
import os, sys, socket, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from disco_diffdock_amd import synthetic, distributed as dd
from disco_diffdock_amd.runtime import Context, Complex
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
ctx = Context(device=0)                                   # libddk.so loaded, a live ddk_ctx on the same device
ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
c = synthetic.make_complex(0, n_res=120)
cx = Complex(ctx, c, 4)
pos = torch.from_numpy(np.stack([c['lig_pos'] + i for i in range(4)]).astype(np.float32)).to(dev)
tr0 = cx.score_forward(pos, 0.6, 0.6, 0.6)[0].clone()
with socket.socket() as so:
    so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
dd.FORCE_COLLECTIVES = True                               # a world of one still runs every collective of the path
n_lig = [c['lig_pos'].shape[0], 7, 19]
poses = {i: torch.randn(4, n, 3, device=dev) for i, n in enumerate(n_lig)}
g = dd.gather_poses(poses, n_lig, 4, dev)
assert all(torch.equal(g[i], poses[i]) for i in poses)
conf = {i: torch.randn(4, device=dev) for i in range(3)}
gc = dd.gather_confidences(conf, 3, dev)
assert all(torch.equal(gc[i], conf[i]) for i in conf)
sl = torch.randn(4, n_lig[0], 3, device=dev)
gs = dd.gather_samples(sl, 4, 0, 1, dev)
assert torch.equal(gs, sl)
t = torch.ones(3, device=dev); dist.all_reduce(t); dist.barrier()
tr1 = cx.score_forward(pos, 0.6, 0.6, 0.6)[0]              # the ddk context still works beside the RCCL communicator
assert torch.allclose(tr0, tr1, rtol=1e-5, atol=1e-6)
torch.cuda.synchronize()
be = dist.get_backend()
dist.destroy_process_group()
print(json.dumps({'backend': be, 'nccl_version': list(torch.cuda.nccl.version()), 'gathers': ['gather_poses', 'gather_confidences', 'gather_samples'], 'ok': True}))
'''


def test_rccl_world_of_one_beside_a_live_context(dev):
    """backend='nccl' (RCCL on ROCm) initialised by THIS code on a GPU box, beside libddk.so and a live ddk_ctx (two HIP runtimes in one
    process was a real failure once: commit 41af4eb), and the path's three gathers executed on device tensors."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, '-c', _RCCL_SCRIPT, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['ok'] and out['backend'] == 'nccl'
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'rccl_world1.json'), 'w') as f:
        json.dump(out, f)
    print('RCCL world-of-one:', out)


def test_bench_force_dist_runs_the_multi_rank_path_on_rccl(dev):
    """``bench.py --gpus 1 --force-dist --backend nccl``: the barrier / all_reduce(max time) / all_reduce(n_lig) / all_gather of the N > 1 path
    execute on RCCL (config 4: poses AND confidences are gathered)."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--backend', 'nccl', '--config', '4', '--steps', '3',
                        '--warmup', '1', '--no-cpu-baseline', '--no-alt', '--no-extras', '--no-device-loop'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0
    with open(os.path.join(ROOT, 'gpurun_out', 'bench_force_dist_nccl.json'), 'w') as f:
        json.dump(d, f)


def test_confidence_layer1_sharing_equal_full_and_engaged(dev):
    """Confidence model, layer 1 (conf.hip): the static groups (atom-atom, atom<-residue, residue-residue, residue<-atom) are evaluated only into
    the receivers whose messages differ from the virtual ligand-free sample's - the receiver or one of its senders received a ligand message in
    layer 0 - and every other (receiver, group) reads the virtual sample's sum.  Must equal the evaluation in every sample
    (ddk_debug_set_layer0_dedup(0)) to the atomic-add order noise, and must actually drop work: a pose far from the receptor contributes no static
    edge at all, poses in the pocket fewer atom-atom edges than the full group."""
    from oracle import confidence_ref as cr
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    c = synthetic.make_complex(41, n_res=120, n_lig=18)
    synthetic.add_receptor_atoms(c, np.random.default_rng(41))
    ctx = Context(device=0, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
    ctx.load_state_dict(cr.random_state_dict(cr.ConfidenceModelConfig(), seed=9))
    B, Bm = 5, 7
    pos = _poses(c, B, np.random.default_rng(12), spread=3.0)
    pos[2] += 300.0           # one pose far away: none of its atoms / residues receives a ligand message
    pos = torch.as_tensor(pos).to(dev)
    cx = Complex(ctx, c, max_batch=Bm)
    cx.set_atoms(c['atom_x'], c['atom_pos'], c['atom_edge_index'], c['atom_rec_index'])
    n_atom, E_aa, E_rr = len(c['atom_x']), c['atom_edge_index'].shape[1], c['rec_edge_index'].shape[1]
    res = {}
    for on in (True, False):
        ctx.debug_set_layer0_dedup(on)
        conf = cx.confidence_forward(pos)
        res[on] = (conf.cpu(), cx.lig_node_features(B, dev).cpu())
        if on:
            full, t0, t1 = cx.confidence_table(0), cx.confidence_table(1), cx.confidence_table(4)
    ctx.debug_set_layer0_dedup(True)
    assert rel_err(res[True][0], res[False][0]) < 1e-5 and rel_err(res[True][1], res[False][1]) < 1e-5      # (atomic-add order noise ~1e-6)
    assert (full['aa'], full['ar'], full['rr'], full['ra']) == (B * E_aa, B * n_atom, B * E_rr, B * n_atom)
    assert (t0['aa'], t0['ar'], t0['rr'], t0['ra']) == (E_aa, n_atom, E_rr, n_atom)                       # layer 0: the virtual sample only
    for g in ('ll', 'lr', 'la', 'al', 'rl'):
        assert t0[g] == full[g] and t1[g] == full[g]                                                     # ligand-dependent groups: untouched
    # layer 1: the virtual sample + at most the four poses near the receptor; the atom-atom group (5 A neighbourhoods) drops most of its edges
    assert E_aa < t1['aa'] < E_aa + (B - 1) * E_aa * 0.9, (t1, E_aa)
    for g, n1 in (('ar', n_atom), ('rr', E_rr), ('ra', n_atom)):
        assert n1 <= t1[g] <= n1 + (B - 1) * n1, (g, t1)


def _cg_conf_model(dev, seed):
    """get_model(args, ..., confidence_mode=True) for a checkpoint WITHOUT all_atoms: the coarse-grained model in confidence_mode
    (utils/model_utils.py:25-68 -> models/score_model.py:110-121)."""
    from argparse import Namespace
    from functools import partial
    from test_gpu_model import ARGS_S
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.diffusion_utils import t_to_sigma
    cargs = Namespace(**{**vars(ARGS_S), 'rmsd_classification_cutoff': [2.0, 5.0]})
    cm = get_model(cargs, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True, confidence_mode=True)
    cfg = smr.ScoreModelConfig(latent_vocab=64, confidence_mode=True, num_confidence_outputs=3)
    cm.score_model.load_state_dict(smr.random_state_dict(cfg, seed=seed), strict=True)
    cm.eval()
    return cm, cargs


def test_cg_confidence_model_golden(dev, golden):
    """the coarse-grained confidence model on the device (ddk_config.confidence_mode, ddk_score_confidence) == the reference's own
    models/score_model.py in confidence_mode (golden produced through the reference's get_model on the stand-ins): confidences [B, 3] and the ligand
    rows after the conv stack, at complex_t = (0.3, 0.25, 0.2) used as sigmas (score_model.py:186-189)."""
    from helpers import complex_from_npz
    from disco_diffdock_amd.data import from_arrays, collate
    from disco_diffdock_amd.diffusion_utils import set_time
    z, c = golden('cg_confidence_model'), complex_from_npz(golden('complex_cg_confidence'))
    cm, _ = _cg_conf_model(dev, int(z['seed']))
    B = int(z['B'])
    b = collate([from_arrays(c) for _ in range(B)])
    b['ligand'].pos = torch.as_tensor(z['pos']).float().reshape(-1, 3)
    b = b.to(dev)
    set_time(b, *[float(x) for x in z['t']], B, False, dev)
    conf = cm(b)
    sm = cm.score_model
    lig = sm.last_complex.lig_node_features(B, dev)
    assert rel_err(lig.cpu(), z['lig_node_attr']) < 1e-4
    assert tuple(conf.shape) == (B, 3) and rel_err(conf.cpu(), z['confidence']) < 1e-4
    # a confidence_mode context has no score heads: the score entry points refuse it
    with pytest.raises(RuntimeError, match='confidence_mode'):
        sm.last_complex.score_forward(b['ligand'].pos.reshape(B, -1, 3), 0.3, 0.3, 0.3)
    # strict loading: the score model's head keys are unexpected, the predictor's are required
    bad = smr.random_state_dict(smr.ScoreModelConfig(latent_vocab=64), seed=1)
    with pytest.raises(RuntimeError):
        sm.load_state_dict(bad, strict=True)


def test_sampling_with_cg_confidence_golden(dev, golden):
    """utils/sampling.py:239-240: sampling(confidence_model=<coarse-grained model>, confidence_data_list=None) - DiffDock-S reverse diffusion followed
    by the confidence model on the score batch itself at the last step's times - against the reference's own sampling() output."""
    from functools import partial
    from helpers import complex_from_npz
    from test_gpu_model import ARGS_S, _ref_noise
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    z, c = golden('trajectory_cg_confidence'), complex_from_npz(golden('complex_cg_confidence'))
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    model.score_model.load_state_dict(smr.random_state_dict(smr.ScoreModelConfig(latent_vocab=64), seed=int(z['score_seed'])), strict=True)
    cm, cargs = _cg_conf_model(dev, int(z['conf_seed']))
    n = len(c['lig_pos'])
    B, steps = len(z['pos0']) // n, int(z['steps'])
    dl = [from_arrays(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = torch.from_numpy(z['pos0'][i * n:(i + 1) * n])
    sched = get_t_schedule(steps)
    noise = [_ref_noise(int(z['seed']), steps, B, int(c['edge_mask'].sum()))]
    out, conf = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B,
                         no_final_step_noise=True, use_latent=False, noise=noise, confidence_model=cm, confidence_data_list=None,
                         confidence_model_args=cargs, temp_sampling=1.0, temp_psi=0.0, temp_sigma_data=0.5)
    assert rel_err(torch.cat([d['ligand'].pos for d in out]).cpu(), z['pos_out']) < 1e-4
    assert tuple(conf.shape) == z['confidence'].shape and rel_err(conf.cpu(), z['confidence']) < 1e-4


def test_cutoff_boundary_pairs_give_consistent_cross_edge_groups(dev):
    """Pairs that sit within a few ulp of the cross cutoff: the counting kernel, the fill kernel's counts, its two write loops and the pair matrix of the
    mirrored features all test the same pair, and must agree - round 4 found them compiled with different fused-multiply-add contractions, so that a
    boundary pair was counted but not written and ONE slot of a cross-edge group kept stale device memory (a garbage node index; a GPU memory fault
    once in a few hundred complexes).  Geometry: all ligand atoms of a sample in one point, the residues on a sphere of radius cutoff * (1 +- k ulp)."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    ctx = Context(device=0)
    ctx.load_state_dict(synthetic.random_score_model_state_dict(seed=0))
    t = 0.37
    sigma = np.float32(0.1) ** np.float32(1.0 - t) * np.float32(19.0) ** np.float32(t)
    cut = np.float32(3.0) * np.float32(sigma) + np.float32(20.0)
    rng = np.random.default_rng(5)
    c = synthetic.make_complex(3, n_res=300, n_lig=24)
    u = rng.normal(size=(300, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    k = rng.integers(-6, 7, size=300)
    c['rec_pos'] = (u * (np.float64(cut) * (1.0 + k * 2.0 ** -24))[:, None]).astype(np.float32)
    B = 16
    cx = Complex(ctx, c, B)
    n_lig, n_bad, n_pairs = 24, 0, 0
    for trial in range(6):
        centre = rng.normal(0, 3e-6, size=(B, 1, 3))
        pos = torch.as_tensor(np.broadcast_to(centre, (B, n_lig, 3)).astype(np.float32).copy()).to(dev)
        ei, off = cx.build_graph(pos, t)
        ei, off = ei.cpu().numpy(), [int(v) for v in off]
        lr, rl = ei[:, off[1]:off[2]], ei[:, off[3]:off[4]]
        assert lr.shape[1] == rl.shape[1]
        N = B * (n_lig + 300)
        assert lr.min() >= 0 and lr.max() < N and rl.min() >= 0 and rl.max() < N          # every slot was written with a node of this batch
        a = set(map(tuple, lr.T.tolist()))
        b_ = set((d, s) for s, d in rl.T.tolist())
        n_bad += len(a ^ b_)
        n_pairs += lr.shape[1]
        assert len(a) == lr.shape[1] and len(b_) == rl.shape[1]                               # no slot written twice / left over
    assert 0 < n_pairs < 6 * B * n_lig * 300          # the sphere really straddles the cutoff
    assert n_bad == 0, f'{n_bad} cross edges without their flipped copy'
