"""Round-3 GPU tests (VERDICT r02 "Next" #1, #2, #5), all through the C ABI / the shipped entry points:

* ``bench.py --gpus N`` starts its N ranks itself and the JSON line carries the executed-work roofline fraction and both floors;
* parity at the size and length of the workload: a 20-step, 300-residue trajectory with unscaled injected noise against
  oracle.sampler_ref, the same trajectory with the receptive-field pruning on vs off, bit-compared in deterministic mode."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from oracle import sampler_ref as spr
from helpers import elem_err, rel_err, to_graph

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


def _record_drift(key, value, **extra):
    """parity figures that are printed are also KEPT (gpurun_out/parity_drift.json on the GPU box, copied to profiles/r0N_parity_drift.json per round):
    a regression from 2e-5 to 9e-4 under a 1e-3 bar must be visible"""
    path = os.path.join(ROOT, 'gpurun_out', 'parity_drift.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = {}
    d[key] = dict(value=float(value), **extra)
    with open(path, 'w') as f:
        json.dump(d, f, indent=1, sort_keys=True)


def _bench(*args):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus2_launches_two_ranks_by_itself(dev):
    """``python bench.py --gpus 2`` with no torchrun in the command runs two ranks (here: both on this box's one GPU over gloo)
    and rank 0 prints ONE line with n_gpus = 2; --gpus beyond the visible devices without --single-device fails loudly."""
    out = _bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-alt',
                 '--no-extras', '--no-device-loop')
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0
    if torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                           env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}, timeout=300)
        assert r.returncode != 0 and 'GPU(s) are visible' in r.stderr


def test_bench_line_reports_executed_work_and_both_floors(dev):
    """roofline.frac is the EXECUTED fraction (<= 1, = flop_per_launch / avg_launch_ms / peak), the pruning-off floor and the
    pocket-bound workload are in the line, and the per-step executed-edge fractions cover the 20 steps."""
    out = _bench('--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-alt', '--no-device-loop', '--no-timesplit')
    rf = out['roofline']
    # (a suite run with DDK_CONV_KERNEL=1 puts the fp32-MFMA kernel under the bench too: its roofline is priced against the fp32 matrix peak)
    peak = 157.3 if os.environ.get('DDK_CONV_KERNEL') == '1' else 2500.0
    assert 0.0 < rf['frac'] <= 1.0 and rf['peak'] == peak and rf['fp32_equivalent_TFLOPs'] > 0
    assert abs(rf['frac'] - rf['flop_per_launch'] / (rf['avg_launch_ms'] * 1e-3) / 1e12 / rf['peak']) < 1e-6 * rf['frac'] + 1e-9
    assert abs(rf['achieved'] - rf['frac'] * rf['peak']) < 1e-6 * rf['achieved']
    ex = out['extra']
    assert ex['pruning_off']['edges_executed_over_unpruned'] == pytest.approx(1.0)
    assert ex['pruning_off']['value'] > 0 and ex['pocket_bound']['value'] > 0
    assert out['value_pocket_bound'] == ex['pocket_bound']['value'] and out['value_pruning_off'] == ex['pruning_off']['value']      # (round 5: beside `value`)
    assert ex['pocket_bound']['min_cross_edges_per_sample_over_steps'] > 0          # the pocket-bound samples never lose contact
    assert ex['pocket_bound']['edges_executed_over_unpruned'] >= rf['edges_executed_over_unpruned'] - 1e-9
    assert len(ex['per_step']) == 20 and all(0 < s['edges_executed_over_unpruned'] <= 1 for s in ex['per_step'])
    # round 6: the median of three timed passes over the same calls (every pass with its own host data_lists: sampling() writes the final poses into the graphs it is
    # given, and a pass that inherited them would start from wandered-off ligands and run 1.5 - 2 x faster - the bug this guards against), the three figures and the
    # HBM-bound boundary-A kernel under keys the driver's record keeps
    hl = ex['headline']
    assert hl['passes'] == 3 and len(hl['pass_elapsed_s']) == 3 and max(hl['pass_elapsed_s']) / min(hl['pass_elapsed_s']) < 1.15, hl['pass_elapsed_s']
    assert out['ms_per_step'] == pytest.approx(1e3 * sorted(hl['pass_elapsed_s'])[1] / out['steps'], rel=1e-3)
    trio = out['config']['value_headline_pruning_off_pocket_bound']
    assert trio == [round(out['value'], 3), round(out['value_pruning_off'], 3), round(out['value_pocket_bound'], 3)]
    tp = rf['tp_boundary_A']
    assert tp is not None and tp['bound'] == 'hbm' and tp['unit'] == 'GB/s' and tp['layer'] == 3 and [p['layer'] for p in tp['per_layer']] == [0, 1, 2, 3]
    assert tp['achieved'] == tp['per_layer'][3]['GBps'] and tp['frac'] == pytest.approx(tp['achieved'] / 8000.0, abs=1e-3)
    assert all(p['GBps'] > 3000 for p in tp['per_layer']), tp['per_layer']      # (round 5's kernel: 3 100 - 4 300; the new one 4 700 - 5 500)


def test_bench_line_carries_the_other_limb_form_of_the_conv_kernel(dev):
    """Round 6: the default conv kernel multiplies two f16 limbs per operand (three products); the line also carries the SAME sampling() bracket with the three-limb /
    six-product form (ddk_config.conv_kernel = 3, the default of rounds 3 - 5) in every context - value_six_limb_products at the top level and under `config` (keys
    the driver's record keeps), the launch time and executed-MFMA fraction under roofline.other_limb_form - and the executed-work accounting follows the form that ran."""
    out = _bench('--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-device-loop', '--no-timesplit', '--no-tp-boundary')
    rf, kern = out['roofline'], int(os.environ.get('DDK_CONV_KERNEL', '0'))
    assert out['config']['conv_kernel'] == kern
    if kern == 1:
        assert rf['other_limb_form'] is None and out['value_six_limb_products'] is None
        return
    o = out['extra']['other_limb_form']
    assert rf['limb_products'] == (6 if kern == 3 else 3) and o['limb_products'] == (3 if kern == 3 else 6) and o['conv_kernel'] == (0 if kern == 3 else 3)
    assert rf['other_limb_form']['value'] == o['value'] and o['value'] > 0
    six_ms, two_ms = (rf['avg_launch_ms'], o['avg_launch_ms']) if kern == 3 else (o['avg_launch_ms'], rf['avg_launch_ms'])
    assert 1.15 < six_ms / two_ms < 1.65, (six_ms, two_ms)            # 27 vs 14 MFMAs per weight tile: measured 1.35 - 1.40
    if kern == 0:
        assert out['value_six_limb_products'] == o['value'] and out['config']['value_six_limb_products_conv_kernel_3'] == round(o['value'], 3)
    else:
        assert out['value_six_limb_products'] == out['value']
    assert o['frac_of_f16_matrix_peak'] == pytest.approx(o['mfma_TFLOPs_executed'] / 2500.0)
    assert out['fallback_fp32_kernel']['value'] > 0


# ------------------------------------------------------------------------------------------------------------------------------------------
# the exact three-limb f16 product of the default conv kernel (VERDICT r02 #3 i, ii)
# ------------------------------------------------------------------------------------------------------------------------------------------
def _adversarial_values(rng, n, binades=36):
    """fp32 values that stress the limb split: full 24-bit mantissas, values one ulp around powers of two and around fp16 rounding
    boundaries (hi rounds up / ties), mixed signs and magnitudes over `binades` binades below each group's maximum"""
    m = rng.integers(1 << 23, 1 << 24, size=n).astype(np.float64)              # every mantissa bit in play
    m[::7] = (1 << 23) + rng.integers(0, 3, size=m[::7].shape)                  # just above a power of two
    m[1::7] = (1 << 24) - 1 - rng.integers(0, 3, size=m[1::7].shape)            # just below
    m[2::7] = ((rng.integers(1 << 10, 1 << 11, size=m[2::7].shape) << 13) | (1 << 12)) + rng.integers(-1, 2, size=m[2::7].shape)   # hi ties
    m[3::7] = (rng.integers(1 << 10, 1 << 11, size=m[3::7].shape) << 13) | ((1 << 12) + (1 << 1) - 1) | (rng.integers(0, 2, size=m[3::7].shape) << 1)   # mid ties
    e = rng.integers(-binades, 1, size=n)
    sgn = rng.choice([-1.0, 1.0], size=n)
    return (sgn * m * np.exp2(e.astype(np.float64) - 23)).astype(np.float32)


@pytest.mark.parametrize('binades', [14, 36])
@pytest.mark.parametrize('scale', [1.0, 3e-9, 7e11])
def test_device_limb_split_is_exact(dev, scale, binades):
    """The in-kernel split (ddk_debug_split3 runs the kernel's own split3 / range_scale).  Round 4: the limbs carry their own weight
    (x * scale == hi + mid + lo, low limbs reach into the fp16 subnormals, which the f16 MFMA honours), so the split is exact BIT FOR BIT
    for every value whose last bit is a multiple of the fp16 subnormal step 2^-24 after scaling - |x * scale| >= 0.5, i.e. within 2^-15 of
    its group's maximum (round 3's 2^11 / 2^22 factors reached 2^-36) - and off by at most 2^-25 absolute = 2^-39 of the group's maximum below
    that, whatever the magnitude of the group (per-edge power-of-two scaling).  The limbs are fp16 values; the scale is a power of two that
    puts the group's maximum into [2^14, 2^15).  binades = 14: every value of every group is inside the exact window."""
    import ctypes as C
    from disco_diffdock_amd.tensor_layers import _shape_context
    ctx = _shape_context(0)
    rng = np.random.default_rng(17)
    group, n = 72, 72 * 4000
    x = _adversarial_values(rng, n, binades) * np.float32(scale)
    xs = torch.from_numpy(x).to(dev)
    hi, mid, lo, sc = [torch.empty(n, device=dev) for _ in range(4)]
    p = lambda t: C.c_void_p(t.data_ptr())
    ctx._check(ctx.L.ddk_debug_split3(ctx.h, p(xs), n, group, p(hi), p(mid), p(lo), p(sc), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'split3')
    hi, mid, lo, sc = [t.cpu().numpy().astype(np.float64) for t in (hi, mid, lo, sc)]
    gmax = np.abs(x.astype(np.float64)).reshape(-1, group).max(axis=1).repeat(group)
    assert np.all(np.log2(sc) == np.round(np.log2(sc)))                                  # powers of two
    assert np.all((gmax * sc >= 2.0 ** 14) & (gmax * sc < 2.0 ** 15))
    for limb in (hi, mid, lo):                                                            # fp16-representable (subnormals included)
        assert np.array_equal(limb.astype(np.float16).astype(np.float64), limb)
    want = x.astype(np.float64) * sc
    back = hi + mid + lo
    big = np.abs(want) >= 0.5
    assert big.mean() > (0.999 if binades == 14 else 0.35)
    assert np.array_equal(back[big], want[big])                                           # exact: not one bit dropped
    assert np.abs(back[~big] - want[~big]).max(initial=0.0) <= 2.0 ** -25                 # <= 2^-39 of the group's maximum


def test_three_limb_product_at_least_as_accurate_as_fp32_chain(dev):
    """(ii): one conv layer at W = 1872 on 20k edges against the fp64 oracle: the three-limb f16 kernel's error (ddk_config.conv_kernel = 3 since round 6) is not
    above the fp32-MFMA kernel's (products are exact and only three of nine limb products, <= 3 * 2^-33, are dropped; both accumulate in fp32).  The default
    two-limb form has its own test (test_gpu_round6.py::test_two_limb_kernel_is_fp32_grade)."""
    from disco_diffdock_amd.runtime import Context
    from test_gpu_ops import _random_case, CFG as OCFG
    l, N, splits = 3, 1000, [0, 3000, 9000, 15000, 20000]
    i_irr, o_irr = OCFG.conv_irreps(l)
    Pl = smr.random_conv_layer_params(OCFG, l, 123, True)
    node, ei, ea, sh = _random_case(l, N, splits, 5, True)
    P = {'L.' + k: v.double() for k, v in Pl.items()}
    ref = smr.tp_conv_layer(P, 'L', node.double(), ei, [ea.double()[splits[i]:splits[i + 1]] for i in range(4)], sh.double(),
                            i_irr, '1x0e+1x1o', o_irr, residual=True, batch_norm=True, faster=True, edge_groups=4)
    args = (l, node.to(dev), ei[0].to(dev), ei[1].to(dev), splits, ea.to(dev), sh.to(dev), smr.irreps_dim(o_irr))
    err = {}
    for kernel in (3, 1):
        ctx = Context(device=0, conv_kernel=kernel)
        ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in Pl.items()})
        out = ctx.conv_forward(*args).cpu()
        err[kernel] = (rel_err(out, ref), elem_err(out, ref))
    print(f'conv layer vs fp64: three-limb f16 {err[3]}, fp32 MFMA {err[1]}')
    assert err[3][0] < 1e-5 and err[3][0] <= 1.25 * err[1][0] + 1e-7 and err[3][1] <= 1.25 * err[1][1] + 1e-6, err


# ------------------------------------------------------------------------------------------------------------------------------------------
# the core parity tests once more under every (conv kernel, scatter) mode, inside the driver's single `pytest -m gpu` (VERDICT r02 #5d)
# ------------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(params=[(0, 0), (0, 1), (1, 0), (1, 1), (3, 0), (3, 1)],
                ids=['x2-atomics', 'x2-deterministic', 'fp32-atomics', 'fp32-deterministic', 'x3-atomics', 'x3-deterministic'])
def mode(request, monkeypatch):
    """every ddk context created inside the test runs the given conv kernel / scatter mode (runtime.Context reads the switches)"""
    kernel, det = request.param
    monkeypatch.setenv('DDK_CONV_KERNEL', str(kernel))
    monkeypatch.setenv('DDK_DETERMINISTIC', str(det))
    return request.param


def test_core_parity_in_every_mode(dev, golden, tables, mode):
    """conv-layer goldens (Tier A, unmodified reference code), score-model goldens at three diffusion times, the reference-produced
    trajectory and a full-size (300 residues) forward against the oracle - in the mode of the fixture."""
    from functools import partial
    from helpers import complex_from_npz, batch_of, chan_err
    from test_gpu_model import ARGS_S, README_S, _dev_batch, _ref_noise
    from test_gpu_ops import CFG as OCFG
    from disco_diffdock_amd.tensor_layers import TensorProductConvLayer
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    for l in range(5):
        for bn in (0, 1):
            z = golden(f'conv_layer_l{l}_bn{bn}')
            i_irr, o_irr = OCFG.conv_irreps(l)
            layer = TensorProductConvLayer(i_irr, '1x0e+1x1o', o_irr, 72, hidden_features=72, residual=True, batch_norm=bool(bn), dropout=0.1,
                                           faster=True, edge_groups=4).eval()
            layer.load_state_dict(smr.random_conv_layer_params(OCFG, l, int(z['param_seed']), bool(bn)), strict=True)
            s = z['splits']
            ea = T(z['edge_attr']).to(dev)
            out = layer(T(z['node']).to(dev), T(z['edge_index']).to(dev), [ea[s[i]:s[i + 1]] for i in range(4)], T(z['sh']).to(dev)).cpu()
            assert int(layer._ctx.cfg.conv_kernel) == mode[0]
            assert rel_err(out, z['out']) < 1e-5, (l, bn)
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    sm = model.score_model
    assert (int(sm.ctx.cfg.conv_kernel), int(sm.ctx.cfg.deterministic)) == mode
    sm.load_state_dict(smr.random_state_dict(CFG, seed=7), strict=True)
    tag = 'diffdockS_score_model'
    c = complex_from_npz(golden(f'complex_{tag}'))
    for t in (1.0, 0.55, 0.05):
        z = golden(f'score_{tag}_t{t}')
        B = int(z['B'])
        tr, rot, tor = sm(_dev_batch(c, B, z['pos'], dev, t), keep_receptor_features=True)
        lig, rec = sm.last_complex.node_features(B, dev)
        assert chan_err(lig.cpu(), z['lig_node_attr']) < 1e-4 and chan_err(rec.cpu(), z['rec_node_attr']) < 1e-4
        for name, a in (('tr', tr), ('rot', rot), ('tor', tor)):
            assert elem_err(a.cpu(), z[name]) < 1e-4, (name, t)
    z = golden(f'trajectory_{tag}')
    B, steps, n = 2, int(z['steps']), len(c['lig_pos'])
    dl = [from_arrays(c) for _ in range(B)]
    for i, d in enumerate(dl):
        d['ligand'].pos = T(z['pos0'][i * n:(i + 1) * n])
    sched = get_t_schedule(steps)
    noise = [_ref_noise(int(z['seed']), steps, B, int(c['edge_mask'].sum()))]
    out, _ = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=B, no_final_step_noise=True,
                      use_latent=False, noise=noise, **README_S)
    assert rel_err(torch.cat([d['ligand'].pos for d in out]).cpu(), z['pos_out']) < 1e-4
    # full size: 300 residues, B = 2, small t (few cross edges: pruning active), scores and ligand rows against the oracle
    cc = synthetic.make_complex(7, n_res=300)
    P = smr.random_state_dict(CFG, seed=5)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    rng = np.random.default_rng(3)
    pos = np.stack([cc['lig_pos'] + rng.normal(0, 4.0, size=(1, 3)) for _ in range(2)]).astype(np.float32)
    cx = Complex(ctx, cc, 2)
    tr, rot, tor = cx.score_forward(T(pos).to(dev), 0.05, 0.05, 0.05)
    lig = cx.lig_node_features(2, dev).cpu()
    b = batch_of(cc, 2, pos)
    spr.set_time(b, 0.05, 0.05, 0.05, 2)
    tr_r, rot_r, tor_r, inter = smr.score_model_forward(P, CFG, b, tables[0], tables[1], return_intermediates=True)
    for name, a, r in (('tr', tr, tr_r), ('rot', rot, rot_r), ('tor', tor, tor_r)):
        assert elem_err(a.cpu(), r) < 1e-4, name
    assert chan_err(lig, inter['lig_node_attr']) < 1e-4


# ------------------------------------------------------------------------------------------------------------------------------------------
# parity at the size and length of the workload (VERDICT r02 #5a, b)
# ------------------------------------------------------------------------------------------------------------------------------------------
def _workload_trajectory(dev, ctx_kwargs, prune=True, seed=11, B=2):
    """20 reverse steps of B samples of a 300-residue complex with the README low-temperature coefficients and UNSCALED injected noise"""
    from functools import partial
    from argparse import Namespace
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    from disco_diffdock_amd.sampling import step_coefficients
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from test_gpu_model import README_S
    args = Namespace(tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03, tor_sigma_max=3.14, no_torsion=False)
    c = synthetic.make_complex(seed, n_res=300)
    P = smr.random_state_dict(CFG, seed=21)
    ctx = Context(device=0, **ctx_kwargs)
    ctx.load_state_dict(P)
    ctx.set_pruning(prune)
    steps = 20
    cx = Complex(ctx, c, B)
    sched = get_t_schedule(steps)
    coeffs = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=args), args, False, False, True, README_S['temp_sampling'],
                               README_S['temp_psi'], README_S['temp_sigma_data'])
    rng = np.random.default_rng(4)
    pos0 = np.stack([c['lig_pos'] + rng.normal(0, 6.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    z = torch.randn(steps, B, 6 + cx.R, generator=torch.Generator().manual_seed(9))
    pos = T(pos0.copy()).to(dev)
    cx.sample(pos, *coeffs, z.to(dev))
    st = cx.graph_stats()
    return c, P, pos0, z, pos.cpu(), st, sched


def test_workload_size_trajectory_vs_oracle(dev, tables):
    """(a) BASELINE config 2's shape and length - 300 residues, 20 reverse steps, unscaled N(0,1) noise, B = 2 - against
    oracle.sampler_ref (about a minute of oracle time).  Poses within 1e-3 of the receptor scale after 20 chaotic steps; the drift is printed."""
    from test_gpu_model import README_S
    c, P, pos0, z, pos, st, sched = _workload_trajectory(dev, {})
    dl = []
    for p in pos0:
        g = to_graph(c)
        g['ligand'].pos = T(p)
        dl.append(g)
    nf = lambda b, t, name, shape: {'tr': z[t, :, 0:3], 'rot': z[t, :, 3:6], 'tor': z[t, :, 6:].reshape(-1)}[name]
    ref, _ = spr.sampling(dl, P, CFG, tables[0], tables[1], 20, sched, sched, sched, noise_fn=nf, batch_size=len(dl), no_final_step_noise=True, **README_S)
    r = torch.cat([g['ligand'].pos for g in ref])
    err = rel_err(pos.reshape(-1, 3), r)
    print(f'20-step, 300-residue, unscaled-noise trajectory drift vs oracle: {err:.2e} (last graph: {st})')
    _record_drift('trajectory_20_steps_300_residues_B2_vs_oracle', err, bar=1e-3)
    assert err < 1e-3


@pytest.mark.parametrize('kernel', [0, 1, 3])
def test_pruned_trajectory_equals_unpruned_over_twenty_steps(dev, kernel):
    """(b) the receptive-field pruning over 20 ACCUMULATING steps at workload size, deterministic scatter on both sides (no atomics: what
    differs is only which dead messages are evaluated and where the 32-edge tile boundaries fall inside the re-ordered rec-rec group, i.e.
    the association of a node's partial sums, amplified by 20 chaotic steps).  Final poses within the north star's 1e-4 per element;
    pruning must have dropped edges."""
    a = _workload_trajectory(dev, dict(deterministic=1, conv_kernel=kernel), prune=True)
    b = _workload_trajectory(dev, dict(deterministic=1, conv_kernel=kernel), prune=False)
    a2 = _workload_trajectory(dev, dict(deterministic=1, conv_kernel=kernel), prune=True)
    assert torch.equal(a[4], a2[4])                                   # deterministic: the same path twice is bit-identical
    E_rr = a[0]['rec_edge_index'].shape[1] * 2
    assert a[5]['E_rr_live'][0] < E_rr and b[5]['E_rr_live'][0] == E_rr      # the last step's level-A segment: pruned vs everything
    err = elem_err(a[4], b[4], floor=1e-2)
    print(f'pruned vs unpruned 20-step trajectory (kernel {kernel}): max element error {err:.2e}, bitwise equal: {torch.equal(a[4], b[4])}')
    assert err < 1e-4


@pytest.mark.parametrize('t', [1.0, 0.2])
def test_build_graph_large_shapes_vs_oracle(dev, t):
    """The graph kernels on shapes the workloads do not reach: 100 ligand atoms (two 64-lane chunks in every wave-per-atom / wave-per-residue
    loop of csrc/k_graph.hip) and 3000 residues (the workgroup's LDS beyond the default 64 KB, the static edge list beyond the register-held
    8192) against the oracle's graph builders (score_model.py:310-408): the edge multiset of every group, group order, every group sorted
    by the receiving node."""
    from collections import Counter
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    from helpers import batch_of
    c = synthetic.make_complex(5, n_res=3000, n_lig=100)
    assert c['lig_pos'].shape[0] == 100 and c['rec_pos'].shape[0] == 3000
    ctx = Context(device=0)
    ctx.load_state_dict(smr.random_state_dict(CFG, seed=2))
    B = 2
    rng = np.random.default_rng(3)
    pos = np.stack([c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    cx = Complex(ctx, c, B)
    ei, off = cx.build_graph(T(pos).to(dev), t)
    ei, off = ei.cpu().long(), [int(v) for v in off]
    b = batch_of(c, B, pos)
    spr.set_time(b, t, t, t, B)
    dt = torch.float32
    _, lig_ei, _, _, lig_sig = smr.build_lig_conv_graph(b, CFG, None, dt)
    _, rec_ei, _, _ = smr.build_rec_conv_graph(b, CFG, None, dt)
    tr_sigma = smr.t_to_sigma(*[b.complex_t[k] for k in ('tr', 'rot', 'tor')], CFG)[0]
    lr_ei, _, _ = smr.build_cross_conv_graph(b, CFG, (tr_sigma * 3 + 20).unsqueeze(1).to(dt), lig_sig, None, dt)
    n_l = B * 100
    lr = torch.stack([lr_ei[0], lr_ei[1] + n_l])
    want = [lig_ei, lr, rec_ei + n_l, torch.flip(lr, dims=[0])]
    assert off[0] == 0 and off[4] == ei.shape[1] == sum(w.shape[1] for w in want)
    assert lr.shape[1] > 64 * 100                       # cross edges well beyond one 64-lane chunk per atom
    for k in range(4):
        sl = ei[:, off[k]:off[k + 1]]
        assert sl.shape[1] == want[k].shape[1], k
        assert Counter(zip(sl[0].tolist(), sl[1].tolist())) == Counter(zip(want[k][0].tolist(), want[k][1].tolist())), k
        assert bool((sl[0, 1:] >= sl[0, :-1]).all()), k


def test_compressed_ligand_keeps_33_neighbours_within_capacity(dev, tables):
    """radius_graph(max_num_neighbors=32) is radius(..., 33) minus the self loop (score_model.py:315): an atom with 33 lower-index atoms inside
    5 A keeps all 33.  A compressed 65-atom ligand in a batch of 3 used to exceed the edge capacity that assumed 32 per atom (found by
    tests/devtools/fuzz_parity.py's boundary sweep); scores against the oracle at the north-star bar."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    from helpers import batch_of
    seed, n_res, n_lig, B, t = 633457205, 63, 65, 3, 0.5
    c = synthetic.make_complex(seed % 100000, n_res=n_res, n_lig=n_lig)
    P = smr.random_state_dict(CFG, seed=seed % 1000)
    base = c['lig_pos'].astype(np.float64)
    cen = base.mean(0, keepdims=True)
    pos = np.stack([cen + 0.35 * (base - cen) for _ in range(B)]).astype(np.float32)
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    cx = Complex(ctx, c, B)
    tr, rot, tor = cx.score_forward(T(pos).to(dev), t, t, t)
    st = cx.graph_stats()
    assert st['E_ll'] > B * (cx.M + 32 * n_lig)          # more than 32 kept neighbours per atom on average: the old bound
    assert st['E'] + st['E_shared'] <= st['cap']
    bt = batch_of(c, B, pos)
    spr.set_time(bt, t, t, t, B)
    tr_r, rot_r, tor_r = smr.score_model_forward(P, CFG, bt, tables[0], tables[1])
    assert rel_err(tr.cpu(), tr_r) < 1e-4 and rel_err(rot.cpu(), rot_r) < 1e-4 and rel_err(tor.cpu(), tor_r) < 1e-4


def test_se3_update_many_rotors_vs_oracle(dev):
    """modify_conformer_batch (diffusion_utils.py:37-55, torsion.py:71-86, geometry.py:126-156) on a 200-atom chain with more than 64 rotatable
    bonds: the device update fetches its rotor table in chunks of 64 (csrc/k_se3.hip) and runs Horn's closed form for the Kabsch step;
    against the oracle's sequential loop + SVD."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    from helpers import batch_of
    c = synthetic.make_complex(21, n_res=30, n_lig=200)
    R = int(c['mask_rotate'].shape[0])
    assert R > 64 and c['lig_pos'].shape[0] == 200
    ctx = Context(device=0)
    ctx.load_state_dict(smr.random_state_dict(CFG, seed=1))
    B = 3
    cx = Complex(ctx, c, B)
    g = torch.Generator().manual_seed(5)
    pos = np.stack([c['lig_pos'] + k for k in range(B)]).astype(np.float32)
    tr, rot = 0.5 * torch.randn(B, 3, generator=g), 0.3 * torch.randn(B, 3, generator=g)
    tor = 0.4 * torch.randn(B * R, generator=g)
    out = cx.se3_update(T(pos).to(dev), tr.to(dev), rot.to(dev), tor.to(dev)).cpu().reshape(-1, 3)
    b = batch_of(c, B, pos)
    ref = spr.modify_conformer_batch(T(pos).reshape(-1, 3), b, tr, rot, tor, T(c['mask_rotate']))
    assert float((out - ref).abs().max()) < 2e-4 * float(ref.abs().max())          # 150+ chained rotations in fp32 on both sides
