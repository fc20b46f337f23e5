"""Round-6 GPU tests (VERDICT r05 "Next" #3, #4, #6), through the C ABI / the shipped entry points:

* an out-of-memory inside ddk_complex_create / an operator workspace surfaces as RuntimeError with ddk_last_error, leaks nothing, and the SAME context then
  completes half the batch with the scores / poses of an uncapped run (the reference's recovery path, evaluate.py:228-231, 394-398);
* the column-owner FasterTensorProduct kernel: every lane / phase combination of the four layer shapes against the fp64 restatement, inputs that make each
  row operand class visible on its own;
* deterministic mode: the samples of a batch do not depend on which other samples share the batch."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from helpers import chan_err, rel_err

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


def _sampler_inputs(c, B, steps=3):
    from functools import partial
    from argparse import Namespace
    from disco_diffdock_amd.sampling import step_coefficients
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from test_gpu_model import README_S
    args = Namespace(tr_sigma_min=0.1, tr_sigma_max=19.0, rot_sigma_min=0.03, rot_sigma_max=1.55, tor_sigma_min=0.03, tor_sigma_max=3.14, no_torsion=False)
    sched = get_t_schedule(20)[:steps]
    coeffs = step_coefficients(steps, sched, sched, sched, partial(t_to_sigma, args=args), args, False, False, True, README_S['temp_sampling'],
                               README_S['temp_psi'], README_S['temp_sigma_data'])
    rng = np.random.default_rng(4)
    pos0 = np.stack([c['lig_pos'] + rng.normal(0, 4.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    return coeffs, pos0


def test_oom_surfaces_and_half_batch_retry_works(dev):
    """VERDICT r05 #3.  The reference halves the batch on ANY exception and retries up to three times (evaluate.py:228-231, 394-398).  Here every byte a
    batch needs is reserved by ddk_complex_create (ddk_sample / ddk_score_forward never allocate: DESIGN.md 2), so the out-of-memory point of the sampling
    path is that call; the operator entry point ddk_conv_forward grows a workspace and is the second point.  Under ddk_debug_set_alloc_limit:
      * a 2 000-residue complex at B = 40 fails with RuntimeError carrying ddk_last_error ('out of device memory ... retry with a smaller batch');
      * nothing leaks: device bytes held / chunks owned before == after, no chunk left owned;
      * the SAME context then runs B = 20 (twice: the two halves) with scores and 3-step poses equal to an uncapped context's B = 20 run;
      * the pool's parked chunks are handed back before giving up (a failed create after a destroyed complex succeeds once the parked chunk is evicted);
      * ddk_conv_forward: workspace growth beyond the cap -> DDK_ERR_NOMEM, a smaller call on the same context still answers correctly."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    det = bool(int(os.environ.get('DDK_DETERMINISTIC', '0')))
    c = synthetic.make_complex(5, n_res=2000)
    P = smr.random_state_dict(CFG, seed=3)
    B = 40
    coeffs, pos0 = _sampler_inputs(c, B)
    z = torch.randn(3, B, 6 + int(np.asarray(c['mask_rotate']).reshape(-1, len(c['lig_x'])).shape[0]), generator=torch.Generator().manual_seed(2))

    # ---- uncapped reference: B = 20 on the two halves; the sizes of a B = 20 and a B = 40 complex ----
    ctx0 = Context(device=0)
    ctx0.load_state_dict(P)
    base0 = ctx0.pool_stats()['device_bytes_held']
    cx20 = Complex(ctx0, c, 20)
    s20 = ctx0.pool_stats()['bytes_owned']
    ref = []
    for h in range(2):
        pos = T(pos0[20 * h:20 * h + 20].copy()).to(dev)
        tr, rot, tor = cx20.score_forward(pos, 0.7, 0.7, 0.7)
        cx20.sample(pos, *coeffs, z[:, 20 * h:20 * h + 20].contiguous().to(dev))
        ref.append((tr.cpu(), rot.cpu(), tor.cpu(), pos.cpu()))
    cx40 = Complex(ctx0, c, 40)
    s40 = ctx0.pool_stats()['bytes_owned'] - s20
    cx20.close(); cx40.close()
    assert s40 > s20 > 0, (s20, s40)      # (chunk size classes are {2^k, 1.5 x 2^k}: a 40-sample complex takes at least 1.5 x the chunk of a 20-sample one)
    del ctx0

    # ---- capped context: B = 40 does not fit, B = 20 does ----
    ctx = Context(device=0)
    ctx.load_state_dict(P)
    st0 = ctx.pool_stats()
    assert st0['device_bytes_held'] == base0 and st0['bytes_owned'] == 0
    ctx.debug_set_alloc_limit(st0['device_bytes_held'] + (s20 + s40) // 2)
    with pytest.raises(RuntimeError) as ei:
        Complex(ctx, c, 40)
    msg = str(ei.value)
    assert 'ddk_complex_create' in msg and 'out of device memory' in msg and 'smaller batch' in msg, msg
    st1 = ctx.pool_stats()
    assert st1 == st0, (st0, st1)                                # nothing allocated, nothing parked, nothing owned: no leak
    cx = Complex(ctx, c, 20)                                      # the retry with half the batch, same context
    for h in range(2):
        pos = T(pos0[20 * h:20 * h + 20].copy()).to(dev)
        tr, rot, tor = cx.score_forward(pos, 0.7, 0.7, 0.7)
        cx.sample(pos, *coeffs, z[:, 20 * h:20 * h + 20].contiguous().to(dev))
        got = (tr.cpu(), rot.cpu(), tor.cpu(), pos.cpu())
        for name, a, b in zip(('tr', 'rot', 'tor', 'pos'), got, ref[h]):
            if det:
                assert torch.equal(a, b), (name, h)
            else:
                assert rel_err(a, b) < 2e-5, (name, h, rel_err(a, b))      # (fp32 atomics: run-to-run noise ~1e-6, three reverse steps)
    # ---- a parked chunk is the context's own slack: it is handed back before the library gives up ----
    cx.close()                                                    # its chunk is parked in the pool now
    st2 = ctx.pool_stats()
    assert st2['bytes_owned'] == 0 and st2['chunks_parked'] >= 1
    ctx.debug_set_alloc_limit(st0['device_bytes_held'] + s40 + (1 << 20))     # room for ONE B = 40 complex, but only without the parked B = 20 chunk
    cx = Complex(ctx, c, 40)
    st3 = ctx.pool_stats()
    assert st3['hipFree_calls'] > st2['hipFree_calls'] and st3['chunks_parked'] == 0, (st2, st3)
    pos = T(pos0.copy()).to(dev)
    tr, rot, tor = cx.score_forward(pos, 0.7, 0.7, 0.7)
    assert rel_err(tr.cpu()[:20], ref[0][0]) < 2e-5 and rel_err(tor.cpu().reshape(B, -1)[20:].reshape(-1), ref[1][2]) < 2e-5
    cx.close()
    ctx.debug_set_alloc_limit(0)

    # ---- the operator boundary: ddk_conv_forward grows a workspace of the context ----
    from disco_diffdock_amd.tensor_layers import TensorProductConvLayer
    i_irr, o_irr = CFG.conv_irreps(3)
    layer = TensorProductConvLayer(i_irr, '1x0e+1x1o', o_irr, 72, hidden_features=72, residual=True, batch_norm=False, dropout=0.0, faster=True, edge_groups=4).eval()
    layer.load_state_dict(smr.random_conv_layer_params(CFG, 3, 5, False), strict=True)
    g = torch.Generator().manual_seed(1)

    def conv_inputs(N, E):
        node = torch.randn(N, 84, generator=g).to(dev)
        ei = torch.randint(0, N, (2, E), generator=g)
        ei = ei[:, torch.argsort(ei[0], stable=True)].to(dev)
        ea = torch.randn(E, 72, generator=g).to(dev)
        sh = torch.randn(E, 4, generator=g).to(dev)
        q = E // 4
        return node, ei, [ea[i * q:(i + 1) * q] for i in range(4)], sh

    small = conv_inputs(2000, 8000)
    want = layer(*small).cpu()                                    # (sizes the workspace for N = 2 000)
    lctx = layer._ctx
    lctx.debug_set_alloc_limit(lctx.pool_stats()['device_bytes_held'] + (8 << 20))
    big = conv_inputs(400000, 8000)                               # 3 x 400 000 x 84 x 4 B of workspace = 400 MB >> 8 MB
    with pytest.raises(RuntimeError) as ei2:
        layer(*big)
    assert 'out of device memory' in str(ei2.value), str(ei2.value)
    again = layer(*small).cpu()                                   # the context survives and re-grows what the failed call had released
    assert rel_err(again, want) < 1e-5
    lctx.debug_set_alloc_limit(0)


def test_sampling_retries_with_half_the_batch_like_evaluate_py(dev, tables):
    """The same recovery at the reference's call surface: the loop of evaluate.py:221-231 / 394-398 - call sampling(), on an exception halve batch_size and try
    again - written around disco_diffdock_amd.sampling.sampling with an allocation cap that a 40-sample batch of a 2 000-residue complex exceeds."""
    from functools import partial
    from argparse import Namespace
    import copy
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.model_utils import get_model
    from disco_diffdock_amd.sampling import sampling
    from disco_diffdock_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from disco_diffdock_amd.data import from_arrays
    from disco_diffdock_amd import score_model as smod
    from test_gpu_model import ARGS_S, README_S
    model = get_model(ARGS_S, dev, partial(t_to_sigma, args=ARGS_S), no_parallel=True)
    sm = model.score_model
    sm.load_state_dict(smr.random_state_dict(CFG, seed=7), strict=True)
    c = synthetic.make_complex(6, n_res=2000)
    N = 40
    rng = np.random.default_rng(0)
    graphs = []
    for i in range(N):
        g = from_arrays(c)
        g['ligand'].pos = T((c['lig_pos'] + rng.normal(0, 3.0, size=(1, 3))).astype(np.float32))
        graphs.append(g)
    steps = 2
    sched = get_t_schedule(20)
    R = int(np.asarray(c['mask_rotate']).reshape(-1, len(c['lig_x'])).shape[0])
    zfull = torch.randn(steps, N, 6 + R, generator=torch.Generator().manual_seed(3))

    def run(batch_size):
        dl = [copy.copy(g) for g in graphs]
        for d_, g in zip(dl, graphs):
            d_['ligand'].pos = g['ligand'].pos.clone()
        noise = [zfull[:, k:k + batch_size].contiguous() for k in range(0, N, batch_size)]
        out, _ = sampling(dl, model, steps, sched, sched, sched, dev, partial(t_to_sigma, args=ARGS_S), ARGS_S, batch_size=batch_size, no_final_step_noise=True,
                          noise=noise, **README_S)
        return torch.stack([d_['ligand'].pos.cpu() for d_ in out])

    want = run(20)                                                # uncapped, two batches of 20
    smod._complex_cache.clear()
    import gc
    gc.collect()
    st = sm.ctx.pool_stats()
    # the cap: weights and workspaces + 1.25 x the chunk of a 20-sample complex (chunk size classes are {2^k, 1.5 x 2^k}: 40 samples need >= 1.5 x that
    # chunk).  The 20-sample chunk of the run above is parked in the pool (or still owned by the complex cache): the library hands parked chunks back
    # before it gives up, so the failing call evicts it and the retry allocates afresh under the cap
    one20 = st['bytes_parked'] + st['bytes_owned']
    assert one20 > 0
    sm.ctx.debug_set_alloc_limit(st['device_bytes_held'] + one20 // 4)
    batch_size, tries, got = 40, 0, None
    failures = []
    while tries < 3 and got is None:                              # evaluate.py:394-398
        try:
            got = run(batch_size)
        except Exception as e:                                    # noqa: BLE001  (the reference catches everything)
            failures.append(str(e))
            batch_size //= 2
            tries += 1
    sm.ctx.debug_set_alloc_limit(0)
    assert got is not None and batch_size == 20 and len(failures) == 1 and 'out of device memory' in failures[0], (batch_size, failures)
    det = bool(int(os.environ.get('DDK_DETERMINISTIC', '0')))
    assert torch.equal(got, want) if det else rel_err(got.reshape(-1, 3), want.reshape(-1, 3)) < 2e-5


@pytest.mark.parametrize('l', range(5))
def test_tp_column_kernel_operand_classes_and_tails(dev, l):
    """tp_col_kernel (k_tp.hip, round 6): every row-operand class on its own - only the 0e inputs non-zero, only 1o, only 1e, only 0o, sh = (s0, 0) and sh = (0, v) -
    so that a wrong row offset, a swapped phase or a sign in one of the cross products cannot hide behind the other terms; edge counts around the persistent
    grid (1, 63, 4095, 4096, 4097, 2 x 4096 + 5: first row only / no second row in flight / exactly one trip / tail of one).  fp64 restatement, per block of
    output columns, 1e-6."""
    from disco_diffdock_amd.tensor_layers import FasterTensorProduct
    i_irr, o_irr = CFG.conv_irreps(l)
    tp = FasterTensorProduct(i_irr, '1x0e+1x1o', o_irr)
    din = smr.irreps_dim(i_irr)
    mul = {'0e': 0, '1o': 0, '1e': 0, '0o': 0}
    for part in i_irr.split('+'):
        m, ir = part.strip().split('x')
        mul[ir] = int(m)
    spans, o = {}, 0
    for ir, d in (('0e', 1), ('1o', 3), ('1e', 3), ('0o', 1)):
        spans[ir] = (o, o + d * mul[ir])
        o += d * mul[ir]
    assert o == din
    g = torch.Generator().manual_seed(500 + l)
    for E in (1, 63, 4095, 4096, 4097, 2 * 4096 + 5):
        x = torch.randn(E, din, generator=g)
        sh = torch.randn(E, 4, generator=g)
        w = torch.randn(E, tp.weight_numel, generator=g)
        cases = [('all', x, sh)]
        if E == 4097:
            for ir, (a, b) in spans.items():
                if b > a:
                    xm = torch.zeros_like(x)
                    xm[:, a:b] = x[:, a:b]
                    cases.append((f'only {ir}', xm, sh))
            s_only, v_only = sh.clone(), sh.clone()
            s_only[:, 1:] = 0
            v_only[:, 0] = 0
            cases += [('sh = (s0, 0)', x, s_only), ('sh = (0, v)', x, v_only)]
        for name, xx, ss in cases:
            out = tp(xx.to(dev), ss.to(dev), w.to(dev)).cpu()
            ref = smr.faster_tensor_product(xx.double(), ss.double(), w.double(), i_irr, o_irr)
            scale = float(ref.abs().max()) + 1e-30
            assert float((out.double() - ref).abs().max()) / scale < 1e-6, (l, E, name)
            if name.startswith('only') or name.startswith('sh'):      # ... and what must vanish does: blocks no operand of the case reaches are exactly zero
                dead = ref.abs().max(0).values == 0
                assert bool((out[:, dead] == 0).all()), (l, E, name)


def test_deterministic_samples_do_not_depend_on_the_batch_around_them(dev):
    """under ddk_config.deterministic = 1 the scores, node rows and 20-step poses of a sample are the same BITS whichever other samples share its
    batch - samples [0:5] and [35:40] of a 40-sample forward equal the two 5-sample forwards, a 13-sample batch cut out of the middle likewise (the layout
    north_star's sample sharding produces: config 5, 40 samples over 8 ranks).  What makes it so: the conv launches' work units are aligned per (edge group,
    level segment, SAMPLE) (det_ranges_kernel), so where the 32-edge tiles cut a node's run of edges depends on the sample's own edge list only."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context, Complex
    kern = int(os.environ.get('DDK_CONV_KERNEL', '0'))
    for n_res, seed in ((300, 2), (2000, 3)):
        c = synthetic.make_complex(seed, n_res=n_res)
        P = smr.random_state_dict(CFG, seed=11)
        ctx = Context(device=0, deterministic=1, conv_kernel=kern)
        ctx.load_state_dict(P)
        B = 40
        steps = 20 if n_res == 300 else 3
        coeffs, pos0 = _sampler_inputs(c, B, steps=steps)
        R = int(np.asarray(c['mask_rotate']).reshape(-1, len(c['lig_x'])).shape[0])
        z = torch.randn(steps, B, 6 + R, generator=torch.Generator().manual_seed(8))
        # poses at three distances: far (few cross edges, heavy pruning), the bench's spread, inside the pocket
        pos0[:14] += np.random.default_rng(1).normal(0, 12.0, size=(14, 1, 3)).astype(np.float32)
        full = Complex(ctx, c, B)

        def run(cx, sl, t):
            pos = T(pos0[sl].copy()).to(dev)
            tr, rot, tor = cx.score_forward(pos, t, t, t)
            nb = pos.shape[0]
            rows = cx.lig_node_features(nb, dev).cpu()
            cx.sample(pos, *coeffs, z[:, sl].contiguous().to(dev))
            return tr.cpu(), rot.cpu(), tor.cpu().reshape(nb, -1), rows.reshape(nb, -1), pos.cpu()

        for t in (1.0, 0.3, 0.05):
            ref = run(full, slice(0, B), t)
            for sl in (slice(0, 5), slice(35, 40), slice(14, 27)):
                small = Complex(ctx, c, sl.stop - sl.start)
                got = run(small, sl, t)
                for name, a, b in zip(('tr', 'rot', 'tor', 'ligand rows', 'poses'), got, ref):
                    assert torch.equal(a, b[sl]), (n_res, t, sl, name, float((a - b[sl]).abs().max()))
                small.close()
        full.close()


@pytest.mark.parametrize('l', range(5))
def test_two_limb_kernel_is_fp32_grade(dev, l):
    """The default conv kernel since round 6 (ddk_config.conv_kernel = 0, k_conv_x2.hip): every operand of the two radial-MLP GEMMs (tensor_layers.py:140-143,154-155)
    as TWO fp16 limbs and the three limb products hi.hi + hi.mid + mid.hi in one fp32 accumulator.  hi = fp16(x) and mid = fp16(x - hi) round to nearest, so
    |x - hi - mid| <= 2^-22 |x| (<= 2^-25 absolute after the per-group range scaling for values more than 2^-3 under their group's maximum); the dropped mid.mid is
    <= 2^-22 |x y| as well: a product carries <= 3 * 2^-22 relative error, a K = 72 dot product less than the classical bound 72 * 2^-24 of an fp32 FMA chain.  Measured
    here on 20k edges of every layer shape against the fp64 oracle, beside the three-limb / six-product kernel (3: products exact to 2^-33) and the fp32-MFMA chains (1):
    the default must stay within 1.5x of the fp32 chains' error (+ 1e-7) and under 1e-5 relative - an order under north_star's 1e-4."""
    from disco_diffdock_amd.runtime import Context
    from test_gpu_ops import _random_case, CFG as OCFG
    from helpers import elem_err
    from test_gpu_round3 import _record_drift
    N, splits = 1000, [0, 3000, 9000, 15000, 20000]
    i_irr, o_irr = OCFG.conv_irreps(l)
    Pl = smr.random_conv_layer_params(OCFG, l, 321 + l, True)
    node, ei, ea, sh = _random_case(l, N, splits, 9 + l, True)
    P = {'L.' + k: v.double() for k, v in Pl.items()}
    ref = smr.tp_conv_layer(P, 'L', node.double(), ei, [ea.double()[splits[i]:splits[i + 1]] for i in range(4)], sh.double(),
                            i_irr, '1x0e+1x1o', o_irr, residual=True, batch_norm=True, faster=True, edge_groups=4)
    args = (l, node.to(dev), ei[0].to(dev), ei[1].to(dev), splits, ea.to(dev), sh.to(dev), smr.irreps_dim(o_irr))
    err, outs = {}, {}
    for kernel in (0, 1, 3):
        ctx = Context(device=0, conv_kernel=kernel)
        assert int(ctx.cfg.conv_kernel) == kernel
        ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in Pl.items()})
        outs[kernel] = ctx.conv_forward(*args).cpu()
        err[kernel] = (rel_err(outs[kernel], ref), elem_err(outs[kernel], ref))
    print(f'conv layer {l} vs fp64: two limbs / three products (default) {err[0]}, fp32 MFMA chains {err[1]}, three limbs / six products {err[3]}')
    _record_drift(f'conv_layer_{l}_20k_edges_vs_fp64_two_limb_default', err[0][0], bar=1e-5,
                  two_limbs_three_products=err[0][0], fp32_chains=err[1][0], six_products=err[3][0], two_limbs_elem=err[0][1], fp32_chains_elem=err[1][1], six_products_elem=err[3][1])
    assert not torch.equal(outs[0], outs[3])                  # (it IS another arithmetic: the mode switch reaches the kernel)
    assert err[0][0] < 1e-5 and err[0][0] <= 1.5 * err[1][0] + 1e-7 and err[0][1] <= 1.5 * err[1][1] + 1e-6, err
