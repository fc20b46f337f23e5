"""CPU tests of the C-ABI library's host side: it loads, exports every declared symbol, packs the radial-MLP
weights into MFMA fragment order correctly (checked by a lane-level emulation of the fused kernel against the
oracle) and refuses to launch without a device."""
import ctypes
import re
import os

import numpy as np
import pytest
import torch

from oracle import score_model_ref as smr
from helpers import rel_err

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
CFG = smr.ScoreModelConfig()


@pytest.fixture(scope='module')
def built():
    from disco_diffdock_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    from disco_diffdock_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'ddk.h')).read()
    declared = set(re.findall(r'\b(ddk_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SYMBOLS)
    L = ctypes.CDLL(built)
    for s in declared:
        assert hasattr(L, s), s
    assert b'gfx950' in _lib.lib().ddk_version()
    # the test hooks bound by _lib.py are declared too (include/ddk_debug.h), and nothing else is bound
    dbg = set(re.findall(r'\b(ddk_debug_[a-z0-9_]+)\s*\(', open(os.path.join(ROOT, 'include', 'ddk_debug.h')).read()))
    assert dbg == set(_lib.DEBUG_SYMBOLS)
    for s in dbg:
        assert hasattr(L, s), s
    bound = set(re.findall(r'L\.(ddk_[a-z0-9_]+)\.', open(os.path.join(ROOT, 'disco_diffdock_amd', '_lib.py')).read()))
    assert bound <= declared | dbg, bound - declared - dbg


def test_host_only_context_refuses_to_launch(built):
    from disco_diffdock_amd.runtime import Context
    ctx = Context(device=-1)
    ctx.finalize()
    rc = ctx.L.ddk_tp_forward(ctx.h, 3, None, None, None, 4, None, None)
    assert rc == -3 and b'host-only' in ctx.L.ddk_last_error(ctx.h)
    go = (ctypes.c_int64 * 5)(0, 0, 0, 0, 0)
    rc = ctx.L.ddk_conv_forward(ctx.h, 3, None, 4, None, None, go, None, None, None, None)
    assert rc == -3


def test_unsupported_config_is_loud(built):
    from disco_diffdock_amd.runtime import Context
    with pytest.raises(RuntimeError, match='ns=24'):
        Context(device=-1, ns=16, nv=4)


def test_missing_or_misshaped_weights_are_loud(built):
    from disco_diffdock_amd.runtime import Context
    P = smr.random_conv_layer_params(CFG, 3, 1, True)
    ctx = Context(device=-1)
    bad = {f'conv_layers.3.{k}': v for k, v in P.items() if k != 'fc.2.4.bias'}
    with pytest.raises(RuntimeError, match='missing state_dict key: conv_layers.3.fc.2.4.bias'):
        ctx.load_state_dict(bad)
    ctx = Context(device=-1)
    bad = {f'conv_layers.3.{k}': (v[:-1] if k == 'fc.0.4.weight' else v) for k, v in P.items()}
    with pytest.raises(RuntimeError, match='shape mismatch'):
        ctx.load_state_dict(bad)


@pytest.mark.parametrize('l', range(5))
def test_packing_by_lane_emulation(built, golden, l):
    """golden conv-layer case (produced by the reference) -> packed weights -> emulated wave algorithm."""
    from disco_diffdock_amd.runtime import Context
    import emu_conv
    z = golden(f'conv_layer_l{l}_bn1')
    P = smr.random_conv_layer_params(CFG, l, int(z['param_seed']), True)
    ctx = Context(device=-1)
    ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in P.items()})
    tiles = len(ctx.export(f'conv.{l}.tiles', np.int32)) // 4
    assert tiles == [24, 34, 42, 66, 66][l]
    node = z['node'].astype(np.float64)
    N, din = node.shape
    x_pad = np.zeros((N, 84))
    x_pad[:, :din] = node
    src, dst = z['edge_index'][0], z['edge_index'][1]
    summed = emu_conv.emulate(ctx, l, x_pad, src, dst, list(z['splits']), z['edge_attr'].astype(np.float64), z['sh'].astype(np.float64))
    deg = np.bincount(src, minlength=N)
    out = emu_conv.finalize(ctx, l, summed, deg, x_pad, z['out'].shape[1])
    assert rel_err(out, z['out']) < 5e-6


@pytest.mark.parametrize('l', range(5))
def test_gemm1_split_packing_by_lane_emulation(built, l):
    """GEMM1 split (ConvLayerDev::wn): W1 [edge_emb | x[src][:24] | x[dst][:24]] with the two node parts taken from the packed
    per-node terms in the accumulator's register order == the unsplit GEMM1, for every group's pair of (receiver, sender) role slots."""
    from disco_diffdock_amd.runtime import Context
    import emu_conv
    P = smr.random_conv_layer_params(CFG, l, 77 + l, True)
    ctx = Context(device=-1)
    ctx.load_state_dict({f'conv_layers.{l}.{k}': v for k, v in P.items()})
    rng = np.random.default_rng(l)
    N, per = 14, 37                     # 37 edges per group: a full and a ragged 32-edge tile
    x_pad = rng.normal(size=(N, 84))
    src, dst = np.sort(rng.integers(0, N, size=4 * per).reshape(4, per), axis=1).reshape(-1), rng.integers(0, N, size=4 * per)
    emb = rng.normal(size=(4 * per, 24))
    edge_attr = np.concatenate([emb, x_pad[src][:, :24], x_pad[dst][:, :24]], 1)
    sh = np.concatenate([np.ones((4 * per, 1)), rng.normal(size=(4 * per, 3))], 1)
    offs = [per * k for k in range(5)]
    full = emu_conv.emulate(ctx, l, x_pad, src, dst, offs, edge_attr, sh)
    split = emu_conv.emulate(ctx, l, x_pad, src, dst, offs, edge_attr, sh, split=True)
    assert np.abs(full).max() > 0.1 and np.abs(full - split).max() < 1e-9 * np.abs(full).max()


@pytest.mark.parametrize('l', range(5))
def test_confidence_layer_packing_by_lane_emulation(built, l):
    """l<=2 e3nn FullyConnectedTensorProduct layers of the confidence model: packed weights (instruction
    offsets, path coefficients, wigner constants, extra 1x2->1 rows) -> emulated wave algorithm == oracle conv layer."""
    from disco_diffdock_amd.runtime import Context
    from oracle import confidence_ref as cr
    import emu_conv
    cfg = cr.ConfidenceModelConfig()
    P = {k: v for k, v in cr.random_state_dict(cfg, seed=40 + l).items() if k.startswith('conv_layers.')}
    ctx = Context(device=-1, all_atoms=1, embedding_scale=10000.0, num_confidence_outputs=2)
    ctx.load_state_dict(P)
    i_irr, o_irr = cfg.conv_irreps(l)
    din, dout = smr.irreps_dim(i_irr), smr.irreps_dim(o_irr)
    g = torch.Generator().manual_seed(l)
    N, per = 12, 7
    from oracle import e3nn_lite as o3
    offs = [per * k for k in range(10)]
    E = offs[-1]
    node = torch.randn(N, din, generator=g)
    ei = torch.randint(0, N, (2, E), generator=g)
    ea = torch.randn(E, 72, generator=g)
    vec = torch.randn(E, 3, generator=g)
    vec[::5] = 0.0                      # zero-length edges: Y1 = Y2 = 0
    sh9 = o3.spherical_harmonics(cfg.sh_irreps, vec, normalize=True, normalization='component')
    x_pad = np.zeros((N, 84)); x_pad[:, :din] = node.numpy()
    summed = emu_conv.emulate(ctx, l, x_pad, ei[0].numpy(), ei[1].numpy(), offs, ea.numpy().astype(np.float64),
                              sh9[:, :4].numpy().astype(np.float64), mode=1, slots=list(range(9)))     # one accumulator slot per conv
    bn_mean, bn_scale, bn_bias = (ctx.export(f'conv.{l}.bn_{k}').reshape(9, 84) for k in ('mean', 'scale', 'bias'))
    for k in range(9):
        sl = slice(offs[k], offs[k + 1])
        want = cr.conv_layer(P, f'conv_layers.{9 * l + k}', cfg, l, node, ei[:, sl], ea[sl], sh9[sl], out_nodes=N).numpy()
        deg = np.bincount(ei[0, sl].numpy(), minlength=N)
        got = (summed[:, k] / np.maximum(deg, 1)[:, None] - bn_mean[k]) * bn_scale[k] + bn_bias[k]
        assert rel_err(got[:, :dout], want) < 5e-6, (l, k)
