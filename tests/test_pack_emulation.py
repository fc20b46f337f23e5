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
    assert tiles == [23, 32, 37, 59, 59][l]          # (58.5 ideal for W = 1872: shared dot-product tails, 6-channel columns packed 4 units per tile)
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


def test_head_layouts_by_lane_emulation(built, tables):
    """tor_bond_conv (12 tiles, W = 288) and final_conv (W = 144, MLP width 48 padded to 72) as layouts of the fused conv
    kernel.  The packed tables (conv.100 / conv.101) driven through the lane-level emulation on the oracle's own head graphs and node
    features must reproduce the oracle's e3nn-style FullyConnectedTensorProduct convolutions (scatter-mean, before BatchNorm)."""
    from disco_diffdock_amd import synthetic
    from disco_diffdock_amd.runtime import Context
    from oracle import e3nn_lite as o3
    from helpers import batch_of
    import emu_conv
    import torch.nn.functional as F
    P = smr.random_state_dict(CFG, seed=19)
    ctx = Context(device=-1)
    ctx.load_state_dict(P)
    assert len(ctx.export('conv.100.tiles', np.int32)) // 4 == 12 and len(ctx.export('conv.101.tiles', np.int32)) // 4 == 18
    c = synthetic.make_complex(9, n_res=30, n_lig=18)
    B, t = 2, 0.4
    rng = np.random.default_rng(0)
    pos = np.stack([c['lig_pos'] + rng.normal(0, 2.0, size=(1, 3)) for _ in range(B)]).astype(np.float32)
    b = batch_of(c, B, pos)
    from oracle import sampler_ref as spr
    spr.set_time(b, t, t, t, B)
    dt = torch.float64
    P64 = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in P.items()}
    for nt in ('ligand', 'receptor'):
        b[nt].pos = b[nt].pos.to(dt)
    lig, rec, tr_s, rot_s, tor_s, lig_sig, graph = smr.embed(P64, CFG, b, dt, True)
    ns = CFG.ns
    sh_irreps = o3.Irreps.spherical_harmonics(CFG.sh_lmax)
    conv_out = CFG.conv_irreps(CFG.num_conv_layers - 1)[1]
    x_pad = lig.numpy()
    # ---- final_conv ----
    ei, ea, esh = smr.build_center_conv_graph(b, CFG, lig_sig, dt)
    ea = torch.cat([smr.mlp2(ea, P64, 'center_edge_embedding', 0, 3), lig[ei[1], :ns]], -1)
    want = smr.tp_conv_layer(P64, 'final_conv', lig, ei, ea, esh, conv_out, sh_irreps, '2x1o + 2x1e', residual=False, batch_norm=False,
                             faster=False, out_nodes=B).numpy()
    attr72 = np.concatenate([ea.numpy(), np.zeros((ea.shape[0], 24))], 1)
    got = emu_conv.emulate(ctx, 101, x_pad, ei[0].numpy(), ei[1].numpy(), [0, ei.shape[1]], attr72, esh.numpy())
    got = got[:B, :12] / np.bincount(ei[0].numpy(), minlength=B)[:, None]
    assert np.abs(want).max() > 1e-3 and rel_err(got, want) < 2e-6
    # ---- tor_bond_conv ----
    bonds, tei, tea, tesh = smr.build_bond_conv_graph(P64, b, CFG, dt)
    bvec = b['ligand'].pos[bonds[1]] - b['ligand'].pos[bonds[0]]
    tp_tor = o3.FullTensorProduct(sh_irreps, '2e')
    full_sh = tp_tor(tesh, smr._sh(bvec, '2e')[tei[0]])
    assert str(tp_tor.irreps_out[0].ir) == '1o'
    Tvec = full_sh[:, :3].numpy()                        # the 1o block: what heads_pre_kernel writes into sh[1:4]
    bond_attr = lig[bonds[0]] + lig[bonds[1]]
    tattr = torch.cat([tea, lig[tei[1], :ns], bond_attr[tei[0], :ns]], -1)
    want = smr.tp_conv_layer(P64, 'tor_bond_conv', lig, tei, tattr, full_sh, conv_out, tp_tor.irreps_out, f'{ns}x0o + {ns}x0e', residual=False,
                             batch_norm=False, faster=False, out_nodes=bonds.shape[1]).numpy()
    sh4 = np.concatenate([np.ones((len(Tvec), 1)), Tvec], 1)
    got = emu_conv.emulate(ctx, 100, np.concatenate([x_pad, np.zeros((bonds.shape[1], 84))]), tei[0].numpy(), tei[1].numpy(), [0, tei.shape[1]],
                           tattr.numpy(), sh4)
    got = got[:bonds.shape[1], :2 * ns] / np.maximum(np.bincount(tei[0].numpy(), minlength=bonds.shape[1]), 1)[:, None]
    assert np.abs(want).max() > 1e-3 and rel_err(got, want) < 2e-6
    # the kernel's closed form of T (k_heads.hip) == the oracle's FullTensorProduct 1o block
    s1 = tesh[:, 1:4].numpy()
    bx, by, bz = (bvec / bvec.norm(dim=-1, keepdim=True))[tei[0]].numpy().T
    s3, s5, ca, cb = np.sqrt(3.0), np.sqrt(5.0), 1 / np.sqrt(10.0), 1 / np.sqrt(30.0)
    y0, y1, y2, y3, y4 = s5 * s3 * bx * bz, s5 * s3 * bx * by, s5 * (by * by - 0.5 * (bx * bx + bz * bz)), s5 * s3 * by * bz, s5 * (s3 * 0.5) * (bz * bz - bx * bx)
    T0 = s3 * (-cb * s1[:, 0] * y2 - ca * s1[:, 0] * y4 + ca * s1[:, 1] * y1 + ca * s1[:, 2] * y0)
    T1 = s3 * (ca * s1[:, 0] * y1 + 2.0 * cb * s1[:, 1] * y2 + ca * s1[:, 2] * y3)
    T2 = s3 * (ca * s1[:, 0] * y0 + ca * s1[:, 1] * y3 - cb * s1[:, 2] * y2 + ca * s1[:, 2] * y4)
    assert np.abs(np.stack([T0, T1, T2], 1) - Tvec).max() < 1e-12
