"""Narrowing Tier B: checks of oracle/e3nn_lite.py that need no e3nn wheel.

The reference calls e3nn for spherical harmonics, Wigner 3j symbols (inside FullyConnectedTensorProduct / FullTensorProduct) and BatchNorm
(models/score_model.py:35,152,295-296,422,436; models/tensor_layers.py:137,145); the wheel is not vendored and not installable here, so the oracle restates them.
What these tests pin WITHOUT the wheel:
* closed forms: w3j(1,1,0) = delta_ij / sqrt3, w3j(1,1,1) = eps_ijk / sqrt6 (signs: the reference's own dot / cross products of
  models/tensor_layers.py:75-83 through test_faster_tp_equals_fctp), w3j(1,2,1) = the symmetric-traceless embedding of the l = 2 basis _sh_l(2, .) defines -
  spherical harmonics and w3j are mutually consistent;
* equivariance of every FullyConnectedTensorProduct / FullTensorProduct instance the models build under random rotations AND inversion (the D matrices of
  l <= 2 are derived from the spherical harmonics themselves; parity of 1o vs 1e is otherwise untested).
What remains unpinned (oracle/__init__.py): the overall sign convention of w3j(1,2,1) (a global sign of the torsion head's 1o x 2e path, absorbed by trained
weights but not by a checkpoint trained against e3nn's sign) and torch_cluster's tie-break under the neighbour cap."""
import math

import numpy as np
import pytest
import torch

from oracle import e3nn_lite as o3l
from oracle import score_model_ref as smr
from oracle.confidence_ref import ConfidenceModelConfig

DT = torch.float64


def _sh(l, v):
    """component-normalised real spherical harmonics of the oracle on UNNORMALISED vectors (homogeneous polynomials of degree l)"""
    return o3l._sh_l(l, v[..., 0], v[..., 1], v[..., 2]) * math.sqrt(2 * l + 1)


def test_w3j_110_and_111_closed_forms():
    w = o3l.wigner_3j(1, 1, 0, DT)[:, :, 0]
    assert torch.allclose(w.abs(), torch.eye(3, dtype=DT) / math.sqrt(3.0), atol=1e-12)
    assert torch.allclose(w, w[0, 0].sign() * torch.eye(3, dtype=DT) / math.sqrt(3.0), atol=1e-12)
    eps = torch.zeros(3, 3, 3, dtype=DT)
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[i, k, j] = 1.0, -1.0
    w = o3l.wigner_3j(1, 1, 1, DT)
    s = w[0, 1, 2].sign()
    assert torch.allclose(w, s * eps / math.sqrt(6.0), atol=1e-12)
    # every symbol has unit Frobenius norm (e3nn's normalisation) and the (l1, l2) <-> (l2, l1) symmetry of the real basis
    for l1, l2, l3 in ((1, 1, 0), (1, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 0), (1, 2, 2), (2, 2, 2), (0, 1, 1), (0, 2, 2)):
        w = o3l.wigner_3j(l1, l2, l3, DT)
        assert abs(float(torch.linalg.norm(w)) - 1.0) < 1e-12
        wt = o3l.wigner_3j(l2, l1, l3, DT).transpose(0, 1)
        assert torch.allclose(w, wt, atol=1e-12) or torch.allclose(w, -wt, atol=1e-12)


def test_w3j_121_is_the_symmetric_traceless_embedding_of_the_l2_basis():
    """sum_jk w3j(1,2,1)_ijk Y2_j(v) u_k  is proportional to  (v v^T - |v|^2 I / 3) u  for every v, u: the l = 2 spherical harmonics and the symbol that contracts
    them in the torsion head (models/score_model.py:295-296: FullTensorProduct(sh, '2e')) describe the same five matrices."""
    g = torch.Generator().manual_seed(0)
    v, u = torch.randn(64, 3, generator=g, dtype=DT), torch.randn(64, 3, generator=g, dtype=DT)
    w = o3l.wigner_3j(1, 2, 1, DT)
    lhs = torch.einsum('ijk,nj,nk->ni', w, _sh(2, v), u)
    M = torch.einsum('ni,nj->nij', v, v) - (v * v).sum(-1)[:, None, None] * torch.eye(3, dtype=DT) / 3.0
    rhs = torch.einsum('nij,nj->ni', M, u)
    c = float((lhs * rhs).sum() / (rhs * rhs).sum())
    assert abs(c) > 0.1 and torch.allclose(lhs, c * rhs, atol=1e-12)
    # the same statement for w3j(1,1,2): the l = 2 part of u (x) v
    w = o3l.wigner_3j(1, 1, 2, DT)
    lhs = torch.einsum('ijk,ni,nj->nk', w, v, v)
    c2 = float((lhs * _sh(2, v)).sum() / (_sh(2, v) ** 2).sum())
    assert abs(c2) > 0.1 and torch.allclose(lhs, c2 * _sh(2, v), atol=1e-12)


def _D(l, R):
    """D^l(R) in the oracle's real basis, from the spherical harmonics themselves: Y_l(R v) = D^l(R) Y_l(v)"""
    if l == 0:
        return torch.ones(1, 1, dtype=DT)
    if l > 2:      # (the oracle has no l = 3 harmonics: 1 (x) 2 -> 3 through the symbol, sum_ij w_ijk w_ijk' = delta_kk' / (2 l + 1))
        w = o3l.wigner_3j(1, l - 1, l, DT)
        return (2 * l + 1) * torch.einsum('ijk,ia,jb,abc->kc', w, _D(1, R), _D(l - 1, R), w)
    g = torch.Generator().manual_seed(l)
    v = torch.randn(40, 3, generator=g, dtype=DT)
    A, Bm = _sh(l, v), _sh(l, v @ R.T)
    D = torch.linalg.lstsq(A, Bm).solution.T
    assert torch.allclose(A @ D.T, Bm, atol=1e-10)
    return D


def _rep(irreps, R, inversion):
    """block-diagonal representation of (rotation R, optionally followed by the inversion) on features with these irreps"""
    blocks = []
    for mul, ir in o3l.Irreps(irreps):
        D = _D(ir.l, R)
        if inversion:
            D = D * ir.p
        blocks += [D] * mul
    return torch.block_diag(*blocks)


def _rotations():
    from scipy.spatial.transform import Rotation
    return [torch.from_numpy(Rotation.random(random_state=s).as_matrix()).to(DT) for s in (0, 1)]


def _tp_instances():
    cfg = smr.ScoreModelConfig(latent_vocab=64)
    ns, nv = cfg.ns, cfg.nv
    sh = '1x0e + 1x1o'
    conv_out = cfg.conv_irreps(cfg.num_conv_layers - 1)[1]
    tor_sh = o3l.FullTensorProduct(sh, '2e').irreps_out
    inst = [('final_conv', conv_out, sh, '2x1o + 2x1e'), ('tor_bond_conv', conv_out, str(tor_sh), f'{ns}x0o + {ns}x0e')]
    for l in range(cfg.num_conv_layers):
        inst.append((f'conv_layers.{l} as FCTP', *[cfg.conv_irreps(l)[0], sh, cfg.conv_irreps(l)[1]]))
    ccfg = ConfidenceModelConfig()
    for l in range(ccfg.num_conv_layers):
        i_irr, o_irr = ccfg.conv_irreps(l)
        inst.append((f'confidence conv {l}', i_irr, ccfg.sh_irreps, o_irr))
    return inst


@pytest.mark.parametrize('name,in1,in2,out', _tp_instances(), ids=[i[0] for i in _tp_instances()])
def test_fctp_instances_are_equivariant_under_rotation_and_inversion(name, in1, in2, out):
    tp = o3l.FullyConnectedTensorProduct(in1, in2, out)
    g = torch.Generator().manual_seed(1)
    n = 5
    x1 = torch.randn(n, o3l.Irreps(in1).dim, generator=g, dtype=DT)
    x2 = torch.randn(n, o3l.Irreps(in2).dim, generator=g, dtype=DT)
    w = torch.randn(n, tp.weight_numel, generator=g, dtype=DT)
    y = tp(x1, x2, w)
    assert float(y.abs().max()) > 1e-3
    for R in _rotations():
        for inv in (False, True):
            y2 = tp(x1 @ _rep(in1, R, inv).T, x2 @ _rep(in2, R, inv).T, w)
            assert torch.allclose(y2, y @ _rep(out, R, inv).T, atol=1e-9), (name, inv)


def test_full_tensor_product_of_the_torsion_head_is_equivariant():
    """FullTensorProduct('1x0e + 1x1o', '2e') (models/score_model.py:295): outputs sorted 2e, 1o, 2o(?)... whatever the order, the map commutes with rotations and
    the inversion, and its 1o block is sqrt3 * w3j(1,2,1) contracted with the two inputs"""
    tp = o3l.FullTensorProduct('1x0e + 1x1o', '2e')
    g = torch.Generator().manual_seed(2)
    x1, x2 = torch.randn(6, 4, generator=g, dtype=DT), torch.randn(6, 5, generator=g, dtype=DT)
    y = tp(x1, x2)
    for R in _rotations():
        for inv in (False, True):
            y2 = tp(x1 @ _rep('1x0e + 1x1o', R, inv).T, x2 @ _rep('2e', R, inv).T)
            assert torch.allclose(y2, y @ _rep(str(tp.irreps_out), R, inv).T, atol=1e-9)
    sl = [s for (mul, ir), s in zip(tp.irreps_out, tp.irreps_out.slices()) if ir.l == 1 and ir.p == -1]
    assert len(sl) == 1
    want = math.sqrt(3.0) * torch.einsum('ijk,ni,nj->nk', o3l.wigner_3j(1, 2, 1, DT), x1[:, 1:], x2)
    assert torch.allclose(y[:, sl[0]], want, atol=1e-12)


def test_spherical_harmonics_component_normalisation_and_parity():
    """|Y_l|^2 = 2l + 1 on the unit sphere (normalization='component', models/score_model.py:326,353,386), Y_l(-v) = (-1)^l Y_l(v), and the l = 1 harmonics
    are sqrt3 * (x, y, z) - the reference mixes sh[1:] with xyz cross products (models/tensor_layers.py:75-83)"""
    g = torch.Generator().manual_seed(3)
    v = torch.nn.functional.normalize(torch.randn(50, 3, generator=g, dtype=DT), dim=-1)
    for l in (0, 1, 2):
        Y = o3l.spherical_harmonics(l, v, normalize=True, normalization='component')
        assert torch.allclose((Y * Y).sum(-1), torch.full((50,), 2.0 * l + 1, dtype=DT), atol=1e-12)
        assert torch.allclose(o3l.spherical_harmonics(l, -v, normalize=True, normalization='component'), (-1) ** l * Y, atol=1e-12)
    assert torch.allclose(o3l.spherical_harmonics(1, v, normalize=True, normalization='component'), math.sqrt(3.0) * v, atol=1e-12)
