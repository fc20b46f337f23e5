"""Restatement of the slice of ``e3nn`` the DiffDock hot path touches.  TEST INFRASTRUCTURE.

e3nn is an un-vendored, un-pinned third-party dependency of the reference (imports at
models/score_model.py:1, models/tensor_layers.py:1,7) and is not installable here, so this
file restates its *published* conventions from memory -> every function here is
"PARITY UNPINNED" (see oracle/__init__.py).  Call sites it serves:

* ``o3.Irreps``                        tensor_layers.py:49-56,67; score_model.py:35
* ``o3.spherical_harmonics``           score_model.py:295,342,371,406,422,436
* ``o3.FullyConnectedTensorProduct``   tensor_layers.py:137 (final_conv, tor_bond_conv)
* ``o3.FullTensorProduct``             score_model.py:152,296
* ``e3nn.nn.BatchNorm`` (eval)         tensor_layers.py:145,161-162

Conventions restated (e3nn >= 0.4):
* l=1 real basis is (x, y, z); real spherical harmonics are the standard ones with the axes
  cyclically relabelled (standard y->x, z->y, x->z), i.e. Y_2 = [sqrt3 xz, sqrt3 xy,
  y^2-(x^2+z^2)/2, sqrt3 yz, sqrt3/2 (z^2-x^2)]; 'component' normalisation multiplies by
  sqrt(2l+1) so that |Y_l|^2 = 2l+1 on the unit sphere.
* wigner_3j(l1,l2,l3): SU(2) Clebsch-Gordan (Racah formula) rotated into the real basis by
  ``change_basis_real_to_complex`` (with the (-i)^l phase), real part, Frobenius norm 1.
* TensorProduct paths: instructions enumerated over (i_in1, i_in2, i_out) in that nesting
  order; 'uvw' weights flattened as [mul1, mul2, mul_out]; path coefficient
  sqrt(dim(ir_out) / sum_{paths into the same output slot} mul1*mul2)
  (irrep_normalization='component', path_normalization='element').
* BatchNorm(eval): only 0e channels have running_mean/bias; every channel is scaled by
  weight / sqrt(running_var + eps).
"""
import math
from fractions import Fraction
from functools import lru_cache

import torch
from torch import nn


# --------------------------------------------------------------------------------------
# Irrep / Irreps
# --------------------------------------------------------------------------------------
class Irrep(tuple):
    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                s = l.strip()
                l, p = int(s[:-1]), {'e': 1, 'o': -1}[s[-1]]
            else:
                l, p = l
        return super().__new__(cls, (int(l), int(p)))

    @property
    def l(self):
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self[0] + 1

    def is_scalar(self):
        return self[0] == 0 and self[1] == 1

    def __mul__(self, other):
        other = Irrep(other)
        p = self.p * other.p
        return [Irrep(l, p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __repr__(self):
        return f"{self[0]}{'e' if self[1] == 1 else 'o'}"

    __str__ = __repr__


class _MulIr(tuple):
    def __new__(cls, mul, ir):
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self[0] * self[1].dim

    def __repr__(self):
        return f"{self[0]}x{self[1]}"


class Irreps(tuple):
    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out = []
        if isinstance(irreps, Irrep):
            out.append(_MulIr(1, irreps))
        elif isinstance(irreps, str):
            if irreps.strip() != '':
                for chunk in irreps.split('+'):
                    chunk = chunk.strip()
                    if 'x' in chunk:
                        mul, ir = chunk.split('x')
                        out.append(_MulIr(int(mul), Irrep(ir)))
                    else:
                        out.append(_MulIr(1, Irrep(chunk)))
        elif irreps is not None:
            for item in irreps:
                if isinstance(item, (str, Irrep)):
                    out.append(_MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    def slices(self):
        s, i = [], 0
        for mul_ir in self:
            s.append(slice(i, i + mul_ir.dim))
            i += mul_ir.dim
        return s

    @property
    def dim(self):
        return sum(mul_ir.dim for mul_ir in self)

    @property
    def num_irreps(self):
        return sum(mul for mul, _ in self)

    def sort(self):
        """Returns (sorted irreps, p, inv) like e3nn: stable sort by the Irrep tuple."""
        order = sorted(range(len(self)), key=lambda i: (self[i].ir, i))
        inv = tuple(order)
        p = [0] * len(order)
        for new, old in enumerate(order):
            p[old] = new
        return Irreps([self[i] for i in order]), tuple(p), inv

    def __eq__(self, other):
        return tuple(self) == tuple(Irreps(other))

    def __hash__(self):
        return hash(tuple(self))

    def __repr__(self):
        return '+'.join(repr(m) for m in self)


# --------------------------------------------------------------------------------------
# spherical harmonics (l <= 2 is all the hot path needs; sh_lmax=1 and the "2e" bond axis)
# --------------------------------------------------------------------------------------
def _sh_l(l, x, y, z):
    if l == 0:
        return torch.ones_like(x).unsqueeze(-1)
    if l == 1:
        return torch.stack([x, y, z], dim=-1)
    if l == 2:
        s3 = math.sqrt(3.0)
        return torch.stack([s3 * x * z, s3 * x * y, y * y - 0.5 * (x * x + z * z),
                            s3 * y * z, (s3 / 2.0) * (z * z - x * x)], dim=-1)
    raise NotImplementedError("oracle e3nn_lite: spherical harmonics only for l <= 2")


def spherical_harmonics(l, x, normalize, normalization='integral'):
    """o3.spherical_harmonics: l may be an int, a list of ints, an irreps string or Irreps."""
    if isinstance(l, int):
        ls = [l]
    elif isinstance(l, (list,)) and all(isinstance(i, int) for i in l):
        ls = list(l)
    else:
        ls = [ir.l for mul, ir in Irreps(l) for _ in range(mul)]
    if normalize:
        x = torch.nn.functional.normalize(x, dim=-1)
    xx, yy, zz = x[..., 0], x[..., 1], x[..., 2]
    out = []
    for li in ls:
        sh = _sh_l(li, xx, yy, zz)
        if normalization == 'component':
            sh = sh * math.sqrt(2 * li + 1)
        elif normalization == 'integral':
            sh = sh * (math.sqrt(2 * li + 1) / math.sqrt(4 * math.pi))
        elif normalization != 'norm':
            raise ValueError(normalization)
        out.append(sh)
    return torch.cat(out, dim=-1)


# --------------------------------------------------------------------------------------
# Wigner 3j in e3nn's real basis
# --------------------------------------------------------------------------------------
def _f(n):
    return math.factorial(round(n))


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3):
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max([-j1 + j2 + m3, -j1 + m1, 0]))
    vmax = int(min([j2 + j3 + m1, j3 - j1 + j2, j3 + m3]))
    C = ((2.0 * j3 + 1.0) * Fraction(
        _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3),
        _f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))) ** 0.5
    S = 0
    for v in range(vmin, vmax + 1):
        S += (-1) ** int(v + j2 + m2) * Fraction(
            _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v),
            _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3))
    return float(C * S)


def _su2_cg(j1, j2, j3):
    mat = torch.zeros(2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1, dtype=torch.float64)
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            if abs(m1 + m2) <= j3:
                mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l):
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _wigner_3j_f64(l1, l2, l3):
    assert abs(l2 - l3) <= l1 <= l2 + l3
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    C = _su2_cg(l1, l2, l3).to(torch.complex128)
    C = torch.einsum('ij,kl,mn,ikn->jlm', Q1, Q2, torch.conj(Q3.T), C)
    assert torch.all(torch.abs(C.imag) < 1e-9)
    C = C.real
    return C / torch.linalg.norm(C)


def wigner_3j(l1, l2, l3, dtype=torch.float32):
    return _wigner_3j_f64(l1, l2, l3).to(dtype)


# --------------------------------------------------------------------------------------
# Tensor products
# --------------------------------------------------------------------------------------
class _TPBase(nn.Module):
    def _blocks(self, x, irreps):
        out = []
        for (mul, ir), sl in zip(irreps, irreps.slices()):
            out.append(x[..., sl].reshape(x.shape[:-1] + (mul, ir.dim)))
        return out


class FullyConnectedTensorProduct(_TPBase):
    """o3.FullyConnectedTensorProduct(in1, in2, out, shared_weights=False) with external weights."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, shared_weights=False, **kwargs):
        super().__init__()
        assert not shared_weights, "oracle restates only the per-edge-weight use (tensor_layers.py:137)"
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        self.instructions = []  # (i1, i2, io, weight offset, weight shape)
        off = 0
        for i1, (mul1, ir1) in enumerate(self.irreps_in1):
            for i2, (mul2, ir2) in enumerate(self.irreps_in2):
                for io, (mulo, iro) in enumerate(self.irreps_out):
                    if iro in ir1 * ir2:
                        shape = (mul1, mul2, mulo)
                        self.instructions.append((i1, i2, io, off, shape))
                        off += mul1 * mul2 * mulo
        self.weight_numel = off
        fan = {}
        for i1, i2, io, _, shape in self.instructions:
            fan[io] = fan.get(io, 0) + shape[0] * shape[1]
        self.coeffs = [math.sqrt(self.irreps_out[io].ir.dim / fan[io]) for (_, _, io, _, _) in self.instructions]

    def forward(self, x1, x2, weight):
        b1, b2 = self._blocks(x1, self.irreps_in1), self._blocks(x2, self.irreps_in2)
        outs = [torch.zeros(x1.shape[:-1] + (mul, ir.dim), dtype=x1.dtype, device=x1.device)
                for mul, ir in self.irreps_out]
        for (i1, i2, io, off, shape), coeff in zip(self.instructions, self.coeffs):
            n = shape[0] * shape[1] * shape[2]
            w = weight[..., off:off + n].reshape(weight.shape[:-1] + shape)
            c = wigner_3j(self.irreps_in1[i1].ir.l, self.irreps_in2[i2].ir.l, self.irreps_out[io].ir.l, x1.dtype)
            outs[io] = outs[io] + coeff * torch.einsum('...uvw,ijk,...ui,...vj->...wk', w, c, b1[i1], b2[i2])
        return torch.cat([o.reshape(o.shape[:-2] + (o.shape[-2] * o.shape[-1],)) for o in outs], dim=-1)   # (explicit size: works for 0 edges)


class FullTensorProduct(_TPBase):
    """o3.FullTensorProduct(in1, in2): weight-less 'uvuv' product, outputs sorted by irrep."""

    def __init__(self, irreps_in1, irreps_in2, **kwargs):
        super().__init__()
        self.irreps_in1, self.irreps_in2 = Irreps(irreps_in1), Irreps(irreps_in2)
        out, ins = [], []
        for i1, (mul1, ir1) in enumerate(self.irreps_in1):
            for i2, (mul2, ir2) in enumerate(self.irreps_in2):
                for iro in ir1 * ir2:
                    ins.append((i1, i2, len(out)))
                    out.append((mul1 * mul2, iro))
        out = Irreps(out)
        self.irreps_out, p, _ = out.sort()
        self.instructions = [(i1, i2, p[io]) for (i1, i2, io) in ins]

    def forward(self, x1, x2):
        b1, b2 = self._blocks(x1, self.irreps_in1), self._blocks(x2, self.irreps_in2)
        outs = [None] * len(self.irreps_out)
        for i1, i2, io in self.instructions:
            iro = self.irreps_out[io].ir
            c = wigner_3j(self.irreps_in1[i1].ir.l, self.irreps_in2[i2].ir.l, iro.l, x1.dtype)
            # each instruction owns its output slot -> 'element' fan-in is 1 -> coeff sqrt(dim_out)
            r = math.sqrt(iro.dim) * torch.einsum('ijk,...ui,...vj->...uvk', c, b1[i1], b2[i2])
            outs[io] = r.reshape(r.shape[:-3] + (-1,))
        return torch.cat(outs, dim=-1)


class BatchNorm(nn.Module):
    """e3nn.nn.BatchNorm(irreps) — evaluation mode only (the sampler runs under model.eval())."""

    def __init__(self, irreps, eps=1e-5, momentum=0.1, affine=True, reduce='mean', instance=False,
                 normalization='component'):
        super().__init__()
        self.irreps = Irreps(irreps)
        self.eps = eps
        num_scalar = sum(mul for mul, ir in self.irreps if ir.is_scalar())
        num_features = self.irreps.num_irreps
        self.register_buffer('running_mean', torch.zeros(num_scalar))
        self.register_buffer('running_var', torch.ones(num_features))
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_scalar))

    def forward(self, x):
        assert not self.training, "oracle BatchNorm restates eval mode only"
        return batch_norm_eval(x, self.irreps, self.weight, self.bias, self.running_mean, self.running_var, self.eps)


def batch_norm_eval(x, irreps, weight, bias, running_mean, running_var, eps=1e-5):
    irreps = Irreps(irreps)
    out, ix, iw, ib = [], 0, 0, 0
    for mul, ir in irreps:
        d = ir.dim
        field = x[:, ix:ix + mul * d].reshape(-1, mul, d)
        ix += mul * d
        if ir.is_scalar():
            field = field - running_mean[ib:ib + mul].reshape(1, mul, 1)
        scale = (running_var[iw:iw + mul] + eps).pow(-0.5) * weight[iw:iw + mul]
        field = field * scale.reshape(1, mul, 1)
        if ir.is_scalar():
            field = field + bias[ib:ib + mul].reshape(1, mul, 1)
            ib += mul
        iw += mul
        out.append(field.reshape(-1, mul * d))
    return torch.cat(out, dim=-1)
