"""Host-side mirror of the reference's models/tensor_layers.py interface for the hot path, backed by
libddk.so.  Same class names, constructor arguments, forward signatures and state_dict keys, so that the
parity tests read like tests of the reference modules:

* ``FasterTensorProduct(in_irreps, sh_irreps, out_irreps).forward(in_, sh, weight)``   tensor_layers.py:39-116
* ``TensorProductConvLayer(...).forward(node_attr, edge_index, edge_attr, edge_sh, out_nodes=None, reduce='mean')``
                                                                                          tensor_layers.py:119-168
* ``GaussianSmearing``                                                                   tensor_layers.py:171-181

Only the configuration the shipped score models use is implemented on the device (ns=24, nv=6, sh_lmax=1,
faster=True, edge_groups=4, residual=True, 2-layer ReLU radial MLP, eval mode); anything else raises.
"""
import math

import torch
from torch import nn

from .runtime import Context

_ORDER = ('0e', '1o', '1e', '0o')
_DIM = {'0e': 1, '1o': 3, '1e': 3, '0o': 1}


def parse_irreps(irreps):
    """'24x0e + 6x1o' -> {'0e':24,'1o':6,'1e':0,'0o':0}; order must be the reference's (0e,1o,1e,0o)."""
    muls = {k: 0 for k in _ORDER}
    seen = []
    for chunk in str(irreps).split('+'):
        chunk = chunk.strip()
        if not chunk:
            continue
        mul, ir = chunk.split('x') if 'x' in chunk else ('1', chunk)
        ir = ir.strip()
        if ir not in muls:
            raise RuntimeError(f'ddk: unsupported irrep {ir!r} (sh_lmax=1 first-order model only)')
        muls[ir] = int(mul)
        seen.append(ir)
    if seen != [k for k in _ORDER if k in seen]:
        raise RuntimeError(f'ddk: irreps {irreps!r} are not in 0e,1o,1e,0o order')
    return muls


def irrep_to_size(irreps):
    return sum(m * _DIM[k] for k, m in parse_irreps(irreps).items())


def layer_index(in_irreps, out_irreps, ns=24, nv=6):
    """Which conv layer of the score model has these irreps (get_irrep_seq, tensor_layers.py:12-27)."""
    seq = [dict(zip(_ORDER, m)) for m in ((ns, 0, 0, 0), (ns, nv, 0, 0), (ns, nv, nv, 0), (ns, nv, nv, ns))]
    i, o = parse_irreps(in_irreps), parse_irreps(out_irreps)
    for l in range(4):
        if seq[min(l, 3)] == i and seq[min(l + 1, 3)] == o:
            return l
    raise RuntimeError(f'ddk: no fused kernel for irreps {in_irreps} -> {out_irreps} (ns={ns}, nv={nv})')


_shape_ctx = {}


def _shape_context(device_index):
    """Context without weights: enough for ddk_tp_forward (shape-only layers)."""
    if device_index not in _shape_ctx:
        ctx = Context(device=device_index)
        ctx.finalize()
        _shape_ctx[device_index] = ctx
    return _shape_ctx[device_index]


class FasterTensorProduct(nn.Module):
    def __init__(self, in_irreps, sh_irreps, out_irreps, **kwargs):
        super().__init__()
        if parse_irreps(sh_irreps) != {'0e': 1, '1o': 1, '1e': 0, '0o': 0}:
            raise AssertionError("sh_irreps don't look like 1st order spherical harmonics")
        self.in_irreps, self.out_irreps = str(in_irreps), str(out_irreps)
        i, o = parse_irreps(in_irreps), parse_irreps(out_irreps)
        self.weight_shapes = {'0e': (i['0e'] + i['1o'], o['0e']), '1o': (i['0e'] + i['1o'] + i['1e'], o['1o']),
                              '1e': (i['1o'] + i['1e'] + i['0o'], o['1e']), '0o': (i['1e'] + i['0o'], o['0o'])}
        self.weight_numel = sum(a * b for a, b in self.weight_shapes.values())
        self.layer = layer_index(in_irreps, out_irreps)
        self.out_size = irrep_to_size(out_irreps)

    def forward(self, in_, sh, weight):
        if not in_.is_cuda:
            raise RuntimeError('ddk FasterTensorProduct runs on the GPU only (no CPU fallback)')
        lead = in_.shape[:-1]
        ctx = _shape_context(in_.device.index or 0)
        out = ctx.tp_forward(self.layer, in_.reshape(-1, in_.shape[-1]), sh.reshape(-1, 4),
                             weight.reshape(-1, self.weight_numel), self.out_size)
        return out.reshape(lead + (self.out_size,))


def FCBlock(in_dim, hidden_dim, out_dim, layers, dropout, activation='relu', batchnorm=False):
    """Same Sequential layout (indices 0 and 4 hold the Linears) as the reference models/layers.py:15-22."""
    if layers != 2 or activation != 'relu' or batchnorm:
        raise RuntimeError('ddk: only the 2-layer ReLU radial MLP is implemented')
    return nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.Identity(), nn.ReLU(), nn.Dropout(dropout), nn.Linear(hidden_dim, out_dim))


class _BatchNormParams(nn.Module):
    """Parameter holder with e3nn.nn.BatchNorm's state_dict keys (weight, bias, running_mean, running_var)."""

    def __init__(self, irreps):
        super().__init__()
        m = parse_irreps(irreps)
        nf, nsc = sum(m.values()), m['0e']
        self.register_buffer('running_mean', torch.zeros(nsc))
        self.register_buffer('running_var', torch.ones(nf))
        self.weight = nn.Parameter(torch.ones(nf))
        self.bias = nn.Parameter(torch.zeros(nsc))


class TensorProductConvLayer(nn.Module):
    def __init__(self, in_irreps, sh_irreps, out_irreps, n_edge_features, residual=True, batch_norm=True, dropout=0.0,
                 hidden_features=None, faster=False, edge_groups=1, tp_weights_layers=2, activation='relu'):
        super().__init__()
        if not faster or edge_groups != 4 or not residual:
            raise RuntimeError('ddk: the fused kernel implements faster=True, edge_groups=4, residual=True conv layers')
        self.in_irreps, self.out_irreps, self.sh_irreps = str(in_irreps), str(out_irreps), sh_irreps
        self.residual, self.edge_groups = residual, edge_groups
        self.out_size = irrep_to_size(out_irreps)
        hidden_features = n_edge_features if hidden_features is None else hidden_features
        self.tp = FasterTensorProduct(in_irreps, sh_irreps, out_irreps)
        self.layer = self.tp.layer
        if n_edge_features != 72 or hidden_features != 72:
            raise RuntimeError('ddk: radial MLP width must be 3*ns = 72')
        self.fc = nn.ModuleList([FCBlock(n_edge_features, hidden_features, self.tp.weight_numel, tp_weights_layers, dropout,
                                         activation) for _ in range(edge_groups)])
        self.batch_norm = _BatchNormParams(out_irreps) if batch_norm else None
        self._ctx, self._ctx_key = None, None

    def _context(self, device):
        key = (device.index or 0, tuple(int(p._version) for p in self.state_dict().values()))
        if self._ctx is None or self._ctx_key != key:
            ctx = Context(device=device.index or 0, batch_norm=int(self.batch_norm is not None))
            ctx.load_state_dict(self.state_dict(), prefix=f'conv_layers.{self.layer}.')
            self._ctx, self._ctx_key = ctx, key
        return self._ctx

    def forward(self, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None, reduce='mean'):
        if self.training:
            raise RuntimeError('ddk: inference (eval mode) only')
        if reduce != 'mean' or (out_nodes is not None and out_nodes != node_attr.shape[0]):
            raise RuntimeError("ddk: conv layers aggregate with reduce='mean' onto all nodes")
        if not node_attr.is_cuda:
            raise RuntimeError('ddk TensorProductConvLayer runs on the GPU only (no CPU fallback)')
        offs = [0]
        for ea in edge_attr:
            offs.append(offs[-1] + ea.shape[0])
        ea = torch.cat(list(edge_attr), dim=0)
        ctx = self._context(node_attr.device)
        return ctx.conv_forward(self.layer, node_attr, edge_index[0], edge_index[1], offs, ea, edge_sh, self.out_size)


class GaussianSmearing(nn.Module):
    """Distance embedding (tensor_layers.py:171-181); kept as a module so that state_dicts carry `offset`."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer('offset', offset)

    def forward(self, dist):
        dist = dist.view(-1, 1) - self.offset.view(1, -1)
        return torch.exp(self.coeff * torch.pow(dist, 2))
