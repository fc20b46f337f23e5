"""Minimal heterogeneous-graph containers for the sampler's call surface (the reference uses torch_geometric's
HeteroData / Batch / DataLoader, utils/sampling.py:5,56; none of that is a dependency here).  Any object with the
same accessors (``d['ligand'].pos``, ``d['ligand','ligand'].edge_index``, ...) - including real PyG graphs - works
with :func:`disco_diffdock_amd.sampling.sampling`."""
import copy

import numpy as np
import torch


class Storage:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, k):
        return k in self.__dict__

    def keys(self):
        return list(self.__dict__)

    @property
    def num_nodes(self):
        for k in ('x', 'pos'):
            if k in self.__dict__:
                return self.__dict__[k].shape[0]
        raise AttributeError('num_nodes')

    @property
    def num_edges(self):
        return self.__dict__['edge_index'].shape[1]


class HeteroData:
    """data['ligand'], data['receptor'], data['ligand', 'ligand'] (== data['ligand','lig_bond','ligand']), ..."""

    def __init__(self):
        object.__setattr__(self, '_stores', {})

    @staticmethod
    def _key(key):
        return (key[0], key[-1]) if isinstance(key, tuple) else key

    def __getitem__(self, key):
        key = self._key(key)
        if key not in self._stores:
            self._stores[key] = Storage()
        return self._stores[key]

    def __contains__(self, key):
        return self._key(key) in self._stores or key in self.__dict__

    def __getattr__(self, name):
        """only reached for attributes that are not set: results that sampling() fills in on first access (``latent_str`` /
        ``latent_pos`` come from a device read-back that is not awaited inside sampling(), see sampling._Bookkeeping)"""
        lazy = self.__dict__.get('_lazy')
        if lazy is not None and not name.startswith('_'):
            lazy.resolve()
            if name in self.__dict__:
                return self.__dict__[name]
        raise AttributeError(name)

    @property
    def node_types(self):
        return [k for k in self._stores if not isinstance(k, tuple)]

    @property
    def edge_types(self):
        return [k for k in self._stores if isinstance(k, tuple)]

    def to(self, device):
        for st in self._stores.values():
            if isinstance(st, _LazyStorage):
                st.materialise_all()
            for k, v in list(st.__dict__.items()):
                if torch.is_tensor(v):
                    st.__dict__[k] = v.to(device)
        for k, v in list(self.__dict__.items()):
            if not k.startswith('_') and torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def clone(self):
        return copy.deepcopy(self)

    def _settle(self):
        """fill the results sampling() left pending on this graph (one device wait): a copy, a deep copy or a pickle must carry values, not
        the bookkeeping object with its CUDA event"""
        lazy = self.__dict__.pop('_lazy', None)
        if lazy is not None:
            self.__dict__['_lazy'] = lazy
            lazy.resolve()
            self.__dict__.pop('_lazy', None)

    def __getstate__(self):
        self._settle()
        return dict(self.__dict__)

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __deepcopy__(self, memo):
        self._settle()
        new = HeteroData.__new__(HeteroData)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __copy__(self):
        """shallow copy with its own storages (tensors are shared): rebinding attributes on the copy leaves the original intact"""
        self._settle()
        new = HeteroData()
        for k, st in self._stores.items():
            cp = type(st).__new__(type(st))          # keeps a collated batch's stores lazy
            cp.__dict__.update(st.__dict__)
            new._stores[k] = cp
        for k, v in self.__dict__.items():
            if not k.startswith('_'):
                new.__dict__[k] = v
        return new


def from_arrays(c, loader_style=True):
    """Build one complex graph from the array dict of :mod:`disco_diffdock_amd.synthetic` (layout of SURVEY.md B.1).
    loader_style=True wraps mask_rotate in a 1-element list like the batch_size=1 loader of evaluate.py:138 does."""
    d = HeteroData()
    d['ligand'].x = torch.as_tensor(c['lig_x']).long()
    d['ligand'].pos = torch.as_tensor(c['lig_pos']).float()
    d['ligand'].edge_mask = torch.as_tensor(c['edge_mask']).bool()
    mr = np.asarray(c['mask_rotate'], dtype=bool)
    d['ligand'].mask_rotate = [mr] if loader_style else mr
    d['ligand', 'lig_bond', 'ligand'].edge_index = torch.as_tensor(c['bond_index']).long()
    d['ligand', 'lig_bond', 'ligand'].edge_attr = torch.as_tensor(c['bond_attr']).float()
    d['receptor'].x = torch.as_tensor(c['rec_x']).float()
    d['receptor'].pos = torch.as_tensor(c['rec_pos']).float()
    d['receptor', 'rec_contact', 'receptor'].edge_index = torch.as_tensor(c['rec_edge_index']).long()
    d.original_center = torch.as_tensor(c.get('original_center', np.zeros((1, 3), np.float32))).float()
    d.name = c.get('name', 'complex')
    if 'atom_x' in c:      # all-atom receptor level of the confidence model's graphs (process_mols.py:474-477)
        d['atom'].x = torch.as_tensor(c['atom_x']).long()
        d['atom'].pos = torch.as_tensor(c['atom_pos']).float()
        d['atom', 'atom_contact', 'atom'].edge_index = torch.as_tensor(c['atom_edge_index']).long()
        d['atom', 'atom_rec_contact', 'receptor'].edge_index = torch.as_tensor(c['atom_rec_index']).long()
    return d


class _LazyStorage(Storage):
    """Store of a collated batch: an attribute is concatenated over the graphs the first time it is read.  A sampling() batch holds
    copies of ONE complex (utils/sampling.py:57), whose receptor features ([n_rec, 1281] per copy) the device path never reads from
    the batch - eager collation spent 10 ms per 40-sample batch copying them."""

    def __init__(self, srcs, kind, off_src=None, off_dst=None):
        self.__dict__['_srcs'] = srcs
        self.__dict__['_kind'] = kind
        self.__dict__['_off'] = (off_src, off_dst)

    def _materialise(self, k):
        srcs, kind = self.__dict__['_srcs'], self.__dict__['_kind']
        if k == 'batch' and kind == 'node':
            v = torch.cat([torch.full((st.num_nodes,), i, dtype=torch.long) for i, st in enumerate(srcs)])
        elif k in self._src_keys():
            vals = [getattr(st, k) for st in srcs]     # (generic accessors: the sources may be PyG storages)
            if k == 'edge_index' and kind == 'edge':
                o0, o1 = self.__dict__['_off']
                vals = [v + torch.tensor([[int(o0[i])], [int(o1[i])]], dtype=v.dtype, device=v.device) for i, v in enumerate(vals)]
                v = torch.cat(vals, 1)
            else:
                v = torch.cat(vals, 0) if torch.is_tensor(vals[0]) else list(vals)
        else:
            raise AttributeError(k)
        self.__dict__[k] = v
        return v

    def _src_keys(self):
        return [k for k in self.__dict__['_srcs'][0].keys() if not str(k).startswith('_')]

    def __getattr__(self, k):          # only reached when k is not materialised yet
        if k.startswith('_'):
            raise AttributeError(k)
        return self._materialise(k)

    def __contains__(self, k):
        return k in self.__dict__ or k in self._src_keys() or (k == 'batch' and self.__dict__['_kind'] == 'node')

    def keys(self):
        ks = self._src_keys()
        ks += [k for k in self.__dict__ if not k.startswith('_') and k not in ks]
        if self.__dict__['_kind'] == 'node' and 'batch' not in ks:
            ks.append('batch')
        return ks

    def materialise_all(self):
        for k in self.keys():
            getattr(self, k)
        return self

    @property
    def num_nodes(self):
        return sum(st.num_nodes for st in self.__dict__['_srcs'])

    @property
    def num_edges(self):
        return sum(st.num_edges for st in self.__dict__['_srcs'])


def collate(data_list):
    """Batch of graphs with the accessors of a PyG ``Batch`` (concatenated node/edge attributes, offset edge_index, ``.batch``,
    ``num_graphs``); attributes are concatenated lazily on first access.  ``batch.first`` is the first graph itself."""
    batch = HeteroData()
    first = data_list[0]
    offs = {nt: np.cumsum([0] + [d[nt].num_nodes for d in data_list]) for nt in first.node_types}
    for nt in first.node_types:
        batch._stores[nt] = _LazyStorage([d[nt] for d in data_list], 'node')
    for et in first.edge_types:
        batch._stores[HeteroData._key(et)] = _LazyStorage([d[et] for d in data_list], 'edge', offs[et[0]], offs[et[-1]])   # PyG edge types are 3-tuples
    for k, v in first.__dict__.items():
        if k.startswith('_'):
            continue
        vals = [d.__dict__[k] for d in data_list]
        batch.__dict__[k] = torch.cat(vals, 0) if torch.is_tensor(v) else list(vals)
    batch.num_graphs = len(data_list)
    batch.first = first
    return batch


class DataLoader:
    def __init__(self, data_list, batch_size=1, shuffle=False, **kw):
        if shuffle:
            raise RuntimeError('inference loader does not shuffle')
        self.data_list, self.batch_size = list(data_list), batch_size

    def __len__(self):
        return (len(self.data_list) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.data_list), self.batch_size):
            yield collate(self.data_list[i:i + self.batch_size])
