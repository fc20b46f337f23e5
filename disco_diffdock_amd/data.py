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

    @property
    def node_types(self):
        return [k for k in self._stores if not isinstance(k, tuple)]

    @property
    def edge_types(self):
        return [k for k in self._stores if isinstance(k, tuple)]

    def to(self, device):
        for st in self._stores.values():
            for k, v in list(st.__dict__.items()):
                if torch.is_tensor(v):
                    st.__dict__[k] = v.to(device)
        for k, v in list(self.__dict__.items()):
            if not k.startswith('_') and torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def clone(self):
        return copy.deepcopy(self)

    def __copy__(self):
        """shallow copy with its own storages (tensors are shared): rebinding attributes on the copy leaves the original intact"""
        new = HeteroData()
        for k, st in self._stores.items():
            new._stores[k] = Storage(**st.__dict__)
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


def collate(data_list):
    batch = HeteroData()
    first = data_list[0]
    offs = {nt: np.cumsum([0] + [d[nt].num_nodes for d in data_list]) for nt in first.node_types}
    for nt in first.node_types:
        st = batch[nt]
        for k in first[nt].keys():
            vals = [getattr(d[nt], k) for d in data_list]
            setattr(st, k, torch.cat(vals, 0) if torch.is_tensor(vals[0]) else list(vals))
        st.batch = torch.cat([torch.full((d[nt].num_nodes,), i, dtype=torch.long) for i, d in enumerate(data_list)])
    for et in first.edge_types:
        st = batch[et]
        for k in first[et].keys():
            vals = [getattr(d[et], k) for d in data_list]
            if k == 'edge_index':
                vals = [v + torch.tensor([[int(offs[et[0]][i])], [int(offs[et[1]][i])]], dtype=v.dtype, device=v.device)
                        for i, v in enumerate(vals)]
                setattr(st, k, torch.cat(vals, 1))
            else:
                setattr(st, k, torch.cat(vals, 0) if torch.is_tensor(vals[0]) else list(vals))
    for k, v in first.__dict__.items():
        if k.startswith('_'):
            continue
        vals = [d.__dict__[k] for d in data_list]
        batch.__dict__[k] = torch.cat(vals, 0) if torch.is_tensor(v) else list(vals)
    batch.num_graphs = len(data_list)
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
