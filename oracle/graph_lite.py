"""Minimal stand-in for the torch_geometric containers the reference sampler uses
(``HeteroData``, collation of a list of graphs into one batch, ``DataLoader``;
utils/sampling.py:5,56,65).  TEST INFRASTRUCTURE, PARITY UNPINNED (un-vendored dependency).

Only the behaviour the hot path relies on is restated: node tensors are concatenated
graph-major, edge indices are offset by the cumulative node counts, ``store.batch`` holds the
graph id per node, numpy attributes (``mask_rotate``) are collected into a python list, and
``num_graphs`` is set on the batch.
"""
import copy

import numpy as np
import torch


class Store:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, k):
        return k in self.__dict__

    def keys(self):
        return list(self.__dict__.keys())

    @property
    def num_nodes(self):
        for k in ('x', 'pos'):
            if k in self.__dict__:
                return self.__dict__[k].shape[0]
        raise AttributeError('num_nodes')

    @property
    def num_edges(self):
        return self.__dict__['edge_index'].shape[1]


def _norm_key(key):
    if isinstance(key, tuple):
        return (key[0], key[-1])
    return key


class HeteroData:
    def __init__(self):
        object.__setattr__(self, '_stores', {})

    def __getitem__(self, key):
        key = _norm_key(key)
        if key not in self._stores:
            self._stores[key] = Store()
        return self._stores[key]

    def __contains__(self, key):
        return _norm_key(key) in self._stores or key in self.__dict__

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
                elif isinstance(v, dict):
                    st.__dict__[k] = {a: (b.to(device) if torch.is_tensor(b) else b) for a, b in v.items()}
        for k, v in list(self.__dict__.items()):
            if k.startswith('_'):
                continue
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self

    def clone(self):
        return copy.deepcopy(self)


def collate(data_list):
    """Batch.from_data_list for graphs with identical schemas."""
    batch = HeteroData()
    first = data_list[0]
    offsets = {nt: [0] for nt in first.node_types}
    for nt in first.node_types:
        for d in data_list:
            offsets[nt].append(offsets[nt][-1] + d[nt].num_nodes)
    for nt in first.node_types:
        st = batch[nt]
        for k in first[nt].keys():
            vals = [d[nt].__dict__[k] for d in data_list]
            if torch.is_tensor(vals[0]):
                setattr(st, k, torch.cat(vals, dim=0))
            else:
                setattr(st, k, list(vals))
        st.batch = torch.cat([torch.full((d[nt].num_nodes,), i, dtype=torch.long) for i, d in enumerate(data_list)])
    for et in first.edge_types:
        st = batch[et]
        for k in first[et].keys():
            vals = [d[et].__dict__[k] for d in data_list]
            if k == 'edge_index':
                setattr(st, k, torch.cat([v + torch.tensor([[offsets[et[0]][i]], [offsets[et[1]][i]]], dtype=v.dtype)
                                          for i, v in enumerate(vals)], dim=1))
            elif torch.is_tensor(vals[0]):
                setattr(st, k, torch.cat(vals, dim=0))
            else:
                setattr(st, k, list(vals))
    for k, v in first.__dict__.items():
        if k.startswith('_'):
            continue
        vals = [d.__dict__[k] for d in data_list]
        if torch.is_tensor(v):
            batch.__dict__[k] = torch.cat(vals, dim=0)
        else:
            batch.__dict__[k] = list(vals)
    batch.num_graphs = len(data_list)
    return batch


class DataLoader:
    def __init__(self, data_list, batch_size=1, shuffle=False, **kw):
        assert not shuffle
        # (a None dataset is legal until the first batch is asked for, as with torch's DataLoader: utils/sampling.py:61 builds
        # iter(DataLoader(confidence_data_list)) even when confidence_data_list is None)
        self.data_list, self.batch_size = (list(data_list) if data_list is not None else None), batch_size

    def __len__(self):
        return (len(self.data_list) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.data_list), self.batch_size):
            yield collate(self.data_list[i:i + self.batch_size])


def make_complex(lig_x, lig_pos, bond_index, bond_attr, edge_mask, mask_rotate, rec_x, rec_pos, rec_edge_index,
                 original_center=None, name='synthetic'):
    """Assemble one complex in the layout ``datasets_utils/process_mols.py`` emits (Appendix B.1)."""
    d = HeteroData()
    d['ligand'].x = torch.as_tensor(lig_x).long()
    d['ligand'].pos = torch.as_tensor(lig_pos).float()
    d['ligand'].edge_mask = torch.as_tensor(edge_mask).bool()
    d['ligand'].mask_rotate = np.asarray(mask_rotate, dtype=bool)
    d['ligand', 'lig_bond', 'ligand'].edge_index = torch.as_tensor(bond_index).long()
    d['ligand', 'lig_bond', 'ligand'].edge_attr = torch.as_tensor(bond_attr).float()
    d['receptor'].x = torch.as_tensor(rec_x).float()
    d['receptor'].pos = torch.as_tensor(rec_pos).float()
    d['receptor', 'rec_contact', 'receptor'].edge_index = torch.as_tensor(rec_edge_index).long()
    d.original_center = torch.zeros(1, 3) if original_center is None else torch.as_tensor(original_center).float()
    d.name = name
    return d


def add_atoms(d, atom_x, atom_pos, atom_edge_index, atom_rec_index):
    """All-atom level of a complex (datasets_utils/process_mols.py:474-477)."""
    d['atom'].x = torch.as_tensor(atom_x).long()
    d['atom'].pos = torch.as_tensor(atom_pos).float()
    d['atom', 'atom_contact', 'atom'].edge_index = torch.as_tensor(atom_edge_index).long()
    d['atom', 'atom_rec_contact', 'receptor'].edge_index = torch.as_tensor(atom_rec_index).long()
    return d
