"""Shared helpers for the parity tests (oracle-side graph assembly from a complex dict)."""
import numpy as np
import torch

from oracle import graph_lite


def complex_from_npz(z):
    return {k: z[k] for k in z.files}


def to_graph(c, loader_style=True):
    g = graph_lite.make_complex(c['lig_x'], c['lig_pos'], c['bond_index'], c['bond_attr'], c['edge_mask'],
                                c['mask_rotate'], c['rec_x'], c['rec_pos'], c['rec_edge_index'], c['original_center'])
    if loader_style:
        g['ligand'].mask_rotate = [g['ligand'].mask_rotate]
    return g


def batch_of(c, B, pos=None):
    b = graph_lite.collate([to_graph(c) for _ in range(B)])
    if pos is not None:
        b['ligand'].pos = torch.as_tensor(pos).float().reshape(-1, 3)
    return b


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
