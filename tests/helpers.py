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


def elem_err(a, b, floor=1e-3):
    """max over ELEMENTS of |a - b| / max(|b|, floor * max|b|): every element is measured against its own magnitude (elements below
    ``floor`` of the largest one against that floor, since a value that is zero by cancellation has no scale of its own).  The global
    ``rel_err`` lets one large element hide the errors of the small ones."""
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    if b.size == 0:
        return 0.0
    scale = np.maximum(np.abs(b), floor * np.abs(b).max() + 1e-30)
    return float((np.abs(a - b) / scale).max())


def chan_err(a, b):
    """max over feature channels of max|a - b| / max|b| of THAT channel (channels smaller than 1e-3 of the largest one are
    measured against 1e-3 of the largest: an all-zero padded channel has no scale of its own)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.abs(b).max(axis=0)
    scale = np.maximum(scale, 1e-3 * scale.max())
    return float((np.abs(a - b).max(axis=0) / scale).max())
