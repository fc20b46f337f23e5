"""Restatement of ``torch_cluster.radius`` / ``radius_graph``.  TEST INFRASTRUCTURE, PARITY UNPINNED.

torch_cluster is an un-vendored, un-pinned dependency of the reference
(models/score_model.py:5; call sites :315, :379-384, :430).  Semantics restated from its
published behaviour (CUDA kernel): for every query point y_i, the x_j of the same batch
element with |x_j - y_i|^2 < r^2 (strict), in ascending j, truncated to the first
``max_num_neighbors``; ``radius`` returns ``[y_index; x_index]``.  ``radius_graph`` calls
``radius(x, x, r, batch, batch, max_num_neighbors + 1)``, flips to [neighbour; centre]
(flow='source_to_target') and drops self loops.  Which neighbours survive when the cap binds
is backend specific in the reference (CPU path = kd-tree order); the synthetic workloads never
bind the cap.
"""
import torch


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    rows, cols = [], []
    r2 = float(r) * float(r)
    nb = int(max(batch_x.max().item() if len(batch_x) else -1, batch_y.max().item() if len(batch_y) else -1)) + 1
    for b in range(nb):
        ix = torch.nonzero(batch_x == b).flatten()
        iy = torch.nonzero(batch_y == b).flatten()
        if len(ix) == 0 or len(iy) == 0:
            continue
        d2 = ((y[iy][:, None, :] - x[ix][None, :, :]) ** 2).sum(-1)
        m = d2 < r2
        # ascending x index, first max_num_neighbors per query
        rank = torch.cumsum(m.long(), dim=1)
        m = m & (rank <= max_num_neighbors)
        yy, xx = torch.nonzero(m, as_tuple=True)
        rows.append(iy[yy])
        cols.append(ix[xx])
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)], 0)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow='source_to_target'):
    ei = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    if flow == 'source_to_target':
        row, col = ei[1], ei[0]
    else:
        row, col = ei[0], ei[1]
    if not loop:
        mask = row != col
        row, col = row[mask], col[mask]
    return torch.stack([row, col], 0)
