"""Restatement of ``torch_scatter.scatter`` / ``scatter_mean``.  TEST INFRASTRUCTURE, PARITY UNPINNED
(un-vendored dependency; call sites models/tensor_layers.py:159, models/score_model.py:265).
mean = sum / max(count, 1)."""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    dim_size = int(dim_size)
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, index.long(), src)
    if reduce in ('sum', 'add'):
        return res
    if reduce == 'mean':
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index.long(), torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
        cnt = cnt.clamp(min=1)
        return res / cnt.reshape((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce='mean')
