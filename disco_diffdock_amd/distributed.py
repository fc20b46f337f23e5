"""Multi-GPU layout of the hot path: complexes are independent units -> shard them across ranks
(one process per GPU) with NO collective on the data path; one final gather of the poses over RCCL
(backend 'nccl' on ROCm) / gloo on CPU tests."""
import torch
import torch.distributed as dist


def shard_indices(costs, rank, world):
    """Greedy longest-processing-time partition: complexes sorted by descending cost (~ n_rec * n_lig), each
    assigned to the currently least loaded rank.  Deterministic; every rank computes the same assignment."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        loads[r] += costs[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_poses(poses, n_lig, samples, device):
    """poses: {complex index: tensor [samples, n_lig_i, 3]} held by this rank -> on every rank the full
    {index: tensor}.  One padded all_gather ([n_complexes, samples, max n_lig, 3] fp32, a few MB)."""
    n = len(n_lig)
    nmax = max(n_lig)
    buf = torch.zeros((n, samples, nmax, 3), dtype=torch.float32, device=device)
    own = torch.zeros(n, dtype=torch.float32, device=device)
    for i, p in poses.items():
        buf[i, :, :n_lig[i]] = p.to(device)
        own[i] = 1.0
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)      # every slot is written by exactly one rank
        dist.all_reduce(own, op=dist.ReduceOp.SUM)
    assert bool((own == 1).all()), 'every complex must be owned by exactly one rank'
    return {i: buf[i, :, :n_lig[i]].clone() for i in range(n)}


def gather_confidences(conf, n_complexes, device):
    """conf: {complex index: tensor [samples] or [samples, k]} (the confidence model's output for this rank's complexes,
    evaluate.py:317-325 ranks poses by it) -> on every rank the full {index: tensor}; rides with the pose gather (disjoint slots)."""
    shape = None
    for v in conf.values():
        shape = tuple(v.shape)
    meta = torch.zeros(3, dtype=torch.int64, device=device)       # a rank without complexes learns the shape from the others
    if shape is not None:
        meta[0], meta[1], meta[2] = len(shape), shape[0], (shape[1] if len(shape) > 1 else 1)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if multi:
        dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    nd, S, k = [int(v) for v in meta.tolist()]
    buf = torch.zeros((n_complexes, S, k), dtype=torch.float32, device=device)
    own = torch.zeros(n_complexes, dtype=torch.float32, device=device)
    for i, v in conf.items():
        buf[i] = v.to(device).float().reshape(S, k)
        own[i] = 1.0
    if multi:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        dist.all_reduce(own, op=dist.ReduceOp.SUM)
    assert bool((own == 1).all()), 'every complex must be owned by exactly one rank'
    return {i: (buf[i, :, 0] if nd == 1 else buf[i]).clone() for i in range(n_complexes)}


def shard_samples(n_samples, rank, world):
    """Large-pocket layout of SURVEY.md §8(e): the samples of ONE complex are independent given the latent, so rank r takes the
    contiguous slice [lo, hi) of the n_samples poses (receptor replicated on every rank, ``Complex(max_batch=hi-lo)``)."""
    base, extra = divmod(n_samples, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_samples(pos, n_samples, rank, world, device):
    """pos: this rank's [hi-lo, n_lig, 3] slice -> on every rank the full [n_samples, n_lig, 3] (disjoint slots, one all_reduce)."""
    lo, hi = shard_samples(n_samples, rank, world)
    buf = torch.zeros((n_samples,) + tuple(pos.shape[1:]), dtype=torch.float32, device=device)
    buf[lo:hi] = pos.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf
