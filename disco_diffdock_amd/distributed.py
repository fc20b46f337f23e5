"""Multi-GPU layout of the hot path: complexes are independent units -> shard them across ranks
(one process per GPU) with NO collective on the data path; one final all_gather of the poses over RCCL
(backend 'nccl' on ROCm) / gloo on CPU tests."""
import torch
import torch.distributed as dist


def complex_cost(n_res, n_lig):
    """Relative cost of one complex (40 samples x 20 reverse steps) for the partition below: affine in the receptor size with a ligand term, fitted to the
    per-complex device times of bench.py --config 4 --complexes 363 on the timesplit-shaped set (profiles/r06_bench_config4_363.json: per-decile table).
    The rec-rec messages (24 per residue and sample) dominate; the cross edges grow with n_lig x the residues within the cutoff; a constant covers the AR
    passes' and the small launches' share.  Only ratios matter."""
    return 1.0 + n_res / 175.0 + max(n_lig, 16) / 300.0      # (per-decile fit: ~18 ms + 0.104 ms per residue; the ligand term is small beside it)


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


FORCE_COLLECTIVES = False     # run the collectives of the path on a world_size-1 group too (bench.py --force-dist, the one-GPU RCCL test)


def _multi():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def _all_gather_rows(rows, idx, device):
    """rows [k, ...] with global row ids idx [k] held by this rank -> (rows, ids) of every rank.  One all_gather of a buffer padded
    to the largest per-rank count (+ one of the ids): each byte crosses each link once, half the traffic of the all_reduce of a
    zero-padded full-size buffer this replaced."""
    k = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    ks = [torch.zeros_like(k) for _ in range(dist.get_world_size())]
    dist.all_gather(ks, k)
    kmax = max(int(v.item()) for v in ks)
    buf = torch.zeros((kmax,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=device)
    ids = torch.full((kmax,), -1, dtype=torch.int64, device=device)
    buf[:rows.shape[0]] = rows
    ids[:rows.shape[0]] = idx
    bufs = [torch.zeros_like(buf) for _ in ks]
    idss = [torch.zeros_like(ids) for _ in ks]
    dist.all_gather(bufs, buf)
    dist.all_gather(idss, ids)
    return torch.cat(bufs), torch.cat(idss)


def gather_poses(poses, n_lig, samples, device):
    """poses: {complex index: tensor [samples, n_lig_i, 3]} held by this rank -> on every rank the full {index: tensor}.
    The ONE exchange of the path: an all_gather of the ranks' padded pose blocks ([k_rank, samples, max n_lig, 3] fp32, a few MB)."""
    n = len(n_lig)
    nmax = max(n_lig)
    own = sorted(poses)
    rows = torch.zeros((len(own), samples, nmax, 3), dtype=torch.float32, device=device)
    for r, i in enumerate(own):
        rows[r, :, :n_lig[i]] = poses[i].to(device)
    ids = torch.tensor(own, dtype=torch.int64, device=device)
    if _multi():
        rows, ids = _all_gather_rows(rows, ids, device)
    out = {}
    for r, i in enumerate(ids.tolist()):
        if i >= 0:
            assert i not in out, 'every complex must be owned by exactly one rank'
            out[i] = rows[r, :, :n_lig[i]].clone()
    assert len(out) == n, 'every complex must be owned by exactly one rank'
    return out


def gather_confidences(conf, n_complexes, device):
    """conf: {complex index: tensor [samples] or [samples, k]} (the confidence model's output for this rank's complexes,
    evaluate.py:317-325 ranks poses by it) -> on every rank the full {index: tensor}; same all_gather as the poses."""
    shape = None
    for v in conf.values():
        shape = tuple(v.shape)
    meta = torch.zeros(3, dtype=torch.int64, device=device)       # a rank without complexes learns the shape from the others
    if shape is not None:
        meta[0], meta[1], meta[2] = len(shape), shape[0], (shape[1] if len(shape) > 1 else 1)
    if _multi():
        dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    nd, S, k = [int(v) for v in meta.tolist()]
    own = sorted(conf)
    rows = torch.zeros((len(own), S, k), dtype=torch.float32, device=device)
    for r, i in enumerate(own):
        rows[r] = conf[i].to(device).float().reshape(S, k)
    ids = torch.tensor(own, dtype=torch.int64, device=device)
    if _multi():
        rows, ids = _all_gather_rows(rows, ids, device)
    out = {i: (rows[r, :, 0] if nd == 1 else rows[r]).clone() for r, i in enumerate(ids.tolist()) if i >= 0}
    assert len(out) == n_complexes, 'every complex must be owned by exactly one rank'
    return out


def shard_samples(n_samples, rank, world):
    """Large-pocket layout of SURVEY.md §8(e): the samples of ONE complex are independent given the latent, so rank r takes the
    contiguous slice [lo, hi) of the n_samples poses (receptor replicated on every rank, ``Complex(max_batch=hi-lo)``)."""
    base, extra = divmod(n_samples, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_samples(pos, n_samples, rank, world, device):
    """pos: this rank's [hi-lo, n_lig, 3] slice -> on every rank the full [n_samples, n_lig, 3] (one all_gather of equally padded slices)."""
    lo, hi = shard_samples(n_samples, rank, world)
    if not _multi():
        return pos.to(device).float()
    per = (n_samples + world - 1) // world
    buf = torch.zeros((per,) + tuple(pos.shape[1:]), dtype=torch.float32, device=device)
    buf[:hi - lo] = pos.to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = []
    for r in range(world):
        l, h = shard_samples(n_samples, r, world)
        out.append(bufs[r][:h - l])
    return torch.cat(out)
