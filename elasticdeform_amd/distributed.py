"""
Multi-GPU: shard a batch of independent volumes across the ranks of a ``torch.distributed`` job,
one process per GPU (backend "nccl" = RCCL over xGMI on MI355X nodes, "gloo" in CPU tests).

The hot path partitions by volume: every output voxel depends on one input
volume and one displacement grid only, and the gradient scatter never leaves its volume.  So a
rank deforms the contiguous slice ``[lo, hi)`` of the batch with no halo, no reduction and NO
data-path collective.  The only communication offered here is the optional hand-back of results
to one rank (``gather_to``), a plain gather of outputs; the reference has nothing comparable (it
is single-process).
"""
from __future__ import absolute_import


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced slice [lo, hi) of ``n_items`` volumes for ``rank``: the first
    ``n_items % world_size`` ranks get one extra volume."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError('invalid rank %r / world size %r' % (rank, world_size))
    base, extra = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def _dist():
    import torch.distributed as dist
    return dist


def deform_batch(volumes, displacements, rank=None, world_size=None, gather_to=None, group=None,
                 compute=None, **kwargs):
    """
    Deform a batch of independent volumes, sharded over the ranks of the current process group.

    volumes        sequence of arrays (the whole batch, or a callable ``i -> array`` that
                   materialises volume ``i`` on demand so that a rank only ever touches its shard)
    displacements  sequence (or callable) of per-volume displacement grids
    gather_to      None: return this rank's outputs only (the data-loader case: no communication);
                   an int: additionally gather every rank's outputs on that rank
                   (``gather_object``, outputs moved to host) and return the full list there
    compute        the per-volume function; defaults to ``elasticdeform_amd.deform_grid``
    kwargs         forwarded to it (order, mode, cval, crop, prefilter, axis, affine, ...)

    Returns ``(indices, outputs)`` for this rank, or on ``gather_to`` the full ordered list.
    """
    if compute is None:
        from . import deform_grid as compute
    if rank is None or world_size is None:
        dist = _dist()
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank(group) if rank is None else rank
            world_size = dist.get_world_size(group) if world_size is None else world_size
        else:
            rank, world_size = 0, 1
    n = len(volumes) if hasattr(volumes, '__len__') else kwargs.pop('n_items')
    lo, hi = shard_bounds(n, rank, world_size)
    get_v = volumes if callable(volumes) else volumes.__getitem__
    get_d = displacements if callable(displacements) else displacements.__getitem__
    outs = [compute(get_v(i), get_d(i), **kwargs) for i in range(lo, hi)]
    idx = list(range(lo, hi))
    if gather_to is None or world_size == 1:
        return (idx, outs) if gather_to is None else outs
    dist = _dist()
    host = [o.cpu() if hasattr(o, 'cpu') else o for o in outs]
    bucket = [None] * world_size if rank == gather_to else None
    dist.gather_object((idx, host), bucket, dst=gather_to, group=group)
    if rank != gather_to:
        return None
    full = [None] * n
    for ids, vals in bucket:
        for i, val in zip(ids, vals):
            full[i] = val
    return full
