"""
Multi-GPU: shard a batch of independent volumes across the ranks of a ``torch.distributed`` job,
one process per GPU (backend "nccl" = RCCL over xGMI on MI355X nodes, "gloo" in CPU tests).

The hot path partitions by volume: every output voxel depends on one input
volume and one displacement grid only, and the gradient scatter never leaves its volume.  So a
rank deforms the contiguous slice ``[lo, hi)`` of the batch with no halo, no reduction and NO
data-path collective; on its shard it runs the single-launch batch kernels
(``deform_grid_batch`` / ``deform_grid_gradient_batch``).

Two entry points:

* :func:`deform_batch_sharded` -- tensors.  The data-loader case (every rank already holds its
  shard) needs no communication at all.  When the whole batch lives on ONE rank
  (``scatter_from=r``), the control grids are broadcast (KBs) and the volumes travel as device
  tensors, point to point: ``r`` posts one ``isend`` per peer and every peer one ``irecv`` in a
  single ``batch_isend_irecv`` group, so the 7 xGMI links of GPU ``r`` run concurrently
  (xGMI is point to point -- a ring or tree collective would be per-link bound).  ``gather_to=r``
  hands the outputs back the same way.  No ``.cpu()``, no pickling: with the "nccl" (RCCL) backend
  the tensors never leave HBM.  (Under "gloo" -- CPU tests, or two test ranks sharing one GPU --
  point-to-point transfers of CUDA tensors are staged through the host, because gloo itself has no
  device send/recv.)
* :func:`deform_batch` -- sequences / callables of per-volume arrays with an injectable per-volume
  ``compute`` (the round-1 interface, kept).

The reference has nothing comparable: it is single-process and its only "many volumes" mechanism
is the step-axis loop (deform.c:405-436,828-838).  No 1 -> 8 GPU scaling curve has been measured
for this module (the build container has no multi-GPU node).
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


def _rank_world(rank, world_size, group):
    if rank is None or world_size is None:
        dist = _dist()
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank(group) if rank is None else rank
            world_size = dist.get_world_size(group) if world_size is None else world_size
        else:
            rank, world_size = 0, 1
    return rank, world_size


def _global_rank(r, group):
    dist = _dist()
    return dist.get_global_rank(group, r) if group is not None else r


def _host_staged(group):
    """gloo has no send / recv for CUDA tensors: stage those through the host (tests only)."""
    return _dist().get_backend(group) == 'gloo'


def _p2p_exchange(sends, recvs, group, device=None):
    """One ``batch_isend_irecv`` group: ``sends`` = [(tensor, peer)], ``recvs`` = [(tensor, peer)]
    (group ranks).  All transfers of the group are in flight together.  Under RCCL ("nccl") every
    tensor handed to the backend must live on the GPU: a host-resident source (the root's batch as it
    came from a data loader) is moved to ``device`` slice by slice, a host-resident destination is
    filled through a device buffer.  gloo (the CPU tests) goes the other way: through the host."""
    import torch
    dist = _dist()
    if not sends and not recvs:
        return
    staged = _host_staged(group)
    if device is None and not staged:
        device = torch.device('cuda', torch.cuda.current_device())
    ops, back = [], []
    for t, peer in sends:
        w = t.contiguous()
        if staged and w.is_cuda:
            w = w.cpu()
        elif not staged and w.device.type != device.type:
            w = w.to(device, non_blocking=True)
        ops.append(dist.P2POp(dist.isend, w, _global_rank(peer, group), group))
    for t, peer in recvs:
        if staged and t.is_cuda:
            w = torch.empty(t.shape, dtype=t.dtype)
            back.append((t, w))
        elif not staged and t.device.type != device.type:
            w = torch.empty(t.shape, dtype=t.dtype, device=device)
            back.append((t, w))
        else:
            w = t
        ops.append(dist.P2POp(dist.irecv, w, _global_rank(peer, group), group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for t, w in back:
        t.copy_(w)


def scatter_batch(full, src, device, rank=None, world_size=None, group=None):
    """The contiguous shard of a stacked tensor ``full`` (B, ...) that lives on rank ``src``:
    returns this rank's ``full[lo:hi]`` on ``device``.  Other ranks pass ``full=None``; shape and
    dtype are announced with one small object broadcast, the data moves point to point."""
    import torch
    dist = _dist()
    rank, world = _rank_world(rank, world_size, group)
    meta = [(tuple(full.shape), full.dtype) if rank == src else None]
    dist.broadcast_object_list(meta, src=_global_rank(src, group), group=group)
    shape, dtype = meta[0]
    lo, hi = shard_bounds(shape[0], rank, world)
    if rank == src:
        sends = []
        for peer in range(world):
            plo, phi = shard_bounds(shape[0], peer, world)
            if peer != src and phi > plo:
                sends.append((full[plo:phi], peer))
        _p2p_exchange(sends, [], group, device)
        return full[lo:hi].to(device)
    mine = torch.empty((hi - lo,) + tuple(shape[1:]), dtype=dtype, device=device)
    if hi > lo:
        _p2p_exchange([], [(mine, src)], group, device)
    return mine


def gather_batch(mine, n_items, dst, rank=None, world_size=None, group=None):
    """Inverse of :func:`scatter_batch`: every rank's shard ends up in one stacked tensor on rank
    ``dst`` (returned there; ``None`` elsewhere)."""
    import torch
    rank, world = _rank_world(rank, world_size, group)
    dev = mine.device if mine.device.type != 'cpu' else None
    if rank != dst:
        if mine.shape[0] > 0:
            _p2p_exchange([(mine, dst)], [], group, dev)
        return None
    full = torch.empty((n_items,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    recvs = []
    for peer in range(world):
        plo, phi = shard_bounds(n_items, peer, world)
        if peer == dst:
            full[plo:phi].copy_(mine)
        elif phi > plo:
            recvs.append((full[plo:phi], peer))
    _p2p_exchange([], recvs, group, dev)
    return full


def deform_batch_sharded(X, displacements, gradient=False, scatter_from=None, gather_to=None,
                         device=None, rank=None, world_size=None, group=None, compute=None, **kwargs):
    """
    Deform a batch of independent volumes, sharded over the ranks of the current process group.

    X              stacked volumes ``(B, ...)`` (``dY`` for ``gradient=True``).  Without
                   ``scatter_from``: THIS RANK'S SHARD, already where it is needed (the data-loader
                   case -- no communication).  With ``scatter_from=r``: the whole batch on rank
                   ``r`` (``None`` on the other ranks).
    displacements  ``(B, naxis, n_0, ...)`` control grids, sharded / rooted like ``X``.  With
                   ``scatter_from`` they are broadcast whole (a few KB) and sliced locally.
    gather_to      None: return this rank's output shard.  An int: additionally gather every shard
                   on that rank (device tensors, point to point) and return the full batch there,
                   ``None`` elsewhere.
    compute        ``(X_shard, D_shard, **kwargs) -> Y_shard``; defaults to
                   ``elasticdeform_amd.deform_grid_batch`` / ``deform_grid_gradient_batch`` (one set
                   of kernel launches for the whole shard).
    kwargs         forwarded to it (order, mode, cval, crop, prefilter, axis, X_shape, ...).
    """
    import torch
    rank, world = _rank_world(rank, world_size, group)
    if compute is None:
        from . import deform_grid_batch, deform_grid_gradient_batch
        compute = deform_grid_gradient_batch if gradient else deform_grid_batch
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() \
            else torch.device('cpu')
    n_items = None
    if scatter_from is not None and world > 1:
        dist = _dist()
        if rank == scatter_from:
            X = torch.as_tensor(X)
            D = torch.as_tensor(displacements).to(device)
        Xs = scatter_batch(X if rank == scatter_from else None, scatter_from, device, rank, world, group)
        meta = [(tuple(D.shape), D.dtype) if rank == scatter_from else None]
        dist.broadcast_object_list(meta, src=_global_rank(scatter_from, group), group=group)
        if rank != scatter_from:
            D = torch.empty(meta[0][0], dtype=meta[0][1], device=device)
        dist.broadcast(D, src=_global_rank(scatter_from, group), group=group)
        n_items = int(D.shape[0])
        lo, hi = shard_bounds(n_items, rank, world)
        Ds = D[lo:hi]
    else:
        Xs = torch.as_tensor(X).to(device)
        Ds = torch.as_tensor(displacements).to(device)
    out = compute(Xs, Ds, **kwargs) if Xs.shape[0] > 0 else None
    if out is None:
        # an empty shard (more ranks than volumes): shape of one output sample from a peer is not
        # known here, so hand back an empty tensor with the input's trailing shape
        out = Xs.new_empty((0,) + tuple(Xs.shape[1:]))
    if gather_to is None or world == 1:
        return out
    if n_items is None:
        dist = _dist()
        counts = [None] * world
        dist.all_gather_object(counts, int(Xs.shape[0]), group=group)
        n_items = int(sum(counts))
        # the shards must be the contiguous balanced partition for the gather to know the sizes
        if [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0]
                for r in range(world)] != counts:
            raise ValueError('gather_to needs the contiguous balanced sharding of shard_bounds()')
    # decided from values every rank holds, BEFORE any rank enters a send / recv: an exception on
    # one rank only would leave the others waiting in the exchange
    glo, ghi = shard_bounds(n_items, gather_to, world)
    if ghi <= glo:
        raise ValueError('the gathering rank must own at least one volume')
    if out.shape[0] == 0:
        return None
    return gather_batch(out, n_items, gather_to, rank, world, group)


def deform_batch(volumes, displacements, rank=None, world_size=None, gather_to=None, group=None,
                 compute=None, **kwargs):
    """
    Per-volume variant: ``volumes`` / ``displacements`` are sequences (or callables ``i -> array``
    that materialise item ``i`` on demand, with ``n_items=`` given) and ``compute`` is applied to
    one volume at a time (default ``elasticdeform_amd.deform_grid``).  Returns ``(indices,
    outputs)`` for this rank, or with ``gather_to`` the full ordered list on that rank (``None``
    elsewhere) -- outputs of equal shape travel as one stacked tensor through
    :func:`gather_batch`, anything else through ``gather_object``.
    """
    if compute is None:
        from . import deform_grid as compute
    rank, world_size = _rank_world(rank, world_size, group)
    n = len(volumes) if hasattr(volumes, '__len__') else kwargs.pop('n_items')
    lo, hi = shard_bounds(n, rank, world_size)
    get_v = volumes if callable(volumes) else volumes.__getitem__
    get_d = displacements if callable(displacements) else displacements.__getitem__
    outs = [compute(get_v(i), get_d(i), **kwargs) for i in range(lo, hi)]
    idx = list(range(lo, hi))
    if gather_to is None or world_size == 1:
        return (idx, outs) if gather_to is None else outs
    import numpy
    import torch
    dist = _dist()
    # same-shaped outputs: one stacked tensor per rank, point to point (device tensors stay on
    # the device under RCCL); ragged outputs: object gather
    sig = [(tuple(o.shape), str(o.dtype)) for o in outs]
    sigs = [None] * world_size
    dist.all_gather_object(sigs, sig, group=group)
    flat = [s for part in sigs for s in part]
    if flat and all(s == flat[0] for s in flat):
        if not sigs[gather_to]:      # (known to every rank: all of them raise, none enters the exchange)
            raise ValueError('the gathering rank must own at least one volume')
        as_numpy = bool(outs) and isinstance(outs[0], numpy.ndarray)
        if outs:
            mine = torch.stack([torch.as_tensor(o) for o in outs])
        else:
            shape, dtype = flat[0]
            mine = torch.empty((0,) + shape, dtype=getattr(torch, dtype.replace('torch.', '')))
        full = gather_batch(mine, n, gather_to, rank, world_size, group)
        if rank != gather_to:
            return None
        return [full[i].numpy() if as_numpy else full[i] for i in range(n)]
    host = [o.cpu() if hasattr(o, 'cpu') else o for o in outs]
    bucket = [None] * world_size if rank == gather_to else None
    dist.gather_object((idx, host), bucket, dst=_global_rank(gather_to, group), group=group)
    if rank != gather_to:
        return None
    full = [None] * n
    for ids, vals in bucket:
        for i, val in zip(ids, vals):
            full[i] = val
    return full
