"""
World-size-2 test of the N > 1 path on CPU (gloo): volumes are sharded by rank with no data-path
collective; the optional hand-back gathers outputs on one rank.  The per-volume compute function
is injected -- here the CPU oracle, so that the test needs no GPU -- exactly where the product
calls elasticdeform_amd.deform_grid on a GPU box.
"""
import os
import socket

import numpy as np
import pytest

from elasticdeform_amd.distributed import deform_batch, shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                seen.extend(range(lo, hi))
            assert seen == list(range(n))                    # contiguous, ordered, complete
            sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0]
                     for r in range(world)]
            assert max(sizes) - min(sizes) <= 1              # balanced
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, out_dir):
    import torch.distributed as dist
    from oracle import ed_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def vol(i):
            return np.random.default_rng(100 + i).random((10, 12))

        def disp(i):
            return np.random.default_rng(200 + i).standard_normal((2, 3, 3)) * 2

        calls = []

        def compute(x, d, **kw):
            calls.append(1)
            return orc.deform_grid(x, d, **kw)

        class Lazy(object):            # a batch that only materialises what a rank asks for
            def __len__(self):
                return n

            def __getitem__(self, i):
                return vol(i)

        idx, outs = deform_batch(Lazy(), disp, compute=compute, order=3, mode="mirror")
        lo, hi = shard_bounds(n, rank, world)
        assert idx == list(range(lo, hi)) and len(calls) == hi - lo
        for i, o in zip(idx, outs):
            np.testing.assert_array_equal(o, orc.deform_grid(vol(i), disp(i), order=3, mode="mirror"))
        full = deform_batch(Lazy(), disp, compute=compute, gather_to=0, order=3, mode="mirror")
        if rank == 0:
            assert len(full) == n
            for i, o in enumerate(full):
                np.testing.assert_array_equal(
                    o, orc.deform_grid(vol(i), disp(i), order=3, mode="mirror"))
        else:
            assert full is None
        dist.barrier()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 5, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
