"""N>1 path on CPU: world_size-2 `gloo` run of the sharding plumbing bench.py uses on GPUs.

The element batch shards as contiguous element ranges with NO data-path collective; only MAX(time)
and SUM(checksum) cross ranks.  Here the per-shard compute is done by the oracle (this is a test),
which lets us assert the property that matters: the union of shard outputs is bit-identical to the
single-rank output, and the reduced checksum equals the single-rank checksum.
"""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, nq, tmpdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle
    pkg = ge.load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nm = nq - 1
    lo, hi = pkg.shard.element_range(total, world, rank)
    b = oracle.fill_basis(nm, nq)
    # each rank generates exactly its slice of the one global array (counter-based generator)
    x = oracle.fill_random((hi - lo) * nm ** 3, 0x5F3759DF, lo * nm ** 3)
    out = oracle.bwdtrans_hex((nq,) * 3, hi - lo, b, b, b, x)
    np.save(os.path.join(tmpdir, f"out{rank}.npy"), out)
    elapsed = 0.25 + 0.5 * rank          # pretend timings: MAX must pick the slowest rank
    tmax, ssum = pkg.shard.reduce_time_and_checksum(dist, torch, elapsed, oracle.sumsq(out), "cpu")
    if rank == 0:
        np.save(os.path.join(tmpdir, "reduced.npy"), np.array([tmax, ssum]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [1001, 64])
def test_two_rank_shards_match_single_rank(tmp_path, oracle, total):
    import torch.multiprocessing as mp
    nq, world = 4, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, nq, str(tmp_path)), nprocs=world, join=True)
    nm = nq - 1
    b = oracle.fill_basis(nm, nq)
    x = oracle.fill_random(total * nm ** 3, 0x5F3759DF, 0)
    full = oracle.bwdtrans_hex((nq,) * 3, total, b, b, b, x)
    union = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(world)])
    assert np.array_equal(union, full)          # bit-identical: same per-element arithmetic order
    tmax, ssum = np.load(tmp_path / "reduced.npy")
    assert tmax == 0.75
    assert abs(ssum - oracle.sumsq(full)) <= 1e-12 * ssum


def test_element_ranges_partition():
    import __graft_entry__ as ge
    shard = ge.load_package().shard
    for total in (0, 1, 7, 8, 1000, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            rs = shard.all_ranges(total, world)
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    assert shard.element_range(10_000_000, 8, 7) == (8_750_000, 10_000_000)
    with pytest.raises(ValueError):
        shard.element_range(10, 2, 2)
    assert abs(shard.aggregate_gdofs(8 << 20, 343, 10, 0.0125) - 8 * 2 ** 20 * 343 * 10 / 0.0125e9) < 1e-9
