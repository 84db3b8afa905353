"""The multi-GPU path: row groups partitioned over ranks, no collective in the data path.
Covered on CPU with a world_size-2 gloo group (each rank 'decodes' its shard with numpy)."""
import os
import socket

import numpy as np
import pytest


def test_assignment_policies():
    from parquet_java_amd.shard import assign_row_groups
    rr = assign_row_groups([1] * 10, 4)
    assert rr == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    sizes = [100, 1, 1, 1, 50, 50, 2, 98]
    g = assign_row_groups(sizes, 2, "greedy")
    assert sorted(g[0] + g[1]) == list(range(8))
    loads = [sum(sizes[i] for i in part) for part in g]
    assert abs(loads[0] - loads[1]) <= 2
    assert assign_row_groups([], 3) == [[], [], []]
    with pytest.raises(ValueError):
        assign_row_groups([1], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_groups, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parquet_java_amd.shard import assign_row_groups, gather_summaries
    mine = assign_row_groups([1] * n_groups, world)[rank]
    # stand-in for the decode of each owned row group: a checksum of its (deterministic) values
    sums = {g: int(np.arange(g * 1000, g * 1000 + 1000, dtype=np.int64).sum()) for g in mine}
    dist.barrier()
    allsums = gather_summaries(dict(rank=rank, groups=mine, sums=sums))
    if rank == 0:
        q.put(allsums)
    dist.destroy_process_group()


def test_two_rank_sharding_over_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_groups = 9
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_groups, q)) for r in range(2)]
    [p.start() for p in procs]
    res = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [r["rank"] for r in res] == [0, 1]
    covered = sorted(g for r in res for g in r["groups"])
    assert covered == list(range(n_groups))  # every row group decoded exactly once
    merged = {}
    for r in res:
        merged.update(r["sums"])
    assert merged == {g: int(np.arange(g * 1000, g * 1000 + 1000, dtype=np.int64).sum()) for g in range(n_groups)}
