"""Row-group sharding across the GPUs of one node.

Row groups (and the column chunks inside them) are independent units of a parquet file
(ParquetFileReader.internalReadRowGroup reads each block on its own,
parquet-hadoop/.../ParquetFileReader.java:1164-1200), so the decode path shards with no exchange
step: every rank decodes its own row groups on its own GPU.  The only cross-rank traffic is a
host-side gather of per-rank summaries (counts, failures), done with torch.distributed objects.
"""
from typing import List, Sequence


def assign_row_groups(sizes: Sequence[int], world_size: int, policy: str = "round_robin") -> List[List[int]]:
    """Returns, per rank, the row-group indices it decodes.

    round_robin: row group r -> rank r % world_size (what bench.py uses for equal-sized groups).
    greedy:      largest-first onto the least-loaded rank (by `sizes`, e.g. total_compressed_size).
    """
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    out: List[List[int]] = [[] for _ in range(world_size)]
    if policy == "round_robin":
        for r in range(len(sizes)):
            out[r % world_size].append(r)
    elif policy == "greedy":
        load = [0] * world_size
        for r in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
            k = min(range(world_size), key=lambda j: (load[j], j))
            out[k].append(r)
            load[k] += sizes[r]
        for lst in out:
            lst.sort()
    else:
        raise ValueError("unknown policy " + policy)
    return out


def gather_summaries(local_summary, group=None):
    """All ranks contribute one picklable summary; every rank receives the list (rank order).
    Host-side only: nothing on the decode path waits for it."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return [local_summary]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local_summary, group=group)
    return out


def decode_row_groups(row_groups, rank: int, world_size: int, stream=None, sizes=None, policy: str = "round_robin"):
    """Decodes the row groups `rank` owns, one ``pqd_decode_pages`` call per row group.

    row_groups[g] is a ctypes array of ``PageDesc`` (host descriptors; data, dictionaries and outputs in
    this rank's device memory).  Returns ``{g: (PageResult array, summary dict)}`` for the owned groups;
    the summaries (pages, values, failed pages) are what ``gather_summaries`` collects."""
    import ctypes as C

    from . import PageResult, check, lib
    mine = assign_row_groups(sizes if sizes is not None else [len(g) for g in row_groups], world_size, policy)[rank]
    out = {}
    for g in mine:
        descs = row_groups[g]
        n = len(descs)
        results = (PageResult * n)()
        check(lib.pqd_decode_pages(descs, results, n, C.c_void_p(stream) if stream else None), "pqd_decode_pages")
        out[g] = (results, dict(group=g, pages=n, values=sum(r.values_written for r in results),
                                failed=[i for i, r in enumerate(results) if r.status != 0]))
    return out
