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
