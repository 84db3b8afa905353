"""The sharded workload at its real one-GPU size: 512 row groups x 20 mixed-encoding column chunks over 8 GPUs,
rank 0's share = 64 row groups x 20 chunks x 250 000 values (dictionary INT32 with 12-bit ids, dictionary INT64
with 16-bit ids, DELTA_BINARY_PACKED with 12-bit deltas, BYTE_STREAM_SPLIT DOUBLE, definition levels on half of
the columns), built on the device (synth_pages.py, checked byte for byte against the CPU encoders in
tests/test_synth_pages.py).  Decoded the way parquet-java reads a file, one row group after the other
(ParquetFileReader.readNextRowGroup / internalReadRowGroup, parquet-hadoop/.../ParquetFileReader.java:1164-1200),
through shard.assign_row_groups + shard.decode_row_groups, and once more with one call per encoding class.
Every page's status and count is checked, every value by re-derivation on the device, and one row group page
by page against the oracle."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_rank0_share_of_the_512_row_group_file(orc):
    import torch
    import parquet_java_amd as pq
    import synth_pages as sp
    from parquet_java_amd.shard import assign_row_groups, decode_row_groups
    import gpu_harness as gh
    pq.init(0)
    world, n_groups = 8, 512
    owned = assign_row_groups([1] * n_groups, world)[0]
    assert owned == list(range(0, n_groups, world)) and len(owned) == 64
    shard = sp.build_mixed_shard(pq, gh.dev(), owned)
    n = 25_000
    assert sum(c["n_pages"] for c in shard.values()) == 64 * 300
    assert sum(c["n_pages"] * n for c in shard.values() if c["encoding"] != pq.ENC_RLE) == 64 * 20 * 250_000
    # ---- row group by row group, as the file reader walks them
    groups = {g: pq.make_descs(sp.shard_row_group_descs(shard, k)) for k, g in enumerate(owned)}
    sizes = [int(sum(int(c["offs"][-1]) for c in shard.values()) // len(owned))] * n_groups
    stream = torch.cuda.current_stream().cuda_stream
    done = decode_row_groups(groups, 0, world, stream, sizes)
    assert sorted(done) == owned
    for g, (results, summary) in done.items():
        assert summary["pages"] == 300 and not summary["failed"] and summary["values"] == 300 * n
        assert all(r.status == 0 and r.values_written == n for r in results)
    torch.cuda.synchronize()
    for name, c in shard.items():
        assert c["expect"](), name
    # ---- one row group against the oracle, page by page
    k = 37
    host_blobs = {name: c["blob"].cpu().numpy() for name, c in shard.items()}
    for name, c in shard.items():
        per_group = c["chunks"] * c["pages_per_chunk"]
        offs = c["offs"][k * per_group:(k + 1) * per_group + 1].tolist()
        out = c["out"][k * per_group:(k + 1) * per_group].cpu().numpy()
        dic = None if c["dictionary"] is None else c["dictionary"].cpu().numpy()
        for i in range(0, per_group, 7):
            page = host_blobs[name][offs[i]:offs[i + 1]]
            if name in ("dict32", "dict64"):
                st, vals = orc.dict_decode(page, dic, n)
            elif name == "delta":
                st, vals, total, used = orc.delta_decode(page, n)
                assert total == n and used == page.size
            elif name == "bss":
                st, raw = orc.bss_decode(page, 8, n)
                vals = raw.view(np.int64)
            else:
                st, vals, _ = orc.rle_decode(2, page, n, prefixed=True)
            assert st == 0 and vals.size == n and (vals == out[i]).all(), (name, i)
    # ---- the same pages with one device call per encoding class
    for c in shard.values():
        c["out"].zero_()
    for name, c in shard.items():
        d_descs = gh.to_dev(pq.descs_to_numpy(pq.make_descs(sp.shard_page_descs(c, 0, c["n_pages"]))))
        d_res = torch.zeros(c["n_pages"] * 16, dtype=torch.uint8, device=gh.dev())
        pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), c["n_pages"], c["encoding"], c["elem"], stream)
        torch.cuda.synchronize()
        status, written, _ = pq.results_from_numpy(d_res.cpu().numpy())
        assert (status == 0).all() and (written == n).all(), name
        assert c["expect"](), name
