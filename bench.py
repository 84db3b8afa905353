#!/usr/bin/env python3
"""Headline benchmark: device-resident decode of dictionary-encoded pages (RLE/bit-packed hybrid ids).

Workload (BASELINE.json configs[1], per GPU): 100 M rows each of a dictionary-encoded INT32 column, a
dictionary-encoded INT64 column and a dictionary-encoded BINARY column (entries of 4..12 bytes), 65 536-entry
dictionaries (id width 16), 5 000 pages of 20 000 values each per column (parquet-java's default page row limit),
ids written as parquet-java writes them: 1-byte bit width, then bit-packed runs of at most 504 values with 1-byte
headers.  One step = one decode pass over the three columns (INT32 / INT64: run-plan kernel, sliced gather kernel,
and the generic kernel for pages the plan leaves to it - none in this workload; BINARY: one kernel, ids ->
offsets + bytes).  Pages, dictionaries and outputs are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line (rank 0).  With N > 1 it is launched by torch.distributed.run, one rank per
GPU; every rank decodes its own shard (row groups are independent: no collective in the data path).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import parquet_java_amd as pq  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
PAGE_VALUES = 20000
DICT_ENTRIES = 65536
ID_WIDTH = 16


def build_column(n_pages, elem, seed, device):
    """Synthesises `n_pages` dictionary-id pages on the device, laid out as parquet-java's
    DictionaryValuesWriter.getBytes + RunLengthBitPackingHybridEncoder produce them for ids that
    never repeat 8 times in a row: [bitWidth][header][<=63 groups * 16 bytes]... with the last group
    of a page zero-padded.  Returns (blob, page_len, ids, dictionary, out)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ids = torch.randint(0, DICT_ENTRIES, (n_pages, PAGE_VALUES), dtype=torch.int32, device=device, generator=g)
    groups = (PAGE_VALUES + 7) // 8
    padded = torch.zeros((n_pages, groups * 8), dtype=torch.int16, device=device)
    padded[:, :PAGE_VALUES] = ids.to(torch.int16)
    payload = padded.view(torch.uint8)  # width 16, LSB first == little-endian uint16
    runs = []
    gleft, goff = groups, 0
    while gleft > 0:
        gr = min(63, gleft)
        runs.append((goff, gr))
        goff += gr
        gleft -= gr
    page_len = 1 + len(runs) + groups * 2 * 8
    blob = torch.empty((n_pages, page_len), dtype=torch.uint8, device=device)
    blob[:, 0] = ID_WIDTH
    pos = 1
    for goff, gr in runs:
        blob[:, pos] = (gr << 1) | 1
        blob[:, pos + 1:pos + 1 + gr * 16] = payload[:, goff * 16:(goff + gr) * 16]
        pos += 1 + gr * 16
    assert pos == page_len
    dt = torch.int32 if elem == 4 else torch.int64
    info = torch.iinfo(dt)
    dictionary = torch.randint(info.min, info.max, (DICT_ENTRIES,), dtype=dt, device=device, generator=g)
    out = torch.zeros((n_pages, PAGE_VALUES), dtype=dt, device=device)
    return blob, page_len, ids, dictionary, out


def build_sorted_runs_column(n_pages, seed, device):
    """`n_pages` dictionary-id pages of sorted, run-heavy ids (a column that is sorted or clustered on its value), as
    RunLengthBitPackingHybridEncoder lays such ids out: 250 times an RLE run of 64..80 equal ids ([varint count << 1]
    [id, two bytes]) followed by one bit-packed group of eight ([0x03][8 x 16 bits]) that holds the end of that run and
    the beginning of the next.  Returns (blob, page_len, ids)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    cells = 250
    delta = torch.randint(0, 9, (n_pages, cells // 2), device=device, generator=g)
    a = torch.empty((n_pages, cells), dtype=torch.int64, device=device)
    a[:, 0::2] = 72 - delta
    a[:, 1::2] = 72 + delta
    k = torch.randint(1, 8, (n_pages, cells), device=device, generator=g)
    run_ids = torch.sort(torch.randint(0, DICT_ENTRIES, (n_pages, cells + 1), device=device, generator=g), dim=1).values
    this_id, next_id = run_ids[:, :-1], run_ids[:, 1:]
    cell = torch.empty((n_pages, cells, 21), dtype=torch.uint8, device=device)
    cell[:, :, 0] = (((a << 1) & 0x7F) | 0x80).to(torch.uint8)
    cell[:, :, 1] = (a >> 6).to(torch.uint8)
    cell[:, :, 2] = (this_id & 0xFF).to(torch.uint8)
    cell[:, :, 3] = (this_id >> 8).to(torch.uint8)
    cell[:, :, 4] = 0x03
    j = torch.arange(8, device=device).view(1, 1, 8)
    group = torch.where(j < k.unsqueeze(-1), this_id.unsqueeze(-1), next_id.unsqueeze(-1))
    cell[:, :, 5::2] = (group & 0xFF).to(torch.uint8)
    cell[:, :, 6::2] = (group >> 8).to(torch.uint8)
    page_len = 1 + cells * 21
    blob = torch.empty((n_pages, page_len), dtype=torch.uint8, device=device)
    blob[:, 0] = ID_WIDTH
    blob[:, 1:] = cell.view(n_pages, cells * 21)
    vals = torch.stack([this_id, next_id], dim=-1).reshape(-1)
    counts = torch.stack([a + k, 8 - k], dim=-1).reshape(-1)
    ids = torch.repeat_interleave(vals, counts).view(n_pages, PAGE_VALUES).to(torch.int32)
    return blob, page_len, ids


def make_descs(blob, page_len, dictionary, out, elem, device, encoding=None, flags=0):
    n_pages = blob.shape[0]
    encoding = pq.ENC_RLE_DICTIONARY if encoding is None else encoding
    pages = []
    for i in range(n_pages):
        pages.append(dict(data=blob.data_ptr() + i * page_len, out=out.data_ptr() + i * PAGE_VALUES * elem,
                          dict=dictionary.data_ptr(), data_len=page_len, num_values=PAGE_VALUES,
                          dict_len=DICT_ENTRIES, encoding=encoding, elem_size=elem, flags=flags))
    arr = pq.make_descs(pages)
    d_descs = torch.from_numpy(pq.descs_to_numpy(arr)).to(device)
    d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=device)
    return d_descs, d_res


def build_binary_column(cols_like, n_pages, device, seed=77):
    """The BINARY column: a dictionary page of DICT_ENTRIES entries of 4..12 random bytes ([length LE32][bytes] ...,
    about 768 KiB), its handle, and per page offsets[PAGE_VALUES + 1] + bytes outputs.  The id pages are those of
    `cols_like` (an id column built by build_column)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(4, 13, DICT_ENTRIES)
    body = rng.integers(0, 256, int(lens.sum()), dtype=np.uint8)
    page = np.zeros(int(lens.sum()) + 4 * DICT_ENTRIES, np.uint8)
    starts = np.cumsum(lens + 4) - lens
    for k in range(4):
        page[starts - 4 + k] = (lens >> (8 * k)) & 0xFF
    dst = np.repeat(starts - (np.cumsum(lens) - lens), lens) + np.arange(body.size)
    page[dst] = body
    handle = C.c_void_p()
    pq.check(pq.lib.pqd_dict_create(pq.TYPE_BYTE_ARRAY, 0, page.ctypes.data_as(C.c_void_p), page.size, DICT_ENTRIES,
                                    C.byref(handle)))
    info = pq.DictInfo()
    pq.check(pq.lib.pqd_dict_get(handle, C.byref(info)))
    c = cols_like
    cap = PAGE_VALUES * 12
    b_offsets = torch.zeros((n_pages, PAGE_VALUES + 1), dtype=torch.int32, device=device)
    b_bytes = torch.zeros((n_pages, cap), dtype=torch.uint8, device=device)
    pages = [dict(data=c["blob"].data_ptr() + i * c["page_len"], dict=info.entries, data_len=c["page_len"], num_values=PAGE_VALUES,
                  dict_len=DICT_ENTRIES, encoding=pq.ENC_RLE_DICTIONARY, elem_size=8) for i in range(n_pages)]
    bd = (pq.BinaryPageDesc * n_pages)()
    for i in range(n_pages):
        bd[i].dict_bytes = info.bytes
        bd[i].offsets = b_offsets.data_ptr() + i * (PAGE_VALUES + 1) * 4
        bd[i].bytes = b_bytes.data_ptr() + i * cap
        bd[i].num_values = PAGE_VALUES
        bd[i].dict_bytes_len = info.bytes_len
        bd[i].bytes_capacity = cap
    return dict(handle=handle, info=info, lens=lens, starts=starts, page=page, cap=cap, offsets=b_offsets, bytes=b_bytes,
                pages=pages, bd=bd, page_descs=torch.from_numpy(pq.descs_to_numpy(pq.make_descs(pages))).to(device),
                bin_descs=torch.from_numpy(np.frombuffer(bytes(bd), np.uint8).copy()).to(device),
                page_res=torch.zeros(n_pages * 16, dtype=torch.uint8, device=device),
                bin_res=torch.zeros(n_pages * 16, dtype=torch.uint8, device=device), ids=c["ids"], page_len=c["page_len"])


def verify_binary(b, device, chk):
    """Results of every page, and offsets + bytes of the first `chk` pages rebuilt with torch."""
    st_p, wr_p, _ = pq.results_from_numpy(b["page_res"].cpu().numpy())
    st_b, wr_b, nb_b = pq.results_from_numpy(b["bin_res"].cpu().numpy())
    ok = bool((st_b == 0).all() and (wr_b == PAGE_VALUES).all() and (st_p == 0).all() and (wr_p == PAGE_VALUES).all())
    t_lens = torch.from_numpy(b["lens"]).to(device)
    t_starts = torch.from_numpy(b["starts"]).to(device)
    t_page = torch.from_numpy(b["page"]).to(device)
    ids_chk = b["ids"][:chk].long()
    l_chk = t_lens[ids_chk]
    exp_off = torch.zeros((chk, PAGE_VALUES + 1), dtype=torch.int64, device=device)
    exp_off[:, 1:] = torch.cumsum(l_chk, 1)
    ok = ok and bool(torch.equal(exp_off.to(torch.int32), b["offsets"][:chk]))
    for i in range(chk):
        src = torch.repeat_interleave(t_starts[ids_chk[i]] - exp_off[i, :-1], l_chk[i]) + torch.arange(int(exp_off[i, -1]), device=device)
        ok = ok and bool(torch.equal(t_page[src], b["bytes"][i, :int(exp_off[i, -1])]))
    return ok, int(nb_b.astype(np.int64).sum())


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        pq.check(pq.lib.pqd_event_create(C.byref(self.h)))

    def record(self, stream):
        pq.check(pq.lib.pqd_event_record(self.h, stream))

    def sync(self):
        pq.check(pq.lib.pqd_event_sync(self.h))

    def ms_since(self, start):
        ms = C.c_float()
        pq.check(pq.lib.pqd_event_elapsed_ms(start.h, self.h, C.byref(ms)))
        return ms.value


def cpu_baseline(seconds, seed, threads=1):
    """Times the CPU restatement of the reference decoders (oracle/) on a bounded sample of the same
    workload: INT64 dictionary pages of 20 000 values, decoded back to back for about `seconds`."""
    import threading
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    dic = rng.integers(-1 << 62, 1 << 62, DICT_ENTRIES).astype(np.int64)
    pages = []
    for _ in range(8):  # 8 distinct pages, cycled
        ids = rng.integers(0, DICT_ENTRIES, PAGE_VALUES).astype(np.int32)
        pages.append((orc.dict_encode_ids(ids, DICT_ENTRIES), ids))
    st, vals = orc.dict_decode(pages[0][0], dic, PAGE_VALUES)
    assert st == 0 and (vals == dic[pages[0][1]]).all()
    counts = [0] * threads
    t0 = time.perf_counter()

    def work(t):
        i = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                orc.dict_decode(pages[i % 8][0], dic, PAGE_VALUES)  # ctypes releases the GIL
                i += 1
        counts[t] = i

    if threads == 1:
        work(0)
    else:
        ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    dt = time.perf_counter() - t0
    n = sum(counts)
    return {"value": round(n * PAGE_VALUES * 8 / dt / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "values_per_s": round(n * PAGE_VALUES / dt, 1),
            "sample": f"{n} INT64 dictionary pages x {PAGE_VALUES} values (64Ki-entry dictionary, id width 16), "
                      f"CPU restatement of DictionaryValuesReader + RunLengthBitPackingHybridDecoder, "
                      f"{threads} thread(s), {dt:.1f} s"}


def splitmix64(index, seed):
    """splitmix64 of (seed + index) as int64 bit patterns (torch's int64 arithmetic wraps)."""
    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def wrap(c):
        return c - (1 << 64) if c >= (1 << 63) else c
    z = (index + seed + 1) * wrap(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * wrap(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * wrap(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def config1_sweep(device, stream, timed, n_pages=256, n_values=1_000_000):
    """BASELINE.json configs[0] on the device: `n_pages` distinct pages of `n_values` INT32 values per bit
    width 1..32, RLE / bit-packing hybrid, decoded by one pqd_decode_pages_dev call per width.  Two flavours:
    A ("all_bit_packed"): random payload bytes behind 1-byte headers of 63 groups = 504 values
       (RunLengthBitPackingHybridEncoder.java:185-235 for data without repeats); two pages per width are re-derived
       from the payload bits with torch and compared.
    B ("java_encoder_layout"): values = splitmix64 & (2^w - 1), laid out as RunLengthBitPackingHybridEncoder
       (:146-273) writes those values (synth_pages.hybrid_encode: eight or more equal values from a group boundary
       become an RLE run, which at widths 1..3 is common); every decoded value is compared with its source."""
    import synth_pages as sp
    groups = n_values // 8
    full, tail = divmod(groups, 63)
    out = torch.empty((n_pages, n_values), dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    res, res_b = {}, {}

    def line(ms, encoded, ok, **more):
        alg = encoded + n_pages * n_values * 4
        return dict({"ms": round(ms, 4), "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS, 3),
                     "decoded_GBps": round(n_pages * n_values * 4 / ms / 1e6, 1), "verified": ok}, **more)

    def run(descs_fields):
        d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(descs_fields))).to(device)
        d_res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=device)
        out.zero_()
        ms = timed(lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), n_pages, pq.ENC_RLE, 4, stream), iters=5)
        status, written, _ = pq.results_from_numpy(d_res.cpu().numpy())
        return ms, bool((status == 0).all() and (written == n_values).all())

    for w in range(1, 33):
        # ---- flavour A
        g.manual_seed(42 + w)
        payload = torch.randint(0, 256, (n_pages, groups * w), dtype=torch.uint8, device=device, generator=g)
        parts = []
        if full:
            body = payload[:, :full * 63 * w].reshape(n_pages, full, 63 * w)
            head = torch.full((n_pages, full, 1), (63 << 1) | 1, dtype=torch.uint8, device=device)
            parts.append(torch.cat([head, body], 2).reshape(n_pages, -1))
        if tail:
            parts.append(torch.full((n_pages, 1), (tail << 1) | 1, dtype=torch.uint8, device=device))
            parts.append(payload[:, full * 63 * w:])
        blob = torch.cat(parts, 1).contiguous()
        page_len = blob.shape[1]
        ms, ok = run([dict(data=blob.data_ptr() + i * page_len, out=out.data_ptr() + i * n_values * 4, data_len=page_len,
                           num_values=n_values, encoding=pq.ENC_RLE, elem_size=4, bit_width=w) for i in range(n_pages)])
        for i in (0, n_pages - 1):
            bits = ((payload[i].to(torch.int64)[:, None] >> torch.arange(8, device=device)) & 1).reshape(-1, w)
            vals = (bits << torch.arange(w, device=device)).sum(1).to(torch.int32)  # width 32 wraps to the bit pattern
            ok = ok and bool(torch.equal(vals, out[i]))
            del bits, vals
        res[str(w)] = line(ms, n_pages * page_len, ok)
        del payload, blob, parts
        # ---- flavour B
        values = splitmix64(torch.arange(n_pages * n_values, dtype=torch.int64, device=device), 1000 * w) & ((1 << w) - 1)
        blob, offs = sp.hybrid_encode_chunked(values.view(n_pages, n_values), w, pages_per_call=32)
        blob = torch.cat([blob, torch.zeros(64, dtype=torch.uint8, device=device)])
        o = offs.tolist()
        ms, ok = run([dict(data=blob.data_ptr() + o[i], out=out.data_ptr() + i * n_values * 4, data_len=o[i + 1] - o[i],
                           num_values=n_values, encoding=pq.ENC_RLE, elem_size=4, bit_width=w) for i in range(n_pages)])
        ok = ok and bool(torch.equal(out.view(-1), values.to(torch.int32)))
        res_b[str(w)] = line(ms, o[-1], ok, encoded_bytes_per_page=round(o[-1] / n_pages, 1))
        del values, blob, offs
    del out
    torch.cuda.empty_cache()
    return {"pages": n_pages, "values_per_page": n_values, "by_bit_width": res, "flavour_of_by_bit_width": "A: all bit-packed, random payload",
            "java_encoder_layout_by_bit_width": res_b,
            "note": "algorithmic bytes = encoded page + 4 bytes per value; distinct pages (2.0 GiB decoded per call: beyond the 256 MiB Infinity Cache)"}


def page_decompression(device, stream, timed, n_pages=256, size=1 << 20):
    """pqd_decompress_pages_dev on n_pages pages of 1 MiB (parquet-java's default page size; 16 distinct ones), compressed
    on the host by the C libraries through pyarrow: GB/s of output, every distinct page compared with its original.
    Replaces the decompressor call in front of initFromPage (ColumnChunkPageReadStore.java:151-181, :232-262)."""
    try:
        import pyarrow as pa
    except ImportError:
        return {"skipped": "pyarrow is not importable here: no compressor to make the pages with"}
    distinct = 16
    words = [bytes(np.random.default_rng(1).integers(97, 123, k, dtype=np.uint8)) for k in np.random.default_rng(2).integers(2, 10, 2000)]
    kinds = {
        "text": lambda r: b" ".join(words[i] for i in r.integers(0, 2000, size // 5))[:size],
        "int32_below_1000": lambda r: r.integers(0, 1000, size // 4).astype("<i4").tobytes(),
    }
    out = {"pages": n_pages, "page_bytes": size}
    for kind, make in kinds.items():
        raws = [make(np.random.default_rng(100 + i)) for i in range(distinct)]
        for codec, num in (("snappy", 1), ("lz4_raw", 7), ("zstd", 6), ("gzip", 2)):
            comp = [np.frombuffer(pa.Codec(codec).compress(r, asbytes=True), np.uint8) for r in raws]
            pages = [comp[i % distinct] for i in range(n_pages)]
            offs = np.cumsum([0] + [p.size for p in pages])
            blob = torch.from_numpy(np.concatenate(pages + [np.zeros(64, np.uint8)])).to(device)
            dst = torch.zeros((n_pages, size), dtype=torch.uint8, device=device)
            d = (pq.DecompressDesc * n_pages)()
            for i in range(n_pages):
                d[i].src, d[i].dst, d[i].src_len, d[i].dst_len, d[i].codec = blob.data_ptr() + int(offs[i]), dst[i].data_ptr(), pages[i].size, size, num
            d_d = torch.from_numpy(np.frombuffer(bytes(d), np.uint8).copy()).to(device)
            res = torch.zeros(n_pages * 16, dtype=torch.uint8, device=device)
            ms = timed(lambda: pq.check(pq.lib.pqd_decompress_pages_dev(d_d.data_ptr(), res.data_ptr(), n_pages, num, stream)), iters=3)
            r = res.cpu().numpy().view(np.uint32).reshape(-1, 4)
            ok = bool((r[:, 0] == 0).all() and (r[:, 1] == size).all())
            for i in range(distinct):
                ok = ok and bytes(dst[n_pages - distinct + i].cpu().numpy()) == raws[(n_pages - distinct + i) % distinct]
            out[f"{codec}_{kind}"] = {"ms": round(ms, 3), "GBps_out": round(n_pages * size / ms / 1e6, 1),
                                      "ratio": round(n_pages * size / float(offs[-1]), 2), "verified": ok}
            del blob, dst
    return out


def config5_shard(device, stream, timed, rank=0):
    """BASELINE.json configs[4], one GPU's share: 512 row groups x 20 mixed-encoding column chunks dealt round
    robin over 8 GPUs = 64 row groups x 20 chunks x 250 000 values here (dictionary INT32 / INT64, DELTA_BINARY_PACKED,
    BYTE_STREAM_SPLIT DOUBLE, definition levels on half of the columns), pages built on the device.  Timed twice:
    one pqd_decode_pages_dev call per encoding class over the whole share (device descriptors, events), and row
    group by row group through shard.decode_row_groups (host descriptors, one pqd_decode_pages call and one host
    wait per row group: wall clock)."""
    import synth_pages as sp
    from parquet_java_amd.shard import assign_row_groups, decode_row_groups
    owned = assign_row_groups([1] * 512, 8)[rank % 8]
    shard = sp.build_mixed_shard(pq, device, owned)
    calls, enc_bytes, out_bytes, values = [], 0, 0, 0
    for name, c in shard.items():
        d_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(sp.shard_page_descs(c, 0, c["n_pages"])))).to(device)
        d_res = torch.zeros(c["n_pages"] * 16, dtype=torch.uint8, device=device)
        calls.append((name, d_descs, d_res, c))
        enc_bytes += int(c["offs"][-1])
        out_bytes += c["n_pages"] * c["page_values"] * c["elem"]
        values += c["n_pages"] * c["page_values"]

    def run():
        for name, d_descs, d_res, c in calls:
            pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), c["n_pages"], c["encoding"], c["elem"], stream)
    ms = timed(run, iters=5)
    ok = True
    per_class = {}
    for name, d_descs, d_res, c in calls:
        status, written, _ = pq.results_from_numpy(d_res.cpu().numpy())
        ok = ok and bool((status == 0).all() and (written == c["page_values"]).all()) and c["expect"]()
        one = lambda: pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), c["n_pages"], c["encoding"], c["elem"], stream)  # noqa: E731
        per_class[name] = round(timed(one, iters=3), 4)
    for c in shard.values():
        c["out"].zero_()
    groups = {g: pq.make_descs(sp.shard_row_group_descs(shard, k)) for k, g in enumerate(owned)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = decode_row_groups(groups, rank % 8, 8, stream, [1] * 512)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ok = ok and all(not s["failed"] for _, s in done.values()) and all(c["expect"]() for c in shard.values())
    alg = enc_bytes + out_bytes
    return {"row_groups": len(owned), "chunks_per_row_group": 20, "values_per_chunk": 250_000, "pages": sum(c["n_pages"] for c in shard.values()),
            "values_incl_levels": values, "decoded_bytes": out_bytes, "encoded_bytes": enc_bytes,
            "ms_one_call_per_encoding_class": round(ms, 4), "decoded_GBps": round(out_bytes / ms / 1e6, 1),
            "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS, 3),
            "ms_per_class_alone": per_class,
            "ms_row_group_by_row_group_host_calls": round(wall_ms, 2), "verified": bool(ok)}


def sources_digest():
    """sha256 over the kernel sources (the same function as in tools/pmc_traffic.py)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "parquet-java_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: starts N fresh processes, one rank per GPU, with the
    environment torch.distributed.run would give them, and returns the worst exit code.  The parent makes
    no GPU call (counting devices does not initialise the runtime) and never execs; rank 0 prints the line.
    PQD_BENCH_ONE_GPU=1 puts every rank on device 0 (a rehearsal of the N-rank path on a one-GPU box)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    one_gpu = bool(os.environ.get("PQD_BENCH_ONE_GPU"))
    if have < n and not one_gpu:
        print(f"bench.py: --gpus {n} but only {have} device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if one_gpu else r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(p.wait() for p in procs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pages", type=int, default=5000, help="pages per column per GPU (5000 x 20000 = 100 M rows)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="length of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--warm-ms", type=float, default=60.0,
                    help="the warm-up lasts at least this long: after the --warmup steps, further untimed steps run until then")
    ap.add_argument("--no-config1", action="store_true", help="skip the 32-width sweep of 1 M-value hybrid pages in extras")
    ap.add_argument("--no-extras", action="store_true",
                    help="the timed workload only (profiler passes: every launch of a kernel in the trace then belongs to it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not os.environ.get("PQD_BENCH_FORCE_DIST"):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # (PQD_BENCH_FORCE_DIST=1 takes the process-group path with a single rank too: a rehearsal on a one-GPU box)
    distributed = world > 1 or bool(os.environ.get("PQD_BENCH_FORCE_DIST"))
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if os.environ.get("PQD_BENCH_ONE_GPU"):  # several ranks on one device: RCCL refuses that, gloo carries the barrier
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    pq.init(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    cols = []
    for elem in (4, 8):
        blob, page_len, ids, dictionary, out = build_column(args.pages, elem, 1234 + 17 * rank + elem, device)
        d_descs, d_res = make_descs(blob, page_len, dictionary, out, elem, device)
        cols.append(dict(elem=elem, blob=blob, page_len=page_len, ids=ids, dictionary=dictionary, out=out,
                         descs=d_descs, res=d_res))
    # the BINARY column: id pages of its own, a dictionary of 4..12-byte entries, offsets + bytes per page
    blob, page_len, ids, _, _ = build_column(args.pages, 4, 1234 + 17 * rank + 12, device)
    binary = build_binary_column(dict(blob=blob, page_len=page_len, ids=ids), args.pages, device)
    binary["blob"] = blob
    torch.cuda.synchronize()

    def decode_fixed(c):
        pq.decode_pages_dev(c["descs"].data_ptr(), c["res"].data_ptr(), args.pages, pq.ENC_RLE_DICTIONARY, c["elem"], stream)

    def decode_binary():
        pq.check(pq.lib.pqd_decode_binary_pages_dev(binary["handle"], binary["page_descs"].data_ptr(), binary["bin_descs"].data_ptr(),
                                                    binary["page_res"].data_ptr(), binary["bin_res"].data_ptr(), args.pages, stream))

    def step():
        decode_fixed(cols[0])
        decode_fixed(cols[1])
        decode_binary()

    # the check against torch comes first, the warm-up steps directly in front of the timed region: the
    # check leaves the GPU idle while the host compares, and the first steps after an idle spell run 5-8 % slower
    verified = None
    binary_bytes = None
    if not args.no_verify:
        step()
        torch.cuda.synchronize()
        verified = True
        for c in cols:
            status, written, _ = pq.results_from_numpy(c["res"].cpu().numpy())
            ok = (status == 0).all() and (written == PAGE_VALUES).all()
            expect = c["dictionary"][c["ids"].long()]
            ok = ok and bool(torch.equal(expect, c["out"]))
            verified = verified and bool(ok)
            del expect
        ok, binary_bytes = verify_binary(binary, device, min(args.pages, 64))
        verified = verified and ok
        for c in cols:
            c["out"].zero_()
        torch.cuda.synchronize()
    if binary_bytes is None:
        binary_bytes = int(torch.from_numpy(binary["lens"]).to(device)[binary["ids"].long()].sum().item())

    # W warm-up steps as asked for, then further untimed steps until the warm-up has lasted WARM_MS of device work: the
    # clocks settle over some tens of milliseconds of load, whatever the number of steps that takes
    t_warm = time.perf_counter()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    extra_warm = 0
    while (time.perf_counter() - t_warm) * 1e3 < args.warm_ms:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        extra_warm += 4

    # ---- timed region: exactly `steps` steps between barriers.  The library brackets the INT64 gather launch and the
    # BINARY launch of every step with two HIP events each, which the launches themselves record on the stream they
    # run on (an event recorded between launches costs the stream about 6 us, so the other launches are timed after
    # the region, not inside it).
    phase = (C.c_float * 3)()
    ncalls = C.c_int()
    for series in (0, 1):
        pq.check(pq.lib.pqd_set_phase_series(series))
        pq.check(pq.lib.pqd_phase_ms_mean(phase, C.byref(ncalls)))  # start a new series
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        decode_fixed(cols[0])
        pq.check(pq.lib.pqd_set_phase_series(0))
        pq.check(pq.lib.pqd_set_phase_timing(2))
        decode_fixed(cols[1])
        pq.check(pq.lib.pqd_set_phase_series(1))
        pq.check(pq.lib.pqd_set_phase_timing(2))
        decode_binary()
        pq.check(pq.lib.pqd_set_phase_timing(0))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if os.environ.get("PQD_BENCH_ONE_GPU") else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    pq.check(pq.lib.pqd_set_phase_series(0))
    pq.check(pq.lib.pqd_set_phase_timing(0))
    pq.check(pq.lib.pqd_phase_ms_mean(phase, C.byref(ncalls)))
    gather_ms, gather_calls = phase[1], ncalls.value
    pq.check(pq.lib.pqd_set_phase_series(1))
    pq.check(pq.lib.pqd_set_phase_timing(0))
    pq.check(pq.lib.pqd_phase_ms_mean(phase, C.byref(ncalls)))
    binary_ms, binary_calls = phase[1], ncalls.value
    pq.check(pq.lib.pqd_set_phase_series(0))
    # for the record, after the region: every column's call between two events, and the three launches of an INT64 call
    per_col = [0.0, 0.0, 0.0]
    ea, eb = Event(), Event()
    for k in range(5):
        for i, fn in enumerate((lambda: decode_fixed(cols[0]), lambda: decode_fixed(cols[1]), decode_binary)):
            ea.record(stream)
            fn()
            eb.record(stream)
            eb.sync()
            per_col[i] += eb.ms_since(ea) / 5
    phase2 = (C.c_float * 3)()
    ncalls2 = C.c_int()
    for k in range(5):
        decode_fixed(cols[0])
        pq.check(pq.lib.pqd_set_phase_timing(1))
        decode_fixed(cols[1])
        pq.check(pq.lib.pqd_set_phase_timing(0))
        decode_binary()
    pq.check(pq.lib.pqd_phase_ms_mean(phase2, C.byref(ncalls2)))
    plan_ms, rest_ms = phase2[0], phase2[2]

    # ---- outside the timed region: the other shapes of this workload, for the record
    extras = {}
    if rank == 0 and not args.no_extras:
        spoiler = torch.zeros(64 << 20, dtype=torch.int64, device=device)
        def timed(fn, iters=10):  # noqa: E306
            """Mean time of fn() by events around every call, with a read of 512 MiB of unrelated data between the
            calls: a loop over one call alone can find its input still in the 256 MiB Infinity Cache.  (Read-only:
            dirty lines left in the caches would be written back on the timed call's account.)"""
            fn()
            torch.cuda.synchronize()
            total = 0.0
            a, b = Event(), Event()
            for _ in range(iters):
                spoiler.sum()
                a.record(stream)
                fn()
                b.record(stream)
                b.sync()
                total += b.ms_since(a)
            return total / iters
        # the BINARY column as the reference hands it out: zero-copy views into the dictionary page
        # (PlainValuesDictionary.java:96-104), i.e. the id decode alone (readValueDictionaryId)
        c = cols[0]
        id_descs, id_res = make_descs(binary["blob"], binary["page_len"], c["dictionary"], c["out"], 4, device,
                                      encoding=pq.ENC_RLE, flags=pq.F_WIDTH_PREFIX)
        ms = timed(lambda: pq.decode_pages_dev(id_descs.data_ptr(), id_res.data_ptr(), args.pages, pq.ENC_RLE, 4, stream))
        ok = bool(torch.equal(c["out"].view(torch.int32), binary["ids"]))
        extras["binary_column_as_dictionary_ids"] = {"ms": round(ms, 4), "values_per_s": round(args.pages * PAGE_VALUES / ms * 1e3, 1),
                                                     "verified": ok}
        # ... and materialised in two steps, as before the one-pass kernel: ids -> 8-byte views (the INT64 gather with the
        # dictionary's views as entries) -> k_binary_pages
        c = cols[1]
        views_out = c["out"]  # 8 bytes per value
        for i, pg in enumerate(binary["pages"]):
            pg["out"] = views_out.data_ptr() + i * PAGE_VALUES * 8
            binary["bd"][i].views = views_out.data_ptr() + i * PAGE_VALUES * 8
        v_descs = torch.from_numpy(pq.descs_to_numpy(pq.make_descs(binary["pages"]))).to(device)
        b_descs = torch.from_numpy(np.frombuffer(bytes(binary["bd"]), np.uint8).copy()).to(device)
        v_res = torch.zeros(args.pages * 16, dtype=torch.uint8, device=device)
        one_pass = (binary["offsets"].clone(), binary["bytes"].clone())
        binary["offsets"].zero_()
        binary["bytes"].zero_()
        ms_views = timed(lambda: pq.decode_pages_dev(v_descs.data_ptr(), v_res.data_ptr(), args.pages, pq.ENC_RLE_DICTIONARY, 8, stream))
        ms_bytes = timed(lambda: pq.check(pq.lib.pqd_binary_materialize_dev(b_descs.data_ptr(), binary["bin_res"].data_ptr(), args.pages, stream)))
        ok = bool(torch.equal(one_pass[0], binary["offsets"]) and torch.equal(one_pass[1], binary["bytes"]))
        del one_pass
        ms_one = timed(decode_binary)
        alg_b = args.pages * binary["page_len"] + args.pages * (PAGE_VALUES + 1) * 4 + binary_bytes
        extras["binary_column_offsets_and_bytes"] = {
            "ms_one_pass_cold_input": round(ms_one, 4), "frac_of_hbm_peak": round(alg_b / ms_one / 1e6 / HBM_PEAK_GBS, 3),
            "two_steps_ms_ids_to_views": round(ms_views, 4), "two_steps_ms_views_to_bytes": round(ms_bytes, 4),
            "two_steps_frac_of_hbm_peak": round(alg_b / (ms_views + ms_bytes) / 1e6 / HBM_PEAK_GBS, 3),
            "mean_value_bytes": round(binary_bytes / (args.pages * PAGE_VALUES), 2), "two_steps_equal_one_pass": ok}

        # the INT64 column again with sorted, run-heavy ids (a column sorted or clustered on its value): such pages
        # are decoded whole by the left-over launch, not through the chunk plan
        r_blob, r_len, r_ids = build_sorted_runs_column(args.pages, 4242, device)
        c = cols[1]
        r_descs, r_res = make_descs(r_blob, r_len, c["dictionary"], c["out"], 8, device)
        c["out"].zero_()
        ms = timed(lambda: pq.decode_pages_dev(r_descs.data_ptr(), r_res.data_ptr(), args.pages, pq.ENC_RLE_DICTIONARY, 8, stream))
        st_r, wr_r, _ = pq.results_from_numpy(r_res.cpu().numpy())
        ok = bool((st_r == 0).all() and (wr_r == PAGE_VALUES).all() and torch.equal(c["out"], c["dictionary"][r_ids.long()]))
        alg = args.pages * r_len + args.pages * PAGE_VALUES * 8
        extras["dictionary_int64_sorted_runs_of_ids"] = {
            "ms": round(ms, 4), "values_per_s": round(args.pages * PAGE_VALUES / ms * 1e3, 1),
            "algorithmic_GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS, 3),
            "page_bytes": r_len, "runs_per_page": 500, "verified": ok}
        del r_blob, r_ids
        binary["offsets"] = binary["bytes"] = None
        torch.cuda.empty_cache()
        extras["config5_shard"] = config5_shard(device, stream, timed, rank)
        extras["page_decompression_256_pages_of_1MiB"] = page_decompression(device, stream, timed)
        if not args.no_config1:
            for c in cols:  # make room: the columns of the timed region are no longer needed
                c["out"] = c["blob"] = c["ids"] = None
            torch.cuda.empty_cache()
            extras["config1_hybrid_int32_pages_1M_values"] = config1_sweep(device, stream, timed)

    n_vals = args.pages * PAGE_VALUES
    encoded_bytes_col = args.pages * cols[0]["page_len"]
    binary_decoded = n_vals * 4 + args.pages * 4 + binary_bytes  # offsets[n + 1] + bytes per page
    decoded_bytes_step = n_vals * (4 + 8) + binary_decoded
    # the kernels of a step, with their algorithmic bytes per launch: the encoded pages once, the dictionary once,
    # the decoded bytes once
    k64_alg_bytes = encoded_bytes_col + n_vals * 8 + DICT_ENTRIES * 8
    kbin_alg_bytes = encoded_bytes_col + binary_decoded + binary["page"].size
    ms_per_step = elapsed / args.steps * 1e3
    value = decoded_bytes_step * world / (elapsed / args.steps) / 1e9

    if rank == 0:
        # HBM-side bytes per launch from the PMC passes (tools/pmc_traffic.sh): numbers measured in
        # another run, so they are reported only while the kernel sources are the ones they were measured on
        traffic = traffic_bin = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("sources_sha256") == sources_digest():
                    traffic = rec.get("k_dict_pages_sliced_8_bytes_per_launch")
                    traffic_bin = rec.get("k_binary_fused_bytes_per_launch")
            except Exception:
                traffic = traffic_bin = None
        ach64 = k64_alg_bytes / (gather_ms * 1e-3) / 1e9
        achb = kbin_alg_bytes / (binary_ms * 1e-3) / 1e9
        step_alg = 3 * encoded_bytes_col + decoded_bytes_step
        line = {
            "metric": "device-resident decoded GB/s (RLE/bitpack+dict INT32 pages) at 1/2/4/8 GPUs",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: dictionary-encoded INT32 + INT64 + BINARY columns, 65536-entry dictionaries "
                                   "(id width 16; BINARY entries of 4..12 bytes), RLE/bit-packed hybrid ids (504-value runs, "
                                   f"uniform random ids), {args.pages} pages x {PAGE_VALUES} values per column per GPU "
                                   f"({n_vals / 1e6:.0f} M rows x 3 columns)",
                       "pages_per_column": args.pages, "page_values": PAGE_VALUES, "dictionary_entries": DICT_ENTRIES,
                       "sharding": "one shard per GPU, no collective"},
            "values_per_s": round(3 * n_vals * world / (elapsed / args.steps), 1),
            "algorithmic_GBps": round(step_alg * world / (elapsed / args.steps) / 1e9, 2),
            "whole_step_frac_of_hbm_peak": round(step_alg / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "warmup_extra_steps_until_%d_ms" % int(args.warm_ms): extra_warm,
            "ms_per_column_after_region": {"int32": round(per_col[0], 4), "int64": round(per_col[1], 4), "binary": round(per_col[2], 4),
                                           "int32_plus_int64": round(per_col[0] + per_col[1], 4)},
            "frac_of_hbm_peak_per_column": {
                "int32": round((encoded_bytes_col + n_vals * 4) / per_col[0] / 1e6 / HBM_PEAK_GBS, 4),
                "int64": round((encoded_bytes_col + n_vals * 8) / per_col[1] / 1e6 / HBM_PEAK_GBS, 4),
                "binary": round((encoded_bytes_col + binary_decoded) / per_col[2] / 1e6 / HBM_PEAK_GBS, 4)},
            "verified_vs_gather": verified,
            "extras": extras,
            "roofline": {"bound": "hbm", "kernel": "k_binary_fused (BINARY dictionary pages -> offsets + bytes): the longest kernel of the step",
                         "achieved": round(achb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achb / HBM_PEAK_GBS, 4), "traffic": traffic_bin,
                         "algorithmic_bytes_per_launch": kbin_alg_bytes, "avg_launch_ms": round(binary_ms, 4),
                         "launches_timed": binary_calls},
            "roofline_int64_gather": {"bound": "hbm", "kernel": "k_dict_pages_sliced<8> (INT64 dictionary pages)",
                                      "achieved": round(ach64, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(ach64 / HBM_PEAK_GBS, 4), "traffic": traffic,
                                      "algorithmic_bytes_per_launch": k64_alg_bytes, "avg_launch_ms": round(gather_ms, 4),
                                      "launches_timed": gather_calls,
                                      "other_launches_ms": {"k_dict_slots + k_rle_plan_chunks": round(plan_ms, 4),
                                                            "k_rle_pages (left-over pages)": round(rest_ms, 4),
                                                            "note": "timed after the region, five more steps"}},
        }
        # the reference's own CPU decoders need a JVM (scalar BytePacker; the Vector API plugin needs
        # avx512vbmi + avx512_vbmi2 as well): probed here, run only where they exist
        import shutil
        flags = ""
        try:
            with open("/proc/cpuinfo") as f:
                flags = next((ln for ln in f if ln.startswith("flags")), "")
        except OSError:
            pass
        line["host"] = {"cpus": os.cpu_count(), "java": shutil.which("java"), "javac": shutil.which("javac"),
                        "avx512vbmi": " avx512vbmi " in flags + " ", "avx512_vbmi2": " avx512_vbmi2 " in flags + " ",
                        "reference_cpu_decoders": "not run: no JVM on this host" if not shutil.which("java") else
                        "JVM present; the Java harness is not part of this repository"}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds, 99, 1)
            line["cpu_baseline_16_threads"] = cpu_baseline(min(args.cpu_seconds, 5.0), 99, 16)
        print(json.dumps(line), flush=True)
    pq.lib.pqd_dict_destroy(binary["handle"])
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
