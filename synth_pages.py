"""Synthetic parquet pages built with torch on whatever device the values live on (the GPU in bench.py and
the GPU tests, the CPU in the tests that compare these builders with the CPU encoders byte for byte).

The layouts are the ones parquet-java's writers produce for the same values:

* hybrid_encode: RunLengthBitPackingHybridEncoder (parquet-column/.../values/rle/
  RunLengthBitPackingHybridEncoder.java:146-273): values are buffered in groups of eight; eight or more equal
  values that START at a group boundary become an RLE run (varint(count << 1), value in ceil(w / 8) bytes),
  everything else is bit-packed, at most 63 groups (504 values) behind one header byte, the last group of a
  page zero-padded.  The sequential rule "from a boundary p: a run of L >= 8 equal values -> RLE run, next
  boundary p + L; else one group, next boundary p + 8" is evaluated for all pages at once: only the long
  runs can move the boundaries' phase (p mod 8), each long run is a map of the 8 phases to 8 phases, and a
  prefix composition of those maps gives the phase at which every long run is entered.
* dictionary_id_pages: DictionaryValuesWriter.getBytes (.../dictionary/DictionaryValuesWriter.java:153-180):
  one byte of bit width, then the hybrid stream.
* delta_pages: DeltaBinaryPackingValuesWriterForLong (.../delta/DeltaBinaryPackingValuesWriterForLong.java:
  75-184), blocks of 128 in 4 miniblocks, for deltas this module shapes so that every miniblock has one width.
* bss_pages: ByteStreamSplitValuesWriter (.../bytestreamsplit/ByteStreamSplitValuesWriter.java): K byte planes.

No file of the decode library or of the oracle is used here.
"""
import torch


def pack_groups(g, w):
    """g: int64 [G, 8], every value below 2^w -> uint8 [G, w]: the eight values LSB-first, back to back
    (ByteBitPackingLE.pack8Values)."""
    out = torch.zeros((g.shape[0], w), dtype=torch.uint8, device=g.device)
    for j in range(w):  # output byte j holds stream bits [8j, 8j + 8)
        t0, t1 = (8 * j) // w, min(7, (8 * j + 7) // w)
        acc = None
        for t in range(t0, t1 + 1):
            sh = t * w - 8 * j  # where value t starts, relative to this byte
            part = (g[:, t] << sh) if sh >= 0 else (g[:, t] >> -sh)
            acc = part if acc is None else acc | part
        out[:, j] = (acc & 0xFF).to(torch.uint8)
    return out


def _varint_len(x):
    n = torch.ones_like(x)
    for k in range(1, 5):
        n = n + (x >= (1 << (7 * k))).to(x.dtype)
    return n


def hybrid_encode(values, w):
    """values: integer tensor [P, N], each 0 <= v < 2^w (w 1..32) -> (blob uint8 [total], offsets int64 [P + 1]):
    page p's stream is blob[offsets[p]:offsets[p + 1]]."""
    assert 1 <= w <= 32
    P, N = values.shape
    dev = values.device
    v = values.reshape(-1).to(torch.int64)
    T = P * N
    idx = torch.arange(T, device=dev)
    change = torch.ones(T, dtype=torch.bool, device=dev)
    change[1:] = v[1:] != v[:-1]
    change[::N] = True
    starts = torch.nonzero(change).squeeze(1)
    ends = torch.cat([starts[1:], torch.tensor([T], device=dev)])
    long_run = (ends - starts) >= 8
    ls, le = starts[long_run], ends[long_run]
    K = ls.numel()
    page_pos = torch.arange(P, device=dev) * N
    # every long run as a map of the phase at which the boundaries arrive to the phase at which they leave
    phi = torch.arange(8, device=dev)[None, :]
    first = ls[:, None] + ((phi - ls[:, None]) % 8)  # first boundary inside the run, per phase
    takes = (le[:, None] - first) >= 8
    f_run = torch.where(takes, (le[:, None] % 8).expand(K, 8), phi.expand(K, 8))
    f_page = (page_pos % 8)[:, None].expand(P, 8)
    keys = torch.cat([page_pos * 2, ls * 2 + 1])
    order = torch.argsort(keys)
    F = torch.cat([f_page, f_run])[order].contiguous()
    M = F.shape[0]
    G = F.clone()
    d = 1
    while d < M:  # inclusive prefix composition: G[i] = F[i] o ... o F[0]
        G = torch.cat([G[:d], torch.gather(G[d:], 1, G[:-d])])
        d *= 2
    enter = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), G[:-1, 0]])
    enter_of = torch.empty(M, dtype=torch.int64, device=dev)
    enter_of[order] = enter
    ph = enter_of[P:]  # phase at which long run k is entered
    r_start = ls + ((ph - ls) % 8)
    is_rle = (le - r_start) >= 8
    r_start, r_end, r_val = r_start[is_rle], le[is_rle], v[ls[is_rle]]
    R = r_start.numel()
    # values inside RLE runs
    cover = torch.zeros(T + 1, dtype=torch.int32, device=dev)
    cover[r_start] += 1
    cover[r_end] -= 1
    in_rle = torch.cumsum(cover[:T], 0) > 0
    packed = ~in_rle
    # stretches of bit-packed values: from a page start or the end of an RLE run to the next RLE run / page end
    mark = torch.zeros(T, dtype=torch.bool, device=dev)
    mark[page_pos] = True
    mark[r_end[r_end < T]] = True
    s_pos = torch.nonzero(mark).squeeze(1)  # start of stretch s
    S = s_pos.numel()
    rank = torch.cumsum(packed.to(torch.int64), 0) - packed.to(torch.int64)  # packed values in front of position i
    rank_end = torch.cat([rank[s_pos[1:]], (rank[-1:] + packed[-1:].to(torch.int64))])
    n_s = rank_end - rank[s_pos]
    g_s = (n_s + 7) // 8
    h_s = (g_s + 62) // 63
    s_bytes = h_s + g_s * w
    count = r_end - r_start
    r_bytes = _varint_len(count << 1) + (w + 7) // 8
    # stream order: by position (an empty stretch may share its position with an RLE run: it has no bytes)
    ekeys = torch.cat([s_pos * 2 + 1, r_start * 2])
    eorder = torch.argsort(ekeys)
    esize = torch.cat([s_bytes, r_bytes])[eorder]
    eoff_sorted = torch.cumsum(esize, 0) - esize
    eoff = torch.empty_like(eoff_sorted)
    eoff[eorder] = eoff_sorted
    s_off, r_off = eoff[:S], eoff[S:]
    total = int(esize.sum().item())
    epage = torch.cat([s_pos, r_start])[eorder] // N
    offsets = torch.zeros(P + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(torch.zeros(P, dtype=torch.int64, device=dev).scatter_add_(0, epage, esize), 0)
    blob = torch.zeros(total, dtype=torch.uint8, device=dev)
    # ---- bit-packed groups
    sid = torch.cumsum(mark.to(torch.int64), 0) - 1  # stretch of position i
    g_base = torch.cumsum(g_s, 0) - g_s              # first group of stretch s among all groups
    n_groups = int(g_s.sum().item())
    if n_groups:
        pi = idx[packed]
        ps = sid[pi]
        q = rank[pi] - rank[s_pos[ps]]  # index of the value inside its stretch
        padded = torch.zeros(n_groups * 8, dtype=torch.int64, device=dev)
        padded[g_base[ps] * 8 + q] = v[pi]
        del pi, ps, q
        payload = pack_groups(padded.view(n_groups, 8), w)  # [n_groups, w]
        del padded
        g_sid = torch.repeat_interleave(torch.arange(S, device=dev), g_s)
        g_in = torch.arange(n_groups, device=dev) - g_base[g_sid]
        g_dst = s_off[g_sid] + (g_in // 63 + 1) + g_in * w
        blob[(g_dst[:, None] + torch.arange(w, device=dev)[None, :]).reshape(-1)] = payload.reshape(-1)
        del payload, g_dst
        # headers: one per 63 groups
        h_base = torch.cumsum(h_s, 0) - h_s
        n_heads = int(h_s.sum().item())
        h_sid = torch.repeat_interleave(torch.arange(S, device=dev), h_s)
        h_in = torch.arange(n_heads, device=dev) - h_base[h_sid]
        h_groups = torch.clamp(g_s[h_sid] - 63 * h_in, max=63)
        blob[s_off[h_sid] + h_in * (63 * w + 1)] = ((h_groups << 1) | 1).to(torch.uint8)
    # ---- RLE runs: varint(count << 1), then the value in ceil(w / 8) bytes
    if R:
        head = count << 1
        nb = _varint_len(head)
        for k in range(5):
            m = nb > k
            byte = ((head >> (7 * k)) & 0x7F) | torch.where(nb > k + 1, 0x80, 0)
            blob[(r_off + k)[m]] = byte[m].to(torch.uint8)
        for k in range((w + 7) // 8):
            blob[r_off + nb + k] = ((r_val >> (8 * k)) & 0xFF).to(torch.uint8)
    return blob, offsets


def dictionary_id_pages(ids, n_dict):
    """ids [P, N] -> (blob, offsets) of dictionary data pages: [bit width][hybrid stream]
    (bit width = 32 - nlz(n_dict - 1), DictionaryValuesWriter.java:154-156)."""
    w = max(1, int(n_dict - 1).bit_length())  # (a dictionary of one entry is written with width 0 by the reference; not used here)
    stream, offs = hybrid_encode(ids, w)
    P = ids.shape[0]
    shift = torch.arange(P + 1, device=ids.device)
    out = torch.zeros(stream.numel() + P, dtype=torch.uint8, device=ids.device)
    new_offs = offs + shift
    out[new_offs[:-1]] = w
    lens = offs[1:] - offs[:-1]
    page_of = torch.repeat_interleave(torch.arange(P, device=ids.device), lens)
    out[torch.arange(stream.numel(), device=ids.device) + page_of + 1] = stream
    return out, new_offs, w


def prefixed(blob, offsets):
    """The same streams behind a 4-byte little-endian length each (RunLengthBitPackingHybridValuesWriter.getBytes,
    .../rle/RunLengthBitPackingHybridValuesWriter.java:66-74: V1 levels, RLE booleans)."""
    P = offsets.numel() - 1
    dev = blob.device
    lens = offsets[1:] - offsets[:-1]
    new_offs = offsets + 4 * torch.arange(P + 1, device=dev)
    out = torch.zeros(blob.numel() + 4 * P, dtype=torch.uint8, device=dev)
    for k in range(4):
        out[new_offs[:-1] + k] = ((lens >> (8 * k)) & 0xFF).to(torch.uint8)
    page_of = torch.repeat_interleave(torch.arange(P, device=dev), lens)
    out[torch.arange(blob.numel(), device=dev) + 4 * (page_of + 1)] = blob
    return out, new_offs


def _zigzag_varint(x, n_bytes):
    """x: int64 [P], non-negative with exactly n_bytes varint bytes of zig-zag -> uint8 [P, n_bytes]."""
    z = x << 1
    cols = []
    for k in range(n_bytes):
        b = (z >> (7 * k)) & 0x7F
        cols.append((b | (0x80 if k + 1 < n_bytes else 0)).to(torch.uint8))
    return torch.stack(cols, 1)


def delta_pages(n_pages, n_values, k_bits, generator, device, first=1_600_000_000_000_000_000, base_delta=1000):
    """DELTA_BINARY_PACKED INT64 pages of timestamps ts[i] = ts[i-1] + base_delta + r, r < 2^k_bits random, with
    r = 2^k_bits - 1 in the first slot of every miniblock and r = 0 in the second slot of every block, so that the
    writer's choices are the same everywhere: min delta = base_delta, miniblock width = k_bits.  Blocks of 128 in 4 miniblocks of 32.
    Returns (pages uint8 [P, page_len], page_len, values int64 [P, n_values])."""
    assert 1 <= k_bits <= 40 and n_values >= 2
    P = n_pages
    nd = n_values - 1
    n_blocks = (nd + 127) // 128
    r = torch.randint(0, 1 << k_bits, (P, n_blocks * 128), dtype=torch.int64, device=device, generator=generator)
    r = r.view(P, n_blocks, 4, 32)
    r[:, :, :, 0] = (1 << k_bits) - 1
    r[:, :, 0, 1] = 0
    r = r.view(P, n_blocks * 128)
    n_last = nd - (n_blocks - 1) * 128  # deltas of the last block
    if n_last == 1:
        r[:, nd - 1] = 0  # a block of one delta: that delta is its minimum
    r[:, nd:] = 0  # what lies behind the last delta is the writer's padding
    firsts = first + torch.arange(P, device=device, dtype=torch.int64) * 1_000_003
    values = torch.empty((P, n_values), dtype=torch.int64, device=device)
    values[:, 0] = firsts
    values[:, 1:] = firsts[:, None] + torch.cumsum(r[:, :nd] + base_delta, 1)
    minis_last = (n_last + 31) // 32  # miniblocks of the last block that hold deltas
    last = r[:, (n_blocks - 1) * 128:].view(P, 4, 32)
    # widths of the last block's miniblocks from what they hold (a partly filled one may miss the planted maximum)
    lw = torch.zeros((P, 4), dtype=torch.int64, device=device)
    lmax = last.max(dim=2).values
    for b in range(1, 41):
        lw += (lmax >= (1 << (b - 1))).to(torch.int64)
    lw[:, minis_last:] = 0
    uniform_last = bool((lw[:, :minis_last] == lw[0, :minis_last]).all())
    assert uniform_last, "last block widths differ between pages"
    lw0 = [int(x) for x in lw[0].tolist()]
    zz_min = base_delta << 1
    min_bytes = []
    while True:
        min_bytes.append((zz_min & 0x7F) | (0x80 if zz_min >= 0x80 else 0))
        zz_min >>= 7
        if not zz_min:
            break
    n_first = 0
    zz = int(firsts.max().item()) << 1
    while zz:
        n_first += 1
        zz >>= 7
    assert (int(firsts.min().item()) << 1) >> (7 * (n_first - 1)), "first values need varints of one length"
    head = [0x80, 0x01, 0x04]
    t = n_values
    while True:
        head.append((t & 0x7F) | (0x80 if t >= 0x80 else 0))
        t >>= 7
        if not t:
            break
    block_bytes = len(min_bytes) + 4 + 4 * 4 * k_bits
    last_bytes = len(min_bytes) + 4 + sum(4 * x for x in lw0[:minis_last])
    page_len = len(head) + n_first + (n_blocks - 1) * block_bytes + last_bytes
    pages = torch.zeros((P, page_len), dtype=torch.uint8, device=device)
    pages[:, :len(head)] = torch.tensor(head, dtype=torch.uint8, device=device)
    pages[:, len(head):len(head) + n_first] = _zigzag_varint(firsts, n_first)
    pos = len(head) + n_first
    if n_blocks > 1:
        body = pages[:, pos:pos + (n_blocks - 1) * block_bytes].view(P, n_blocks - 1, block_bytes)
        body[:, :, :len(min_bytes)] = torch.tensor(min_bytes, dtype=torch.uint8, device=device)
        body[:, :, len(min_bytes):len(min_bytes) + 4] = k_bits
        full = r[:, :(n_blocks - 1) * 128].reshape(-1, 8)
        body[:, :, len(min_bytes) + 4:] = pack_groups_wide(full, k_bits).view(P, n_blocks - 1, 16 * k_bits)
        pos += (n_blocks - 1) * block_bytes
    pages[:, pos:pos + len(min_bytes)] = torch.tensor(min_bytes, dtype=torch.uint8, device=device)
    pos += len(min_bytes)
    pages[:, pos:pos + 4] = torch.tensor(lw0, dtype=torch.uint8, device=device)
    pos += 4
    for m in range(minis_last):
        if lw0[m]:
            pages[:, pos:pos + 4 * lw0[m]] = pack_groups_wide(last[:, m].reshape(-1, 8), lw0[m]).view(P, 4 * lw0[m])
        pos += 4 * lw0[m]
    assert pos == page_len
    return pages, page_len, values


def pack_groups_wide(g, w):
    """pack_groups for widths up to 64 (BytePackerForLong): int64 [G, 8] -> uint8 [G, w].  A byte takes bits of at
    most two values once w >= 8; shifts stay below 64."""
    if w <= 32:
        return pack_groups(g, w)
    out = torch.zeros((g.shape[0], w), dtype=torch.uint8, device=g.device)
    for j in range(w):
        t0, t1 = (8 * j) // w, min(7, (8 * j + 7) // w)
        acc = None
        for t in range(t0, t1 + 1):
            sh = t * w - 8 * j
            # bits of value t that fall into this byte: value bit b lands at stream bit t*w + b
            part = ((g[:, t] << sh) & 0xFF) if sh >= 0 else ((g[:, t] >> -sh) & 0xFF)
            acc = part if acc is None else acc | part
        out[:, j] = acc.to(torch.uint8)
    return out


def bss_pages(values_bytes):
    """values_bytes: uint8 [P, N, K] (the values' little-endian bytes) -> uint8 [P, K * N]: byte plane s holds byte
    s of every value."""
    P, N, K = values_bytes.shape
    return values_bytes.transpose(1, 2).contiguous().view(P, K * N)


def hybrid_encode_chunked(values, w, pages_per_call=1024):
    """hybrid_encode over slices of pages (bounds the temporaries): same result."""
    blobs, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=values.device)], 0
    for a in range(0, values.shape[0], pages_per_call):
        b, o = hybrid_encode(values[a:a + pages_per_call], w)
        blobs.append(b)
        offs.append(o[1:] + base)
        base += int(b.numel())
    return torch.cat(blobs), torch.cat(offs)


# ---- the sharded workload: 512 row groups x 20 mixed-encoding column chunks over 8 GPUs ------------------------
SHARD_COLUMNS = (["dict32"] * 5) + (["dict64"] * 5) + (["delta"] * 5) + (["bss"] * 5)
SHARD_LEVEL_COLUMNS = tuple(range(0, 20, 2))  # definition levels on half of the columns


def build_mixed_shard(enc, device, row_groups, chunk_values=250_000, page_values=25_000, seed=8):
    """The pages of `row_groups` (global row-group numbers: this rank's share) of the sharded workload, built on
    `device`: per row group 20 column chunks of `chunk_values` values in pages of `page_values` (5 x dictionary
    INT32 with 12-bit ids over 4096 entries, 5 x dictionary INT64 with 16-bit ids over 65 536 entries, 5 x
    DELTA_BINARY_PACKED INT64 with 12-bit deltas, 5 x BYTE_STREAM_SPLIT DOUBLE) and V1 definition levels
    (maximum 3, 10 % nulls at each level, 4-byte length prefix) for every second column.
    `enc`: the library's encoding numbers (a namespace with ENC_RLE, ENC_RLE_DICTIONARY, ENC_DELTA_BINARY_PACKED,
    ENC_BYTE_STREAM_SPLIT, F_LENGTH_PREFIX).  Returns {class name: dict}: pages of one encoding class side by
    side (page index = (local row group * chunks of the class + chunk) * pages per chunk + page), with
    blob / offs (page bytes), out (decoded values go here), dictionary, and expect() -> bool (device check)."""
    G = len(row_groups)
    ppc = chunk_values // page_values
    n = page_values
    gen = torch.Generator(device=device)
    gen.manual_seed(seed * 1_000_003 + int(row_groups[0]) if G else seed)
    shard = {}

    def dict_class(name, n_dict, elem):
        P = G * 5 * ppc
        ids = torch.randint(0, n_dict, (P, n), dtype=torch.int32, device=device, generator=gen)
        blob, offs, w = dictionary_id_pages(ids, n_dict)
        dt = torch.int32 if elem == 4 else torch.int64
        info = torch.iinfo(dt)
        dic = torch.randint(info.min, info.max, (n_dict,), dtype=dt, device=device, generator=gen)
        out = torch.zeros((P, n), dtype=dt, device=device)
        shard[name] = dict(encoding=enc.ENC_RLE_DICTIONARY, elem=elem, bit_width=0, flags=0, blob=blob, offs=offs, out=out,
                           dictionary=dic, dict_len=n_dict, n_pages=P, chunks=5,
                           expect=lambda: bool(torch.equal(out, dic[ids.long()])))

    dict_class("dict32", 4096, 4)
    dict_class("dict64", 65536, 8)

    P = G * 5 * ppc
    pages, page_len, values = delta_pages(P, n, 12, gen, device)
    d_out = torch.zeros((P, n), dtype=torch.int64, device=device)
    shard["delta"] = dict(encoding=enc.ENC_DELTA_BINARY_PACKED, elem=8, bit_width=0, flags=0, blob=pages.view(-1),
                          offs=torch.arange(P + 1, device=device, dtype=torch.int64) * page_len, out=d_out, dictionary=None,
                          dict_len=0, n_pages=P, chunks=5, expect=lambda: bool(torch.equal(d_out, values)))

    dbl = torch.randn((P, n), dtype=torch.float64, device=device, generator=gen)
    dbl.view(torch.int64)[:, 7] = 0x7FF8_0000_0000_1234  # a NaN with a payload, -0.0 and a denormal in every page
    dbl.view(torch.int64)[:, 8] = -0x8000_0000_0000_0000
    dbl.view(torch.int64)[:, 9] = 3
    b_pages = bss_pages(dbl.view(torch.uint8).view(P, n, 8))
    b_out = torch.zeros((P, n), dtype=torch.int64, device=device)
    shard["bss"] = dict(encoding=enc.ENC_BYTE_STREAM_SPLIT, elem=8, bit_width=0, flags=0, blob=b_pages.view(-1),
                        offs=torch.arange(P + 1, device=device, dtype=torch.int64) * (8 * n), out=b_out, dictionary=None,
                        dict_len=0, n_pages=P, chunks=5, expect=lambda: bool(torch.equal(b_out, dbl.view(torch.int64))))

    PL = G * len(SHARD_LEVEL_COLUMNS) * ppc
    u = torch.rand((3, PL, n), device=device, generator=gen)
    lv = ((u[0] >= 0.1).to(torch.int32) * (1 + (u[1] >= 0.1).to(torch.int32) * (1 + (u[2] >= 0.1).to(torch.int32))))
    del u
    l_blob, l_offs = prefixed(*hybrid_encode_chunked(lv, 2))
    l_out = torch.zeros((PL, n), dtype=torch.int32, device=device)
    shard["levels"] = dict(encoding=enc.ENC_RLE, elem=4, bit_width=2, flags=enc.F_LENGTH_PREFIX, blob=l_blob, offs=l_offs,
                           out=l_out, dictionary=None, dict_len=0, n_pages=PL, chunks=len(SHARD_LEVEL_COLUMNS),
                           expect=lambda: bool(torch.equal(l_out, lv)))
    for c in shard.values():
        c["page_values"] = n
        c["pages_per_chunk"] = ppc
        c["blob"] = torch.cat([c["blob"], torch.zeros(64, dtype=torch.uint8, device=device)])  # room for whole-dword reads
    return shard


def shard_page_descs(c, first_page, n_pages):
    """Descriptor fields (dicts for make_descs) of pages [first_page, first_page + n_pages) of one class."""
    offs = c["offs"][first_page:first_page + n_pages + 1].tolist()
    n, elem = c["page_values"], c["elem"]
    base, out = c["blob"].data_ptr(), c["out"].data_ptr()
    dic = 0 if c["dictionary"] is None else c["dictionary"].data_ptr()
    return [dict(data=base + offs[i], out=out + (first_page + i) * n * elem, dict=dic, data_len=offs[i + 1] - offs[i],
                 num_values=n, dict_len=c["dict_len"], encoding=c["encoding"], elem_size=elem, bit_width=c["bit_width"],
                 flags=c["flags"]) for i in range(n_pages)]


def shard_row_group_descs(shard, local_group):
    """All page descriptors of one row group of the shard (every class's chunks of that group), as dicts."""
    pages = []
    for c in shard.values():
        per_group = c["chunks"] * c["pages_per_chunk"]
        pages += shard_page_descs(c, local_group * per_group, per_group)
    return pages
