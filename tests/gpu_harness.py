"""Helpers for the GPU parity tests: put pages on the device, call the C ABI, fetch results."""
import ctypes as C

import numpy as np
import torch

import parquet_java_amd as pq


def dev():
    return torch.device("cuda:0")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


class Page:
    def __init__(self, data, num_values, encoding, elem_size=4, bit_width=0, flags=0, dictionary=None):
        self.data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if isinstance(data, (bytes, bytearray)) else data,
                                         dtype=np.uint8)
        self.num_values = int(num_values)
        self.encoding = encoding
        self.elem_size = elem_size
        self.bit_width = bit_width
        self.flags = flags
        self.dictionary = None if dictionary is None else np.ascontiguousarray(dictionary)


def decode_pages(pages, misalign=True, fill=0xA5):
    """Runs one pqd_decode_pages_dev launch over `pages` (all of one encoding class).

    Page bytes are packed back to back into one device blob (so most pages start at odd addresses when
    `misalign`), followed by a guard region of 0xEE bytes; outputs are pre-filled with `fill`.
    Returns a list of (status, values_written, bytes_consumed, out_bytes ndarray[uint8]).
    """
    enc, elem = pages[0].encoding, pages[0].elem_size
    offs, pos = [], 1 if misalign else 0
    for p in pages:
        if not misalign:
            pos = (pos + 15) & ~15
        offs.append(pos)
        pos += p.data.size
    blob = np.full(pos + 64, 0xEE, np.uint8)
    for p, o in zip(pages, offs):
        blob[o:o + p.data.size] = p.data
    d_blob = to_dev(blob)
    out_offs, opos = [], 0
    for p in pages:
        out_offs.append(opos)
        opos += (p.num_values * p.elem_size + 15) & ~15
        opos += 16
    d_out = torch.full((max(opos, 16),), fill, dtype=torch.uint8, device=dev())
    dicts = {}
    descs = []
    for p, o, oo in zip(pages, offs, out_offs):
        dd = 0
        dl = 0
        if p.dictionary is not None:
            key = id(p.dictionary)
            if key not in dicts:
                dicts[key] = to_dev(p.dictionary.view(np.uint8)) if p.dictionary.size else torch.zeros(16, dtype=torch.uint8, device=dev())
            dd = dicts[key].data_ptr()
            dl = p.dictionary.size
        descs.append(dict(data=d_blob.data_ptr() + o, out=d_out.data_ptr() + oo, dict=dd, data_len=p.data.size,
                          num_values=p.num_values, dict_len=dl, encoding=p.encoding, elem_size=p.elem_size,
                          bit_width=p.bit_width, flags=p.flags))
    arr = pq.make_descs(descs)
    d_descs = to_dev(pq.descs_to_numpy(arr))
    d_res = torch.zeros(len(pages) * 16, dtype=torch.uint8, device=dev())
    stream = torch.cuda.current_stream().cuda_stream
    pq.decode_pages_dev(d_descs.data_ptr(), d_res.data_ptr(), len(pages), enc, elem, stream)
    torch.cuda.synchronize()
    status, written, consumed = pq.results_from_numpy(d_res.cpu().numpy())
    out = d_out.cpu().numpy()
    res = []
    for i, (p, oo) in enumerate(zip(pages, out_offs)):
        nbytes = p.num_values * p.elem_size
        res.append((int(status[i]), int(written[i]), int(consumed[i]), out[oo:oo + nbytes].copy(),
                    out[oo + nbytes:oo + ((nbytes + 15) & ~15) + 16].copy()))
    return res


def unpack_dev(w, packed, n, long=False, offset=0):
    """pqd_unpack_i32_dev / _i64_dev with the input placed `offset` bytes past a 256-byte aligned address."""
    packed = np.ascontiguousarray(packed, np.uint8)
    buf = np.full(offset + packed.size + 32, 0xEE, np.uint8)
    buf[offset:offset + packed.size] = packed
    d_in = to_dev(buf)
    d_out = torch.full((max(n, 1) + 4,), -0x5A5A5A5B, dtype=torch.int64 if long else torch.int32, device=dev())
    f = pq.lib.pqd_unpack_i64_dev if long else pq.lib.pqd_unpack_i32_dev
    st = f(w, d_in.data_ptr() + offset, packed.size, d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    return st, out[:n], out[n:]


def materialize_binary(cases, fill=0xA5):
    """One pqd_binary_materialize_dev launch.  cases: (views uint64[n], dict_bytes uint8[], capacity).
    Outputs sit at odd addresses between guard bytes.  Returns per case
    (status, values_written, bytes_written, offsets int32[n+1], bytes uint8[capacity], [guards])."""
    dicts, blobs = {}, []
    o_pos, b_pos, v_pos = 0, 0, 0
    lay = []
    for views, dict_bytes, cap in cases:
        n = len(views)
        o_pos = (o_pos + 3) & ~3
        lay.append((v_pos, o_pos + 4, b_pos + 3))
        v_pos += n * 8
        o_pos += 4 + (n + 1) * 4 + 12
        b_pos += 3 + cap + 13
    all_views = np.concatenate([np.ascontiguousarray(c[0], np.uint64) for c in cases] + [np.zeros(1, np.uint64)])
    d_views = to_dev(all_views)
    d_offs = torch.full((o_pos + 16,), fill, dtype=torch.uint8, device=dev())
    d_bytes = torch.full((b_pos + 16,), fill, dtype=torch.uint8, device=dev())
    descs = (pq.BinaryPageDesc * len(cases))()
    for i, ((views, dict_bytes, cap), (vp, op, bp)) in enumerate(zip(cases, lay)):
        key = id(dict_bytes)
        if key not in dicts:
            dicts[key] = to_dev(dict_bytes) if len(dict_bytes) else torch.zeros(16, dtype=torch.uint8, device=dev())
        descs[i].views = d_views.data_ptr() + vp
        descs[i].dict_bytes = dicts[key].data_ptr()
        descs[i].offsets = d_offs.data_ptr() + op
        descs[i].bytes = d_bytes.data_ptr() + bp
        descs[i].num_values = len(views)
        descs[i].dict_bytes_len = len(dict_bytes)
        descs[i].bytes_capacity = cap
    d_descs = to_dev(np.frombuffer(bytes(descs), np.uint8).copy())
    d_res = torch.zeros(len(cases) * 16, dtype=torch.uint8, device=dev())
    st = pq.lib.pqd_binary_materialize_dev(d_descs.data_ptr(), d_res.data_ptr(), len(cases),
                                           torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    status, written, consumed = pq.results_from_numpy(d_res.cpu().numpy())
    offs, data = d_offs.cpu().numpy(), d_bytes.cpu().numpy()
    res = []
    for i, ((views, dict_bytes, cap), (vp, op, bp)) in enumerate(zip(cases, lay)):
        n = len(views)
        w = int(written[i])
        o = offs[op:op + (n + 1) * 4].view(np.int32)
        guards = [offs[op - 4:op], offs[op + (w + 1) * 4:op + (n + 1) * 4 + 12] if status[i] else offs[op + (n + 1) * 4:op + (n + 1) * 4 + 12],
                  data[bp - 3:bp], data[bp + cap:bp + cap + 13], data[bp + int(consumed[i]):bp + cap]]
        res.append((int(status[i]), w, int(consumed[i]), o.copy(), data[bp:bp + cap].copy(), guards))
    return res
