"""The torch page builders of synth_pages.py (what bench.py and the large GPU tests synthesise their pages
with, on the device) against the oracle's restatements of parquet-java's writers, byte for byte, on the CPU."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_pages as sp  # noqa: E402


def value_sets(rng, w, n):
    hi = 1 << w
    yield rng.integers(0, hi, n)
    yield np.repeat(rng.integers(0, hi, n // 5 + 1), rng.integers(1, 12, n // 5 + 1))[:n]        # runs around the limit of 8
    yield np.repeat(rng.integers(0, hi, n // 40 + 1), rng.geometric(1 / 40.0, n // 40 + 1))[:n]   # long runs
    yield np.zeros(n, np.int64)
    v = rng.integers(0, hi, n)
    v[n // 3:n // 3 + 9] = v[n // 3]
    v[-8:] = v[-1]
    yield v
    yield np.where(rng.random(n) < 0.9, hi - 1, rng.integers(0, hi, n))                          # level-like


@pytest.mark.parametrize("w", [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 20, 24, 25, 31, 32])
def test_hybrid_encode_is_the_java_layout(orc, w):
    rng = np.random.default_rng(100 + w)
    for n in (1, 7, 8, 9, 503, 504, 505, 1000, 4099):
        sets = [v for v in value_sets(rng, w, n) if v.size == n]
        vals = np.stack(sets).astype(np.int64)
        blob, offs = sp.hybrid_encode(torch.from_numpy(vals), w)
        blob, offs = blob.numpy(), offs.numpy()
        for p in range(vals.shape[0]):
            want = np.asarray(orc.rle_encode(w, (vals[p] & 0xFFFFFFFF).astype(np.uint32).view(np.int32)))
            got = blob[offs[p]:offs[p + 1]]
            assert got.size == want.size and (got == want).all(), (w, n, p)


def test_dictionary_pages_and_prefix(orc):
    rng = np.random.default_rng(7)
    ids = np.stack([rng.integers(0, 4096, 3000), np.sort(rng.integers(0, 4096, 3000))]).astype(np.int64)
    blob, offs, w = sp.dictionary_id_pages(torch.from_numpy(ids), 4096)
    assert w == 12
    for p in range(2):
        want = np.asarray(orc.dict_encode_ids(ids[p].astype(np.int32), 4096))
        assert (blob.numpy()[offs[p]:offs[p + 1]] == want).all()
    lv = (rng.random((3, 5000)) < 0.8).astype(np.int64) * 3
    blob, offs = sp.prefixed(*sp.hybrid_encode(torch.from_numpy(lv), 2))
    for p in range(3):
        body = bytes(orc.rle_encode(2, lv[p].astype(np.int32)))
        assert blob.numpy()[offs[p]:offs[p + 1]].tobytes() == len(body).to_bytes(4, "little") + body


@pytest.mark.parametrize("k_bits,n", [(12, 25000), (4, 20000), (20, 129), (40, 130), (12, 2), (7, 3), (12, 161), (1, 1000)])
def test_delta_pages_are_what_the_writer_writes(orc, k_bits, n):
    g = torch.Generator()
    g.manual_seed(5)
    pages, page_len, values = sp.delta_pages(3, n, k_bits, g, torch.device("cpu"))
    for p in range(3):
        want = np.asarray(orc.delta_encode(values[p].numpy()))
        got = pages[p].numpy()
        # (bytes behind the last delta of a partly filled miniblock are padding: the writer leaves what its buffer
        # held, this builder zeros; both decode alike, which is what is compared where they differ)
        st, vals, total, used = orc.delta_decode(got, n)
        assert st == 0 and total == n and used == page_len and (vals[:n] == values[p].numpy()).all()
        assert got.size == want.size
        # everything but the padding of the last miniblock and the width bytes of the miniblocks the last block
        # does not have (the writer leaves there what its arrays held) is the writer's bytes
        nd = n - 1
        n_last = nd - (nd - 1) // 128 * 128
        minis = (n_last + 31) // 32
        differ = np.nonzero(got != want)[0]
        pad_bits = (-n_last) % 32 * k_bits
        pad_from = got.size - (pad_bits + 7) // 8
        n_blocks = (nd + 127) // 128
        varint_len = lambda x: max(1, (int(x).bit_length() + 6) // 7)  # noqa: E731
        head = 3 + varint_len(n) + varint_len(int(values[p, 0]) << 1)
        widths_at = head + (n_blocks - 1) * (2 + 4 + 16 * k_bits) + 2
        allowed = set(range(widths_at + minis, widths_at + 4)) | set(range(pad_from - 1, got.size))
        assert set(differ.tolist()) <= allowed, (differ[:8], widths_at, pad_from)


def test_bss_pages(orc):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 500))
    raw = torch.from_numpy(x.view(np.uint8).reshape(2, 500, 8).copy())
    pages = sp.bss_pages(raw)
    for p in range(2):
        assert (pages[p].numpy() == np.asarray(orc.bss_encode(x[p].view(np.uint8), 8))).all()
