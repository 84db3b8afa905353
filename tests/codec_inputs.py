"""Seeded byte strings the codec vectors were made from (tests/golden/make_codec_golden.py compressed
them; the tests regenerate them and check their SHA-256 against the one stored beside each vector)."""
import numpy as np


def inputs():
    rng = np.random.default_rng(20240611)
    words = [bytes(rng.integers(97, 123, rng.integers(2, 10), dtype=np.uint8)) for _ in range(300)]
    text = b" ".join(words[i] for i in rng.integers(0, 300, 40000))
    yield "empty", b""
    yield "one_byte", b"x"
    yield "short", b"hello hello hello hello"
    yield "run_1", b"a" * 100000  # offset 1 copies, long matches
    yield "run_3", b"abc" * 30000
    yield "run_7", bytes(range(7)) * 9000
    yield "period_70", bytes(rng.integers(0, 256, 70, dtype=np.uint8)) * 1500  # offset just above a wave
    yield "text", text[:150000]  # many short matches
    yield "random", bytes(rng.integers(0, 256, 66000, dtype=np.uint8))  # literals only, more than one 64 KiB fragment
    yield "ints", rng.integers(0, 1000, 40000).astype("<i4").tobytes()  # PLAIN int32 page
    yield "timestamps", (1700000000000 + np.cumsum(rng.integers(0, 50, 30000))).astype("<i8").tobytes()
    yield "doubles", rng.normal(size=4000).astype("<f8").tobytes()
    far = bytearray(rng.integers(0, 4, 140000, dtype=np.uint8) + 97)  # two-bit symbols: matches everywhere
    far[70000:70400] = rng.integers(0, 256, 400, dtype=np.uint8).tobytes()
    far[134000:134400] = far[70000:70400]  # a match 64000 back
    far[10000:10300] = far[70050:70350]
    yield "far_matches", bytes(far)
    yield "mixed", text[:30000] + bytes(rng.integers(0, 256, 5000, dtype=np.uint8)) + b"\0" * 20000 + text[1000:20000]
