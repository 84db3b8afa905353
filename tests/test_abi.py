"""CPU-side checks of the C ABI library: it loads, exports every symbol include/pqd.h declares, and
validates arguments without touching a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pqd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pqd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import parquet_java_amd as pq
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(pq.lib, n), f"{n} declared in include/pqd.h but not exported"
    assert set(names) == set(pq._native.EXPORTS), "ctypes table and header disagree"
    assert pq.lib.pqd_version() == 100


def test_struct_layout_matches_header():
    import parquet_java_amd as pq
    assert C.sizeof(pq.PageDesc) == 40 and C.sizeof(pq.PageResult) == 16
    assert pq.PageDesc.data_len.offset == 24 and pq.PageDesc.encoding.offset == 36 and pq.PageDesc.flags.offset == 39


def test_status_strings_and_argument_checks_without_gpu():
    import parquet_java_amd as pq
    for s in range(0, 8):
        assert pq.lib.pqd_status_string(s)
    # host-side validation happens before any device call
    assert pq.lib.pqd_unpack_i32(33, None, 0, None, 0) == pq.BAD_ARG
    assert pq.lib.pqd_unpack_i32(3, None, 0, None, 12) == pq.BAD_ARG
    assert pq.lib.pqd_unpack_i32(3, None, 2, None, 8) == pq.EOF
    assert pq.lib.pqd_unpack_i64(65, None, 0, None, 0) == pq.BAD_ARG
    assert pq.lib.pqd_unpack_i32(3, None, 0, None, 0) == pq.OK
    assert b"shorter" in pq.lib.pqd_last_error() or pq.lib.pqd_last_error() is not None
    assert pq.lib.pqd_decode_pages_dev(None, None, 0, pq.ENC_RLE, 4, None) == pq.OK
    assert pq.lib.pqd_decode_pages_dev(None, None, 3, pq.ENC_RLE, 4, None) == pq.BAD_ARG


def test_no_cpu_fallback_when_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import parquet_java_amd as pq
    assert pq.device_count() == 0
    out = np.zeros(8, np.int32)
    data = np.zeros(3, np.uint8)
    st = pq.lib.pqd_unpack_i32(3, data.ctypes.data, 3, out.ctypes.data, 8)
    assert st in (pq.NO_DEVICE, pq.HIP_ERROR)  # fails loudly: nothing decodes on the CPU
    with pytest.raises(pq.PqdError):
        pq.ParquetReadRouter.read(3, pq.ByteBufferInputStream.wrap(data), 8, out)
    with pytest.raises(pq.PqdError):
        pq.init(0)


def test_stream_mirror():
    import parquet_java_amd as pq
    s = pq.ByteBufferInputStream.wrap(bytes(range(10)))
    assert s.available() == 10 and s.read() == 0 and s.position() == 1
    assert s.slice(3).tolist() == [1, 2, 3] and s.position() == 4
    sub = s.sliceStream(2)
    assert sub.available() == 2 and sub.read() == 4 and s.position() == 6
    with pytest.raises(pq.EOFException):
        s.slice(5)
    rest = s.remainingStream()
    assert rest.available() == 4 and s.available() == 0
    with pytest.raises(pq.EOFException):
        s.read()


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "parquet-java_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("no cpu", ""), f"{f} mentions the oracle"
