"""Runs the C++ reader mirror's own test program (include/pqd_readers.hpp, tests/cpp/test_readers.cpp):
known answers from the reference's tests, decoded through the C ABI from compiled host code."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_readers")


def test_mirror_header_cites_the_reference():
    text = open(os.path.join(ROOT, "include", "pqd_readers.hpp")).read()
    for name in ("ParquetReadRouter", "RunLengthBitPackingHybridValuesReader", "DictionaryValuesReader",
                 "DeltaBinaryPackingValuesReader", "ByteStreamSplitValuesReader", "ByteBitPackingValuesReader"):
        assert ("class " + name in text) or ("struct " + name in text), name
    assert ".java:" in text


@pytest.mark.gpu
def test_cpp_reader_mirror_known_answers():
    assert os.path.exists(BIN), "tests/cpp/test_readers is not built: run `make`"
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "all reader-mirror checks passed" in p.stdout
