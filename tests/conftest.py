import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(params=["generic", "sliced"])
def dict_path(request, monkeypatch):
    """Runs a test once per dictionary-page kernel path of the library (PQD_DICT_PATH is read on
    every call; monkeypatch.setenv reaches the C library's getenv through os.putenv)."""
    monkeypatch.setenv("PQD_DICT_PATH", request.param)
    return request.param


@pytest.fixture(params=["generic", "planned", "planned-no-small"])
def rle_path(request, monkeypatch):
    """Runs a test once per kernel path of plain hybrid pages (PQD_RLE_PATH / PQD_RLE_SMALL, read on every
    call): the generic kernel alone; the planned kernels with the small-page kernel in front; the planned
    kernels alone."""
    monkeypatch.setenv("PQD_RLE_PATH", "generic" if request.param == "generic" else "planned")
    monkeypatch.setenv("PQD_RLE_SMALL", "0" if request.param == "planned-no-small" else "1")
    return request.param
