import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # PQD_SEED_SHIFT=n: every numpy generator seeded with an integer gets seed + n, so that the tests which compare
    # the library with the oracle on seeded inputs run once more on other inputs (fixtures and known answers are not
    # touched; a test that depends on what one seed happens to produce may then fail for that reason alone).
    shift = int(os.environ.get("PQD_SEED_SHIFT", "0") or 0)
    if shift:
        import numpy as np
        plain = np.random.default_rng

        def shifted(seed=None, *a, **k):
            # (codec_inputs.py regenerates the inputs of the committed compressed fixtures: those stay as they are)
            fixed = sys._getframe(1).f_code.co_filename.endswith("codec_inputs.py")
            if isinstance(seed, (int, np.integer)) and not fixed:
                seed = int(seed) + shift
            return plain(seed, *a, **k)

        np.random.default_rng = shifted


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
