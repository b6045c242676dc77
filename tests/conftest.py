import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the checker's one parallel region (rows of a roll are independent: oracle/haf_oracle.c, HAFO_THREADS; per-row arithmetic and
    # order untouched) on a few cores: the decision stage of the 4096- and 8964-SV parity cases is most of the suite's wall time
    os.environ.setdefault("HAFO_THREADS", str(max(1, min(8, (os.cpu_count() or 2) // 2))))
    # The product library and the checker are built in-tree; build them when the suite runs on a fresh checkout
    # (hipcc cross-compiles gfx950 without a GPU, ~15 s).  Nothing here falls back to another implementation.
    from haf_grasping_amd import build as _b
    if not os.path.exists(_b.LIB):
        _b.build()
    from oracle import oracle as _o
    if not os.path.exists(os.path.join(os.path.dirname(_o.__file__), "libhaforacle.so")):
        _o.build()


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
