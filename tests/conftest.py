import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_path(name):
    p = os.path.join(GOLDEN, name)
    if os.path.exists(p):
        return p
    if os.path.exists(p + ".gz"):
        return p + ".gz"
    raise FileNotFoundError(p)


def golden_text(name):
    p = golden_path(name)
    if p.endswith(".gz"):
        with gzip.open(p, "rt") as f:
            return f.read()
    with open(p) as f:
        return f.read()


def golden_lines(name):
    """Lines without the trailing newline (the final 14-column rows keep their trailing TAB)."""
    return golden_text(name).split("\n")[:-1]


@pytest.fixture(scope="session")
def golden():
    class G:
        path = staticmethod(golden_path)
        text = staticmethod(golden_text)
        lines = staticmethod(golden_lines)
    return G
