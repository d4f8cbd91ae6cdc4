import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a per-test time limit where pytest-timeout is installed (it is in the build image): a hung multi-process test (gloo rendezvous, a worker waiting for a
    GPU that another process holds) fails after its limit instead of taking the whole tier with it.  CPU tests: 10 minutes, GPU tests: 25 (the slowest one, the
    C4 oracle comparison, takes 4-6)."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1500 if item.get_closest_marker("gpu") else 600))
