import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def coracle():
    from oracle.sampler import COracle

    return COracle()


@pytest.fixture(scope="session")
def g2_graph():
    """BASELINE configs[3]'s parent graph (10M nodes / 200M edges requested), generated once per session (about a minute on
    the GPU box's host cores; GCC_AMD_GRAPH_CACHE keeps it between processes of one call)."""
    from gcc_amd.graphgen import powerlaw_graph

    os.environ.setdefault("GCC_AMD_GRAPH_CACHE", "/tmp/graphs")
    return powerlaw_graph(10_000_000, 200_000_000, 0)
