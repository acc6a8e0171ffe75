import os
import sys

import pytest
import torch  # noqa: F401  -- first, so that torch's bundled HIP runtime is the one the process shares with libptx_amd.so (as in bench.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built from oracle/pt_oracle.c on first use."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates ptx contexts on cuda:0; fails loudly if libptx_amd.so is missing."""
    from path_tracer_amd import capi
    capi.load()
    made = []

    def make(**kw):
        c = capi.Context(0, **kw)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
