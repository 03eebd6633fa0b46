import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Both libraries are built in-tree before any test runs (hipcc cross-compiles without a GPU)."""
    from oracle import build as obuild
    from valida_amd import build

    build.build_vgpu()
    obuild.build_oracle()


@pytest.fixture(scope="session")
def rc():
    import valida_amd as va

    return va.poseidon_round_constants()


@pytest.fixture(scope="session")
def fib25():
    import valida_amd as va

    return va.Workload.fib(25)


@pytest.fixture(scope="session")
def machine():
    import valida_amd as va

    return va.Machine.basic()


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def prover(machine, rc):
    import valida_amd as va

    return va.Prover(machine, rc)


def first_mismatch(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    if a.shape != b.shape:
        return "shape %s vs %s" % (a.shape, b.shape)
    d = np.nonzero(a != b)[0]
    if d.size == 0:
        return None
    return "%d/%d words differ, first at %d: %d vs %d" % (d.size, a.size, d[0], a[d[0]], b[d[0]])
