import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The oracle's OpenMP runtime must wait PASSIVELY and use no more threads than this process may run on.  Test modules import torch at collection
# time, and torch's own libgomp.so.1 is then the copy the oracle library binds to — initialised before oracle/pyoracle.py gets to set these
# variables: with spinning waits on a box whose cores are shared the verifier tests (thousands of tiny parallel regions) took minutes instead of
# seconds.  conftest.py is imported before any test module: set them here.
def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")


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


def pow_trace(n, degree, x0):
    """Trace of the oracle's synthetic POW5 / POW9 AIRs (oracle/chips.hpp): x counts up from x0, y = x^degree."""
    import valida_amd as va

    x = (np.arange(n, dtype=np.uint64) + x0) % va.P
    y = np.array([pow(int(v), degree, va.P) for v in x], dtype=np.uint64)
    return np.stack([x, y], axis=1).astype(np.uint32)


def pow_machine(specs):
    """The same AIRs captured through the vgpu_air_* FFI: specs = [(name, degree, pin_first)], one chip each."""
    import ctypes

    import valida_amd as va

    L, u = va.lib(), ctypes.c_uint32
    m = ctypes.c_void_p()
    assert L.vgpu_machine_new(ctypes.byref(m)) == 0
    codes = []
    for name, degree, pin_first in specs:
        air = ctypes.c_void_p()
        assert L.vgpu_air_new(name.encode(), u(2), u(0), ctypes.byref(air)) == 0
        x = L.vgpu_air_variable(air, u(0), u(0), u(0))
        y = L.vgpu_air_variable(air, u(0), u(1), u(0))
        xn = L.vgpu_air_variable(air, u(0), u(0), u(1))
        p = x
        for _ in range(degree - 1):
            p = L.vgpu_air_mul(air, u(p), u(x))
        L.vgpu_air_assert_zero(air, u(L.vgpu_air_sub(air, u(y), u(p))))
        one = L.vgpu_air_constant(air, u(1))
        step = L.vgpu_air_sub(air, u(xn), u(L.vgpu_air_add(air, u(x), u(one))))
        L.vgpu_air_assert_zero(air, u(L.vgpu_air_mul(air, u(L.vgpu_air_is_transition(air)), u(step))))
        if pin_first:
            three = L.vgpu_air_constant(air, u(3))
            L.vgpu_air_assert_zero(air, u(L.vgpu_air_mul(air, u(L.vgpu_air_is_first_row(air)), u(L.vgpu_air_sub(air, u(x), u(three))))))
        codes.append(L.vgpu_machine_push_air(m, air))
        L.vgpu_air_free(air)
    return va.Machine(m), codes
