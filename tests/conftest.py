import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import rpf_pkg  # noqa: E402

rpf_pkg.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suite tests the in-tree library: hip.py honours RPF_HIP_LIB (profiling variants), which would silently point the
    # tests at another build
    if os.environ.get("RPF_HIP_LIB"):
        raise pytest.UsageError("RPF_HIP_LIB is set (%s): the tests must load raytracer-rpf_amd/lib/librpf_hip.so" % os.environ["RPF_HIP_LIB"])


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hipmod():
    from raytracer_rpf_amd import hip
    hip.load()  # raises if librpf_hip.so was not built: there is no fallback
    return hip


@pytest.fixture(scope="session")
def ctx(hipmod):
    c = hipmod.Context(0)
    yield c
    c.close()
