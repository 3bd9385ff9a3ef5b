import os
import sys

import pytest

# Tests that run a tensor-parallel group as several device contexts of ONE process (tests/test_hip_tp_p2p.py's local groups, the
# single-device simulation) give every rank its own stream; the ranks' kernels poll for one another, so two of them must never
# share a hardware queue (the second would sit behind the first until its poll times out).  The HIP runtime maps streams onto 4
# hardware queues by default (GPU_MAX_HW_QUEUES); with the session's other devices alive a 4-rank group then collides on some
# boxes.  Set before the runtime loads; a production group is one process -- one stream -- per rank and is not affected.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope="session")
def odev(oracle):
    return oracle.OracleDevice(thread_num=1, use_avx2=False)


@pytest.fixture(scope="session")
def ca():
    """The product package.  No skip: on the GPU box a missing extension must fail loudly."""
    import crabml_amd
    return crabml_amd


@pytest.fixture(scope="session")
def hdev(ca):
    return ca.HipTensorDevice(device_ordinal=0, debug_named_tensor=False)
