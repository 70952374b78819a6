import os
import sys

import pytest

# torch first: it ships its own copies of the HIP / HSA / RCCL runtimes (torch/lib).  A process that maps the system
# copies first (libgaast_hip.so links /opt/rocm/lib) and torch's afterwards ends up with two HSA runtimes and no visible
# GPU; loaded in this order the library reuses the copies torch already mapped (same sonames), whatever test file or
# test order pytest is given.  A host without torch (tests/cpp/abi_host.c, the Rust shim) only ever sees the system copies.
import torch  # noqa: F401  (import order matters, see above)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
