import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build the checker (liboracle.so) and the product (libpire_hip.so) if they are not there yet."""
    from oracle import binding as ob
    import pire_amd

    if not os.path.exists(ob.ORACLE_SO):
        ob.build()
    if not os.path.exists(pire_amd.lib_path()):
        pire_amd.build()
    yield


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
