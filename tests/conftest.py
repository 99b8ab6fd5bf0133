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
    # The tests pin WHICH rows are dense (cold-state tests, trap counters, kernel choices): the automatic re-ranking
    # (pire_hip_config.auto_adapt, on by default in the library) is off here and switched on by the tests that are
    # about it (test_auto_adapt_*).
    from pire_amd import binding as pb

    pb.set_config(auto_adapt=1)
    yield


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


class _Cfg:
    """pire_hip_config through the ABI (pire_amd.binding.set_config), restored when the test ends.  For
    segment_warmup / segment_budget a value of 0 means "really zero" here (the struct's 0 is "the default")."""

    def __init__(self):
        from pire_amd import binding as pb

        self.pb = pb
        self.saved = pb.get_config()

    def set(self, **fields):
        for k in ("segment_warmup", "segment_budget"):
            if k in fields and int(fields[k]) == 0:
                fields[k] = self.pb.NONE
        self.pb.set_config(**{k: int(v) for k, v in fields.items()})

    def unset(self, *names):
        self.pb.set_config(**{k: 0 for k in names})

    def restore(self):
        import ctypes as C

        self.pb._check(self.pb.lib().pire_hip_config_set(C.byref(self.saved)))


@pytest.fixture
def cfg():
    c = _Cfg()
    yield c
    c.restore()
