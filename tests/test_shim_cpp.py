"""The C++ drop-in shim (include/pire_hip/batch_runner.hpp) compiled against the UNMODIFIED reference headers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "bin", "shim_test")
REF_PRESENT = os.path.exists("/root/reference/pire/run.h")


@pytest.mark.skipif(not REF_PRESENT, reason="/root/reference not present (GPU box): the prebuilt binary is used there")
def test_shim_compiles_against_reference_headers():
    from oracle import binding as ob

    ob.build()
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_shim_matches_reference_runner_on_gpu():
    if not os.path.exists(BIN):
        pytest.skip("tests/cpp/bin/shim_test was not built (needs /root/reference at build time)")
    r = subprocess.run([BIN], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "OK(shim" in r.stdout
