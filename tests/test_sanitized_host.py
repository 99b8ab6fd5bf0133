"""The host side of libpire_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer ("run host
code under ASan in CI"; the reference's analogue is its checked build, tests/Makefile.am:40-48, and ValidateSkip,
pire/scanners/multi.h:925-934).  `make -C pire_amd/csrc asan` compiles the same sources with -fsanitize=address,undefined
(device code untouched) into pire_amd/libpire_hip_asan.so; the host-only test modules -- blob ingestion and its fuzzing,
the accessors, Glue, the mode relations: everything that parses untrusted bytes or juggles the host tables -- are then run
in a child interpreter with the sanitizer runtime preloaded and PIRE_HIP_LIB pointing at that library.  On a GPU box
tools/gpu_scripts/r04_final.sh runs the GPU stress tests the same way (profiles/r04_asan_*.log)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
MODULES = ["tests/test_abi.py", "tests/test_fuzz_blobs.py", "tests/test_glue.py", "tests/test_mode_relations.py"]


def asan_runtime():
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return found[-1] if found else None


def sanitized_env():
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan_runtime(), PIRE_HIP_LIB=os.path.join(ROOT, "pire_amd", "libpire_hip_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",   # (CPython itself leaks by design)
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONPATH=ROOT)
    return env


@pytest.mark.skipif(not os.path.exists(HIPCC) or asan_runtime() is None, reason="hipcc / the ASan runtime is not installed")
def test_host_only_suites_are_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "pire_amd", "csrc"), "asan"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + MODULES,
                       cwd=ROOT, env=sanitized_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    out = r.stdout
    log = os.path.join(ROOT, "gpurun_out", "asan_cpu_suite.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as f:
        f.write(out)
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0 and " passed" in out, out[-3000:]
    # the child really ran the sanitized build
    probe = subprocess.run([sys.executable, "-c", "from pire_amd import binding as b; print(b.lib_path()); b.lib()"], cwd=ROOT,
                           env=sanitized_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert probe.returncode == 0 and "libpire_hip_asan.so" in probe.stdout, probe.stdout[-2000:]
    syms = subprocess.run(["nm", "-D", os.path.join(ROOT, "pire_amd", "libpire_hip_asan.so")], stdout=subprocess.PIPE, text=True)
    assert "__asan_init" in syms.stdout and "__ubsan_handle" in syms.stdout
