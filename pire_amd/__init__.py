"""pire_amd -- MI355X (gfx950) implementation of Pire's Runner(...).Run() scan path.

The product is the C-ABI shared library ``pire_amd/libpire_hip.so`` (sources in ``pire_amd/csrc``, ABI in
``include/pire_hip.h``) plus the header-only C++ shim ``include/pire_hip/batch_runner.hpp`` that keeps the
``Pire::Scanner`` / ``Pire::Runner`` vocabulary.  This Python package is a thin ctypes binding used by the
test-suite and ``bench.py``; it contains no scan logic and no CPU fallback.
"""
from .binding import (  # noqa: F401
    FLAG_BEGIN,
    FLAG_END,
    FLAG_ON_DEVICE,
    FLAG_GENERIC,
    PireHipError,
    Table,
    SlowTable,
    CountingTable,
    BatchRunner,
    MultiRunner,
    build,
    corpus_fill_device,
    device_count,
    lib,
    lib_path,
)
