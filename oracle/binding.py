"""ctypes bindings for the parity checkers.  TEST INFRASTRUCTURE -- not the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

* ``OracleScanner``  -- oracle/liboracle.so, the plain-C restatement (pire_oracle.c)
* ``RefScanner``     -- oracle/_ref/libpire_ref.so, the UNMODIFIED reference library
                        (present wherever `make -C oracle ref` ran; built in the dev
                        container, shipped to the GPU box as a prebuilt .so)
* ``corpus_*``       -- host mirror of the on-device synthetic corpus generator
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libpire_ref.so")
REFERENCE_ROOT = "/root/reference"

FLAG_BEGIN = 1
FLAG_END = 2
BEGIN_MARK = 258
END_MARK = 259

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> None:
    """Compile liboracle.so (always) and oracle/_ref (only where the reference tree exists)."""
    targets = ["oracle"]
    if os.path.exists(os.path.join(REFERENCE_ROOT, "pire", "run.h")):
        targets += ["ref"]
    cmd = ["make", "-C", HERE, "-j8"] + (["-B"] if force else []) + targets
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def _ptr(a: Optional[np.ndarray], typ):
    if a is None:
        return None
    return a.ctypes.data_as(typ)


def _as_text(text) -> np.ndarray:
    if isinstance(text, (bytes, bytearray)):
        return np.frombuffer(bytes(text), dtype=np.uint8)
    a = np.ascontiguousarray(text, dtype=np.uint8)
    return a


def pack_strings(strings: Sequence[bytes]):
    """Concatenate byte strings -> (text u8[], offsets u64[n+1])."""
    offs = np.zeros(len(strings) + 1, dtype=np.uint64)
    if strings:
        offs[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
    text = np.frombuffer(b"".join(strings), dtype=np.uint8) if strings else np.zeros(0, np.uint8)
    return text, offs


# --------------------------------------------------------------------------- oracle (C port)

_oracle_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        L.oracle_scanner_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.oracle_scanner_load.restype = C.c_int
        L.oracle_scanner_free.argtypes = [C.c_void_p]
        for name in ("oracle_size", "oracle_letters_count", "oracle_regexps_count", "oracle_initial_index",
                     "oracle_row_stride", "oracle_header_size"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint32
        L.oracle_empty.argtypes = [C.c_void_p]
        L.oracle_empty.restype = C.c_int
        L.oracle_letter_class.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_letter_class.restype = C.c_uint32
        L.oracle_next_index.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.oracle_next_index.restype = C.c_uint32
        L.oracle_final.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_final.restype = C.c_int
        L.oracle_dead.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_dead.restype = C.c_int
        L.oracle_accepted_regexps.argtypes = [C.c_void_p, C.c_uint32, u64p, C.c_size_t]
        L.oracle_accepted_regexps.restype = C.c_size_t
        L.oracle_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u32p, u8p, C.c_int]
        L.oracle_run.restype = None
        L.oracle_run_shortcut.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u32p, u8p]
        L.oracle_run_shortcut.restype = None
        L.oracle_visit_counts.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u64p]
        L.oracle_visit_counts.restype = None
        L.oracle_prefix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_int, i64p]
        L.oracle_prefix.restype = None
        L.oracle_suffix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_int, i64p]
        L.oracle_suffix.restype = None
        L.corpus_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                  C.c_void_p, C.c_int]
        L.corpus_fill.restype = None
        L.oracle_slow_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.oracle_slow_load.restype = C.c_int
        L.oracle_slow_free.argtypes = [C.c_void_p]
        for name in ("oracle_slow_size", "oracle_slow_letters"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint32
        L.oracle_slow_empty.argtypes = [C.c_void_p]
        L.oracle_slow_empty.restype = C.c_int
        L.oracle_slow_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u8p, u32p]
        L.oracle_slow_run.restype = None
        L.oracle_run_half_final.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u8p, u64p]
        L.oracle_run_half_final.restype = None
        L.oracle_count_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.oracle_count_load.restype = C.c_int
        L.oracle_count_free.argtypes = [C.c_void_p]
        for name in ("oracle_count_size", "oracle_count_letters", "oracle_count_regexps", "oracle_count_initial_index"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint32
        L.oracle_count_letter.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_count_letter.restype = C.c_uint32
        L.oracle_count_next.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u32p]
        L.oracle_count_next.restype = C.c_uint32
        L.oracle_count_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u64p]
        L.oracle_count_run.restype = None
        L.oracle_capture_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u8p, u8p, i64p, i64p]
        L.oracle_capture_run.restype = None
        L.oracle_simple_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.oracle_simple_load.restype = C.c_int
        L.oracle_simple_free.argtypes = [C.c_void_p]
        for name in ("oracle_simple_size", "oracle_simple_initial_index"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint32
        L.oracle_simple_empty.argtypes = [C.c_void_p]
        L.oracle_simple_empty.restype = C.c_int
        L.oracle_simple_next_index.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.oracle_simple_next_index.restype = C.c_uint32
        L.oracle_simple_final.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_simple_final.restype = C.c_int
        L.oracle_simple_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u32p, u8p]
        L.oracle_simple_run.restype = None
        _oracle_lib = L
    return _oracle_lib


class OracleScanner:
    """The C restatement, loaded from a Scanner::Save() blob."""

    def __init__(self, blob: bytes):
        L = oracle_lib()
        self._L = L
        self.blob = bytes(blob)
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        if L.oracle_scanner_load(self.blob, len(self.blob), C.byref(h), err, 256) != 0:
            raise ValueError(err.value.decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_scanner_free(self._h)
            self._h = None

    size = property(lambda s: s._L.oracle_size(s._h))
    letters = property(lambda s: s._L.oracle_letters_count(s._h))
    regexps = property(lambda s: s._L.oracle_regexps_count(s._h))
    initial = property(lambda s: s._L.oracle_initial_index(s._h))
    empty = property(lambda s: bool(s._L.oracle_empty(s._h)))
    row_stride = property(lambda s: s._L.oracle_row_stride(s._h))
    header_size = property(lambda s: s._L.oracle_header_size(s._h))

    def letter_class(self, ch: int) -> int:
        return self._L.oracle_letter_class(self._h, ch)

    def next(self, idx: int, ch: int) -> int:
        return self._L.oracle_next_index(self._h, idx, ch)

    def final(self, idx: int) -> bool:
        return bool(self._L.oracle_final(self._h, idx))

    def dead(self, idx: int) -> bool:
        return bool(self._L.oracle_dead(self._h, idx))

    def accepted(self, idx: int):
        buf = (C.c_uint64 * 256)()
        n = self._L.oracle_accepted_regexps(self._h, idx, buf, 256)
        return [int(buf[i]) for i in range(min(n, 256))]

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, init_idx=None, threads=1, shortcut=False):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        tp = text.ctypes.data if text.size else None
        if shortcut:
            self._L.oracle_run_shortcut(self._h, tp, _ptr(offsets, u64p), n, flags, _ptr(init, u32p),
                                        _ptr(idx, u32p), _ptr(fin, u8p))
        else:
            self._L.oracle_run(self._h, tp, _ptr(offsets, u64p), n, flags, _ptr(init, u32p),
                               _ptr(idx, u32p), _ptr(fin, u8p), threads)
        return idx, fin

    def visit_counts(self, text, offsets, flags=FLAG_BEGIN | FLAG_END) -> np.ndarray:
        """u64[states]: how many text bytes' steps ended in each state (working-set measurements)."""
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        counts = np.zeros(self.size, dtype=np.uint64)
        self._L.oracle_visit_counts(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), len(offsets) - 1,
                                    flags, _ptr(counts, u64p))
        return counts

    def run_strings(self, strings: Sequence[bytes], **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)

    def run_half_final(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        """The table walked as a Pire::HalfFinalScanner: (idx, final, results[n, regexps])."""
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        res = np.zeros((n, max(self.regexps, 1)), dtype=np.uint64)
        self._L.oracle_run_half_final(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                      _ptr(idx, u32p), _ptr(fin, u8p), _ptr(res, u64p))
        return idx, fin, res[:, :self.regexps]

    def prefix(self, text, offsets, longest: bool, through_begin=False, through_end=False):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.empty(n, dtype=np.int64)
        self._L.oracle_prefix(self._h, int(longest), text.ctypes.data if text.size else None,
                              _ptr(offsets, u64p), n, int(through_begin), int(through_end), _ptr(out, i64p))
        return out


def _oracle_suffix(self, text, offsets, longest: bool, through_end=False, through_begin=False):
    """LongestSuffix / ShortestSuffix (run.h:313-362): suffix length per string, -1 = the reference's null."""
    text = _as_text(text)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.empty(n, dtype=np.int64)
    self._L.oracle_suffix(self._h, int(longest), text.ctypes.data if text.size else None,
                          _ptr(offsets, u64p), n, int(through_end), int(through_begin), _ptr(out, i64p))
    return out


OracleScanner.suffix = _oracle_suffix


class OracleSlowScanner:
    """C restatement of Pire::SlowScanner, loaded from SlowScanner::Save() bytes."""

    def __init__(self, blob: bytes):
        L = oracle_lib()
        self._L = L
        self.blob = bytes(blob)
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        if L.oracle_slow_load(self.blob, len(self.blob), C.byref(h), err, 256) != 0:
            raise ValueError(err.value.decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_slow_free(self._h)
            self._h = None

    size = property(lambda s: s._L.oracle_slow_size(s._h))
    letters = property(lambda s: s._L.oracle_slow_letters(s._h))
    empty = property(lambda s: bool(s._L.oracle_slow_empty(s._h)))
    words = property(lambda s: (s.size + 31) // 32)

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        fin = np.empty(n, dtype=np.uint8)
        bits = np.empty((n, self.words), dtype=np.uint32)
        self._L.oracle_slow_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                _ptr(fin, u8p), _ptr(bits, u32p))
        return fin, bits

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class OracleSimpleScanner:
    """C restatement of Pire::SimpleScanner, loaded from SimpleScanner::Save() bytes."""

    def __init__(self, blob: bytes):
        L = oracle_lib()
        self._L = L
        self.blob = bytes(blob)
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        if L.oracle_simple_load(self.blob, len(self.blob), C.byref(h), err, 256) != 0:
            raise ValueError(err.value.decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_simple_free(self._h)
            self._h = None

    size = property(lambda s: s._L.oracle_simple_size(s._h))
    empty = property(lambda s: bool(s._L.oracle_simple_empty(s._h)))
    initial = property(lambda s: s._L.oracle_simple_initial_index(s._h))
    regexps = property(lambda s: 0 if s.empty else 1)

    def next(self, idx: int, ch: int) -> int:
        return self._L.oracle_simple_next_index(self._h, idx, ch)

    def final(self, idx: int) -> bool:
        return bool(self._L.oracle_simple_final(self._h, idx))

    def accepted(self, idx: int):
        return [0] if self.final(idx) else []          # simple.h:66-68

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, init_idx=None, threads=1):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        self._L.oracle_simple_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                  _ptr(init, u32p), _ptr(idx, u32p), _ptr(fin, u8p))
        return idx, fin

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class OracleCountingScanner:
    """C restatement of Pire::CountingScanner (kind 0) / AdvancedCountingScanner (kind 1), from Save() bytes."""

    BASIC, ADVANCED = 0, 1

    def __init__(self, blob: bytes, kind: int):
        L = oracle_lib()
        self._L = L
        self.kind = kind
        self.blob = bytes(blob)
        h = C.c_void_p()
        err = C.create_string_buffer(256)
        if L.oracle_count_load(self.blob, len(self.blob), C.byref(h), err, 256) != 0:
            raise ValueError(err.value.decode())
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_count_free(self._h)
            self._h = None

    size = property(lambda s: s._L.oracle_count_size(s._h))
    letters = property(lambda s: s._L.oracle_count_letters(s._h))
    regexps = property(lambda s: s._L.oracle_count_regexps(s._h))
    initial = property(lambda s: s._L.oracle_count_initial_index(s._h))

    def letter(self, ch: int) -> int:
        return self._L.oracle_count_letter(self._h, ch)

    def next(self, idx: int, letter: int):
        a = C.c_uint32()
        n = self._L.oracle_count_next(self._h, idx, letter, C.byref(a))
        return n, a.value

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        res = np.zeros((n, max(self.regexps, 1)), dtype=np.uint64)
        self._L.oracle_count_run(self._h, self.kind, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n,
                                 flags, _ptr(idx, u32p), _ptr(res, u64p))
        return idx, res[:, :self.regexps]

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)

    def capture(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        """The table walked as a Pire::CapturingScanner: (idx, final, captured, begin, end)."""
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        cap = np.empty(n, dtype=np.uint8)
        b = np.empty(n, dtype=np.int64)
        e = np.empty(n, dtype=np.int64)
        self._L.oracle_capture_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                   _ptr(idx, u32p), _ptr(fin, u8p), _ptr(cap, u8p), _ptr(b, i64p), _ptr(e, i64p))
        return idx, fin, cap, b, e


# --------------------------------------------------------------------------- reference library

_ref_lib = None


def ref_available() -> bool:
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref_lib
    if _ref_lib is None:
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (run `make -C oracle ref` where /root/reference exists)")
        L = C.CDLL(REF_SO)
        L.pire_ref_last_error.restype = C.c_char_p
        L.pire_ref_compile.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_size_t]
        L.pire_ref_compile.restype = C.c_void_p
        L.pire_ref_load.argtypes = [C.c_void_p, C.c_size_t]
        L.pire_ref_load.restype = C.c_void_p
        L.pire_ref_compile_dictionary.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int]
        L.pire_ref_compile_dictionary.restype = C.c_void_p
        L.pire_ref_free.argtypes = [C.c_void_p]
        L.pire_ref_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.pire_ref_save.restype = C.c_size_t
        for name in ("pire_ref_size", "pire_ref_letters", "pire_ref_regexps", "pire_ref_bufsize"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_size_t
        L.pire_ref_empty.argtypes = [C.c_void_p]
        L.pire_ref_empty.restype = C.c_int
        L.pire_ref_initial_index.argtypes = [C.c_void_p]
        L.pire_ref_initial_index.restype = C.c_uint32
        L.pire_ref_next.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.pire_ref_next.restype = C.c_uint32
        L.pire_ref_final.argtypes = [C.c_void_p, C.c_uint32]
        L.pire_ref_final.restype = C.c_int
        L.pire_ref_dead.argtypes = [C.c_void_p, C.c_uint32]
        L.pire_ref_dead.restype = C.c_int
        L.pire_ref_accepted.argtypes = [C.c_void_p, C.c_uint32, u64p, C.c_size_t]
        L.pire_ref_accepted.restype = C.c_size_t
        L.pire_ref_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u32p,
                                   u8p, C.c_int]
        L.pire_ref_run.restype = C.c_int
        L.pire_ref_prefix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_int, i64p]
        L.pire_ref_prefix.restype = C.c_int
        L.pire_ref_suffix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_int, i64p]
        L.pire_ref_suffix.restype = C.c_int
        L.pire_ref_slow_compile.argtypes = [C.c_char_p, C.c_char_p]
        L.pire_ref_slow_compile.restype = C.c_void_p
        L.pire_ref_slow_load.argtypes = [C.c_void_p, C.c_size_t]
        L.pire_ref_slow_load.restype = C.c_void_p
        L.pire_ref_slow_free.argtypes = [C.c_void_p]
        L.pire_ref_slow_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.pire_ref_slow_save.restype = C.c_size_t
        for name in ("pire_ref_slow_size", "pire_ref_slow_letters"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_size_t
        L.pire_ref_slow_empty.argtypes = [C.c_void_p]
        L.pire_ref_slow_empty.restype = C.c_int
        L.pire_ref_slow_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u8p, u32p, C.c_int]
        L.pire_ref_slow_run.restype = C.c_int
        L.pire_ref_half_compile.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_char_p]
        L.pire_ref_half_compile.restype = C.c_void_p
        L.pire_ref_half_load.argtypes = [C.c_void_p, C.c_size_t]
        L.pire_ref_half_load.restype = C.c_void_p
        L.pire_ref_half_free.argtypes = [C.c_void_p]
        L.pire_ref_half_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.pire_ref_half_save.restype = C.c_size_t
        for name in ("pire_ref_half_size", "pire_ref_half_regexps"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_size_t
        L.pire_ref_half_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u8p, u64p, C.c_int]
        L.pire_ref_half_run.restype = C.c_int
        L.pire_ref_count_compile.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_char_p]
        L.pire_ref_count_compile.restype = C.c_void_p
        L.pire_ref_count_load.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.pire_ref_count_load.restype = C.c_void_p
        L.pire_ref_count_free.argtypes = [C.c_void_p]
        L.pire_ref_count_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.pire_ref_count_save.restype = C.c_size_t
        for name in ("pire_ref_count_size", "pire_ref_count_regexps", "pire_ref_count_letters"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_size_t
        L.pire_ref_count_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u64p, C.c_int]
        L.pire_ref_count_run.restype = C.c_int
        L.pire_ref_capture_compile.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.pire_ref_capture_compile.restype = C.c_void_p
        L.pire_ref_capture_load.argtypes = [C.c_void_p, C.c_size_t]
        L.pire_ref_capture_load.restype = C.c_void_p
        L.pire_ref_capture_free.argtypes = [C.c_void_p]
        L.pire_ref_capture_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.pire_ref_capture_save.restype = C.c_size_t
        L.pire_ref_capture_size.argtypes = [C.c_void_p]
        L.pire_ref_capture_size.restype = C.c_size_t
        L.pire_ref_capture_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u8p, u8p, i64p, i64p]
        L.pire_ref_capture_run.restype = C.c_int
        L.pire_ref_simple_compile.argtypes = [C.c_char_p, C.c_char_p]
        L.pire_ref_simple_compile.restype = C.c_void_p
        L.pire_ref_simple_empty.argtypes = []
        L.pire_ref_simple_empty.restype = C.c_void_p
        L.pire_ref_simple_load.argtypes = [C.c_void_p, C.c_size_t]
        L.pire_ref_simple_load.restype = C.c_void_p
        L.pire_ref_simple_free.argtypes = [C.c_void_p]
        L.pire_ref_simple_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.pire_ref_simple_save.restype = C.c_size_t
        for name in ("pire_ref_simple_size", "pire_ref_simple_regexps"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_size_t
        L.pire_ref_simple_empty_flag.argtypes = [C.c_void_p]
        L.pire_ref_simple_empty_flag.restype = C.c_int
        L.pire_ref_simple_initial.argtypes = [C.c_void_p]
        L.pire_ref_simple_initial.restype = C.c_uint32
        L.pire_ref_simple_next.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.pire_ref_simple_next.restype = C.c_uint32
        L.pire_ref_simple_final.argtypes = [C.c_void_p, C.c_uint32]
        L.pire_ref_simple_final.restype = C.c_int
        L.pire_ref_simple_run.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_uint32, u32p, u32p, u8p, C.c_int]
        L.pire_ref_simple_run.restype = C.c_int
        _ref_lib = L
    return _ref_lib


class RefSlowScanner:
    """The real Pire::SlowScanner behind a C ABI."""

    def __init__(self, handle):
        self._L = ref_lib()
        if not handle:
            raise ValueError("reference: " + self._L.pire_ref_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def compile(cls, pattern, options=""):
        L = ref_lib()
        p = pattern.encode("latin-1") if isinstance(pattern, str) else pattern
        return cls(L.pire_ref_slow_compile(p, options.encode()))

    @classmethod
    def load(cls, blob: bytes):
        L = ref_lib()
        return cls(L.pire_ref_slow_load(bytes(blob), len(blob)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pire_ref_slow_free(self._h)
            self._h = None

    size = property(lambda s: s._L.pire_ref_slow_size(s._h))
    letters = property(lambda s: s._L.pire_ref_slow_letters(s._h))
    empty = property(lambda s: bool(s._L.pire_ref_slow_empty(s._h)))
    words = property(lambda s: (s.size + 31) // 32)

    def save(self) -> bytes:
        n = self._L.pire_ref_slow_save(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.pire_ref_slow_save(self._h, buf, n)
        return buf.raw

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, threads=1):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        fin = np.empty(n, dtype=np.uint8)
        bits = np.empty((n, self.words), dtype=np.uint32)
        rc = self._L.pire_ref_slow_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                       _ptr(fin, u8p), _ptr(bits, u32p), threads)
        if rc != 0:
            raise RuntimeError(self._L.pire_ref_last_error().decode())
        return fin, bits

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class RefCapturingScanner:
    """The real Pire::CapturingScanner behind a C ABI (built like tests/capture_ut.cpp:39-53)."""

    def __init__(self, handle):
        self._L = ref_lib()
        if not handle:
            raise ValueError("reference: " + self._L.pire_ref_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def compile(cls, pattern, index, options="i"):
        p = pattern.encode("latin-1") if isinstance(pattern, str) else pattern
        return cls(ref_lib().pire_ref_capture_compile(p, index, options.encode()))

    @classmethod
    def load(cls, blob: bytes):
        return cls(ref_lib().pire_ref_capture_load(bytes(blob), len(blob)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pire_ref_capture_free(self._h)
            self._h = None

    size = property(lambda s: s._L.pire_ref_capture_size(s._h))

    def save(self) -> bytes:
        n = self._L.pire_ref_capture_save(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.pire_ref_capture_save(self._h, buf, n)
        return buf.raw

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        cap = np.empty(n, dtype=np.uint8)
        b = np.empty(n, dtype=np.int64)
        e = np.empty(n, dtype=np.int64)
        self._L.pire_ref_capture_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                     _ptr(idx, u32p), _ptr(fin, u8p), _ptr(cap, u8p), _ptr(b, i64p), _ptr(e, i64p))
        return idx, fin, cap, b, e

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class RefCountingScanner:
    """The real Pire::CountingScanner (kind 0) / AdvancedCountingScanner (kind 1) behind a C ABI."""

    BASIC, ADVANCED = 0, 1

    def __init__(self, handle, kind):
        self._L = ref_lib()
        if not handle:
            raise ValueError("reference: " + self._L.pire_ref_last_error().decode())
        self._h = C.c_void_p(handle)
        self.kind = kind

    @classmethod
    def compile(cls, kind, res, seps, options="u"):
        L = ref_lib()
        enc = lambda p: p.encode("utf-8") if isinstance(p, str) else p
        a = (C.c_char_p * len(res))(*[enc(p) for p in res])
        b = (C.c_char_p * len(seps))(*[enc(p) for p in seps])
        return cls(L.pire_ref_count_compile(kind, a, b, len(res), options.encode()), kind)

    @classmethod
    def load(cls, kind, blob: bytes):
        return cls(ref_lib().pire_ref_count_load(kind, bytes(blob), len(blob)), kind)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pire_ref_count_free(self._h)
            self._h = None

    size = property(lambda s: s._L.pire_ref_count_size(s._h))
    regexps = property(lambda s: s._L.pire_ref_count_regexps(s._h))
    letters = property(lambda s: s._L.pire_ref_count_letters(s._h))

    def save(self) -> bytes:
        n = self._L.pire_ref_count_save(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.pire_ref_count_save(self._h, buf, n)
        return buf.raw

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, threads=1):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        res = np.zeros((n, max(self.regexps, 1)), dtype=np.uint64)
        self._L.pire_ref_count_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                   _ptr(idx, u32p), _ptr(res, u64p), threads)
        return idx, res[:, :self.regexps]

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class RefHalfFinalScanner:
    """The real Pire::HalfFinalScanner behind a C ABI (built like tests/count_ut.cpp:503-527)."""

    GREEDY_SIMPLE, GREEDY, NONGREEDY_SIMPLE, NONGREEDY, NONGREEDY_NOINTERSECT, PLAIN = range(6)

    def __init__(self, handle):
        self._L = ref_lib()
        if not handle:
            raise ValueError("reference: " + self._L.pire_ref_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def compile(cls, patterns, modes, options="u"):
        L = ref_lib()
        pats = (C.c_char_p * len(patterns))(*[p.encode("utf-8") if isinstance(p, str) else p for p in patterns])
        md = (C.c_int * len(modes))(*modes)
        return cls(L.pire_ref_half_compile(pats, md, len(patterns), options.encode()))

    @classmethod
    def load(cls, blob: bytes):
        return cls(ref_lib().pire_ref_half_load(bytes(blob), len(blob)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pire_ref_half_free(self._h)
            self._h = None

    size = property(lambda s: s._L.pire_ref_half_size(s._h))
    regexps = property(lambda s: s._L.pire_ref_half_regexps(s._h))

    def save(self) -> bytes:
        n = self._L.pire_ref_half_save(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.pire_ref_half_save(self._h, buf, n)
        return buf.raw

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, threads=1):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        res = np.zeros((n, max(self.regexps, 1)), dtype=np.uint64)
        self._L.pire_ref_half_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                  _ptr(idx, u32p), _ptr(fin, u8p), _ptr(res, u64p), threads)
        return idx, fin, res[:, :self.regexps]

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class RefSimpleScanner:
    """The real Pire::SimpleScanner behind a C ABI."""

    def __init__(self, handle):
        self._L = ref_lib()
        if not handle:
            raise ValueError("reference: " + self._L.pire_ref_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def compile(cls, pattern, options=""):
        L = ref_lib()
        p = pattern.encode("latin-1") if isinstance(pattern, str) else pattern
        return cls(L.pire_ref_simple_compile(p, options.encode()))

    @classmethod
    def empty_scanner(cls):
        return cls(ref_lib().pire_ref_simple_empty())

    @classmethod
    def load(cls, blob: bytes):
        L = ref_lib()
        return cls(L.pire_ref_simple_load(bytes(blob), len(blob)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pire_ref_simple_free(self._h)
            self._h = None

    size = property(lambda s: s._L.pire_ref_simple_size(s._h))
    regexps = property(lambda s: s._L.pire_ref_simple_regexps(s._h))
    empty = property(lambda s: bool(s._L.pire_ref_simple_empty_flag(s._h)))
    initial = property(lambda s: s._L.pire_ref_simple_initial(s._h))

    def save(self) -> bytes:
        n = self._L.pire_ref_simple_save(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.pire_ref_simple_save(self._h, buf, n)
        return buf.raw

    def next(self, idx: int, ch: int) -> int:
        return self._L.pire_ref_simple_next(self._h, idx, ch)

    def final(self, idx: int) -> bool:
        return bool(self._L.pire_ref_simple_final(self._h, idx))

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, init_idx=None, threads=1):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        self._L.pire_ref_simple_run(self._h, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n, flags,
                                    _ptr(init, u32p), _ptr(idx, u32p), _ptr(fin, u8p), threads)
        return idx, fin

    def run_strings(self, strings, **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)


class RefScanner:
    """The real Pire::Scanner (+ NonrelocScanner twin) behind a C ABI."""

    SCANNER = 0
    NONRELOC = 1

    def __init__(self, handle):
        self._L = ref_lib()
        if not handle:
            raise ValueError("reference: " + self._L.pire_ref_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def compile(cls, patterns: Sequence[str], options: Optional[Sequence[str]] = None, glue_max: int = 0):
        L = ref_lib()
        n = len(patterns)
        pats = (C.c_char_p * n)(*[p.encode("latin-1") if isinstance(p, str) else p for p in patterns])
        opts = (C.c_char_p * n)(*[(o or "").encode() for o in (options or [""] * n)])
        return cls(L.pire_ref_compile(pats, opts, n, glue_max))

    @classmethod
    def compile_dictionary(cls, words: Sequence[bytes], surround: bool = False, utf8: bool = False):
        """samples/blacklist/blacklist.cpp:65-76: the words as fixed strings joined with |=, wrapped as the sample wraps
        them (scheme, subdomains, path) or -- `surround` -- searched anywhere in the text (Fsm::Surround)."""
        L = ref_lib()
        arr = (C.c_char_p * len(words))(*[bytes(w) for w in words])
        h = L.pire_ref_compile_dictionary(arr, len(words), 2 if utf8 else 1 if surround else 0)
        if not h:
            raise RuntimeError(L.pire_ref_last_error().decode())
        return cls(h)

    @classmethod
    def load(cls, blob: bytes):
        L = ref_lib()
        return cls(L.pire_ref_load(bytes(blob), len(blob)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pire_ref_free(self._h)
            self._h = None

    size = property(lambda s: s._L.pire_ref_size(s._h))
    letters = property(lambda s: s._L.pire_ref_letters(s._h))
    regexps = property(lambda s: s._L.pire_ref_regexps(s._h))
    bufsize = property(lambda s: s._L.pire_ref_bufsize(s._h))
    initial = property(lambda s: s._L.pire_ref_initial_index(s._h))
    empty = property(lambda s: bool(s._L.pire_ref_empty(s._h)))

    def save(self) -> bytes:
        n = self._L.pire_ref_save(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.pire_ref_save(self._h, buf, n)
        return buf.raw

    def next(self, idx: int, ch: int) -> int:
        return self._L.pire_ref_next(self._h, idx, ch)

    def final(self, idx: int) -> bool:
        return bool(self._L.pire_ref_final(self._h, idx))

    def dead(self, idx: int) -> bool:
        return bool(self._L.pire_ref_dead(self._h, idx))

    def accepted(self, idx: int):
        buf = (C.c_uint64 * 256)()
        n = self._L.pire_ref_accepted(self._h, idx, buf, 256)
        return [int(buf[i]) for i in range(min(n, 256))]

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, init_idx=None, kind=0, threads=1):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        rc = self._L.pire_ref_run(self._h, kind, text.ctypes.data if text.size else None, _ptr(offsets, u64p), n,
                                  flags, _ptr(init, u32p), _ptr(idx, u32p), _ptr(fin, u8p), threads)
        if rc != 0:
            raise RuntimeError(self._L.pire_ref_last_error().decode())
        return idx, fin

    def run_strings(self, strings: Sequence[bytes], **kw):
        text, offs = pack_strings(strings)
        return self.run(text, offs, **kw)

    def prefix(self, text, offsets, longest: bool, through_begin=False, through_end=False):
        text = _as_text(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.empty(n, dtype=np.int64)
        rc = self._L.pire_ref_prefix(self._h, int(longest), text.ctypes.data if text.size else None,
                                     _ptr(offsets, u64p), n, int(through_begin), int(through_end), _ptr(out, i64p))
        if rc != 0:
            raise RuntimeError(self._L.pire_ref_last_error().decode())
        return out


def _ref_suffix(self, text, offsets, longest: bool, through_end=False, through_begin=False):
    text = _as_text(text)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    out = np.empty(n, dtype=np.int64)
    rc = self._L.pire_ref_suffix(self._h, int(longest), text.ctypes.data if text.size else None,
                                 _ptr(offsets, u64p), n, int(through_end), int(through_begin), _ptr(out, i64p))
    if rc != 0:
        raise RuntimeError(self._L.pire_ref_last_error().decode())
    return out


RefScanner.suffix = _ref_suffix


# --------------------------------------------------------------------------- corpus (host mirror)

CORPUS_MAX_PLANTS = 16
CORPUS_PLANT_BYTES = 64


class CorpusPlants(C.Structure):
    _fields_ = [
        ("nplants", C.c_uint32),
        ("len", C.c_uint32 * CORPUS_MAX_PLANTS),
        ("at_tail", C.c_uint32 * CORPUS_MAX_PLANTS),
        ("bytes", (C.c_uint8 * CORPUS_PLANT_BYTES) * CORPUS_MAX_PLANTS),
    ]


def make_plants(plants: Sequence[tuple]) -> CorpusPlants:
    """plants: sequence of (witness_bytes, at_tail_bool)."""
    p = CorpusPlants()
    assert len(plants) <= CORPUS_MAX_PLANTS
    p.nplants = len(plants)
    for i, (w, tail) in enumerate(plants):
        assert len(w) <= CORPUS_PLANT_BYTES
        p.len[i] = len(w)
        p.at_tail[i] = 1 if tail else 0
        for k, b in enumerate(w):
            p.bytes[i][k] = b
    return p


def corpus_fill(seed: int, first: int, count: int, length: int, plants: Optional[CorpusPlants] = None,
                threads: int = 1) -> np.ndarray:
    """Strings [first, first+count) of the synthetic corpus as a (count, length) u8 array."""
    out = np.empty((count, length), dtype=np.uint8)
    L = oracle_lib()
    L.corpus_fill(seed, first, count, length, length, C.byref(plants) if plants is not None else None,
                  out.ctypes.data, threads)
    return out
