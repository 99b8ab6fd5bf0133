"""ctypes binding of libpire_hip.so (include/pire_hip.h).  No scan logic lives here."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, "libpire_hip.so")
if os.environ.get("PIRE_HIP_LIB"):   # an alternative build of the same library (tools/ab: tuning / older builds)
    _LIB_PATH = os.path.abspath(os.environ["PIRE_HIP_LIB"])

FLAG_BEGIN = 1
FLAG_END = 2
FLAG_ON_DEVICE = 4
FLAG_GENERIC = 8
FLAG_HOST_OFFSETS = 16
FLAG_NO_PEEK = 32

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class PireHipError(RuntimeError):
    """Counterpart of Pire::Error (pire/stub/stl.h:213-217) for the C ABI's negative return codes."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"pire_hip error {code}: {msg}")
        self.code = code


class TableInfo(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("states", C.c_uint32),
        ("letters", C.c_uint32),
        ("regexps", C.c_uint32),
        ("initial", C.c_uint32),
        ("empty", C.c_uint32),
        ("header_size", C.c_uint32),
        ("row_stride", C.c_uint32),
        ("hot_states", C.c_uint32),
        ("lds_table_bytes", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("ref_buf_size", C.c_uint64),
        ("last_trap_samples", C.c_uint64),
        ("adaptations", C.c_uint32),
        ("compact_states", C.c_uint32),
        ("scanner_type", C.c_uint32),
        ("zip_full_states", C.c_uint32),
        ("wide_states", C.c_uint32),
        ("wide_lds_bytes", C.c_uint32),
        ("outside_dense_share", C.c_float),
        ("outside_wide_share", C.c_float),
        ("shares_measured", C.c_uint32),
        ("zip_outside_share", C.c_float),
        ("last_wide_trap_chunks", C.c_uint64),
        ("wide_outside_chunk_share", C.c_float),
        ("zip_plain_outside_share", C.c_float),
    ]


class CountingInfo(C.Structure):
    _fields_ = [("states", C.c_uint32), ("letters", C.c_uint32), ("regexps", C.c_uint32), ("initial", C.c_uint32)]


class SlowInfo(C.Structure):
    _fields_ = [("states", C.c_uint32), ("letters", C.c_uint32), ("start", C.c_uint32), ("words", C.c_uint32),
                ("empty", C.c_uint32), ("reserved", C.c_uint32), ("mask_bytes", C.c_uint64)]


def lib_path() -> str:
    return _LIB_PATH


def build(force: bool = False) -> str:
    """Compile libpire_hip.so in-tree for gfx950 with hipcc (works without a GPU)."""
    cmd = ["make", "-j8", "-C", os.path.join(HERE, "csrc")] + (["-B"] if force else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libpire_hip.so failed:\n" + r.stdout)
    return _LIB_PATH


_lib = None

class Config(C.Structure):
    """pire_hip_config (include/pire_hip.h): the library's runtime knobs, set through the ABI."""
    _fields_ = [
        ("size", C.c_uint32),
        ("tiled_variant", C.c_uint32),
        ("checked", C.c_uint32),
        ("no_compact", C.c_uint32),
        ("prior_flat", C.c_uint32),
        ("ragged_act_always", C.c_uint32),
        ("no_ragged_act", C.c_uint32),
        ("no_segments", C.c_uint32),
        ("segment_no_grid", C.c_uint32),
        ("segment_stats", C.c_uint32),
        ("segment_modes", C.c_uint32),
        ("segment_bytes", C.c_uint64),
        ("segment_warmup", C.c_uint64),
        ("segment_budget", C.c_uint64),
        ("host_chunk_bytes", C.c_uint64),
        ("host_one_shot", C.c_uint32),
        ("no_rccl", C.c_uint32),
        ("slow_sets_in_memory", C.c_uint32),
        ("slow_no_list", C.c_uint32),
        ("auto_adapt", C.c_uint32),
        ("auto_adapt_min_traps", C.c_uint32),
        ("ragged_variant", C.c_uint32),
        ("host_staging", C.c_uint32),
        ("no_offsets_peek", C.c_uint32),
        ("segment_no_pair", C.c_uint32),
        ("segment_no_product", C.c_uint32),
        ("segment_no_derive", C.c_uint32),
        ("no_length_order", C.c_uint32),
        ("capture_by_length", C.c_uint32),
        ("force_rccl", C.c_uint32),
        ("counting_variant", C.c_uint32),
        ("slow_stats", C.c_uint32),
        ("walk_variant", C.c_uint32),
        ("selftest", C.c_uint32),
        ("zip_variant", C.c_uint32),
    ]


NONE = (1 << 64) - 1   # PIRE_HIP_SEGMENT_WARMUP_NONE / PIRE_HIP_SEGMENT_BUDGET_NONE: "really zero", 0 being "the default"


ABI_VERSION = 6   # include/pire_hip.h PIRE_HIP_ABI_VERSION

# every symbol include/pire_hip.h declares: (name, restype, argtypes)
ABI = [
    ("pire_hip_build_info", C.c_char_p, []),
    ("pire_hip_config_get", C.c_int, [C.POINTER(Config)]),
    ("pire_hip_config_set", C.c_int, [C.POINTER(Config)]),
    ("pire_hip_table_create", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_table_mmap", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("pire_hip_table_create_from_file", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("pire_hip_table_glue", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_table_glue_gpu", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_table_upload", C.c_int, [C.c_void_p]),
    ("pire_hip_table_adapt", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("pire_hip_table_destroy", None, [C.c_void_p]),
    ("pire_hip_table_get_info", C.c_int, [C.c_void_p, C.POINTER(TableInfo)]),
    ("pire_hip_table_final", C.c_int, [C.c_void_p, C.c_uint32]),
    ("pire_hip_table_dead", C.c_int, [C.c_void_p, C.c_uint32]),
    ("pire_hip_table_accepted_regexps", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    ("pire_hip_table_letter_class", C.c_int, [C.c_void_p, C.c_uint32]),
    ("pire_hip_table_next", C.c_int64, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("pire_hip_table_layout", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_table_wide_layout", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint32)]),
    ("pire_hip_table_get_info_sized", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("pire_hip_abi_version", C.c_uint32, []),
    ("pire_hip_selftested_kernels", C.c_char_p, []),
    ("pire_hip_table_config_set", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pire_hip_table_config_get", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pire_hip_table_zip_layout", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("pire_hip_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_run_strided", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_step", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    ("pire_hip_run_half_final", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_prefix", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                  C.c_void_p, C.c_void_p]),
    ("pire_hip_suffix", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                  C.c_void_p, C.c_void_p]),
    ("pire_hip_slow_table_create", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_slow_table_destroy", None, [C.c_void_p]),
    ("pire_hip_slow_table_get_info", C.c_int, [C.c_void_p, C.POINTER(SlowInfo)]),
    ("pire_hip_slow_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    ("pire_hip_slow_run_strided", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_counting_table_create", C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_counting_table_destroy", None, [C.c_void_p]),
    ("pire_hip_counting_table_get_info", C.c_int, [C.c_void_p, C.POINTER(CountingInfo)]),
    ("pire_hip_counting_table_forms", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32 * 8)]),
    ("pire_hip_counting_run", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    ("pire_hip_capture_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    ("pire_hip_multi_destroy", None, [C.c_void_p]),
    ("pire_hip_multi_device_count", C.c_int, [C.c_void_p]),
    ("pire_hip_multi_reduce_backend", C.c_char_p, [C.c_void_p]),
    ("pire_hip_multi_run_strided", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("pire_hip_multi_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("pire_hip_multi_run_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_multi_last_split", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    ("pire_hip_multi_run_strided_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                                  C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_last_kernel", C.c_char_p, []),
    ("pire_hip_last_kernel_symbol", C.c_char_p, []),
    ("pire_hip_set_timing", C.c_int, [C.c_int]),
    ("pire_hip_last_kernel_ms", C.c_float, []),
    ("pire_hip_last_error", C.c_char_p, []),
    ("pire_hip_device_count", C.c_int, []),
    ("pire_hip_run_pair", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_run_pair_strided", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                            C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pire_hip_table_check_failures", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("pire_hip_host_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_host_free", None, [C.c_void_p]),
    ("pire_hip_device_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("pire_hip_device_free", None, [C.c_void_p]),
    ("pire_hip_copy_to_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("pire_hip_copy_to_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("pire_hip_memset_device", C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    ("pire_hip_stream_synchronize", C.c_int, [C.c_void_p]),
    ("pire_hip_corpus_fill", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.c_void_p, C.c_void_p]),
]


def lib():
    """Load the native library.  Fails loudly if it has not been built: there is no Python/CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pire_amd/csrc` (hipcc, gfx950). pire_amd has no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64.so.7; a process must hold exactly ONE HIP runtime or the second
        # one finds no devices.  Let torch's copy load first so that our NEEDED libamdhip64.so.7 binds to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        for name, res, args in ABI:
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("PIRE_HIP_LIB"):   # an older build under tools/ab: entry points added since are absent
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        # the structs declared above are THIS header's (include/pire_hip.h PIRE_HIP_ABI_VERSION): a library of another version
        # would write a pire_hip_table_info of another size into them
        if not os.environ.get("PIRE_HIP_LIB") and hasattr(L, "pire_hip_abi_version") and L.pire_hip_abi_version() != ABI_VERSION:
            raise ImportError(f"{_LIB_PATH} is ABI version {L.pire_hip_abi_version()}, this binding is {ABI_VERSION}: rebuild")
        _lib = L
    return _lib


def get_config() -> Config:
    c = Config()
    c.size = C.sizeof(Config)
    _check(lib().pire_hip_config_get(C.byref(c)))
    return c


def set_config(**fields) -> Config:
    """Change the named pire_hip_config fields (process-wide); returns the configuration as it was before."""
    old = get_config()
    new = get_config()
    for k, v in fields.items():
        if k not in dict(Config._fields_) or k == "size":
            raise KeyError(k)
        setattr(new, k, int(v))
    _check(lib().pire_hip_config_set(C.byref(new)))
    return old


class config:
    """`with config(tiled_variant=22): ...` -- the fields changed inside the block, restored after it."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.old = set_config(**self.fields)
        return self

    def __exit__(self, *exc):
        _check(lib().pire_hip_config_set(C.byref(self.old)))
        return False


def device_count() -> int:
    return lib().pire_hip_device_count()


def _check(rc: int):
    if rc < 0:
        raise PireHipError(rc, lib().pire_hip_last_error().decode(errors="replace"))
    return rc


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


class Table:
    """An ingested Pire::Scanner / NonrelocScanner / SimpleScanner (from Save() bytes).  Mirrors the public getters."""

    def __init__(self, blob: bytes = None, _handle=None):
        L = lib()
        h = _handle
        if h is None:
            h = C.c_void_p()
            blob = bytes(blob)
            _check(L.pire_hip_table_create(blob, len(blob), C.byref(h)))
        self._h = h
        self._info = None

    @property
    def info(self):
        """pire_hip_table_info (fetched on first use: for a glued table it triggers the dense-row ranking)."""
        if self._info is None:
            info = TableInfo()
            _check(lib().pire_hip_table_get_info(self._h, C.byref(info)))
            self._info = info
        return self._info

    @classmethod
    def from_file(cls, path: str):
        """Scanner::Mmap deployment flow (samples/blacklist/blacklist.cpp:86-93): ingest a file written by Save()."""
        h = C.c_void_p()
        _check(lib().pire_hip_table_create_from_file(os.fsencode(path), C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def mmap(cls, image: bytes):
        """Ingest the scanner at the start of `image`; returns (table, bytes consumed) like Scanner::Mmap's pointer."""
        h = C.c_void_p()
        used = C.c_size_t()
        image = bytes(image)
        _check(lib().pire_hip_table_mmap(image, len(image), C.byref(h), C.byref(used)))
        return cls(_handle=h), used.value

    @classmethod
    def glue(cls, lhs: "Table", rhs: "Table", max_size: int = 0, gpu: bool = False):
        """Scanner::Glue(lhs, rhs, maxSize) on two ingested tables (reference numbering); gpu=True: BFS on the device."""
        h = C.c_void_p()
        fn = lib().pire_hip_table_glue_gpu if gpu else lib().pire_hip_table_glue
        _check(fn(lhs._h, rhs._h, max_size, C.byref(h)))
        return cls(_handle=h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:      # at interpreter shutdown the module globals may already be gone
            try:
                _lib.pire_hip_table_destroy(h)
            except Exception:
                pass
            self._h = None

    # --- Scanner getters (multi.h:134-161)
    Size = property(lambda s: s.info.states)
    LettersCount = property(lambda s: s.info.letters)
    RegexpsCount = property(lambda s: s.info.regexps)
    Empty = property(lambda s: bool(s.info.empty))
    initial = property(lambda s: s.info.initial)

    def refresh_info(self):
        self._info = None
        return self.info

    def Final(self, idx: int) -> bool:
        return bool(_check(lib().pire_hip_table_final(self._h, idx)))

    def Dead(self, idx: int) -> bool:
        return bool(_check(lib().pire_hip_table_dead(self._h, idx)))

    def AcceptedRegexps(self, idx: int):
        b = u64p()
        n = C.c_size_t()
        _check(lib().pire_hip_table_accepted_regexps(self._h, idx, C.byref(b), C.byref(n)))
        return [int(b[i]) for i in range(n.value)]

    def letter_class(self, ch: int) -> int:
        return _check(lib().pire_hip_table_letter_class(self._h, ch))

    def Next(self, idx: int, ch: int) -> int:
        """Table accessor (Scanner::Next on indices); not a scan loop."""
        return _check(lib().pire_hip_table_next(self._h, idx, ch))

    def layout(self):
        """(orig_of_perm u32[states], hot_rows u8[hot+1, 256]) -- the device numbering and dense LDS rows."""
        o = np.empty(self.info.states, dtype=np.uint32)
        h = np.empty((self.info.hot_states + 1, 256), dtype=np.uint8)
        _check(lib().pire_hip_table_layout(self._h, o.ctypes.data, h.ctypes.data))
        return o, h

    def wide_layout(self):
        """(rows u16[wide + 1, pitch / 2], wide_states, pitch, rows_offset): the class-indexed walk's LDS image (wide.hip);
        rows is None for a table that fits the dense rows."""
        w, pitch, off = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _check(lib().pire_hip_table_wide_layout(self._h, None, 0, C.byref(w), C.byref(pitch), C.byref(off)))
        if not w.value:
            return None, 0, pitch.value, off.value
        rows = np.empty((w.value + 1, pitch.value // 2), dtype=np.uint16)
        _check(lib().pire_hip_table_wide_layout(self._h, rows.ctypes.data, rows.size, C.byref(w), C.byref(pitch), C.byref(off)))
        return rows, w.value, pitch.value, off.value

    def zip_layout(self):
        """None, or the ZIPPED image of the class-indexed walk (pire_hip_table_zip_layout): a dict with tier, full, pitch,
        rows_offset and the decoded pieces rows u16[full + 1, pitch / 2], headers u32[tier + 1], targets u16[tier - full, 3]."""
        g = (C.c_uint32 * 8)()
        _check(lib().pire_hip_table_zip_layout(self._h, None, 0, g))
        tier, full, pitch, rows_off, h_off, x_off, end, k = list(g)
        if not full:
            return None
        img = np.empty((end - rows_off) // 2, dtype=np.uint16)
        _check(lib().pire_hip_table_zip_layout(self._h, img.ctypes.data, img.size, g))
        raw = img.view(np.uint8)
        rows = raw[:(full + 1) * pitch].view(np.uint16).reshape(full + 1, pitch // 2)
        headers = raw[h_off - rows_off:h_off - rows_off + 4 * (tier + 1)].view(np.uint32)
        targets = raw[x_off - rows_off:x_off - rows_off + 2 * k * (tier - full)].view(np.uint16).reshape(tier - full, k)
        return {"tier": tier, "full": full, "pitch": pitch, "rows_offset": rows_off, "rows": rows, "headers": headers,
                "targets": targets}

    def set_config(self, **fields):
        """pire_hip_table_config_set: this table's own configuration = the current one with `fields` changed; no fields: back to the
        process-wide configuration."""
        if not fields:
            _check(lib().pire_hip_table_config_set(self._h, None))
            return None
        c = self.get_config()
        for k, v in fields.items():
            setattr(c, k, v)
        c.size = C.sizeof(Config)
        _check(lib().pire_hip_table_config_set(self._h, C.byref(c)))
        return c

    def get_config(self) -> "Config":
        c = Config()
        c.size = C.sizeof(Config)
        _check(lib().pire_hip_table_config_get(self._h, C.byref(c)))
        return c

    def adapt(self) -> int:
        """Re-rank the LDS rows from the visit counters of earlier scans; returns the number of rows promoted."""
        n = C.c_uint32(0)
        _check(lib().pire_hip_table_adapt(self._h, C.byref(n)))
        self.refresh_info()
        return n.value

    def upload(self):
        _check(lib().pire_hip_table_upload(self._h))

    # --- host-pointer runs (numpy in, numpy out): the PCIe-inclusive convenience mode
    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, init_idx=None, counts=False):
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        cnt = np.zeros(self.RegexpsCount + 2, dtype=np.uint64) if counts else None
        _check(lib().pire_hip_run(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n,
                                  flags & ~FLAG_ON_DEVICE, _np_ptr(init), idx.ctypes.data, fin.ctypes.data,
                                  _np_ptr(cnt), None))
        return (idx, fin, cnt) if counts else (idx, fin)

    def run_strings(self, strings, **kw):
        offs = np.zeros(len(strings) + 1, dtype=np.uint64)
        if strings:
            offs[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
        text = np.frombuffer(b"".join(strings), dtype=np.uint8)
        return self.run(text, offs, **kw)

    def run_strided_host(self, text2d: np.ndarray, flags=FLAG_BEGIN | FLAG_END, init_idx=None, counts=False):
        text2d = np.ascontiguousarray(text2d, dtype=np.uint8)
        n, length = text2d.shape
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        cnt = np.zeros(self.RegexpsCount + 2, dtype=np.uint64) if counts else None
        _check(lib().pire_hip_run_strided(self._h, text2d.ctypes.data if text2d.size else None, n, length, length,
                                          flags & ~FLAG_ON_DEVICE, _np_ptr(init), idx.ctypes.data, fin.ctypes.data,
                                          _np_ptr(cnt), None))
        return (idx, fin, cnt) if counts else (idx, fin)

    # --- device-pointer runs (raw addresses; torch tensors' data_ptr()): only enqueue on `stream`
    def run_strided_device(self, text_ptr: int, n: int, length: int, stride: int, flags, out_idx_ptr=0,
                           out_final_ptr=0, out_counts_ptr=0, init_ptr=0, stream: int = 0):
        _check(lib().pire_hip_run_strided(self._h, text_ptr or None, n, length, stride, flags | FLAG_ON_DEVICE,
                                          init_ptr or None, out_idx_ptr or None, out_final_ptr or None,
                                          out_counts_ptr or None, stream or None))

    def run_device_host_offsets(self, text_ptr: int, offsets: np.ndarray, flags, out_idx_ptr=0, out_final_ptr=0,
                                out_counts_ptr=0, init_ptr=0, stream: int = 0):
        """Resident text (device pointer), offsets on the host (PIRE_HIP_RUN_HOST_OFFSETS); outputs on the device."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(lib().pire_hip_run(self._h, text_ptr or None, offsets.ctypes.data, len(offsets) - 1,
                                  flags | FLAG_ON_DEVICE | FLAG_HOST_OFFSETS, init_ptr or None, out_idx_ptr or None,
                                  out_final_ptr or None, out_counts_ptr or None, stream or None))

    def run_device(self, text_ptr: int, offsets_ptr: int, n: int, flags, out_idx_ptr=0, out_final_ptr=0,
                   out_counts_ptr=0, init_ptr=0, stream: int = 0):
        _check(lib().pire_hip_run(self._h, text_ptr or None, offsets_ptr or None, n, flags | FLAG_ON_DEVICE,
                                  init_ptr or None, out_idx_ptr or None, out_final_ptr or None,
                                  out_counts_ptr or None, stream or None))

    def run_half_final(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        """The table walked as a Pire::HalfFinalScanner: (StateIndex, Final, Result[n, regexps]) for host strings."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        res = np.zeros((n, max(self.RegexpsCount, 1)), dtype=np.uint32)
        _check(lib().pire_hip_run_half_final(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n,
                                             flags & ~FLAG_ON_DEVICE, idx.ctypes.data, fin.ctypes.data,
                                             res.ctypes.data, None))
        return idx, fin, res[:, :self.RegexpsCount]

    def run_half_final_device(self, text_ptr: int, offsets_ptr: int, n: int, flags, out_idx_ptr=0, out_final_ptr=0,
                              out_results_ptr=0, stream: int = 0):
        _check(lib().pire_hip_run_half_final(self._h, text_ptr or None, offsets_ptr or None, n, flags | FLAG_ON_DEVICE,
                                             out_idx_ptr or None, out_final_ptr or None, out_results_ptr or None,
                                             stream or None))

    def run_half_final_device_host_offsets(self, text_ptr: int, offsets: np.ndarray, flags, out_idx_ptr=0,
                                           out_final_ptr=0, out_results_ptr=0, stream: int = 0):
        """Resident text, offsets on the host (PIRE_HIP_RUN_HOST_OFFSETS); outputs on the device."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(lib().pire_hip_run_half_final(self._h, text_ptr or None, offsets.ctypes.data, len(offsets) - 1,
                                             flags | FLAG_ON_DEVICE | FLAG_HOST_OFFSETS, out_idx_ptr or None,
                                             out_final_ptr or None, out_results_ptr or None, stream or None))

    def prefix(self, text, offsets, longest: bool, through_begin=False, through_end=False, generic=False):
        """LongestPrefix / ShortestPrefix lengths (-1 = no prefix) for host strings."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.empty(n, dtype=np.int64)
        _check(lib().pire_hip_prefix(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n,
                                     int(longest), int(through_begin), int(through_end), FLAG_GENERIC if generic else 0,
                                     out.ctypes.data, None))
        return out

    def prefix_device(self, text_ptr: int, offsets_ptr: int, n: int, longest: bool, out_len_ptr: int,
                      through_begin=False, through_end=False, stream: int = 0, generic=False):
        _check(lib().pire_hip_prefix(self._h, text_ptr or None, offsets_ptr or None, n, int(longest), int(through_begin),
                                     int(through_end), FLAG_ON_DEVICE | (FLAG_GENERIC if generic else 0),
                                     out_len_ptr or None, stream or None))

    def suffix(self, text, offsets, longest: bool, through_end=False, through_begin=False):
        """LongestSuffix / ShortestSuffix lengths (-1 = the reference's null) for host strings, walked backwards."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.empty(n, dtype=np.int64)
        _check(lib().pire_hip_suffix(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n,
                                     int(longest), int(through_end), int(through_begin), 0, out.ctypes.data, None))
        return out

    def suffix_device(self, text_ptr: int, offsets_ptr: int, n: int, longest: bool, out_len_ptr: int,
                      through_end=False, through_begin=False, stream: int = 0):
        _check(lib().pire_hip_suffix(self._h, text_ptr or None, offsets_ptr or None, n, int(longest), int(through_end),
                                     int(through_begin), FLAG_ON_DEVICE, out_len_ptr or None, stream or None))

    def check_failures(self) -> int:
        """Failures counted by the checked kernel build (PIRE_HIP_CHECKED=1) since the last call."""
        v = C.c_uint64(0)
        _check(lib().pire_hip_table_check_failures(self._h, C.byref(v)))
        return int(v.value)

    def step_device(self, state_ptr: int, n: int, ch: int, stream: int = 0):
        _check(lib().pire_hip_step(self._h, state_ptr, n, ch, stream or None))


class SlowTable:
    """An ingested Pire::SlowScanner (from SlowScanner::Save() bytes)."""

    def __init__(self, blob: bytes):
        L = lib()
        h = C.c_void_p()
        blob = bytes(blob)
        _check(L.pire_hip_slow_table_create(blob, len(blob), C.byref(h)))
        self._h = h
        self.info = SlowInfo()
        _check(L.pire_hip_slow_table_get_info(h, C.byref(self.info)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            try:
                _lib.pire_hip_slow_table_destroy(h)
            except Exception:
                pass
            self._h = None

    Size = property(lambda s: s.info.states)
    LettersCount = property(lambda s: s.info.letters)
    Empty = property(lambda s: bool(s.info.empty))
    words = property(lambda s: s.info.words)

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END, counts=False):
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        fin = np.empty(n, dtype=np.uint8)
        bits = np.empty((n, self.words), dtype=np.uint32)
        cnt = np.zeros(2, dtype=np.uint64) if counts else None
        _check(lib().pire_hip_slow_run(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n,
                                       flags & ~FLAG_ON_DEVICE, fin.ctypes.data, bits.ctypes.data, _np_ptr(cnt), None))
        return (fin, bits, cnt) if counts else (fin, bits)

    def run_strings(self, strings, **kw):
        offs = np.zeros(len(strings) + 1, dtype=np.uint64)
        if strings:
            offs[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
        return self.run(np.frombuffer(b"".join(strings), dtype=np.uint8), offs, **kw)

    def run_strided_device(self, text_ptr, n, length, stride, flags, out_final_ptr=0, out_bits_ptr=0, out_counts_ptr=0,
                           stream=0):
        _check(lib().pire_hip_slow_run_strided(self._h, text_ptr or None, n, length, stride, flags | FLAG_ON_DEVICE,
                                               out_final_ptr or None, out_bits_ptr or None, out_counts_ptr or None,
                                               stream or None))


class CountingTable:
    """An ingested Pire::CountingScanner / AdvancedCountingScanner (LoadedScanner::Save() bytes)."""

    BASIC, ADVANCED = 0, 1

    def __init__(self, blob: bytes, kind: int):
        L = lib()
        h = C.c_void_p()
        blob = bytes(blob)
        _check(L.pire_hip_counting_table_create(blob, len(blob), C.byref(h)))
        self._h = h
        self.kind = kind
        self.info = CountingInfo()
        _check(L.pire_hip_counting_table_get_info(h, C.byref(self.info)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            try:
                _lib.pire_hip_counting_table_destroy(h)
            except Exception:
                pass
            self._h = None

    Size = property(lambda s: s.info.states)
    LettersCount = property(lambda s: s.info.letters)
    RegexpsCount = property(lambda s: s.info.regexps)
    initial = property(lambda s: s.info.initial)

    def forms(self) -> dict:
        """Which device forms the table has (pire_hip_counting_table_forms)."""
        out = (C.c_uint32 * 8)()
        _check(lib().pire_hip_counting_table_forms(self._h, C.byref(out)))
        return {"packed_nreg": out[0], "packed_lds": out[1], "byte_rows": bool(out[2]), "byte_rows_lds": out[3],
                "letter_rows_nreg": out[4], "letter_rows_lds": out[5], "letter_rows_actions": out[6]}

    def run(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        """(StateIndex[n], Result[n, regexps]) for host strings."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        res = np.zeros((n, max(self.RegexpsCount, 1)), dtype=np.uint32)
        _check(lib().pire_hip_counting_run(self._h, self.kind, text.ctypes.data if text.size else None,
                                           offsets.ctypes.data, n, flags & ~FLAG_ON_DEVICE, idx.ctypes.data,
                                           res.ctypes.data, None))
        return idx, res[:, :self.RegexpsCount]

    def run_strings(self, strings, **kw):
        offs = np.zeros(len(strings) + 1, dtype=np.uint64)
        if strings:
            offs[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
        return self.run(np.frombuffer(b"".join(strings), dtype=np.uint8), offs, **kw)

    def capture(self, text, offsets, flags=FLAG_BEGIN | FLAG_END):
        """The table walked as a Pire::CapturingScanner: (StateIndex, Final, captured, begin, end) for host strings."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        b = np.empty(n, dtype=np.int64)
        e = np.empty(n, dtype=np.int64)
        _check(lib().pire_hip_capture_run(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n,
                                          flags & ~FLAG_ON_DEVICE, idx.ctypes.data, fin.ctypes.data, b.ctypes.data,
                                          e.ctypes.data, None))
        return idx, fin, ((b >= 0) & (e >= 0)).astype(np.uint8), b, e

    def capture_device(self, text_ptr: int, offsets_ptr: int, n: int, flags, out_idx_ptr=0, out_final_ptr=0,
                       out_begin_ptr=0, out_end_ptr=0, stream: int = 0):
        _check(lib().pire_hip_capture_run(self._h, text_ptr or None, offsets_ptr or None, n, flags | FLAG_ON_DEVICE,
                                          out_idx_ptr or None, out_final_ptr or None, out_begin_ptr or None,
                                          out_end_ptr or None, stream or None))

    def run_device(self, text_ptr: int, offsets_ptr: int, n: int, flags, out_idx_ptr=0, out_results_ptr=0, stream: int = 0):
        _check(lib().pire_hip_counting_run(self._h, self.kind, text_ptr or None, offsets_ptr or None, n,
                                           flags | FLAG_ON_DEVICE, out_idx_ptr or None, out_results_ptr or None,
                                           stream or None))


class BatchRunner:
    """Batched twin of Pire::Runner / RunHelper (run.h:365-392):  BatchRunner(t).Begin().Run(text, offs).End()."""

    def __init__(self, table: Table, init_idx=None):
        self.table = table
        self._flags = 0
        self._init = init_idx
        self._text = None
        self._offsets = None
        self._result = None

    def Begin(self):
        self._flags |= FLAG_BEGIN
        return self

    def Run(self, text, offsets):
        self._text, self._offsets = text, offsets
        return self

    def End(self):
        self._flags |= FLAG_END
        return self

    def _go(self):
        if self._result is None:
            if self._text is None:
                raise ValueError("Run() was not called")
            self._result = self.table.run(self._text, self._offsets, flags=self._flags, init_idx=self._init)
        return self._result

    def State(self):
        """StateIndex of every string's end state (run.h:378)."""
        return self._go()[0]

    def Final(self):
        """operator bool of RunHelper, per string (run.h:380)."""
        return self._go()[1].astype(bool)


class Shard(C.Structure):
    """pire_hip_shard (include/pire_hip.h)."""
    _fields_ = [("text", C.c_void_p), ("n", C.c_uint64), ("len", C.c_uint64), ("stride", C.c_uint64),
                ("init_state_idx", C.c_void_p), ("out_state_idx", C.c_void_p), ("out_final", C.c_void_p)]


class ShardOffsets(C.Structure):
    """pire_hip_shard_offsets: one device's part of an offset batch (device pointers of that device)."""
    _fields_ = [("text", C.c_void_p), ("offsets", C.c_void_p), ("n", C.c_uint64), ("init_state_idx", C.c_void_p),
                ("out_state_idx", C.c_void_p), ("out_final", C.c_void_p)]


class MultiRunner:
    """pire_hip_multi: one process, several GPUs, strings sharded by index, match counters reduced over RCCL."""

    def __init__(self, devices=None, ndev: int = 0):
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().pire_hip_multi_create(arr, len(devices), C.byref(h)))
        else:
            _check(lib().pire_hip_multi_create(None, ndev, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.pire_hip_multi_destroy(h)

    @property
    def device_count(self) -> int:
        return lib().pire_hip_multi_device_count(self._h)

    @property
    def reduce_backend(self) -> str:
        return lib().pire_hip_multi_reduce_backend(self._h).decode()

    def run_shards(self, table: "Table", shards, flags=FLAG_BEGIN | FLAG_END, counts=True):
        """shards: one (text_ptr, n, len, stride, init_ptr, out_idx_ptr, out_final_ptr) per device (device pointers)."""
        arr = (Shard * len(shards))()
        for i, (text, n, length, stride, init, oi, of) in enumerate(shards):
            arr[i] = Shard(text or None, n, length, stride, init or None, oi or None, of or None)
        cnt = np.zeros(table.RegexpsCount + 2, dtype=np.uint64) if counts else None
        _check(lib().pire_hip_multi_run_strided(self._h, table._h, arr, flags, _np_ptr(cnt)))
        return cnt

    def run_offset_shards(self, table: "Table", shards, flags=FLAG_BEGIN | FLAG_END, counts=True):
        """shards: one (text_ptr, offsets_ptr, n, init_ptr, out_idx_ptr, out_final_ptr) per device (device pointers)."""
        arr = (ShardOffsets * len(shards))()
        for i, (text, offs, n, init, oi, of) in enumerate(shards):
            arr[i] = ShardOffsets(text or None, offs or None, n, init or None, oi or None, of or None)
        cnt = np.zeros(table.RegexpsCount + 2, dtype=np.uint64) if counts else None
        _check(lib().pire_hip_multi_run(self._h, table._h, arr, flags, _np_ptr(cnt)))
        return cnt

    def run_host(self, table: "Table", text, offsets, flags=FLAG_BEGIN | FLAG_END, init_idx=None):
        """A host batch of ragged strings sharded over the devices by BYTES; (idx, fin, counts) in string order."""
        text = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray))
                                    else text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        cnt = np.zeros(table.RegexpsCount + 2, dtype=np.uint64)
        _check(lib().pire_hip_multi_run_host(self._h, table._h, text.ctypes.data if text.size else None,
                                             offsets.ctypes.data, n, flags, _np_ptr(init), idx.ctypes.data,
                                             fin.ctypes.data, cnt.ctypes.data))
        return idx, fin, cnt

    def last_split(self):
        """First string of every shard of the last host-pointer call, then n."""
        out = (C.c_uint64 * (self.device_count + 1))()
        k = lib().pire_hip_multi_last_split(self._h, out, self.device_count + 1)
        _check(min(k, 0))
        return [int(out[i]) for i in range(k)]

    def run_strided_host(self, table: "Table", text2d: np.ndarray, flags=FLAG_BEGIN | FLAG_END, init_idx=None):
        text2d = np.ascontiguousarray(text2d, dtype=np.uint8)
        n, length = text2d.shape
        idx = np.empty(n, dtype=np.uint32)
        fin = np.empty(n, dtype=np.uint8)
        init = None if init_idx is None else np.ascontiguousarray(init_idx, dtype=np.uint32)
        cnt = np.zeros(table.RegexpsCount + 2, dtype=np.uint64)
        _check(lib().pire_hip_multi_run_strided_host(self._h, table._h, text2d.ctypes.data if text2d.size else None, n,
                                                     length, length, flags, _np_ptr(init), idx.ctypes.data,
                                                     fin.ctypes.data, cnt.ctypes.data))
        return idx, fin, cnt


def run_pair_strided_device(t1: "Table", t2: "Table", text_ptr, n, length, stride, flags, out_idx1_ptr=0, out_idx2_ptr=0,
                            out_final_ptr=0, stream=0):
    """Pire::Run(sc1, sc2, ...) over device-resident fixed-length records: one fused pass (pair.hip)."""
    _check(lib().pire_hip_run_pair_strided(t1._h, t2._h, text_ptr or None, n, length, stride, flags | FLAG_ON_DEVICE,
                                           out_idx1_ptr or None, out_idx2_ptr or None, out_final_ptr or None, stream or None))


def run_pair_device(t1: "Table", t2: "Table", text_ptr, offsets_ptr, n, flags, out_idx1_ptr=0, out_idx2_ptr=0,
                    out_final_ptr=0, stream=0):
    _check(lib().pire_hip_run_pair(t1._h, t2._h, text_ptr or None, offsets_ptr or None, n, flags | FLAG_ON_DEVICE,
                                   out_idx1_ptr or None, out_idx2_ptr or None, out_final_ptr or None, stream or None))


def build_info() -> str:
    return lib().pire_hip_build_info().decode()


def last_kernel() -> str:
    return lib().pire_hip_last_kernel().decode()


def selftested_kernels():
    """The kernel names that have passed a first-use self-test in this process."""
    return [k for k in lib().pire_hip_selftested_kernels().decode().split(",") if k]


def last_kernel_symbol() -> str:
    return lib().pire_hip_last_kernel_symbol().decode()


def set_timing(enabled: bool):
    lib().pire_hip_set_timing(1 if enabled else 0)


def last_kernel_ms() -> float:
    return float(lib().pire_hip_last_kernel_ms())


CORPUS_MAX_PLANTS = 16
CORPUS_PLANT_BYTES = 64


class CorpusPlants(C.Structure):
    """pire_hip_corpus_plants (include/pire_hip.h)."""
    _fields_ = [
        ("nplants", C.c_uint32),
        ("len", C.c_uint32 * CORPUS_MAX_PLANTS),
        ("at_tail", C.c_uint32 * CORPUS_MAX_PLANTS),
        ("bytes", (C.c_uint8 * CORPUS_PLANT_BYTES) * CORPUS_MAX_PLANTS),
    ]


def make_plants(plants) -> CorpusPlants:
    """plants: sequence of (witness_bytes, at_tail_bool)."""
    p = CorpusPlants()
    if len(plants) > CORPUS_MAX_PLANTS:
        raise ValueError("too many plants")
    p.nplants = len(plants)
    for i, (w, tail) in enumerate(plants):
        if len(w) > CORPUS_PLANT_BYTES:
            raise ValueError("plant too long")
        p.len[i] = len(w)
        p.at_tail[i] = 1 if tail else 0
        for k, b in enumerate(w):
            p.bytes[i][k] = b
    return p


def corpus_fill_device(out_ptr: int, seed: int, first: int, count: int, length: int, stride: int, plants=None,
                       stream: int = 0):
    """Generate the synthetic corpus in device memory (plants: a CorpusPlants of this module or of oracle.binding --
    the same layout -- or None)."""
    p = C.byref(plants) if plants is not None else None
    _check(lib().pire_hip_corpus_fill(out_ptr, seed, first, count, length, stride, p, stream or None))
