"""Benchmark workloads: the committed scanner images and corpus plants the benchmarks run on.

The product does not compile regexps (that stays in the reference library on the host), so
a benchmark needs serialised scanners: the fixtures `tests/golden/make_golden.py` generated with the unmodified
reference (Scanner::Save() bytes + the witnesses planted into the synthetic corpus).  This module only reads them;
it needs neither the oracle nor the test helpers."""
import gzip
import json
import os

from .binding import make_plants

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
_cache = None


def _golden():
    global _cache
    if _cache is None:
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            _cache = json.load(f)
    return _cache


def load_blob(rel: str) -> bytes:
    with open(os.path.join(GOLDEN, rel), "rb") as f:
        data = f.read()
    return gzip.decompress(data) if rel.endswith(".gz") else data


def pattern_set(name: str) -> dict:
    """One of the glued 8-regexp sets (set_a = the headline, set_b, set_d, c2_single): its fixture record."""
    for b in _golden()["big"]:
        if b["name"] == name:
            return b
    raise KeyError(name)


def slow_case(name: str) -> dict:
    for c in _golden()["slow"]:
        if c["name"] == name:
            return c
    with open(os.path.join(GOLDEN, "slow_wide.json")) as f:   # SlowScanners of more than 256 NFA states
        for c in json.load(f)["slow_wide"]:
            if c["name"] == name:
                return c
    raise KeyError(name)


def plants_for(big: dict):
    """Corpus plants of a pattern set: witness r at the tail / at a generated offset, as the fixture says."""
    return make_plants([(bytes.fromhex(h), t) for h, t in zip(big["witnesses_hex"], big["witness_at_tail"])])


def ref_bench_file() -> bytes:
    """The reference's benchmark corpus, tools/bench/test_file (20 485 bytes of C++ text), shipped as a data fixture
    (tests/golden/make_ref_corpus.py); its SHA-256 is checked."""
    import hashlib

    data = load_blob("ref_bench_test_file.gz")
    with open(os.path.join(GOLDEN, "ref_bench_test_file.json")) as f:
        meta = json.load(f)
    assert len(data) == meta["bytes"] and hashlib.sha256(data).hexdigest() == meta["sha256"], "corpus fixture damaged"
    return data


def ref_bench_corpus(total_bytes: int) -> bytes:
    """The big file of the reference's tools/bench/run-bench:126-138: test_file doubled (`cat big big > big.new`) until
    it is at least `total_bytes` long -- NOT truncated there, exactly like the script; callers cut what they need."""
    data = ref_bench_file()
    while len(data) < total_bytes:
        data = data + data
    return data
