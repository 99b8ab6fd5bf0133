"""Shared test helpers: golden fixtures, string packing."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_cache = None


def golden():
    global _cache
    if _cache is None:
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            _cache = json.load(f)
    return _cache


def load_blob(rel: str) -> bytes:
    path = os.path.join(GOLDEN, rel)
    with open(path, "rb") as f:
        data = f.read()
    return gzip.decompress(data) if rel.endswith(".gz") else data


def case_strings(case):
    return [bytes.fromhex(h) for h in case["strings_hex"]]


def all_cases():
    return golden()["cases"]


def big_sets():
    return golden()["big"]


def pack(strings):
    offs = np.zeros(len(strings) + 1, dtype=np.uint64)
    if strings:
        offs[1:] = np.cumsum([len(s) for s in strings], dtype=np.uint64)
    text = np.frombuffer(b"".join(strings), dtype=np.uint8) if strings else np.zeros(0, np.uint8)
    return text, offs


def plants_for(big):
    from oracle.binding import make_plants

    return make_plants([(bytes.fromhex(h), t) for h, t in zip(big["witnesses_hex"], big["witness_at_tail"])])


def random_strings(rng, n, max_len, alphabet=None):
    out = []
    for _ in range(n):
        k = int(rng.randint(0, max_len + 1))
        if alphabet is None:
            out.append(bytes(rng.randint(0, 256, size=k, dtype=np.uint8)))
        else:
            out.append(bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=k)))
    return out
