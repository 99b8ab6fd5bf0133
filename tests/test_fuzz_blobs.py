"""Malformed table images must be refused, never crash: the reference's Load()/Mmap() validate the header and the
sizes of a Save() image (multi.h:244-279, scanner_io.cpp:51-69, 113-170) and raise Pire::Error; the ingestion behind
pire_hip_table_create / _slow_table_create / _counting_table_create reads untrusted bytes (files, mmaps) and has to
hold the same line.  Deterministic mutations (truncations, byte flips, dword overwrites) of golden blobs of every
format run in a CHILD process so that a crash shows up as a test failure, not as a dead test runner."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, random, sys
sys.path.insert(0, %(root)r)
from pire_amd import binding as pb
from tests import helpers as H

g = H.golden()
jobs = []
for c in g["cases"][:6] + g["half_final"][:2]:
    jobs.append(("scanner", c["blob"], lambda b: pb.Table(b)))
for c in g["simple"][:3]:
    jobs.append(("simple", c["blob"], lambda b: pb.Table(b)))
for c in g["slow"]:
    jobs.append(("slow", c["blob"], lambda b: pb.SlowTable(b)))
kinds = {"basic": 0, "advanced": 1, "noglue": 2}
for c in g["counting"][:8]:
    k = kinds.get(c["kind"], 0)
    jobs.append(("counting", c["blob"], lambda b, k=k: pb.CountingTable(b, k)))
for c in g["capturing"][:2]:
    jobs.append(("capturing", c["blob"], lambda b: pb.CountingTable(b, 0)))

rng = random.Random(20260925)
stats = {"ok": 0, "refused": 0}
for name, rel, make in jobs:
    blob = H.load_blob(rel)
    make(blob)                                    # the untouched image loads
    muts = []
    for cut in sorted({0, 1, 7, 8, 23, 24, 31, 32, 40, 47, 48, len(blob) // 2, len(blob) - 8, len(blob) - 1}):
        if 0 <= cut < len(blob):
            muts.append(blob[:cut])
    for _ in range(%(rounds)d):
        b = bytearray(blob)
        kind = rng.randrange(4)
        if kind == 0:                             # one byte anywhere
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == 1:                           # a dword in the header / size area
            pos = rng.randrange(0, min(len(b) - 4, 160)) & ~3
            b[pos:pos + 4] = rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 65536, rng.getrandbits(32)]).to_bytes(4, "little")
        elif kind == 2:                           # a dword anywhere (transitions, tags, action lists)
            pos = rng.randrange(0, len(b) - 4)
            b[pos:pos + 4] = rng.choice([0xFFFFFFFF, 0x7FFFFFF0, 0x80000000, rng.getrandbits(32)]).to_bytes(4, "little")
        else:                                     # a burst of random bytes
            pos = rng.randrange(len(b))
            for i in range(pos, min(len(b), pos + rng.randrange(1, 64))):
                b[i] = rng.getrandbits(8)
        muts.append(bytes(b))
    for m in muts:
        try:
            t = make(m)
            stats["ok"] += 1                      # a mutation that keeps the image well-formed
            del t
        except pb.PireHipError as e:
            assert e.code < 0
            stats["refused"] += 1
print(json.dumps(stats))
"""


@pytest.mark.timeout(900)
def test_mutated_images_are_refused_or_loaded_never_crash():
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "rounds": 150}], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, f"ingestion crashed or raised (rc={r.returncode}):\n{r.stderr[-3000:]}"
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    assert stats["refused"] > 500 and stats["ok"] > 0, stats


CHILD_GPU = r"""
import json, random, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pire_amd import binding as pb
from oracle import binding as ob
from tests import helpers as H

g = H.golden()
rng = random.Random(777)
nrng = np.random.RandomState(3)
strings = H.random_strings(nrng, 300, 90, b"abcdehlorw xyzHIAZ019-() \xd0\xb6") + [b"", b"hello world"]
stats = {"scanned": 0, "compared": 0, "refused": 0}


def mutate(blob):
    b = bytearray(blob)
    for _ in range(rng.randrange(1, 4)):
        kind = rng.randrange(3)
        if kind == 0:
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            pos = rng.randrange(0, len(b) - 4)
            b[pos:pos + 4] = rng.choice([0, 1, 2, 3, 7, 8, 64, 0xFFFFFFFF, rng.getrandbits(16)]).to_bytes(4, "little")
        else:
            pos = rng.randrange(48, len(b) - 8) & ~7
            b[pos:pos + 8] = rng.choice([0, 1, 5, 9, 2 ** 64 - 1, rng.getrandbits(10)]).to_bytes(8, "little")
    return bytes(b)


for c in g["cases"][:10] + g["half_final"][:3] + [x for x in g["cases"] if "glue" in x["name"].lower()][:3]:
    blob = H.load_blob(c["blob"])
    for _ in range(%(rounds)d):
        m = mutate(blob)
        try:
            t = pb.Table(m)
        except pb.PireHipError:
            stats["refused"] += 1
            continue
        idx, fin, cnt = t.run_strings(strings, counts=True)          # must not fault, whatever the image says
        text, offs = H.pack(strings)
        t.run_half_final(text, offs)                                   # the path that indexes per-regexp results
        t.run(text, offs, flags=pb.FLAG_BEGIN | pb.FLAG_END | pb.FLAG_GENERIC, counts=True)
        stats["scanned"] += 1
        assert int(cnt[1]) == len(strings)
        try:
            o = ob.OracleScanner(m)
        except Exception:
            continue
        oi, of = o.run_strings(strings)
        assert (idx == oi).all() and (fin == of).all(), c["name"]
        stats["compared"] += 1
print(json.dumps(stats))
"""


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_mutated_images_that_load_also_scan_without_faulting():
    """an image that passes ingestion must be safe to RUN -- regexp ids index counters, state ids index
    rows -- so every mutation that loads is scanned (with counters); where the oracle accepts the same bytes the
    results must agree."""
    r = subprocess.run([sys.executable, "-c", CHILD_GPU % {"root": ROOT, "rounds": 40}], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, f"scan of a mutated image crashed (rc={r.returncode}):\n{r.stderr[-3000:]}"
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    assert stats["scanned"] > 50 and stats["compared"] > 20, stats
