#!/usr/bin/env python3
"""Latency of small host-pointer calls (the C++ shim's BatchRunner / pigrep use): pire_hip_run with host text."""
import time

import numpy as np

import pire_amd
from oracle import binding as ob
from tests import helpers as H

big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
t = pire_amd.Table(H.load_blob(big["blob"]))
t.upload()
rng = np.random.RandomState(1)
# the text: the benchmark corpus (plain text with planted matches), cut where the strings end -- uniformly random bytes
# would keep the walk in states no text visits, outside the dense rows, and time the cold path instead
CORPUS = ob.corpus_fill(big["corpus"]["seed"], 0, 2048, 4096, H.plants_for(big)).reshape(-1)


def text_of(total):
    return np.ascontiguousarray(CORPUS[:total])


for n, ln in ((10, 100), (1000, 100), (10000, 100), (1000, 4096), (100000, 100)):
    lens = rng.randint(ln // 2, ln + 1, size=n)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    text = text_of(int(offs[-1]))
    t.run(text, offs)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        t.run(text, offs)
        ts.append(time.perf_counter() - t0)
    print("%7d strings x ~%4d B (%8.1f KiB): median %.0f us, min %.0f us per call" % (n, ln, offs[-1] / 1024, 1e6 * np.median(ts), 1e6 * min(ts)))

# the other host-pointer entry points (prefix, half-final): pire_hip_config.host_staging = 0 device blocks cached between
# calls (the default), 1 hipMalloc + hipFree per call (round 2), 2 the stream-ordered pool (hipMallocAsync)
from pire_amd import binding as pb

n, ln = 10, 100
lens = rng.randint(ln // 2, ln + 1, size=n)
offs = np.zeros(n + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
text = text_of(int(offs[-1]))
for mode in (0, 1, 2):
    pb.set_config(host_staging=mode)
    for name, fn in (("prefix", lambda: t.prefix(text, offs, True)), ("half_final", lambda: t.run_half_final(text, offs))):
        fn()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        print("%-10s %d strings x ~%d B: median %.0f us, min %.0f us per call (host_staging=%d)" % (
            name, n, ln, 1e6 * np.median(ts), 1e6 * min(ts), mode))
pb.set_config(host_staging=0)
