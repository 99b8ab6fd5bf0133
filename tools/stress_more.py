#!/usr/bin/env python3
"""One-off stress on the GPU box: segmented scan with random knobs, SlowScanner with random patterns, counting scanners
with random regexp/separator pairs -- everything against the oracle."""
import os
import sys

import numpy as np

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

rng = np.random.RandomState(2026)
fails = 0

# ---- segmented scan
tables = []
for name in ("survey_known_answer", "set_a", "set_d", "set_b", "rep_dot_3_10"):
    c = [x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
    tables.append((name, H.load_blob(c["blob"])))
tables.append(("parity", ob.RefScanner.compile(["(b*ab*a)*b*", "(a*ba*b)*a*"], ["n", "n"]).save()))
tables.append(("mod3", ob.RefScanner.compile(["((b|c)*a(b|c)*a(b|c)*a)*(b|c)*"], ["n"]).save()))
ALPHA = b"abc ABCDEFGHIJKLMNOPQRSTUVWXYZ hello wd0123456789-()@net"
runs = 0
for it in range(120):
    name, blob = tables[it % len(tables)]
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    seg = int(rng.choice([32, 48, 64, 100, 128, 256, 1000, 4096]))
    knobs = dict(segment_bytes=seg, segment_warmup=int(rng.choice([0, 4, 16, 64, 256])) or pb.NONE,
                 segment_modes=int(rng.randint(1, 7)), segment_budget=int(rng.choice([0, 1, 3, 32])) or pb.NONE)
    pb.set_config(**knobs)
    a = np.frombuffer(b"ab" if name == "parity" else b"abc" if name == "mod3" else ALPHA, dtype=np.uint8)
    strings = [a[rng.randint(0, len(a), size=int(k))].tobytes() for k in rng.randint(0, 6000, size=int(rng.randint(1, 30)))]
    text, offs = H.pack(strings)
    flags = int(rng.choice([0, 1, 2, 3]))
    oi, of = o.run(*ob.pack_strings(strings), flags=flags)
    gi, gf, cnt = t.run(text, offs, flags=flags, counts=True)
    runs += 1
    if not ((gi == oi).all() and (gf == of).all() and cnt[0] == int(of.sum()) and cnt[1] == len(strings)):
        print("SEGMENTED FAIL", name, knobs, flags)
        fails += 1
pb.set_config(segment_bytes=0, segment_warmup=0, segment_modes=0, segment_budget=0)
print("segmented runs:", runs, "fails:", fails)

# ---- SlowScanner, random patterns
ATOMS = ["a", "b", "ab", "[ab]", ".", "x", "(ab|ba)", "a*", "b+", ".{3}", ".{1,4}", "c?", "(a|ab|abc)"]
sruns = 0
for it in range(60):
    pat = "".join(ATOMS[rng.randint(0, len(ATOMS))] for _ in range(rng.randint(2, 7)))
    if rng.randint(0, 2):
        pat += "$"
    try:
        r = ob.RefSlowScanner.compile(pat, "")
    except Exception:
        continue
    blob = r.save()
    try:
        t = pire_amd.SlowTable(blob)
    except pire_amd.PireHipError:
        continue
    o = ob.OracleSlowScanner(blob)
    strings = H.random_strings(rng, 800, 150, b"abcx") + [b"", b"ab" * 200]
    for flags in (3, 0):
        of, obits = o.run_strings(strings, flags=flags)
        gf, gb = t.run_strings(strings, flags=flags)
        sruns += 1
        if not ((gf == of).all() and (gb == obits).all()):
            print("SLOW FAIL", pat, flags)
            fails += 1
print("slow runs:", sruns, "fails:", fails)

# ---- counting scanners, random pairs
RES = ["a", "b", "ab", "[ab]+", "c", "bc", "d", "abc", "ca", "a+b", "[a-c]"]
SEPS = [".*", "\\s", "c", "[ ,]", "d"]
cruns = 0
for it in range(60):
    k = int(rng.randint(1, 7))
    res_ = [RES[rng.randint(0, len(RES))] for _ in range(k)]
    seps = [SEPS[rng.randint(0, len(SEPS))] for _ in range(k)]
    kind = int(rng.randint(0, 3))
    try:
        blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
    except Exception:
        continue
    t, o = pire_amd.CountingTable(blob, kind), ob.OracleCountingScanner(blob, kind)
    strings = H.random_strings(rng, 1500, 200, b"abcd ,\n") + [b""]
    for flags in (3, 0):
        oi, orr = o.run_strings(strings, flags=flags)
        gi, gr = t.run_strings(strings, flags=flags)
        cruns += 1
        if not ((gi == oi).all() and (gr == orr).all()):
            print("COUNTING FAIL", res_, seps, kind, flags)
            fails += 1
print("counting runs:", cruns, "fails:", fails)
sys.exit(1 if fails else 0)
