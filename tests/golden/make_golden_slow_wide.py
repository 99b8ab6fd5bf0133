#!/usr/bin/env python3
"""Golden fixtures for SlowScanners of MORE than 256 NFA states (the GPU's wave-per-string form), generated with the
unmodified reference (oracle/_ref).  Separate from make_golden.py so that the other fixtures stay byte-identical.

    python tests/golden/make_golden_slow_wide.py      (needs /root/reference: runs in the build container only)

Writes tests/golden/slow_wide.json and the blobs it names.  The reference has no test with such an automaton
(pire_ut.cpp:707-714 uses a.{30}$); these are the same patterns with longer counted gaps, plus the verdicts a reader can
check by eye (x then exactly N characters then the end)."""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.binding import RefSlowScanner  # noqa: E402

A, D = True, False


def write_blob(name, blob):
    rel = os.path.join("blobs", name + ".slow.blob.gz")
    os.makedirs(os.path.join(HERE, "blobs"), exist_ok=True)
    with open(os.path.join(HERE, rel), "wb") as f:
        f.write(gzip.compress(blob, 9, mtime=0))
    return rel


def main():
    cases = []
    zh = "ж".encode("utf-8")
    SLOW = [
        ("slow_x300", "pire_ut.cpp:707-714 with a longer gap", "x.{300}$", "",
         [(b"zzx" + b"y" * 300, A), (b"zzx" + b"y" * 299, D), (b"x" * 301, A), (b"x" + b"y" * 301, D)]),
        ("slow_x400_utf8", "BASELINE config 5b with a longer gap", "x.{400}$", "u",
         [(b"zzx" + b"y" * 400, A), (b"zzx" + b"y" * 399, D), (b"x" + zh * 400, A), (b"x" + zh * 401, D)]),
        ("slow_ab_gap", "several counted gaps", "(ab|cd)+e.{120}f.{150}$", "",
         [(b"abcde" + b"." * 120 + b"f" + b"." * 150, A), (b"abcde" + b"." * 119 + b"f" + b"." * 150, D)]),
    ]
    rng = np.random.RandomState(177)
    for name, source, pat, opt, items in SLOW:
        sc = RefSlowScanner.compile(pat, opt)
        assert sc.size > 256, (name, sc.size)
        strings = [s_ for s_, _ in items]
        strings += [bytes(rng.choice(np.frombuffer(b"ax.yd ef\xd0\xb6bc", dtype=np.uint8), size=int(k)))
                    for k in rng.randint(0, 700, size=30)]
        strings += [b"x" + bytes(rng.choice(np.frombuffer(b"xy", dtype=np.uint8), size=int(k))) for k in (299, 300, 301, 399, 400, 401)]
        fin, bits = sc.run_strings(strings)
        for (s_, verdict), f_ in zip(items, fin):
            assert bool(f_) == verdict, (name, s_[:20], f_, verdict)
        blob = sc.save()
        cases.append({"name": name, "source": source, "pattern": pat, "options": opt,
                      "geometry": {"states": sc.size, "letters": sc.letters, "words": sc.words},
                      "blob": write_blob(name, blob), "blob_sha256": hashlib.sha256(blob).hexdigest(),
                      "strings_hex": [x.hex() for x in strings], "ref_expect": [v for _, v in items],
                      "final": [int(x) for x in fin],
                      "bits_sha256": [hashlib.sha256(bytes(np.ascontiguousarray(b))).hexdigest() for b in bits]})
        print(name, "states", sc.size, "letters", sc.letters, "words", sc.words, "blob", len(blob), "finals", int(sum(fin)))
    with open(os.path.join(HERE, "slow_wide.json"), "w") as f:
        json.dump({"slow_wide": cases}, f, indent=1)


if __name__ == "__main__":
    main()
