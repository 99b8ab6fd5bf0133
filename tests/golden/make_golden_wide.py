#!/usr/bin/env python3
"""Generates tests/golden/wide.json + the dictionary scanner images (round 5): tables whose scans visit thousands of states.

Run where /root/reference exists (needs oracle/_ref, the unmodified reference library):

    python tests/golden/make_golden_wide.py

* dictionary scanners built the way samples/blacklist/blacklist.cpp:65-76 builds one (oracle/ref/ref_capi.cpp
  pire_ref_compile_dictionary): `blacklist_<n>` = the sample's own wrapping (scheme, subdomains, path; anchored),
  `dict_<n>` = the same words Surround()ed (searched anywhere in a record), n = 1 000 and 10 000 made-up domains
  (pire_amd/workloads.py synthetic_domains; there is no network for a real list).  100 000 words do not compile: the
  reference's own determinisation gives up ("regexp pattern too complicated", fsm.cpp:1018).
* `set_b_mix`: the glued 8 952-state table of BASELINE config 5a (tests/golden/set_b.blob.gz) with a corpus of fragments
  that keep several patterns half matched.
For every (set, corpus) the REFERENCE's results on a small sample are recorded (state index + Final of every record / URL),
and how many distinct states the sample's walks visit.
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.binding import OracleScanner, RefScanner  # noqa: E402
from pire_amd import workloads as W  # noqa: E402

SAMPLE_SEED = 0x5EED5EED
SAMPLE_N, SAMPLE_LEN = 128, 1024


def write_gz(name, data):
    with open(os.path.join(HERE, name), "wb") as f:
        f.write(gzip.compress(data, 9, mtime=0))
    return name


def sample_records(entry, corpus, sc, blob):
    rec = W.wide_records(entry, corpus, SAMPLE_SEED, SAMPLE_N, SAMPLE_LEN)
    offs = np.arange(SAMPLE_N + 1, dtype=np.uint64) * SAMPLE_LEN
    idx, fin = sc.run(rec.reshape(-1), offs)
    i2, f2 = sc.run(rec.reshape(-1), offs, kind=RefScanner.NONRELOC)
    assert (idx == i2).all() and (fin == f2).all()
    visits = OracleScanner(blob).visit_counts(rec.reshape(-1), offs)
    order = np.sort(visits)[::-1].astype(np.float64)
    cum = np.cumsum(order) / order.sum()
    return {"seed": SAMPLE_SEED, "n": SAMPLE_N, "len": SAMPLE_LEN, "sha256": hashlib.sha256(rec.tobytes()).hexdigest(),
            "idx": [int(i) for i in idx], "final": [int(f) for f in fin], "distinct_states_visited": int((visits > 0).sum()),
            "share_of_steps_in_top_255_states": round(float(cum[min(254, len(cum) - 1)]), 6),
            "share_of_steps_in_top_1700_states": round(float(cum[min(1699, len(cum) - 1)]), 6)}


def utf8_entries():
    """Large-alphabet dictionary scanners (round 6; BASELINE config 5's wording): mixed-script words (pire_amd/workloads.py
    synthetic_words_utf8), every word parsed by the reference's lexer with Encodings::Utf8() (pire/encoding.cpp:99-111; as
    tests/pire_ut.cpp:181-209 build their UTF-8 scanners), joined with Fsm::operator|= and Surround()ed: > 100 letter classes."""
    words5k = W.synthetic_words_utf8(5000, seed=2)
    words_file = write_gz("dict_words_utf8_5k.txt.gz", b"\n".join(words5k) + b"\n")
    out = []
    for n in (1000, 5000):
        name = "dict_utf8_%dk" % (n // 1000)
        sc = RefScanner.compile_dictionary(words5k[:n], True, utf8=True)
        blob = sc.save()
        entry = {"name": name, "kind": "dictionary", "mode": "surround", "script": "utf8", "words": n, "words_file": words_file,
                 "source": "samples/blacklist/blacklist.cpp:65-76 with every word through Lexer + Encodings::Utf8() (encoding.cpp:99-111)",
                 "geometry": {"states": sc.size, "letters": sc.letters, "regexps": sc.regexps, "initial": sc.initial, "bufsize": sc.bufsize},
                 "blob": write_gz(name + ".blob.gz", blob), "blob_sha256": hashlib.sha256(blob).hexdigest(), "samples": {}}
        for corpus in ("k32", "k512", "k%d" % n):
            entry["samples"][corpus] = sample_records(entry, corpus, sc, blob)
            print(name, corpus, {k: v for k, v in entry["samples"][corpus].items() if k.startswith(("distinct", "share"))}, flush=True)
        out.append(entry)
    return out


def main():
    if "--only-utf8" in sys.argv:   # keep the other entries of wide.json as they are (their determinisation takes minutes)
        with open(os.path.join(HERE, "wide.json")) as f:
            doc = json.load(f)
        doc["wide"] = [w for w in doc["wide"] if w.get("script") != "utf8"] + utf8_entries()
        with open(os.path.join(HERE, "wide.json"), "w") as f:
            json.dump(doc, f, indent=1)
        print("wrote", len(doc["wide"]), "wide sets")
        return
    words10k = W.synthetic_domains(10000, seed=1)
    words_file = write_gz("dict_words_10k.txt.gz", b"\n".join(words10k) + b"\n")
    wide = []
    for n in (1000, 10000):
        words = words10k[:n]
        for mode, surround in (("dict", True), ("blacklist", False)):
            name = "%s_%dk" % (mode, n // 1000)
            sc = RefScanner.compile_dictionary(words, surround)
            blob = sc.save()
            entry = {"name": name, "kind": "dictionary", "mode": "surround" if surround else "blacklist", "words": n,
                     "words_file": words_file, "source": "samples/blacklist/blacklist.cpp:65-76",
                     "geometry": {"states": sc.size, "letters": sc.letters, "regexps": sc.regexps, "initial": sc.initial,
                                  "bufsize": sc.bufsize},
                     "blob": write_gz(name + ".blob.gz", blob), "blob_sha256": hashlib.sha256(blob).hexdigest(), "samples": {}}
            if surround:
                for corpus in ("k32", "k128", "k512", "k2048", "k%d" % n):
                    entry["samples"][corpus] = sample_records(entry, corpus, sc, blob)
                    print(name, corpus, {k: v for k, v in entry["samples"][corpus].items() if k.startswith(("distinct", "share"))}, flush=True)
            else:
                text, offs = W.wide_urls(entry, SAMPLE_SEED, 512)
                idx, fin = sc.run(text, offs)
                visits = OracleScanner(blob).visit_counts(text, offs)
                entry["samples"]["urls"] = {"seed": SAMPLE_SEED, "n": 512, "sha256": hashlib.sha256(text.tobytes()).hexdigest(),
                                            "idx": [int(i) for i in idx], "final": [int(f) for f in fin],
                                            "distinct_states_visited": int((visits > 0).sum())}
                print(name, "urls", int(fin.sum()), "of 512 listed,", int((visits > 0).sum()), "states visited", flush=True)
            wide.append(entry)
    # the glued table of config 5a with a corpus that keeps its patterns half matched
    big = W.pattern_set("set_b")
    blob = W.load_blob(big["blob"])
    sc = RefScanner.load(blob)
    entry = {"name": "set_b_mix", "kind": "glued", "patterns": big["patterns"], "geometry": big["geometry"], "blob": big["blob"],
             "blob_sha256": big["blob_sha256"], "samples": {}}
    entry["samples"]["mix"] = sample_records(entry, "mix", sc, blob)
    print("set_b_mix", {k: v for k, v in entry["samples"]["mix"].items() if k.startswith(("distinct", "share"))}, flush=True)
    wide.append(entry)
    wide += utf8_entries()
    with open(os.path.join(HERE, "wide.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_wide.py", "reference": "yandex/pire v0.0.6 (oracle/_ref)", "wide": wide}, f, indent=1)
    print("wrote", len(wide), "wide sets")


if __name__ == "__main__":
    main()
