#!/usr/bin/env python3
"""Generate tests/golden/* from the UNMODIFIED reference library (oracle/_ref/libpire_ref.so).

Run in the dev container (where /root/reference exists):   python tests/golden/make_golden.py
The fixtures travel with the repo; the reference does not exist on the GPU box.

Two kinds of knowledge are recorded per case:
  * ``ref_expect``  -- the accept/deny verdict WRITTEN IN the reference's own unit tests
                       (/root/reference/tests/pire_ut.cpp, line cited per case): the known answers;
  * ``idx`` / ``final`` / ``accepted`` -- what the compiled reference itself returned for
    Runner(sc).Begin().Run(s).End() on each string (StateIndex, Final, AcceptedRegexps): these pin
    bit-exact state ids, which the reference's tests do not.
The generator asserts that the two agree (ACCEPTS <=> AcceptedRegexps non-empty, tests/common.h:171-177).
"""
import base64
import gzip
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.binding import RefScanner, RefSlowScanner, corpus_fill, make_plants  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

A, D = True, False   # ACCEPTS / DENIES

# (name, source, patterns, options, [(string, verdict-or-None)])
CASES = [
    ("string", "pire_ut.cpp:38-45", ["abc"], [""], [(b"def abc ghi", A), (b"abc", A), (b"def abd ghi", D)]),
    ("boundaries_begin", "pire_ut.cpp:49-52", ["^abc"], [""], [(b"abc ghi", A), (b"def abc", D)]),
    ("boundaries_end", "pire_ut.cpp:54-57", ["abc$"], [""], [(b"abc ghi", D), (b"def abc", A)]),
    ("prim_alt", "pire_ut.cpp:62-66", ["abc|def"], [""], [(b"def", A), (b"abc", A), (b"deb", D)]),
    ("prim_star", "pire_ut.cpp:68-74", ["ad*e"], [""],
     [(b"xaez", A), (b"xadez", A), (b"xaddez", A), (b"xadddddddddddddddddddddddez", A), (b"xafez", D)]),
    ("prim_plus", "pire_ut.cpp:76-82", ["ad+e"], [""],
     [(b"xaez", D), (b"xadez", A), (b"xaddez", A), (b"xadddddddddddddddddddddddez", A), (b"xafez", D)]),
    ("prim_opt", "pire_ut.cpp:84-89", ["ad?e"], [""], [(b"xaez", A), (b"xadez", A), (b"xaddez", D), (b"xafez", D)]),
    ("prim_count", "pire_ut.cpp:91-95", ["a.{1}e"], [""], [(b"axe", A), (b"ae", D), (b"axye", D)]),
] + [
    ("mass_alt_%d" % i, "pire_ut.cpp:98-118", [p], [""],
     [(b"abc", A), (b"def", A), (b"ghi", A), (b"klm", A), (b"aei", D), (b"klc", D)])
    for i, p in enumerate(["((abc|def)|ghi)|klm", "(abc|def)|(ghi|klm)", "abc|(def|(ghi|klm))", "abc|(def|ghi)|klm"])
] + [
    ("composition_slashes", "pire_ut.cpp:122-132", ["^/([^\\\\/]|\\\\.)*/[a-z]*$"], [""],
     [(b"/regexp/i", A), (b"/regexp2/", A), (b"regexp", D), (b"/dir\\/file/", A), (b"/dir/file/", D),
      (b"/dir\\\\/", A), (b"/dir\\\\/file/", D)]),
    ("composition_head_tail", "pire_ut.cpp:134-139", ["Head(Inner)*Tail"], [""],
     [(b"HeadInnerTail", A), (b"HeadInnerInnerTail", A), (b"HeadInneInnerTail", D), (b"HeadTail", A)]),
    ("rep_3_6", "pire_ut.cpp:144-151", ["^x{3,6}$"], [""],
     [(b"xx", D), (b"xxx", A), (b"xxxx", A), (b"xxxxx", A), (b"xxxxxx", A), (b"xxxxxxx", D)]),
    ("rep_3_inf", "pire_ut.cpp:153-159", ["^x{3,}$"], [""],
     [(b"xx", D), (b"xxx", A), (b"xxxx", A), (b"x" * 11, A), (b"x" * 47, A)]),
    ("rep_3", "pire_ut.cpp:161-168", ["^x{3}$"], [""],
     [(b"x", D), (b"xx", D), (b"xxx", A), (b"xxxx", D), (b"xxxxx", D), (b"x" * 47, D)]),
    ("rep_dot_3_10", "pire_ut.cpp:170-178", ["x.{3,10}$"], [""],
     [(b"b" * (2 * n) + b"x" + b"e" * n, A if 3 <= n <= 10 else D) for n in range(20)]),
    ("utf8_dot", "pire_ut.cpp:183-206", ["^.$"], ["u"],
     [(b"\x41", A), (b"\x81", D), (b"\xC1\x81", A), (b"\xC1", D), (b"\xC1\x41", D), (b"\xC1\xC2", D),
      (b"\xC1\x81\x82", D), (b"\xE1\x81\x82", A), (b"\xE1", D), (b"\xE1\x42", D), (b"\xE1\x42\x43", D),
      (b"\xE1\xC2\xC3", D), (b"\xE1\x82", D), (b"\xE1\x82\x83\x84", D), (b"\xF1\x81\x82\x83", A)]),
    ("utf8_literal", "pire_ut.cpp:208", ["x\xD0\xA4y"], ["u"], [(b"x\xD0\xA4y", A)]),
    ("andnot_1", "pire_ut.cpp:213-219", ["<([0-9]+&~123&~456)>"], ["a"],
     [(b"<111>", A), (b"<124>", A), (b"<123>", D), (b"<456>", D), (b"<abc>", D)]),
    ("andnot_2", "pire_ut.cpp:221-224", ["[0-9]+\\&1+"], ["a"], [(b"111", D), (b"123&111", A)]),
    ("misc_1", "pire_ut.cpp:240", ["^[^\\s=/>]*$"], ["n"], [(b"a", A)]),
    ("misc_tab", "pire_ut.cpp:241", ["\\t"], [""], [(b"\t", A)]),
    ("ranges", "pire_ut.cpp:253-256", ["a\\W"], [""], [(b"a,", A), (b"ab", D)]),
    ("shortcuts_aaa", "pire_ut.cpp:630-634", ["aaa"], [""],
     [(b"." * 38 + b"aaa" + b"." * 13, A), (b"." * 38 + b"aab" + b"." * 13, D), (b"." * 54, D)]),
    ("shortcuts_ab3", "pire_ut.cpp:635-640", ["[ab]{3}"], [""],
     [(b"." * 38 + b"aaa" + b"." * 13, A), (b"." * 38 + b"aab" + b"." * 13, A), (b"." * 38 + b"bbb" + b"." * 13, A),
      (b"." * 54, D)]),
    ("shortcuts_utf8", "pire_ut.cpp:641-645", ["\xD0\xB0"], ["u"],
     [(b"." * 38 + b"\xD0\xB0" + b"." * 15, A), (b"." * 35 + b"\xD0\xB0" + b"." * 18, A),
      (b"." * 32 + b"\xD0\xB0" + b"." * 21, A)]),
    ("aligned_xy", "pire_ut.cpp:733-743", ["xy"], [""],
     [(b"xy", A), (b"yz", D), (b"......xy", A), (b"......yz", D)]),
    ("aligned_abcde", "pire_ut.cpp:745-755", ["abcde"], [""],
     [(b"ZZZZZabcdeZZZZZZ", A), (b"ZZZZZabcdfZZZZZZ", D), (b"ZabcdeZZZ", A), (b"ZxbcdeZZZ", D),
      (b"ZZZZZZZZZZZZZabcde", A), (b"ZZZZZZZZZZZZZabcdf", D)]),
    ("serialization_pattern", "pire_ut.cpp:534-538, 555", ["^regexp$"], [""],
     [(b"regexp", A), (b"regxp", D), (b"regexp t", D)]),
    ("copying_pattern", "pire_ut.cpp:503-517", ["^r$"], [""], [(b"r", A), (b"p", D)]),
    ("empty_scanner", "pire_ut.cpp:760-830", [], [], [(b"a strin", D), (b"", D)]),
    ("null_fsm", "pire_ut.cpp:832-837 (Fsm() is not exposed; the empty pattern, unsurrounded, is the same automaton)",
     [""], ["n"], [(b"", None)]),
    ("survey_known_answer", "SURVEY.md 8c / README:60", ["hello\\s+w.+d$"], [""],
     [(b"hello world", A), (b"Hello world", D), (b"say hello   wod", A), (b"hello world!", D), (b"hello wd", D),
      (b"", D), (b"xxhello\tw--d", A)]),
]

# Glue (pire_ut.cpp:648-705): expected AcceptedRegexps lists are written in the test itself.
GLUE_CASES = [
    ("glue_aaa_bbb", "pire_ut.cpp:651-674", ["aaa", "bbb"], ["", ""],
     [(b"aaa", [0]), (b"bbb", [1]), (b"aaabbb", [0, 1]), (b"ccc", [])]),
    # the test glues (ccc, (aaa,bbb)); a left fold gives the same regexp numbering ccc=0, aaa=1, bbb=2
    ("glue_ccc_aaa_bbb", "pire_ut.cpp:676-683", ["ccc", "aaa", "bbb"], ["", "", ""],
     [(b"ccc", [0]), (b"aaa", [1]), (b"aaabbb", [1, 2]), (b"xyz", [])]),
    ("glue_nonfinal", "pire_ut.cpp:684-692", ["a", "c"], ["n", "n"], [(b"ac", [])]),
    ("inline_glue3", "inline_ut.cpp:60-91 (patterns)", ["foo", "bar", "http://([a-z0-9]+\\.)+[a-z]{2,4}/?"], ["", "", ""],
     [(b"foo", [0]), (b"bar", [1]), (b"foobar", [0, 1]), (b"see http://aba.caba.ru/ ok", [2]), (b"none", [])]),
]

# The 8-regexp sets of SURVEY.md section 8d (patterns from tools/bench/run-bench:42-80, README:60, pire_ut.cpp).
SET_A = ["hello\\s+w.+d$", "ABCDEFGHIJKLMNOPQRSTUVWXYZ$", "[XYZ]ABCDEFGHIJKLMNOPQRSTUVWXYZ$",
         "[ -~]*ABCDEFGHIJKLMNOPQRSTUVWXYZ$", "(\\d{3}-|\\(\\d{3}\\)\\s+)(\\d{3}-\\d{4})$", "[@QZ]$", "[ABC]$", "[net]$"]
SET_A_WITNESS = [(b"hello  world", True), (b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", True),
                 (b"XABCDEFGHIJKLMNOPQRSTUVWXYZ", True), (b"abc ABCDEFGHIJKLMNOPQRSTUVWXYZ", True),
                 (b"(123)  456-7890", True), (b"Q", True), (b"B", True), (b"t", True)]
SET_D = ["hello\\s+w.+d$", "abc", "abc|def", "ad*e", "ad+e", "Head(Inner)*Tail", "^x{3,6}$", "aaa"]
SET_D_WITNESS = [(b"hello  world", True), (b"abc", False), (b"def", False), (b"addde", False), (b"ade", False),
                 (b"HeadInnerInnerTail", False), (b"xxxx", False), (b"aaa", False)]
# BASELINE config 5a: a glued table far too big for LDS (8 952 states x 60 letters, 3.2 MB reference buffer; SURVEY 8d "Set B")
SET_B = ["hello\\s+w.+d$", "ABCDEFGHIJKLMNOPQRSTUVWXYZ$", "[XYZ]ABCDEFGHIJKLMNOPQRSTUVWXYZ$",
         "[ -~]*ABCDEFGHIJKLMNOPQRSTUVWXYZ$", "(\\d{3}-|\\(\\d{3}\\)\\s+)(\\d{3}-\\d{4})$",
         "http://([a-z0-9]+\\.)+[a-z]{2,4}/?", "^[a-z]+@[a-z]+\\.com", "foo(bar|baz)+qux"]
SET_B_WITNESS = [(b"hello  world", True), (b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", True),
                 (b"XABCDEFGHIJKLMNOPQRSTUVWXYZ", True), (b"abc ABCDEFGHIJKLMNOPQRSTUVWXYZ", True),
                 (b"(123)  456-7890", True), (b" http://aba.caba.ru/ ", False), (b"foobarbazbarqux", False),
                 (b"http://x.yz/foobazqux", False)]
CORPUS_SEED = 0x5EED5EED


def write_blob(name, blob):
    if len(blob) > 65536:
        path = name + ".blob.gz"
        with open(os.path.join(OUT, path), "wb") as f:
            f.write(gzip.compress(blob, 9, mtime=0))
    else:
        path = name + ".blob"
        with open(os.path.join(OUT, path), "wb") as f:
            f.write(blob)
    return path


def record(sc, strings):
    idx, fin = sc.run_strings(strings)
    idx_n, fin_n = sc.run_strings(strings, kind=RefScanner.NONRELOC)
    assert (idx == idx_n).all() and (fin == fin_n).all(), "Scanner vs NonrelocScanner disagree"
    acc = [sc.accepted(int(i)) for i in idx]
    return [int(i) for i in idx], [int(f) for f in fin], acc


def geometry(sc):
    return {"states": sc.size, "letters": sc.letters, "regexps": sc.regexps, "initial": sc.initial,
            "bufsize": sc.bufsize, "empty": sc.empty}


def main():
    cases = []
    for name, source, pats, opts, items in CASES:
        sc = RefScanner.compile(pats, opts)
        strings = [s for s, _ in items]
        idx, fin, acc = record(sc, strings)
        for (s, verdict), a in zip(items, acc):
            if verdict is not None:
                assert (len(a) > 0) == verdict, (name, s, a, verdict)
        blob = sc.save()
        cases.append({"name": name, "source": source, "patterns": pats, "options": opts, "geometry": geometry(sc),
                      "blob": write_blob(name, blob), "blob_sha256": hashlib.sha256(blob).hexdigest(),
                      "strings_hex": [s.hex() for s in strings],
                      "ref_expect": [v for _, v in items], "idx": idx, "final": fin, "accepted": acc})
    for name, source, pats, opts, items in GLUE_CASES:
        sc = RefScanner.compile(pats, opts)
        strings = [s for s, _ in items]
        idx, fin, acc = record(sc, strings)
        for (s, want), a in zip(items, acc):
            assert a == want, (name, s, a, want)
        blob = sc.save()
        cases.append({"name": name, "source": source, "patterns": pats, "options": opts, "geometry": geometry(sc),
                      "blob": write_blob(name, blob), "blob_sha256": hashlib.sha256(blob).hexdigest(),
                      "strings_hex": [s.hex() for s in strings],
                      "ref_expect_accepted": [w for _, w in items], "idx": idx, "final": fin, "accepted": acc})

    # big glued sets: geometry + results on seeded corpus strings and on raw random bytes (all 256 values)
    big = []
    for name, pats, wit in (("set_a", SET_A, SET_A_WITNESS), ("set_d", SET_D, SET_D_WITNESS),
                            ("set_b", SET_B, SET_B_WITNESS)):
        sc = RefScanner.compile(pats, [""] * len(pats))
        blob = sc.save()
        plants = make_plants(wit)
        n, length = 96, 1024
        data = corpus_fill(CORPUS_SEED, 0, n, length, plants)
        offs = np.arange(n + 1, dtype=np.uint64) * length
        idx, fin = sc.run(data.reshape(-1), offs)
        rng = np.random.RandomState(1234)
        lens = rng.randint(0, 300, size=64)
        raw = [bytes(rng.randint(0, 256, size=int(k), dtype=np.uint8)) for k in lens]
        ridx, rfin, racc = record(sc, raw)
        big.append({"name": name, "patterns": pats, "witnesses_hex": [w.hex() for w, _ in wit],
                    "witness_at_tail": [bool(t) for _, t in wit], "geometry": geometry(sc),
                    "blob": write_blob(name, blob), "blob_sha256": hashlib.sha256(blob).hexdigest(),
                    "corpus": {"seed": CORPUS_SEED, "n": n, "len": length,
                               "sha256": hashlib.sha256(data.tobytes()).hexdigest(),
                               "idx": [int(i) for i in idx], "final": [int(f) for f in fin],
                               "accepted": [sc.accepted(int(i)) for i in idx]},
                    "raw": {"numpy_randomstate_seed": 1234, "strings_hex": [r.hex() for r in raw], "idx": ridx,
                            "final": rfin, "accepted": racc}})

    # BASELINE config C2: the single pattern, with a planted witness so that full-size runs are not vacuous
    sc = RefScanner.compile(["hello\\s+w.+d$"], [""])
    blob = sc.save()
    wit = [(b"hello  world", True), (b"hello w-d", True), (b"hello\tworld", False)]
    plants2 = make_plants(wit)
    n, length = 96, 1024
    data = corpus_fill(CORPUS_SEED, 0, n, length, plants2)
    idx, fin = sc.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length)
    big.append({"name": "c2_single", "patterns": ["hello\\s+w.+d$"], "witnesses_hex": [w.hex() for w, _ in wit],
                "witness_at_tail": [bool(t) for _, t in wit], "geometry": geometry(sc),
                "blob": write_blob("c2_single", blob), "blob_sha256": hashlib.sha256(blob).hexdigest(),
                "corpus": {"seed": CORPUS_SEED, "n": n, "len": length,
                           "sha256": hashlib.sha256(data.tobytes()).hexdigest(),
                           "idx": [int(i) for i in idx], "final": [int(f) for f in fin],
                           "accepted": [sc.accepted(int(i)) for i in idx]},
                "raw": {"numpy_randomstate_seed": 1234, "strings_hex": [], "idx": [], "final": [], "accepted": []}})

    # corpus generator pin: bytes of a few strings, with and without plants
    plants = make_plants(SET_A_WITNESS)
    c0 = corpus_fill(CORPUS_SEED, 0, 4, 100, None)
    c1 = corpus_fill(CORPUS_SEED, 5, 12, 77, plants)
    corpus = {"seed": CORPUS_SEED, "noplant_first4_len100_b64": base64.b64encode(c0.tobytes()).decode(),
              "planted_from5_count12_len77_b64": base64.b64encode(c1.tobytes()).decode()}

    # SlowScanner (scanners/slow.h): the Slow test of pire_ut.cpp:707-714 + the config-5b style patterns
    slow = []
    SLOW = [
        ("slow_a30", "pire_ut.cpp:707-714", "a.{30}$", "",
         [(b"....a" + b"." * 30, A), (b"....a" + b"." * 31, D), (b"....a" + b"." * 29, D)]),
        ("slow_x40_utf8", "BASELINE config 5 / SURVEY 8a a12", "x.{40}$", "u",
         [(b"zzx" + b"y" * 40, A), (b"zzx" + b"y" * 39, D), (b"x" + "\u0436".encode("utf-8") * 40, A),
          (b"x" + "\u0436".encode("utf-8") * 41, D)]),
        ("slow_alt", "pire_ut.cpp:62-74 patterns through SlowScanner", "abc|ad*e", "",
         [(b"def", D), (b"abc", A), (b"xadddez", A), (b"xafez", D)]),
    ]
    rng = np.random.RandomState(77)
    for name, source, pat, opt, items in SLOW:
        sc = RefSlowScanner.compile(pat, opt)
        strings = [s_ for s_, _ in items]
        strings += [bytes(rng.choice(np.frombuffer(b"ax.yd e\xd0\xb6bc", dtype=np.uint8), size=int(k)))
                    for k in rng.randint(0, 120, size=40)]
        fin, bits = sc.run_strings(strings)
        for (s_, verdict), f_ in zip(items, fin):
            assert bool(f_) == verdict, (name, s_, f_, verdict)
        blob = sc.save()
        slow.append({"name": name, "source": source, "pattern": pat, "options": opt,
                     "geometry": {"states": sc.size, "letters": sc.letters, "words": sc.words},
                     "blob": write_blob(name, blob), "blob_sha256": hashlib.sha256(blob).hexdigest(),
                     "strings_hex": [x.hex() for x in strings], "ref_expect": [v for _, v in items],
                     "final": [int(x) for x in fin], "bits_hex": [bytes(np.ascontiguousarray(b)).hex() for b in bits]})

    # SimpleScanner (scanners/simple.h): tests/common.h:80-119 runs every SCANNER() block through it as well, so the
    # verdicts of tests/pire_ut.cpp apply unchanged
    from oracle.binding import RefSimpleScanner
    simple = []
    SIMPLE = [
        ("simple_hello", "README:60 / SURVEY 8c known answer", r"hello\s+w.+d$", "",
         [(b"hello world", A), (b"Hello world", D), (b"say hello   wod", A), (b"hello world!", D), (b"hello wd", D),
          (b"", D), (b"xxhello\tw--d", A)]),
        ("simple_alt", "pire_ut.cpp:62-74", "abc|def", "", [(b"abc", A), (b"def", A), (b"xxabcyy", A), (b"ab", D), (b"", D)]),
        ("simple_count", "pire_ut.cpp:117-131 (Repetition)", "^x{3,6}$", "", [(b"xxx", A), (b"xxxxxx", A), (b"xx", D),
                                                                               (b"xxxxxxx", D), (b"axxx", D)]),
        ("simple_head_tail", "pire_ut.cpp:76-91 (Composition)", "Head(Inner)*Tail", "",
         [(b"HeadTail", A), (b"HeadInnerInnerTail", A), (b"HeadInneTail", D)]),
        ("simple_utf8_dot", "pire_ut.cpp:181-209 (UTF8)", "^.$", "u", [("\u0436".encode("utf-8"), A), (b"a", A), (b"ab", D),
                                                                        (b"\xd0", D)]),
    ]
    rng = np.random.RandomState(78)
    for name, source, pat, opt, items in SIMPLE:
        sc = RefSimpleScanner.compile(pat, opt)
        strings = [s_ for s_, _ in items]
        strings += [bytes(rng.choice(np.frombuffer(b"abcdefxHeadInrTl w\t\xd0\xb6", dtype=np.uint8), size=int(k)))
                    for k in rng.randint(0, 90, size=40)]
        bi, bf = sc.run_strings(strings)
        ni, nf = sc.run_strings(strings, flags=0)
        for (s_, verdict), f_ in zip(items, bf):
            assert bool(f_) == verdict, (name, s_, f_, verdict)
        blob = sc.save()
        simple.append({"name": name, "source": source, "pattern": pat, "options": opt, "states": sc.size,
                       "initial": sc.initial, "empty": sc.empty, "blob": write_blob(name, blob),
                       "blob_sha256": hashlib.sha256(blob).hexdigest(), "strings_hex": [x.hex() for x in strings],
                       "accepts_hex": [x.hex() for x, v in items if v], "denies_hex": [x.hex() for x, v in items if not v],
                       "be": {"idx": [int(x) for x in bi], "final": [int(x) for x in bf]},
                       "none": {"idx": [int(x) for x in ni], "final": [int(x) for x in nf]}})
    sc = RefSimpleScanner.empty_scanner()
    strings = [b"", b"abc", b"\x00\xff"]
    bi, bf = sc.run_strings(strings)
    ni, nf = sc.run_strings(strings, flags=0)
    blob = sc.save()
    simple.append({"name": "simple_empty", "source": "pire_ut.cpp:760-830 (EmptyScanner) for SimpleScanner()", "pattern": None,
                   "options": "", "states": sc.size, "initial": sc.initial, "empty": True, "blob": write_blob("simple_empty", blob),
                   "blob_sha256": hashlib.sha256(blob).hexdigest(), "strings_hex": [x.hex() for x in strings],
                   "accepts_hex": [], "denies_hex": [x.hex() for x in strings],
                   "be": {"idx": [int(x) for x in bi], "final": [int(x) for x in bf]},
                   "none": {"idx": [int(x) for x in ni], "final": [int(x) for x in nf]}})

    # HalfFinalScanner (scanners/half_final.h): the reference's own vectors, tests/count_ut.cpp:541-550 and 575.
    # Five counter flavours per regexp (count_ut.cpp:503-519) plus all five glued (520-523).
    from oracle.binding import RefHalfFinalScanner
    HALF = [
        ("ab+", b"abbabbbabbbbbb", [3, 3, 3, 11, 3]),
        ("(ab)+", b"ababbababbab", [3, 3, 5, 5, 5]),
        ("(abab)+", b"ababababab", [1, 1, 4, 4, 2]),
        ("ab+c|b", b"abbbbbbbbbb", [1, 10, 10, 10, 10]),
        ("ab+c|b", b"abbbbbbbbbbb", [1, 10, 11, 11, 11]),
        ("ab+c|b", b"abbbbbbbbbbc", [1, 1, 10, 11, 10]),
        ("ab+c|b", b"abbbbbbbbbbbc", [1, 1, 11, 12, 11]),
        ("a\\w+c|b", b"abbbdbbbdbbc", [1, 1, 8, 9, 8]),
        ("a\\w+c|b", b"abbbdbbbdbb", [1, 8, 8, 8, 8]),
        ("a[a-z]+c|b", b"abeeeebeeeeeeeeeceeaeebeeeaeecceebeeaeebeeb", [2, 4, 7, 9, 7]),
        ("(\\w\\w)+", b"ab abbb ababa a", [3, 3, 8, 8, 5]),
    ]
    half = []
    seen = {}
    rng = np.random.RandomState(79)
    for pat, text, expect in HALF:
        if pat not in seen:
            glued = RefHalfFinalScanner.compile([pat] * 5, list(range(5)))
            name = "half_%d" % len(seen)
            blob = glued.save()
            extra = [bytes(rng.choice(np.frombuffer(b"abcde w", dtype=np.uint8), size=int(k))) for k in rng.randint(0, 60, size=25)]
            seen[pat] = {"name": name, "source": "tests/count_ut.cpp:503-550, 575", "pattern": pat, "states": glued.size,
                         "regexps": glued.regexps, "blob": write_blob(name, blob),
                         "blob_sha256": hashlib.sha256(blob).hexdigest(), "vectors": [], "_sc": glued, "_extra": extra}
            half.append(seen[pat])
        entry = seen[pat]
        idx, fin, res = entry["_sc"].run_strings([text])
        assert res[0].tolist() == expect, (pat, text, res[0].tolist(), expect)
        for m in range(5):     # the single (unglued) scanners state the same numbers, count_ut.cpp:530-532
            one = RefHalfFinalScanner.compile([pat], [m])
            assert int(one.run_strings([text])[2][0, 0]) == expect[m]
        entry["vectors"].append({"text_hex": text.hex(), "expect": expect})
    for entry in half:
        sc, extra = entry.pop("_sc"), entry.pop("_extra")
        strings = [bytes.fromhex(v["text_hex"]) for v in entry["vectors"]] + extra
        for key, flags in (("be", 3), ("none", 0)):
            idx, fin, res = sc.run_strings(strings, flags=flags)
            entry[key] = {"idx": [int(x) for x in idx], "final": [int(x) for x in fin], "results": res.tolist()}
        entry["strings_hex"] = [x.hex() for x in strings]

    # CountingScanner / AdvancedCountingScanner (extra/count.h): the reference's vectors, tests/count_ut.cpp:95-103,
    # plus a 3-regexp glue (count_ut.cpp:203-260 style)
    from oracle.binding import RefCountingScanner
    COUNT = [
        ("[a-z]+", "\\s", b"abc def, abc def ghi, abc", 3),
        ("\\w", "", b"abc abcdef abcd abcdefgh ac", 8),
        ("http", ".*", b"http://aaa, http://bbb, something in the middle, http://ccc, end", 3),
        ("abc", ".*", b"abcabcabcabc", 4),
        ("[a-z]+", ".*", b"abc def\0 abc\0 def ghi, abc\0", 6),
    ]
    counting = []
    rng = np.random.RandomState(80)
    extra = [bytes(rng.choice(np.frombuffer(b"abc def,http:/\n", dtype=np.uint8), size=int(k))) for k in rng.randint(0, 90, size=40)]
    for k, (re_, sep, text, expect) in enumerate(COUNT):
        for kind, kname in ((0, "basic"), (1, "advanced"), (2, "noglue")):
            sc = RefCountingScanner.compile(kind, [re_], [sep])
            strings = [text] + extra
            idx, res = sc.run_strings(strings)
            assert int(res[0, 0]) == expect, (re_, sep, kind, res[0], expect)
            nidx, nres = sc.run_strings(strings, flags=0)
            name = "count%d_%s" % (k, kname)
            blob = sc.save()
            counting.append({"name": name, "source": "tests/count_ut.cpp:95-103", "re": [re_], "sep": [sep], "kind": kind,
                             "states": sc.size, "letters": sc.letters, "regexps": sc.regexps, "expect_first": expect,
                             "blob": write_blob(name, blob), "strings_hex": [x.hex() for x in strings],
                             "be": {"idx": [int(x) for x in idx], "results": res.tolist()},
                             "none": {"idx": [int(x) for x in nidx], "results": nres.tolist()}})
    for kind, kname in ((0, "basic"), (1, "advanced"), (2, "noglue")):
        res_, seps_ = ["[a-z]+", "http", "abc"], ["\\s", ".*", ".*"]
        sc = RefCountingScanner.compile(kind, res_, seps_)
        strings = [c[2] for c in COUNT] + extra
        idx, res = sc.run_strings(strings)
        nidx, nres = sc.run_strings(strings, flags=0)
        name = "count_glued3_%s" % kname
        blob = sc.save()
        counting.append({"name": name, "source": "CountingScanner::Glue of three count_ut.cpp:95-103 scanners", "re": res_,
                         "sep": seps_, "kind": kind, "states": sc.size, "letters": sc.letters, "regexps": sc.regexps,
                         "expect_first": None, "blob": write_blob(name, blob), "strings_hex": [x.hex() for x in strings],
                         "be": {"idx": [int(x) for x in idx], "results": res.tolist()},
                         "none": {"idx": [int(x) for x in nidx], "results": nres.tolist()}})

    # NoGlueLimitCountingScanner with more than MAX_RE_COUNT = 16 regexps (what the class exists for, count.h:330-344)
    res_ = [c for c in "abcdefghijklmnopq"]
    seps_ = ["\\s"] * len(res_)
    sc = RefCountingScanner.compile(2, res_, seps_)
    wide_strings = [b"abc abc def qqq, a b c", b"", b"zzz"] + [bytes(rng.choice(np.frombuffer(b"abcdefghijklmnopqrs  \n", dtype=np.uint8), size=int(k)))
                                                             for k in rng.randint(0, 120, size=40)]
    idx, res = sc.run_strings(wide_strings)
    nidx, nres = sc.run_strings(wide_strings, flags=0)
    blob = sc.save()
    counting.append({"name": "count_wide17_noglue", "source": "NoGlueLimitCountingScanner::Glue of 17 scanners (count.h:330-344)",
                     "re": res_, "sep": seps_, "kind": 2, "states": sc.size, "letters": sc.letters, "regexps": sc.regexps,
                     "expect_first": None, "blob": write_blob("count_wide17_noglue", blob),
                     "strings_hex": [x.hex() for x in wide_strings],
                     "be": {"idx": [int(x) for x in idx], "results": res.tolist()},
                     "none": {"idx": [int(x) for x in nidx], "results": nres.tolist()}})

    # CapturingScanner (extra/capture.h): the reference's vectors, tests/capture_ut.cpp:93-153
    from oracle.binding import RefCapturingScanner
    CAPTURE = [
        ("capture_google", "capture_ut.cpp:93-129 (Trivial, Sequential)", "google_id\\s*=\\s*['\"]([a-z0-9]+)['\"]\\s*;", 1,
         [(b"google_id = 'abcde';", b"abcde"), (b"var google_id = 'abcde'; eval(google_id);", b"abcde"),
          (b"google_id != 'abcde';", None), (b"google_id = 'abcde'; google_id = 'xyz';", b"abcde"),
          (b"var google_id = 'abc de'; google_id = 'xyz';", b"xyz")]),
        ("capture_digits", "capture_ut.cpp:131-141 (NegatedTerminator)", "=(\\d+)[^\\d]", 1, [(b"=12345;", b"12345")]),
        ("capture_path", "capture_ut.cpp:143-153 (FakeEdges)", "(/to-match-with)", 1,
         [(b"/some/table/path/to-match-with", b"/to-match-with")]),
    ]
    capturing = []
    for name, source, pat, index, items in CAPTURE:
        sc = RefCapturingScanner.compile(pat, index, "i")
        strings = [s_ for s_, _ in items]
        strings += [bytes(rng.choice(np.frombuffer(b"google_id = 'ab1'; /to-match-with =12;x", dtype=np.uint8), size=int(k)))
                    for k in rng.randint(0, 100, size=30)]
        strings += [s_ + b" tail" for s_, _ in items] + [b"xx " + s_ for s_, _ in items]
        idx, fin, cap, b, e = sc.run_strings(strings)
        for (s_, want), c_, b_, e_ in zip(items, cap, b, e):
            got = s_[b_ - 1:e_ - 1] if c_ else None           # capture_ut.cpp:85-91
            assert got == want, (name, s_, got, want)
        blob = sc.save()
        capturing.append({"name": name, "source": source, "pattern": pat, "index": index, "states": sc.size,
                          "blob": write_blob(name, blob), "strings_hex": [x.hex() for x in strings],
                          "expect_hex": [None if w is None else w.hex() for _, w in items],
                          "idx": [int(x) for x in idx], "final": [int(x) for x in fin], "captured": [int(x) for x in cap],
                          "begin": [int(x) for x in b], "end": [int(x) for x in e]})

    # Scanner::Glue parts: every pattern of set_a / set_d compiled on its own (bench.cpp:114-129 glues such scanners
    # left to right); gluing these blobs must reproduce the big sets' tables, state for state.
    glue_parts = []
    for name, pats in (("set_a", SET_A), ("set_d", SET_D)):
        parts = []
        for k, pat in enumerate(pats):
            one = RefScanner.compile([pat], [""])
            blob = one.save()
            parts.append({"pattern": pat, "states": one.size, "letters": one.letters,
                          "blob": write_blob("%s_part%d" % (name, k), blob)})
        glue_parts.append({"name": name, "parts": parts})

    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference": "yandex/pire @ /root/reference (v0.0.6)",
                   "cases": cases, "big": big, "slow": slow, "simple": simple, "half_final": half, "counting": counting, "capturing": capturing, "glue_parts": glue_parts, "corpus": corpus}, f, indent=1)
    print("wrote", len(cases), "cases,", len(big), "big sets,", len(slow), "slow scanners,", len(simple), "simple scanners")


if __name__ == "__main__":
    main()
