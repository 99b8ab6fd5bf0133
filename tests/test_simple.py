"""Pire::SimpleScanner: oracle restatement vs the unmodified reference, product ingestion of
SimpleScanner::Save() bytes (host logic, no GPU), the Mmap-style entry points, and GPU parity.

Patterns and verdicts follow the reference's own tests: tests/common.h runs every SCANNER() block through
SimpleScanner as well (common.h:80-119, 158-221), so the ACCEPTS/DENIES of tests/pire_ut.cpp apply unchanged."""
import os

import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H



@pytest.fixture(scope="module")
def pa():
    import pire_amd

    return pire_amd


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available()
    return torch


# (pattern, options, accepted strings, denied strings) -- tests/pire_ut.cpp:40-161, 630 and SURVEY 8c known answers
CASES = [
    (r"hello\s+w.+d$", "", [b"hello world", b"say hello   wod", b"xxhello\tw--d"], [b"Hello world", b"hello world!", b"hello wd", b""]),
    (r"abc|def", "", [b"abc", b"def", b"xxabcyy"], [b"ab", b"de", b"", b"abd"]),
    (r"^x{3,6}$", "", [b"xxx", b"xxxx", b"xxxxxx"], [b"xx", b"xxxxxxx", b"", b"axxx"]),
    (r"ad*e", "", [b"ae", b"ade", b"addde", b"zaddez"], [b"a", b"ad", b"dde", b""]),
    (r"Head(Inner)*Tail", "", [b"HeadTail", b"HeadInnerTail", b"xHeadInnerInnerTailx"], [b"HeadInnerTai", b"HeadInneTail", b""]),
    (r"^abc$", "n", [b"abc"], [b"xabc", b"abcx", b"", b"ab"]),
    (r"[A-Z]+\d", "i", [b"abc1", b"Z9", b"xx q7 "], [b"123", b"abc", b""]),
]


def ref(pattern, options):
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return ob.RefSimpleScanner.compile(pattern, options)


def golden_simple():
    return H.golden().get("simple", [])


@pytest.mark.parametrize("case", golden_simple(), ids=lambda c: c["name"])
def test_oracle_matches_golden(case):
    """Committed fixtures (generated from the unmodified reference by tests/golden/make_golden.py)."""
    o = ob.OracleSimpleScanner(H.load_blob(case["blob"]))
    assert o.size == case["states"] and o.initial == case["initial"] and o.empty == case["empty"]
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for flags, key in ((3, "be"), (0, "none")):
        idx, fin = o.run_strings(strings, flags=flags)
        assert idx.tolist() == case[key]["idx"] and fin.tolist() == case[key]["final"]
    for s, f in zip(strings, case["be"]["final"]):
        if s in [bytes.fromhex(h) for h in case.get("accepts_hex", [])]:
            assert f == 1
        if s in [bytes.fromhex(h) for h in case.get("denies_hex", [])]:
            assert f == 0


@pytest.mark.parametrize("pattern,options,acc,den", CASES, ids=[c[0] for c in CASES])
def test_oracle_vs_reference(pattern, options, acc, den):
    r = ref(pattern, options)
    blob = r.save()
    o = ob.OracleSimpleScanner(blob)
    assert (o.size, o.initial, o.empty, o.regexps) == (r.size, r.initial, r.empty, r.regexps)
    ri, rf = r.run_strings(acc + den)
    assert rf.tolist() == [1] * len(acc) + [0] * len(den)          # the verdicts the reference's tests state
    rng = np.random.RandomState(len(pattern))
    strings = acc + den + H.random_strings(rng, 300, 60) + H.random_strings(rng, 300, 40, b"abcdexHeadInrTl w0123456789\t ")
    for flags in (3, 0, 1, 2):
        ri, rf = r.run_strings(strings, flags=flags)
        oi, of = o.run_strings(strings, flags=flags)
        assert (ri == oi).all() and (rf == of).all()
    for idx in range(r.size):
        assert o.final(idx) == r.final(idx)
        for ch in list(range(0, 256, 7)) + [258, 259]:
            assert o.next(idx, ch) == r.next(idx, ch)
    # Save -> Load round trip in the reference gives the same bytes (scanner_io.cpp:35-69)
    assert ob.RefSimpleScanner.load(blob).save() == blob


def test_empty_simple_scanner():
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    r = ob.RefSimpleScanner.empty_scanner()
    blob = r.save()
    o = ob.OracleSimpleScanner(blob)
    assert r.empty and o.empty and r.regexps == 0 and o.size == r.size
    strings = [b"", b"abc", b"\x00\xff"]
    assert r.run_strings(strings)[1].tolist() == [0, 0, 0] == o.run_strings(strings)[1].tolist()
    assert (r.run_strings(strings)[0] == o.run_strings(strings)[0]).all()


# ------------------------------------------------------------------ product ingestion (host only)
@pytest.mark.parametrize("case", golden_simple(), ids=lambda c: c["name"])
def test_product_ingests_simple_scanner(case, pa):
    blob = H.load_blob(case["blob"])
    t = pa.Table(blob)
    o = ob.OracleSimpleScanner(blob)
    assert t.info.scanner_type == 2
    assert (t.Size, t.initial, t.Empty, t.RegexpsCount) == (o.size, o.initial, o.empty, o.regexps)
    assert t.info.row_stride == 265 * 8 and t.info.header_size == 1
    for idx in range(o.size):
        assert t.Final(idx) == o.final(idx) and not t.Dead(idx)          # Dead() is always false, simple.h:64
        assert list(t.AcceptedRegexps(idx)) == o.accepted(idx)
        for ch in list(range(256)) + [258, 259]:
            assert t.Next(idx, ch) == o.next(idx, ch)
    # letter classes are OUR folding of identical columns: fewer than 264, and consistent with Next
    assert 1 <= t.LettersCount <= 264


def test_mmap_entry_points(pa, tmp_path):
    """pire_hip_table_mmap reports what Scanner::Mmap would consume; several scanners back to back can be walked;
    pire_hip_table_create_from_file = the blacklist sample's deployment flow."""
    cases = golden_simple()
    a = H.load_blob(cases[0]["blob"])
    b = H.load_blob([c for c in H.all_cases() if c["name"] == "survey_known_answer"][0]["blob"])
    image = a + b + a
    pos = 0
    kinds = []
    while pos < len(image):
        t, used = pa.Table.mmap(image[pos:])
        assert used > 0 and used % 8 == 0
        kinds.append((t.info.scanner_type, t.Size))
        pos += used
    assert pos == len(image) and [k for k, _ in kinds] == [2, 1, 2]
    path = os.path.join(tmp_path, "scanner.bin")
    with open(path, "wb") as f:
        f.write(b)
    t = pa.Table.from_file(path)
    assert t.info.scanner_type == 1 and t.Size == 11 and t.initial == 8
    with pytest.raises(pa.PireHipError):
        pa.Table.from_file(os.path.join(tmp_path, "missing.bin"))
    with open(path, "wb") as f:
        f.write(b[:100])
    with pytest.raises(pa.PireHipError):
        pa.Table.from_file(path)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("case", golden_simple(), ids=lambda c: c["name"])
def test_gpu_simple_scanner_parity(case, pa, torch_cuda):
    torch = torch_cuda
    blob = H.load_blob(case["blob"])
    t, o = pa.Table(blob), ob.OracleSimpleScanner(blob)
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for flags, key in ((3, "be"), (0, "none")):
        gi, gf = t.run_strings(strings, flags=flags)
        assert gi.tolist() == case[key]["idx"] and gf.tolist() == case[key]["final"]
    rng = np.random.RandomState(11)
    many = strings + H.random_strings(rng, 3000, 200, b"abcdexHeadInrTl w0123456789\t hello") + H.random_strings(rng, 500, 300)
    for flags in (3, 0, 1, 2):
        oi, of = o.run_strings(many, flags=flags)
        gi, gf, cnt = t.run(*H.pack(many), flags=flags, counts=True)      # ragged kernel
        assert (gi == oi).all() and (gf == of).all()
        assert cnt[0] == int(of.sum()) and cnt[1] == len(many)
        if o.regexps:
            assert cnt[2] == int(of.sum())
    # fixed-length records: the tiled kernel
    n, length = 4096, 512
    alphabet = np.frombuffer(b"abcdexHeadInrTl w0123456789\t hello", dtype=np.uint8)
    data = alphabet[rng.randint(0, len(alphabet), size=(n, length))].astype(np.uint8)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs)
    gi, gf = t.run_strided_host(data)
    from pire_amd import binding as pb
    assert pb.last_kernel() == "tiled"
    assert (gi == oi).all() and (gf == of).all()
