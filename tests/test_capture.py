"""Pire::CapturingScanner (extra/capture.h): the substring matched by one pair of parentheses.  Known answers are the
reference's own, tests/capture_ut.cpp:93-153."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H


def cases():
    return H.golden().get("capturing", [])


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    return pire_amd


def check(case, result):
    idx, fin, cap, b, e = result
    assert idx.tolist() == case["idx"] and fin.tolist() == case["final"] and cap.tolist() == case["captured"]
    assert b.tolist() == case["begin"] and e.tolist() == case["end"]
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for s, want, c_, b_, e_ in zip(strings, case["expect_hex"], cap, b, e):
        got = s[b_ - 1:e_ - 1] if c_ else None               # tests/capture_ut.cpp:85-91
        assert got == (None if want is None else bytes.fromhex(want))


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_oracle_matches_golden_and_reference(case):
    blob = H.load_blob(case["blob"])
    o = ob.OracleCountingScanner(blob, 0)
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    check(case, o.capture(*ob.pack_strings(strings)))
    if ob.ref_available():
        r = ob.RefCapturingScanner.load(blob)
        assert r.save() == blob
        rng = np.random.RandomState(3)
        pool = [bytes.fromhex(h) for h in case["strings_hex"][:5]]
        many = []
        for _ in range(400):
            parts = [pool[rng.randint(0, len(pool))] if rng.randint(0, 2) else
                     bytes(rng.choice(np.frombuffer(b"google_id ='\";x1/", dtype=np.uint8), size=rng.randint(0, 12)))
                     for _ in range(rng.randint(0, 4))]
            many.append(b"".join(parts))
        for flags in (3, 0, 1, 2):
            a, b = r.run_strings(many, flags=flags), o.capture(*ob.pack_strings(many), flags=flags)
            assert all((x == y).all() for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_gpu_capture_parity(case, pa, cfg):
    assert pa.device_count() > 0
    blob = H.load_blob(case["blob"])
    t, o = pa.CountingTable(blob, 0), ob.OracleCountingScanner(blob, 0)
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    check(case, t.capture(*H.pack(strings)))
    rng = np.random.RandomState(5)
    pool = strings[:5]
    many = []
    for _ in range(5000):
        parts = [pool[rng.randint(0, len(pool))] if rng.randint(0, 2) else
                 bytes(rng.choice(np.frombuffer(b"google_id ='\";x1/", dtype=np.uint8), size=rng.randint(0, 40)))
                 for _ in range(rng.randint(0, 6))]
        many.append(b"".join(parts))
    from pire_amd import binding as pb

    # long strings among the short ones, the brackets in every position relative to the 16-byte chunks and 128-byte
    # windows of the ragged kernel, strings that are nothing but brackets
    for k in range(0, 300, 7):
        many.append(b"x" * k + pool[k % len(pool)] + b"y" * (300 - k))
    many += [pool[0] * 40, pool[1] * 3 + b"z" * 5000 + pool[2], b""]
    cfg.set(ragged_act_always=1)   # whatever the scanner's share of action states (the library's own choice: below)
    for flags in (3, 0, 1, 2):
        a = o.capture(*ob.pack_strings(many), flags=flags)
        b = t.capture(*H.pack(many), flags=flags)                       # >= 256 strings: the ragged kernel with actions
        assert pb.last_kernel() == "ragged_capture"
        assert all((x == y).all() for x, y in zip(a, b)), flags
        c = t.capture(*H.pack(many), flags=flags | pb.FLAG_GENERIC)     # the plain one-string-per-lane kernel
        assert pb.last_kernel() == "capture"
        assert all((x == y).all() for x, y in zip(a, c)), flags
        cfg.set(ragged_act_always=0, no_ragged_act=1)                   # ... and its dense-row form
        d = t.capture(*H.pack(many), flags=flags)
        assert pb.last_kernel() == "capture_dense"
        assert all((x == y).all() for x, y in zip(a, d)), flags
        cfg.set(counting_variant=2)                                     # ... and that on whole text lines (round 4)
        r = t.capture(*H.pack(many), flags=flags)
        assert pb.last_kernel() == ("capture_rows" if t.Size <= 34 else "capture_dense")
        assert all((x == y).all() for x, y in zip(a, r)), flags
        cfg.set(ragged_act_always=1, no_ragged_act=0, counting_variant=0)
    assert a[2].sum() > 0
    # left to itself the library keeps the one-string-per-lane kernel for scanners that are in an action state on most
    # bytes of text (=(\d+)[^\d] re-arms BeginCapture all the time), and takes the ragged one for the others
    cfg.set(ragged_act_always=0)
    b = t.capture(*H.pack(many))
    assert pb.last_kernel() == {"capture_digits": "capture_dense", "capture_google": "ragged_capture"}.get(case["name"], pb.last_kernel())
    assert all((x == y).all() for x, y in zip(o.capture(*ob.pack_strings(many)), b))
