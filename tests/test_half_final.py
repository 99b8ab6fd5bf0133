"""Pire::HalfFinalScanner (scanners/half_final.h): per-regexp match counting through TakeAction.

Known answers are the reference's own: tests/count_ut.cpp:541-550 and 575 (five counter flavours per regexp, glued
into one 5-regexp scanner as count_ut.cpp:503-527 does)."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H


def half_cases():
    return H.golden().get("half_final", [])


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    return pire_amd


@pytest.mark.parametrize("case", half_cases(), ids=lambda c: c["name"])
def test_oracle_matches_reference_vectors(case):
    o = ob.OracleScanner(H.load_blob(case["blob"]))
    assert o.regexps == case["regexps"] == 5
    for v in case["vectors"]:
        text = bytes.fromhex(v["text_hex"])
        idx, fin, res = o.run_half_final(*ob.pack_strings([text]))
        assert res[0].tolist() == v["expect"]            # the numbers written in tests/count_ut.cpp
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for key, flags in (("be", 3), ("none", 0)):
        idx, fin, res = o.run_half_final(*ob.pack_strings(strings), flags=flags)
        assert idx.tolist() == case[key]["idx"] and fin.tolist() == case[key]["final"]
        assert res.tolist() == case[key]["results"]


@pytest.mark.parametrize("case", half_cases(), ids=lambda c: c["name"])
def test_oracle_vs_live_reference(case):
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    blob = H.load_blob(case["blob"])
    r, o = ob.RefHalfFinalScanner.load(blob), ob.OracleScanner(blob)
    assert r.save() == blob
    rng = np.random.RandomState(5)
    strings = H.random_strings(rng, 400, 80, b"abcde w") + H.random_strings(rng, 100, 50)
    for flags in (3, 0, 1, 2):
        ri, rf, rr = r.run_strings(strings, flags=flags)
        oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=flags)
        assert (ri == oi).all() and (rf == of).all() and (rr == orr).all()


def test_plain_half_final_scanner_and_many_regexps():
    """HalfFinalScanner(Fsm) (MakeScanner, half_final.h:38-46) and a glue of more than 8 regexps (the unpacked
    counter path of the kernel)."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    pats = ["ab+", "b", "a", "(ab)+", "c", "bc", "abc", "b+", "[ab]c", "ca"]
    plain = ob.RefHalfFinalScanner.compile(pats[:3], [ob.RefHalfFinalScanner.PLAIN] * 3)
    po = ob.OracleScanner(plain.save())
    probe = [b"ab", b"abb", b"b", b"a", b"ba", b""]
    a, b = plain.run_strings(probe), po.run_half_final(*ob.pack_strings(probe))
    assert all((x == y).all() for x, y in zip(a, b))
    r = ob.RefHalfFinalScanner.compile(pats, [ob.RefHalfFinalScanner.NONGREEDY_SIMPLE] * len(pats))
    o = ob.OracleScanner(r.save())
    assert r.regexps == len(pats) == o.regexps
    rng = np.random.RandomState(6)
    strings = H.random_strings(rng, 300, 60, b"abc")
    ri, rf, rr = r.run_strings(strings)
    oi, of, orr = o.run_half_final(*ob.pack_strings(strings))
    assert (ri == oi).all() and (rf == of).all() and (rr == orr).all() and rr.sum() > 0


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("case", half_cases(), ids=lambda c: c["name"])
def test_gpu_half_final_golden(case, pa):
    assert pa.device_count() > 0
    blob = H.load_blob(case["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    for v in case["vectors"]:
        idx, fin, res = t.run_half_final(*H.pack([bytes.fromhex(v["text_hex"])]))
        assert res[0].tolist() == v["expect"]
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for key, flags in (("be", 3), ("none", 0)):
        idx, fin, res = t.run_half_final(*H.pack(strings), flags=flags)
        assert idx.tolist() == case[key]["idx"] and fin.tolist() == case[key]["final"]
        assert res.tolist() == case[key]["results"]
    rng = np.random.RandomState(17)
    many = H.random_strings(rng, 5000, 300, b"abcde w") + [b""] * 3 + H.random_strings(rng, 500, 100)
    for flags in (3, 0, 1, 2):
        oi, of, orr = o.run_half_final(*ob.pack_strings(many), flags=flags)
        gi, gf, gr = t.run_half_final(*H.pack(many), flags=flags)
        assert (gi == oi).all() and (gf == of).all() and (gr == orr).all()
    # the same table through the plain Scanner entry point still gives Scanner results
    oi, of = o.run(*ob.pack_strings(many))
    gi, gf = t.run(*H.pack(many))
    assert (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
def test_gpu_half_final_many_regexps_and_big_table(pa, tmp_path):
    """> 8 regexps (counters in HBM rows instead of registers) and a table with cold states (set_a walked as a
    half-final scanner: every step that ends in a Final state counts)."""
    if ob.ref_available():
        pats = ["ab+", "b", "a", "(ab)+", "c", "bc", "abc", "b+", "[ab]c", "ca"]
        blob = ob.RefHalfFinalScanner.compile(pats, [2] * len(pats)).save()
        t, o = pa.Table(blob), ob.OracleScanner(blob)
        rng = np.random.RandomState(8)
        many = H.random_strings(rng, 3000, 200, b"abc")
        oi, of, orr = o.run_half_final(*ob.pack_strings(many))
        gi, gf, gr = t.run_half_final(*H.pack(many))
        assert (gi == oi).all() and (gf == of).all() and (gr == orr).all() and gr.sum() > 0
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    c = big["corpus"]
    data = ob.corpus_fill(c["seed"], 0, 512, 1024, H.plants_for(big))
    offs = np.arange(513, dtype=np.uint64) * 1024
    oi, of, orr = o.run_half_final(data.reshape(-1), offs)
    gi, gf, gr = t.run_half_final(data.reshape(-1), offs)
    assert (gi == oi).all() and (gf == of).all() and (gr == orr).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", half_cases(), ids=lambda c: c["name"])
def test_gpu_half_final_on_the_row_kernel(case, pa):
    """Dense HalfFinal counting on the counting scanners' row kernel (round 4: whole text lines per lane, the increments
    of the target state as the step's action): every Begin/End combination, strings longer than the 16-bit counters hold
    (they go through the one-string-per-lane kernel, from the row kernel's list), empty strings -- against the oracle
    and against the kernels of round 3."""
    from pire_amd import binding as pb

    blob = H.load_blob(case["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(23)
    many = H.random_strings(rng, 5000, 400, b"abcde w") + [b""] * 70 + H.random_strings(rng, 300, 100)
    many += [b"ab ab c " * 9000, b"b" * 65000, b"a" * 65001, b"abc" * 30000]
    took = 0
    for flags in (3, 0, 1, 2):
        oi, of, orr = o.run_half_final(*ob.pack_strings(many), flags=flags)
        with pb.config(counting_variant=2, no_ragged_act=1):
            gi, gf, gr = t.run_half_final(*H.pack(many), flags=flags)
            took += pb.last_kernel() == "half_final_rows"
        assert (gi == oi).all() and (gf == of).all() and (gr == orr).all(), (flags, pb.last_kernel())
        with pb.config(counting_variant=1, no_ragged_act=1):
            hi, hf, hr = t.run_half_final(*H.pack(many), flags=flags)
            assert pb.last_kernel() == "half_final"
        assert (hi == oi).all() and (hf == of).all() and (hr == orr).all(), flags
    info = t.info
    if info.regexps <= 8 and info.letters <= 255:
        assert took == 4, "a table of %d states x %d letters, %d regexps should have taken the row kernel" % (info.states, info.letters, info.regexps)
    assert orr.sum() > 0
