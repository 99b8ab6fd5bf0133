"""Scanner::Glue on ingested tables (host side): pire_hip_table_glue must number the product
automaton exactly as the reference does (multi.h:1005-1103, glue.h:35-159, determine.h:91-137), so that state indices
coming back from the GPU are the reference's."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    return pire_amd


def glue_all(pa, blobs, max_size=0):
    t = pa.Table(blobs[0])
    for b in blobs[1:]:
        t = pa.Table.glue(t, pa.Table(b), max_size)
    return t


def assert_same_table(t, o, sample_states=None):
    """t: product Table (ours), o: OracleScanner of the reference-glued blob: identical in every accessor."""
    assert (t.Size, t.LettersCount, t.RegexpsCount, t.initial, t.Empty) == (o.size, o.letters, o.regexps, o.initial, o.empty)
    for ch in range(264):
        if ch != 257:
            assert t.letter_class(ch) == o.letter_class(ch), ch
    states = range(o.size) if sample_states is None else sample_states
    chars = [0, 9, 32, 65, 90, 97, 100, 104, 119, 122, 200, 255, 258, 259] + list(range(40, 64))
    for s in states:
        assert t.Final(s) == o.final(s) and t.Dead(s) == o.dead(s)
        assert list(t.AcceptedRegexps(s)) == list(o.accepted(s))
        for ch in chars:
            assert t.Next(s, ch) == o.next(s, ch)


@pytest.mark.parametrize("name", ["set_d", "set_a"])
def test_glue_reproduces_reference_tables(pa, name):
    """The committed single-pattern blobs glued left to right == the committed reference-glued table (8 regexps)."""
    parts = [g for g in H.golden()["glue_parts"] if g["name"] == name][0]["parts"]
    big = [b for b in H.big_sets() if b["name"] == name][0]
    t = glue_all(pa, [H.load_blob(p["blob"]) for p in parts])
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    rng = np.random.RandomState(1)
    assert_same_table(t, o, sample_states=sorted(set(rng.randint(0, o.size, size=400).tolist() + [0, o.initial, o.size - 1])))
    assert t.info.ref_buf_size == len(H.load_blob(big["blob"])) - 80      # BufSize() of the scanner the reference glued
    assert t.info.row_stride == o.row_stride and t.info.header_size == o.header_size


def test_glue_small_exhaustive_and_edge_cases(pa):
    """Every state / letter of small products; empty operands (multi.h:1094-1097); max_size failure -> empty scanner."""
    parts = [g for g in H.golden()["glue_parts"] if g["name"] == "set_d"][0]["parts"]
    blobs = [H.load_blob(p["blob"]) for p in parts]
    if ob.ref_available():
        pats = [p["pattern"] for p in parts]
        for k in (2, 3, 5):
            ref = ob.RefScanner.compile(pats[:k], [""] * k)
            assert_same_table(glue_all(pa, blobs[:k]), ob.OracleScanner(ref.save()))
    empty = [c for c in H.all_cases() if c.get("geometry", {}).get("empty")]
    if empty:
        e = pa.Table(H.load_blob(empty[0]["blob"]))
        a = pa.Table(blobs[0])
        for t in (pa.Table.glue(e, a), pa.Table.glue(a, e)):
            assert (t.Size, t.LettersCount, t.RegexpsCount, t.initial) == (a.Size, a.LettersCount, a.RegexpsCount, a.initial)
    # the product of the first two set_d scanners needs more than 3 new states
    t = pa.Table.glue(pa.Table(blobs[0]), pa.Table(blobs[1]), 3)
    assert t.Empty and t.RegexpsCount == 0
    full = pa.Table.glue(pa.Table(blobs[0]), pa.Table(blobs[1]))
    exact = pa.Table.glue(pa.Table(blobs[0]), pa.Table(blobs[1]), full.Size - 1)      # exactly enough new states
    assert not exact.Empty and exact.Size == full.Size
    assert pa.Table.glue(pa.Table(blobs[0]), pa.Table(blobs[1]), full.Size - 2).Empty


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["set_d", "set_a"])
def test_gpu_glued_table_scans_like_the_reference_one(pa, name):
    assert pa.device_count() > 0
    parts = [g for g in H.golden()["glue_parts"] if g["name"] == name][0]["parts"]
    big = [b for b in H.big_sets() if b["name"] == name][0]
    t = glue_all(pa, [H.load_blob(p["blob"]) for p in parts])
    c = big["corpus"]
    data = ob.corpus_fill(c["seed"], 0, c["n"], c["len"], H.plants_for(big))
    idx, fin, cnt = t.run_strided_host(data, counts=True)
    assert idx.tolist() == c["idx"] and fin.tolist() == c["final"]          # the reference's own state indices
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    rng = np.random.RandomState(3)
    many = H.random_strings(rng, 3000, 200, b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet")
    oi, of = o.run_strings(many)
    gi, gf = t.run_strings(many)
    assert (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["set_d", "set_a"])
def test_gpu_side_glue_equals_host_glue_and_reference(pa, name):
    """pire_hip_table_glue_gpu: the level-synchronous BFS on the device numbers the product exactly like the
    sequential loop (every intermediate of the 8-way glue is produced on the GPU and glued further)."""
    assert pa.device_count() > 0
    parts = [g for g in H.golden()["glue_parts"] if g["name"] == name][0]["parts"]
    blobs = [H.load_blob(p["blob"]) for p in parts]
    t = pa.Table(blobs[0])
    for b in blobs[1:]:
        t = pa.Table.glue(t, pa.Table(b), gpu=True)
    big = [b for b in H.big_sets() if b["name"] == name][0]
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    rng = np.random.RandomState(2)
    assert_same_table(t, o, sample_states=sorted(set(rng.randint(0, o.size, size=400).tolist() + [0, o.initial, o.size - 1])))
    # and the failure rule (determine.h:112-113) is the same on both sides
    a, b2 = pa.Table(blobs[0]), pa.Table(blobs[1])
    full = pa.Table.glue(a, b2, gpu=True)
    host = pa.Table.glue(a, b2)
    assert full.Size == host.Size
    assert not pa.Table.glue(a, b2, full.Size - 1, gpu=True).Empty
    assert pa.Table.glue(a, b2, full.Size - 2, gpu=True).Empty
    c = big["corpus"]
    data = ob.corpus_fill(c["seed"], 0, c["n"], c["len"], H.plants_for(big))
    idx, fin = t.run_strided_host(data)
    assert idx.tolist() == c["idx"] and fin.tolist() == c["final"]


def test_gpu_glue_without_device_fails_loudly(pa):
    if pa.device_count() > 0:
        pytest.skip("a GPU is present")
    parts = [g for g in H.golden()["glue_parts"] if g["name"] == "set_d"][0]["parts"]
    a, b = pa.Table(H.load_blob(parts[0]["blob"])), pa.Table(H.load_blob(parts[1]["blob"]))
    with pytest.raises(pa.PireHipError):
        pa.Table.glue(a, b, gpu=True)
