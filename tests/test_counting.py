"""Pire::CountingScanner / AdvancedCountingScanner (extra/count.h): the first scanners on the path
whose Action is not a no-op.  Known answers are the reference's own, tests/count_ut.cpp:95-103."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H


def cases():
    return H.golden().get("counting", [])


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    return pire_amd


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_oracle_matches_golden(case):
    o = ob.OracleCountingScanner(H.load_blob(case["blob"]), case["kind"])
    assert (o.size, o.letters, o.regexps) == (case["states"], case["letters"], case["regexps"])
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for key, flags in (("be", 3), ("none", 0)):
        idx, res = o.run_strings(strings, flags=flags)
        assert idx.tolist() == case[key]["idx"] and res.tolist() == case[key]["results"]
    if case["expect_first"] is not None:
        assert case["be"]["results"][0][0] == case["expect_first"]          # the number written in count_ut.cpp


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_oracle_vs_live_reference(case):
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    blob = H.load_blob(case["blob"])
    r, o = ob.RefCountingScanner.load(case["kind"], blob), ob.OracleCountingScanner(blob, case["kind"])
    assert r.save() == blob
    rng = np.random.RandomState(9)
    strings = H.random_strings(rng, 500, 120, b"abc def,http:/\n") + H.random_strings(rng, 100, 60)
    for flags in (3, 0, 1, 2):
        ri, rr = r.run_strings(strings, flags=flags)
        oi, orr = o.run_strings(strings, flags=flags)
        assert (ri == oi).all() and (rr == orr).all()


def test_product_ingests_counting_tables(pa):
    for case in cases():
        blob = H.load_blob(case["blob"])
        t = pa.CountingTable(blob, case["kind"])
        o = ob.OracleCountingScanner(blob, case["kind"])
        assert (t.Size, t.LettersCount, t.RegexpsCount, t.initial) == (o.size, o.letters, o.regexps, o.initial)
    with pytest.raises(pa.PireHipError):
        pa.CountingTable(H.load_blob(cases()[0]["blob"])[:100], 0)
    # a NoGlueLimitCountingScanner blob cannot be run as one of the other classes (its actions are list indices)
    noglue = [c for c in cases() if c["kind"] == 2][0]
    with pytest.raises(pa.PireHipError):
        pa.CountingTable(H.load_blob(noglue["blob"]), 0).run_strings([b"abc"])
    with pytest.raises(pa.PireHipError):                                   # a Scanner blob is not a LoadedScanner
        pa.CountingTable(H.load_blob([c for c in H.all_cases() if c["name"] == "survey_known_answer"][0]["blob"]), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_gpu_counting_parity(case, pa):
    assert pa.device_count() > 0
    blob = H.load_blob(case["blob"])
    t, o = pa.CountingTable(blob, case["kind"]), ob.OracleCountingScanner(blob, case["kind"])
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    for key, flags in (("be", 3), ("none", 0)):
        idx, res = t.run_strings(strings, flags=flags)
        assert idx.tolist() == case[key]["idx"] and res.tolist() == case[key]["results"]
    rng = np.random.RandomState(21)
    many = H.random_strings(rng, 6000, 400, b"abc def,http:/\n") + [b""] * 3 + H.random_strings(rng, 500, 100)
    for flags in (3, 0, 1, 2):
        oi, orr = o.run_strings(many, flags=flags)
        gi, gr = t.run_strings(many, flags=flags)
        assert (gi == oi).all() and (gr == orr).all()
    assert orr.sum() > 0


@pytest.mark.gpu
def test_gpu_counting_sixteen_regexps(pa):
    """The 16-counter instantiation (more than 8 regexps glued)."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    res_ = ["a", "b", "c", "ab", "bc", "ca", "[ab]+", "[bc]+", "d", "abc"]
    seps = [".*"] * len(res_)
    for kind in (0, 1):
        blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
        t, o = pa.CountingTable(blob, kind), ob.OracleCountingScanner(blob, kind)
        assert t.RegexpsCount == len(res_)
        rng = np.random.RandomState(22)
        many = H.random_strings(rng, 3000, 200, b"abcd ")
        oi, orr = o.run_strings(many)
        gi, gr = t.run_strings(many)
        assert (gi == oi).all() and (gr == orr).all() and gr.sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 9])
def test_gpu_counting_every_counter_instantiation(pa, k):
    """The counters live in 1, 2, 4, 8 or 16 register slots depending on the number of regexps: every instantiation,
    all three scanner classes, against the oracle (and through it the reference)."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    res_ = ["a", "b", "ab", "[ab]+", "c", "bc", "d", "abc", "ca"][:k]
    seps = [".*", "\\s", ".*", "c", ".*", ".*", "\\s", ".*", ".*"][:k]
    rng = np.random.RandomState(30 + k)
    many = H.random_strings(rng, 2000, 300, b"abcd \n") + [b"", b"a", b"ab ab ab"]
    for kind in (0, 1, 2):
        try:
            blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
        except ValueError:
            continue            # this class cannot glue that many
        t, o = pa.CountingTable(blob, kind), ob.OracleCountingScanner(blob, kind)
        assert t.RegexpsCount == k
        for flags in (3, 0):
            oi, orr = o.run_strings(many, flags=flags)
            gi, gr = t.run_strings(many, flags=flags)
            assert (gi == oi).all() and (gr == orr).all(), (k, kind, flags)
        assert orr.sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["count0_advanced", "count0_basic", "count_glued3_advanced", "count0_noglue"])
def test_gpu_counting_big_batches(pa, name):
    """Tens of thousands of strings (several per lane): every length from 0 up, unaligned starts, strings that end
    inside a 16-byte block."""
    case = [c for c in cases() if c["name"] == name]
    if not case:
        pytest.skip("fixture not present")
    case = case[0]
    blob = H.load_blob(case["blob"])
    t, o = pa.CountingTable(blob, case["kind"]), ob.OracleCountingScanner(blob, case["kind"])
    rng = np.random.RandomState(40)
    many = ([b"", b"a", b"ab cd"] + H.random_strings(rng, 30000, 120, b"abc def,http:/\n") +
            H.random_strings(rng, 200, 3000, b"abcdefgh \n") + [b""] * 5)
    for flags in (3, 0):
        oi, orr = o.run_strings(many, flags=flags)
        gi, gr = t.run_strings(many, flags=flags)
        assert (gi == oi).all() and (gr == orr).all(), (name, flags)
    assert orr.sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3, 4, 7, 16])
def test_gpu_counting_packed_kernel_and_its_overflow_list(pa, k):
    """CountingScanner / AdvancedCountingScanner take the dense-row kernel with packed 16-bit counters first
    (CountingPackedKernel: 1, 2, 4 or 8 registers of counter pairs); strings longer than 65 000 bytes -- a count could
    outgrow 16 bits -- go onto its overflow list and through the 32-bit kernel.  Both kernels (PIRE_HIP_RUN_GENERIC
    keeps the 32-bit one alone) against the oracle, with counts beyond 65 535 in the batch."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    from pire_amd import binding as pb

    res_ = (["a", "b", "ab", "[ab]+", "c", "bc", "d", "abc", "ca"] + ["[a-c]%d" % i for i in range(7)])[:k]
    seps = ([".*", "\\s", ".*", "c", ".*", ".*", "\\s", ".*", ".*"] + [".*"] * 7)[:k]
    rng = np.random.RandomState(50 + k)
    many = H.random_strings(rng, 1500, 300, b"abcd \n012") + [b"", b"a"]
    many += [b"a " * 40000, bytes(rng.choice(np.frombuffer(b"abcd \n", dtype=np.uint8), size=70001)), b"ab" * 32499]
    for kind in (0, 1):
        try:
            blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
        except ValueError:
            continue            # this class cannot glue that many
        t, o = pa.CountingTable(blob, kind), ob.OracleCountingScanner(blob, kind)
        for flags in (3, 0):
            oi, orr = o.run_strings(many, flags=flags)
            gi, gr = t.run_strings(many, flags=flags)
            packed = pb.last_kernel() == "counting_packed"
            assert (gi == oi).all() and (gr == orr).all(), (k, kind, flags, pb.last_kernel())
            hi, hr = t.run_strings(many, flags=flags | pb.FLAG_GENERIC)
            assert pb.last_kernel() == "counting"
            assert (hi == oi).all() and (hr == orr).all(), (k, kind, flags)
        assert packed or t.Size > 255 or t.Size * 512 > 130 * 1024, "a table of %d states should have taken the packed kernel" % t.Size
        assert orr.max() > 20000


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3, 4, 7, 8])
def test_gpu_counting_row_kernel(pa, k):
    """CountingRowKernel (whole lines of text per lane, entries that are LDS addresses, no branch around the action;
    tables of up to 64 states and eight regexps): all three counter-register instantiations, both scanner classes, all four
    Begin/End combinations, the overflow list -- against the oracle and against the 16-bit-entry kernel (pire_hip_config.counting_variant 2 / 1)."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    from pire_amd import binding as pb

    res_ = ["a", "b", "ab", "[ab]+", "c", "bc", "d", "abc"][:k]
    seps = [".*", "\\s", ".*", "c", ".*", ".*", "\\s", ".*"][:k]
    rng = np.random.RandomState(70 + k)
    many = H.random_strings(rng, 5000, 300, b"abcd \n012") + [b"", b"a", b"ab ab ab"] + [b""] * 70
    many += [b"a " * 40000, b"ab" * 32499, b"abc " * 16250]
    took_rows = 0
    for kind in (0, 1):
        try:
            blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
        except ValueError:
            continue            # this class cannot glue that many
        t, o = pa.CountingTable(blob, kind), ob.OracleCountingScanner(blob, kind)
        for flags in (3, 0, 1, 2):
            oi, orr = o.run_strings(many, flags=flags)
            with pb.config(counting_variant=2):
                gi, gr = t.run_strings(many, flags=flags)
                rows = pb.last_kernel() in ("counting_rows", "counting_letter_rows")
            # rows indexed by the byte for tables of up to 64 states, by the table's letters for any other that fits the LDS
            assert pb.last_kernel() == ("counting_rows" if t.Size <= 64 else "counting_letter_rows"), (k, kind, t.Size, pb.last_kernel())
            took_rows += rows
            assert (gi == oi).all() and (gr == orr).all(), (k, kind, flags)
            with pb.config(counting_variant=1):
                hi, hr = t.run_strings(many, flags=flags)
                assert pb.last_kernel() in ("counting_packed", "counting")
            assert (hi == oi).all() and (hr == orr).all(), (k, kind, flags)
        assert orr.max() > 20000
    assert took_rows, "no table of this size took a row kernel"


@pytest.mark.gpu
def test_gpu_counting_row_kernel_is_the_default_for_batches_that_fill_the_gpu(pa):
    from pire_amd import binding as pb

    case = [c for c in cases() if c["name"] == "count_glued3_advanced"][0]
    blob = H.load_blob(case["blob"])
    t, o = pa.CountingTable(blob, case["kind"]), ob.OracleCountingScanner(blob, case["kind"])
    rng = np.random.RandomState(81)
    many = H.random_strings(rng, 70000, 60, b"abc def,http:/\n")
    oi, orr = o.run_strings(many)
    gi, gr = t.run_strings(many)
    assert pb.last_kernel() == "counting_rows"
    assert (gi == oi).all() and (gr == orr).all()
    few = many[:3000]
    gi, gr = t.run_strings(few)
    assert pb.last_kernel() == "counting_packed"
    assert (gi == oi[:3000]).all() and (gr == orr[:3000]).all()


@pytest.mark.gpu
def test_gpu_counting_letter_rows_on_large_tables(pa):
    """CountingRowKernel with rows indexed by the table's letters: glued scanners of hundreds of states (7 regexps:
    143 states as CountingScanner, 573 as AdvancedCountingScanner, more than 255 distinct actions), all four Begin/End
    combinations, the overflow list -- against the oracle and the 32-bit kernel."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    from pire_amd import binding as pb

    res_ = ["[a-z]+", "http", "abc", "[0-9]+", "e", "th", "ing"]
    seps = ["\\s", ".*", ".*", "\\s", ".*", ".*", ".*"]
    rng = np.random.RandomState(91)
    many = H.random_strings(rng, 6000, 400, b"abc the thing http://e 0123 \n") + [b"", b"e", b"thing 42"] + [b""] * 70
    many += [b"the thing " * 7000, b"e" * 64999]
    for kind in (0, 1):
        blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
        t, o = pa.CountingTable(blob, kind), ob.OracleCountingScanner(blob, kind)
        assert t.Size > 64
        for flags in (3, 0, 1, 2):
            oi, orr = o.run_strings(many, flags=flags)
            with pb.config(counting_variant=2):
                gi, gr = t.run_strings(many, flags=flags)
                assert pb.last_kernel() == "counting_letter_rows", (kind, t.Size, pb.last_kernel())
            assert (gi == oi).all() and (gr == orr).all(), (kind, flags)
            hi, hr = t.run_strings(many, flags=flags | pb.FLAG_GENERIC)
            assert pb.last_kernel() == "counting"
            assert (hi == oi).all() and (hr == orr).all(), (kind, flags)
        assert orr.max() > 20000


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [1, 17, 127])
def test_gpu_row_kernels_with_text_at_any_alignment(pa, shift):
    """The row kernels read whole 128-byte lines by ABSOLUTE address: device text that starts `shift` bytes into a line
    (first line of the first string reaches in front of the text, last line behind it), offsets on the device --
    counting (rows by byte and by letter) and capture against the oracle."""
    import torch
    from pire_amd import binding as pb

    rng = np.random.RandomState(100 + shift)
    many = H.random_strings(rng, 3000, 500, b"abc def,http:/\n=123 ") + [b"", b"a"] + H.random_strings(rng, 40, 3000, b"ab =12 \n")
    text, offs = ob.pack_strings(many)
    buf = torch.zeros(len(text) + 256, dtype=torch.uint8, device="cuda")
    buf[shift:shift + len(text)] = torch.as_tensor(np.asarray(text, dtype=np.uint8).copy())
    doffs = torch.as_tensor(np.asarray(offs, dtype=np.uint64).astype(np.int64), device="cuda")
    n = len(many)
    stream = torch.cuda.current_stream().cuda_stream
    cases_ = [c for c in cases() if c["name"] in ("count_glued3_advanced", "count0_basic")]
    assert cases_
    with pb.config(counting_variant=2):
        for case in cases_:
            blob = H.load_blob(case["blob"])
            t, o = pa.CountingTable(blob, case["kind"]), ob.OracleCountingScanner(blob, case["kind"])
            oi, orr = o.run_strings(many)
            idx = torch.empty(n, dtype=torch.int32, device="cuda")
            res = torch.empty((n, t.RegexpsCount), dtype=torch.int32, device="cuda")
            t.run_device(buf.data_ptr() + shift, doffs.data_ptr(), n, 3, idx.data_ptr(), res.data_ptr(), stream)
            torch.cuda.synchronize()
            assert pb.last_kernel() == "counting_rows"
            assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (res.cpu().numpy().astype(np.uint32) == orr).all(), case["name"]
        cap = [c for c in H.golden()["capturing"] if c["name"] == "capture_digits"][0]
        blob = H.load_blob(cap["blob"])
        t, o = pa.CountingTable(blob, 0), ob.OracleCountingScanner(blob, 0)
        want = o.capture(*ob.pack_strings(many))
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        fin = torch.empty(n, dtype=torch.uint8, device="cuda")
        bg = torch.empty(n, dtype=torch.int64, device="cuda")
        en = torch.empty(n, dtype=torch.int64, device="cuda")
        with pb.config(no_ragged_act=1):
            t.capture_device(buf.data_ptr() + shift, doffs.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), bg.data_ptr(), en.data_ptr(), stream)
            torch.cuda.synchronize()
            assert pb.last_kernel() == "capture_rows"
        widx, wfin, _, wbeg, wend = want
        assert (idx.cpu().numpy().astype(np.uint32) == widx).all() and (fin.cpu().numpy() == wfin).all()
        assert (bg.cpu().numpy() == wbeg).all() and (en.cpu().numpy() == wend).all() and (wbeg >= 0).sum() > 0


def test_counting_table_forms_on_the_host(pa):
    """Which device forms BuildDenseCounting / BuildLetterRows give a table (no GPU needed): the golden tables get the
    16-bit-entry form, the byte-indexed rows (<= 64 states) and the letter-indexed rows; glued scanners of hundreds of
    states keep the letter-indexed rows as long as (states + 1) x (letters + 1) entries and their actions fit 150 KB."""
    for case in cases():
        t = pa.CountingTable(H.load_blob(case["blob"]), case["kind"])
        f = t.forms()
        if case["kind"] == 2 or t.RegexpsCount > 8:      # NoGlueLimitCountingScanner: action lists, not bit words
            assert f["packed_nreg"] == 0 and f["letter_rows_nreg"] == 0 and not f["byte_rows"], (case["name"], f)
            continue
        nreg = 1 if t.RegexpsCount <= 2 else 2 if t.RegexpsCount <= 4 else 4
        assert f["packed_nreg"] == nreg and f["letter_rows_nreg"] == nreg, (case["name"], f)
        assert f["byte_rows"] == (t.Size <= 64) and f["byte_rows_lds"] == (((t.Size + 1) * 2056 + 15) // 16 * 16 + 256 * 8 * nreg if t.Size <= 64 else 0)
        assert f["letter_rows_lds"] <= 150 * 1024 and f["letter_rows_actions"] >= 1
        assert f["letter_rows_lds"] == ((t.Size + 1) * (t.LettersCount + 1) * 8 + 15) // 16 * 16 + (f["letter_rows_actions"] + 1) * 8 * nreg + 512
    if not ob.ref_available():
        return
    res_ = ["[a-z]+", "http", "abc", "[0-9]+", "e", "th", "ing"]
    seps = ["\\s", ".*", ".*", "\\s", ".*", ".*", ".*"]
    small = pa.CountingTable(ob.RefCountingScanner.compile(0, res_, seps).save(), 0)
    big = pa.CountingTable(ob.RefCountingScanner.compile(1, res_, seps).save(), 1)
    fs, fb = small.forms(), big.forms()
    assert 64 < small.Size <= 255 and fs["packed_nreg"] == 4 and not fs["byte_rows"] and fs["letter_rows_nreg"] == 4
    assert big.Size > 255 and fb["packed_nreg"] == 0 and fb["letter_rows_nreg"] == 4 and fb["letter_rows_actions"] > 255
    assert fb["letter_rows_lds"] <= 150 * 1024
    # a table whose rows do not fit: 16 glued regexps have no packed form at all
    many = ["a", "b", "c", "ab", "bc", "ca", "[ab]+", "[bc]+", "d", "abc"]
    wide = pa.CountingTable(ob.RefCountingScanner.compile(1, many, [".*"] * len(many)).save(), 1)
    assert wide.forms()["letter_rows_nreg"] == 0
