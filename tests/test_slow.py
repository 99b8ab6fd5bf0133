"""Pire::SlowScanner (BASELINE config 5b): oracle vs golden / live reference (CPU), GPU kernel vs oracle (gpu)."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

SLOW = H.golden()["slow"]
BE = ob.FLAG_BEGIN | ob.FLAG_END


def _bits(case):
    w = case["geometry"]["words"]
    return np.array([np.frombuffer(bytes.fromhex(h), dtype=np.uint32) for h in case["bits_hex"]]).reshape(-1, w)


@pytest.mark.parametrize("case", SLOW, ids=lambda c: c["name"])
def test_slow_oracle_matches_golden(case):
    o = ob.OracleSlowScanner(H.load_blob(case["blob"]))
    g = case["geometry"]
    assert (o.size, o.letters, o.words) == (g["states"], g["letters"], g["words"])
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    fin, bits = o.run_strings(strings)
    assert fin.tolist() == case["final"]
    assert (bits == _bits(case)).all()
    for f, want in zip(fin, case["ref_expect"]):
        assert bool(f) == want


@pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libpire_ref.so not built")
@pytest.mark.parametrize("pat,opt", [("a.{30}$", ""), ("x.{40}$", "u"), ("(ab|cd)*e.{3}f", ""), ("^a.{5}b", ""),
                                     ("[a-c]+d.{2,9}$", "i")])
def test_slow_oracle_vs_reference_random(pat, opt):
    r = ob.RefSlowScanner.compile(pat, opt)
    o = ob.OracleSlowScanner(r.save())
    assert (o.size, o.letters) == (r.size, r.letters)
    rng = np.random.RandomState(5)
    strings = H.random_strings(rng, 400, 150, b"ax.bcdef \xd0\xb0AB") + H.random_strings(rng, 100, 60) + [b""]
    for flags in (BE, 0, ob.FLAG_BEGIN, ob.FLAG_END):
        rf, rb = r.run_strings(strings, flags=flags)
        of, obits = o.run_strings(strings, flags=flags)
        assert (rf == of).all() and (rb == obits).all()


@pytest.mark.parametrize("case", SLOW, ids=lambda c: c["name"])
def test_slow_table_ingest_without_gpu(case):
    import pire_amd

    t = pire_amd.SlowTable(H.load_blob(case["blob"]))
    g = case["geometry"]
    assert (t.Size, t.LettersCount, t.words) == (g["states"], g["letters"], g["words"])
    blob = bytearray(H.load_blob(case["blob"]))
    blob[16] = 1      # Type = Scanner, not SlowScanner
    with pytest.raises(pire_amd.PireHipError):
        pire_amd.SlowTable(bytes(blob))
    with pytest.raises(pire_amd.PireHipError):
        pire_amd.SlowTable(bytes(H.load_blob(case["blob"])[:100]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", SLOW, ids=lambda c: c["name"])
def test_slow_gpu_matches_golden_and_oracle(case):
    import pire_amd

    blob = H.load_blob(case["blob"])
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    fin, bits, cnt = t.run_strings(strings, counts=True)
    assert fin.tolist() == case["final"]
    assert (bits == _bits(case)).all()
    assert cnt.tolist() == [sum(case["final"]), len(strings)]
    rng = np.random.RandomState(9)
    more = H.random_strings(rng, 3000, 200, b"ax.yd e\xd0\xb6bcx") + [b""] * 3 + H.random_strings(rng, 500, 100)
    for flags in (BE, 0, ob.FLAG_BEGIN, ob.FLAG_END):
        of, obits = o.run_strings(more, flags=flags)
        gf, gbits = t.run_strings(more, flags=flags)
        assert (gf == of).all() and (gbits == obits).all()


@pytest.mark.gpu
def test_slow_gpu_strided_device_batch():
    """Config 5b shape at a test size: x.{40}$ (UTF-8), fixed-length records resident on the device."""
    import torch
    import pire_amd

    case = [c for c in SLOW if c["name"] == "slow_x40_utf8"][0]
    blob = H.load_blob(case["blob"])
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    n, length = 20000, 512
    rng = np.random.RandomState(3)
    data = rng.choice(np.frombuffer(b"xyzw abc", dtype=np.uint8), size=(n, length)).astype(np.uint8)
    data[::3, length - 41] = ord("x")          # plant matches: x then 40 single-byte characters to the end
    of, obits = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length)
    d = torch.as_tensor(data, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    bits = torch.empty((n, t.words), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
    t.run_strided_device(d.data_ptr(), n, length, length, BE, fin.data_ptr(), bits.data_ptr(), cnt.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (fin.cpu().numpy() == of).all()
    assert (bits.cpu().numpy().astype(np.uint32) == obits).all()
    assert cnt.cpu().numpy().tolist() == [int(of.sum()), n]
    assert of.sum() >= n // 3


@pytest.mark.gpu
@pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libpire_ref.so not built")
@pytest.mark.parametrize("pat,opt,alphabet", [
    ("a.{30}$", "", b"ab"),                      # every second byte starts a chain: lists overflow at once
    ("x.{40}$", "u", b"xyz\xd0\xb0 "),            # UTF-8 automaton, dense triggers
    ("(ab|cd)*e.{3}f", "", b"abcdef"),           # rows with several targets
    ("(a|ab|abc|abcd|abcde)+f", "", b"abcdef"),   # many targets per row, sets that stay large
    ("[a-c]+d.{2,9}$", "i", b"abcdABCD. "),
    ("^a.{5}b", "", b"ab"),
    ("hello.{20}world", "", b"helowrd abcxyz0123456789"),   # sparse: the list form nearly all the time
])
def test_slow_gpu_list_and_bitset_forms_agree_with_the_oracle(pat, opt, alphabet):
    """The kernel keeps a lane's active set as a short list while it fits and as a bitset otherwise; dense and sparse
    triggers push lanes through both forms and back.  Final and the full state set must be the oracle's."""
    import pire_amd

    r = ob.RefSlowScanner.compile(pat, opt)
    blob = r.save()
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    rng = np.random.RandomState(31)
    strings = H.random_strings(rng, 3000, 200, alphabet) + [b"", b"a", alphabet * 40]
    for flags in (BE, 0, ob.FLAG_BEGIN, ob.FLAG_END):
        of, obits = o.run_strings(strings, flags=flags)
        gf, gb = t.run_strings(strings, flags=flags)
        assert (gf == of).all(), (pat, flags)
        assert (gb == obits).all(), (pat, flags)
