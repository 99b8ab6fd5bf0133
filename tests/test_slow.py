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
def test_slow_gpu_list_and_bitset_forms_agree_with_the_oracle(pat, opt, alphabet, cfg):
    """A lane's active set is a list of states while it fits -- 16 slots in the list kernel that every batch takes
    first (round 3), 4 slots in the bitset kernel that takes the strings the list kernel gave up on (and every string
    with pire_hip_config.slow_no_list) -- and a bitset otherwise; dense and sparse triggers push strings and lanes
    through all forms.  Final and the full state set must be the oracle's, either way."""
    import pire_amd
    from pire_amd import binding as pb

    r = ob.RefSlowScanner.compile(pat, opt)
    blob = r.save()
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    rng = np.random.RandomState(31)
    strings = H.random_strings(rng, 3000, 200, alphabet) + [b"", b"a", alphabet * 40]
    for no_list in (0, 1):
        cfg.set(slow_no_list=no_list)
        for flags in (BE, 0, ob.FLAG_BEGIN, ob.FLAG_END):
            of, obits = o.run_strings(strings, flags=flags)
            gf, gb, cnt = t.run_strings(strings, flags=flags, counts=True)
            assert pb.last_kernel() == ("slow" if no_list else "slow_list")
            assert (gf == of).all(), (pat, flags, no_list)
            assert (gb == obits).all(), (pat, flags, no_list)
            assert cnt.tolist() == [int(of.sum()), len(strings)]


# ---- more than 256 NFA states: the wave-per-string form (slow.hip SlowWideKernel) ------------------------------------
import hashlib
import json
import os


def _wide_cases():
    with open(os.path.join(H.GOLDEN, "slow_wide.json")) as f:
        return json.load(f)["slow_wide"]


WIDE = _wide_cases()


@pytest.mark.parametrize("case", WIDE, ids=lambda c: c["name"])
def test_slow_wide_oracle_matches_golden(case):
    """The oracle restates slow.h for any number of states; the fixtures come from the unmodified reference."""
    o = ob.OracleSlowScanner(H.load_blob(case["blob"]))
    g = case["geometry"]
    assert (o.size, o.letters, o.words) == (g["states"], g["letters"], g["words"]) and o.size > 256
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    fin, bits = o.run_strings(strings)
    assert fin.tolist() == case["final"]
    assert [hashlib.sha256(bytes(np.ascontiguousarray(b))).hexdigest() for b in bits] == case["bits_sha256"]
    for f, want in zip(fin, case["ref_expect"]):
        assert bool(f) == want


@pytest.mark.parametrize("case", WIDE, ids=lambda c: c["name"])
def test_slow_wide_table_ingest_without_gpu(case):
    import pire_amd

    t = pire_amd.SlowTable(H.load_blob(case["blob"]))
    g = case["geometry"]
    assert (t.Size, t.LettersCount, t.words) == (g["states"], g["letters"], g["words"])
    blob = bytearray(H.load_blob(case["blob"]))
    blob[-3] = blob[-7] = 0xFF      # the last jump target (or the padding word behind it and the one before): far outside
    with pytest.raises(pire_amd.PireHipError):
        pire_amd.SlowTable(bytes(blob))


@pytest.mark.gpu
@pytest.mark.parametrize("no_list", [0, 1], ids=["list_first", "wave_per_string"])
@pytest.mark.parametrize("case", WIDE, ids=lambda c: c["name"])
def test_slow_wide_gpu_matches_golden_and_oracle(case, no_list, cfg):
    """Both forms for automata of more than 256 states: the list kernel with the wave-per-string kernel behind it for
    the strings whose sets outgrow 16 slots (the default), and the wave-per-string kernel alone
    (pire_hip_config.slow_no_list)."""
    import pire_amd
    from pire_amd import binding as pb

    cfg.set(slow_no_list=no_list)
    blob = H.load_blob(case["blob"])
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    strings = [bytes.fromhex(h) for h in case["strings_hex"]]
    fin, bits, cnt = t.run_strings(strings, counts=True)
    assert pb.last_kernel() == ("slow_wide" if no_list else "slow_list")
    assert fin.tolist() == case["final"]
    assert [hashlib.sha256(bytes(np.ascontiguousarray(b))).hexdigest() for b in bits] == case["bits_sha256"]
    assert cnt.tolist() == [sum(case["final"]), len(strings)]
    rng = np.random.RandomState(19)
    more = H.random_strings(rng, 600, 900, b"ax.yd ef\xd0\xb6bcx") + [b""] * 3 + H.random_strings(rng, 100, 100)
    for flags in (BE, 0, ob.FLAG_BEGIN, ob.FLAG_END):
        of, obits = o.run_strings(more, flags=flags)
        gf, gbits = t.run_strings(more, flags=flags)
        assert (gf == of).all() and (gbits == obits).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", WIDE, ids=lambda c: c["name"])
def test_slow_list_kernel_hands_crowded_strings_to_the_wide_kernel(case):
    """The list kernel keeps a lane's set in 16 slots.  Text with one trigger letter in every few bytes has dozens of
    threads alive (x.{300}: one per 'x' of the last 300 bytes), text with one in a hundred a handful: a batch of
    both -- sparse strings stay in the list form, crowded ones are walked by the wave-per-string kernel from their
    start -- must come out as the oracle says, string by string, counters included, with offsets that are not
    multiples of 16."""
    import pire_amd

    blob = H.load_blob(case["blob"])
    t, o = pire_amd.SlowTable(blob), ob.OracleSlowScanner(blob)
    rng = np.random.RandomState(23)
    sparse = np.frombuffer(b"x" + b"abcdefghijklmnopqrstuvwyz ABCDEFGHIJKLMNOPQRSTUVWYZ0123456789.,;:-_" * 2, dtype=np.uint8)
    crowded = np.frombuffer(b"xxxyab(cd)ef", dtype=np.uint8)
    strings = []
    for i in range(700):
        alpha = crowded if i % 5 == 0 else sparse
        strings.append(bytes(alpha[rng.randint(0, len(alpha), size=int(rng.randint(0, 1400)))]))
    strings += [b"", b"x", b"x" * 700, b"zx" + b"y" * 299, b"zx" + b"y" * 300, b"zx" + b"y" * 301]
    of, obits = o.run_strings(strings)
    gf, gbits, cnt = t.run_strings(strings, counts=True)
    assert (gf == of).all()
    assert (gbits == obits).all()
    assert cnt.tolist() == [int(of.sum()), len(strings)]


@pytest.mark.gpu
@pytest.mark.skipif(not ob.ref_available(), reason="oracle/_ref/libpire_ref.so not built")
def test_slow_wide_gpu_sets_in_device_memory():
    """x.{6000}$ has 12 007 states (two sets of 376 words per wave); the UTF-8 x.{1500}$ 7 507.  Both run with the sets
    in LDS and -- forced with pire_hip_config.slow_sets_in_memory -- with the sets in device memory, the form
    automata too large for the LDS take.  All against the oracle on strings around the gap length."""
    import pire_amd
    from pire_amd import binding as pb

    for pat, opt, gap in (("x.{6000}$", "", 6000), ("x.{1500}$", "u", 1500)):
        r = ob.RefSlowScanner.compile(pat, opt)
        blob = r.save()
        o = ob.OracleSlowScanner(blob)
        rng = np.random.RandomState(5)
        strings = [b"zx" + b"y" * k for k in (gap - 1, gap, gap + 1)]
        strings += [b"x" + bytes(rng.choice(np.frombuffer(b"xyz", dtype=np.uint8), size=k)) for k in (gap, 17)]
        strings += [bytes(rng.choice(np.frombuffer(b"xy", dtype=np.uint8), size=gap + 40)) for _ in range(6)] + [b""]
        of, obits = o.run_strings(strings)
        for knob in (0, 1):
            with pb.config(slow_sets_in_memory=knob, slow_no_list=1):
                t = pire_amd.SlowTable(blob)
                gf, gb = t.run_strings(strings)
            assert (gf == of).all() and (gb == obits).all(), (pat, knob)
        assert of[1] == 1 and of[0] == 0 and of[2] == 0
