"""Randomly generated scanners (compiled by the unmodified reference, oracle/_ref) through every scan kernel: table
shapes the fixed fixtures do not cover -- few states, hundreds of letter classes (compact tier off), tables with heavy
traffic outside the 255 dense rows, UTF-8 and case-insensitive classes, glued sets of 2..5 patterns."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu

ATOMS = ["a", "b", "c", "ab", "ba", "abc", "[a-c]", "[^a]", ".", "\\d", "\\w", "\\s", "x", "[a-z]", "(ab|cd)", "[0-9a-f]",
         "ж", "[а-я]", "q+", "(xy)*"]
QUANT = ["", "", "", "*", "+", "?", "{2}", "{1,3}"]


def random_pattern(rng):
    parts = []
    for _ in range(rng.randint(1, 5)):
        parts.append(ATOMS[rng.randint(0, len(ATOMS))] + QUANT[rng.randint(0, len(QUANT))])
    pat = "".join(parts)
    if rng.randint(0, 4) == 0:
        pat = "^" + pat
    if rng.randint(0, 3) == 0:
        pat = pat + "$"
    return pat


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    assert pire_amd.device_count() > 0
    return pire_amd


@pytest.mark.parametrize("seed", range(24))
def test_random_scanner_all_kernels(pa, seed, cfg):
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    import torch
    from pire_amd import binding as pb

    rng = np.random.RandomState(1000 + seed)
    k = int(rng.randint(1, 6))
    opts = ["", "i", "u", "iu"][int(rng.randint(0, 4))]
    for attempt in range(20):
        pats = [random_pattern(rng) for _ in range(k)]
        try:
            ref = ob.RefScanner.compile(pats, [opts] * k)
        except ValueError:
            continue                      # too big to glue / not compilable: draw again
        if not ref.empty:
            break
    else:
        pytest.skip("no compilable set drawn")
    blob = ref.save()
    o, t = ob.OracleScanner(blob), pa.Table(blob)
    assert (t.Size, t.LettersCount, t.RegexpsCount) == (o.size, o.letters, o.regexps)
    alphabet = np.frombuffer("abcdxyq01 f\tжаZ".encode("utf-8"), dtype=np.uint8)
    # ragged batch (ragged kernel), with counts
    strings = [bytes(alphabet[rng.randint(0, len(alphabet), size=int(n))]) for n in rng.randint(0, 260, size=1500)]
    text, offs = H.pack(strings)
    for flags in (3, 0):
        oi, of = o.run(text, offs, flags=flags, threads=4)
        gi, gf, cnt = t.run(text, offs, flags=flags, counts=True)
        assert pb.last_kernel() == "ragged"
        assert (gi == oi).all() and (gf == of).all(), (pats, opts)
        assert cnt[0] == int(of.sum()) and cnt[1] == len(strings)
    # fixed-length records (tiled kernel) and the generic kernel on the same bytes
    n, length = 1024, 384
    data = alphabet[rng.randint(0, len(alphabet), size=(n, length))].astype(np.uint8)
    fo = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), fo, threads=4)
    gi, gf = t.run_strided_host(data)
    assert pb.last_kernel() == "tiled"
    assert (gi == oi).all() and (gf == of).all(), (pats, opts)
    d = torch.as_tensor(data, device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    t.run_strided_device(d.data_ptr(), n, length, length, 3 | pb.FLAG_GENERIC, idx.data_ptr(), fin.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize()
    assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()
    # adapt() must never change results
    t.adapt()
    gi, gf = t.run_strided_host(data)
    assert (gi == oi).all() and (gf == of).all()
    # prefix searches
    lp, gp = o.prefix(text, offs, True), t.prefix(text, offs, True)
    assert (lp == gp).all()
    sp, gs = o.prefix(text, offs, False), t.prefix(text, offs, False)
    assert (sp == gs).all()
    # the segmented scan (few long strings), forced onto longer strings of the same alphabet with segments and
    # warm-ups small enough for the guesses to fail often; and the half-final counting where the table allows it
    long_strings = [bytes(alphabet[rng.randint(0, len(alphabet), size=int(n))]) for n in rng.randint(0, 5000, size=40)]
    ltext, loffs = H.pack(long_strings)
    oi, of = o.run(ltext, loffs, threads=4)
    cfg.set(segment_bytes=str(int(rng.choice([48, 128, 400]))))
    cfg.set(segment_warmup=str(int(rng.choice([0, 8, 64]))))
    gi, gf, cnt = t.run(ltext, loffs, counts=True)
    assert pb.last_kernel().startswith("segmented")
    assert (gi == oi).all() and (gf == of).all(), (pats, opts)
    assert cnt[0] == int(of.sum()) and cnt[1] == len(long_strings)
    cfg.unset("segment_bytes")
    if t.RegexpsCount <= 8:
        hi, hf, hr = o.run_half_final(text, offs)
        gi, gf, gr = t.run_half_final(text, offs)
        assert (gi == hi).all() and (gf == hf).all() and (gr == hr).all(), (pats, opts)


def test_many_letter_classes_disable_the_compact_tier(pa):
    """More than 127 letter classes: the compact tier is off (its per-byte class table holds 2*class in a u8) and the
    exact re-walk goes straight to the full table; more states than dense rows, so cold traffic is certain."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    from pire_amd import binding as pb

    # 138 symbols with pairwise different behaviour: 64 Cyrillic letters (64 different UTF-8 continuation bytes) and 74
    # ASCII bytes; symbol i must be followed by the 8-bit code of i written in 'a'/'b', then 'z'
    symbols = [chr(0x410 + k) for k in range(64)] + list("0123456789ABCDEFGHIJKLMNOPQRSTUVWXYcdefghijklmnopqrstuvwxy_,;:!@#%=<>/' ")
    assert len(symbols) > 130 and len(set(symbols)) == len(symbols)

    def word(i):
        return symbols[i] + "".join("ab"[(i >> k) & 1] for k in range(8)) + "z"

    ref = ob.RefScanner.compile([("(" + "|".join(word(i)[:-1] for i in range(len(symbols))) + ")z").encode("utf-8")], ["u"])
    blob = ref.save()
    o, t = ob.OracleScanner(blob), pa.Table(blob)
    assert t.LettersCount > 127 and t.info.compact_states == 0 and t.Size > 255
    rng = np.random.RandomState(4)
    strings = []
    for _ in range(3000):
        parts = []
        for _ in range(rng.randint(0, 12)):
            i = rng.randint(0, len(symbols))
            parts.append(word(i).encode("utf-8") if rng.randint(0, 3)
                         else bytes(rng.randint(0, 256, size=rng.randint(1, 9), dtype=np.uint8)))
        strings.append(b"".join(parts))
    text, offs = H.pack(strings)
    oi, of = o.run(text, offs, threads=4)
    gi, gf = t.run(text, offs)
    assert pb.last_kernel() == "ragged"
    assert (gi == oi).all() and (gf == of).all() and 0 < of.sum() < len(strings)
    n, length = 512, 256
    data = np.frombuffer(text.tobytes()[:n * length].ljust(n * length, b"a"), dtype=np.uint8).reshape(n, length)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    gi, gf = t.run_strided_host(data)
    assert pb.last_kernel() == "tiled"
    assert (gi == oi).all() and (gf == of).all()
