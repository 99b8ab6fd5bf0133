"""HalfFinalScanner counting and the prefix searches through the ragged kernel (ragged.hip, "scans with actions"):
the fast path only notices that a 16-byte chunk touched a Final state (or left the dense rows) and re-walks exactly
those chunks with the action.  Checked against the oracle AND against the one-string-per-lane kernels of exact.hip
(PIRE_HIP_RUN_GENERIC), on length mixes that hit every window shape of the ragged kernel: empty strings, < 16 bytes,
exact multiples of 16 and of 128, long strings, unaligned starts, the last bytes of the buffer."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def always_ragged(cfg):
    """The prefix searches pick the one-string-per-lane kernel for scanners whose searches end after a few bytes
    (a heuristic, exact.hip LaunchPrefix); these tests are about the ragged kernel, whatever the scanner."""
    cfg.set(ragged_act_always="1")


def length_mix(rng, n, alphabet, long_every=97):
    lens = []
    special = [0, 1, 2, 15, 16, 17, 31, 32, 33, 112, 113, 127, 128, 129, 143, 144, 145, 255, 256, 257, 1000]
    for i in range(n):
        if i % 7 == 0:
            lens.append(special[(i // 7) % len(special)])
        elif i % long_every == 0:
            lens.append(int(rng.randint(1500, 6000)))
        else:
            lens.append(int(rng.randint(0, 300)))
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return [a[rng.randint(0, len(a), size=k)].tobytes() for k in lens]


def half_tables():
    g = H.golden()
    out = [(c["name"], H.load_blob(c["blob"]), b"abcde w") for c in g["half_final"] if c["regexps"] <= 8][:4]
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    out.append(("set_a", H.load_blob(big["blob"]), b"ABCDEFGHIJKLMNOPQRSTUVWXYZ hello wd0123456789-() @net"))
    return out


@pytest.mark.parametrize("name,blob,alphabet", half_tables(), ids=[t[0] for t in half_tables()])
def test_ragged_half_final_matches_oracle_and_exact_kernel(name, blob, alphabet):
    import pire_amd
    from pire_amd import binding as pb

    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(101)
    strings = length_mix(rng, 3000, alphabet)
    text, offs = H.pack(strings)
    for flags in (3, 0, 1, 2):
        oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=flags)
        gi, gf, gr = t.run_half_final(text, offs, flags=flags)
        assert pb.last_kernel() == "ragged_half_final"
        assert (gi == oi).all() and (gf == of).all(), (name, flags)
        assert (gr == orr).all(), (name, flags, np.nonzero((gr != orr).any(axis=1))[0][:5])
        xi, xf, xr = t.run_half_final(text, offs, flags=flags | pb.FLAG_GENERIC)
        assert pb.last_kernel() == "half_final"
        assert (xi == oi).all() and (xf == of).all() and (xr == orr).all()
    assert orr.sum() > 0
    # after adapt() (other dense rows, other hotFinalLo) the answers are the same
    t.run(text, offs)
    t.adapt()
    gi, gf, gr = t.run_half_final(text, offs, flags=3)
    oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=3)
    assert (gi == oi).all() and (gf == of).all() and (gr == orr).all()
    # below 256 strings: the exact kernel
    gi, gf, gr = t.run_half_final(*H.pack(strings[:100]), flags=3)
    assert pb.last_kernel() == "half_final"
    assert (gr == orr[:100]).all()


def prefix_tables():
    out = []
    for name in ("survey_known_answer", "inline_glue3", "rep_dot_3_10", "set_d", "set_a"):
        c = [x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
        out.append((name, H.load_blob(c["blob"])))
    return out


@pytest.mark.parametrize("name,blob", prefix_tables(), ids=[t[0] for t in prefix_tables()])
def test_ragged_prefix_matches_oracle_and_exact_kernel(name, blob):
    import pire_amd
    from pire_amd import binding as pb

    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(55)
    strings = length_mix(rng, 2500, b"abcdefhelo wrdxHTailnI0123 \t/.:fbABCXYZ@-()", long_every=61)
    strings += [b"hello  world", b"say hello   wod and more", b"aaab", b"xxabc", b""]
    text, offs = H.pack(strings)
    for longest in (True, False):
        for tb, te in ((False, False), (True, False), (False, True), (True, True)):
            want = o.prefix(text, offs, longest, tb, te)
            got = t.prefix(text, offs, longest, tb, te)
            assert pb.last_kernel() == "ragged_prefix"
            assert (got == want).all(), (name, longest, tb, te, np.nonzero(got != want)[0][:5])
            ex = t.prefix(text, offs, longest, tb, te, generic=True)
            assert pb.last_kernel() == "prefix"
            assert (ex == want).all()
    assert (want >= 0).any() or name == "set_a"


def test_ragged_prefix_plain_scanners_and_dead_states():
    """Unsurrounded scanners (the way lexers use prefix searches): Dead states end the search, Final states are dense."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref/libpire_ref.so not built")
    import pire_amd

    rng = np.random.RandomState(9)
    for pats in (["a+b"], ["[a-c]+"], ["abc|abcabc|b+"], ["a*"], ["(ab)*c?"]):
        r = ob.RefScanner.compile(pats, ["n"] * len(pats))
        blob = r.save()
        t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
        strings = length_mix(rng, 1200, b"abc", long_every=53) + [b"a" * 700 + b"b", b"ab" * 400 + b"c", b"c" * 300]
        text, offs = H.pack(strings)
        for longest in (True, False):
            want = o.prefix(text, offs, longest)
            assert (t.prefix(text, offs, longest) == want).all(), (pats, longest)
            assert (r.prefix(text, offs, longest) == want).all()


def test_ragged_actions_on_device_buffers_of_exact_size():
    """Device pointers, the text allocation ends with the last string: nothing may be read past its last 16-byte block."""
    import torch
    import pire_amd

    g = H.golden()
    c = [c for c in g["half_final"] if c["regexps"] <= 8][0]
    blob = H.load_blob(c["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(77)
    for tail in (1, 15, 16, 100, 127, 129):
        strings = length_mix(rng, 700, b"abcde w") + [b"ab" * (tail // 2) + b"a" * (tail % 2)]
        text, offs = H.pack(strings)
        d = torch.as_tensor(np.array(text), device="cuda")
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        n = len(strings)
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        fin = torch.empty(n, dtype=torch.uint8, device="cuda")
        res = torch.empty((n, t.RegexpsCount), dtype=torch.int32, device="cuda")
        ln = torch.empty(n, dtype=torch.int64, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        t.run_half_final_device(d.data_ptr(), do.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), res.data_ptr(), s)
        t.prefix_device(d.data_ptr(), do.data_ptr(), n, True, ln.data_ptr(), stream=s)
        torch.cuda.synchronize()
        oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=3)
        assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (res.cpu().numpy().astype(np.uint32) == orr).all()
        assert (ln.cpu().numpy() == o.prefix(text, offs, True)).all()


def test_prefix_kernel_choice_follows_the_scanner(cfg):
    """A lexer-like scanner (Dead right behind the token) keeps the exact kernel, a Surround()ed one takes the ragged."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref/libpire_ref.so not built")
    import pire_amd
    from pire_amd import binding as pb

    cfg.unset("ragged_act_always")
    rng = np.random.RandomState(5)
    strings = length_mix(rng, 600, b"abc 019")
    text, offs = H.pack(strings)
    for pats, opts, kernel in ((["[a-z]+|[0-9]+| +"], ["n"], "prefix"), (["b+c"], [""], "ragged_prefix")):
        blob = ob.RefScanner.compile(pats, opts).save()
        t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
        got = t.prefix(text, offs, True)
        assert pb.last_kernel() == kernel
        assert (got == o.prefix(text, offs, True)).all()


def test_ragged_half_final_with_counters_that_do_not_pack():
    """More than 8 regexps: the lane read-modify-writes its row of the result array in the exact re-walks."""
    if not ob.ref_available():
        pytest.skip("oracle/_ref/libpire_ref.so not built")
    import pire_amd
    from pire_amd import binding as pb

    pats = ["ab+", "b", "a", "(ab)+", "c", "bc", "abc", "b+", "[ab]c", "ca", "d+"]
    blob = ob.RefHalfFinalScanner.compile(pats, [ob.RefHalfFinalScanner.NONGREEDY_SIMPLE] * len(pats)).save()
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(3)
    for alphabet in (b"abcd", b"abcdefghijklmnopqrstuvwxyz    "):
        strings = length_mix(rng, 3000, alphabet)
        text, offs = H.pack(strings)
        for flags in (3, 0):
            oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=flags)
            gi, gf, gr = t.run_half_final(text, offs, flags=flags)
            assert pb.last_kernel() == "ragged_half_final"
            assert (gi == oi).all() and (gf == of).all() and (gr == orr).all()
            xi, xf, xr = t.run_half_final(text, offs, flags=flags | pb.FLAG_GENERIC)
            assert (xr == orr).all()
        assert orr.sum() > 0
