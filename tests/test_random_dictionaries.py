"""Randomly drawn dictionaries (compiled by the unmodified reference, oracle/_ref, the way samples/blacklist/blacklist.cpp:65-76
builds its scanner) through every kernel of the class-indexed walk: plain rows and zipped image, one and two strings per lane,
the ragged kernel and the stream kernel on the walk, the prefix searches and the half-final counting on it, before and after the
table has ranked its rows from the scans -- table shapes and texts the fixed fixtures do not hold (round 6: the one wrong-result
bug of round 5's kernels was found by a corpus nobody had tried).  tools/stress_dict.py runs the same over hundreds of seeds."""
import numpy as np
import pytest

from oracle import binding as ob
from tests.test_gpu_parity import dev_run_strided, expected_counts, pa, stream_lengths, torch_cuda  # noqa: F401  (fixtures)
from tests.test_wide import dev_run_offsets

pytestmark = pytest.mark.gpu

BE = ob.FLAG_BEGIN | ob.FLAG_END
ALPHABETS = [b"abcdefghijklmnopqrstuvwxyz", b"abcdefghijklmnopqrstuvwxyz0123456789-_", b"abcdefgh", b"ACGT",
             "абвгдежзиклмнопрстуфхцчшщэюя".encode("utf-8")]


def draw_dictionary(rng):
    """(words, mode): 150..2500 words of 3..12 symbols over one of the alphabets; mode 0 blacklist wrapping, 1 Surround(), 2 UTF-8."""
    a = ALPHABETS[int(rng.randint(0, len(ALPHABETS)))]
    utf8 = a[0] >= 0x80
    symbols = [a[i:i + 2] for i in range(0, len(a), 2)] if utf8 else [a[i:i + 1] for i in range(len(a))]
    n = int(rng.choice([150, 400, 900, 2500]))
    n = min(n, len(symbols) ** 3 // 2)
    words = set()
    while len(words) < n:
        words.add(b"".join(symbols[int(k)] for k in rng.randint(0, len(symbols), size=int(rng.randint(3, 13)))))
    return sorted(words), symbols, (2 if utf8 else int(rng.randint(0, 2)))


def draw_text(rng, words, symbols, total):
    """Text that walks the dictionary's trie: whole words, prefixes of words, random symbols, separators."""
    out = bytearray()
    seps = [b" ", b"/", b".", b"\n", b"=", b"http://", b"www."]
    while len(out) < total:
        r = rng.randint(0, 10)
        w = words[int(rng.randint(0, len(words)))]
        if r < 1:
            out += w
        elif r < 6:
            out += w[:int(rng.randint(1, len(w) + 1))]
        elif r < 8:
            out += b"".join(symbols[int(k)] for k in rng.randint(0, len(symbols), size=int(rng.randint(1, 6))))
        out += seps[int(rng.randint(0, len(seps)))] if rng.randint(0, 3) else b""
    return np.frombuffer(bytes(out[:total]), dtype=np.uint8).copy()


def run_seed(pa, torch, cfg, seed, verbose=False):
    from pire_amd import binding as pb

    def say(*a):
        if verbose:
            print(*a, flush=True)

    rng = np.random.RandomState(7000 + seed)
    words, symbols, mode = draw_dictionary(rng)
    ref = ob.RefScanner.compile_dictionary(words, surround=(mode == 1), utf8=(mode == 2))
    blob = ref.save()
    o = ob.OracleScanner(blob)
    if o.size <= 300 or o.letters > 127:
        return "skipped"
    what = (seed, len(words), mode, o.size, o.letters)
    say(what)
    # ---- fixed-length records: both wide kernels, both images
    n, length = 2048, int(rng.choice([384, 1024, 1152]))
    data = draw_text(rng, words, symbols, n * length).reshape(n, length)
    fo = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), fo, threads=4)
    d = torch.as_tensor(data, device="cuda")
    # ---- offset batches
    kind = ["urls", "tiny", "mixed", "lines", "edges", "aligned"][int(rng.randint(0, 6))]
    m = int(rng.choice([300, 5000, 20000]))
    ln = stream_lengths(rng, kind, m).astype(np.uint64)
    offs = np.zeros(m + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(ln)
    text = draw_text(rng, words, symbols, max(int(offs[-1]), 1))[:int(offs[-1])]
    roi, rof = o.run(text, offs, threads=4)
    want_prefix = {(lg, tb): o.prefix(text, offs, lg, tb, tb) for lg in (True, False) for tb in (True, False)}
    hi, hf, hr = o.run_half_final(text, offs)
    for zipv in (1, 2):
        cfg.set(zip_variant=zipv, auto_adapt=1, no_offsets_peek=1, ragged_act_always=1)
        t = pa.Table(blob)
        for round_ in range(2):
            for walk in (2, 3):
                cfg.set(walk_variant=walk)
                say("zip", zipv, "round", round_, "strided walk", walk, n, length)
                gi, gf, cnt = dev_run_strided(torch, t, d)
                assert pb.last_kernel() == "wide", (what, pb.last_kernel())
                assert (gi == oi).all() and (gf == of).all(), (what, zipv, round_, walk, "strided")
                assert (cnt == expected_counts(o, oi, of)).all()
            cfg.set(walk_variant=2)
            for raggedv in (1, 2):
                cfg.set(ragged_variant=raggedv)
                if round_ == 0 and raggedv == 2:
                    t.upload()   # (the stream kernel's image is built for tables uploaded while it is asked for)
                say("offsets, ragged_variant", raggedv, kind, m)
                gi, gf, cnt = dev_run_offsets(torch, t, text, offs)
                assert pb.last_kernel() in ("ragged_wide", "stream_wide", "generic"), (what, pb.last_kernel())
                assert (gi == roi).all() and (gf == rof).all(), (what, zipv, round_, raggedv, kind, m, pb.last_kernel())
                assert (cnt == expected_counts(o, roi, rof)).all()
            cfg.set(ragged_variant=0)
            for (lg, tb), want in want_prefix.items():
                say("prefix", lg, tb)
                got = t.prefix(text, offs, lg, tb, tb)
                assert (got == want).all(), (what, zipv, round_, "prefix", lg, tb, pb.last_kernel())
            if t.RegexpsCount <= 8:
                say("half-final")
                gi, gf, gr = t.run_half_final(text, offs)
                assert (gi == hi).all() and (gf == hf).all() and (gr == hr).all(), (what, zipv, round_, "half-final", pb.last_kernel())
            t.adapt()   # the second round: rows ranked from what these scans saw
    return "ok"


@pytest.mark.parametrize("seed", range(10))
def test_random_dictionary_through_every_kernel_of_the_wide_walk(pa, torch_cuda, cfg, seed):
    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    if run_seed(pa, torch_cuda, cfg, seed) == "skipped":
        pytest.skip("the dictionary drawn compiles to a table the wide walk is not for")


def test_the_ragged_kernel_waits_for_its_last_window_before_it_leaves(pa, torch_cuda, cfg):
    """Found by tools/stress_dict.py (round 6, seed 206): ScanRaggedKernel left its window loop with the (dummy) loads of the
    next window still on their way; the epilogue's barrier waits for LDS only, and a line that landed late overwrote a register
    of the counter flush -- a memory fault (or an atomic add somewhere) once in a few hundred launches, whenever a wave's LAST
    iteration has nothing to walk: prefix searches whose strings all die at their first byte (a blacklist scanner walked
    without BeginMark), batches that end in empty strings.  Every instantiation since round 2 had the window; the kernel now
    waits on its way out (ragged.hip RaggedPhase).  Hundreds of such launches, every answer the oracle's, no fault."""
    from pire_amd import binding as pb

    if not ob.ref_available():
        pytest.skip("oracle/_ref not built")
    torch = torch_cuda
    rng = np.random.RandomState(7000 + 206)
    words, symbols, mode = draw_dictionary(rng)
    assert mode == 0
    blob = ob.RefScanner.compile_dictionary(words, surround=False, utf8=False).save()
    o = ob.OracleScanner(blob)
    ln = stream_lengths(np.random.RandomState(3), "edges", 20000).astype(np.uint64)
    offs = np.zeros(len(ln) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(ln)
    text = draw_text(rng, words, symbols, int(offs[-1]))
    want = o.prefix(text, offs, False, False, False)
    assert (want < 0).all()                      # dead at the first byte, every one of them
    empty = np.zeros(5001, dtype=np.uint64)      # 5 000 empty strings
    eoi, eof = o.run(text[:0], empty, threads=1)
    for walk in (2, 1):
        cfg.set(walk_variant=walk, zip_variant=1, auto_adapt=1, ragged_act_always=1, no_offsets_peek=1, ragged_variant=1)
        t = pa.Table(blob)
        for _ in range(150):
            assert (t.prefix(text, offs, False, False, False) == want).all()
        assert pb.last_kernel() == ("ragged_prefix_wide" if walk == 2 else "ragged_prefix")
        for _ in range(150):
            gi, gf, _ = dev_run_offsets(torch, t, np.zeros(16, dtype=np.uint8), empty)
            assert (gi == eoi).all() and (gf == eof).all()
        assert pb.last_kernel() in ("ragged_wide", "ragged")
