"""Walks with actions on the class-indexed walk (round 6; ragged.hip WideChunkAct): LongestPrefix /
ShortestPrefix (run.h:277-311) and the HalfFinalScanner counting (half_final.h:137-164) on tables whose scans visit thousands
of states -- the wide rows instead of the 255 dense ones, plain and zipped images -- against the oracle and the unmodified
reference."""
import numpy as np
import pytest

from oracle import binding as ob
from pire_amd import workloads as W
from tests.test_gpu_parity import pa, torch_cuda  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _lines(entry, corpus, seed, n):
    """Records of the corpus cut into strings of 0..700 bytes at any alignment, empty ones and tiny ones among them."""
    rng = np.random.RandomState(seed)
    lens = rng.randint(0, 700, size=n).astype(np.uint64)
    lens[rng.randint(0, n, size=n // 16)] = rng.randint(0, 16, size=n // 16)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    text = W.wide_records(entry, corpus, seed, (total + 1023) // 1024 + 1, 1024).reshape(-1)[:total].copy()
    # the corpora hold no whole word of the dictionary: one written into every 7th line that has room (behind it a Surround()ed
    # scanner is in ONE Final state for good -- PrefixAct's short cut -- and every chunk is one "with a Final state")
    words = W.dictionary_words(entry)
    for i in range(0, n, 7):
        w = words[rng.randint(0, len(words))]
        if int(lens[i]) > len(w):
            at = int(offs[i]) + rng.randint(0, int(lens[i]) - len(w))
            text[at:at + len(w)] = np.frombuffer(w, dtype=np.uint8)
    return text, offs


CASES = [
    # scanner, corpus ("urls" or a records corpus), zip_variant (1 plain rows, 2 zipped)
    ("blacklist_1k", "urls", 1),
    ("blacklist_1k", "urls", 2),
    ("blacklist_10k", "urls", 2),
    ("dict_1k", "k128", 1),
    ("dict_1k", "k1000", 2),
    ("dict_10k", "k10000", 1),
    ("dict_10k", "k10000", 2),
    ("dict_utf8_1k", "k1000", 2),
]


@pytest.mark.parametrize("name,corpus,zipv", CASES, ids=lambda v: str(v))
def test_prefix_searches_and_half_final_counting_on_the_wide_walk(pa, torch_cuda, cfg, name, corpus, zipv):
    from pire_amd import binding as pb

    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    o = ob.OracleScanner(blob)
    ref = ob.RefScanner.load(blob) if ob.ref_available() else None
    if corpus == "urls":
        text, offs = W.wide_urls(entry, 9, 20000)
        lens = np.diff(offs).astype(np.int64)
        lens[::89] = 0
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        text = text[:int(offs[-1])]
    else:
        text, offs = _lines(entry, corpus, 17, 6000)
    cfg.set(walk_variant=2, zip_variant=zipv, ragged_act_always=1, auto_adapt=1)
    t = pa.Table(blob)
    for round_ in range(2):
        for longest in (True, False):
            for tb, te in ((True, True), (True, False), (False, False)):
                want = o.prefix(text, offs, longest, tb, te)
                got = t.prefix(text, offs, longest, tb, te)
                assert pb.last_kernel() == "ragged_prefix_wide", pb.last_kernel()
                assert ("zipped" in pb.last_kernel_symbol()) == (zipv == 2), pb.last_kernel_symbol()
                bad = np.nonzero(got != want)[0]
                assert len(bad) == 0, (round_, longest, tb, te, len(bad), bad[:8], got[bad[:8]], want[bad[:8]])
                if ref is not None and round_ == 0:
                    assert (ref.prefix(text, offs, longest, tb, te) == got).all()
        for flags in (ob.FLAG_BEGIN | ob.FLAG_END, 0):
            oi, of, orr = o.run_half_final(text, offs, flags=flags)
            gi, gf, gr = t.run_half_final(text, offs, flags=flags)
            assert pb.last_kernel() == "ragged_half_final_wide", pb.last_kernel()
            assert (gi == oi).all() and (gf == of).all() and (gr == orr).all(), (round_, flags)
        # a plain scan of the same text, then the rows ranked from what all these walks saw: the second round on the new image
        t.run(text, offs)
        t.adapt()
    assert t.refresh_info().shares_measured


def test_the_walks_with_actions_follow_the_table_to_the_wide_walk(pa, torch_cuda, cfg):
    """Routing under the defaults: a dictionary scanner's first prefix call takes the dense rows (nothing known yet); once the
    table has seen its scans leave them (adapt()), the same call takes the wide rows; walk_variant = 1 keeps the dense ones."""
    from pire_amd import binding as pb

    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    o = ob.OracleScanner(blob)
    text, offs = _lines(entry, "k128", 5, 6000)
    want = o.prefix(text, offs, True, True, True)
    cfg.set(ragged_act_always=1, auto_adapt=1, walk_variant=0, zip_variant=0)
    t = pa.Table(blob)
    assert (t.prefix(text, offs, True, True, True) == want).all()
    first = pb.last_kernel()
    for _ in range(3):
        t.run(text, offs)
        t.adapt()
    assert (t.prefix(text, offs, True, True, True) == want).all()
    assert first == "ragged_prefix" and pb.last_kernel() == "ragged_prefix_wide", (first, pb.last_kernel())
    cfg.set(walk_variant=1)
    assert (t.prefix(text, offs, True, True, True) == want).all()
    assert pb.last_kernel() == "ragged_prefix"
    assert (want >= 0).mean() > 0.05 and (want == np.diff(offs).astype(np.int64))[want >= 0].all()   # behind a match: Final for good
