"""The class-indexed walk (pire_amd/csrc/wide.hip, round 5): tables whose scans visit thousands of states.

CPU part: the fixtures of tests/golden/wide.json (dictionary scanners compiled by the unmodified reference the way
samples/blacklist/blacklist.cpp:65-76 builds one) pin the oracle, and the walk's LDS image is walked on the host exactly
as the kernel walks it -- rows for the first states of the ranking, the escape row, the exact table behind it.
GPU part (-m gpu): the kernel against the oracle and the recorded reference results, through the C ABI."""
import hashlib

import numpy as np
import pytest

from oracle import binding as ob
from pire_amd import workloads as W
from tests import helpers as H
from tests.test_gpu_parity import dev_run_strided, expected_counts, pa, torch_cuda  # noqa: F401  (fixtures)

BE = ob.FLAG_BEGIN | ob.FLAG_END
WIDE = [(w["name"], c) for w in W.wide_sets() for c in w["samples"]]


def sample_of(name, corpus):
    entry = W.wide_set(name)
    s = entry["samples"][corpus]
    if corpus == "urls":
        text, offs = W.wide_urls(entry, s["seed"], s["n"])
    else:
        text = W.wide_records(entry, corpus, s["seed"], s["n"], s["len"]).reshape(-1)
        offs = np.arange(s["n"] + 1, dtype=np.uint64) * s["len"]
    assert hashlib.sha256(text.tobytes()).hexdigest() == s["sha256"], "the corpus generator no longer builds the recorded bytes"
    return entry, s, text, offs


@pytest.mark.parametrize("name,corpus", WIDE)
def test_oracle_against_the_recorded_reference_results(name, corpus):
    """oracle/pire_oracle.c on the wide corpora == what the compiled reference answered when the fixture was made."""
    entry, s, text, offs = sample_of(name, corpus)
    blob = W.load_blob(entry["blob"])
    assert hashlib.sha256(blob).hexdigest() == entry["blob_sha256"]
    o = ob.OracleScanner(blob)
    assert (o.size, o.letters) == (entry["geometry"]["states"], entry["geometry"]["letters"])
    idx, fin = o.run(text, offs, threads=4)
    assert idx.tolist() == s["idx"] and fin.tolist() == s["final"]
    visits = o.visit_counts(text, offs)
    assert int((visits > 0).sum()) == s["distinct_states_visited"]
    if ob.ref_available():
        ri, rf = ob.RefScanner.load(blob).run(text, offs)
        assert (ri == idx).all() and (rf == fin).all()


def walk_image(t, strings):
    """The kernel's walk on the host (wide.hip WideChunk / WideTrapChunk): device ids through the rows of the image while
    the state has a row, through the exact table (the accessors) while it has none; StateIndex of the end state per string."""
    rows, wide, pitch, off = t.wide_layout()
    orig_of_perm, _ = t.layout()
    perm_of_orig = np.empty_like(orig_of_perm)
    perm_of_orig[orig_of_perm] = np.arange(len(orig_of_perm), dtype=np.uint32)
    out = []
    for s in strings:
        st = int(perm_of_orig[t.Next(t.info.initial, 258)])     # Begin()
        for b in s:
            e = int(rows[st, t.letter_class(b)]) if st < wide else wide
            st = e if e != wide else int(perm_of_orig[t.Next(int(orig_of_perm[st]), b)])
        out.append(int(t.Next(int(orig_of_perm[st]), 259)))   # End()
    return out


@pytest.mark.parametrize("name,corpus", [("dict_1k", "k512"), ("dict_10k", "k2048"), ("blacklist_1k", "urls"), ("set_b_mix", "mix")])
def test_wide_image_walks_like_the_reference(name, corpus):
    """The LDS image of the wide walk (pire_hip_table_wide_layout): walked the way the kernel walks it, a string ends in
    the state the reference ends in -- rows, escape row, own-id field and the table behind them are consistent."""
    import pire_amd

    entry, s, text, offs = sample_of(name, corpus)
    t = pire_amd.Table(W.load_blob(entry["blob"]))
    info = t.info
    assert info.wide_states > info.hot_states and info.wide_lds_bytes <= 160 * 1024
    rows, wide, pitch, off = t.wide_layout()
    assert wide == info.wide_states and rows.shape == (wide + 1, pitch // 2) and pitch % 8 != 0 and off == 256   # (rows start in every bank in turn)
    assert (rows[wide, :info.letters] == wide).all() and (rows[:, :info.letters] <= wide).all()   # the escape row is absorbing
    k = min(24, len(offs) - 1)
    strings = [bytes(text[int(offs[i]):int(offs[i + 1])])[:300] for i in range(k)]
    o = ob.OracleScanner(W.load_blob(entry["blob"]))
    want, _ = o.run_strings(strings)
    assert walk_image(t, strings) == want.tolist()


def test_walk_choice_follows_the_measured_share(cfg):
    """pire_hip_config.walk_variant and the table's measured shares decide; tables that fit the dense rows have no image."""
    import pire_amd

    small = pire_amd.Table(H.load_blob([b for b in H.big_sets() if b["name"] == "c2_single"][0]["blob"]))
    assert small.info.wide_states == 0 and small.wide_layout()[0] is None
    t = pire_amd.Table(W.load_blob(W.wide_set("dict_1k")["blob"]))
    i = t.info
    assert i.wide_states >= 1700 and not i.shares_measured
    assert 0.0 <= i.outside_wide_share <= i.outside_dense_share <= 1.0


# ------------------------------------------------------------------------------------------------------------ GPU


def records_of(entry, corpus, seed, n, length):
    return W.wide_records(entry, corpus, seed, n, length)


@pytest.mark.gpu
@pytest.mark.parametrize("name,corpus", [("dict_1k", "k32"), ("dict_1k", "k1000"), ("dict_10k", "k2048"), ("dict_10k", "k10000"),
                                         ("set_b_mix", "mix"), ("dict_utf8_5k", "k5000")])
@pytest.mark.parametrize("n,length", [(64, 256), (65, 4096), (1000, 1024), (333, 128 * 5 + 16), (4096 + 7, 512), (128, 4096 + 48)])
def test_wide_kernel_vs_oracle(pa, torch_cuda, cfg, name, corpus, n, length):
    """pire_hip_run_strided with walk_variant = 2: partial waves, odd tile counts, tails shorter than a tile, counters."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    data = records_of(entry, corpus, n * 7 + length, n, length)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs, threads=4)
    d = torch.as_tensor(data, device="cuda")
    for variant, symbol, task in ((2, "ScanWideKernel", 64), (3, "ScanWide2Kernel", 128)):   # one / two strings per lane
        cfg.set(walk_variant=variant)
        gi, gf, cnt = dev_run_strided(torch, t, d)
        assert pb.last_kernel() in ("wide", "generic"), pb.last_kernel()   # "generic": the remainder of a task behind it
        assert symbol in pb.last_kernel_symbol() or n % task
        assert (gi == oi).all() and (gf == of).all(), variant
        assert (cnt == expected_counts(o, oi, of)).all()
    cfg.set(walk_variant=1)
    di, df, _ = dev_run_strided(torch, t, d)
    assert pb.last_kernel() in ("tiled", "generic")
    assert (di == oi).all() and (df == of).all()


_EARLY = {}


@pytest.mark.gpu
@pytest.mark.parametrize("walk,zipv", [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2)])
def test_early_out_between_chained_tasks(pa, torch_cuda, cfg, walk, zipv):
    """Several tasks per wave, an even number of tiles per record (the ring of two register tiles runs straight through task
    boundaries), and two tasks out of three whose 64 records ALL reach the absorbing state in their first tile: the wave-wide
    early-out (multi.h:955-958) ends such a task with a tile still on its way, and the next task must not see it.  Round 5's
    ScanWideKernel waited for that tile behind the loop, where hipcc had already copied the slot: 557 of 2^20 strings of a
    corpus like this one came out wrong (round 6).  Every fixed-length kernel, every string against the oracle."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(walk_variant=walk, zip_variant=zipv)
    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n, length = 2 * cus * 16 * 64 + 64 * 37, 512          # two tasks (of 64 strings) for every wave slot of the chip, and a few more
    key = (n, length)
    if _EARLY.get("key") != key:
        data = records_of(entry, "k128", 12345, n, length).copy()
        word = np.frombuffer(W.dictionary_words(entry)[7], dtype=np.uint8)
        planted = ((np.arange(n) // 64) % 3) != 0
        data[planted, 3:3 + len(word)] = word
        oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=8)
        assert of[planted].all() and not of[~planted].any()
        _EARLY.update(key=key, data=data, oi=oi, of=of)
    data, oi, of = _EARLY["data"], _EARLY["oi"], _EARLY["of"]
    d = torch.as_tensor(data, device="cuda")
    for _ in range(2):
        gi, gf, cnt = dev_run_strided(torch, t, d)
        assert pb.last_kernel() in ("wide", "tiled") and ("zipped" in pb.last_kernel_symbol()) == (zipv == 2 and walk != 1)
        bad = np.nonzero((gi != oi) | (gf != of))[0]
        assert len(bad) == 0, (len(bad), bad[:10].tolist(), (bad[:10] // 64).tolist())
        assert (cnt == expected_counts(o, oi, of)).all()
        t.adapt()


@pytest.mark.gpu
def test_wide_kernel_on_the_recorded_samples(pa, torch_cuda, cfg):
    """... and against what the compiled reference recorded in tests/golden/wide.json."""
    torch = torch_cuda
    cfg.set(walk_variant=2)
    for w in W.wide_sets():
        for corpus, s in w["samples"].items():
            if corpus == "urls":
                continue
            t = pa.Table(W.load_blob(w["blob"]))
            rec = W.wide_records(w, corpus, s["seed"], s["n"], s["len"])
            gi, gf, _ = dev_run_strided(torch, t, torch.as_tensor(rec, device="cuda"))
            assert gi.tolist() == s["idx"] and gf.tolist() == s["final"], (w["name"], corpus)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, ob.FLAG_BEGIN, ob.FLAG_END, BE])
def test_wide_kernel_flags_and_resume_states(pa, torch_cuda, cfg, flags):
    """Begin / End optional, resume states per string (also states WITHOUT a row), raw bytes of every value."""
    torch = torch_cuda
    entry = W.wide_set("dict_10k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 640, 768
    rng = np.random.RandomState(flags + 11)
    data = records_of(entry, "k2048", 99 + flags, n, length).copy()
    data[::5, 100:400] = rng.randint(0, 256, size=(len(data[::5]), 300), dtype=np.uint8)
    init = rng.randint(0, o.size, size=n).astype(np.uint32)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    d = torch.as_tensor(data, device="cuda")
    for ini in (None, init):
        oi, of = o.run(data.reshape(-1), offs, flags=flags, init_idx=ini, threads=4)
        for variant in (2, 3):
            cfg.set(walk_variant=variant)
            gi, gf, cnt = dev_run_strided(torch, t, d, flags=flags, init=ini)
            assert (gi == oi).all() and (gf == of).all(), variant
            assert (cnt == expected_counts(o, oi, of)).all()


@pytest.mark.gpu
def test_wide_kernel_with_a_ranking_that_knows_nothing(pa, torch_cuda, cfg):
    """prior_flat: rows for the first states BY INDEX -- most of the walk happens outside them, through the table in memory
    (both widths of it: u16 for dict_1k, u32 for a table of more than 65 536 states is not in the fixtures)."""
    from pire_amd import binding as pb

    torch = torch_cuda
    for name, corpus, variant in (("dict_1k", "k1000", 2), ("dict_10k", "k10000", 2), ("dict_1k", "k1000", 3), ("dict_10k", "k10000", 3)):
        cfg.set(prior_flat=1, walk_variant=variant)
        entry = W.wide_set(name)
        blob = W.load_blob(entry["blob"])
        t, o = pa.Table(blob), ob.OracleScanner(blob)
        n, length = 512, 1024
        data = records_of(entry, corpus, 4242, n, length)
        offs = np.arange(n + 1, dtype=np.uint64) * length
        oi, of = o.run(data.reshape(-1), offs, threads=4)
        gi, gf, _ = dev_run_strided(torch, t, torch.as_tensor(data, device="cuda"))
        assert pb.last_kernel() == "wide"
        assert (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
def test_wide_walk_is_chosen_once_the_scans_are_seen_leaving_the_dense_rows(pa, torch_cuda, cfg):
    """Default configuration: the first passes take the dense rows (the a-priori ranking says little about a dictionary),
    adapt() sees a third of the steps outside them, the next pass takes the wide walk -- same results; a table whose text
    stays inside the dense rows (set_a, headline corpus) never does."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(walk_variant=0)
    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 4096, 1024
    data = records_of(entry, "k512", 5, n, length)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs, threads=4)
    d = torch.as_tensor(data, device="cuda")
    kernels = []
    for _ in range(3):
        gi, gf, _c = dev_run_strided(torch, t, d)
        kernels.append(pb.last_kernel())
        assert (gi == oi).all() and (gf == of).all()
        t.adapt()
    info = t.refresh_info()
    assert info.shares_measured and info.outside_dense_share > 0.05, (info.outside_dense_share, kernels)
    assert kernels[-1] == "wide", kernels
    # after the wide walk has run, its own visit counters keep the ranking: still wide, shares still measured
    t.adapt()
    gi, gf, _c = dev_run_strided(torch, t, d)
    assert pb.last_kernel() == "wide" and (gi == oi).all()
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    ta = pa.Table(H.load_blob(big["blob"]))
    da = torch.as_tensor(ob.corpus_fill(3, 0, 2048, 1024, H.plants_for(big), threads=4), device="cuda")
    for _ in range(2):
        dev_run_strided(torch, ta, da)
        assert pb.last_kernel() == "tiled"
        ta.adapt()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 3])
def test_measured_share_outside_the_wide_rows_is_what_the_oracle_counts(pa, torch_cuda, cfg, variant):
    """pire_hip_table_info.outside_wide_share after adapt(): of the wide walk's visit samples (one lane per wave and tile,
    chosen by a hash of the wave's tile count) those that found their lane outside the rows.  Against the oracle's visit
    counts of the same batch under the best possible ranking: the first sampler took lane l at tile l of its string and
    nowhere else, and the share the library reported for a batch like this one was a twentieth of the truth."""
    torch = torch_cuda
    cfg.set(walk_variant=variant, zip_variant=1)   # (the plain rows: left to itself the library zips this table, tests/test_zip.py)
    entry = W.wide_set("dict_10k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 65536, 1024
    data = records_of(entry, "k2048", 9, n, length)
    d = torch.as_tensor(data, device="cuda")
    for _ in range(4):                      # the ranking settles (the estimates are remembered from adapt() to adapt())
        dev_run_strided(torch, t, d)
        t.adapt()
    info = t.refresh_info()
    v = np.sort(o.visit_counts(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length))[::-1].astype(np.float64)
    ideal = 1.0 - v[:info.wide_states].sum() / v.sum()
    assert ideal > 0.1, ideal                # a third of the steps of this corpus have no row whatever the ranking
    assert info.shares_measured
    assert 0.8 * ideal <= info.outside_wide_share <= 1.4 * ideal, (ideal, info.outside_wide_share)


# ---- offset batches of wide tables: the ragged kernel on the class-indexed walk ------------------------------------------


def dev_run_offsets(torch, t, text, offs, flags=BE, init=None, want_idx=True):
    from pire_amd import binding as pb  # noqa: F401

    n = len(offs) - 1
    d = torch.as_tensor(np.ascontiguousarray(text), device="cuda")
    doffs = torch.as_tensor(offs.astype(np.int64), device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
    init_t = None if init is None else torch.as_tensor(np.asarray(init, dtype=np.int32), device="cuda")
    t.run_device(d.data_ptr(), doffs.data_ptr(), n, flags, idx.data_ptr() if want_idx else 0, fin.data_ptr(), cnt.data_ptr(),
                 init_t.data_ptr() if init_t is not None else 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return idx.cpu().numpy().astype(np.uint32), fin.cpu().numpy(), cnt.cpu().numpy().astype(np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["blacklist_1k", "blacklist_10k"])
def test_ragged_kernel_on_the_wide_walk_urls(pa, torch_cuda, cfg, name):
    """URL batches of a blacklist scanner (samples/blacklist/blacklist.cpp:78-85, one Runner per URL) through pire_hip_run:
    walk_variant = 2 routes offset batches to the ragged kernel on the class-indexed walk; the recorded reference results,
    the oracle on a larger batch with empty strings in it, resume states, counters; and the dense rows for comparison."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    s = entry["samples"]["urls"]
    text, offs = W.wide_urls(entry, s["seed"], s["n"])
    cfg.set(walk_variant=2)
    gi, gf, _ = dev_run_offsets(torch, t, text, offs)
    assert pb.last_kernel() == "ragged_wide", pb.last_kernel()
    assert gi.tolist() == s["idx"] and gf.tolist() == s["final"]
    text, offs = W.wide_urls(entry, 77, 20000)
    lens = np.diff(offs).astype(np.int64)
    lens[::97] = 0                                   # empty strings
    offs2 = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs2[1:] = np.cumsum(lens)
    rng = np.random.RandomState(5)
    init = rng.randint(0, o.size, size=len(lens)).astype(np.uint32)
    for flags in (BE, 0):
        for ini in (None, init):
            oi, of = o.run(text, offs2, flags=flags, init_idx=ini, threads=4)
            gi, gf, cnt = dev_run_offsets(torch, t, text, offs2, flags=flags, init=ini)
            assert pb.last_kernel() == "ragged_wide"
            assert (gi == oi).all() and (gf == of).all(), (flags, ini is not None)
            assert (cnt == expected_counts(o, oi, of)).all()
            # without a StateIndex array the end states cannot be parked in it: the kernel finishes every string itself
            _, gf, cnt = dev_run_offsets(torch, t, text, offs2, flags=flags, init=ini, want_idx=False)
            assert (gf == of).all() and (cnt == expected_counts(o, oi, of)).all()
    cfg.set(walk_variant=1)
    oi, of = o.run(text, offs2, threads=4)
    gi, gf, _ = dev_run_offsets(torch, t, text, offs2)
    assert pb.last_kernel() in ("ragged", "stream") and (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
def test_ragged_kernel_on_the_wide_walk_records_cut_anywhere(pa, torch_cuda, cfg):
    """Dictionary records cut into strings of 0..700 bytes at any alignment: windows that start anywhere in a line, partial
    last chunks, strings shorter than a chunk, lanes that leave the rows (dict_10k, a corpus of 10 000 labels)."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(walk_variant=2)
    for name, corpus in (("dict_1k", "k512"), ("dict_10k", "k10000")):
        entry = W.wide_set(name)
        blob = W.load_blob(entry["blob"])
        t, o = pa.Table(blob), ob.OracleScanner(blob)
        text = W.wide_records(entry, corpus, 31, 2048, 1024).reshape(-1)
        rng = np.random.RandomState(8)
        lens = rng.randint(0, 700, size=5000).astype(np.uint64)
        lens[rng.randint(0, len(lens), size=300)] = rng.randint(0, 16, size=300)
        offs = np.zeros(len(lens) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(lens)
        assert int(offs[-1]) <= text.size
        oi, of = o.run(text, offs, threads=4)
        gi, gf, cnt = dev_run_offsets(torch, t, text[:int(offs[-1])], offs)
        assert pb.last_kernel() == "ragged_wide"
        assert (gi == oi).all() and (gf == of).all(), name
        assert (cnt == expected_counts(o, oi, of)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("zipv", [1, 2], ids=["plain rows", "zipped"])
@pytest.mark.parametrize("kind,n", [("urls", 70001), ("tiny", 30000), ("mixed", 20000), ("uniform", 1025), ("empty", 5000),
                                    ("aligned", 6000), ("edges", 70001), ("lines", 9000), ("skewed", 2500), ("urls", 300)])
def test_ragged_kernel_on_the_wide_walk_string_lengths_of_every_kind(pa, torch_cuda, cfg, kind, n, zipv):
    """The ragged kernel on the class-indexed walk, plain rows and zipped: URL-sized strings, runs of empty and tiny ones, ranges
    that end in the middle of a wave's grab, strings that all fill their window, a few very long ones.  (Written for round 6's
    experiment of handing a wave's strings out in the order of their lengths -- parity-green, 6 % slower, not kept: DESIGN.md 7 --
    and kept as the parity test it is.)"""
    from pire_amd import binding as pb
    from tests.test_gpu_parity import stream_lengths

    torch = torch_cuda
    cfg.set(no_offsets_peek=1, ragged_variant=1, walk_variant=2, zip_variant=zipv, auto_adapt=1)
    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(n + len(kind))
    ln = stream_lengths(rng, kind, n).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(ln)
    total = int(offs[-1])
    text = W.wide_records(entry, "k512", 23, (total + 1023) // 1024 + 1, 1024).reshape(-1)[:total].copy()
    for flags in (BE, 0):
        oi, of = o.run(text, offs, flags=flags, threads=4)
        gi, gf, cnt = dev_run_offsets(torch, t, text, offs, flags=flags)
        assert pb.last_kernel() == "ragged_wide", pb.last_kernel()
        bad = np.nonzero((gi != oi) | (gf != of))[0]
        assert len(bad) == 0, (flags, len(bad), bad[:10], ln[bad[:10]])
        assert (cnt == expected_counts(o, oi, of)).all()


STREAM_WIDE_CASES = [
    # scanner, corpus, zip_variant (1 plain rows, 2 zipped), lengths, strings, lead
    ("dict_1k", "k128", 1, "urls", 70001, 0),
    ("dict_1k", "k1000", 2, "urls", 70001, 77),
    ("dict_1k", "k512", 1, "tiny", 30000, 5),
    ("dict_1k", "k512", 2, "tiny", 30000, 3),
    ("dict_10k", "k10000", 1, "mixed", 20000, 1),
    ("dict_10k", "k10000", 2, "mixed", 20000, 127),
    ("dict_10k", "k2048", 2, "lines", 9000, 128),
    ("dict_10k", "k512", 1, "aligned", 6000, 112),
    ("dict_10k", "k512", 2, "aligned", 6000, 0),
    ("dict_utf8_5k", "k5000", 2, "edges", 70001, 3),
    ("dict_utf8_1k", "k1000", 1, "empty", 5000, 9),
    ("dict_1k", "k1000", 2, "uniform", 64, 0),
    ("dict_10k", "k10000", 2, "urls", 300000, 64),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,corpus,zipv,kind,n,lead", STREAM_WIDE_CASES, ids=lambda v: str(v))
def test_stream_kernel_on_the_wide_walk_vs_oracle(pa, torch_cuda, cfg, name, corpus, zipv, kind, n, lead):
    """the stream kernel (runs of consecutive strings per lane, boundaries inside the chunk walk) on the
    class-indexed walk -- the image with the smaller tier (internal.h StreamWideTier), plain rows and zipped -- against the
    oracle: text that visits thousands of states and leaves the tier (dict_10k / k10000), boundaries of every kind (inside a
    chunk, on a chunk, on a line, several per chunk, runs of empty strings), a text that starts anywhere in a line, both flag
    combinations, match counters; before and after the table has ranked its rows from these very scans."""
    from pire_amd import binding as pb
    from tests.test_gpu_parity import stream_lengths

    torch = torch_cuda
    cfg.set(no_offsets_peek=1, ragged_variant=2, walk_variant=2, zip_variant=zipv, auto_adapt=1)
    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(n + len(kind) + lead)
    ln = stream_lengths(rng, kind, n).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[0] = lead
    offs[1:] = lead + np.cumsum(ln)
    total = int(offs[-1])
    records = (total + 1023) // 1024 + 1
    text = W.wide_records(entry, corpus, 41 + lead, records, 1024).reshape(-1)[:total].copy()
    for round_ in range(2):
        for flags in (BE, 0):
            oi, of = o.run(text, offs, flags=flags, threads=4)
            gi, gf, cnt = dev_run_offsets(torch, t, text, offs, flags=flags)
            # (a zipped image whose rows leave no room for the strings' positions -- 113 letter classes, 230-byte rows -- keeps the
            # ragged kernel: internal.h StreamWideTier)
            either = name == "dict_utf8_5k" and zipv == 2   # (fits as created, not once the ranking has given it more rows)
            assert pb.last_kernel() in (("generic",) if n < 256 else ("stream_wide", "ragged_wide") if either else ("stream_wide",)), pb.last_kernel()
            assert ("zipped" in pb.last_kernel_symbol()) == (zipv == 2 and n >= 256), pb.last_kernel_symbol()
            bad = np.nonzero((gi != oi) | (gf != of))[0]
            assert len(bad) == 0, (round_, flags, len(bad), bad[:10], ln[bad[:10]], offs[bad[:10]])
            assert (cnt == expected_counts(o, oi, of)).all()
        t.adapt()   # the second round: rows ranked from what the stream kernel's samples said
    # resume states keep the ragged kernel (the stream kernel starts every string in the same state)
    init = rng.randint(0, t.Size, size=n).astype(np.uint32)
    oi, of = o.run(text, offs, flags=ob.FLAG_END, init_idx=init, threads=4)
    gi, gf, _ = dev_run_offsets(torch, t, text, offs, flags=ob.FLAG_END, init=init)
    assert pb.last_kernel() in ("ragged_wide", "generic")
    assert (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
def test_stream_kernel_on_the_wide_walk_is_opt_in(pa, torch_cuda, cfg):
    """Routing: measured slower than the ragged kernel on the same walk wherever lanes leave the tier, so a
    blacklist scanner's URL batch of a million strings keeps ragged_wide by default; ragged_variant = 2 asks for stream_wide;
    same answers."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry = W.wide_set("blacklist_1k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    cfg.set(no_offsets_peek=1, walk_variant=2, ragged_variant=2)
    t.upload()                                   # (the stream kernel's image is made for tables uploaded while it is asked for)
    for n in ((1 << 20) + 5, 3000):
        text, offs = W.wide_urls(entry, 5, n)
        oi, of = o.run(text, offs, threads=4)
        for variant, want in ((0, "ragged_wide"), (2, "stream_wide"), (1, "ragged_wide")):
            cfg.set(ragged_variant=variant)
            gi, gf, cnt = dev_run_offsets(torch, t, text, offs)
            assert pb.last_kernel() == want, (n, variant, pb.last_kernel())
            assert (gi == oi).all() and (gf == of).all()
            assert (cnt == expected_counts(o, oi, of)).all()


@pytest.mark.gpu
def test_two_tables_with_configurations_of_their_own(pa, torch_cuda, cfg):
    """pire_hip_table_config_set (round 6): two users of the library in one process, each with its own routing -- one table
    pinned to the dense rows, one to the class-indexed walk, the process-wide configuration untouched; same answers."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    o = ob.OracleScanner(blob)
    n, length = 2048, 1024
    data = records_of(entry, "k128", 3, n, length)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    d = torch.as_tensor(data, device="cuda")
    dense, wide = pa.Table(blob), pa.Table(blob)
    dense.set_config(walk_variant=1)
    wide.set_config(walk_variant=2)
    for _ in range(2):
        for t, kernel in ((dense, "tiled"), (wide, "wide"), (dense, "tiled")):
            gi, gf, _c = dev_run_strided(torch, t, d)
            assert pb.last_kernel() == kernel and (gi == oi).all() and (gf == of).all()
    assert pb.get_config().walk_variant == 0
