"""The ZIPPED image of the class-indexed walk (round 6; pire_amd/csrc/internal.h MakeWideLayout, table.cpp PlanZip / ChooseZip,
wide_common.h WideEntry<ZIP>): a row of their own for at most 1 022 states, every other state of the tier a 4-byte header (the row it
is equal to except in <= 3 letters, and those letters) + three targets -- four times the states of a dictionary scanner
(samples/blacklist/blacklist.cpp:65-76) in a CU's LDS.

CPU part: the image is walked on the host exactly as the kernel walks it (the header's bit fields, the multiply that compares
three letters at once, the escape state, the exact table behind the tier) against the oracle.
GPU part (-m gpu): the kernels on the zipped image against the oracle and the recorded reference results, through the C ABI."""
import numpy as np
import pytest

from oracle import binding as ob
from pire_amd import workloads as W
from tests import helpers as H
from tests.test_gpu_parity import dev_run_strided, expected_counts, pa, torch_cuda  # noqa: F401  (fixtures)
from tests.test_wide import dev_run_offsets, records_of, sample_of

BE = ob.FLAG_BEGIN | ob.FLAG_END


def zip_step(z, st, c):
    """wide_common.h WideEntry<true>: header, the three letter fields compared by one multiply and one xor, ONE u16 read."""
    h = int(z["headers"][st])
    x = h ^ ((2 * c) * 0x4081)
    if not (x & 0x0000FE):
        return int(z["targets"][st - z["full"], 0])
    if not (x & 0x007F00):
        return int(z["targets"][st - z["full"], 1])
    if not (x & 0x3F8000):
        return int(z["targets"][st - z["full"], 2])
    return int(z["rows"][h >> 22, c])


def walk_zip(t, z, strings):
    orig_of_perm, _ = t.layout()
    perm_of_orig = np.empty_like(orig_of_perm)
    perm_of_orig[orig_of_perm] = np.arange(len(orig_of_perm), dtype=np.uint32)
    tier = z["tier"]
    out, inside, steps = [], 0, 0
    for s in strings:
        st = int(perm_of_orig[t.Next(t.info.initial, 258)])     # Begin()
        for b in s:
            e = zip_step(z, st, t.letter_class(b)) if st < tier else tier
            inside += e != tier
            steps += 1
            st = e if e != tier else int(perm_of_orig[t.Next(int(orig_of_perm[st]), b)])
        out.append(int(t.Next(int(orig_of_perm[st]), 259)))   # End()
    return out, inside / max(1, steps)


@pytest.mark.parametrize("name,corpus", [("dict_1k", "k1000"), ("dict_10k", "k2048"), ("dict_10k", "k10000"), ("blacklist_10k", "urls"),
                                         ("set_b_mix", "mix"), ("dict_utf8_1k", "k1000"), ("dict_utf8_5k", "k5000")])
def test_zipped_image_walks_like_the_reference(cfg, name, corpus):
    import pire_amd

    cfg.set(zip_variant=2)
    entry, s, text, offs = sample_of(name, corpus)
    blob = W.load_blob(entry["blob"])
    t = pire_amd.Table(blob)
    z = t.zip_layout()
    info = t.refresh_info()
    assert z is not None and info.zip_full_states == z["full"] and info.wide_states == z["tier"]
    assert info.wide_lds_bytes <= 160 * 1024 - 64
    assert info.hot_states <= z["full"] <= 1022 < z["tier"] <= info.states
    assert t.wide_layout()[0] is None                           # the plain accessor does not describe a zipped image
    rows, hdr = z["rows"], z["headers"]
    assert (rows[z["full"], :info.letters] == z["tier"]).all() and (rows[:, :info.letters] <= z["tier"]).all()
    assert (hdr[:z["full"]] >> 22 == np.arange(z["full"])).all()          # a state with a row leans on itself ...
    assert ((hdr[:z["full"]] & 0x3FFFFE) == 0x3FFFFE).all()               # ... and has no exceptions
    assert int(hdr[z["tier"]]) >> 22 == z["full"]                           # the escape state: the escape row
    assert ((hdr[z["full"]:z["tier"]] >> 22) < z["full"]).all() and (z["targets"] <= z["tier"]).all()
    # every zipped state's row, rebuilt from header + targets + base row, is the table's row (clamped to the tier)
    orig_of_perm, _ = t.layout()
    perm_of_orig = np.empty_like(orig_of_perm)
    perm_of_orig[orig_of_perm] = np.arange(len(orig_of_perm), dtype=np.uint32)
    # bit 0 of a header: "Final" -- what the walks with actions look for in a chunk (ragged.hip WideChunkAct)
    flagged = np.array([t.Final(int(x)) for x in orig_of_perm[:z["tier"]]])
    assert ((hdr[:z["tier"]] & 1).astype(bool) == flagged).all() and not (int(hdr[z["tier"]]) & 1)
    o = ob.OracleScanner(blob)
    byte_of_class = {}
    for ch in list(range(256)) + [258, 259]:
        byte_of_class.setdefault(t.letter_class(ch), ch)
    rng = np.random.RandomState(3)
    for st in rng.randint(z["full"], z["tier"], size=200):
        for c, ch in byte_of_class.items():
            want = min(int(perm_of_orig[o.next(int(orig_of_perm[st]), ch)]), z["tier"])
            assert zip_step(z, int(st), c) == want, (st, c)
    k = min(24, len(offs) - 1)
    strings = [bytes(text[int(offs[i]):int(offs[i + 1])])[:300] for i in range(k)]
    want, _ = o.run_strings(strings)
    got, inside = walk_zip(t, z, strings)
    assert got == want.tolist()
    assert inside > 0.5   # (an a-priori ranking: most steps of these corpora are on shallow states, which it knows)


def test_zip_needs_a_measurement_or_an_order(cfg):
    """Default configuration: a fresh table keeps the plain rows (the byte model says little about thousands of states);
    zip_variant = 1 never zips; tables that fit the plain rows never do."""
    import pire_amd

    blob = W.load_blob(W.wide_set("dict_10k")["blob"])
    t = pire_amd.Table(blob)
    assert t.zip_layout() is None and t.info.zip_full_states == 0 and t.info.wide_states >= 1700
    cfg.set(zip_variant=2)
    small = pire_amd.Table(H.load_blob([b for b in H.big_sets() if b["name"] == "c2_single"][0]["blob"]))
    assert small.zip_layout() is None and small.info.wide_states == 0
    cfg.set(zip_variant=1)
    assert pire_amd.Table(blob).zip_layout() is None


# ------------------------------------------------------------------------------------------------------------ GPU


@pytest.mark.gpu
@pytest.mark.parametrize("name,corpus", [("dict_1k", "k1000"), ("dict_10k", "k2048"), ("dict_10k", "k10000"), ("set_b_mix", "mix"),
                                         ("dict_utf8_1k", "k1000"), ("dict_utf8_5k", "k5000")])
@pytest.mark.parametrize("n,length", [(64, 256), (65, 4096), (1000, 1024), (333, 128 * 5 + 16), (4096 + 7, 512), (128, 4096 + 48)])
def test_zipped_kernels_vs_oracle(pa, torch_cuda, cfg, name, corpus, n, length):
    """pire_hip_run_strided on the zipped image, one and two strings per lane: partial waves, odd tile counts, tails, counters."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(zip_variant=2)
    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    data = records_of(entry, corpus, n * 7 + length, n, length)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs, threads=4)
    d = torch.as_tensor(data, device="cuda")
    for variant, symbol, task in ((2, "ScanWideKernel", 64), (3, "ScanWide2Kernel", 128)):
        cfg.set(walk_variant=variant)
        gi, gf, cnt = dev_run_strided(torch, t, d)
        assert pb.last_kernel() in ("wide", "generic"), pb.last_kernel()
        assert (symbol in pb.last_kernel_symbol() and "zipped" in pb.last_kernel_symbol()) or n % task
        assert (gi == oi).all() and (gf == of).all(), variant
        assert (cnt == expected_counts(o, oi, of)).all()
    assert t.refresh_info().zip_full_states > 0


@pytest.mark.gpu
def test_zipped_kernels_on_the_recorded_samples(pa, torch_cuda, cfg):
    """... against what the compiled reference recorded in tests/golden/wide.json (records and URL batches)."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(zip_variant=2, walk_variant=3)
    for w in W.wide_sets():
        for corpus, s in w["samples"].items():
            t = pa.Table(W.load_blob(w["blob"]))
            if corpus == "urls":
                text, offs = W.wide_urls(w, s["seed"], s["n"])
                gi, gf, _ = dev_run_offsets(torch, t, text, offs)
                assert pb.last_kernel() == "ragged_wide" and "zipped" in pb.last_kernel_symbol()
            else:
                rec = W.wide_records(w, corpus, s["seed"], s["n"], s["len"])
                gi, gf, _ = dev_run_strided(torch, t, torch.as_tensor(rec, device="cuda"))
            assert gi.tolist() == s["idx"] and gf.tolist() == s["final"], (w["name"], corpus)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, ob.FLAG_BEGIN, ob.FLAG_END, BE])
def test_zipped_kernels_flags_and_resume_states(pa, torch_cuda, cfg, flags):
    """Begin / End optional, resume states per string (states with a row, zipped states, states outside the tier), raw bytes."""
    torch = torch_cuda
    cfg.set(zip_variant=2)
    entry = W.wide_set("dict_10k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 640, 768
    rng = np.random.RandomState(flags + 11)
    data = records_of(entry, "k10000", 99 + flags, n, length).copy()
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
def test_zipped_kernels_with_a_ranking_that_knows_nothing(pa, torch_cuda, cfg):
    """prior_flat: the tier = the first states BY INDEX, whatever leans on whatever -- most of the walk goes through the table."""
    from pire_amd import binding as pb

    torch = torch_cuda
    for name, corpus, variant in (("dict_1k", "k1000", 2), ("dict_10k", "k10000", 2), ("dict_10k", "k10000", 3)):
        cfg.set(prior_flat=1, zip_variant=2, walk_variant=variant)
        entry = W.wide_set(name)
        blob = W.load_blob(entry["blob"])
        t, o = pa.Table(blob), ob.OracleScanner(blob)
        n, length = 512, 1024
        data = records_of(entry, corpus, 4242, n, length)
        offs = np.arange(n + 1, dtype=np.uint64) * length
        oi, of = o.run(data.reshape(-1), offs, threads=4)
        gi, gf, _ = dev_run_strided(torch, t, torch.as_tensor(data, device="cuda"))
        assert pb.last_kernel() == "wide" and "zipped" in pb.last_kernel_symbol()
        assert (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
def test_the_library_zips_a_table_whose_scans_leave_the_plain_rows(pa, torch_cuda, cfg):
    """Default configuration (zip_variant = 0): dict_10k / k2048 leaves the 2 207 plain rows in a third of its steps; after the
    adaptations that see that the image is zipped, the tier holds thousands of states more, the share outside it is a fraction --
    and every answer is what it was.  dict_1k / k128 (1 500 states visited: they fit the plain rows) stays on the plain rows."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(walk_variant=0, zip_variant=0)
    entry = W.wide_set("dict_10k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 32768, 1024
    data = records_of(entry, "k2048", 5, n, length)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs, threads=4)
    d = torch.as_tensor(data, device="cuda")
    seen = []
    for _ in range(5):
        gi, gf, _c = dev_run_strided(torch, t, d)
        assert (gi == oi).all() and (gf == of).all()
        t.adapt()
        i = t.refresh_info()
        seen.append((pb.last_kernel_symbol(), i.zip_full_states, i.wide_states, round(i.zip_plain_outside_share, 4), round(i.zip_outside_share, 4)))
    assert i.zip_full_states > 0 and i.wide_states > 5000, seen
    assert i.zip_outside_share < 0.6 * i.zip_plain_outside_share, seen
    gi, gf, _c = dev_run_strided(torch, t, d)
    assert "zipped" in pb.last_kernel_symbol() and (gi == oi).all() and (gf == of).all(), seen
    t.adapt()
    assert t.refresh_info().outside_wide_share < 0.12, t.info.outside_wide_share   # (the plain rows: 0.29)
    entry = W.wide_set("dict_1k")
    t1, o1 = pa.Table(W.load_blob(entry["blob"])), ob.OracleScanner(W.load_blob(entry["blob"]))
    data = records_of(entry, "k128", 5, n, length)
    oi, of = o1.run(data.reshape(-1), offs, threads=4)
    d = torch.as_tensor(data, device="cuda")
    for _ in range(4):
        gi, gf, _c = dev_run_strided(torch, t1, d)
        assert (gi == oi).all() and (gf == of).all()
        t1.adapt()
    assert t1.refresh_info().zip_full_states == 0 and "zipped" not in pb.last_kernel_symbol()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["blacklist_1k", "blacklist_10k"])
def test_ragged_kernel_on_the_zipped_image_urls(pa, torch_cuda, cfg, name):
    """URL batches (samples/blacklist/blacklist.cpp:78-85) through pire_hip_run on the zipped image: empty strings, resume
    states, counters, with and without a StateIndex array."""
    from pire_amd import binding as pb

    torch = torch_cuda
    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    cfg.set(zip_variant=2, walk_variant=2)
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    text, offs = W.wide_urls(entry, 77, 20000)
    lens = np.diff(offs).astype(np.int64)
    lens[::97] = 0
    offs2 = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs2[1:] = np.cumsum(lens)
    rng = np.random.RandomState(5)
    init = rng.randint(0, o.size, size=len(lens)).astype(np.uint32)
    for flags in (BE, 0):
        for ini in (None, init):
            oi, of = o.run(text, offs2, flags=flags, init_idx=ini, threads=4)
            gi, gf, cnt = dev_run_offsets(torch, t, text, offs2, flags=flags, init=ini)
            assert pb.last_kernel() == "ragged_wide" and "zipped" in pb.last_kernel_symbol()
            assert (gi == oi).all() and (gf == of).all(), (flags, ini is not None)
            assert (cnt == expected_counts(o, oi, of)).all()
            _, gf, cnt = dev_run_offsets(torch, t, text, offs2, flags=flags, init=ini, want_idx=False)
            assert (gf == of).all() and (cnt == expected_counts(o, oi, of)).all()


@pytest.mark.gpu
def test_ragged_kernel_on_the_zipped_image_records_cut_anywhere(pa, torch_cuda, cfg):
    """Dictionary records cut into strings of 0..700 bytes at any alignment, lanes that leave the tier (dict_10k / k10000)."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(zip_variant=2, walk_variant=2)
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
        oi, of = o.run(text, offs, threads=4)
        gi, gf, cnt = dev_run_offsets(torch, t, text[:int(offs[-1])], offs)
        assert pb.last_kernel() == "ragged_wide" and "zipped" in pb.last_kernel_symbol()
        assert (gi == oi).all() and (gf == of).all(), name
        assert (cnt == expected_counts(o, oi, of)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 3])
def test_measured_share_outside_the_zipped_tier_is_what_the_oracle_counts(pa, torch_cuda, cfg, variant):
    """pire_hip_table_info.outside_wide_share of a zipped table: of the walk's visit samples (the states with a row: counters in
    LDS; zipped states: straight to memory; the escape state) those that found their lane outside the tier -- against the oracle's
    visit counts of the same batch over the states the device's tier really holds."""
    torch = torch_cuda
    cfg.set(walk_variant=variant, zip_variant=2)
    entry = W.wide_set("dict_10k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 65536, 1024
    data = records_of(entry, "k10000", 9, n, length)
    d = torch.as_tensor(data, device="cuda")
    for _ in range(4):
        dev_run_strided(torch, t, d)
        t.adapt()
    dev_run_strided(torch, t, d)
    orig_of_perm, _ = t.layout()                       # the numbering the last pass ran with
    tier = t.refresh_info().wide_states
    t.adapt()
    info = t.refresh_info()
    v = o.visit_counts(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length).astype(np.float64)
    true = 1.0 - v[orig_of_perm[:tier]].sum() / v.sum()
    assert info.zip_full_states > 0 and tier > 5000
    assert 0.05 < true < 0.3, true                     # 30 000 states visited, a tier of ~10 000
    assert 0.75 * true <= info.outside_wide_share <= 1.35 * true, (true, info.outside_wide_share)


@pytest.mark.gpu
def test_a_zipped_table_goes_back_to_the_plain_rows_when_its_traffic_fits_them(pa, torch_cuda, cfg):
    """The zipped walk is the slower one inside its tier (three LDS instructions per byte, not two): a table that was zipped for
    one corpus and then sees text that stays inside ~1 500 states is given its plain rows back by adapt()."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(walk_variant=0, zip_variant=0)
    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length = 32768, 1024
    offs = np.arange(n + 1, dtype=np.uint64) * length
    big = records_of(entry, "k1000", 5, n, length)
    d = torch.as_tensor(big, device="cuda")
    for _ in range(5):
        dev_run_strided(torch, t, d)
        t.adapt()
    assert t.refresh_info().zip_full_states > 0, "k1000 visits 4 000 states: the library zips"
    small = records_of(entry, "k32", 6, n, length)
    oi, of = o.run(small.reshape(-1), offs, threads=4)
    d = torch.as_tensor(small, device="cuda")
    for _ in range(8):   # (the estimates of the old corpus are halved at every adapt())
        gi, gf, _c = dev_run_strided(torch, t, d)
        assert (gi == oi).all() and (gf == of).all()
        t.adapt()
    assert t.refresh_info().zip_full_states == 0
    gi, gf, _c = dev_run_strided(torch, t, d)
    assert "zipped" not in pb.last_kernel_symbol() and (gi == oi).all() and (gf == of).all()
