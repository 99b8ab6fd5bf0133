"""The SHIPPED configuration: the session fixture of conftest.py pins auto_adapt off because most tests pin
which rows are dense, so the parity evidence described a configuration a drop-in user does not run.  Here the library's
defaults are restored and the host-pointer entry points -- the ones in which the default policy re-ranks the dense rows by
itself -- are run on tables that WILL adapt (a prior that knows nothing, a low trigger), from several host threads at
once, every result against the oracle: pire_hip_run (ragged, fixed-length, few long strings = segmented scan), the prefix
and suffix searches and HalfFinal counting, whose start states are derived on the host from the numbering of the image a
call holds (api.cpp FillParams: ScanParams::hostPermOfOrig) while another thread's adaptation replaces the table's."""
import threading

import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    assert pire_amd.device_count() > 0, "GPU tests need a HIP device; the library has no CPU fallback"
    return pire_amd


def _table(pa, cfg, name, min_traps=8):
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    cfg.set(prior_flat=1, auto_adapt=0, auto_adapt_min_traps=min_traps, no_offsets_peek=0)
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.layout()               # ranks the rows now: a prior that knows nothing
    cfg.set(prior_flat=0)
    return big, t, o


def test_shipped_defaults_are_what_the_header_says(pa):
    from pire_amd import binding as pb

    c = pb.get_config()
    # conftest pins auto_adapt = 1 for the session; everything else is the library's default
    for f in ("tiled_variant", "checked", "no_compact", "no_segments", "ragged_variant", "no_offsets_peek", "host_staging"):
        assert getattr(c, f) == 0, f


@pytest.mark.parametrize("name", ["set_a", "set_d"])
def test_host_entry_points_under_the_default_policy(pa, cfg, name):
    """One thread, the default policy, every host-pointer entry point in turn on a table that adapts underneath."""
    big, t, o = _table(pa, cfg, name)
    rng = np.random.RandomState(5)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet"
    strings = H.random_strings(rng, 3000, 300, alphabet)
    # (every fifth string gets a witness: the walk leaves the know-nothing prior's rows often enough for a trigger of 8 sampled traps
    # whatever lanes the samples fall on -- with 40 strings per witness it was a matter of which lane was looked at when)
    for w in [bytes.fromhex(h) for h in big["witnesses_hex"]]:
        for k in rng.randint(0, len(strings), size=600 // max(1, len(big["witnesses_hex"])) * 2):
            strings[k] = strings[k][: len(strings[k]) // 2] + w
    text, offs = H.pack(strings)
    want = o.run(text, offs, threads=4)
    want_hf = o.run_half_final(text, offs)
    for rep in range(4):
        gi, gf = t.run(text, offs)
        assert (gi == want[0]).all() and (gf == want[1]).all(), rep
        for longest in (True, False):
            for tb, te in ((False, False), (True, True)):
                assert (t.prefix(text, offs, longest, tb, te) == o.prefix(text, offs, longest, tb, te)).all(), (rep, longest, tb, te)
        assert (t.suffix(text, offs, True, True, False) == o.suffix(text, offs, True, True, False)).all()
        hi, hf, hr = t.run_half_final(text, offs)
        assert (hi == want_hf[0]).all() and (hf == want_hf[1]).all() and (hr == want_hf[2]).all(), rep
    if name == "set_a":   # (set_d's walks stay inside the know-nothing prior's rows on this text: nothing to adapt to)
        assert t.refresh_info().adaptations >= 1   # it did adapt along the way


def test_every_host_entry_point_while_the_table_adapts_on_other_threads(pa, cfg):
    """Eight host threads, one table from the know-nothing prior with a low trigger: ragged runs, fixed-length runs, few
    long strings (segmented scan, mode learning included), prefix / suffix searches and HalfFinal counting -- each thread
    one kind, all at once, every result against the oracle, while the automatic adaptations (up to 6) replace the images
    and the numbering underneath the calls in flight."""
    big, t, o = _table(pa, cfg, "set_a", min_traps=4)
    rng = np.random.RandomState(9)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet"
    witnesses = [bytes.fromhex(h) for h in big["witnesses_hex"]]

    def batch(n, maxlen):
        strings = H.random_strings(rng, n, maxlen, alphabet)
        for w in witnesses:
            for k in rng.randint(0, len(strings), size=max(2, n // 60)):
                strings[k] = strings[k][: len(strings[k]) // 2] + w
        return H.pack(strings)

    jobs = []
    tx, ofs = batch(2500, 250)
    jobs.append(("run", tx, ofs, o.run(tx, ofs, threads=2)))
    data = ob.corpus_fill(321, 0, 2048, 512, H.plants_for(big), threads=2)
    jobs.append(("strided", data, None, o.run(data.reshape(-1), np.arange(2049, dtype=np.uint64) * 512, threads=2)))
    long_data = ob.corpus_fill(654, 0, 6, 1 << 18, H.plants_for(big), threads=2)   # 6 x 256 KiB: the segmented scan
    long_offs = np.arange(7, dtype=np.uint64) * (1 << 18)
    jobs.append(("long", long_data.reshape(-1), long_offs, o.run(long_data.reshape(-1), long_offs, threads=2)))
    tx, ofs = batch(1500, 200)
    jobs.append(("prefix", tx, ofs, o.prefix(tx, ofs, True, True, True)))
    tx, ofs = batch(1500, 200)
    jobs.append(("shortest", tx, ofs, o.prefix(tx, ofs, False, False, False)))
    tx, ofs = batch(1500, 200)
    jobs.append(("suffix", tx, ofs, o.suffix(tx, ofs, True, True, True)))
    tx, ofs = batch(2000, 220)
    jobs.append(("half_final", tx, ofs, o.run_half_final(tx, ofs)))
    tx, ofs = batch(400, 90)
    jobs.append(("half_final_small", tx, ofs, o.run_half_final(tx, ofs)))
    errors = []

    def worker(job):
        kind, tx, ofs, want = job
        try:
            for rep in range(10):
                if kind == "run" or kind == "long":
                    got = t.run(tx, ofs)
                    ok = (got[0] == want[0]).all() and (got[1] == want[1]).all()
                elif kind == "strided":
                    got = t.run_strided_host(tx)
                    ok = (got[0] == want[0]).all() and (got[1] == want[1]).all()
                elif kind == "prefix":
                    ok = (t.prefix(tx, ofs, True, True, True) == want).all()
                elif kind == "shortest":
                    ok = (t.prefix(tx, ofs, False, False, False) == want).all()
                elif kind == "suffix":
                    ok = (t.suffix(tx, ofs, True, True, True) == want).all()
                else:
                    got = t.run_half_final(tx, ofs)
                    ok = all((g == w).all() for g, w in zip(got, want))
                if not ok:
                    errors.append((kind, rep))
        except Exception as e:   # noqa: BLE001
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    assert t.refresh_info().adaptations >= 1


def test_scans_while_another_thread_calls_adapt(pa, cfg):
    """pire_hip_table_adapt() itself, called from a fifth thread while four threads scan.  Rounds 1-3
    documented that as "must not run concurrently": an adaptation frees the images, and a scan could be between copying
    their pointers and launching.  Now every entry point holds the table's lock shared until its kernels are enqueued
    (internal.h TableUse) and adapt() takes it exclusively and drains the devices before it frees anything: legal, and
    every result must still equal the oracle's."""
    big, t, o = _table(pa, cfg, "set_a", min_traps=1 << 30)   # no automatic adaptation: the fifth thread does it
    rng = np.random.RandomState(21)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet"
    witnesses = [bytes.fromhex(h) for h in big["witnesses_hex"]]
    jobs = []
    for k in range(4):
        strings = H.random_strings(rng, 1500, 220, alphabet)
        for w in witnesses:
            for j in rng.randint(0, len(strings), size=20):
                strings[j] = strings[j][: len(strings[j]) // 2] + w
        tx, ofs = H.pack(strings)
        jobs.append((k, tx, ofs, o.run(tx, ofs, threads=2), o.prefix(tx, ofs, True, True, True), o.run_half_final(tx, ofs)))
    errors, stop = [], threading.Event()
    adapts = [0]

    def scanner(job):
        k, tx, ofs, want, want_p, want_hf = job
        try:
            for rep in range(80):
                if rep >= 12 and adapts[0] >= 3:   # (at least 12 rounds, and on until three adaptations ran underneath them:
                    break                          #  an adaptation uploads more since round 5 -- the wide walk's image)
                if k % 2 == 0:
                    got = t.run(tx, ofs)
                    ok = (got[0] == want[0]).all() and (got[1] == want[1]).all()
                elif k == 1:
                    ok = (t.prefix(tx, ofs, True, True, True) == want_p).all()
                else:
                    got = t.run_half_final(tx, ofs)
                    ok = all((g == w).all() for g, w in zip(got, want_hf))
                if not ok:
                    errors.append((k, rep))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    def adapter():
        try:
            while not stop.is_set():
                t.adapt()
                adapts[0] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(("adapt", repr(e)))

    threads = [threading.Thread(target=scanner, args=(j,)) for j in jobs]
    ad = threading.Thread(target=adapter)
    ad.start()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    stop.set()
    ad.join()
    assert not errors, errors[:5]
    assert adapts[0] >= 3
