"""Few long strings (segmented.hip): strings are cut into segments that are scanned in parallel from GUESSED start
states (one guess per learned mode of the automaton); the host follows the chain of segments and accepts only what was
computed from the state the chain is really in -- so the results must be the reference's whatever the automaton and
however bad the guesses.  Tiny segments and warm-ups (environment knobs) make wrong guesses common on short test
strings; a parity automaton never forgets its start state, so its guesses are coin tosses until every state is a mode."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu


def tables():
    out = []
    for name in ("survey_known_answer", "inline_glue3", "set_a", "set_d"):
        c = [x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
        out.append((name, H.load_blob(c["blob"])))
    if ob.ref_available():
        # even number of a's / b's: the state after any text depends on where the walk started, for ever
        out.append(("parity", ob.RefScanner.compile(["(b*ab*a)*b*", "(a*ba*b)*a*"], ["n", "n"]).save()))
    return out


ALPHABET = b"abc ABCDEFGHIJKLMNOPQRSTUVWXYZ hello wd0123456789-()@net"


def strings_for(rng, name):
    lens = [0, 1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4096, 4097, 20000] + [int(x) for x in rng.randint(0, 3000, size=40)]
    a = np.frombuffer(b"ab" if name == "parity" else ALPHABET, dtype=np.uint8)
    out = [a[rng.randint(0, len(a), size=k)].tobytes() for k in lens]
    if name != "parity":
        tails = [b" hello   world", b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", b" http://yandex.ru/", b"(123) 456-7890", b"abc", b"xn"]
        out = [s + (tails[i % len(tails)] if i % 3 == 0 else b"") for i, s in enumerate(out)]
    return out


@pytest.mark.parametrize("name,blob", tables(), ids=[n for n, _ in tables()])
@pytest.mark.parametrize("seg,warm,modes,budget", [(64, 0, 6, 32), (100, 16, 1, 0), (256, 256, 6, 32), (128, 32, 2, 1),
                                                   (4096, 256, 3, 4)])
def test_segmented_scan_is_exact(name, blob, seg, warm, modes, budget, cfg):
    import pire_amd
    from pire_amd import binding as pb

    cfg.set(segment_bytes=str(seg))
    cfg.set(segment_warmup=str(warm))
    cfg.set(segment_modes=str(modes))      # 1: nothing is learned
    cfg.set(segment_budget=str(budget))    # 0: every surprise ends in the plain walk
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(seg * 7 + warm)
    strings = strings_for(rng, name)
    text, offs = H.pack(strings)
    for flags in (3, 0, 1, 2):
        oi, of = o.run(*ob.pack_strings(strings), flags=flags)
        gi, gf, cnt = t.run(text, offs, flags=flags, counts=True)
        assert pb.last_kernel().startswith("segmented")      # "+plain" where the budget of repairs ran out
        assert (gi == oi).all() and (gf == of).all(), (name, flags, np.nonzero(gi != oi)[0][:5])
        assert cnt[0] == int(of.sum()) and cnt[1] == len(strings)
        xi, xf = t.run(text, offs, flags=flags | pb.FLAG_GENERIC)
        assert pb.last_kernel() == "generic"
        assert (xi == oi).all() and (xf == of).all()
    # resume states (RunHelper(sc, st), run.h:391-392): every string from a state reached by another text
    pre = [s[:7] for s in strings]
    init, _ = o.run(*ob.pack_strings(pre), flags=1)
    oi, of = o.run(*ob.pack_strings(strings), flags=2, init_idx=init)
    gi, gf = t.run(text, offs, flags=2, init_idx=init)
    assert (gi == oi).all() and (gf == of).all()


def test_segmented_fixed_length_records_host_and_device(cfg):
    import torch
    import pire_amd
    from pire_amd import binding as pb

    cfg.set(segment_bytes="512")
    cfg.set(segment_warmup="64")
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    n, length = 37, 5000 + 8          # not a multiple of the segment size; stride 16-byte aligned
    data = ob.corpus_fill(big["corpus"]["seed"], 0, n, length, H.plants_for(big))
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs)
    gi, gf, cnt = t.run_strided_host(data, counts=True)
    assert pb.last_kernel() == "segmented"
    assert (gi == oi).all() and (gf == of).all() and cnt[1] == n and cnt[0] == int(of.sum())
    d = torch.as_tensor(np.array(data), device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    t.run_strided_device(d.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert pb.last_kernel() == "segmented"
    assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()


def test_one_long_string_takes_the_segmented_path_by_itself():
    """No knobs: 8 strings of 4 MiB are few and long -- the library cuts them up on its own, and on this (forgetful)
    automaton every chain must resolve without the sequential walk, whatever the other strings' chains do."""
    import torch
    import pire_amd
    from pire_amd import binding as pb

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    n, length = 8, 4 << 20
    data = ob.corpus_fill(big["corpus"]["seed"], 0, n * (length // 4096), 4096, H.plants_for(big)).reshape(n, length)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs)
    d = torch.as_tensor(np.array(data), device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
    t.run_strided_device(d.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), cnt.data_ptr(), 0,
                         torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert pb.last_kernel() == "segmented"
    assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()
    assert int(cnt[1]) == n and int(cnt[0]) == int(of.sum())
    gi, gf = t.run(data.reshape(-1), offs)            # host pointers, offsets
    assert pb.last_kernel() == "segmented"
    assert (gi == oi).all() and (gf == of).all()


@pytest.mark.parametrize("length,seg", [((1 << 20) + 1000, 2048), ((1 << 20), 4096), (300000, 1024)])
def test_single_string_grid_segments_through_the_tiled_kernel_plus_tail(length, seg, cfg):
    """One string on the device: its full segments are fixed-length records (tiled kernel), the tail is not."""
    import torch
    import pire_amd
    from pire_amd import binding as pb

    cfg.set(segment_bytes=str(seg))
    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    data = ob.corpus_fill(big["corpus"]["seed"], 0, (length + 4095) // 4096, 4096, H.plants_for(big)).reshape(-1)[:length]
    oi, of = o.run(data, np.array([0, length], dtype=np.uint64))
    d = torch.as_tensor(np.array(data), device="cuda")
    idx = torch.empty(1, dtype=torch.int32, device="cuda")
    fin = torch.empty(1, dtype=torch.uint8, device="cuda")
    for grid in (True, False):
        if not grid:
            cfg.set(segment_no_grid="1")
        t.run_strided_device(d.data_ptr(), 1, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0,
                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert pb.last_kernel() == "segmented"
        assert int(idx[0]) == int(oi[0]) and int(fin[0]) == int(of[0])


def test_resident_text_with_host_offsets():
    """PIRE_HIP_RUN_HOST_OFFSETS: the text stays on the device, the offsets come from the host -- which is what lets
    a batch of few long documents take the segmented scan; many short ones take the ragged kernel as usual."""
    import torch
    import pire_amd
    from pire_amd import binding as pb

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    total = 6 << 20
    data = ob.corpus_fill(big["corpus"]["seed"], 0, total // 4096, 4096, H.plants_for(big)).reshape(-1)
    d = torch.as_tensor(np.array(data), device="cuda")
    rng = np.random.RandomState(12)
    stream = torch.cuda.current_stream().cuda_stream
    for n, kernel in ((5, "segmented"), (20000, "ragged")):
        cuts = np.sort(rng.choice(np.arange(1, total), size=n - 1, replace=False)).astype(np.uint64)
        offs = np.concatenate([[0], cuts, [total]]).astype(np.uint64)
        oi, of = o.run(data, offs, threads=4)
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        fin = torch.empty(n, dtype=torch.uint8, device="cuda")
        cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
        t.run_device_host_offsets(d.data_ptr(), offs, 3, idx.data_ptr(), fin.data_ptr(), cnt.data_ptr(), 0, stream)
        assert pb.last_kernel() == kernel
        torch.cuda.synchronize()
        assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()
        assert int(cnt[1]) == n and int(cnt[0]) == int(of.sum())
    with pytest.raises(pb.PireHipError):
        t.run_device_host_offsets(d.data_ptr(), np.array([0, 10, 5], dtype=np.uint64), 3, 0, 0, 0, 0, stream)


def test_device_offsets_are_peeked_at_so_that_few_long_strings_take_the_segmented_scan(cfg):
    """Offsets on the DEVICE: the host does not know the lengths, and round 2 left such batches one string per lane
    whatever their shape.  A batch of fewer than 65 536 strings is now peeked at (first and last offset read back): few
    long documents take the segmented scan, many short ones the ragged kernel, and pire_hip_config.no_offsets_peek
    keeps the call enqueue-only (one string per lane).  Same results every way."""
    import torch
    import pire_amd
    from pire_amd import binding as pb

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    total = 6 << 20
    data = ob.corpus_fill(big["corpus"]["seed"], 0, total // 4096, 4096, H.plants_for(big)).reshape(-1)
    d = torch.as_tensor(np.array(data), device="cuda")
    rng = np.random.RandomState(13)
    stream = torch.cuda.current_stream().cuda_stream
    for n, lead, kernel in ((5, 0, "segmented"), (3, 4097, "segmented"), (20000, 0, "ragged")):
        cuts = np.sort(rng.choice(np.arange(lead + 1, total), size=n - 1, replace=False)).astype(np.uint64)
        offs = np.concatenate([[lead], cuts, [total]]).astype(np.uint64)        # offsets[0] need not be 0
        oi, of = o.run(data, offs, threads=4)
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        for peek in (True, False):
            cfg.set(no_offsets_peek=0 if peek else 1)
            idx = torch.full((n,), -1, dtype=torch.int32, device="cuda")
            fin = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
            cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
            t.run_device(d.data_ptr(), do.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), cnt.data_ptr(), 0, stream)
            torch.cuda.synchronize()
            assert pb.last_kernel().startswith(kernel if peek else ("generic" if n < 256 else "ragged")), (n, peek, pb.last_kernel())
            assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all(), (n, peek)
            assert int(cnt[1]) == n and int(cnt[0]) == int(of.sum())



def test_concurrent_threads_segmented_scans_on_one_table():
    """Four host threads, each on its own stream, run segmented scans of different texts on ONE table (learning and
    sharing its modes) at the same time; every result must equal the oracle's."""
    import threading

    import torch
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    n, length = 3, 1 << 20
    jobs = []
    for k in range(4):
        data = ob.corpus_fill(1000 + k, 0, n * (length // 4096), 4096, H.plants_for(big), threads=4).reshape(n, length)
        jobs.append((data, o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=2)))
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            data, (oi, of) = jobs[k]
            d = torch.as_tensor(np.array(data), device="cuda")
            idx = torch.empty(n, dtype=torch.int32, device="cuda")
            fin = torch.empty(n, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for _ in range(10):
                t.run_strided_device(d.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0,
                                     stream.cuda_stream)
                stream.synchronize()
                if not ((idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()):
                    errors.append("mismatch in thread %d" % k)
        except Exception as e:   # noqa: BLE001 -- reported below
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("seg,modes,budget", [(96, 6, 32), (256, 1, 0), (4096, 6, 32)])
def test_segmented_half_final_counting(seg, modes, budget, cfg):
    """HalfFinalScanner counting of few long strings: the chain gives every segment its true start state, the segments
    are then counted in parallel; a chain that had to fall back to the plain walk hands the batch to the ordinary
    half-final kernel.  Either way Result(r), StateIndex and Final are the oracle's."""
    import pire_amd
    from pire_amd import binding as pb

    cfg.set(segment_bytes=str(seg))
    cfg.set(segment_modes=str(modes))
    cfg.set(segment_budget=str(budget))
    g = H.golden()
    tables = [(c["name"], H.load_blob(c["blob"]), b"abcde w") for c in g["half_final"] if c["regexps"] <= 8][:3]
    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    tables.append(("set_d", H.load_blob(big["blob"]), ALPHABET))
    rng = np.random.RandomState(seg)
    for name, blob, alphabet in tables:
        t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
        a = np.frombuffer(alphabet, dtype=np.uint8)
        strings = [a[rng.randint(0, len(a), size=k)].tobytes() for k in (0, 1, 95, 96, 97, 1000, 5000, 20000, 333)]
        text, offs = H.pack(strings)
        for flags in (3, 0, 1, 2):
            oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=flags)
            gi, gf, gr = t.run_half_final(text, offs, flags=flags)
            if budget:
                assert pb.last_kernel().startswith("segmented")
            assert (gi == oi).all() and (gf == of).all(), (name, flags)
            assert (gr == orr).all(), (name, flags, gr.tolist(), orr.tolist())
        assert orr.sum() > 0


@pytest.mark.parametrize("name", ["set_a", "set_d", "parity"])
@pytest.mark.parametrize("n,length,seg", [(1, (1 << 21) + 777, 2048), (3, 1 << 19, 4096), (1, 100 * 1024 + 5, 1024)])
def test_two_modes_share_one_pass_of_the_pair_kernel(name, n, length, seg, cfg):
    """Once a table knows a second mode, the scan proper of mode 0 and of that mode is ONE pass of the fused pair kernel
    over the grid segments (the same table twice, the two guesses as start states): the same answers as with
    pire_hip_config.segment_no_pair, and as the reference's."""
    import torch
    import pire_amd
    from pire_amd import binding as pb

    tabs = dict(tables())
    if name not in tabs:
        pytest.skip("needs oracle/_ref to compile the parity scanner")
    cfg.set(segment_bytes=str(seg))   # the warm-up stays the default 256 B: whole pairs of tiles, so it can be fused
    t, o = pire_amd.Table(tabs[name]), ob.OracleScanner(tabs[name])
    rng = np.random.RandomState(n * 1000 + seg)
    if name == "parity":
        data = np.frombuffer(b"ab", dtype=np.uint8)[rng.randint(0, 2, size=n * length)]
    else:
        big = [b for b in H.big_sets() if b["name"] == name][0]
        data = ob.corpus_fill(big["corpus"]["seed"], 0, (n * length + 4095) // 4096, 4096, H.plants_for(big)).reshape(-1)[:n * length]
    data = np.ascontiguousarray(data)
    oi, of = o.run(data, np.arange(n + 1, dtype=np.uint64) * length)
    d = torch.as_tensor(data, device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    fused = one_mode = derived = product = 0
    for rep in range(6):
        # the last two: every mode walked, and by the pair kernel (no derived modes, no product automaton)
        cfg.set(segment_no_derive=1 if rep >= 4 else 0, segment_no_product=1 if rep >= 4 else 0)
        cfg.set(segment_no_pair=1 if rep == 3 else 0)
        idx.fill_(-1)
        t.run_strided_device(d.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
        torch.cuda.synchronize()
        assert pb.last_kernel().startswith("segmented")
        fused += pb.last_kernel_symbol() == "pirehip::ScanPairTiledKernel"
        one_mode += pb.last_kernel_symbol() == "pirehip::ScanTiledSegKernel"
        derived += pb.last_kernel_symbol() == "pirehip::ScanTiledSegKernel+derived"
        product += pb.last_kernel_symbol() == "pirehip::ScanTiledSegKernel+product"
        assert rep != 3 or pb.last_kernel_symbol() not in ("pirehip::ScanPairTiledKernel", "pirehip::ScanTiledSegKernel")
        assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()
    if name != "parity":
        # the first call learned the second mode from the planted matches; calls two and three walked the two modes as one
        # walk of their product automaton -- or, where the learned mode is a function of mode 0 (ModeFunction; with
        # `$`-anchored patterns it mostly is not), walked one and derived the other; the last two walked both in one pass of
        # the pair kernel (the learning call itself may already derive a mode it has just learned)
        assert derived + product >= 2 and fused == 2, (derived, product, fused, one_mode)
    else:                      # surrounded patterns over {a, b} forget at once: ONE mode, its warm-up inside the tiled pass
        assert fused == 0 and derived == 0 and product == 0 and one_mode == 5


def test_a_mode_that_is_a_function_of_mode_zero_is_not_walked(cfg):
    """Unanchored patterns (grep's use): "error was seen" is a sticky mode, and the state under it is a FUNCTION of the
    state of the walk from the start state (the same walk with one more pattern marked as seen) -- ModeFunction proves it
    on the product automaton, and the segmented scan then fills that mode's slots from mode 0's by a table lookup
    instead of walking the text a second time.  Results against the reference with the derivation on and off."""
    import torch
    import pire_amd
    from pire_amd import binding as pb

    if not ob.ref_available():
        pytest.skip("needs oracle/_ref to compile the scanner")
    blob = ob.RefScanner.compile(["error", "time ?out", "fa+tal"]).save()   # surrounded: .*error.* and so on
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(21)
    n, length = 1, (1 << 21) + 4096
    a = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz   .,:/", dtype=np.uint8)
    data = a[rng.randint(0, len(a), size=n * length)].copy()
    for pos, word in ((700000, b" error "), (1300000, b" timeout "), (1900000, b" faaatal ")):
        data[pos:pos + len(word)] = np.frombuffer(word, dtype=np.uint8)
    oi, of = o.run(data, np.array([0, length], dtype=np.uint64))
    d = torch.as_tensor(data, device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    derived = 0
    for rep in range(6):
        cfg.set(segment_no_derive=1 if rep >= 4 else 0)
        idx.fill_(-1)
        t.run_strided_device(d.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0,
                             torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert pb.last_kernel().startswith("segmented")
        derived += pb.last_kernel_symbol().endswith("+derived")
        assert rep < 4 or not pb.last_kernel_symbol().endswith("+derived")
        assert int(idx[0]) == int(oi[0]) and int(fin[0]) == int(of[0]), rep
    assert derived >= 2, derived     # the calls after the one that learned the modes
