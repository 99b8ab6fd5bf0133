"""Adaptation in the background (round 6; table.cpp BackgroundAdaptStep, pire_hip_config.auto_adapt = 0): a caller that only
enqueues (PIRE_HIP_RUN_ON_DEVICE) never calls pire_hip_table_adapt() and never lets the library drain the device -- VERDICT r5:
such a table stayed at the speed of its a-priori ranking for ever.  Now its launch boundaries start a worker thread (counters
copied on a stream of its own, a copy of the table re-ranked, the new image uploaded) and swap the result in; the calls stay
legal inside a stream capture.  The reference's Run() needs no tuning call either (run.h:271-275)."""
import time

import numpy as np
import pytest

from oracle import binding as ob
from pire_amd import workloads as W
from tests.test_gpu_parity import dev_run_strided, pa, torch_cuda  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _batch(torch, name, corpus, n, length, seed=5):
    entry = W.wide_set(name)
    blob = W.load_blob(entry["blob"])
    data = W.wide_records(entry, corpus, seed, n, length)
    o = ob.OracleScanner(blob)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    return blob, torch.as_tensor(data, device="cuda"), oi, of


def test_an_enqueue_only_caller_gets_an_adapted_table_without_ever_asking(pa, torch_cuda, cfg):
    """dict_1k / k128: a third of the steps outside the 255 dense rows.  Enqueue-only passes, the caller waiting only for its own
    stream between them: within a handful of passes the library has taken the class-indexed walk (from the trap signal alone, before
    any ranking) and swapped in a table ranked from the scans (adaptations >= 1) -- no adapt() call, every answer what it was."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(auto_adapt=0, walk_variant=0, zip_variant=0)
    blob, d, oi, of = _batch(torch, "dict_1k", "k128", 32768, 1024)
    t = pa.Table(blob)
    t.upload()
    kernels, adapts = [], []
    for i in range(12):
        gi, gf, _ = dev_run_strided(torch, t, d)          # run_strided_device + the caller's own synchronize
        assert (gi == oi).all() and (gf == of).all(), i
        kernels.append(pb.last_kernel())
        adapts.append(t.refresh_info().adaptations)
        if adapts[-1] >= 1 and kernels[-1] == "wide" and i >= 3:
            break
        time.sleep(0.05)                                   # (the worker thread ranks 4 000 states and uploads 600 KB)
    assert kernels[0] == "tiled", kernels                  # the a-priori ranking: dense rows
    assert "wide" in kernels[:4], kernels                  # the trap signal of the first passes is enough to change the walk
    assert adapts[-1] >= 1 and kernels[-1] == "wide", (kernels, adapts)
    info = t.refresh_info()
    assert info.shares_measured and info.outside_dense_share > 0.1
    gi, gf, _ = dev_run_strided(torch, t, d)
    assert (gi == oi).all() and (gf == of).all()


def test_round_5s_default_never_adapts_in_an_enqueue_only_call(pa, torch_cuda, cfg):
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(auto_adapt=3, walk_variant=0)
    blob, d, oi, of = _batch(torch, "dict_1k", "k128", 16384, 1024)
    t = pa.Table(blob)
    for i in range(6):
        gi, gf, _ = dev_run_strided(torch, t, d)
        assert (gi == oi).all() and (gf == of).all()
        time.sleep(0.02)
    assert t.refresh_info().adaptations == 0


def test_enqueue_only_calls_stay_capturable_while_the_table_adapts_underneath(pa, torch_cuda, cfg):
    """A HIP graph captured from an enqueue-only call while background adaptations are allowed: the capture succeeds (the call makes
    no synchronising call), the replays give the oracle's answers before and after the table under them has been swapped (the
    replaced image stays alive), and so do plain calls in between."""
    from pire_amd import binding as pb

    torch = torch_cuda
    cfg.set(auto_adapt=0, walk_variant=0, zip_variant=0)
    blob, d, oi, of = _batch(torch, "dict_1k", "k512", 32768, 1024)
    n, length = d.shape
    t = pa.Table(blob)
    t.upload()
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    flags = pb.FLAG_BEGIN | pb.FLAG_END

    def call(stream):
        t.run_strided_device(d.data_ptr(), n, length, length, flags, idx.data_ptr(), fin.data_ptr(), 0, 0, stream.cuda_stream)

    with torch.cuda.stream(side):
        call(side)                                     # the first use of the kernel (its self-test runs outside the capture)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call(side)
    swaps = 0
    for i in range(10):
        idx.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all(), ("replay", i)
        gi, gf, _ = dev_run_strided(torch, t, d)       # plain enqueue-only calls: they start / swap in the adaptation
        assert (gi == oi).all() and (gf == of).all(), ("call", i)
        swaps = t.refresh_info().adaptations
        if swaps >= 1 and i >= 2:
            break
        time.sleep(0.05)
    assert swaps >= 1, "the table never adapted"
    for _ in range(2):
        idx.zero_()
        g.replay()                                     # the graph still holds the image it was captured with
        torch.cuda.synchronize()
        assert (idx.cpu().numpy().astype(np.uint32) == oi).all()


def test_explicit_adapt_and_destroy_wait_for_the_worker(pa, torch_cuda, cfg):
    """pire_hip_table_adapt() and pire_hip_table_destroy() while a worker may be on its way: no crash, right answers."""
    torch = torch_cuda
    cfg.set(auto_adapt=0, walk_variant=0, auto_adapt_min_traps=8)
    blob, d, oi, of = _batch(torch, "dict_10k", "k2048", 16384, 1024)
    for round_ in range(3):
        t = pa.Table(blob)
        for i in range(3):
            gi, gf, _ = dev_run_strided(torch, t, d)
            assert (gi == oi).all() and (gf == of).all()
        if round_ == 0:
            t.adapt()                                  # joins the worker, ranks synchronously
            gi, gf, _ = dev_run_strided(torch, t, d)
            assert (gi == oi).all() and (gf == of).all()
        del t                                          # destroy right behind the last launch boundary


def test_background_adaptation_through_the_c_abi_alone_from_several_threads(cfg):
    """No torch: device buffers from pire_hip_device_alloc, four host threads making enqueue-only calls on ONE table (each on the
    default stream, each waiting for its own results) while the table adapts in the background underneath them -- every answer the
    oracle's.  This is the test the ThreadSanitizer build of the library runs (tools/gpu_final_r06.sh: torch's own HIP
    initialisation does not survive the preloaded sanitizer runtime, the library's does)."""
    import ctypes as C
    import threading

    import pire_amd
    from pire_amd import binding as pb

    if pire_amd.device_count() == 0:
        pytest.skip("needs a HIP device")
    cfg.set(auto_adapt=0, walk_variant=0, zip_variant=0, auto_adapt_min_traps=64)
    entry = W.wide_set("dict_1k")
    blob = W.load_blob(entry["blob"])
    n, length = 16384, 1024
    data = W.wide_records(entry, "k512", 21, n, length)
    o = ob.OracleScanner(blob)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    L = pb.lib()
    t = pire_amd.Table(blob)
    t.upload()
    dtext = C.c_void_p()
    assert L.pire_hip_device_alloc(data.size, C.byref(dtext)) == 0
    assert L.pire_hip_copy_to_device(dtext, data.ctypes.data, data.size, None) == 0 and L.pire_hip_stream_synchronize(None) == 0
    errors = []

    def worker(k):
        didx, dfin = C.c_void_p(), C.c_void_p()
        try:
            assert L.pire_hip_device_alloc(n * 4, C.byref(didx)) == 0 and L.pire_hip_device_alloc(n, C.byref(dfin)) == 0
            gi, gf = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint8)
            for i in range(10):
                t.run_strided_device(dtext.value, n, length, length, pb.FLAG_BEGIN | pb.FLAG_END, didx.value, dfin.value, 0, 0, 0)
                assert L.pire_hip_copy_to_host(gi.ctypes.data, didx, n * 4, None) == 0
                assert L.pire_hip_copy_to_host(gf.ctypes.data, dfin, n, None) == 0
                assert L.pire_hip_stream_synchronize(None) == 0
                if not ((gi == oi).all() and (gf == of).all()):
                    errors.append((k, i, int((gi != oi).sum())))
                time.sleep(0.01)
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))
        finally:
            L.pire_hip_device_free(didx)
            L.pire_hip_device_free(dfin)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    L.pire_hip_device_free(dtext)
    assert not errors, errors
    assert t.refresh_info().adaptations >= 1, "forty passes over a corpus that leaves the a-priori rows: the table never adapted"
