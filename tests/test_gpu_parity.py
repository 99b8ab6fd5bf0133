"""Parity of the HIP path (through the C ABI) with the oracle / golden vectors.  Needs an MI355X.

Bit-exact bar: StateIndex, Final flag and AcceptedRegexps of every string equal the reference's."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu

BE = ob.FLAG_BEGIN | ob.FLAG_END


@pytest.fixture(scope="module")
def pa():
    import pire_amd

    assert pire_amd.device_count() > 0, "GPU tests need a HIP device; the library has no CPU fallback"
    return pire_amd


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available()
    return torch


def dev_run_strided(torch, t, data2d, flags=BE, init=None, counts=True, generic=False):
    """data2d: torch uint8 [n, stride] on cuda (len == stride).  Returns numpy idx, final, counts."""
    from pire_amd import binding as pb

    n, length = data2d.shape
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
    init_t = None if init is None else torch.as_tensor(np.asarray(init, dtype=np.int32), device="cuda")
    t.run_strided_device(data2d.data_ptr(), n, length, data2d.stride(0), flags | (pb.FLAG_GENERIC if generic else 0),
                         idx.data_ptr(), fin.data_ptr(), cnt.data_ptr() if counts else 0,
                         init_t.data_ptr() if init_t is not None else 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return idx.cpu().numpy().astype(np.uint32), fin.cpu().numpy(), cnt.cpu().numpy().astype(np.uint64)


def expected_counts(o, idx, fin):
    cnt = np.zeros(o.regexps + 2, dtype=np.uint64)
    cnt[0] = int(fin.sum())
    cnt[1] = len(idx)
    cache = {}
    for i in idx.tolist():
        if i not in cache:
            cache[i] = o.accepted(i)
        for r in cache[i]:
            cnt[2 + r] += 1
    return cnt


@pytest.mark.parametrize("case", H.all_cases(), ids=lambda c: c["name"])
def test_golden_cases_through_c_abi(pa, case):
    """Every known-answer vector of the reference's unit tests, via pire_hip_run (host pointers, generic kernel)."""
    t = pa.Table(H.load_blob(case["blob"]))
    strings = H.case_strings(case)
    idx, fin, cnt = t.run_strings(strings, counts=True)
    assert idx.tolist() == case["idx"]
    assert fin.tolist() == case["final"]
    assert [t.AcceptedRegexps(int(i)) for i in idx] == case["accepted"]
    for i, want in zip(idx, case.get("ref_expect", [])):
        if want is not None:
            assert (len(t.AcceptedRegexps(int(i))) > 0) == want
    assert cnt[0] == sum(case["final"]) and cnt[1] == len(strings)
    for r in range(t.RegexpsCount):
        assert cnt[2 + r] == sum(1 for a in case["accepted"] if r in a)
    # the batched Runner vocabulary
    text, offs = H.pack(strings)
    br = pa.BatchRunner(t).Begin().Run(text, offs).End()
    assert br.State().tolist() == case["idx"] and br.Final().tolist() == [bool(f) for f in case["final"]]


@pytest.mark.parametrize("big", H.big_sets(), ids=lambda b: b["name"])
def test_big_sets_golden_tiled_and_generic(pa, torch_cuda, big):
    torch = torch_cuda
    t = pa.Table(H.load_blob(big["blob"]))
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    c = big["corpus"]
    data = ob.corpus_fill(c["seed"], 0, c["n"], c["len"], H.plants_for(big))
    d = torch.as_tensor(data, device="cuda")
    from pire_amd import binding as pb

    for generic in (False, True):
        idx, fin, cnt = dev_run_strided(torch, t, d, generic=generic)
        assert pb.last_kernel() == ("generic" if generic else "tiled")
        assert idx.tolist() == c["idx"] and fin.tolist() == c["final"]
        assert (cnt == expected_counts(o, idx, fin)).all()
    raw = [bytes.fromhex(h) for h in big["raw"]["strings_hex"]]
    idx, fin = t.run_strings(raw)
    assert idx.tolist() == big["raw"]["idx"] and fin.tolist() == big["raw"]["final"]


@pytest.mark.parametrize("name", ["set_a", "set_d", "set_b", "survey_known_answer", "inline_glue3", "utf8_dot"])
def test_random_ragged_batches_vs_oracle(pa, name):
    """Ragged, unaligned, empty strings; bytes 0..255; every flag combination."""
    case = [c for c in H.all_cases() + H.big_sets() if c["name"] == name][0]
    blob = H.load_blob(case["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(99)
    strings = (H.random_strings(rng, 700, 300) + [b""] * 5 +
               H.random_strings(rng, 700, 200, b"abcdefghxyzHeadInrTl ABCXYZ0123456789()-wo\t\xd0\xb0\xc1\x81@Qnet"))
    for flags in (BE, 0, ob.FLAG_BEGIN, ob.FLAG_END):
        oi, of = o.run_strings(strings, flags=flags)
        gi, gf = t.run_strings(strings, flags=flags)
        assert (gi == oi).all() and (gf == of).all()


def dev_run_ragged(torch, t, text, offs, flags=BE, init=None, generic=False):
    """text/offs: numpy; staged by torch, run with device pointers.  Returns numpy idx, final, counts."""
    from pire_amd import binding as pb

    n = len(offs) - 1
    d = torch.as_tensor(np.ascontiguousarray(text), device="cuda") if len(text) else torch.zeros(16, dtype=torch.uint8, device="cuda")
    do = torch.as_tensor(offs.astype(np.int64), device="cuda")
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
    init_t = None if init is None else torch.as_tensor(np.asarray(init, dtype=np.int32), device="cuda")
    t.run_device(d.data_ptr(), do.data_ptr(), n, flags | (pb.FLAG_GENERIC if generic else 0), idx.data_ptr(),
                 fin.data_ptr(), cnt.data_ptr(), init_t.data_ptr() if init_t is not None else 0,
                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return idx.cpu().numpy().astype(np.uint32), fin.cpu().numpy(), cnt.cpu().numpy().astype(np.uint64)


def ragged_lengths(rng, kind, n):
    if kind == "uniform":
        return rng.randint(0, 700, size=n)
    if kind == "edges":      # every tile / chunk boundary and its neighbours
        base = np.array([0, 1, 15, 16, 17, 31, 32, 112, 113, 127, 128, 129, 143, 144, 145, 255, 256, 257, 383, 384, 400])
        return base[rng.randint(0, len(base), size=n)]
    if kind == "skewed":     # a few very long strings among many short ones: lanes must be re-used
        ln = rng.randint(0, 40, size=n)
        ln[rng.randint(0, n, size=6)] = rng.randint(20000, 60000, size=6)
        return ln
    if kind == "empty":
        return np.zeros(n, dtype=np.int64)
    raise ValueError(kind)


@pytest.mark.parametrize("name", ["set_a", "set_d", "c2_single"])
@pytest.mark.parametrize("kind,n", [("uniform", 5000), ("edges", 3000), ("skewed", 2500), ("empty", 300),
                                    ("uniform", 256), ("edges", 70001)])
def test_ragged_kernel_vs_oracle(pa, torch_cuda, name, kind, n, cfg):
    """The dynamically scheduled ragged kernel (offset batches of >= 256 strings): every length class, lane re-use,
    strings that end exactly at the end of the buffer, all flag combinations, match counts, resumed states."""
    from pire_amd import binding as pb

    # the test is about the ragged kernel: no look at the offsets that could choose another, and not the stream kernel
    # (stream.hip, tested below), which takes large offset batches by default
    cfg.set(no_offsets_peek=1, ragged_variant=1)
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(n + len(kind))
    ln = ragged_lengths(rng, kind, n).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(ln)
    total = int(offs[-1])
    alphabet = np.frombuffer(b"abcdeaxHedInrTailhello w0123456789/.-_?=&%:@\n", dtype=np.uint8)
    text = alphabet[rng.randint(0, len(alphabet), size=total)].astype(np.uint8)
    for w in [bytes.fromhex(h) for h in big["witnesses_hex"]]:
        for _ in range(50):
            if total > len(w) + 1:
                q = rng.randint(0, total - len(w))
                text[q:q + len(w)] = np.frombuffer(w, dtype=np.uint8)
    for flags in (BE, 0):
        oi, of = o.run(text, offs, flags=flags, threads=4)
        gi, gf, cnt = dev_run_ragged(torch, t, text, offs, flags=flags)
        assert pb.last_kernel() == "ragged"    # device offsets: size unknown to the host, always ragged for n >= 256
        assert (gi == oi).all() and (gf == of).all()
        assert (cnt == expected_counts(o, oi, of)).all()
    # resume: every string continues from an arbitrary reachable state
    init = rng.randint(0, t.Size, size=n).astype(np.uint32)
    oi, of = o.run(text, offs, flags=ob.FLAG_END, init_idx=init, threads=4)
    gi, gf, _ = dev_run_ragged(torch, t, text, offs, flags=ob.FLAG_END, init=init)
    assert (gi == oi).all() and (gf == of).all()
    # host-pointer entry point takes the same kernel
    if total >= 4096:
        gi, gf = t.run(text, offs, flags=ob.FLAG_END, init_idx=init)
        assert pb.last_kernel() == "ragged"
        assert (gi == oi).all() and (gf == of).all()
    gi, gf, _ = dev_run_ragged(torch, t, text, offs, flags=ob.FLAG_END, init=init, generic=True)
    assert pb.last_kernel() == "generic"
    assert (gi == oi).all() and (gf == of).all()


def stream_lengths(rng, kind, n):
    if kind == "urls":
        return rng.randint(20, 200, size=n)
    if kind == "tiny":       # several strings per 16-byte chunk, empty ones among them: the exact multi-boundary walk
        return rng.randint(0, 9, size=n)
    if kind == "lines":
        return rng.randint(64, 1024, size=n)
    if kind == "mixed":      # short strings, empty runs, line-sized ones, a few long ones
        ln = rng.randint(0, 64, size=n)
        ln[rng.randint(0, n, size=n // 7)] = 0
        ln[rng.randint(0, n, size=n // 9)] = rng.randint(100, 3000, size=n // 9)
        ln[rng.randint(0, n, size=3)] = rng.randint(30000, 90000, size=3)
        return ln
    if kind == "aligned":    # every string ends on a 128-byte line (boundaries at the end of a window) or on a chunk
        return rng.randint(0, 5, size=n) * 128 + rng.randint(0, 3, size=n) * 16
    return ragged_lengths(rng, kind, n)


@pytest.mark.parametrize("name", ["set_a", "set_d", "c2_single"])
@pytest.mark.parametrize("kind,n,lead", [("urls", 70001, 0), ("urls", 5000, 77), ("tiny", 30000, 5), ("lines", 9000, 128),
                                         ("mixed", 20000, 1), ("aligned", 6000, 0), ("aligned", 6000, 112),
                                         ("edges", 70001, 3), ("empty", 5000, 9), ("skewed", 2500, 0), ("uniform", 64, 0),
                                         ("uniform", 1025, 31), ("urls", 300000, 64)])
def test_stream_kernel_vs_oracle(pa, torch_cuda, name, kind, n, lead, cfg):
    """The stream kernel (stream.hip: every lane walks a run of consecutive strings, boundaries inside the chunk walk)
    against the oracle: boundary positions of every kind -- inside a chunk, on a chunk, on a line, several per chunk,
    empty strings in runs -- text that starts anywhere in a line (lead), a buffer that ends with the last string, batches
    of one sub-task and of hundreds, both flag combinations, match counts; and against the ragged kernel it replaces."""
    from pire_amd import binding as pb

    cfg.set(no_offsets_peek=1, ragged_variant=2)
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(n + len(kind) + lead)
    ln = stream_lengths(rng, kind, n).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[0] = lead
    offs[1:] = lead + np.cumsum(ln)
    total = int(offs[-1])
    alphabet = np.frombuffer(b"abcdeaxHedInrTailhello w0123456789/.-_?=&%:@\n", dtype=np.uint8)
    text = alphabet[rng.randint(0, len(alphabet), size=max(total, 1))].astype(np.uint8)
    for w in [bytes.fromhex(h) for h in big["witnesses_hex"]]:
        for _ in range(200):
            if total > len(w) + 1:
                q = rng.randint(0, total - len(w))
                text[q:q + len(w)] = np.frombuffer(w, dtype=np.uint8)
    text = text[:total] if total else text[:0]
    for flags in (BE, 0):
        oi, of = o.run(text, offs, flags=flags, threads=4)
        gi, gf, cnt = dev_run_ragged(torch, t, text, offs, flags=flags)
        assert pb.last_kernel() == ("stream" if n >= 256 else "generic")
        bad = np.nonzero((gi != oi) | (gf != of))[0]
        assert len(bad) == 0, (len(bad), bad[:10], ln[bad[:10]], offs[bad[:10]])
        assert (cnt == expected_counts(o, oi, of)).all()
    # resume states keep the ragged kernel (the stream kernel starts every string in the same state)
    init = rng.randint(0, t.Size, size=n).astype(np.uint32)
    oi, of = o.run(text, offs, flags=ob.FLAG_END, init_idx=init, threads=4)
    gi, gf, _ = dev_run_ragged(torch, t, text, offs, flags=ob.FLAG_END, init=init)
    assert pb.last_kernel() in ("ragged", "generic")
    assert (gi == oi).all() and (gf == of).all()


def test_stream_kernel_is_the_default_for_large_offset_batches(pa, torch_cuda, cfg):
    """Routing: offsets on the device and a million strings -> stream; fewer -> ragged; pire_hip_config.ragged_variant = 1
    -> ragged; host offsets: by the bytes the host then knows."""
    from pire_amd import binding as pb

    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(11)
    for n, want in (((1 << 20) + 5, "stream"), (3000, "ragged")):
        ln = rng.randint(0, 40, size=n).astype(np.uint64)
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum(ln)
        text = rng.randint(32, 127, size=int(offs[-1])).astype(np.uint8)
        oi, of = o.run(text, offs, threads=4)
        cfg.set(no_offsets_peek=1)
        gi, gf, _ = dev_run_ragged(torch, t, text, offs)
        assert pb.last_kernel() == want
        assert (gi == oi).all() and (gf == of).all()
        cfg.set(ragged_variant=1)
        gi, gf, _ = dev_run_ragged(torch, t, text, offs)
        assert pb.last_kernel() == "ragged"
        assert (gi == oi).all() and (gf == of).all()
        cfg.set(ragged_variant=0)
        # the host-pointer form knows the bytes: 20 MB of text is the ragged kernel's whatever the string count
        gi, gf = t.run(text, offs)
        assert pb.last_kernel() == "ragged"
        assert (gi == oi).all() and (gf == of).all()


def test_ragged_kernel_offsets_not_from_zero(pa, torch_cuda):
    """offsets[0] > 0 and a buffer that ends exactly with the last string (no readable slack behind it)."""
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(3)
    n = 1000
    ln = rng.randint(0, 300, size=n).astype(np.uint64)
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[0] = 777
    offs[1:] = 777 + np.cumsum(ln)
    text = rng.randint(0, 256, size=int(offs[-1]), dtype=np.uint8)
    oi, of = o.run(text, offs, threads=4)
    gi, gf, _ = dev_run_ragged(torch, t, text, offs)
    assert (gi == oi).all() and (gf == of).all()


@pytest.mark.parametrize("name", ["set_a", "set_d", "set_b", "c2_single"])
@pytest.mark.parametrize("n,length", [(1, 128), (63, 256), (64, 4096), (65, 384), (1000, 1024), (4097, 128 * 3 + 16),
                                      (300, 100), (129, 4096 + 48), (128, 128), (192, 256 + 16)])
def test_tiled_kernel_shapes_vs_oracle(pa, torch_cuda, name, n, length):
    """Tile-count / lane-count edge cases of the tiled kernel: partial waves, odd tile counts, tails shorter than
    a tile, lengths below one tile (routed to the generic kernel)."""
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    data = ob.corpus_fill(n * 31 + length, 0, n, length, H.plants_for(big), threads=4)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs, threads=4)
    d = torch.as_tensor(data, device="cuda")
    gi, gf, cnt = dev_run_strided(torch, t, d)
    assert (gi == oi).all() and (gf == of).all()
    assert (cnt == expected_counts(o, oi, of)).all()


@pytest.mark.parametrize("name", ["set_a", "set_d", "set_b", "c2_single"])
def test_tiled_kernel_with_rotated_columns(pa, torch_cuda, name, cfg):
    """pire_hip_config.tiled_variant = 23: the dense rows with their columns in rotated byte order (bank = byte & 63, the
    variant tables with spread-out traffic take by themselves) -- fast path, traps through the compact tier
    and the full table (text that leaves the dense rows), tails shorter than a tile -- against the oracle and against the
    plain rows (24)."""
    from pire_amd import binding as pb

    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(23)
    for n, length in ((64 * 9 + 3, 1024), (256, 128 * 3 + 16), (64, 4096 + 48)):
        data = ob.corpus_fill(n + length, 0, n, length, H.plants_for(big), threads=4)
        # raw bytes into a third of the strings: the walk leaves the dense rows
        k = rng.randint(0, n, size=n // 3)
        data[k, : length // 2] = rng.randint(0, 256, size=(len(k), length // 2), dtype=np.uint8)
        offs = np.arange(n + 1, dtype=np.uint64) * length
        oi, of = o.run(data.reshape(-1), offs, threads=4)
        d = torch.as_tensor(data, device="cuda")
        for variant, symbol in ((23, "rotated columns"), (24, "16,2,nt,5>")):
            cfg.set(tiled_variant=variant)
            gi, gf, cnt = dev_run_strided(torch, t, d)
            assert symbol in pb.last_kernel_symbol(), pb.last_kernel_symbol()
            assert (gi == oi).all() and (gf == of).all(), (variant, n, length)
            assert (cnt == expected_counts(o, oi, of)).all()


def test_cold_states_are_exact(pa, torch_cuda):
    """Text that drives set_d far outside its 255 dense rows: the trap / exact re-walk path must stay bit-exact."""
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(5)
    alphabet = np.frombuffer(b"abcdeaxHedInrTailhello w", dtype=np.uint8)
    n, length = 2048, 1024
    data = alphabet[rng.randint(0, len(alphabet), size=(n, length))].astype(np.uint8)
    # splice whole witnesses in so the product automaton wanders deep
    for i in range(n):
        for w in (b"HeadInnerInner", b"abc", b"aaa", b"adddde", b"hello   w"):
            p = rng.randint(0, length - 20)
            data[i, p:p + len(w)] = np.frombuffer(w, dtype=np.uint8)
    offs = np.arange(n + 1, dtype=np.uint64) * length
    oi, of = o.run(data.reshape(-1), offs, threads=4)
    orig_of_perm, _ = t.layout()
    hot_orig = set(orig_of_perm[:t.info.hot_states].tolist())
    assert len(set(oi.tolist()) - hot_orig) > 0, "test must end in states outside the dense rows"
    gi, gf, cnt = dev_run_strided(torch, t, torch.as_tensor(data, device="cuda"))
    assert (gi == oi).all() and (gf == of).all()
    assert (cnt == expected_counts(o, oi, of)).all()
    gi2, gf2 = t.run(data.reshape(-1), offs)   # host-pointer entry point, uniform offsets -> tiled as well
    assert (gi2 == oi).all() and (gf2 == of).all()


def test_resume_from_state_indices(pa, torch_cuda):
    """Chunked scanning: Runner(sc, st) resume (run.h:368, 391-392) == scanning the whole string."""
    torch = torch_cuda
    big = H.big_sets()[0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    n, length, cut = 500, 1024, 384
    data = ob.corpus_fill(11, 0, n, length, H.plants_for(big))
    whole_i, whole_f = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length)
    a = torch.as_tensor(np.ascontiguousarray(data[:, :cut]), device="cuda")
    b = torch.as_tensor(np.ascontiguousarray(data[:, cut:]), device="cuda")
    i1, _, _ = dev_run_strided(torch, t, a, flags=ob.FLAG_BEGIN, counts=False)
    i2, f2, _ = dev_run_strided(torch, t, b, flags=ob.FLAG_END, init=i1, counts=False)
    assert (i2 == whole_i).all() and (f2 == whole_f).all()
    # the same through the generic kernel and host pointers
    i1h, _ = t.run(np.ascontiguousarray(data[:, :cut]).reshape(-1), np.arange(n + 1, dtype=np.uint64) * cut,
                   flags=ob.FLAG_BEGIN)
    assert (i1h == i1).all()


def test_step_kernel_is_pire_step(pa, torch_cuda):
    torch = torch_cuda
    case = [c for c in H.all_cases() if c["name"] == "survey_known_answer"][0]
    blob = H.load_blob(case["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    st = torch.arange(t.Size, dtype=torch.int32, device="cuda")
    for ch in (258, ord("h"), 32, 259):
        cur = st.cpu().numpy()
        t.step_device(st.data_ptr(), t.Size, ch, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert st.cpu().numpy().tolist() == [o.next(int(s), ch) for s in cur]


def test_empty_inputs_and_empty_scanner(pa):
    case = [c for c in H.all_cases() if c["name"] == "empty_scanner"][0]
    t = pa.Table(H.load_blob(case["blob"]))
    assert t.Empty and t.RegexpsCount == 0
    idx, fin = t.run_strings([b"a strin", b"", b"x" * 500])
    assert idx.tolist() == [0, 0, 0] and fin.tolist() == [0, 0, 0]
    t2 = pa.Table(H.load_blob(H.all_cases()[0]["blob"]))
    idx, fin = t2.run_strings([])                      # n == 0
    assert len(idx) == 0
    idx, fin = t2.run_strings([b"", b"", b""])          # null-ish range, pire_ut.cpp:832-837
    o = ob.OracleScanner(H.load_blob(H.all_cases()[0]["blob"]))
    assert idx.tolist() == o.run_strings([b"", b"", b""])[0].tolist()


@pytest.mark.parametrize("name,length", [("c2_single", 4096), ("set_a", 4096), ("set_b", 16384)],
                         ids=["C2_single_1Mx4K", "C3_glued8_1Mx4K", "C5a_bigtable_1Mx16K"])
def test_full_size_config_properties(pa, torch_cuda, name, length):
    """BASELINE.json configs at FULL size (C2: single Scanner, C3: 8 glued regexps, C5a: a glued table that spills
    LDS -- HBM-resident transitions -- with 16 KiB strings; 2^20 strings each) on the device-generated corpus:
    sampled strings are re-generated on the host and checked bit-exactly against the oracle, and whole-batch
    counters must be consistent with the per-string outputs (a checksum of checksums)."""
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    plants = H.plants_for(big)
    n, seed = 1 << 20, 0x5EED5EED
    buf = torch.empty((n, length), dtype=torch.uint8, device="cuda")
    pa.corpus_fill_device(buf.data_ptr(), seed, 0, n, length, length, plants, torch.cuda.current_stream().cuda_stream)
    idx, fin, cnt = dev_run_strided(torch, t, buf)
    # sampled bit-exact check
    rng = np.random.RandomState(0)
    k = 3000 if length <= 4096 else 700
    sample = np.unique(np.concatenate([rng.randint(0, n, k), np.arange(0, 600 if length <= 4096 else 150),
                                       np.arange(n - 300, n)]))
    for lo in range(0, len(sample), 512):
        sel = sample[lo:lo + 512]
        host = np.stack([ob.corpus_fill(seed, int(s), 1, length, plants)[0] for s in sel])
        oi, of = o.run(host.reshape(-1), np.arange(len(sel) + 1, dtype=np.uint64) * length, threads=4)
        assert (idx[sel] == oi).all() and (fin[sel] == of).all()
    # global consistency: counters == histogram of per-string outputs
    assert cnt[1] == n and cnt[0] == int(fin.sum())
    states, hist = np.unique(idx, return_counts=True)
    per = np.zeros(t.RegexpsCount, dtype=np.uint64)
    for s, h in zip(states.tolist(), hist.tolist()):
        assert t.Final(s) or len(t.AcceptedRegexps(s)) == 0   # only Final states accept (multi.h:149-158)
        for r in t.AcceptedRegexps(s):
            per[r] += h
    assert (per == cnt[2:]).all()
    # plants: string s carries witness (s % (P+1)) - 1, so every regexp is matched by about n/(P+1) strings at least
    nplants = len(big["witnesses_hex"])
    if name == "set_a":          # witness r is a witness of regexp r
        for r in range(t.RegexpsCount):
            assert cnt[2 + r] >= n // (nplants + 1) - 1
    tails = sum(1 for x in big["witness_at_tail"] if x)
    assert cnt[0] >= (n // (nplants + 1)) * tails - nplants   # every tail-anchored witness is a match
    assert len(states) >= (8 if t.RegexpsCount >= 8 else 2)


@pytest.mark.parametrize("name", ["set_d", "set_b", "set_a"])
def test_adapt_promotes_visited_rows_and_keeps_results(pa, torch_cuda, name, cfg):
    """pire_hip_table_adapt(): after one representative batch the rows the data really visits move into LDS.
    Results must be bit-identical before and after; the trap counter must collapse.  The table starts from a prior
    that knows nothing (PIRE_HIP_PRIOR_FLAT: dense rows = the first 255 states by index) so that there is something to
    learn whatever the shipped prior already gets right."""
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    cfg.set(prior_flat="1")
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.layout()                                   # ranks the rows now, under the knob
    cfg.unset("prior_flat")
    n, length = 8192, 2048
    data = ob.corpus_fill(77, 0, n, length, H.plants_for(big), threads=4)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    d = torch.as_tensor(data, device="cuda")
    gi, gf, cnt = dev_run_strided(torch, t, d)
    assert (gi == oi).all() and (gf == of).all()
    changed = t.adapt()
    first_traps = t.info.last_trap_samples
    assert first_traps > 0 and changed > 0
    gi2, gf2, cnt2 = dev_run_strided(torch, t, d)
    assert (gi2 == oi).all() and (gf2 == of).all() and (cnt2 == cnt).all()
    t.adapt()
    assert t.info.last_trap_samples * 20 < first_traps, "adapted table must trap at least 20x less on the same data"
    gi3, gf3, _ = dev_run_strided(torch, t, d)
    assert (gi3 == oi).all() and (gf3 == of).all()
    # the host-side view stays in the reference's numbering
    assert t.Final(int(oi[0])) == bool(of[0])


def test_concurrent_host_threads_share_one_table(pa, torch_cuda):
    """A table handle is immutable and shareable between host threads: four threads, each on its own
    stream, run ragged and tiled batches on ONE table at the same time; every result must equal the oracle's."""
    import threading

    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(12)
    jobs = []
    for k in range(4):
        strings = H.random_strings(rng, 3000, 300, b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet")
        text, offs = H.pack(strings)
        jobs.append((text, offs, o.run(text, offs, threads=2)))
    n, length = 2048, 512
    data = ob.corpus_fill(99, 0, n, length, H.plants_for(big), threads=4)
    ref_tiled = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    errors = []

    def worker(k):
        try:
            stream = torch.cuda.Stream()
            text, offs, (oi, of) = jobs[k]
            d = torch.as_tensor(np.array(text), device="cuda")
            do = torch.as_tensor(offs.astype(np.int64), device="cuda")
            dd = torch.as_tensor(data, device="cuda")
            idx = torch.empty(len(offs) - 1, dtype=torch.int32, device="cuda")
            fin = torch.empty(len(offs) - 1, dtype=torch.uint8, device="cuda")
            idx2 = torch.empty(n, dtype=torch.int32, device="cuda")
            fin2 = torch.empty(n, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for _ in range(20):
                t.run_device(d.data_ptr(), do.data_ptr(), len(offs) - 1, BE, idx.data_ptr(), fin.data_ptr(), 0, 0,
                             stream.cuda_stream)
                t.run_strided_device(dd.data_ptr(), n, length, length, BE, idx2.data_ptr(), fin2.data_ptr(), 0, 0,
                                     stream.cuda_stream)
                stream.synchronize()
                if not ((idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()):
                    errors.append("ragged mismatch in thread %d" % k)
                if not ((idx2.cpu().numpy().astype(np.uint32) == ref_tiled[0]).all() and
                        (fin2.cpu().numpy() == ref_tiled[1]).all()):
                    errors.append("tiled mismatch in thread %d" % k)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["set_d", "set_a"])
def test_auto_adapt_reranks_at_a_launch_boundary(pa, torch_cuda, name, cfg):
    """pire_hip_config.auto_adapt (the library's default): a table whose scans keep leaving the dense rows re-ranks
    itself at the next launch -- no call by the user (a drop-in user of BatchRunner never makes one).  The
    table starts from a prior that knows nothing; the first launch traps, the second notices (host-visible trap total),
    adapts and runs on the new image; results identical throughout, traps collapse, and with the policy off nothing
    changes by itself."""
    torch = torch_cuda
    big = [b for b in H.big_sets() if b["name"] == name][0]
    blob = H.load_blob(big["blob"])
    n, length = 8192, 2048
    data = ob.corpus_fill(78, 0, n, length, H.plants_for(big), threads=4)
    o = ob.OracleScanner(blob)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    d = torch.as_tensor(data, device="cuda")
    # (policy, host-pointer calls, adapts by itself): the default re-ranks at the start of a host-pointer call (an adaptation
    # there drains the device); inside a call that only enqueues work it never waits -- 3 (round 5's default): no adaptation at
    # all there, 0 (since round 6): in the background (tests/test_background_adapt.py; whether one has been swapped in after four
    # launches is a matter of timing: None); 2 = at every launch boundary, draining
    for policy, host_calls, expect in ((1, False, False), (3, False, False), (0, False, None), (2, False, True), (0, True, True), (1, True, False)):
        cfg.set(prior_flat=1, auto_adapt=policy)
        t = pa.Table(blob)
        t.layout()                                   # ranks the rows now, under the knob
        cfg.set(prior_flat=0)
        rows_before = set(t.layout()[0][:t.info.hot_states].tolist())
        for launch in range(4):
            if host_calls:
                gi, gf = t.run_strided_host(data)
            else:
                gi, gf, cnt = dev_run_strided(torch, t, d)
            assert (gi == oi).all() and (gf == of).all(), (policy, host_calls, launch)
        info = t.refresh_info()
        rows_after = set(t.layout()[0][:info.hot_states].tolist())
        if expect is None:
            assert info.adaptations <= 4
        elif expect:
            assert 1 <= info.adaptations <= 3, info.adaptations
            assert rows_after != rows_before
            t.adapt()                                # reads the counters of the launches since the last re-ranking
            # the traps have collapsed (counts of a few dozen since round 6: a sample is the state in front of ONE drawn step of a
            # re-walked chunk, if that has no row -- not every re-walked chunk that ends outside the rows)
            assert t.info.last_trap_samples * 8 < info.last_trap_samples + 20
        else:
            assert info.adaptations == 0 and rows_after == rows_before, (policy, host_calls)
        gi, gf, cnt = dev_run_strided(torch, t, d)
        assert (gi == oi).all() and (gf == of).all()


@pytest.mark.gpu
def test_auto_adapt_under_concurrent_launches(pa, cfg):
    """The automatic re-ranking replaces the device images while other host threads may hold their pointers: the
    old images are retired, not freed (table.cpp RetireAllDeviceTables), and the host layout is copied under a shared
    lock.  Six threads, one table that starts from the know-nothing prior, ragged and fixed-length calls: every result
    equals the oracle's while the table adapts itself underneath."""
    import threading

    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    blob = H.load_blob(big["blob"])
    cfg.set(prior_flat=1, auto_adapt=0, auto_adapt_min_traps=8)
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.layout()
    cfg.set(prior_flat=0)
    jobs = []
    for k in range(6):
        data = ob.corpus_fill(100 + k, 0, 2048, 512, H.plants_for(big), threads=2)
        want = o.run(data.reshape(-1), np.arange(2049, dtype=np.uint64) * 512, threads=2)
        jobs.append((data, want))
    errors = []

    def worker(k):
        try:
            data, want = jobs[k]
            for rep in range(12):
                gi, gf = t.run_strided_host(data)
                if not ((gi == want[0]).all() and (gf == want[1]).all()):
                    errors.append((k, rep))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert t.refresh_info().adaptations >= 1


@pytest.mark.gpu
def test_ragged_work_counters_survive_thousands_of_launches(pa, torch_cuda):
    """The ragged kernels take string ranges from a counter that has to be zero at launch; no launch pays a memset:
    launch k uses slot k % 1024 and its last block out puts the slot back to zero (internal.h WorkSlotOf).  More than
    two full rounds of launches on one table, two batches of different sizes alternating, two streams: every launch
    must hand out every string exactly once."""
    torch = torch_cuda
    from pire_amd import binding as pb

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(77)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet"
    batches = []
    for n in (300, 1500):
        strings = H.random_strings(rng, n, 150, alphabet)
        text, offs = H.pack(strings)
        d = torch.as_tensor(np.array(text), device="cuda")
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        want = torch.as_tensor(o.run(text, offs)[0].astype(np.int64), device="cuda").to(torch.int32)
        batches.append((n, d, do, want, torch.empty(n, dtype=torch.int32, device="cuda")))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    mismatches = torch.zeros(1, dtype=torch.int64, device="cuda")
    for k in range(2300):
        n, d, do, want, idx = batches[k & 1]
        st = streams[(k >> 1) & 1]
        with torch.cuda.stream(st):
            idx.fill_(-1)
            t.run_device(d.data_ptr(), do.data_ptr(), n, 3, idx.data_ptr(), 0, 0, 0, st.cuda_stream)
            mismatches += (idx != want).sum()
        if k % 100 == 99:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    assert pb.last_kernel() == "ragged"
    assert int(mismatches.item()) == 0


@pytest.mark.gpu
def test_ragged_work_counters_with_mixed_kinds_of_launches(pa, torch_cuda):
    """every launch of a table takes a slot NUMBER, but only ragged kernels use the slot's counter.
    Ragged, tiled, small generic, empty and flag-forced generic launches interleaved on one table for more than two
    rounds of the 1 024 slots: a slot a ragged launch used must be clean again for the next ragged launch that comes
    round to it, whatever went in between (the last block of a ragged launch zeroes its own slot)."""
    torch = torch_cuda
    from pire_amd import binding as pb

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(78)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet"

    def batch(n, max_len):
        strings = H.random_strings(rng, n, max_len, alphabet)
        text, offs = H.pack(strings)
        d = torch.as_tensor(np.array(text), device="cuda")
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        want = torch.as_tensor(o.run(text, offs)[0].astype(np.int64), device="cuda").to(torch.int32)
        return n, d, do, want, torch.empty(n, dtype=torch.int32, device="cuda")

    ragged, small = batch(1200, 150), batch(40, 60)               # ragged kernel / generic kernel (n < 256)
    n_t, len_t = 128, 256
    rec = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=(n_t, len_t)).astype(np.uint8)
    want_t = torch.as_tensor(o.run(rec.reshape(-1), np.arange(n_t + 1, dtype=np.uint64) * len_t)[0].astype(np.int64),
                             device="cuda").to(torch.int32)
    drec = torch.as_tensor(rec, device="cuda")
    idx_t = torch.empty(n_t, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    mism = torch.zeros(1, dtype=torch.int64, device="cuda")
    kinds = set()
    # the period (7) is coprime to the 1 024 slots: over 2 500 launches every slot sees every kind of neighbour
    for k in range(2500):
        kind = k % 7
        if kind in (0, 3):
            n, d, do, want, idx = ragged
            idx.fill_(-1)
            t.run_device(d.data_ptr(), do.data_ptr(), n, 3, idx.data_ptr(), 0, 0, 0, stream)
            mism += (idx != want).sum()
        elif kind in (1, 5):
            idx_t.fill_(-1)
            t.run_strided_device(drec.data_ptr(), n_t, len_t, len_t, 3, idx_t.data_ptr(), 0, 0, 0, stream)
            mism += (idx_t != want_t).sum()
        elif kind == 2:
            n, d, do, want, idx = small
            idx.fill_(-1)
            t.run_device(d.data_ptr(), do.data_ptr(), n, 3, idx.data_ptr(), 0, 0, 0, stream)
            mism += (idx != want).sum()
        elif kind == 4:
            n, d, do, want, idx = ragged
            t.run_device(d.data_ptr(), do.data_ptr(), 0, 3, idx.data_ptr(), 0, 0, 0, stream)   # n == 0: no launch
        else:
            n, d, do, want, idx = ragged
            idx.fill_(-1)
            t.run_device(d.data_ptr(), do.data_ptr(), n, 3 | pb.FLAG_GENERIC, idx.data_ptr(), 0, 0, 0, stream)
            mism += (idx != want).sum()
        if k < 14:
            kinds.add(pb.last_kernel())
        if k % 250 == 249:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    assert {"ragged", "tiled", "generic"} <= kinds, kinds
    assert int(mism.item()) == 0


@pytest.mark.gpu
def test_empty_fixed_length_records_in_host_mode(pa):
    """n > 0 records of length 0 with stride 0 and a null text pointer (what
    Table.run_strided_host(np.zeros((n, 0))) passes) reached a division by the stride.  Every record is the empty
    string: Begin() then End() from the initial state (pire_ut.cpp:832-837 semantics)."""
    for case in H.all_cases()[:6]:
        blob = H.load_blob(case["blob"])
        t, o = pa.Table(blob), ob.OracleScanner(blob)
        for n in (1, 5, 3000):
            idx, fin = t.run_strided_host(np.zeros((n, 0), dtype=np.uint8))
            oi, of = o.run(np.zeros(0, np.uint8), np.zeros(n + 1, dtype=np.uint64))
            assert (idx == oi).all() and (fin == of).all(), case["name"]


@pytest.mark.gpu
def test_concurrent_host_pointer_calls_share_the_staging_pool(pa, cfg):
    """Host-pointer calls from several threads at once, ragged and fixed-length, small and cut into chunks
    (PIRE_HIP_HOST_CHUNK_BYTES): the pooled staging arenas of api.cpp are taken, grown and returned under contention;
    every result must equal the oracle's, repeatedly."""
    import threading

    cfg.set(host_chunk_bytes=str(256 * 1024))
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(2)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCXYZ@Qnet"
    jobs = []
    for k in range(6):
        if k % 2 == 0:
            strings = H.random_strings(rng, 500 + 1500 * k, 40 + 150 * k, alphabet)
            text, offs = H.pack(strings)
            jobs.append(("ragged", np.array(text), offs, o.run(text, offs, threads=2)))
        else:
            n, length = 300 * k, 256 * k
            data = ob.corpus_fill(7 + k, 0, n, length, H.plants_for(big), threads=2)
            jobs.append(("strided", data, None,
                         o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=2)))
    errors = []

    def worker(k):
        try:
            kind, data, offs, (oi, of) = jobs[k]
            for rep in range(6):
                gi, gf = t.run(data, offs)[:2] if kind == "ragged" else t.run_strided_host(data)[:2]
                if not ((gi == oi).all() and (gf == of).all()):
                    errors.append((k, rep, int((gi != oi).sum())))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


@pytest.mark.gpu
def test_checked_kernel_build_confirms_the_early_out(pa, torch_cuda, cfg):
    """PIRE_HIP_CHECKED=1 (the analogue of the reference's ValidateSkip, multi.h:925-934): the wave-wide early-out is
    only noted, the text is walked to the end, and lanes whose state still moved are counted.  On a batch in which every
    wave does go absorbing (an unanchored pattern matched early in every string) the count must be 0 and the results
    those of the ordinary kernel and of the oracle; the ordinary kernel must really have been faster on it."""
    torch = torch_cuda
    case = [c for c in H.all_cases() if c["name"] == "string"][0]      # /abc/, surrounded: absorbing once it matched
    blob = H.load_blob(case["blob"])
    o = ob.OracleScanner(blob)
    n, length = 4096, 2048
    rng = np.random.RandomState(8)
    data = rng.choice(np.frombuffer(b"abdefgh ", dtype=np.uint8), size=(n, length)).astype(np.uint8)
    absorbing = b"xxabc"
    idx = int(o.run_strings([absorbing], flags=ob.FLAG_BEGIN)[0][0])
    assert all(o.next(idx, c) == idx for c in range(256)) and o.final(idx)
    data[:, :len(absorbing)] = np.frombuffer(absorbing, dtype=np.uint8)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    d = torch.as_tensor(data, device="cuda")
    t = pa.Table(blob)
    gi, gf, cnt = dev_run_strided(torch, t, d)
    assert (gi == oi).all() and (gf == of).all()
    from pire_amd import binding as pb

    assert pb.last_kernel_symbol().endswith("nt,5>")   # the default instantiation (tiled.hip LaunchTiled)
    cfg.set(checked="1")
    ci, cf, ccnt = dev_run_strided(torch, t, d)
    assert "checked" in pb.last_kernel_symbol()
    assert (ci == oi).all() and (cf == of).all() and (ccnt == cnt).all()
    assert t.check_failures() == 0
    assert t.check_failures() == 0      # cleared by the read, and nothing ran in between


@pytest.mark.gpu
def test_host_pointer_mode_in_chunks_equals_one_shot(pa, torch_cuda, cfg):
    """The host-pointer mode cuts big batches into chunks of whole strings that go H2D -> scan -> D2H on alternating
    streams of a pooled staging arena (api.cpp RunHostPipelined).  With a tiny chunk size (knob) a small batch takes
    dozens of chunks: ragged strings with empty ones, strings longer than most chunks' average, offsets that do not
    start at 0, resume states, counters accumulated into; fixed-length records too.  Everything must equal the oracle
    and the one-shot path."""
    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    blob = H.load_blob(big["blob"])
    t, o = pa.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(41)
    strings = H.random_strings(rng, 5000, 300, b"abcdeaxHedInrTail hello w") + [b""] * 7 + \
        [bytes(rng.choice(np.frombuffer(b"abcx ", dtype=np.uint8), size=9000))] + H.random_strings(rng, 2000, 40, b"adexx")
    rng.shuffle(strings)
    text, offs = H.pack(strings)
    pad = np.frombuffer(b"#" * 37, dtype=np.uint8)
    text2 = np.concatenate([pad, text])           # offsets not starting at 0, odd alignment
    offs2 = offs + np.uint64(len(pad))
    init = rng.randint(0, o.size, size=len(strings)).astype(np.uint32)
    want = {}
    for name, kw in (("plain", {}), ("resume", {"init_idx": init})):
        want[name] = o.run(text2, offs2, threads=4, **kw)
    cfg.set(host_one_shot="1")
    one = t.run(text2, offs2, counts=True)
    cfg.unset("host_one_shot")
    for chunk in ("4096", "20000", "65536"):
        cfg.set(host_chunk_bytes=chunk)
        gi, gf, cnt = t.run(text2, offs2, counts=True)
        assert (gi == want["plain"][0]).all() and (gf == want["plain"][1]).all(), chunk
        assert (gi == one[0]).all() and (cnt == one[2]).all()
        ri, rf = t.run(text2, offs2, init_idx=init)
        assert (ri == want["resume"][0]).all() and (rf == want["resume"][1]).all(), chunk
    # fixed-length records: 3 000 x 512 B in chunks of 64 KiB
    data = ob.corpus_fill(5, 0, 3000, 512, H.plants_for(big), threads=4)
    oi, of = o.run(data.reshape(-1), np.arange(3001, dtype=np.uint64) * 512, threads=4)
    cfg.set(host_chunk_bytes="65536")
    si, sf = t.run_strided_host(data)
    assert (si == oi).all() and (sf == of).all()
