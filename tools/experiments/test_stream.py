"""The stream kernel (stream.hip, PIRE_HIP_RUN_SHORT): an offset batch walked as one contiguous text cut into spans,
every lane answering for the strings that START in its span.  Checked against the oracle on what the span logic can get
wrong: strings that end exactly at chunk / tile / span / task borders, strings shorter than a chunk, empty strings (alone,
in runs, at the very start and the very end), a string longer than many spans among short ones, a batch that does not
start at offset 0 or at an aligned address, the last bytes of the buffer, tasks with more strings than the LDS slice
holds (the per-string fallback), counters, every flag combination, a table that traps."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu

ALPHABET = b"ABCDEFGHIJKLMNOPQRSTUVWXYZ hello wd0123456789-() @net"


def run_stream(t, o, strings, flags=3, lead=0, counts=True, span=None, monkeypatch=None):
    import torch
    from pire_amd import binding as pb

    text, offs = H.pack(strings)
    text = np.asarray(text, dtype=np.uint8)
    rng = np.random.RandomState(len(strings) + lead)
    buf = rng.randint(0, 256, size=lead + text.size + 64).astype(np.uint8)
    buf[lead:lead + text.size] = text
    d = torch.as_tensor(buf, device="cuda")
    do = torch.as_tensor((np.asarray(offs, dtype=np.uint64) + np.uint64(lead)).astype(np.int64), device="cuda")
    n = len(strings)
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    idx.fill_(-1)
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    fin.fill_(0xA5)
    cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t.run_device(d.data_ptr(), do.data_ptr(), n, flags | pb.FLAG_SHORT, idx.data_ptr(), fin.data_ptr(),
                 cnt.data_ptr() if counts else 0, 0, s)
    torch.cuda.synchronize()
    assert pb.last_kernel() == "stream"
    oi, of = o.run(*ob.pack_strings(strings), flags=flags, threads=4)
    oc = np.zeros(t.RegexpsCount + 2, dtype=np.uint64)     # [final, strings, per regexp] as the kernels count them
    oc[0], oc[1] = int((of != 0).sum()), n
    states, times = np.unique(oi, return_counts=True)
    for st, k in zip(states, times):
        for r in o.accepted(int(st)):
            oc[2 + r] += np.uint64(k)
    gi, gf = idx.cpu().numpy().astype(np.uint32), fin.cpu().numpy()
    bad = np.nonzero(gi != oi)[0]
    assert bad.size == 0, (len(strings), lead, flags, bad[:8], [len(strings[i]) for i in bad[:8]])
    assert (gf == of).all(), (len(strings), lead, flags)
    if counts:
        assert (cnt.cpu().numpy().astype(np.uint64) == oc).all(), (cnt.cpu().numpy(), oc)


@pytest.fixture(scope="module")
def table():
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    return t, o


def mix(rng, n, lo, hi):
    a = np.frombuffer(ALPHABET, dtype=np.uint8)
    return [a[rng.randint(0, len(a), size=int(rng.randint(lo, hi)))].tobytes() for _ in range(n)]


def test_urls_and_log_lines(table):
    t, o = table
    rng = np.random.RandomState(1)
    for lead in (0, 48, 1, 1000):
        run_stream(t, o, mix(rng, 20000, 20, 200), lead=lead)
    run_stream(t, o, mix(rng, 6000, 64, 1024), lead=16)
    for flags in (0, 1, 2):
        run_stream(t, o, mix(rng, 5000, 0, 300), flags=flags, lead=7)


def test_border_lengths(table):
    """Lengths around every border the kernel has: 16 (chunk), 128 (tile), 1024 (span), 65536 (task)."""
    t, o = table
    rng = np.random.RandomState(2)
    a = np.frombuffer(ALPHABET, dtype=np.uint8)
    special = [0, 1, 2, 15, 16, 17, 31, 32, 33, 127, 128, 129, 255, 256, 1023, 1024, 1025, 2047, 2048, 4096]
    strings = []
    for i in range(9000):
        k = special[i % len(special)] if i % 3 == 0 else int(rng.randint(0, 260))
        strings.append(a[rng.randint(0, len(a), size=k)].tobytes())
    for lead in (0, 16, 127, 128):
        run_stream(t, o, strings, lead=lead)
    # every string exactly 16 / 128 / 1024 bytes: every boundary on a border
    for k in (16, 128, 1024):
        run_stream(t, o, [a[rng.randint(0, len(a), size=k)].tobytes() for _ in range(2000)], lead=0)


def test_empty_and_tiny_strings(table):
    t, o = table
    rng = np.random.RandomState(3)
    strings = mix(rng, 3000, 0, 6)                       # several ends per chunk, many empty
    run_stream(t, o, strings, lead=5)
    run_stream(t, o, [b""] * 5000, lead=0)               # no text at all
    run_stream(t, o, [b""] * 700 + mix(rng, 3000, 10, 90) + [b""] * 900, lead=3)      # runs of empties at both ends
    run_stream(t, o, mix(rng, 40000, 0, 12), lead=0)     # > 1024 string starts per task: the per-string fallback
    s2 = mix(rng, 4000, 30, 120)
    for i in range(0, len(s2), 7):
        s2[i] = b""
    run_stream(t, o, s2, lead=64)


def test_long_strings_among_short_ones(table):
    t, o = table
    rng = np.random.RandomState(4)
    strings = mix(rng, 6000, 20, 200)
    a = np.frombuffer(ALPHABET, dtype=np.uint8)
    for pos, k in ((10, 5000), (2000, 70000), (2001, 1024), (5999, 300000), (3000, 65536)):
        strings[pos] = a[rng.randint(0, len(a), size=k)].tobytes()
    run_stream(t, o, strings, lead=32)


def test_a_table_that_traps(table):
    """set_d on prose-like text leaves the dense rows often before adapt(): the exact routine with boundaries in it."""
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(5)
    text = b"the quick brown fox jumps over the lazy dog, http://example.com/a/b?c=d&e=f 0123 user@example.net "
    a = np.frombuffer(text, dtype=np.uint8)
    strings = [a[rng.randint(0, len(a), size=int(rng.randint(0, 180)))].tobytes() for _ in range(15000)]
    run_stream(t, o, strings, lead=9)
