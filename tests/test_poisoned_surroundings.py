"""The text of a batch embedded in memory that is NOT zero: random bytes before the first string and after the last
one, the batch starting at any distance from a 128-byte line, the output arrays filled with a poison pattern before
the call.  A kernel that lets bytes outside a string reach its result, leaves an output element unwritten, or leans
on a freshly allocated (zeroed) buffer fails here and nowhere else: every other test hands over buffers the driver
has just zeroed.  (Found the hard way: a staging pool that reuses memory made one check of the C++ shim test flaky.)"""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

pytestmark = pytest.mark.gpu

LEADS = (0, 16, 48, 112, 128, 1, 7, 77)


def short_mix(rng, n, alphabet, top=70):
    """The C++ shim test's mix: a few hundred strings of 0..69 characters."""
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return [a[rng.randint(0, len(a), size=int(rng.randint(0, top)))].tobytes() for _ in range(n)]


def long_mix(rng, n, alphabet):
    lens = [int(rng.randint(0, 300)) if i % 37 else int(rng.randint(1000, 5000)) for i in range(n)]
    a = np.frombuffer(alphabet, dtype=np.uint8)
    return [a[rng.randint(0, len(a), size=k)].tobytes() for k in lens]


class Embedded:
    """`strings` packed into the middle of a device buffer of random bytes; two ways to address them:
    pointer = start of the batch and offsets from 0, or pointer = start of the buffer and offsets[0] = lead."""

    def __init__(self, strings, lead, rng, shifted_offsets):
        import torch

        text, offs = H.pack(strings)
        text = np.asarray(text, dtype=np.uint8)
        self.text, self.offs, self.n = text, np.asarray(offs, dtype=np.uint64), len(strings)
        tail = 300
        buf = rng.randint(0, 256, size=lead + text.size + tail).astype(np.uint8)
        buf[lead:lead + text.size] = text
        self.buf = torch.as_tensor(buf, device="cuda")
        if shifted_offsets:
            self.ptr = self.buf.data_ptr()
            self.dev_offs = torch.as_tensor((self.offs + np.uint64(lead)).astype(np.int64), device="cuda")
        else:
            self.ptr = self.buf.data_ptr() + lead
            self.dev_offs = torch.as_tensor(self.offs.astype(np.int64), device="cuda")


def poisoned(shape, dtype):
    import torch

    t = torch.empty(shape, dtype=dtype, device="cuda")
    t.view(torch.uint8).fill_(0xA5)
    return t


def half_blobs():
    g = H.golden()
    return [(c["name"], H.load_blob(c["blob"])) for c in g["half_final"] if c["regexps"] <= 8][:3]


@pytest.mark.parametrize("name,blob", half_blobs(), ids=[b[0] for b in half_blobs()])
def test_half_final_and_prefix_in_poisoned_memory(name, blob, cfg):
    import torch
    import pire_amd

    cfg.set(ragged_act_always="1")
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(2024)
    s = torch.cuda.current_stream().cuda_stream
    for k, lead in enumerate(LEADS):
        for strings in (short_mix(rng, 508, b"abcde "), long_mix(rng, 900, b"abcde w"), short_mix(rng, 100, b"abcde ")):
            e = Embedded(strings, lead, rng, shifted_offsets=bool(k & 1))
            n = e.n
            idx, fin = poisoned(n, torch.int32), poisoned(n, torch.uint8)
            res = poisoned((n, t.RegexpsCount), torch.int32)
            ln = poisoned(n, torch.int64)
            for flags in (3, 0):
                t.run_half_final_device(e.ptr, e.dev_offs.data_ptr(), n, flags, idx.data_ptr(), fin.data_ptr(),
                                        res.data_ptr(), s)
                torch.cuda.synchronize()
                oi, of, orr = o.run_half_final(*ob.pack_strings(strings), flags=flags)
                assert (idx.cpu().numpy().astype(np.uint32) == oi).all(), (name, lead, n, flags)
                assert (fin.cpu().numpy() == of).all(), (name, lead, n, flags)
                got = res.cpu().numpy().astype(np.uint32)
                assert (got == orr).all(), (name, lead, n, flags, np.nonzero((got != orr).any(axis=1))[0][:8])
                idx.view(torch.uint8).fill_(0xA5), fin.fill_(0xA5), res.view(torch.uint8).fill_(0xA5)
            for longest in (True, False):
                t.prefix_device(e.ptr, e.dev_offs.data_ptr(), n, longest, ln.data_ptr(), through_begin=True,
                                through_end=True, stream=s)
                torch.cuda.synchronize()
                want = o.prefix(e.text, e.offs, longest, True, True)
                assert (ln.cpu().numpy() == want).all(), (name, lead, n, longest)
                ln.view(torch.uint8).fill_(0xA5)
                t.suffix_device(e.ptr, e.dev_offs.data_ptr(), n, longest, ln.data_ptr(), stream=s)
                torch.cuda.synchronize()
                want = o.suffix(e.text, e.offs, longest)
                assert (ln.cpu().numpy() == want).all(), (name, lead, n, longest, "suffix")
                ln.view(torch.uint8).fill_(0xA5)


def test_run_in_poisoned_memory():
    import torch
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    rng = np.random.RandomState(7)
    s = torch.cuda.current_stream().cuda_stream
    alphabet = b"ABCDEFGHIJKLMNOPQRSTUVWXYZ hello wd0123456789-() @net"
    for k, lead in enumerate(LEADS):
        for strings in (short_mix(rng, 508, alphabet), long_mix(rng, 900, alphabet), short_mix(rng, 60, alphabet)):
            e = Embedded(strings, lead, rng, shifted_offsets=bool(k & 1))
            n = e.n
            idx, fin = poisoned(n, torch.int32), poisoned(n, torch.uint8)
            for flags in (3, 0):
                t.run_device(e.ptr, e.dev_offs.data_ptr(), n, flags, idx.data_ptr(), fin.data_ptr(), stream=s)
                torch.cuda.synchronize()
                oi, of = o.run(*ob.pack_strings(strings), flags=flags)[:2]
                assert (idx.cpu().numpy().astype(np.uint32) == oi).all(), (lead, n, flags)
                assert (fin.cpu().numpy() == of).all(), (lead, n, flags)
                idx.view(torch.uint8).fill_(0xA5), fin.fill_(0xA5)
    # fixed-length records with a stride, embedded the same way (the tiled kernel and its remainder kernel)
    for lead in (0, 16, 128, 4096 + 48):
        for n, length, stride in ((64 * 20, 256, 256), (64 * 16 + 5, 512, 640), (37, 128, 128)):
            body = rng.randint(0, len(alphabet), size=(n, stride))
            rec = np.frombuffer(alphabet, dtype=np.uint8)[body]
            buf = rng.randint(0, 256, size=lead + n * stride + 300).astype(np.uint8)
            buf[lead:lead + n * stride] = rec.reshape(-1)
            d = torch.as_tensor(buf, device="cuda")
            idx, fin = poisoned(n, torch.int32), poisoned(n, torch.uint8)
            t.run_strided_device(d.data_ptr() + lead, n, length, stride, 3, idx.data_ptr(), fin.data_ptr(), stream=s)
            torch.cuda.synchronize()
            strings = [rec[i, :length].tobytes() for i in range(n)]
            oi, of = o.run(*ob.pack_strings(strings), flags=3)[:2]
            assert (idx.cpu().numpy().astype(np.uint32) == oi).all(), (lead, n, length, stride)
            assert (fin.cpu().numpy() == of).all(), (lead, n, length, stride)


def test_counting_in_poisoned_memory():
    import torch
    import pire_amd

    g = H.golden()
    cases = g.get("counting", [])[:3]
    if not cases:
        pytest.skip("no counting cases in the golden file")
    rng = np.random.RandomState(11)
    s = torch.cuda.current_stream().cuda_stream
    for c in cases:
        blob = H.load_blob(c["blob"])
        t = pire_amd.CountingTable(blob, c["kind"])
        o = ob.OracleCountingScanner(blob, c["kind"])
        for k, lead in enumerate(LEADS[:6]):
            for strings in (short_mix(rng, 508, b"abcde w"), long_mix(rng, 700, b"abcde w")):
                e = Embedded(strings, lead, rng, shifted_offsets=bool(k & 1))
                n = e.n
                idx = poisoned(n, torch.int32)
                res = poisoned((n, t.RegexpsCount), torch.int32)
                t.run_device(e.ptr, e.dev_offs.data_ptr(), n, 3, idx.data_ptr(), res.data_ptr(), s)
                torch.cuda.synchronize()
                oi, orr = o.run(e.text, e.offs, flags=3)
                assert (idx.cpu().numpy().astype(np.uint32) == oi).all(), (c["name"], lead, n)
                assert (res.cpu().numpy().astype(np.uint32).astype(np.uint64) == orr).all(), (c["name"], lead, n)
