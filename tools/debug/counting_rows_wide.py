#!/usr/bin/env python3
"""CountingRowKernel with 5-8 regexps (four counter registers) against the 16-bit-entry kernel: parity and time."""
import numpy as np
import torch
import pire_amd
from pire_amd import binding as pb
from oracle import binding as ob

res_ = ["[a-z]+", "http", "abc", "[0-9]+", "e", "th", "ing"]
seps = ["\\s", ".*", ".*", "\\s", ".*", ".*", ".*"]
for kind in (1, 0):
    blob = ob.RefCountingScanner.compile(kind, res_, seps).save()
    t = pire_amd.CountingTable(blob, kind)
    m = 1 << 20
    rng = np.random.RandomState(3)
    lens = rng.randint(64, 1024, size=m).astype(np.uint64)
    offs = np.zeros(m + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz    .,:/http0123", dtype=np.uint8)
    text = alphabet[rng.randint(0, len(alphabet), size=total)].astype(np.uint8)
    d = torch.as_tensor(text, device="cuda")
    do = torch.as_tensor(offs.astype(np.int64), device="cuda")
    R = t.RegexpsCount
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for variant in (1, 0):
        idx = torch.empty(m, dtype=torch.int32, device="cuda")
        res = torch.empty((m, R), dtype=torch.int32, device="cuda")
        with pb.config(counting_variant=variant):
            for _ in range(30):
                t.run_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), res.data_ptr(), stream)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in ev:
                a.record()
                t.run_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), res.data_ptr(), stream)
                b.record()
            torch.cuda.synchronize()
            ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            out[variant] = (idx.cpu().numpy(), res.cpu().numpy())
            print("kind %d, %d regexps, %d states: variant %d [%s] %.3f ms -> %.1f GB/s" % (kind, R, t.Size, variant, pb.last_kernel(), ms, total / ms / 1e6))
    print("   equal results: %s; totals %s" % (bool((out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all()), out[0][1].sum(axis=0).tolist()))
