#!/usr/bin/env python3
"""HalfFinalScanner counting and LongestPrefix/ShortestPrefix: the ragged kernel with actions against the
one-string-per-lane kernels (PIRE_HIP_RUN_GENERIC), device pointers, dense- and sparse-Final scanners."""
import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

stream = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def batch(m, lo, hi, alphabet, seed):
    rng = np.random.RandomState(seed)
    lens = rng.randint(lo, hi, size=m).astype(np.uint64)
    offs = np.zeros(m + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    a = np.frombuffer(alphabet, dtype=np.uint8)
    text = a[rng.randint(0, len(a), size=total)].astype(np.uint8)
    return text, offs, total


LOG = b"abcdefghijklmnopqrstuvwxyz    .,0123456789-/:"
cases = []
g = H.golden()
half5 = [c for c in g["half_final"] if c["name"] == "half_5"][0]
cases.append(("half_5 dense (" + half5["pattern"] + ")", H.load_blob(half5["blob"]), b"abcdefghijklmnopqrstuvwxyz    .,"))
if ob.ref_available():
    words = ["error", "timeout", "get /index", "[0-9]{3}-[0-9]{4}", "warn(ing)?"]
    r = ob.RefHalfFinalScanner.compile(words, [ob.RefHalfFinalScanner.NONGREEDY_SIMPLE] * len(words))
    cases.append(("5 words sparse", r.save(), LOG))

for shape, m, lo, hi in (("log lines 64..1023 B", 1 << 20, 64, 1024), ("URLs 20..199 B", 1 << 22, 20, 200)):
    for name, blob, alphabet in cases:
        t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
        t.upload()
        text, offs, total = batch(m, lo, hi, alphabet, 3)
        d = torch.as_tensor(text, device="cuda")
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        idx = torch.empty(m, dtype=torch.int32, device="cuda")
        fin = torch.empty(m, dtype=torch.uint8, device="cuda")
        res = torch.empty((m, t.RegexpsCount), dtype=torch.int32, device="cuda")
        for adapt in (False, True):
            if adapt:
                for _ in range(2):
                    t.run_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
                    torch.cuda.synchronize()
                    t.adapt()
            for gflag in (0, pb.FLAG_GENERIC):
                ms = timeit(lambda: t.run_half_final_device(d.data_ptr(), do.data_ptr(), m, 3 | gflag, idx.data_ptr(),
                                                            fin.data_ptr(), res.data_ptr(), stream))
                k = 4096
                oi, of, orr = o.run_half_final(text[:int(offs[k])], offs[:k + 1], flags=3)
                ok = bool((res[:k].cpu().numpy().astype(np.uint32) == orr).all())
                print("half_final %-34s %-22s %-18s%s %.3f GiB: %.3f ms -> %7.1f GB/s; matches/KB %.2f; parity(first %d) %s"
                      % (name, shape, pb.last_kernel(), " adapted" if adapt else "", total / 2**30, ms, total / ms / 1e6,
                         float(res.sum().item()) / total * 1000, k, ok))

# prefix searches
pcases = []
big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
pcases.append(("set_a (8 glued, $-anchored)", H.load_blob(big["blob"]), LOG + b"ABCXYZ@() "))
if ob.ref_available():
    pcases.append(("lexer [a-z]+|[0-9]+| +", ob.RefScanner.compile(["[a-z]+|[0-9]+| +"], ["n"]).save(), LOG))
    pcases.append(("surrounded 'error|timeout'", ob.RefScanner.compile(["error|timeout"], [""]).save(), LOG))
for shape, m, lo, hi in (("log lines 64..1023 B", 1 << 18, 64, 1024), ("URLs 20..199 B", 1 << 22, 20, 200)):
    for name, blob, alphabet in pcases:
        t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
        t.upload()
        text, offs, total = batch(m, lo, hi, alphabet, 4)
        d = torch.as_tensor(text, device="cuda")
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        out = torch.empty(m, dtype=torch.int64, device="cuda")
        for longest in (True, False):
            for generic in (False, True):
                ms = timeit(lambda: t.prefix_device(d.data_ptr(), do.data_ptr(), m, longest, out.data_ptr(), stream=stream,
                                                    generic=generic))
                k = 4096
                want = o.prefix(text[:int(offs[k])], offs[:k + 1], longest)
                ok = bool((out[:k].cpu().numpy() == want).all())
                print("%-8s %-30s %-22s %-14s %.3f GiB: %.3f ms -> %7.1f GB/s of text; found %.1f%%; parity(first %d) %s"
                      % ("longest" if longest else "shortest", name, shape, pb.last_kernel(), total / 2**30, ms,
                         total / ms / 1e6, 100.0 * float((out >= 0).float().mean().item()), k, ok))
