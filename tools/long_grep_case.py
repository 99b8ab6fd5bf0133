#!/usr/bin/env python3
"""One long string under grep-like (unanchored) patterns: the sticky modes ("error was seen") are functions of the walk
from the start state, so the segmented scan walks the text once and derives them (segmented.hip, ModeFunction).
Needs oracle/_ref to compile the scanner.  tools/long_grep_case.py [log2 bytes]"""
import sys

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
blob = ob.RefScanner.compile(["error", "time ?out", "fa+tal"]).save()
t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
t.upload()
total = 1 << lg
rng = np.random.RandomState(5)
a = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz   .,:/", dtype=np.uint8)
host = a[rng.randint(0, len(a), size=1 << 24)]
d = torch.as_tensor(host, device="cuda").repeat(total >> 24)
for frac, word in ((0.3, b" error "), (0.6, b" timeout "), (0.8, b" faaatal ")):
    pos = int(total * frac)
    d[pos:pos + len(word)] = torch.as_tensor(np.frombuffer(word, dtype=np.uint8).copy(), device="cuda")
idx = torch.empty(1, dtype=torch.int32, device="cuda")
fin = torch.empty(1, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
run = lambda: t.run_strided_device(d.data_ptr(), 1, total, total, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)


def timeit(reps=5):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        x.record()
        run()
        y.record()
        torch.cuda.synchronize()
        best = min(best, x.elapsed_time(y))
    return best


for _ in range(3):   # learn the modes
    run()
    torch.cuda.synchronize()
ms = timeit()
sym = pb.last_kernel_symbol()
pb.set_config(segment_no_derive=1)
ms2 = timeit()
sym2 = pb.last_kernel_symbol()
pb.set_config(segment_no_derive=0)
k = 1 << 26
oi, of = o.run(d[:k].cpu().numpy(), np.array([0, k], dtype=np.uint64))
idx2 = torch.empty(1, dtype=torch.int32, device="cuda")
t.run_strided_device(d.data_ptr(), 1, k, k, 3, idx2.data_ptr(), fin.data_ptr(), 0, 0, stream)
torch.cuda.synchronize()
print("grep-like patterns, one string of %d B: %.3f ms -> %.1f GB/s (%s); every mode walked: %.3f ms -> %.1f GB/s (%s); parity on the first 64 MiB: %s"
      % (total, ms, total / ms / 1e6, sym, ms2, total / ms2 / 1e6, sym2, int(idx2[0]) == int(oi[0])))
