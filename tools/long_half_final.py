#!/usr/bin/env python3
"""HalfFinalScanner match counting in few long strings (resident text, host offsets): the segmented scan resolves the
segments' start states, then the segments are counted in parallel."""
import os

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

stream = torch.cuda.current_stream().cuda_stream
total = 1 << int(os.environ.get("LONG_TOTAL_LOG2", "30"))
LOG = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz    .,0123456789-/:", dtype=np.uint8)
rng = np.random.RandomState(5)
text = LOG[rng.randint(0, len(LOG), size=total)].astype(np.uint8)
# plant some words
for w in (b"error", b"timeout", b"warning"):
    pos = rng.randint(0, total - 16, size=total // 4096)
    for k, ch in enumerate(w):
        text[pos + k] = ch
d = torch.as_tensor(text, device="cuda")
cases = []
g = H.golden()
half5 = [c for c in g["half_final"] if c["name"] == "half_5"][0]
cases.append(("half_5 dense (" + half5["pattern"] + ")", H.load_blob(half5["blob"])))
if ob.ref_available():
    words = ["error", "timeout", "get /index", "[0-9]{3}-[0-9]{4}", "warn(ing)?"]
    cases.append(("5 words", ob.RefHalfFinalScanner.compile(words, [ob.RefHalfFinalScanner.NONGREEDY_SIMPLE] * len(words)).save()))


def timeit(fn, reps=3):
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


for name, blob in cases:
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    t.upload()
    for n in (1, 64):
        offs = (np.arange(n + 1, dtype=np.uint64) * (total // n)).astype(np.uint64)
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        fin = torch.empty(n, dtype=torch.uint8, device="cuda")
        res = torch.empty((n, t.RegexpsCount), dtype=torch.int32, device="cuda")
        run = lambda: t.run_half_final_device_host_offsets(d.data_ptr(), offs, 3, idx.data_ptr(), fin.data_ptr(), res.data_ptr(), stream)
        ms = timeit(run)
        kernel = pb.last_kernel()
        # parity on the first string's first 64 MiB is not possible (counts are per string): check the whole first string
        # when it is short enough for the oracle, else the sum over a truncated copy
        k = min(total // n, 64 << 20)
        oi, of, orr = o.run_half_final(text[:k], np.array([0, k], dtype=np.uint64), flags=1)
        t2 = timeit(lambda: t.run_half_final_device_host_offsets(d.data_ptr(), np.array([0, k], dtype=np.uint64), 1, idx.data_ptr(),
                                                                 fin.data_ptr(), res.data_ptr(), stream), reps=1)
        ok = bool((res[0].cpu().numpy().astype(np.uint32) == orr[0]).all())
        print("half_final %-30s %5d x %10d B: %-10s %8.3f ms -> %7.1f GB/s; parity on the first %d MiB as one string: %s (counts %s)"
              % (name, n, total // n, kernel, ms, total / ms / 1e6, k >> 20, ok, orr[0].tolist()), flush=True)
