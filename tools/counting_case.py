#!/usr/bin/env python3
"""CountingScanner throughput: pire_hip_counting_run (device pointers) vs the reference on the host cores."""
import os
import sys
import time

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from tests import helpers as H

name = sys.argv[1] if len(sys.argv) > 1 else "count_glued3_advanced"
case = [c for c in H.golden()["counting"] if c["name"] == name][0]
blob = H.load_blob(case["blob"])
t = pire_amd.CountingTable(blob, case["kind"])
m = 1 << int(os.environ.get("COUNTING_LOG2_STRINGS", "20"))
rng = np.random.RandomState(3)
# COUNTING_FIXED_LEN=544: every string that long (how much of the time is lanes waiting for the longest string of their wave)
lens = rng.randint(64, 1024, size=m).astype(np.uint64)
if os.environ.get("COUNTING_FIXED_LEN"):
    lens[:] = int(os.environ["COUNTING_FIXED_LEN"])
offs = np.zeros(m + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
total = int(offs[-1])
alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz    .,:/http", dtype=np.uint8)
text = alphabet[rng.randint(0, len(alphabet), size=total)].astype(np.uint8)
d = torch.as_tensor(text, device="cuda")
do = torch.as_tensor(offs.astype(np.int64), device="cuda")
R = t.RegexpsCount
idx = torch.empty(m, dtype=torch.int32, device="cuda")
res = torch.empty((m, R), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
if os.environ.get("NO_LENGTH_ORDER"):
    pire_amd.binding.set_config(no_length_order=1)
generic = pire_amd.binding.FLAG_GENERIC if len(sys.argv) > 2 and sys.argv[2] == "generic" else 0   # the 32-bit kernel alone


def launch():
    t.run_device(d.data_ptr(), do.data_ptr(), m, 3 | generic, idx.data_ptr(), res.data_ptr(), stream)


# settled clocks: ~40 ms of back-to-back launches, then 20 timed ones back to back
for _ in range(50):
    launch()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record()
    launch()
    b.record()
torch.cuda.synchronize()
ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
print("counting[%s] %s (re %s sep %s, %d states x %d letters, %d regexps): %d strings, %.3f GiB: %.3f ms -> %.1f GB/s; totals %s"
      % (pire_amd.binding.last_kernel(), name, case["re"], case["sep"], t.Size, t.LettersCount, R, m, total / 2**30, ms, total / ms / 1e6, res.sum(dim=0).tolist()))
if ob.ref_available():
    r = ob.RefCountingScanner.load(case["kind"], blob)
    k = m   # the whole batch (round 4; rounds 1-3 compared the first 2^17 strings)
    cores = min(os.cpu_count() or 1, 64)
    t0 = time.perf_counter()
    ri, rr = r.run(text, offs[:k + 1], threads=cores)
    dt = time.perf_counter() - t0
    print("reference, %d threads, first %d strings (%.1f MiB): %.3f s -> %.2f GB/s; parity on the sample: %s"
          % (cores, k, int(offs[k]) / 2**20, dt, int(offs[k]) / dt / 1e9,
             bool((rr == res[:k].cpu().numpy()).all() and (ri == idx[:k].cpu().numpy().astype(np.uint32)).all())))
    t0 = time.perf_counter()
    r.run(text, offs[:(k >> 3) + 1], threads=1)
    dt = time.perf_counter() - t0
    print("reference, 1 thread: %.3f GB/s" % (int(offs[k >> 3]) / dt / 1e9))
