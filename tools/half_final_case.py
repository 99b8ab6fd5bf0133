#!/usr/bin/env python3
"""HalfFinalScanner throughput: pire_hip_run_half_final (device pointers) vs the reference on the host cores."""
import os
import sys
import time

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

name = sys.argv[1] if len(sys.argv) > 1 else "half_5"
case = [c for c in H.golden()["half_final"] if c["name"] == name][0]
blob = H.load_blob(case["blob"])
t = pire_amd.Table(blob)
t.upload()
m, lo, hi = 1 << int(__import__("os").environ.get("HALF_FINAL_LOG2_STRINGS", "20")), 64, 1024
rng = np.random.RandomState(3)
lens = rng.randint(lo, hi, size=m).astype(np.uint64)
offs = np.zeros(m + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
total = int(offs[-1])
alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz    .,", dtype=np.uint8)
text = alphabet[rng.randint(0, len(alphabet), size=total)].astype(np.uint8)
d = torch.as_tensor(text, device="cuda")
do = torch.as_tensor(offs.astype(np.int64), device="cuda")
R = t.RegexpsCount
idx = torch.empty(m, dtype=torch.int32, device="cuda")
fin = torch.empty(m, dtype=torch.uint8, device="cuda")
res = torch.empty((m, R), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
if len(sys.argv) > 2 and sys.argv[2] == "adapt":
    # let a plain Scanner pass over the same text feed the visit counters, then re-rank the dense rows
    t.run_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
    torch.cuda.synchronize()
    print("adapt: rows changed", t.adapt())
    t.run_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
    torch.cuda.synchronize()
    print("adapt: rows changed", t.adapt())
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    t.run_half_final_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), fin.data_ptr(), res.data_ptr(), stream)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = min(ts[1:])
print("half_final[%s] %s (pattern %s, %d states, %d regexps): %d strings, %.3f GiB: %.3f ms -> %.1f GB/s; matches counted: %s"
      % (pb.last_kernel(), name, case["pattern"], t.Size, R, m, total / 2**30, ms, total / ms / 1e6, res.sum(dim=0).tolist()))
# reference on the host (a sample), same bytes
if ob.ref_available():
    r = ob.RefHalfFinalScanner.load(blob)
    k = 1 << 16
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    ri, rf, rr = r.run(text, offs[:k + 1], threads=min(cores, 64))
    dt = time.perf_counter() - t0
    print("reference HalfFinalScanner, %d threads, first %d strings (%.1f MiB): %.3f s -> %.2f GB/s; parity on the sample: %s"
          % (min(cores, 64), k, int(offs[k]) / 2**20, dt, int(offs[k]) / dt / 1e9,
             bool((rr == res[:k].cpu().numpy()).all() and (ri == idx[:k].cpu().numpy().astype(np.uint32)).all())))
