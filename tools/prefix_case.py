#!/usr/bin/env python3
"""LongestPrefix / ShortestPrefix throughput: pire_hip_prefix with device pointers vs the reference on the host.  set_a table, 2^18 strings of 64..1023 B."""
import os
import time

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
blob = H.load_blob(big["blob"])
t = pire_amd.Table(blob)
t.upload()
import sys
LOG2 = int(os.environ.get("PREFIX_LOG2_STRINGS", "18"))   # 18: the batch of profiles/r04_final_prefix.log (0.133 GiB); 21: 1.06 GiB
SETTLE = int(os.environ.get("PREFIX_SETTLE", "0"))        # untimed launches in front of the timed ones (settled clocks)
m = 1 << LOG2
rng = np.random.RandomState(4)
lens = rng.randint(64, 1024, size=m).astype(np.uint64)
offs = np.zeros(m + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
total = int(offs[-1])
text = ob.corpus_fill(0x5EED5EED, 0, (total + 4095) // 4096, 4096, H.plants_for(big), threads=8).reshape(-1)[:total]
d = torch.as_tensor(text, device="cuda")
do = torch.as_tensor(offs.astype(np.int64), device="cuda")
dout = torch.empty(m, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
import sys
if len(sys.argv) > 1 and sys.argv[1] == "adapt":
    # a plain Scanner pass over the same text feeds the visit counters, then the dense rows are re-ranked
    idx = torch.empty(m, dtype=torch.int32, device="cuda")
    fin = torch.empty(m, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        t.run_device(d.data_ptr(), do.data_ptr(), m, 0, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
        torch.cuda.synchronize()
        print("adapt: rows changed", t.adapt())
if len(sys.argv) > 1 and sys.argv[1] == "adapt_by_prefix":
    # the searches' own visit samples: two prefix calls, then adapt() (what the automatic adaptation does for a caller of
    # pire_hip_prefix alone)
    for _ in range(2):
        for _k in range(2):
            t.prefix_device(d.data_ptr(), do.data_ptr(), m, True, dout.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        print("adapt after prefix calls only: rows changed", t.adapt())
for longest in (True, False):
    best = 1e9
    for _ in range(SETTLE):
        t.prefix_device(d.data_ptr(), do.data_ptr(), m, longest, dout.data_ptr(), stream=stream)
    for _ in range(4 if not SETTLE else 10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t.prefix_device(d.data_ptr(), do.data_ptr(), m, longest, dout.data_ptr(), stream=stream)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    out = dout.cpu().numpy()
    name = "LongestPrefix" if longest else "ShortestPrefix"
    scanned = int(np.where(out >= 0, out, lens.astype(np.int64)).sum()) if not longest else total
    print("%s (%s): %d strings, %.3f GiB: kernel %.3f ms -> %.1f GB/s of text (%.1f GB/s of bytes actually walked)"
          % (name, pb.last_kernel(), m, total / 2**30, best, total / best / 1e6, scanned / best / 1e6))
    if ob.ref_available():
        r = ob.RefScanner.load(blob)
        k = 1 << 15
        t0 = time.perf_counter()
        ref = r.prefix(text, offs[:k + 1], longest)
        dt = time.perf_counter() - t0
        print("  reference %s, 1 thread, first %d strings (%.1f MiB): %.3f s -> %.3f GB/s; parity on the sample: %s"
              % (name, k, int(offs[k]) / 2**20, dt, int(offs[k]) / dt / 1e9, bool((ref == out[:k]).all())))
