#!/usr/bin/env python3
"""CapturingScanner throughput: pire_hip_capture_run with device pointers, 2^20 log-line-like strings."""
import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

stream = torch.cuda.current_stream().cuda_stream
m = 1 << 20
rng = np.random.RandomState(3)
lens = rng.randint(64, 1024, size=m).astype(np.uint64)
offs = np.zeros(m + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
total = int(offs[-1])
alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz    .,=0123456789/'\";_", dtype=np.uint8)
text = alphabet[rng.randint(0, len(alphabet), size=total)].astype(np.uint8)
for w in (b"google_id = 'abc123';", b"=12345x", b"/to-match-with"):
    pos = offs[:-1][::7][: m // 7].astype(np.int64) + 10
    pos = pos[pos + 32 < total]
    for k, ch in enumerate(w):
        text[pos + k] = ch
d = torch.as_tensor(text, device="cuda")
do = torch.as_tensor(offs.astype(np.int64), device="cuda")
idx = torch.empty(m, dtype=torch.int32, device="cuda")
fin = torch.empty(m, dtype=torch.uint8, device="cuda")
bg = torch.empty(m, dtype=torch.int64, device="cuda")
en = torch.empty(m, dtype=torch.int64, device="cuda")
for case in H.golden()["capturing"]:
    blob = H.load_blob(case["blob"])
    t, o = pire_amd.CountingTable(blob, 0), ob.OracleCountingScanner(blob, 0)
    best = 1e9
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t.capture_device(d.data_ptr(), do.data_ptr(), m, 3, idx.data_ptr(), fin.data_ptr(), bg.data_ptr(), en.data_ptr(), stream)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    k = 4096
    oi, of, oc, obg, oen = o.capture(text[:int(offs[k])], offs[:k + 1], flags=3)
    ok = bool((bg[:k].cpu().numpy() == obg).all() and (en[:k].cpu().numpy() == oen).all() and (fin[:k].cpu().numpy() == of).all())
    print("capture %-16s (%s, %d states): %d strings, %.3f GiB: %.3f ms -> %.1f GB/s; captured in %.1f%% of the strings; parity(first %d) %s"
          % (case["name"], case["pattern"], case["states"], m, total / 2**30, best, total / best / 1e6,
             100.0 * float(((bg >= 0) & (en >= 0)).float().mean().item()), k, ok))
