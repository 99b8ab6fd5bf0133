#!/usr/bin/env python3
"""SlowScanner on a ragged batch (log-line-like strings, device pointers): the list kernel with the strings in the caller's
order and by length class (order.hip).  tools/slow_ragged_case.py [fixture]"""
import ctypes as C
import sys

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

name = sys.argv[1] if len(sys.argv) > 1 else "slow_x40_utf8"
case = [c for c in H.golden()["slow"] if c["name"] == name][0]
blob = H.load_blob(case["blob"])
t = pire_amd.SlowTable(blob)
m = 1 << 18
rng = np.random.RandomState(3)
lens = rng.randint(64, 1024, size=m).astype(np.uint64)
offs = np.zeros(m + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
total = int(offs[-1])
alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz    .,:/x", dtype=np.uint8)
text = alphabet[rng.randint(0, len(alphabet), size=total)].astype(np.uint8)
d = torch.as_tensor(text, device="cuda")
do = torch.as_tensor(offs.astype(np.int64), device="cuda")
fin = torch.empty(m, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def launch():
    pb._check(pb.lib().pire_hip_slow_run(t._h, C.c_void_p(d.data_ptr()), C.c_void_p(do.data_ptr()), m, 3 | pb.FLAG_ON_DEVICE,
                                         C.c_void_p(fin.data_ptr()), None, None, C.c_void_p(stream)))


want = None
for off in (0, 1):
    pb.set_config(no_length_order=off)
    for _ in range(5):
        launch()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for a, b in ev:
        a.record()
        launch()
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    got = fin.cpu().numpy().copy()
    same = True if want is None else bool((got == want).all())
    want = got
    print("slow[%s] %s, %d strings of 64..1023 B, %.3f GiB, strings %s: %.3f ms -> %.1f GB/s; finals %d; equal to the other order: %s"
          % (pb.last_kernel(), name, m, total / 2**30, "in the caller's order" if off else "by length class", ms, total / ms / 1e6,
             int(got.sum()), same))
