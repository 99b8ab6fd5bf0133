#!/usr/bin/env python3
"""Fused ScannerPair pass (pair.hip) against two tiled passes over the same resident text: tools/pair_case.py"""
import numpy as np
import torch
import pire_amd
from pire_amd import binding as pb
from pire_amd import workloads as W

t1 = pire_amd.Table(W.load_blob(W.pattern_set("set_a")["blob"]))
t2 = pire_amd.Table(W.load_blob(W.pattern_set("set_d")["blob"]))
n, L = 1 << 20, 4096
buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
pire_amd.corpus_fill_device(buf.data_ptr(), 0x5EED5EED, 0, n, L, L, W.plants_for(W.pattern_set("set_a")), stream)
i1 = torch.empty(n, dtype=torch.int32, device="cuda"); i2 = torch.empty_like(i1)
f1 = torch.empty(n, dtype=torch.uint8, device="cuda"); f2 = torch.empty_like(f1)
flags = pb.FLAG_BEGIN | pb.FLAG_END


def two():
    t1.run_strided_device(buf.data_ptr(), n, L, L, flags, i1.data_ptr(), f1.data_ptr(), 0, 0, stream)
    t2.run_strided_device(buf.data_ptr(), n, L, L, flags, i2.data_ptr(), f2.data_ptr(), 0, 0, stream)


def fused():
    pb.run_pair_strided_device(t1, t2, buf.data_ptr(), n, L, L, flags, i1.data_ptr(), i2.data_ptr(), f1.data_ptr(), stream)


for _ in range(3):
    two()
torch.cuda.synchronize()
t1.adapt(); t2.adapt()
ref = None
for name, fn in (("two tiled passes", two), ("fused pair pass ", fused), ("two tiled passes", two), ("fused pair pass ", fused)):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(10):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    res = (i1.cpu().numpy().copy(), i2.cpu().numpy().copy())
    if ref is None:
        ref = res
    same = (res[0] == ref[0]).all() and (res[1] == ref[1]).all()
    print("%s: %d x %d B with set_a and set_d: median %.3f ms, min %.3f ms -> %.0f GB/s of text per pair; state indices equal to the first run: %s"
          % (name, n, L, float(np.median(ts)), min(ts), n * L / float(np.median(ts)) / 1e6, same))
