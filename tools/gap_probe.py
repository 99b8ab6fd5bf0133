#!/usr/bin/env python3
"""How long an idle gap may be before the GPU's power management restarts its transient (fast launches, a dip, recovery)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pire_amd
from pire_amd import workloads as W
big = W.pattern_set("set_a"); table = pire_amd.Table(W.load_blob(big["blob"])); table.upload()
n, L = 1 << 20, 4096
text = torch.empty((n, L), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
pire_amd.corpus_fill_device(text.data_ptr(), 0x5EED5EED, 0, n, L, L, W.plants_for(big), stream)
idx = torch.empty(n, dtype=torch.int32, device="cuda"); fin = torch.empty(n, dtype=torch.uint8, device="cuda")
def burst(k):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    for a, b in ev:
        a.record(); table.run_strided_device(text.data_ptr(), n, L, L, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream); b.record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ev])
burst(100)
for gap in (0.0, 0.0002, 0.001, 0.003, 0.01, 0.03):
    outs = []
    for rep in range(3):
        burst(60)
        if gap: time.sleep(gap)
        ms = burst(20)
        outs.append("%.4f(max %.3f)" % (ms.mean(), ms.max()))
    print("gap %.4f s after a sync: 20 launches mean %s" % (gap, " ".join(outs)))
# warm-up length needed after a long idle
for w in (5, 10, 20, 30, 50):
    time.sleep(0.3); burst(w); ms = burst(20)
    print("0.3 s idle, %d warm-up launches, sync, then 20: mean %.4f max %.3f" % (w, ms.mean(), ms.max()))
