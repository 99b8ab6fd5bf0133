#!/usr/bin/env python3
"""Per-launch kernel time of the headline workload from a cold GPU: how long the clocks take to settle, and what an
idle gap costs.  usage: warmup_curve.py [launches] [idle seconds]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pire_amd
from pire_amd import binding as pb, workloads as W

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 400
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
big = W.pattern_set("set_a"); table = pire_amd.Table(W.load_blob(big["blob"])); table.upload()
n, L = 1 << 20, 4096
text = torch.empty((n, L), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
pire_amd.corpus_fill_device(text.data_ptr(), 0x5EED5EED, 0, n, L, L, W.plants_for(big), stream)
idx = torch.empty(n, dtype=torch.int32, device="cuda"); fin = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
time.sleep(1.0)

def burst(k):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        table.run_strided_device(text.data_ptr(), n, L, L, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / k * 1e3
    return np.array([a.elapsed_time(b) for a, b in ev]), wall

ms, wall = burst(launches)
pts = [0, 1, 2, 3, 5, 8, 12, 16, 20, 30, 40, 60, 80, 120, 160, 240, 320, launches - 1]
print("cold start:", " ".join("%d:%.3f" % (i, ms[i]) for i in pts if i < launches), "| wall/launch %.4f" % wall)
print("means: first 20 %.4f, 20-40 %.4f, 40-80 %.4f, 80-160 %.4f, last 100 %.4f, min %.4f" % (ms[:20].mean(), ms[20:40].mean(), ms[40:80].mean(), ms[80:160].mean(), ms[-100:].mean(), ms.min()))
for gap in (0.02, 0.1, idle, 2.0):
    time.sleep(gap)
    ms2, wall2 = burst(40)
    print("after %.2f s idle: first 5 %s  mean first 20 %.4f last 20 %.4f" % (gap, " ".join("%.3f" % v for v in ms2[:5]), ms2[:20].mean(), ms2[20:].mean()))
table.adapt()
ms3, _ = burst(100)
print("after adapt(): mean %.4f min %.4f" % (ms3.mean(), ms3.min()))
