#!/usr/bin/env python3
"""A/B of tiled-kernel variants (pire_hip_config.tiled_variant) inside ONE process, alternating bursts, so
that box, clocks and temperature are shared.  usage: ab_variants.py "0,22,21" [rounds] [set] [extra env k=v ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pire_amd
from pire_amd import binding as pb, workloads as W

variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "21"]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
setname = sys.argv[3] if len(sys.argv) > 3 else "set_a"
log2n = int(os.environ.get("AB_LOG2N", "20"))
big = W.pattern_set(setname); table = pire_amd.Table(W.load_blob(big["blob"])); table.upload()
n, L = 1 << log2n, 4096
text = torch.empty((n, L), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
pire_amd.corpus_fill_device(text.data_ptr(), 0x5EED5EED, 0, n, L, L, W.plants_for(big), stream)
idx = torch.empty(n, dtype=torch.int32, device="cuda"); fin = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

def burst(k, timed=True):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)] if timed else None
    for i in range(k):
        if timed: ev[i][0].record()
        table.run_strided_device(text.data_ptr(), n, L, L, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
        if timed: ev[i][1].record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in ev]) if timed else None

pb.set_config(tiled_variant=int(variants[0]))
burst(60, False); table.adapt(); burst(60, False)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in (variants if r % 2 == 0 else variants[::-1]):
        pb.set_config(tiled_variant=int(v))
        burst(10, False)
        ms = burst(60)
        res[v].append((ms.mean(), np.median(ms), ms.min()))
for v in variants:
    a = np.array(res[v])
    print("variant %-3s steady: mean %.4f (per round %s) median %.4f min %.4f" % (v, a[:, 0].mean(), " ".join("%.4f" % x for x in a[:, 0]), a[:, 1].mean(), a[:, 2].min()))
# the driver's shape: idle, 5 warm-up launches, sync, 20 timed launches
for v in variants:
    pb.set_config(tiled_variant=int(v))
    outs = []
    for rep in range(3):
        time.sleep(0.3)
        burst(5, False)
        outs.append(burst(20).mean())
    print("variant %-3s idle -> 5 warm-up -> 20 timed: %s" % (v, " ".join("%.4f" % x for x in outs)))
