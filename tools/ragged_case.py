#!/usr/bin/env python3
"""One ragged batch through pire_hip_run (device pointers), for rocprofv3: tools/ragged_case.py <case> [reps]."""
import sys
import numpy as np
import torch
import pire_amd
from pire_amd import binding as pb
from tests import helpers as H

CASES = {  # name: (lo, hi, log2 strings, multiplier)
    "uniform8k": (0, 8192, 17, 1), "uniform8k_al128": (0, 64, 17, 128), "urls": (20, 200, 22, 1),
    "loglines": (64, 1024, 20, 1), "fixed4096": (32, 33, 18, 128), "uniform2k": (0, 2048, 20, 1),
    # the same shapes at four times the size (3.4 GiB of text): how much of the small batches' time is their tail
    # smaller batches of the same shapes: where the stream kernel's fixed part (search, table, first line) stops paying
    "urls_16k": (20, 200, 14, 1), "urls_64k": (20, 200, 16, 1), "urls_256k": (20, 200, 18, 1), "urls_1m": (20, 200, 20, 1),
    "loglines_16k": (64, 1024, 14, 1), "loglines_64k": (64, 1024, 16, 1), "loglines_256k": (64, 1024, 18, 1),
    "urls_x4": (20, 200, 24, 1), "loglines_x4": (64, 1024, 22, 1), "uniform2k_x4": (0, 2048, 22, 1),
}
case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
no_out = bool(__import__("os").environ.get("NO_OUT"))
flags = 3 | (pb.FLAG_GENERIC if len(sys.argv) > 3 and sys.argv[3] == "generic" else 0)
lo, hi, lg, mul = CASES[case]
big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
t = pire_amd.Table(H.load_blob(big["blob"]))
t.upload()
stream = torch.cuda.current_stream().cuda_stream
n, L = (1 << 20, 4096) if case.endswith("_x4") else (1 << 18, 4096)
buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
pire_amd.corpus_fill_device(buf.data_ptr(), 0x5EED5EED, 0, n, L, L, H.plants_for(big), stream)
m = 1 << lg
lens = (np.random.RandomState(1).randint(lo, hi, size=m) * mul).astype(np.uint64)
offs = np.zeros(m + 1, dtype=np.uint64)
offs[1:] = np.cumsum(lens)
total = int(offs[-1])
assert total <= n * L
doffs = torch.as_tensor(offs.astype(np.int64), device="cuda")
idx = torch.empty(m, dtype=torch.int32, device="cuda")
fin = torch.empty(m, dtype=torch.uint8, device="cuda")
def launch():
    t.run_device(buf.data_ptr(), doffs.data_ptr(), m, flags, 0 if no_out else idx.data_ptr(), 0 if no_out else fin.data_ptr(), 0, 0, stream)


# two adaptation rounds from the batch itself (the estimates are remembered and settle), then the GPU's
# clocks settled by ~40 ms of back-to-back launches, then `reps` x 10 timed launches back to back
for r in range(2):
    launch()
    torch.cuda.synchronize()
    print("adapt: rows changed", t.adapt(), "hot", t.info.hot_states)
settle = max(20, int(40.0 / max(total / 2.5e9, 0.05)))
for _ in range(settle):
    launch()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps * 10)]
for a, b in ev:
    a.record()
    launch()
    b.record()
torch.cuda.synchronize()
try:   # tuning build (PIRE_HIP_LIB=tools/ab/libpire_hip_tuning.so PIRE_HIP_DEBUG_STREAM_CLOCKS=1): the stream kernel's stage clocks
    import ctypes as C
    out = (C.c_double * 8)()
    if pb.lib().pire_hip_debug_stream_clocks(out) == 0:
        print("stream clocks over the settled launches, us per wave: search %.2f | table copy %.2f | positions + lane search %.2f | "
              "window loop %.2f | flush %.2f | whole wave %.2f ; %.1f phases per wave, %d wave-launches" % (*out[:7], int(out[7])))
except AttributeError:
    pass
ts = [a.elapsed_time(b) for a, b in ev]
ch = t.adapt()
print("after the timed runs: adapt changed", ch, "rows; trap samples seen", t.refresh_info().last_trap_samples)
print("%s %s: %d strings, %.3f GiB, %d launches at settled clocks: mean %.3f ms -> %.1f GB/s (min %.3f ms -> %.1f GB/s)" % (
    pb.last_kernel(), case, m, total / 2**30, len(ts), float(np.mean(ts)), total / float(np.mean(ts)) / 1e6, min(ts),
    total / min(ts) / 1e6))
