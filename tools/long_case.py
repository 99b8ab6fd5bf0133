#!/usr/bin/env python3
"""Few long strings (segmented.hip): fixed-length records on the device, set_a table; the segmented scan against the
one-string-per-lane kernels (PIRE_HIP_NO_SEGMENTS=1) where those finish in reasonable time."""
import os
import sys

import numpy as np
import torch

import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

name = sys.argv[1] if len(sys.argv) > 1 else "set_a"
big = [b for b in H.big_sets() if b["name"] == name][0]
blob = H.load_blob(big["blob"])
t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
t.upload()
plants = H.plants_for(big)
stream = torch.cuda.current_stream().cuda_stream
total = int(os.environ.get('LONG_TOTAL_LOG2', '28')) and (1 << int(os.environ.get('LONG_TOTAL_LOG2', '28')))
buf = torch.empty(total, dtype=torch.uint8, device="cuda")
pire_amd.corpus_fill_device(buf.data_ptr(), 0x5EED5EED, 0, total // 4096, 4096, 4096, plants, stream)
torch.cuda.synchronize()


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


# adapt the dense rows on the same text first (as bench.py does)
idx = torch.empty(total // 4096, dtype=torch.int32, device="cuda")
fin = torch.empty(total // 4096, dtype=torch.uint8, device="cuda")
for _ in range(2):
    t.run_strided_device(buf.data_ptr(), total // 4096, 4096, 4096, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
    torch.cuda.synchronize()
    t.adapt()

for n in [int(x) for x in os.environ.get('LONG_NS', '1,8,64,1024,16384').split(',')]:
    length = total // n
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    run = lambda: t.run_strided_device(buf.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
    pb.set_config(segment_stats=1)
    run()
    torch.cuda.synchronize()
    pb.set_config(segment_stats=0)
    ms0 = timeit(run)
    # long strings live in other states than 4 KiB ones (sticky modes): let the dense rows follow
    rows = t.adapt()
    run()
    torch.cuda.synchronize()
    rows += t.adapt()
    ms = timeit(run)
    kernel = pb.last_kernel() + {"pirehip::ScanPairTiledKernel": "(2 modes fused)", "pirehip::ScanTiledSegKernel": "(1 mode)",
                                 "pirehip::ScanTiledSegKernel+derived": "(modes derived)",
                                 "pirehip::ScanTiledSegKernel+product": "(2 modes, product walk)"}.get(pb.last_kernel_symbol(), "")
    pb.set_config(segment_no_derive=1, segment_no_product=1)
    ms_noderive = timeit(run)
    pb.set_config(segment_no_derive=0, segment_no_product=0)
    pb.set_config(segment_no_pair=1, segment_no_derive=1)
    ms_nopair = timeit(run)
    pb.set_config(segment_no_pair=0, segment_no_derive=0)
    pb.set_config(segment_stats=1)
    run()
    torch.cuda.synchronize()
    pb.set_config(segment_stats=0)
    # parity: the oracle on the host over the same bytes (first strings only when there are many)
    k = min(n, 4)
    host = buf[:k * length].cpu().numpy()
    oi, of = o.run(host, np.arange(k + 1, dtype=np.uint64) * length, threads=min(k, 4))
    ok = bool((idx[:k].cpu().numpy().astype(np.uint32) == oi).all() and (fin[:k].cpu().numpy() == of).all())
    line = "%6d x %10d B: %-9s %8.3f ms -> %7.1f GB/s (%.3f ms with segment_no_derive / _no_product, %.3f with segment_no_pair too, %.3f before adapt(), %d rows changed); parity(first %d) %s" % (
        n, length, kernel, ms, total / ms / 1e6, ms_noderive, ms_nopair, ms0, rows, k, ok)
    if length <= (1 << 20):
        pb.set_config(no_segments=1)
        ms2 = timeit(run, reps=2)
        line += "   | one string per lane (%s): %9.3f ms -> %7.1f GB/s" % (pb.last_kernel(), ms2, total / ms2 / 1e6)
        pb.set_config(no_segments=0)
    print(line, flush=True)
