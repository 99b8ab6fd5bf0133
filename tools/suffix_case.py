#!/usr/bin/env python3
"""LongestSuffix / ShortestSuffix throughput (never measured): pire_hip_suffix with device pointers on the batch of
tools/prefix_case.py -- set_a table (its patterns are $-anchored: walked backwards from the end they are what a reversed scanner
sees first), 2^LOG2 strings of 64..1023 B -- against the oracle on a sample.  One string per lane, backwards (exact.hip SuffixKernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from tests import helpers as H

big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
blob = H.load_blob(big["blob"])
t = pire_amd.Table(blob)
t.upload()
LOG2 = int(os.environ.get("PREFIX_LOG2_STRINGS", "20"))
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
o = ob.OracleScanner(blob)
for longest in (True, False):
    for _ in range(30):
        t.suffix_device(d.data_ptr(), do.data_ptr(), m, longest, dout.data_ptr(), stream=stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record()
        t.suffix_device(d.data_ptr(), do.data_ptr(), m, longest, dout.data_ptr(), stream=stream)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    out = dout.cpu().numpy()
    k = 1 << 14
    want = o.suffix(text[:int(offs[k])], offs[:k + 1], longest)
    walked = int(np.where(out >= 0, np.minimum(out + 1, lens.astype(np.int64)), lens.astype(np.int64)).sum()) if not longest else total
    print("%s (%s): %d strings, %.3f GiB: kernel %.3f ms -> %.1f GB/s of text (%.1f GB/s of the bytes a search has to walk); "
          "parity with the oracle on the first %d strings: %s" % ("LongestSuffix" if longest else "ShortestSuffix", pb.last_kernel(), m, total / 2**30, ms,
                                                                total / ms / 1e6, walked / ms / 1e6, k, bool((want == out[:k]).all())))
