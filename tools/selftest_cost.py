#!/usr/bin/env python3
"""What the first-use self-test costs (pire_hip_config.selftest): wall-clock of the FIRST pire_hip_run_strided of a fresh table
with it on and off, and of the second call (tables uploaded beforehand, so that the upload is in neither)."""
import time
import numpy as np
import torch
import pire_amd
from pire_amd import binding as pb
from pire_amd import workloads as W
from tests import helpers as H

big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
blobs = {"set_a (4552 states)": H.load_blob(big["blob"]), "dict_10k (30202 states, wide walk)": W.load_blob(W.wide_set("dict_10k")["blob"])}
n, length = 4096, 1024
text = torch.randint(32, 127, (n, length), dtype=torch.uint8, device="cuda")
idx = torch.empty(n, dtype=torch.int32, device="cuda")
fin = torch.empty(n, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for name, blob in blobs.items():
    for mode, label in ((1, "off"), (0, "on")):
        first, second = [], []
        for _ in range(5):
            pb.set_config(selftest=mode, walk_variant=2 if "wide" in name else 0)
            t = pire_amd.Table(blob)
            t.upload()
            torch.cuda.synchronize()
            for out in (first, second):
                t0 = time.perf_counter()
                t.run_strided_device(text.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, stream)
                torch.cuda.synchronize()
                out.append((time.perf_counter() - t0) * 1e3)
        print("%s, self-test %s: first call %.2f ms (min of 5 tables; median %.2f), second call %.3f ms" % (
            name, label, min(first), float(np.median(first)), min(second)))
