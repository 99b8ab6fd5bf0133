#!/usr/bin/env python3
"""Which strings does the row kernel count differently from the 16-bit-entry kernel?  (diagnostic)"""
import numpy as np
import pire_amd
from pire_amd import binding as pb
from tests import helpers as H

case = [c for c in H.golden()["counting"] if c["name"] == "count0_advanced"][0]
t = pire_amd.CountingTable(H.load_blob(case["blob"]), case["kind"])
rng = np.random.RandomState(5)
alphabet = b"abcdefghijklmnopqrstuvwxyz    .,:/http"
strings = [bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=int(n))) for n in rng.randint(0, 700, size=4096)]
with pb.config(counting_variant=1):
    ri, rr = t.run_strings(strings)
with pb.config(counting_variant=2):
    gi, gr = t.run_strings(strings)
    print("kernel", pb.last_kernel())
offs = np.concatenate([[0], np.cumsum([len(s) for s in strings])])
bad = np.nonzero((gr != rr).any(axis=1) | (gi != ri))[0]
print("mismatching strings: %d of %d" % (len(bad), len(strings)))
for i in bad[:40]:
    print("string %5d  off%%128 %3d  len %4d  end%%128 %3d  windows %d  ref %s idx %d  got %s idx %d" % (
        i, offs[i] % 128, len(strings[i]), offs[i + 1] % 128, (offs[i] % 128 + len(strings[i]) + 127) // 128,
        rr[i].tolist(), ri[i], gr[i].tolist(), gi[i]))
lens = np.array([len(s) for s in strings])
print("mismatch share by window count:", {int(w): "%d/%d" % (int(((offs[:-1] % 128 + lens + 127) // 128 == w)[bad].sum()), int(((offs[:-1] % 128 + lens + 127) // 128 == w).sum())) for w in range(0, 8)})
