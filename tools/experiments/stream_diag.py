#!/usr/bin/env python3
"""Which strings does the stream kernel get wrong?  (debugging aid)"""
import numpy as np, torch
import pire_amd
from pire_amd import binding as pb
from oracle import binding as ob
from tests import helpers as H
big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
blob = H.load_blob(big["blob"])
t, o = pire_amd.Table(blob), ob.OracleScanner(blob); t.upload()
A = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ hello wd0123456789-() @net", dtype=np.uint8)
rng = np.random.RandomState(1)
strings = [A[rng.randint(0, len(A), size=int(rng.randint(20, 200)))].tobytes() for _ in range(20000)]
text, offs = H.pack(strings)
d = torch.as_tensor(np.array(text), device="cuda"); do = torch.as_tensor(offs.astype(np.int64), device="cuda")
n = len(strings)
idx = torch.full((n,), -1, dtype=torch.int32, device="cuda"); fin = torch.zeros(n, dtype=torch.uint8, device="cuda")
cnt = torch.zeros(t.RegexpsCount + 2, dtype=torch.int64, device='cuda')
t.run_device(d.data_ptr(), do.data_ptr(), n, 3 | pb.FLAG_SHORT, idx.data_ptr(), fin.data_ptr(), cnt.data_ptr(), 0, 0)
torch.cuda.synchronize()
oi, of = o.run(text, offs, flags=3, threads=4)
gi = idx.cpu().numpy().astype(np.uint32)
bad = np.nonzero(gi != oi)[0]
print("kernel", pb.last_kernel(), "bad", bad.size, "strings counted", int(cnt[1]), "of", n, "unwritten", int((gi == 0xFFFFFFFF).sum()))
ends = offs[1:].astype(np.int64); starts = offs[:-1].astype(np.int64)
line_of_end = ends // 128
cnt = np.bincount(line_of_end)
for i in bad[:25]:
    same = np.nonzero(line_of_end == line_of_end[i])[0]
    print(i, "len", len(strings[i]), "start%1024", starts[i] % 1024, "end%128", ends[i] % 128, "end%16", ends[i] % 16,
          "ends in its tile", cnt[line_of_end[i]], "rank among them", int(np.nonzero(same == i)[0][0]), "got", gi[i], "want", oi[i],
          "unwritten" if gi[i] == 0xFFFFFFFF else "")

sets = []
for rep in range(5):
    idx.fill_(-1)
    t.run_device(d.data_ptr(), do.data_ptr(), n, 3 | pb.FLAG_SHORT, idx.data_ptr(), fin.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize()
    g = idx.cpu().numpy().astype(np.uint32)
    sets.append(set(np.nonzero(g != oi)[0].tolist()))
print("bad per run", [len(x) for x in sets], "in all runs", len(set.intersection(*sets)), "in any", len(set.union(*sets)))
u = sorted(set.union(*sets))
lanes = [(int(starts[i]) // 1024) % 64 for i in u]
print("lanes of the failing strings:", np.bincount(lanes, minlength=64).tolist())
print("tasks:", sorted(set(int(starts[i]) // 65536 for i in u)))
print("wrong values:", np.unique(gi[bad], return_counts=True))
