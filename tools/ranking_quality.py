"""How far is the ranking adapt() learns from the best one?  For a wide set and its URL batch: the share of the steps (oracle's
visit counts on a held-out sample) that fall outside the first `tier` states of (a) the table's ranking after k rounds of
scan + adapt(), (b) the ranking by the oracle's own visit counts of another sample."""
import sys
import numpy as np
import torch
import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from pire_amd import workloads as W

name = sys.argv[1] if len(sys.argv) > 1 else "blacklist_1k"
dense = len(sys.argv) > 2 and sys.argv[2] == "dense"   # the 255 dense rows under the ragged kernel instead of the wide tier
entry = W.wide_set(name)
blob = W.load_blob(entry["blob"])
o = ob.OracleScanner(blob)
t1, o1 = W.wide_urls(entry, 0x5EED5EED, 1 << 16)
t2, o2 = W.wide_urls(entry, 77, 1 << 16)
def lookups(text, offs):
    """How often each state's ROW is looked up: the state in front of every byte -- the oracle's visit counts are of the states
    BEHIND the bytes, so the state every string starts in is added once per (non-empty) string and the state behind a string's
    last byte taken off once (nothing is looked up in it: End() has its own record)."""
    v = o.visit_counts(text, offs).astype(np.float64)
    ends, _ = o.run(text, offs, flags=ob.FLAG_BEGIN, threads=4)
    start, _ = o.run(text[:0], np.zeros(2, dtype=np.uint64), flags=ob.FLAG_BEGIN)
    nonempty = np.diff(offs) > 0
    np.subtract.at(v, ends[nonempty], 1.0)
    v[int(start[0])] += float(nonempty.sum())
    return v


v1 = lookups(t1, o1)
v2 = lookups(t2, o2)
rep = 64
lens = np.tile(np.diff(o1), rep)
offs = np.zeros(len(lens) + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
text = torch.as_tensor(np.ascontiguousarray(t1), device="cuda").repeat(rep).contiguous()
doffs = torch.as_tensor(offs.astype(np.int64), device="cuda")
n = len(lens)
idx = torch.empty(n, dtype=torch.int32, device="cuda"); fin = torch.empty(n, dtype=torch.uint8, device="cuda")
pb.set_config(walk_variant=1 if dense else 2, zip_variant=1, auto_adapt=1, ragged_variant=1)
t = pire_amd.Table(blob); t.upload()
tier = t.info.hot_states if dense else t.info.wide_states
ideal = np.argsort(-v1)
print(name, "tier", tier, "ideal ranking (other sample): outside %.4f" % (1 - v2[ideal[:tier]].sum() / v2.sum()))
for k in range(8):
    t.run_device(text.data_ptr(), doffs.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t.adapt()
    orig_of_perm, _ = t.layout()
    inside = v2[orig_of_perm[:tier]].sum() / v2.sum()
    i2 = t.refresh_info()
    print("after %d x (scan of %d URLs + adapt()): outside %.4f ; the library's own measure %.4f" % (k + 1, n, 1 - inside, i2.outside_dense_share if dense else i2.outside_wide_share))
    perm_of_orig = np.empty_like(orig_of_perm); perm_of_orig[orig_of_perm] = np.arange(len(orig_of_perm), dtype=orig_of_perm.dtype)
    out = [(v2[s0] / v2.sum(), int(s0), int(perm_of_orig[s0]), int(np.nonzero(ideal == s0)[0][0])) for s0 in np.argsort(-v2)[:3000] if perm_of_orig[s0] >= tier][:6]
    print("   heaviest states outside the tier (share of steps, state, its place in the library's ranking, in the ideal one):",
          [("%.4f" % a, b, c, d2) for a, b, c, d2 in out])
