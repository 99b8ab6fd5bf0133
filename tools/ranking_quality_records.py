"""tools/ranking_quality.py for fixed-length records (the class-indexed walk's kernels of wide.hip): the share of the row lookups
(oracle's counts on a held-out sample) outside the first `tier` states of the table's ranking after k rounds of scan + adapt(),
against the ranking by the oracle's own counts of another sample."""
import sys
import numpy as np
import torch
import pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from pire_amd import workloads as W

name, corpus = sys.argv[1], sys.argv[2]
zipv = int(sys.argv[3]) if len(sys.argv) > 3 else 1
entry = W.wide_set(name)
blob = W.load_blob(entry["blob"])
o = ob.OracleScanner(blob)
n, length = 1 << 18, 4096
nb = 8192


def lookups(data):
    text = data.reshape(-1)
    offs = np.arange(data.shape[0] + 1, dtype=np.uint64) * data.shape[1]
    v = o.visit_counts(text, offs).astype(np.float64)
    ends, _ = o.run(text, offs, flags=ob.FLAG_BEGIN, threads=8)
    start, _ = o.run(text[:0], np.zeros(2, dtype=np.uint64), flags=ob.FLAG_BEGIN)
    np.subtract.at(v, ends, 1.0)
    v[int(start[0])] += float(data.shape[0])
    return v


dense = len(sys.argv) > 4 and sys.argv[4] == "dense"


def url_records(seed, count, width):
    """URLs as records of a fixed layout: one per record, padded with line feeds -- states that live at ONE offset of every record."""
    text, offs = W.wide_urls(entry, seed, count)
    out = np.full((count, width), 10, dtype=np.uint8)
    for i in range(count):
        u = text[int(offs[i]):int(offs[i + 1])][:width]
        out[i, :len(u)] = u
    return out


if corpus.startswith("urls"):
    length = int(corpus[4:] or 256)
    d1, d2 = url_records(0x5EED5EED, nb, length), url_records(77, nb, length)
else:
    d1 = W.wide_records(entry, corpus, 0x5EED5EED, nb, length)
    d2 = W.wide_records(entry, corpus, 77, nb, length)
v1, v2 = lookups(d1), lookups(d2)
order = W.rotated_repeat_order(n, nb)
text = torch.as_tensor(d1, device="cuda").index_select(0, torch.as_tensor(order, device="cuda")).contiguous()
idx = torch.empty(n, dtype=torch.int32, device="cuda"); fin = torch.empty(n, dtype=torch.uint8, device="cuda")
pb.set_config(walk_variant=1 if dense else 0, zip_variant=zipv, auto_adapt=1)
t = pire_amd.Table(blob); t.upload()
ideal = np.argsort(-v1)
for k in range(8):
    t.run_strided_device(text.data_ptr(), n, length, length, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t.adapt()
    i2 = t.refresh_info()
    tier = i2.hot_states if dense else i2.wide_states
    orig_of_perm, _ = t.layout()
    inside = v2[orig_of_perm[:tier]].sum() / v2.sum()
    best = v2[ideal[:tier]].sum() / v2.sum()
    print("%s %s zip_variant %d after %d x (scan + adapt()): kernel %s, tier %d (rows %d): outside %.4f (best possible with %d states: %.4f) ; the library's own measure %.4f"
          % (name, corpus, zipv, k + 1, pb.last_kernel(), tier, i2.zip_full_states or tier, 1 - inside, tier, 1 - best, i2.outside_dense_share if dense else i2.outside_wide_share), flush=True)
