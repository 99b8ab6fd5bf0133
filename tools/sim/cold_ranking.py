import sys, os, numpy as np, time
sys.path.insert(0, "/root/repo")
from oracle import binding as ob
from pire_amd import workloads as W
import pire_amd
setname = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
big = W.pattern_set(setname)
blob = W.load_blob(big["blob"])
fn = f"/tmp/pire_sim/nxt_{setname}.npy"
o = ob.OracleScanner(blob)
if not os.path.exists(fn):
    S = o.size
    cls = np.array([o.letter_class(c) for c in range(256)])
    rep = {}
    for c in range(256): rep.setdefault(cls[c], c)
    nxt = np.zeros((S, 256), dtype=np.int32)
    for s in range(S):
        row = {k: o.next(s, r) for k, r in rep.items()}
        nxt[s] = [row[cls[c]] for c in range(256)]
    np.save(fn, nxt)
nxt = np.load(fn)
begin = o.next(o.initial, 258)
n = 1024
plants = ob.make_plants([(bytes.fromhex(h), t) for h, t in zip(big["witnesses_hex"], big["witness_at_tail"])])
text = ob.corpus_fill(0x5EED5EED, 0, n, L, plants, threads=8)
st = np.full(n, begin, dtype=np.int32)
states = np.zeros((n, L + 1), dtype=np.int32)
for i in range(L):
    states[:, i] = st
    st = nxt[st, text[:, i]]
states[:, L] = st
t = pire_amd.Table(blob)
lay = t.layout() if hasattr(t, "layout") else None
print(type(lay), [a for a in dir(t) if not a.startswith("_")][:40])
orig_of_perm, hot_rows = lay
hot = t.info.hot_states
hotset = np.zeros(nxt.shape[0], dtype=bool); hotset[np.asarray(orig_of_perm[:hot])] = True
cold = ~hotset[states[:, 1:]]            # state entered by byte i is not hot
print("hot rows", hot, "distinct visited", len(np.unique(states)), "visited & not hot", len(set(np.unique(states)) - set(np.asarray(orig_of_perm[:hot]).tolist())))
print("per-lane cold-step share %.5f" % cold.mean())
# chunk-level: a wave (64 strings) re-walks a chunk if any lane enters a cold state in it or starts cold
c = cold.reshape(n // 64, 64, L // 16, 16).any(axis=3)      # [wave, lane, chunk]
print("per-lane trapped-chunk share %.5f   per-wave chunks with a trap %.5f" % (c.mean(), c.any(axis=1).mean()))
pos = np.where(cold.any(axis=0))[0]
print("cold positions: first", pos[:10], "last", pos[-10:], "count", len(pos))
u, cnt = np.unique(states[:, 1:][cold], return_counts=True)
print("top cold states", sorted(zip(cnt.tolist(), u.tolist()), reverse=True)[:12])
