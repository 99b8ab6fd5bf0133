import sys, numpy as np, time
sys.path.insert(0, "/root/repo")
from oracle import binding as ob
from pire_amd import workloads as W
import glob, os

setname = sys.argv[1] if len(sys.argv) > 1 else "set_a"
corpus = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
big = W.pattern_set(setname)
blob = W.load_blob(big["blob"])
o = ob.OracleScanner(blob)
S = o.size
# class of each byte, and next[state][class] via representative bytes
cls = np.array([o.letter_class(c) for c in range(256)])
ncls = cls.max() + 1
rep = {}
for c in range(256):
    rep.setdefault(cls[c], c)
print("states", S, "classes(bytes)", len(rep), file=sys.stderr)
nxt = np.zeros((S, 256), dtype=np.int32)
t0 = time.time()
for s in range(S):
    row = {k: o.next(s, r) for k, r in rep.items()}
    nxt[s] = [row[cls[c]] for c in range(256)]
print("table built", time.time() - t0, file=sys.stderr)
begin = o.next(o.initial, 256)  # BeginMark = 256? check
np.save(f"/tmp/pire_sim/nxt_{setname}.npy", nxt)
n, L = 4096, 4096
if corpus == "synthetic":
    plants = W.plants_for(big)
    # make_plants from pire_amd.binding; ob.corpus_fill wants ob struct: rebuild
    plants = ob.make_plants([(bytes.fromhex(h), t) for h, t in zip(big["witnesses_hex"], big["witness_at_tail"])])
    text = ob.corpus_fill(0x5EED5EED, 0, n, L, plants, threads=8)
else:
    files = sorted(glob.glob("/root/repo/pire_amd/csrc/*") + glob.glob("/root/repo/include/*.h") + glob.glob("/root/repo/include/pire_hip/*") + glob.glob("/root/repo/oracle/*.c") + glob.glob("/root/repo/tests/cpp/*.cpp"))
    data = b"".join(open(f, "rb").read() for f in files if os.path.isfile(f))
    base = np.frombuffer(data, dtype=np.uint8)
    text = np.resize(base, n * L).reshape(n, L)
st = np.full(n, begin, dtype=np.int32)
states = np.zeros((n, L), dtype=np.int32)
for i in range(L):
    states[:, i] = st
    st = nxt[st, text[:, i]]
np.save(f"/tmp/pire_sim/states_{setname}_{corpus}.npy", states)
np.save(f"/tmp/pire_sim/text_{setname}_{corpus}.npy", text)
u, c = np.unique(states, return_counts=True)
order = np.argsort(-c)
print("distinct states", len(u))
cum = np.cumsum(c[order]) / c.sum()
for k in (1, 2, 4, 8, 16, 19, 32, 48, 63, 85, 127):
    if k <= len(u):
        print("top", k, "coverage %.5f" % cum[k - 1])
