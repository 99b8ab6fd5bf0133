"""Does adapt() see a state that is visited at ONE fixed position of every string?  A blacklist scanner, a batch of URLs without a
scheme ("shop.<host>/..."): the states behind "sh", "shop" are looked up once per URL.  Rank them out of the tier first (a few
rounds on ordinary URLs), then scan the special batch and adapt(): where do they stand?"""
import numpy as np, torch, pire_amd
from oracle import binding as ob
from pire_amd import binding as pb
from pire_amd import workloads as W

entry = W.wide_set("blacklist_1k"); blob = W.load_blob(entry["blob"]); o = ob.OracleScanner(blob)
BEGIN = 258
def state_after(prefix):
    st = o.next(o.initial, BEGIN)
    for b in prefix: st = o.next(st, b)
    return st
targets = {p: state_after(p) for p in (b"", b"s", b"sh", b"sho", b"shop", b"shop.")}
print("states", targets)
t0, o0 = W.wide_urls(entry, 5, 1 << 16)
urls = [bytes(t0[int(o0[i]):int(o0[i + 1])]) for i in range(len(o0) - 1)]
special = [b"shop." + u.split(b"//", 1)[-1] for u in urls if b"//" in u][:40000]
def pack(strs, rep):
    text = np.frombuffer(b"".join(strs), dtype=np.uint8); lens = np.array([len(x) for x in strs], dtype=np.uint64)
    lens = np.tile(lens, rep); offs = np.zeros(len(lens) + 1, dtype=np.uint64); offs[1:] = np.cumsum(lens)
    return torch.as_tensor(text.copy(), device="cuda").repeat(rep).contiguous(), torch.as_tensor(offs.astype(np.int64), device="cuda"), len(lens)
pb.set_config(walk_variant=2, zip_variant=1, auto_adapt=1)
t = pire_amd.Table(blob); t.upload()
tier = t.info.wide_states
def places():
    orig_of_perm, _ = t.layout()
    perm = np.empty_like(orig_of_perm); perm[orig_of_perm] = np.arange(len(orig_of_perm), dtype=orig_of_perm.dtype)
    return {p.decode() or "<start>": int(perm[s]) for p, s in targets.items()}
for name, strs, rounds in (("ordinary URLs (with a scheme)", [u for u in urls if b"//" in u], 4), ("URLs that start with shop.", special, 4)):
    text, doffs, n = pack(strs, 32)
    idx = torch.empty(n, dtype=torch.int32, device="cuda"); fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    for r in range(rounds):
        t.run_device(text.data_ptr(), doffs.data_ptr(), n, 3, idx.data_ptr(), fin.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); ch = t.adapt()
        print(name, "round", r + 1, "tier", tier, "rows changed", ch, "kernel", pb.last_kernel(), "places", places(), flush=True)
