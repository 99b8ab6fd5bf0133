import sys, numpy as np
setname, corpus = sys.argv[1], sys.argv[2]
states = np.load(f"/tmp/pire_sim/states_{setname}_{corpus}.npy")
text = np.load(f"/tmp/pire_sim/text_{setname}_{corpus}.npy")
n, L = states.shape
# remap states to dense ids by frequency
u, inv = np.unique(states, return_inverse=True)
cnt = np.bincount(inv.ravel())
rank = np.empty(len(u), dtype=np.int64); rank[np.argsort(-cnt)] = np.arange(len(u))
sid = rank[inv.reshape(n, L)]
Lsub = 512  # positions sampled
pos = np.linspace(0, L - 1, Lsub).astype(int)
sid = sid[:, pos]; by = text[:, pos].astype(np.int64)

def cost(dw):  # dw: [n, Lsub] dword addresses; returns avg cycles per wave instr (2 halves)
    nw = n // 32
    d = dw.reshape(nw, 32, Lsub).transpose(0, 2, 1).reshape(-1, 32)   # [halfwaves*pos, 32]
    bank = d & 63
    # distinct dwords per bank: sort by (bank, dw), count unique
    key = bank * (1 << 40) + d
    key.sort(axis=1)
    newdw = np.ones_like(key, dtype=bool); newdw[:, 1:] = key[:, 1:] != key[:, :-1]
    b = key >> 40
    tot = np.zeros(len(key))
    mx = np.zeros(len(key), dtype=np.int64)
    for k in range(64):
        c = ((b == k) & newdw).sum(axis=1)
        mx = np.maximum(mx, c)
    return 2 * mx.mean()

lane = (np.arange(n) % 64)[:, None]
print("baseline            %.3f" % cost(sid * 64 + (by >> 2)))
for name, masks in [("2rep 0/80", [0, 0x80]), ("4rep 0/40/80/C0", [0, 0x40, 0x80, 0xC0]), ("2rep 0/40", [0, 0x40]),
                    ("4rep 0/20/80/A0", [0,0x20,0x80,0xA0]), ("8rep", [0,0x20,0x40,0x60,0x80,0xA0,0xC0,0xE0]),
                    ("4rep 0/80/10/90", [0,0x80,0x10,0x90]), ("4rep 0/80/60/E0",[0,0x80,0x60,0xE0])]:
    G = len(masks)
    m = np.array(masks)[lane % G]
    rep = lane % G
    dw = (sid + 1000 * rep) * 64 + ((by ^ m) >> 2)
    print("%-20s %.3f" % (name, cost(dw)))
# pitch variants (no replicas): 352-byte pitch
print("pitch 352           %.3f" % cost(sid * 88 + (by >> 2)))
print("pitch 384           %.3f" % cost(sid * 96 + (by >> 2)))
print("rot2 byte           %.3f" % cost(sid * 64 + ((((by << 2) & 0xFF) | (by >> 6)) >> 2)))
print("--- u16 entries: addr = rank*P + 2*b")
for P in (512, 520, 528, 544, 560, 576, 592, 608, 640, 672, 704, 736, 768):
    print("u16 pitch %4d       %.3f" % (P, cost((sid * P + 2 * by) >> 2)))
print("--- u8 entries, pitch P: addr = rank*P + b")
for P in (256, 260, 272, 288, 320, 352, 384, 416, 448):
    print("u8 pitch %4d        %.3f" % (P, cost((sid * P + by) >> 2)))
