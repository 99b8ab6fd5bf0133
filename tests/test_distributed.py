"""The N>1 path on CPU: world_size-2 gloo.  Sharding by string index + the match-count all-reduce must reproduce
the single-process totals; the per-rank scanner in this TEST is the oracle (the GPU kernel needs a GPU)."""
import os
import socket

import numpy as np
import pytest

from pire_amd.distributed import shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1000, 1 << 20, (1 << 26) + 3):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, length, q):
    import torch
    import torch.distributed as dist

    from oracle import binding as ob
    from pire_amd import distributed as pd
    from tests import helpers as H

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    pd.init("gloo")
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    lo, hi = pd.shard_range(n_total, rank, world)
    data = ob.corpus_fill(0x5EED5EED, lo, hi - lo, length, H.plants_for(big))       # global string ids lo..hi
    idx, fin = o.run(data.reshape(-1), np.arange(hi - lo + 1, dtype=np.uint64) * length)
    counts = np.zeros(o.regexps + 2, dtype=np.int64)
    counts[0], counts[1] = int(fin.sum()), hi - lo
    for i in idx.tolist():
        for r in o.accepted(i):
            counts[2 + r] += 1
    t = torch.from_numpy(counts)
    t2 = t.clone()
    pd.allreduce_counts(t)
    work = pd.allreduce_counts(t2, async_op=True)      # what bench.py uses to overlap the reduce with the next scan
    assert work is not None
    work.wait()
    assert (t2 == t).all()
    slowest = pd.max_over_ranks(1.0 + rank)
    who = pd.describe_ranks()                          # bench.py's `per_rank_device`: one entry per rank, in rank order
    assert [w["rank"] for w in who] == list(range(world)) and pd.ranks_sharing_a_device(
        [dict(w, device=w["rank"]) for w in who]) == []
    pd.barrier()
    if rank == 0:
        q.put((t.numpy().tolist(), slowest))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_count_reduce():
    import torch.multiprocessing as mp

    from oracle import binding as ob
    from tests import helpers as H

    n_total, length, world = 301, 256, 2          # odd total: ranks get 151 / 150 strings
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, length, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, slowest = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    data = ob.corpus_fill(0x5EED5EED, 0, n_total, length, H.plants_for(big))
    idx, fin = o.run(data.reshape(-1), np.arange(n_total + 1, dtype=np.uint64) * length)
    want = np.zeros(o.regexps + 2, dtype=np.int64)
    want[0], want[1] = int(fin.sum()), n_total
    for i in idx.tolist():
        for r in o.accepted(i):
            want[2 + r] += 1
    assert got == want.tolist()
    assert slowest == 2.0      # MAX over ranks of (1.0, 2.0)


def test_ranks_on_one_device_are_noticed():
    """bench.py refuses an RCCL line whose ranks share a GPU: the check behind it."""
    from pire_amd import distributed as pd

    ranks = [{"rank": r, "host": "h", "device": r % 2, "pci_bus_id": "0000:%02x:00" % (r % 2), "name": "x"} for r in range(4)]
    assert pd.ranks_sharing_a_device(ranks) == [(0, 2), (1, 3)]
    assert pd.ranks_sharing_a_device(ranks[:2]) == []
    other_host = [dict(r, host="h%d" % r["rank"]) for r in ranks]
    assert pd.ranks_sharing_a_device(other_host) == []
