"""One process, several GPUs (pire_hip_multi_*, SURVEY 8e / BASELINE C4) and the per-device table images behind it.

CPU: the entry points exist and refuse loudly without a device.  GPU (one-GPU box): the runner is built over device 0
listed two or three times -- each "device" slot gets its own stream, counter buffer and shard, the counters are summed
on the host because RCCL refuses duplicate devices -- so sharding, reduction and result placement are exercised
through the real kernels; with two or more GPUs the same tests also run over distinct devices with RCCL."""
import subprocess
import sys
import os

import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multi_create_without_gpu_fails_loudly():
    import pire_amd

    if pire_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(pire_amd.PireHipError):
        pire_amd.MultiRunner(ndev=2)


def _expected_counts(o, idx, fin):
    cnt = np.zeros(o.regexps + 2, dtype=np.uint64)
    cnt[0], cnt[1] = int(fin.sum()), len(idx)
    for i in idx.tolist():
        for r in o.accepted(i):
            cnt[2 + r] += 1
    return cnt


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 2, 3])
@pytest.mark.parametrize("n,length", [(1000, 512), (5, 256), (64 * 40 + 7, 1024), (0, 256)])
def test_host_batch_sharded_over_device_slots(slots, n, length):
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    m = pire_amd.MultiRunner(devices=[0] * slots)
    assert m.device_count == slots
    assert m.reduce_backend.startswith("host")          # one device listed several times: no RCCL communicator
    data = ob.corpus_fill(77 + n, 0, n, length, H.plants_for(big), threads=4) if n else np.zeros((0, length), np.uint8)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    gi, gf, cnt = m.run_strided_host(t, data)
    assert (gi == oi).all() and (gf == of).all()
    assert (cnt == _expected_counts(o, oi, of)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 2, 4])
@pytest.mark.parametrize("shape", ["urls", "skewed", "tiny", "empties", "resume"])
def test_ragged_host_batch_sharded_by_bytes_over_device_slots(slots, shape):
    """pire_hip_multi_run_host: what the reference's callers have is ragged lines (pigrep.cpp:38-45), so the batch is
    cut by BYTES into one run of whole strings per device, staged through the runner's pooled per-device buffers
    (the second call of each runner allocates nothing), scanned concurrently, results in string order, counters summed.
    "skewed": a few long strings among many short ones -- equal string counts would give one device most of the text."""
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(len(shape) * 7 + slots)
    alphabet = b"abcdeaxHedInrTailhello w0123456789()- ABCDEFGHIJKLMNOPQRSTUVWXYZ@Qnet"
    if shape == "urls":
        strings = H.random_strings(rng, 5000, 200, alphabet)
    elif shape == "skewed":
        strings = H.random_strings(rng, 3000, 60, alphabet)
        for k in (5, 1500, 2990):
            strings[k] = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=60000))
    elif shape == "tiny":
        strings = H.random_strings(rng, 3, 40, alphabet)          # fewer strings than device slots (for 4)
    elif shape == "empties":
        strings = [b""] * 700
    else:
        strings = H.random_strings(rng, 2000, 300, alphabet)
    text, offs = H.pack(strings)
    # offsets that do not start at 0, text that does not start at a 256-byte boundary
    lead = 77
    text = np.concatenate([np.full(lead, ord("x"), np.uint8), text])
    offs = offs + np.uint64(lead)
    init = None
    if shape == "resume":
        init = rng.randint(0, o.size, size=len(strings)).astype(np.uint32)
        oi, of = o.run(text, offs, threads=4, init_idx=init)
    else:
        oi, of = o.run(text, offs, threads=4)
    m = pire_amd.MultiRunner(devices=[0] * slots)
    for rep in range(2):
        gi, gf, cnt = m.run_host(t, text, offs, init_idx=init)
        assert (gi == oi).all() and (gf == of).all(), (shape, slots, rep)
        assert (cnt == _expected_counts(o, oi, of)).all()
    split = m.last_split()
    assert split[0] == 0 and split[-1] == len(strings) and split == sorted(split) and len(split) == slots + 1
    total = int(offs[-1] - offs[0])
    if shape in ("urls", "skewed", "resume") and slots > 1:
        longest = max(len(x) for x in strings)
        for g in range(slots):   # every shard within one string of its fair share of the BYTES
            got = int(offs[split[g + 1]] - offs[split[g]])
            assert abs(got - total / slots) <= longest, (g, got, total / slots)


@pytest.mark.gpu
def test_device_resident_offset_shards():
    """pire_hip_multi_run: offset shards already resident on their devices (here: three slots of device 0)."""
    import torch

    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    rng = np.random.RandomState(3)
    m = pire_amd.MultiRunner(devices=[0, 0, 0])
    shards, keep, want = [], [], []
    for g, n in enumerate((900, 0, 2500)):
        strings = H.random_strings(rng, n, 250, b"abcdehello w0123456789()- ABCXYZ@Qnet")
        text, offs = H.pack(strings)
        d = torch.as_tensor(np.array(text) if len(text) else np.zeros(16, np.uint8), device="cuda")
        do = torch.as_tensor(offs.astype(np.int64), device="cuda")
        idx = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
        fin = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
        keep += [d, do, idx, fin]
        shards.append((d.data_ptr(), do.data_ptr(), n, 0, idx.data_ptr(), fin.data_ptr()))
        want.append((n, idx, fin) + tuple(o.run(text, offs, threads=2)))
    cnt = m.run_offset_shards(t, shards)
    total = np.zeros(o.regexps + 2, dtype=np.uint64)
    for n, idx, fin, oi, of in want:
        assert (idx[:n].cpu().numpy().astype(np.uint32) == oi).all() and (fin[:n].cpu().numpy() == of).all()
        total += _expected_counts(o, oi, of)
    assert (cnt == total).all()


@pytest.mark.gpu
def test_device_resident_shards_and_rccl_when_there_are_two_gpus():
    import torch

    import pire_amd
    from pire_amd import binding as pb

    ndev = torch.cuda.device_count()
    devices = [0, 1] if ndev >= 2 else [0, 0]
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    m = pire_amd.MultiRunner(devices=devices)
    if ndev >= 2:
        assert m.reduce_backend.startswith("rccl"), m.reduce_backend
    per, length, plants = 4096 + 64, 1024, H.plants_for(big)
    keep, shards = [], []
    for g, d in enumerate(devices):
        with torch.cuda.device(d):
            buf = torch.empty((per, length), dtype=torch.uint8, device=f"cuda:{d}")
            pire_amd.corpus_fill_device(buf.data_ptr(), 5, g * per, per, length, length, plants,
                                        torch.cuda.current_stream().cuda_stream)
            idx = torch.empty(per, dtype=torch.int32, device=f"cuda:{d}")
            fin = torch.empty(per, dtype=torch.uint8, device=f"cuda:{d}")
            torch.cuda.synchronize()
        keep.append((buf, idx, fin))
        shards.append((buf.data_ptr(), per, length, length, 0, idx.data_ptr(), fin.data_ptr()))
    for _ in range(2):      # twice: the counters must be totals of ONE call, not accumulated
        cnt = m.run_shards(t, shards)
    host = ob.corpus_fill(5, 0, per * len(devices), length, plants, threads=4)
    oi, of = o.run(host.reshape(-1), np.arange(per * len(devices) + 1, dtype=np.uint64) * length, threads=4)
    gi = np.concatenate([k[1].cpu().numpy().astype(np.uint32) for k in keep])
    gf = np.concatenate([k[2].cpu().numpy() for k in keep])
    assert (gi == oi).all() and (gf == of).all()
    assert (cnt == _expected_counts(o, oi, of)).all()
    assert pb.last_kernel() in ("tiled", "generic")


@pytest.mark.gpu
def test_one_table_handle_on_every_device_from_two_host_threads():
    """The table keeps one image per device: two host threads driving two devices (or the same one) with ONE handle
    must not free each other's image."""
    import threading

    import torch

    import pire_amd
    from pire_amd import binding as pb

    ndev = torch.cuda.device_count()
    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    n, length = 2048, 512
    host = ob.corpus_fill(9, 0, n, length, H.plants_for(big), threads=4)
    oi, of = o.run(host.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    errors = []

    def worker(k):
        d = k % max(ndev, 1)
        try:
            with torch.cuda.device(d):
                buf = torch.as_tensor(host, device=f"cuda:{d}")
                idx = torch.empty(n, dtype=torch.int32, device=f"cuda:{d}")
                fin = torch.empty(n, dtype=torch.uint8, device=f"cuda:{d}")
                s = torch.cuda.Stream(device=d)
                for _ in range(20):
                    t.run_strided_device(buf.data_ptr(), n, length, length, pb.FLAG_BEGIN | pb.FLAG_END, idx.data_ptr(),
                                         fin.data_ptr(), 0, 0, s.cuda_stream)
                    s.synchronize()
                    if not ((idx.cpu().numpy().astype(np.uint32) == oi).all() and (fin.cpu().numpy() == of).all()):
                        errors.append(f"mismatch in thread {k} on device {d}")
                        return
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.gpu
def test_bench_two_ranks_through_the_real_kernel():
    """`python bench.py --gpus 2` launches its own ranks; with gloo both ranks share the GPU(s) of this box, scan their
    shards with the real kernel and reduce the counters: strings == 2 shards, results identical to one rank's x2
    where the shards are the same size (the corpus differs per shard, so only the string count is exactly 2x)."""
    import json

    import torch

    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--log2-strings", "12", "--steps", "3", "--warmup", "2",
            "--no-cpu", "--backend", backend]
    one = subprocess.run(base + ["--gpus", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run(base + ["--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    r2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert r1["n_gpus"] == 1 and r2["n_gpus"] == 2
    assert r2["match_counts"]["strings"] == 2 * r1["match_counts"]["strings"] == 2 << 12
    # rank 0's shard is the single rank's batch: the two-rank totals contain it
    assert all(b >= a for a, b in zip(r1["match_counts"]["per_regexp"], r2["match_counts"]["per_regexp"]))
    # and the whole thing equals the oracle on the 2 x 2^12 strings of the global corpus
    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    o = ob.OracleScanner(H.load_blob(big["blob"]))
    host = ob.corpus_fill(0x5EED5EED, 0, 2 << 12, 4096, H.plants_for(big), threads=4)
    oi, of = o.run(host.reshape(-1), np.arange((2 << 12) + 1, dtype=np.uint64) * 4096, threads=4)
    want = _expected_counts(o, oi, of)
    assert r2["match_counts"]["final"] == int(want[0])
    assert r2["match_counts"]["per_regexp"] == [int(x) for x in want[2:]]


@pytest.mark.gpu
def test_rccl_runs_as_a_one_rank_communicator_on_one_gpu(cfg):
    """ncclCommInitAll / the grouped ncclAllReduce(uint64, sum) had never executed -- every box this code has
    seen has one GPU, and pire_hip_multi_create only built a communicator for two or more distinct devices.  With
    pire_hip_config.force_rccl the one-device runner builds a ONE-rank communicator through the same dlsym'ed entry points
    and enum values (multi.cpp: copied from rccl.h) and sends the counters through the all-reduce: a sum over one rank is
    the identity, so the reduced counters must equal the single-device ones -- and a wrong datatype / op constant or a
    signature that does not match librccl shows here instead of in the driver's 8-GPU run."""
    import pire_amd

    big = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    blob = H.load_blob(big["blob"])
    t, o = pire_amd.Table(blob), ob.OracleScanner(blob)
    cfg.set(force_rccl=1)
    m = pire_amd.MultiRunner(devices=[0])
    assert m.reduce_backend.startswith("rccl"), m.reduce_backend
    n, length = 64 * 30 + 5, 768
    data = ob.corpus_fill(4242, 0, n, length, H.plants_for(big), threads=4)
    oi, of = o.run(data.reshape(-1), np.arange(n + 1, dtype=np.uint64) * length, threads=4)
    for rep in range(3):   # the communicator is reused
        gi, gf, cnt = m.run_strided_host(t, data)
        assert (gi == oi).all() and (gf == of).all()
        assert (cnt == _expected_counts(o, oi, of)).all(), rep
    assert m.reduce_backend.startswith("rccl"), m.reduce_backend   # no fallback to the host sum happened on the way
    # ragged host batch through the same runner
    rng = np.random.RandomState(3)
    ln = rng.randint(0, 300, size=5000).astype(np.uint64)
    offs = np.zeros(len(ln) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(ln)
    text = data.reshape(-1)[: int(offs[-1])]
    oi, of = o.run(text, offs, threads=4)
    gi, gf, cnt = m.run_host(t, text, offs)
    assert (gi == oi).all() and (gf == of).all() and (cnt == _expected_counts(o, oi, of)).all()
    assert m.reduce_backend.startswith("rccl"), m.reduce_backend
