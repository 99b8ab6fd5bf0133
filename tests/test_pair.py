"""Pire::Run(scanner1, scanner2, ...) / ScannerPair (run.h:229-241, scanners/pair.h:33-94) in one fused pass
(pair.hip) and, where the batch is not made of fixed-length records, as two passes behind the same call: every result
must be the pair of what the oracle returns for the two scanners alone, Final = either."""
import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H

BE = ob.FLAG_BEGIN | ob.FLAG_END


def _table(name):
    c = [x for x in H.all_cases() + H.big_sets() if x["name"] == name][0]
    return H.load_blob(c["blob"]), c


def test_pair_needs_device_pointers_and_fails_loudly_without_a_gpu():
    import pire_amd
    from pire_amd import binding as pb

    a = pire_amd.Table(_table("string")[0])
    with pytest.raises(pire_amd.PireHipError):       # host pointers: refused whatever the box
        pb._check(pb.lib().pire_hip_run_pair_strided(a._h, a._h, None, 64, 256, 256, BE, None, None, None, None))


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [("set_a", "set_d"), ("set_d", "set_a"), ("survey_known_answer", "string"),
                                   ("set_b", "c2_single"), ("inline_glue3", "set_d")])
def test_fused_pair_equals_two_oracle_runs(n1, n2):
    import torch
    import pire_amd
    from pire_amd import binding as pb

    b1, c1 = _table(n1)
    b2, _ = _table(n2)
    t1, t2 = pire_amd.Table(b1), pire_amd.Table(b2)
    o1, o2 = ob.OracleScanner(b1), ob.OracleScanner(b2)
    big = [b for b in H.big_sets() if b["name"] == "set_d"][0]
    stream = torch.cuda.current_stream().cuda_stream
    for n, length, stride, flags in ((4096 + 37, 1024, 1024, BE), (2048, 512, 640, BE), (1000, 256, 256, ob.FLAG_BEGIN),
                                     (640, 4096, 4096, BE), (64, 256, 256, 0)):
        data = np.zeros((n, stride), dtype=np.uint8)
        data[:, :length] = ob.corpus_fill(11, 0, n, length, H.plants_for(big), threads=4)
        rng = np.random.RandomState(n)
        data[::5, :length] = rng.choice(np.frombuffer(b"abcdehlo wHeadInrTailfoobar.:/xq", dtype=np.uint8), size=(len(data[::5]), length))
        offs = np.arange(n + 1, dtype=np.uint64) * stride
        ends_text = np.ascontiguousarray(data[:, :length]).reshape(-1)
        eo = np.arange(n + 1, dtype=np.uint64) * length
        w1 = o1.run(ends_text, eo, flags=flags, threads=4)
        w2 = o2.run(ends_text, eo, flags=flags, threads=4)
        d = torch.as_tensor(data, device="cuda")
        i1 = torch.empty(n, dtype=torch.int32, device="cuda")
        i2 = torch.empty(n, dtype=torch.int32, device="cuda")
        fin = torch.empty(n, dtype=torch.uint8, device="cuda")
        pb.run_pair_strided_device(t1, t2, d.data_ptr(), n, length, stride, flags, i1.data_ptr(), i2.data_ptr(),
                                   fin.data_ptr(), stream)
        torch.cuda.synchronize()
        if n % 64 == 0:
            assert pb.last_kernel() == "pair_tiled"      # one fused pass (a remainder would add two generic ones)
        assert (i1.cpu().numpy().astype(np.uint32) == w1[0]).all(), (n1, n2, n, length)
        assert (i2.cpu().numpy().astype(np.uint32) == w2[0]).all(), (n1, n2, n, length)
        assert (fin.cpu().numpy() == (w1[1] | w2[1])).all()
    # ragged strings: two passes behind the same call
    rng = np.random.RandomState(3)
    strings = H.random_strings(rng, 3000, 200, b"abcdehlo wHeadInrTailfoobar.:/xq") + [b""] * 3
    text, offs = H.pack(strings)
    w1, w2 = o1.run(text, offs, threads=4), o2.run(text, offs, threads=4)
    dt = torch.as_tensor(text, device="cuda")
    do = torch.as_tensor(offs.astype(np.int64), device="cuda")
    n = len(strings)
    i1 = torch.empty(n, dtype=torch.int32, device="cuda")
    i2 = torch.empty(n, dtype=torch.int32, device="cuda")
    fin = torch.empty(n, dtype=torch.uint8, device="cuda")
    pb.run_pair_device(t1, t2, dt.data_ptr(), do.data_ptr(), n, BE, i1.data_ptr(), i2.data_ptr(), fin.data_ptr(), stream)
    torch.cuda.synchronize()
    assert (i1.cpu().numpy().astype(np.uint32) == w1[0]).all() and (i2.cpu().numpy().astype(np.uint32) == w2[0]).all()
    assert (fin.cpu().numpy() == (w1[1] | w2[1])).all()
