"""The synthetic corpus generator: host mirror pinned by golden bytes; device generator == host mirror."""
import base64

import numpy as np
import pytest

from oracle import binding as ob
from tests import helpers as H


def test_host_corpus_matches_golden_bytes():
    c = H.golden()["corpus"]
    set_a = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    a = ob.corpus_fill(c["seed"], 0, 4, 100, None)
    assert a.tobytes() == base64.b64decode(c["noplant_first4_len100_b64"])
    b = ob.corpus_fill(c["seed"], 5, 12, 77, H.plants_for(set_a))
    assert b.tobytes() == base64.b64decode(c["planted_from5_count12_len77_b64"])
    assert a.min() >= 0x20 and a.max() <= 0x7E
    # plants land where the definition says: slot = s % 9, tail-anchored witnesses end the string
    wit = [bytes.fromhex(h) for h in set_a["witnesses_hex"]]
    for i in range(12):
        s = 5 + i
        slot = s % 9
        if slot:
            w = wit[slot - 1]
            assert b[i].tobytes().endswith(w)
    # threads do not change the bytes; string s does not depend on which batch it was generated in
    big = ob.corpus_fill(c["seed"], 0, 64, 77, H.plants_for(set_a), threads=4)
    assert (big[5:17] == b).all()


@pytest.mark.gpu
def test_device_corpus_equals_host_corpus():
    import torch
    import pire_amd

    set_a = [b for b in H.big_sets() if b["name"] == "set_a"][0]
    plants = H.plants_for(set_a)
    for first, count, length, stride in ((0, 1000, 4096, 4096), (12345, 333, 77, 80), (7, 65, 5, 5), (0, 3, 1, 16)):
        buf = torch.zeros(count * stride, dtype=torch.uint8, device="cuda")
        pire_amd.corpus_fill_device(buf.data_ptr(), 0xABCDEF, first, count, length, stride, plants,
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        dev = buf.cpu().numpy().reshape(count, stride)[:, :length]
        host = ob.corpus_fill(0xABCDEF, first, count, length, plants, threads=4)
        assert (dev == host).all()
