"""The benchmark batches that repeat a base of records (bench.py --set dict_*, tools/wide_case.py): every repeat is rotated,
so that no two waves of a CU ever walk the same records -- a plain repeat's period met the kernels' task order once, and
the heavy corpora measured twice their rate (lesson 21)."""
import numpy as np
import pytest

from pire_amd import workloads as W


@pytest.mark.parametrize("n,nbase", [(1 << 20, 16384), (1 << 20, 4096), (1 << 18, 4096), (1 << 22, 16384)])
def test_every_repeat_is_the_base_rotated(n, nbase):
    order = W.rotated_repeat_order(n, nbase)
    assert order.shape == (n,) and order.min() == 0 and order.max() == nbase - 1
    reps = order.reshape(n // nbase, nbase)
    assert (reps[0] == np.arange(nbase)).all()                       # the first repeat IS the base (the CPU sample's strings)
    for r in range(1, n // nbase):
        assert (reps[r] == np.roll(np.arange(nbase), -((1237 * r) % nbase))).all()
    assert len({int(x[0]) for x in reps}) == min(n // nbase, nbase)   # no two repeats start at the same record


@pytest.mark.parametrize("per_task,blocks", [(64, 256), (128, 256), (64, 304), (128, 128)])
@pytest.mark.parametrize("n,nbase", [(1 << 20, 16384), (1 << 22, 16384)])
def test_no_two_waves_of_a_block_walk_the_same_records(n, nbase, per_task, blocks):
    """Both task orders the kernels have used: wave w of block b on task 16 b + w, or on task b + blocks * w."""
    order = W.rotated_repeat_order(n, nbase)
    tasks = order.reshape(n // per_task, per_task)
    ntasks = tasks.shape[0]
    for name, task_of in (("block-major", lambda b, w: 16 * b + w), ("round the blocks", lambda b, w: b + blocks * w)):
        for b in range(0, blocks, 7):
            seen = set()
            for w in range(16):
                t = task_of(b, w)
                if t >= ntasks:
                    continue
                recs = set(tasks[t].tolist())
                assert not (seen & recs), (name, b, w)
                seen |= recs


def test_first_record_of_a_shard_follows_the_global_index():
    a = W.rotated_repeat_order(1 << 16, 4096, first=0)
    b = W.rotated_repeat_order(1 << 16, 4096, first=3 << 16)          # rank 3 of a weak-scaling run
    assert ((b - a) % 4096 == (3 << 16) % 4096).all()
