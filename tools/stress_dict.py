import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import pire_amd
import tests.test_random_dictionaries as T
from tests.conftest import _Cfg
first, last = int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 110
ok = skipped = 0
for seed in range(first, last):
    mp = _Cfg()
    print('seed', seed, flush=True)
    try:
        r = T.run_seed(pire_amd, torch, mp, seed, verbose=bool(os.environ.get('STRESS_VERBOSE')))
        ok += r == "ok"
        skipped += r == "skipped"
    except BaseException as e:
        print("FAIL seed", seed, repr(e)[:600]); break
    finally:
        mp.restore()
print("random dictionary seeds ok:", ok, "skipped:", skipped)
