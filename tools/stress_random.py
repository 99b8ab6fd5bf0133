import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytest
import tests.test_random_scanners as T
import pire_amd
from tests.conftest import _Cfg
ok=0; skipped=0
for seed in range(24, 324):
    mp=_Cfg()
    try:
        T.test_random_scanner_all_kernels(pire_amd, seed, mp); ok+=1
    except pytest.skip.Exception:
        skipped+=1
    except BaseException as e:
        print("FAIL seed", seed, repr(e)[:300]); break
    finally:
        mp.restore()
print("random scanner seeds ok:", ok, "skipped:", skipped)
