import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import pytest
import tests.test_random_scanners as T
import pire_amd
class MP:
    def __init__(self): self.saved={}
    def setenv(self,k,v): self.saved.setdefault(k,os.environ.get(k)); os.environ[k]=v
    def delenv(self,k): self.saved.setdefault(k,os.environ.get(k)); os.environ.pop(k,None)
    def undo(self):
        for k,v in self.saved.items():
            if v is None: os.environ.pop(k,None)
            else: os.environ[k]=v
        self.saved={}
ok=0; skipped=0
for seed in range(24, 324):
    mp=MP()
    try:
        T.test_random_scanner_all_kernels(pire_amd, seed, mp); ok+=1
    except pytest.skip.Exception:
        skipped+=1
    except BaseException as e:
        print("FAIL seed", seed, repr(e)[:300]); break
    finally:
        mp.undo()
print("random scanner seeds ok:", ok, "skipped:", skipped)
