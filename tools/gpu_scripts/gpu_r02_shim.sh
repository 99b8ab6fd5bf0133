#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/r02shim
mkdir -p $OUT
timeout 600 tests/cpp/bin/shim_test 2>&1 | tail -5 | tee $OUT/shim.log
echo "== host-pointer mode through the python binding (pageable numpy): one-shot against pipelined"
timeout 600 python - <<'PY' 2>&1 | tee $OUT/pcie.log
import os, time, numpy as np, torch
import pire_amd
from oracle import binding as ob
from tests import helpers as H
big=[b for b in H.big_sets() if b["name"]=="set_a"][0]
t=pire_amd.Table(H.load_blob(big["blob"])); t.upload()
for n,L in ((1<<16,4096),(1<<18,4096),(1<<19,4096)):
    data=ob.corpus_fill(0x5EED5EED,0,n,L,H.plants_for(big),threads=32)
    t.run_strided_host(data[:1024])
    ref=None
    for mode in ("one-shot","pipelined"):
        if mode=="one-shot": os.environ["PIRE_HIP_HOST_ONE_SHOT"]="1"
        else: os.environ.pop("PIRE_HIP_HOST_ONE_SHOT",None)
        best=1e9
        for _ in range(3):
            t0=time.perf_counter(); idx,fin=t.run_strided_host(data); dt=time.perf_counter()-t0; best=min(best,dt)
        if ref is None: ref=(idx.copy(),fin.copy())
        ok=(idx==ref[0]).all() and (fin==ref[1]).all()
        print("host-pointer mode %-9s: %d x %d B (%.0f MiB pageable): %.1f ms -> %.2f GB/s, same results %s" % (mode,n,L,n*L/2**20,best*1e3,n*L/best/1e9, ok))
    # ragged through offsets: log-line like strings
    lens=np.random.RandomState(3).randint(64,1024,size=n).astype(np.uint64)
    offs=np.zeros(n+1,dtype=np.uint64); offs[1:]=np.cumsum(lens)
    flat=data.reshape(-1)[:int(offs[-1])]
    res={}
    for mode in ("one-shot","pipelined"):
        if mode=="one-shot": os.environ["PIRE_HIP_HOST_ONE_SHOT"]="1"
        else: os.environ.pop("PIRE_HIP_HOST_ONE_SHOT",None)
        t0=time.perf_counter(); idx,fin=t.run(flat,offs)[:2]; dt=time.perf_counter()-t0
        res[mode]=(idx,fin)
        print("host-pointer mode %-9s ragged: %d strings, %.0f MiB: %.1f ms -> %.2f GB/s" % (mode,n,int(offs[-1])/2**20,dt*1e3,int(offs[-1])/dt/1e9))
    print("ragged same results", (res["one-shot"][0]==res["pipelined"][0]).all() and (res["one-shot"][1]==res["pipelined"][1]).all())
PY
timeout 600 python -m pytest tests/test_shim_cpp.py tests/test_abi.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
