#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
python - <<'PY'
import time, numpy as np, torch
import pire_amd
from pire_amd import binding as pb
from oracle import binding as ob
from tests import helpers as H
big=[b for b in H.big_sets() if b["name"]=="set_a"][0]
t=pire_amd.Table(H.load_blob(big["blob"])); t.upload()
plants=H.plants_for(big)
stream=torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(reps):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
n,L=1<<18,4096
buf=torch.empty((n,L),dtype=torch.uint8,device="cuda")
pire_amd.corpus_fill_device(buf.data_ptr(),0x5EED5EED,0,n,L,L,plants,stream)
idx=torch.empty(n,dtype=torch.int32,device="cuda"); fin=torch.empty(n,dtype=torch.uint8,device="cuda")
ms=timeit(lambda: t.run_strided_device(buf.data_ptr(),n,L,L,3|pb.FLAG_GENERIC,idx.data_ptr(),fin.data_ptr(),0,0,stream))
print("generic kernel, strided 2^18 x 4096: %.3f ms -> %.1f GB/s"%(ms,n*L/ms/1e6))
ms=timeit(lambda: t.run_strided_device(buf.data_ptr(),n,L,L,3,idx.data_ptr(),fin.data_ptr(),0,0,stream))
print("tiled kernel,   strided 2^18 x 4096: %.3f ms -> %.1f GB/s"%(ms,n*L/ms/1e6))
for name,lo,hi,m,mul in (("uniform 0..8191",0,8192,n//2,1),("uniform 0..8191 al16",0,512,n//2,16),("uniform 0..8191 al128",0,64,n//2,128),("short 20..200 (URLs)",20,200,1<<22,1),("log lines 64..1024",64,1024,1<<20,1),("log lines al16",4,64,1<<20,16),("log lines al128",1,8,1<<20,128),("fixed 4096",32,33,n,128)):
    rng=np.random.RandomState(1)
    lens=(rng.randint(lo,hi,size=m)*mul).astype(np.uint64)
    offs=np.zeros(m+1,dtype=np.uint64); offs[1:]=np.cumsum(lens)
    total=int(offs[-1]); assert total<=n*L, total
    doffs=torch.as_tensor(offs.astype(np.int64),device="cuda")
    idx=torch.empty(m,dtype=torch.int32,device="cuda"); fin=torch.empty(m,dtype=torch.uint8,device="cuda")
    for g in (0, pb.FLAG_GENERIC):
        ms=timeit(lambda: t.run_device(buf.data_ptr(),doffs.data_ptr(),m,3|g,idx.data_ptr(),fin.data_ptr(),0,0,stream))
        print("%-8s %-22s %8d strings %.2f GiB: %.3f ms -> %.1f GB/s"%(pb.last_kernel(),name,m,total/2**30,ms,total/ms/1e6))
PY
