import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("EPS_TUNING_FROM_ENV", "1")   # (scripts steer the library's engine switches through the environment: vectordb_amd/_lib.py)
import vectordb_amd as amd
n,d=1_000_000,768
g=torch.Generator(device="cuda").manual_seed(42)
X=torch.empty((n,d),device="cuda")
for s in range(0,n,1<<19):
    e=min(n,s+(1<<19)); X[s:e]=torch.rand((e-s,d),generator=g,device="cuda")
Q=torch.rand((256,d),generator=torch.Generator(device="cuda").manual_seed(43),device="cuda")
ix=amd.GpuIndex(d,0).use_torch_stream(); ix.attach_rows(X); ix.build()
o=(torch.empty((256,10),dtype=torch.int64,device="cuda"),torch.empty((256,10),device="cuda"),torch.empty((256,),dtype=torch.int32,device="cuda"))
for nq in (1,256):
    for T in (1,4):
        kw=dict(mode=amd.MODE_GRAPH,intra_threads=T,master_queue=500,local_queue=500)
        oo=tuple(x[:nq] for x in o)
        ix.search(Q[:nq],10,out=oo,**kw); torch.cuda.synchronize()
        os.environ["EPS_TRV_PROF"]="1"
        ix.search(Q[:nq],10,out=oo,**kw); torch.cuda.synchronize()
        del os.environ["EPS_TRV_PROF"]
        lat=[]
        for i in range(20):
            torch.cuda.synchronize(); t=time.perf_counter(); ix.search(Q[i:i+nq] if nq==1 else Q[:nq],10,out=oo,**kw); torch.cuda.synchronize(); lat.append(time.perf_counter()-t)
        print(json.dumps({"nq":nq,"T":T,"p50_ms":1e3*float(np.median(lat)),"kernel_ms":ix.stats()["main_kernel_ms"]}),flush=True)
