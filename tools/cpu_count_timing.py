import sys,time,os,numpy as np
sys.path.insert(0,'oracle')
import pyoracle as po
rng=np.random.default_rng(1)
n=int(sys.argv[1])
s=np.frombuffer(b"ACGT",dtype=np.uint8)[rng.integers(0,4,n,dtype=np.uint8)]
cores=len(os.sched_getaffinity(0))
for nt in (cores, cores//2, 64):
    t=time.perf_counter(); k,c=po.count(s,15,3,nthreads=nt); print("count nt=%d"%nt,time.perf_counter()-t, len(k), flush=True)
