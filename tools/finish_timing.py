import numpy as np, time, sys
sys.path.insert(0, "/root/repo")
import ngsdist_amd as N
n=65*124750
rng=np.random.default_rng(0)
cnt=np.full(n,500000,dtype=np.uint64)
s=rng.random(n)*0.3*500000
out=np.empty(n)
for rep in range(4):
    t=time.perf_counter(); N.finish(s,cnt,0,1,out=out); t1=time.perf_counter()-t
    t=time.perf_counter()
    k=n//8
    for c in range(8): N.finish(s[c*k:(c+1)*k],cnt[c*k:(c+1)*k],0,1,out=out[c*k:(c+1)*k])
    t8=time.perf_counter()-t
    t=time.perf_counter()
    for c in range(65): N.finish(s[c*124750:(c+1)*124750],cnt[c*124750:(c+1)*124750],0,1,out=out[c*124750:(c+1)*124750])
    t65=time.perf_counter()-t
    m=62437
    t=time.perf_counter(); N.finish(s[:m],cnt[:m],0,1,out=out[:m]); ts=time.perf_counter()-t
    print("one call %.2f ms, 8 calls %.2f ms, 65 calls %.2f ms; 62k cells %.3f ms"%(t1*1e3,t8*1e3,t65*1e3,ts*1e3))
import os; print("cpus", os.cpu_count())
