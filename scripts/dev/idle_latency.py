import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
from spearmint_amd.engine import Engine
from spearmint_amd.synthetic import synthetic_problem
eng=Engine(0)
comp,cand,vals,hyp=synthetic_problem(256,16,8,8,1)
eng.set_observations(comp,vals); eng.set_hypers(hyp); eng.gp_logprob()
for what in ("set_observations","gp_logprob"):
  for d in (0.0,0.002,0.01,0.03,0.1,0.5):
    ts=[]
    for i in range(40 if d<0.5 else 12):
        if d: time.sleep(d)
        t=time.perf_counter()
        if what=="set_observations": eng.set_observations(comp,vals)
        else: eng.set_hypers(hyp); eng.gp_logprob()
        ts.append((time.perf_counter()-t)*1e3)
    ts=np.array(ts); print("%-16s idle %.3f s: median %.3f ms  max %.3f  >1ms: %d of %d"%(what,d,np.median(ts),ts.max(),(ts>1).sum(),len(ts)),flush=True)
# busy host (numpy work) instead of sleep
for what in ("set_observations",):
    ts=[]
    for i in range(40):
        a=np.random.rand(400,400); b=a@a; c=np.linalg.cholesky(b@b.T+400*np.eye(400))
        t=time.perf_counter(); eng.set_observations(comp,vals); ts.append((time.perf_counter()-t)*1e3)
    ts=np.array(ts); print("%-16s after numpy BLAS work: median %.3f ms max %.3f >1ms: %d of %d"%(what,np.median(ts),ts.max(),(ts>1).sum(),len(ts)))
