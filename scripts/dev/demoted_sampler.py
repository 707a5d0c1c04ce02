import sys, tempfile; sys.path.insert(0,'/root/repo')
import numpy as np, numpy.random as npr
from spearmint_amd.engine import Engine
from spearmint_amd.chooser import GPEIOptChooser
import spearmint_amd.chooser._base as b, spearmint_amd.chooser.GPEIOptChooser as o
b.log=o.log=lambda *a: None
N,D=40,3
rs=np.random.RandomState(N); G=400
grid=rs.rand(N+G,D); values=np.full(N+G,np.nan); values[:N]=np.sin(3*grid[:N]).sum(axis=1)+0.05*rs.randn(N)
complete,candidates,pending=np.arange(N),np.arange(N,N+G),np.array([],dtype=int)
def run(extra, provoke):
    ch=GPEIOptChooser.init(tempfile.mkdtemp(),"mcmc_iters=4,burnin=6,grid_subset=3,use_multiprocessing=0,"+extra)
    eng=ch.engine()
    if provoke:
        comp=grid[:N]; vals=values[:N]
        eng.set_observations(comp,vals)
        row=np.concatenate(([vals.mean(),1e-3,1.0],np.ones(D)))[None,:]
        eng.set_option("flow_spin_limit",1)
        eng.set_hypers(np.repeat(row,17,axis=0)); eng.gp_logprob()
        eng.set_option("flow_spin_limit",0)
        print("  after provoked call: fallbacks",eng.stat("flow_fallbacks"),"enabled",eng.stat("flow_enabled"), eng.last_warning())
    npr.seed(77)
    try:
        job=ch.next(grid,values,np.ones(N+G),candidates,pending,complete)
        res=(job, np.array([np.concatenate(([h[0],h[1],h[2]],h[3])) for h in ch.hyper_samples]))
    except Exception as ex:
        res=("ERR "+type(ex).__name__+": "+str(ex)[:80], None)
    print("  ", extra, "provoke" if provoke else "", "->", res[0], "fallbacks", eng.stat("flow_fallbacks"), ch.sampler_stats.get("calls"))
    return res
a=run("sampler=python,lookahead=6,follow=0:0", False)
for extra in ("sampler=native,lookahead=6,follow=0:0","sampler=python,lookahead=6,follow=0:0","sampler=native"):
    for provoke in (False, True):
        r=run(extra, provoke)
        if r[1] is not None: print("     same chain:", np.array_equal(r[1],a[1]))
