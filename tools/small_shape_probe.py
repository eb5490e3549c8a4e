import numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from lia_ral_amd import capi
from oracle import oracle as orc
ctx=capi.Context(0)
rng=np.random.default_rng(0)
for C,D,T in ((2,1,26),(2,1,1000),(2,2,26),(3,1,50),(16,1,100),(2,4,26),(1,1,10),(1,3,10),(5,2,33)):
    w=np.full(C,1.0/C); mean=np.linspace(-2,2,C*D).reshape(C,D); cov=np.ones((C,D))
    x=rng.normal(size=(T,D)).astype(np.float32)
    g=ctx.gmm(w,mean,1/cov)
    og=orc.Gmm(w,mean,1/cov)
    for opts in ({}, {"stats_z":0}, {"short_calls":0}):
        for k,v in opts.items(): ctx.set_option(k,v)
        a=g.split_acc(g.em_accumulate(x)); r=orc.em_accumulate(og,x.astype(np.float64))
        l=g.llk(x); lo=orc.llk(og,x.astype(np.float64))
        d=g.llk_determine_top(x,min(C,2)); do=orc.llk_determine_top(og,x.astype(np.float64),min(C,2),True)
        N=np.zeros((1,C)); F=np.zeros((1,C*D)); g.tv_stats(x,np.array([0,T]),N,F)
        print(C,D,T,opts,"occ",np.abs(a["occ"]-r["occ"]).max(),"sx",np.abs(a["sx"]-r["sx"]).max(),"sxx",np.abs(a["sxx"]-r["sxx"]).max(),"llk",np.abs(l-lo).max(),"idx",np.array_equal(d["idx"],do["idx"]),"N",np.abs(N[0]-r["occ"]).max(),"F",np.abs(F[0]-r["sx"].ravel()).max())
        for k,v in opts.items(): ctx.set_option(k,{"stats_z":1,"short_calls":1}[k])
    g.close()
