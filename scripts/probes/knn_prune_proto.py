"""Developer probe (CPU, numpy): how many refs would a tile-level pruned exact search still have to visit?  k-means cells,
per-query lower bounds (|q - centre| - radius)^2 reduced over a 128-query block, compared with the block's final k-th distance."""
import numpy as np, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import bench
from scipy.spatial import cKDTree
def blobs(n, d, seed, scale):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 10, size=n)
    return rng.normal(size=(10, d))[lab] * scale + rng.normal(size=(n, d))
def kmeans(X, nc, iters=4, seed=0):
    rng=np.random.default_rng(seed)
    C=X[rng.choice(len(X),nc,replace=False)].copy()
    for it in range(iters):
        d2=(X*X).sum(1)[:,None]+(C*C).sum(1)[None,:]-2*X@C.T
        a=d2.argmin(1)
        for c in range(nc):
            m=a==c
            if m.any(): C[c]=X[m].mean(0)
    d2=(X*X).sum(1)[:,None]+(C*C).sum(1)[None,:]-2*X@C.T
    a=d2.argmin(1)
    return C,a
def run(name,X,k,cellsize=512,BQ=128):
    n,d=X.shape
    nc=max(8,n//cellsize)
    t0=time.time(); C,a=kmeans(X,nc); 
    order=np.argsort(a,kind='stable'); Xs=X[order]; a_s=a[order]
    rad=np.zeros(nc); cnt=np.bincount(a_s,minlength=nc)
    dc=np.sqrt(((Xs-C[a_s])**2).sum(1)); np.maximum.at(rad,a_s,dc)
    # true kth distances (sample of blocks)
    tree=cKDTree(Xs); 
    nb=(n+BQ-1)//BQ
    rng=np.random.default_rng(1); blocks=rng.choice(nb,size=min(nb,60),replace=False)
    fr=[]; fr_ideal=[]
    for b in blocks:
        q=Xs[b*BQ:(b+1)*BQ]
        dk=tree.query(q,k=k)[0][:,-1]
        T_B=(dk**2).max()
        dqc=np.sqrt(np.maximum(0,(q*q).sum(1)[:,None]+(C*C).sum(1)[None,:]-2*q@C.T))
        lb=np.maximum(0,dqc-rad[None,:])**2
        LBmin=lb.min(0)
        visit=LBmin<T_B          # with the FINAL thresholds (best case)
        fr.append(cnt[visit].sum()/n)
        # with thresholds 2x the final (early, looser)
        fr_ideal.append(cnt[LBmin<2*T_B].sum()/n)
    print('%s: n=%d d=%d cells=%d (kmeans %.1fs): refs visited with final thresholds %.1f%% (thresholds x2: %.1f%%), mean cell radius %.2f, kth dist %.2f'%(name,n,d,nc,time.time()-t0,100*np.mean(fr),100*np.mean(fr_ideal),rad.mean(),np.sqrt(T_B)))
X2=bench.make_features(bench.load_labels(70000)); run('config2',X2,11)
run('config2 cells 128',X2,11,cellsize=128)
run('config3',blobs(60000,32,1,1.2),21)
run('config4-shape 200k',blobs(200000,64,2,4.0),11)
run('no clusters d=20',np.random.default_rng(5).normal(size=(50000,20)),11)
