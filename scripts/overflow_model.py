"""Scaled model of the compact table's overflow (k = 14, m = 11, the engine's own scramblers from minimizer_sim.py): share of
the k-mers that do not fit their 12-slot bucket as a function of the mean load AND of r = k-mers per canonical m-mer.  The data
behind cpt_displaced_share (lmat_api.cpp) and behind the diagnosis of the 2^32-bucket build that did not finish in round 2
(profiles/r03_overflow_model.txt)."""
import sys, numpy as np
sys.path.insert(0,'/root/repo/scripts')
import importlib.util
spec=importlib.util.spec_from_file_location('ms','/root/repo/scripts/minimizer_sim.py'); ms=importlib.util.module_from_spec(spec)
sys.argv=['x','12','6.4','engine']; spec.loader.exec_module(ms)
rng=np.random.default_rng(5)
m=11; k=m+3
def run(N, W):
    seq=rng.integers(0,4,N+k-1,dtype=np.uint8)
    keys=ms.canon(ms.words(seq,k),k)
    mm=ms.feistel(ms.canon(ms.words(seq,m),m),m,ms.SCR)
    P=len(seq)-k+1
    best=mm[0:P].copy()
    for j in range(1,4): best=np.minimum(best,mm[j:P+j])
    uk,idx=np.unique(keys,return_index=True)
    um=best[idx]
    sp=ms.feistel(um,m,ms.MIX)
    hi=sp>>np.uint64(2)
    b=(hi//np.uint64(W)).astype(np.int64)
    nb=(1<<(2*m-2))//W
    cnt=np.bincount(b,minlength=nb)
    disp=np.maximum(cnt-12,0).sum()/len(uk)
    return len(uk), nb, len(uk)/nb, disp, cnt.max()
space=4**m//2
for ratio in (0.7,2.24):
    for W in (1,2):
        n,nb,load,disp,mx=run(int(ratio*space),W)
        model=min(0.6,0.061*(load/6.4)**2.5+0.01)
        print(f"k-mers/canonical m-mers {ratio}: W={W} buckets {nb} load {load:.2f} displaced {disp*100:.1f}%  (sizing model {model*100:.1f}% x2)  max bucket {mx}")
print("---- grid")
import itertools
rows=[]
for ratio in (0.125,0.25,0.5,1.0,1.5,2.24,3.0):
    for W in (1,2,4,8):
        n,nb,load,disp,mx=run(int(ratio*space),W)
        rows.append((ratio,W,load,disp))
        print(f"{ratio:6.3f} W={W} load {load:6.2f} displaced {disp*100:5.1f}%  old model {min(0.6,0.061*(load/6.4)**2.5+0.01)*100:5.1f}%")
