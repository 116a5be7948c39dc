import sys, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, pyoracle as po
rtc = importlib.import_module('embree-compressed_amd').rtc
d=np.load('/root/repo/assets/bomberman.mesh.npz'); v,fs,fi=d['verts'],d['face_sizes'],d['face_index']
accel=sys.argv[1]; L=int(sys.argv[2]); Cl=int(sys.argv[3]); mode={'bvh4.compressed.box':3,'bvh4.compressed.leaf':4,'bvh4.compressed.grid':5,'default':2}[accel]
dev=rtc.Device('subdiv_accel='+accel); sc=rtc.Scene(dev); sc.add_subdiv(v,fs,fi); sc.set_levels(L,Cl); sc.commit()
st=sc.stats(); orc=po.SubdivScene(sc.accel_data(2), st['primBytes'], mode, Cl, qnodes=sc.accel_data(0), root=sc.accel_root())
lo,hi=v.min(0),v.max(0)
want=po.make_random_rays(200000,lo,hi,seed=0,double_eval=True); got=want.copy()
orc.intersect1M(want,nthreads=8); sc.intersect1M(got)
gh=got['geomID']!=0xFFFFFFFF; wh=want['geomID']!=0xFFFFFFFF
print('hits',gh.sum(),wh.sum(),'hitdiff',(gh!=wh).sum(),'primdiff',(got['primID'][gh&wh]!=want['primID'][gh&wh]).sum())
both=gh&wh
for f in ('tfar','u','v'):
    a=got[f][both].astype(np.float64); b=want[f][both].astype(np.float64)
    err=np.abs(a-b); rel=err/np.maximum(np.abs(b),1e-3)
    k=np.argsort(-rel)[:5]
    print(f,'max abs',err.max(),'max rel',rel.max(),'n>1e-4',(rel>1e-4).sum(),'n exact',(err==0).sum(),'of',both.sum())
    for i in k: print('   ',a[i],b[i],err[i])
