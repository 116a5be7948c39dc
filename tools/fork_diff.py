"""Development aid: rays on which the HIP kernels and the oracle (product arithmetic) differ on a fork cBVH mode, field by field."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import pyoracle as po
rtc = importlib.import_module('embree-compressed_amd').rtc
d = np.load(os.path.join(ROOT, 'assets', 'bomberman.mesh.npz')); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
accel = sys.argv[1] if len(sys.argv) > 1 else 'bvh4.compressed.leaf'
L, C, n = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 3, 1000000)
mode = {'bvh4.compressed.box': 3, 'bvh4.compressed.leaf': 4, 'bvh4.compressed.grid': 5}[accel]
dev = rtc.Device('subdiv_accel=' + accel); sc = rtc.Scene(dev)
sc.add_subdiv(v, fs, fi); sc.set_levels(L, C); sc.commit()
orc = po.SubdivScene(sc.accel_data(2), sc.stats()['primBytes'], mode, C, qnodes=sc.accel_data(0), root=sc.accel_root())
got = po.make_random_rays(n, v.min(0), v.max(0), seed=0, double_eval=True)
want = got.copy()
sc.intersect1M(got)
with po.fork_arith(1):
    orc.intersect1M(want, nthreads=8)
gw, ww = got.view(np.uint32).reshape(n, 20), want.view(np.uint32).reshape(n, 20)
bad = np.unique(np.nonzero(gw != ww)[0])
print(accel, 'L', L, 'C', C, ':', len(bad), 'records differ of', n)
names = got.dtype.names
import ctypes as C
L = po.lib()
L.orc_set_fork_trace.argtypes = [C.c_int]
src = po.make_random_rays(n, v.min(0), v.max(0), seed=0, double_eval=True)
for i in bad[:6]:
    cols = np.nonzero(gw[i] != ww[i])[0]
    print('  ray', i, ' '.join('%s got %r (%s) want %r (%s)' % (names[c], got[i][names[c]], hex(gw[i][c]), want[i][names[c]], hex(ww[i][c])) for c in cols), flush=True)
    print('  ray record:', ' '.join(float(x).hex() for x in src[i:i + 1].view(np.float32).reshape(-1)[:9]), flush=True)
    one = src[i:i + 1].copy()
    with po.fork_arith(1):
        L.orc_set_fork_trace(1)
        orc.intersect1M(one, nthreads=1)
        L.orc_set_fork_trace(0)
    sys.stdout.flush()
