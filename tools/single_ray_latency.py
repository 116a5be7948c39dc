"""Development probe: latency of ONE ray alone on the chip as a function of its work (node steps, leaf visits).
Launches of a single ray (count = 1): kernel time = launch floor + the ray's own critical path.  A least-squares fit
time = a + b*nodes + c*leaves over rays of different depth gives the cost of one dependent node step / leaf visit.
usage: single_ray_latency.py [workload cbvh.leaf|eager|tri] [n candidate rays]"""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
wl = sys.argv[1] if len(sys.argv) > 1 else 'cbvh.leaf'
ncand = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
if wl == 'tri':
    dev = rtc.Device('gpu=0,tri_accel=bvh8.triangle4v'); sc = rtc.Scene(dev)
    tris = []
    o = 0
    for n in fs:
        for k in range(1, n - 1): tris.append((fi[o], fi[o + k], fi[o + k + 1]))
        o += n
    sc.add_triangles(v, np.array(tris, dtype=np.uint32)); sc.commit()
else:
    dev = rtc.Device('gpu=0,subdiv_accel=' + ('bvh4.compressed.leaf' if wl == 'cbvh.leaf' else 'bvh4.subdivpatch1eager')); sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
rays = raygen.make_random_rays(ncand, v.min(0), v.max(0), seed=1)
rows = []
for i in range(ncand):
    b = torch.from_numpy(rays[i:i + 1].copy()).cuda()
    c = sc.intersect1M_counted(b)
    rows.append((c['nodeVisits'], c['leafVisits'] if 'leafVisits' in c else c['primTests'], i))
rows.sort()
# pick ~24 rays spread over the depth range
pick = [rows[int(k * (len(rows) - 1) / 23)] for k in range(24)] + rows[-8:]
A, T = [], []
for nodes, leaves, i in pick:
    bs = [torch.from_numpy(rays[i:i + 1].copy()).cuda() for _ in range(40)]
    for x in bs[:5]: sc.intersect1M(x, check=False)
    dev.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for x in bs[5:]: sc.intersect1M(x, check=False)
    dev.synchronize()
    us = (time.perf_counter() - t0) * 1e6 / 35
    A.append((1.0, nodes, leaves)); T.append(us)
    print('ray %6d: nodes %3d leaves %3d: %.2f us per launch' % (i, nodes, leaves, us))
A, T = np.array(A), np.array(T)
x, *_ = np.linalg.lstsq(A, T, rcond=None)
print('fit: %.2f us floor + %.3f us per node step + %.3f us per leaf visit (rms %.2f us)' % (x[0], x[1], x[2], np.sqrt(np.mean((A @ x - T) ** 2))))
