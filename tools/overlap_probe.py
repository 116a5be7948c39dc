"""Development probe: do two ray batches in flight (two HIP streams) hide each other's drain?
Builds the cbvh.leaf scene on two RTCDevice objects (each has its own stream, queue heads and spill area) and alternates
steps between them; compares with the same number of steps on one device."""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
K = 16
def scene():
    dev = rtc.Device('gpu=0,subdiv_accel=bvh4.compressed.leaf'); sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
    return dev, sc
pairs = [scene() for _ in range(4)]
bufs = [torch.from_numpy(raygen.make_random_rays(1000000, lo, hi, seed=100 + s)).cuda() for s in range(2 * K)]
torch.cuda.synchronize()
def run(ndev, off):
    for i in range(ndev): pairs[i][1].intersect1M(bufs[off + i], check=False)
    for p in pairs: p[0].synchronize()
    t0 = time.perf_counter()
    for s in range(K): pairs[s % ndev][1].intersect1M(bufs[off + s], check=False)
    for p in pairs: p[0].synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for n in (1, 2, 3, 4):
    us = run(n, 0 if n % 2 else K)
    print('%d batches in flight: %.1f us per 1M-ray step -> %.0f Mrays/s' % (n, us, 1e6 / us))
