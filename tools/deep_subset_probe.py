"""Development probe (GPU box): how much of the metric kernel's duration is the critical path of its deepest rays?
Per-ray cost (BVH8 node steps, blob visits, quadtree steps) of one 1 M-ray batch comes from the oracle (dev aid only); the plain
kernel is then timed (HIP events, one stream, fresh records per launch) on: the K costliest rays alone, the batch without them
(replaced by copies of cheap rays), the batch with the costliest rays moved to the front / to the back.
    python tools/deep_subset_probe.py [rays] > gpurun_out/.../deep_subset.txt"""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import torch
import pyoracle as po
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = np.load(os.path.join(ROOT, 'assets', 'bomberman.mesh.npz')); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
dev = rtc.Device('gpu=0,subdiv_accel=bvh4.compressed.leaf'); sc = rtc.Scene(dev)
sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); dev.set_stream(stream.cuda_stream)
orc = po.SubdivScene(sc.accel_data(2), sc.stats()['primBytes'], 4, 3, qnodes=sc.accel_data(0), root=sc.accel_root())
L = po.lib(); L.orc_get_fork_inner.restype = C.c_ulonglong
raw = np.ascontiguousarray(raygen.make_random_rays(n, lo, hi, seed=4242))  # uint8 [n, 80]
rays = raw.copy().reshape(-1).view(rtc.RAYHIT_DTYPE)
t0 = time.time()
out = np.zeros((n, 3), np.int64)
with po.fork_arith(1):
    for i in range(n):
        L.orc_reset_fork_inner()
        orc.intersect1M(rays[i:i + 1])
        c = orc.counters()
        out[i] = (c["nodes"], c["leaves"], L.orc_get_fork_inner())
nodes, blobs, inner = out[:, 0], out[:, 1], out[:, 2]
cost = nodes * 1.0 + blobs * 1.5 + inner * 0.5
order = np.argsort(-cost, kind='stable')
print('# per-ray cost by the oracle: %.0f s; nodes/ray %.3f blobs/ray %.3f inner/ray %.3f; costliest ray: nodes %d blobs %d inner %d' %
      (time.time() - t0, nodes.mean(), blobs.mean(), inner.mean(), nodes[order[0]], blobs[order[0]], inner[order[0]]), flush=True)
for q in (50, 90, 99, 99.9, 99.99):
    print('#   percentile %.2f: nodes %d blobs %d inner %d cost %.0f' % (q, np.percentile(nodes, q), np.percentile(blobs, q), np.percentile(inner, q), np.percentile(cost, q)))

def timed(batch, label, reps=20):
    src = torch.from_numpy(np.ascontiguousarray(batch)).to('cuda')
    work = [src.clone() for _ in range(reps + 3)]
    torch.cuda.synchronize()
    for k in range(3):
        sc.intersect1M(work[k], check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(3, reps + 3):
        sc.intersect1M(work[k], check=False)
    e1.record(stream)
    torch.cuda.synchronize()
    print('%-58s %8d rays  %8.1f us per launch' % (label, batch.shape[0], e0.elapsed_time(e1) / reps * 1e3), flush=True)

cheap = order[::-1]
timed(raw, 'the batch as generated')
for K in (64, 1000, 10000, 100000):
    timed(raw[order[:K]], 'the %d costliest rays alone' % K)
for K in (100, 1000, 10000, 100000):
    b = raw.copy(); b[order[:K]] = raw[cheap[:K]]
    timed(b, 'batch with the %d costliest rays replaced by cheap ones' % K)
timed(raw[order], 'batch sorted: costliest rays first')
timed(raw[order[::-1]], 'batch sorted: costliest rays last')
perm = np.random.RandomState(1).permutation(n)
timed(raw[perm], 'batch shuffled')
b = raw[perm].copy(); K = 4096; b[:K] = raw[order[:K]]; rest = np.setdiff1d(perm, order[:K], assume_unique=False)
timed(np.concatenate([raw[order[:K]], raw[rest]]), 'the 4096 costliest first, rest shuffled')
