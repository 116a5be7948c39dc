"""Development probe (CPU, oracle): per-ray work of the metric workload - outer BVH8 node visits, blob visits and quadtree steps inside the blobs -
for the deepest rays of a sample, i.e. what the critical path of a batch consists of.  Usage: python tools/ray_cost_profile.py [sample]"""
import ctypes as C, importlib, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np
import pyoracle as po
rtc = importlib.import_module('embree-compressed_amd').rtc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
dev = rtc.Device('gpu=none,subdiv_accel=bvh4.compressed.leaf'); sc = rtc.Scene(dev)
sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
orc = po.SubdivScene(sc.accel_data(2), sc.stats()['primBytes'], 4, 3, qnodes=sc.accel_data(0), root=sc.accel_root())
L = po.lib(); L.orc_get_fork_inner.restype = C.c_ulonglong
rays = po.make_random_rays(n, lo, hi, seed=0)
out = np.zeros((n, 3), np.int64)
with po.fork_arith(1):
    for i in range(n):
        L.orc_reset_fork_inner()
        orc.intersect1M(rays[i:i + 1])
        c = orc.counters()
        out[i] = (c["nodes"], c["leaves"], L.orc_get_fork_inner())
nodes, blobs, inner = out[:, 0], out[:, 1], out[:, 2]
print('rays %d: nodes/ray %.3f blobs/ray %.3f inner/ray %.3f inner/blob %.2f' % (n, nodes.mean(), blobs.mean(), inner.mean(), inner.sum() / max(1, blobs.sum())))
cost = nodes * 1.0 + blobs * 1.5 + inner * 0.5   # rough us: octet node step, blob setup, quadtree step (tools/single_ray_latency.py scale)
order = np.argsort(-cost)
print('deepest rays (rough cost model us = nodes*1.0 + blobs*1.5 + inner*0.5):')
for k in order[:12]:
    print('  ray %7d: nodes %3d blobs %3d inner %4d (%.1f per blob)  cost %.0f us, of which quadtree steps %.0f %%' % (k, nodes[k], blobs[k], inner[k], inner[k] / max(1, blobs[k]), cost[k], 100 * inner[k] * 0.5 / cost[k]))
for q in (50, 90, 99, 99.9, 99.99):
    print('  percentile %.2f: nodes %d blobs %d inner %d' % (q, np.percentile(nodes, q), np.percentile(blobs, q), np.percentile(inner, q)))
top = order[: max(1, n // 1000)]
print('top 0.1 %% of rays: nodes %.1f blobs %.1f inner %.1f; share of quadtree steps in their cost %.0f %%' % (nodes[top].mean(), blobs[top].mean(), inner[top].mean(), 100 * (inner[top] * 0.5).sum() / cost[top].sum()))
