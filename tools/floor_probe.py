"""Development probe: time rtcIntersect1M on device-resident batches for degenerate scenes to find fixed costs."""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
def timeit(sc, dev, bufs, K=10):
    for b in bufs[:2]: sc.intersect1M(b, check=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for b in bufs[2:2+K]: sc.intersect1M(b, check=False)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lo, hi = np.array([-242.4, -2.1, -246.1], np.float32), np.array([243.6, 21.4, 239.9], np.float32)
bufs = [torch.from_numpy(raygen.make_random_rays(m, lo, hi, seed=s)).cuda() for s in range(12)]
# (a) empty scene: kernel not launched at all (root empty)
dev = rtc.Device('gpu=0'); dev.set_stream(stream.cuda_stream); sc = rtc.Scene(dev); sc.commit()
print('empty scene (no launch)      %.4f ms' % timeit(sc, dev, bufs)); sc.release(); dev.release()
# (b) one far-away triangle: every ray misses the root box -> ray I/O + one leaf-root test
dev = rtc.Device('gpu=0,tri_accel=bvh8.triangle4v'); dev.set_stream(stream.cuda_stream); sc = rtc.Scene(dev)
sc.add_triangles(np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6]], np.float32), np.array([[0, 1, 2]], np.uint32)); sc.commit()
print('1 far triangle (root = leaf) %.4f ms' % timeit(sc, dev, bufs)); sc.release(); dev.release()
# (c) 100 far triangles: root node test, all children missed
v = (np.random.RandomState(1).rand(300, 3) * 10 + 1e5).astype(np.float32)
dev = rtc.Device('gpu=0,tri_accel=bvh8.triangle4v'); dev.set_stream(stream.cuda_stream); sc = rtc.Scene(dev)
sc.add_triangles(v, np.arange(300, dtype=np.uint32).reshape(-1, 3)); sc.commit()
print('100 far triangles (1 node)   %.4f ms' % timeit(sc, dev, bufs)); sc.release(); dev.release()
# (d) torch copy of the same bytes as a streaming yardstick
t = bufs[0].clone(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for k in range(10): t.copy_(bufs[k])
e1.record(stream); torch.cuda.synchronize()
print('torch copy 80 MB             %.4f ms' % (e0.elapsed_time(e1) / 10))
