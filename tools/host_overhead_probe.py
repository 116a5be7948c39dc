"""Development probe: host-side cost of one device-resident rtcIntersect1M call (Python/ctypes + launch-context pick +
memset + launch + event record), measured by enqueuing many tiny batches without synchronising in between."""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
dev = rtc.Device('gpu=0,tri_accel=bvh8.triangle4v'); sc = rtc.Scene(dev)
sc.add_triangles(v, rtc.fan_triangulate(fs, fi)); sc.commit()
buf = torch.from_numpy(raygen.make_random_rays(64, v.min(0), v.max(0), seed=1)).cuda()
streams = [torch.cuda.Stream() for _ in range(4)]
for n_streams in (1, 4):
    for _ in range(50): sc.intersect1M(buf, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 2000
    for i in range(N):
        if n_streams > 1: dev.set_stream(streams[i % n_streams].cuda_stream)
        sc.intersect1M(buf, check=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%d stream(s): %.1f us host time per call (enqueue only), %.1f us per call including the final sync' % (n_streams, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
