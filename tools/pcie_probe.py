"""Development probe: rate of one rtcIntersect1M on HOST records (staging + H2D + traversal + D2H + scatter), pipelined path vs
the unpipelined one, pageable numpy memory.  Usage: python tools/pcie_probe.py [rays]"""
import importlib, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
rays = raygen.make_random_rays(n, lo, hi, seed=1).reshape(-1).view(rtc.RAYHIT_DTYPE)
for label, env in (('unpipelined', {'RTAMD_PIPE_MIN': '2000000000'}), ('pipelined 128k x 8 threads', {}), ('pipelined 64k', {'RTAMD_PIPE_CHUNK': '65536'}),
                   ('pipelined 256k', {'RTAMD_PIPE_CHUNK': '262144'}), ('pipelined 128k x 4 threads', {'RTAMD_HOST_THREADS': '4'}),
                   ('pipelined 128k x 12 threads', {'RTAMD_HOST_THREADS': '12'}), ('pipelined 128k x 16 threads', {'RTAMD_HOST_THREADS': '16'})):
    for k in ('RTAMD_PIPE_MIN', 'RTAMD_PIPE_CHUNK', 'RTAMD_HOST_THREADS'): os.environ.pop(k, None)
    os.environ.update(env)
    dev = rtc.Device('gpu=0,subdiv_accel=bvh4.compressed.leaf'); sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
    best = 1e9
    for rep in range(5):
        buf = rtc.aligned_rayhits(n); buf[:] = rays
        t0 = time.perf_counter(); sc.intersect1M(buf); best = min(best, time.perf_counter() - t0)
    print('%-28s %d rays: %.2f ms  %.0f Mrays/s  (hits %d)' % (label, n, best * 1e3, n / best / 1e6, int((buf['geomID'] != 0xFFFFFFFF).sum())), flush=True)
    sc.release(); dev.release()
