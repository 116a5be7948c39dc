"""Development check: digest of the hit records of one seeded batch (metric workload by default), plain and counted kernels, repeated.
Run it once per library build / knob setting (RTAMD_LIB=..., RTAMD_*=...) and compare the lines: equal digests = byte-identical hits.
usage: hits_digest.py [rays] [reps] [levels,C] [accel]"""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
d = np.load(os.path.join(root, 'assets/bomberman.mesh.npz')); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L, C = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '6,3').split(','))
accel = sys.argv[4] if len(sys.argv) > 4 else 'bvh4.compressed.leaf'
dev = rtc.Device('gpu=0,subdiv_accel=' + accel); sc = rtc.Scene(dev)
sc.add_subdiv(v, fs, fi); sc.set_levels(L, C); sc.commit()
rays = raygen.make_random_rays(n, v.min(0), v.max(0), seed=0)
out = []
for rep in range(reps):
    for counted in (False, True):
        b = torch.from_numpy(rays.copy()).cuda()
        c = sc.intersect1M_counted(b) if counted else sc.intersect1M(b)
        dev.synchronize()
        g = b.cpu().numpy().view(np.uint32).reshape(n, 20)
        out.append('%s hits %d %s%s' % ('counted' if counted else 'plain', int((g[:, 18] != 0xFFFFFFFF).sum()), hashlib.sha256(g.tobytes()).hexdigest()[:16],
                                         (' leaves %d prims %d inner %d' % (c['leafVisits'], c['primTests'], c.get('innerVisits', 0))) if counted else ''))
print(os.environ.get('RTAMD_LIB', 'lib').split('/')[-2] if 'RTAMD_LIB' in os.environ else 'lib', accel, (L, C), ' | '.join(sorted(set(out))))
