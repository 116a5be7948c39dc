"""Development check: the child-parallel (octet) node step against the lane-per-ray node step on the same rays, uncounted and
counted kernel twins, repeated (ray-to-wave assignment is dynamic, so repetitions exercise different lane mixes)."""
import importlib, os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def build(accel, octmax):
    os.environ['RTAMD_OCT_MAX'] = str(octmax)
    dev = rtc.Device('gpu=0,subdiv_accel=' + accel); sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
    return dev, sc
rays = raygen.make_random_rays(n, v.min(0), v.max(0), seed=0)
for accel in ('default', 'bvh4.compressed.leaf'):
    d0, s0 = build(accel, 0)
    ref = torch.from_numpy(rays.copy()).cuda(); s0.intersect1M(ref); d0.synchronize()
    ref = ref.cpu().numpy()
    refw = ref.view(np.uint32).reshape(n, 20)
    print(accel, 'reference hits', int((refw[:, 18] != 0xFFFFFFFF).sum()))
    for octmax in (0, 8, 16):
        d1, s1 = build(accel, octmax)
        for rep in range(reps):
            for counted in (False, True):
                b = torch.from_numpy(rays.copy()).cuda()
                c = s1.intersect1M_counted(b) if counted else s1.intersect1M(b)
                d1.synchronize()
                g = b.cpu().numpy()
                gw = g.view(np.uint32).reshape(n, 20)
                bad = np.unique(np.nonzero(gw != refw)[0])
                msg = '' if not counted else ' counter hits %d nodes %d' % (c['hits'], c['nodeVisits'])
                print('  oct %2d rep %d %s: %d rays differ%s' % (octmax, rep, 'counted' if counted else 'plain  ', len(bad), msg))
                for i in bad[:4]:
                    print('     ray %d: got geom %d prim %d t %.6g | ref geom %d prim %d t %.6g' % (i, gw[i, 18], gw[i, 17], gw[i, 8:9].view(np.float32)[0], refw[i, 18], refw[i, 17], refw[i, 8:9].view(np.float32)[0]))
