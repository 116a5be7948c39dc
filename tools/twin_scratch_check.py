"""Bug hunt (round 2): reproduce the scratch-dependent wrong records of the instrumented eager twin on a historic build
and classify them.  usage: hunt.py <tree root> [reps]   (tree root contains embree-compressed_amd/ and assets come from /root/repo)"""
import importlib, os, sys
import numpy as np, torch
root = os.path.abspath(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sys.path.insert(0, root)
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
print('library', rtc.LIB_PATH, flush=True)
d = np.load(os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'assets', 'bomberman.mesh.npz'))
v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
n = 1000000
rays = raygen.make_random_rays(n, v.min(0), v.max(0), seed=0)
perq = (n + 63) // 64
import itertools
for accel, octmax in itertools.product(('default', 'bvh4.compressed.leaf'), (0, 16)):
    os.environ['RTAMD_OCT_MAX'] = str(octmax)
    dev = rtc.Device('gpu=0,subdiv_accel=' + accel); sc = rtc.Scene(dev)
    sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
    ref = torch.from_numpy(rays.copy()).cuda(); sc.intersect1M(ref); dev.synchronize()
    refw = ref.cpu().numpy().view(np.uint32).reshape(n, 20)
    refhit = refw[:, 18] != 0xFFFFFFFF
    print(accel, 'octmax', octmax, 'reference hits', int(refhit.sum()), flush=True)
    for rep in range(reps):
        b = torch.from_numpy(rays.copy()).cuda()
        c = sc.intersect1M_counted(b); dev.synchronize()
        gw = b.cpu().numpy().view(np.uint32).reshape(n, 20)
        bad = np.unique(np.nonzero(gw != refw)[0])
        gothit = gw[:, 18] != 0xFFFFFFFF
        lost = int((refhit[bad] & ~gothit[bad]).sum()); false_ = int((~refhit[bad] & gothit[bad]).sum()); other = len(bad) - lost - false_
        print('  rep %d: %d rays differ (lost hit %d, false hit %d, other %d) | counters hits %d nodes %d rays %d spills %d | canary: badLanes %d observedThread %d ownThread %d badWaves %d | lateWaves %d waves %d'
              % (rep, len(bad), lost, false_, other, c['hits'], c['nodeVisits'], c['rays'], c['stackSpills'], c['cyclesFetch'], c['cyclesNode'], c['cyclesLeaf'], c['cyclesPop'], c['reserved'], c['waves']), flush=True)
        if len(bad):
            q = bad // perq
            pos = (bad - q * perq) / perq
            print('     queues', sorted(set(q.tolist())), 'position in queue min %.3f max %.3f' % (pos.min(), pos.max()), 'ray range', bad.min(), bad.max(),
                  'chunks', sorted(set(((bad - q * perq) // 256).tolist()))[:12], flush=True)
    sc.release(); dev.release()
