"""Development probe: per-phase shader-clock breakdown of one counted batch (rtcamdIntersect1MCounted)."""
import importlib, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
only = sys.argv[1:]
for name, cfg, sub in (('cbvh.leaf', 'subdiv_accel=bvh4.compressed.leaf', True), ('eager', 'subdiv_accel=default', True), ('tri', 'tri_accel=bvh8.triangle4v', False)):
    if only and name not in only: continue
    dev = rtc.Device('gpu=0,' + cfg); sc = rtc.Scene(dev)
    if sub: sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3)
    else: sc.add_triangles(v, rtc.fan_triangulate(fs, fi))
    sc.commit()
    for kind in ('random', 'primary'):
        rays = raygen.make_random_rays(1000000, lo, hi, seed=1) if kind == 'random' else raygen.make_primary_rays()
        buf = torch.from_numpy(rays).cuda()
        sc.intersect1M_counted(torch.from_numpy(rays).cuda())
        import time
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = sc.intersect1M_counted(buf)
        wall = (time.perf_counter() - t0) * 1e6
        fresh = [torch.from_numpy(rays).cuda() for _ in range(5)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in fresh: sc.intersect1M(b)
        dev.synchronize() if hasattr(dev, 'synchronize') else torch.cuda.synchronize()
        plain = (time.perf_counter() - t0) * 1e6 / 5
        print('counted call wall %.0f us, plain call wall %.0f us' % (wall, plain))
        tot = max(c['cyclesTotal'], 1)
        print('%-9s %-7s waves %d iters/wave %.0f leafPhases/wave %.0f cycles/wave %.0f | fetch %.2f node %.2f leaf %.2f pop %.2f | rays/iter %.1f' % (
            name, kind, c['waves'], c['iterations'] / c['waves'], c['leafPhases'] / c['waves'], tot / c['waves'],
            c['cyclesFetch'] / tot, c['cyclesNode'] / tot, c['cyclesLeaf'] / tot, c['cyclesPop'] / tot, c['rays'] / max(c['iterations'], 1)))
        print('   loop occupancy %.3f; wave end times (4 us buckets): %s' % (c['activeLaneIters'] / (64.0 * c['iterations']), ' '.join(str(x) for x in c['waveEndHist'])))
        print('   stack spills %d;' % c['stackSpills'], end=' ')
        print('longest ray %d iterations; drain (last grab -> wave end): mean %.1f us, max %.1f us' % (c['maxRaySteps'], c['drainTicksSum'] / c['waves'] / 100.0, c['drainTicksMax'] / 100.0))
        print('   iterations per wave (buckets of 2): %s' % ' '.join(str(x) for x in c['waveIterHist']))
    sc.release(); dev.release()
