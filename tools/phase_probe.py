"""Development probe: per-phase shader-clock breakdown of one counted batch (rtcamdIntersect1MCounted)."""
import importlib, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
for name, cfg, sub in (('cbvh.leaf', 'subdiv_accel=bvh4.compressed.leaf', True), ('eager', 'subdiv_accel=default', True), ('tri', 'tri_accel=bvh8.triangle4v', False)):
    dev = rtc.Device('gpu=0,' + cfg); sc = rtc.Scene(dev)
    if sub: sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3)
    else: sc.add_triangles(v, rtc.fan_triangulate(fs, fi))
    sc.commit()
    for kind in ('random', 'primary'):
        rays = raygen.make_random_rays(1000000, lo, hi, seed=1) if kind == 'random' else raygen.make_primary_rays()
        buf = torch.from_numpy(rays).cuda()
        sc.intersect1M_counted(torch.from_numpy(rays).cuda())
        c = sc.intersect1M_counted(buf)
        tot = max(c['cyclesTotal'], 1)
        print('%-9s %-7s waves %d iters/wave %.0f leafPhases/wave %.0f cycles/wave %.0f | fetch %.2f node %.2f leaf %.2f pop %.2f | rays/iter %.1f' % (
            name, kind, c['waves'], c['iterations'] / c['waves'], c['leafPhases'] / c['waves'], tot / c['waves'],
            c['cyclesFetch'] / tot, c['cyclesNode'] / tot, c['cyclesLeaf'] / tot, c['cyclesPop'] / tot, c['rays'] / max(c['iterations'], 1)))
    sc.release(); dev.release()
