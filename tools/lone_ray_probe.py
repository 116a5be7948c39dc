"""Development probe: cost of one loop iteration for a wave that traces a single deep ray (64 copies of it, so that the wave
is alone on its SIMD and all lanes follow the same path): the critical path that bounds a batch's drain.  Uses the phase
stamps when the library is built with EXTRA=-DTRACE_PHASE_STAMPS=1 (RTAMD_LIB=...)."""
import importlib, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
d = np.load('/root/repo/assets/bomberman.mesh.npz'); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
dev = rtc.Device('gpu=0,subdiv_accel=bvh4.compressed.leaf'); sc = rtc.Scene(dev)
sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
rays = raygen.make_random_rays(200000, v.min(0), v.max(0), seed=1)
buf = torch.from_numpy(rays).cuda(); sc.intersect1M(buf); dev.synchronize()
hit = (buf.view(torch.int32)[:, 18] != -1).cpu().numpy()
cand = np.nonzero(hit)[0][:400]
best, bestIt = None, 0
for i in cand:
    b = torch.from_numpy(np.repeat(rays[i:i + 1], 64, 0)).cuda()
    c = sc.intersect1M_counted(b)
    if c['iterations'] > bestIt: bestIt, best = c['iterations'], i
print('deepest of %d hit rays: ray %d, %d iterations (nodes %d, leaf visits %d, inner %d per ray)' % (len(cand), best, bestIt, c['nodeVisits'] // 64, c['primTests'] // 64, c['innerVisits'] // 64))
b = torch.from_numpy(np.repeat(rays[best:best + 1], 64, 0)).cuda()
c = sc.intersect1M_counted(b)
tot = max(c['cyclesTotal'], 1)
print('one wave, 64 copies: %d iterations, %d leaf phases, %.0f cycles total = %.0f cycles per iteration; nodes/ray %d, blobs/ray %d, inner/ray %d' % (
    c['iterations'], c['leafPhases'], tot, tot / c['iterations'], c['nodeVisits'] // 64, c['primTests'] // 64, c['innerVisits'] // 64))
if c['cyclesNode']:
    print('phase cycles per iteration: fetch %.0f node %.0f leaf(per leaf phase) %.0f pop %.0f' % (c['cyclesFetch'] / c['iterations'], c['cyclesNode'] / c['iterations'], c['cyclesLeaf'] / max(c['leafPhases'], 1), c['cyclesPop'] / c['iterations']))
# the same ray alone in its wave (1 copy) and as 8 copies: the drain's case - child-parallel node steps, quad-form blob walk
for copies in (1, 8):
    b = torch.from_numpy(np.repeat(rays[best:best + 1], copies, 0)).cuda()
    c = sc.intersect1M_counted(b)
    tot = max(c['cyclesTotal'], 1)
    print('%d cop%s: %d iterations, %d leaf phases, %.0f cycles total = %.0f cycles per iteration' % (copies, 'y' if copies == 1 else 'ies', c['iterations'], c['leafPhases'], tot, tot / c['iterations']))
    if c['cyclesNode']:
        print('   phase cycles per iteration: fetch %.0f node %.0f leaf(per leaf phase) %.0f pop %.0f' % (c['cyclesFetch'] / c['iterations'], c['cyclesNode'] / c['iterations'], c['cyclesLeaf'] / max(c['leafPhases'], 1), c['cyclesPop'] / c['iterations']))
# wall time of the single-wave kernel
for _ in range(3): sc.intersect1M(torch.from_numpy(np.repeat(rays[best:best + 1], 64, 0)).cuda())
dev.synchronize()
bs = [torch.from_numpy(np.repeat(rays[best:best + 1], 64, 0)).cuda() for _ in range(20)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for x in bs: sc.intersect1M(x, check=False)
dev.synchronize(); print('wall per 64-ray launch: %.1f us' % ((time.perf_counter() - t0) / 20 * 1e6))
