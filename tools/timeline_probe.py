"""Development probe (GPU box, -DTRACE_TIMELINE build of the library, RTAMD_TIMELINE=1): wave timeline of the PLAIN metric kernel.
    RTAMD_TIMELINE=1 RTAMD_LIB=.../lib_wtl/libembree3.so python tools/timeline_probe.py [rays]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
rtc = importlib.import_module('embree-compressed_amd').rtc
raygen = importlib.import_module('embree-compressed_amd.raygen')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = np.load(os.path.join(ROOT, 'assets', 'bomberman.mesh.npz')); v, fs, fi = d['verts'], d['face_sizes'], d['face_index']
lo, hi = v.min(0), v.max(0)
dev = rtc.Device('gpu=0,subdiv_accel=bvh4.compressed.leaf'); sc = rtc.Scene(dev)
sc.add_subdiv(v, fs, fi); sc.set_levels(6, 3); sc.commit()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); dev.set_stream(stream.cuda_stream)
bufs = [torch.from_numpy(np.ascontiguousarray(raygen.make_random_rays(n, lo, hi, seed=100 + k))).to('cuda') for k in range(6)]
torch.cuda.synchronize()
for k in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); sc.intersect1M(bufs[k], check=False); e1.record(stream)
    torch.cuda.synchronize()
    if k < 3:
        continue
    log = np.zeros(16384 * 8, np.uint64)
    got = rtc.lib().rtcamdDebugReadWaveLog(dev.handle, log.ctypes.data, log.nbytes)
    w = log.reshape(-1, 8)
    w = w[w[:, 3] != 0]
    t0 = w[:, 0].min()
    us = lambda a: (a.astype(np.int64) - int(t0)) / 100.0
    begin, first, grab, end = us(w[:, 0]), us(w[:, 1]), us(w[:, 2]), us(w[:, 3])
    it, dit, dl, rays, adopt = w[:, 4].astype(np.int64), w[:, 5].astype(np.int64), w[:, 6].astype(np.int64), (w[:, 7] & np.uint64(0xFFFFFFFF)).astype(np.int64), (w[:, 7] >> np.uint64(32)).astype(np.int64)
    print('launch %d: %.1f us by events; %d waves; rays handed out %d' % (k, e0.elapsed_time(e1) * 1e3, len(w), rays.sum()))
    print('  wave start   : min %.1f  median %.1f  max %.1f us' % (begin.min(), np.median(begin), begin.max()))
    print('  last grab    : min %.1f  median %.1f  max %.1f us (waves with rays: %d)' % (grab[rays > 0].min(), np.median(grab[rays > 0]), grab[rays > 0].max(), (rays > 0).sum()))
    print('  wave end     : min %.1f  p10 %.1f median %.1f  p90 %.1f  p99 %.1f  max %.1f us' % (end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
    print('  iterations   : mean %.1f  p90 %d  max %d; after exhaustion: mean %.1f max %d; lanes per drain iteration %.2f' % (it.mean(), np.percentile(it, 90), it.max(), dit.mean(), dit.max(), dl.sum() / max(1, dit.sum())))
    busy = (grab - first)[rays > 0]; drain = (end - grab)[rays > 0]
    print('  per wave     : first rays -> last grab %.1f us, last grab -> end %.1f us (max %.1f); us per iteration overall %.2f, in the drain %.2f' % (
        busy.mean(), drain.mean(), drain.max(), ((end - first)[rays > 0] / np.maximum(it[rays > 0], 1)).mean(), (drain / np.maximum(dit[rays > 0], 1)).mean()))
    print('  adoptions    : %d rays adopted by %d waves (max %d per wave)' % (adopt.sum(), (adopt > 0).sum(), adopt.max()))
    hist, edges = np.histogram(end, bins=np.arange(0, end.max() + 10, 10))
    alive = len(w) - np.cumsum(hist)
    print('  waves alive at t (10 us steps): ' + ' '.join('%d' % a for a in np.concatenate([[len(w)], alive])))
