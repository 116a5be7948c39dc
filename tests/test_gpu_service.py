"""The persistent consumer for small calls (config key service=1; trace_service.hip.h, rt_trace.cpp service_trace):
a resident kernel answers calls of up to 64 rays through a ring of slots in host-mapped memory - no kernel launch per call.  It runs the same
traversal code as every batch (trace_body), so single-ray calls, short streams and any-hit queries must be byte-identical to ONE big
rtcIntersect1M / rtcOccluded1M over the same rays; calls from several threads use different slots; a scene change (other accel arrays, same kind)
is picked up per job; the kernel leaves by itself when idle and comes back on demand."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INVALID = 0xFFFFFFFF


def _scene(rtc, cfg, bomberman, kind):
    verts, fs, fi = bomberman
    dev = rtc.Device(cfg)
    sc = rtc.Scene(dev)
    if kind == "tri":
        sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    else:
        sc.add_subdiv(verts, fs, fi)
        sc.set_levels(4, 3)
    sc.commit()
    return dev, sc


@pytest.mark.parametrize("cfg,kind", [("tri_accel=bvh8.triangle4v", "tri"), ("tri_accel=bvh8.triangle4", "tri"), ("subdiv_accel=default", "sub"),
                                      ("subdiv_accel=bvh4.compressed.leaf", "sub"), ("subdiv_accel=bvh4.compressed.grid", "sub")])
def test_service_answers_small_calls_like_one_big_batch(rtc, po, bomberman, cfg, kind):
    verts, fs, fi = bomberman
    dev0, sc0 = _scene(rtc, cfg, bomberman, kind)             # launches
    dev, sc = _scene(rtc, "service=1," + cfg, bomberman, kind)  # resident kernel
    n = 4096
    src = po.make_random_rays(n, verts.min(0), verts.max(0), seed=21)
    want = rtc.aligned_rayhits(n)
    want[:] = src
    sc0.intersect1M(want)
    wocc = rtc.aligned_rays(n)
    for f in wocc.dtype.names:
        wocc[f] = src[f]
    occ = wocc.copy()
    sc0.occluded1M(wocc)
    got = rtc.aligned_rayhits(n)
    got[:] = src
    T, errors = 4, []

    def worker(t):
        try:
            lo, hi = t * n // T, (t + 1) * n // T
            i = lo
            while i < hi:  # single rays, short streams of 2..64 rays, any-hit in between
                m = (1, 1, 1, 7, 64, 33, 1, 2)[(i // 3) % 8]
                m = min(m, hi - i)
                if m == 1:
                    sc.intersect1(got[i:i + 1])
                    sc.occluded1(occ[i:i + 1])
                else:
                    sc.intersect1M(got[i:i + m])
                    sc.occluded1M(occ[i:i + m])
                i += m
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    served = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_SERVICE_CALLS)
    assert served > n // 8, served  # the calls went through the resident kernel, not through launches
    assert got.tobytes() == want.tobytes()
    assert occ.tobytes() == wocc.tobytes()
    assert (got["geomID"] != INVALID).sum() > 100
    # idle exit and restart: after a pause longer than the kernel's idle limit the next call is still answered (by a fresh kernel)
    time.sleep(0.15)
    again = rtc.aligned_rayhits(64)
    again[:] = src[:64]
    sc.intersect1M(again)
    assert again.tobytes() == want[:64].tobytes()
    for s_, d_ in ((sc, dev), (sc0, dev0)):
        s_.release()
        d_.release()


def test_service_follows_a_recommitted_scene(rtc, po, bomberman):
    """Jobs carry the scene's device arrays: after a second scene (same accel kind, other geometry) is built on the same device, calls on either
    scene are answered with that scene's geometry."""
    verts, fs, fi = bomberman
    tris = rtc.fan_triangulate(fs, fi)
    dev = rtc.Device("service=1,tri_accel=bvh8.triangle4v")
    a = rtc.Scene(dev)
    a.add_triangles(verts, tris)
    a.commit()
    b = rtc.Scene(dev)
    b.add_triangles(verts + np.array([0, 30, 0], np.float32), tris)  # the same mesh lifted by 30
    b.commit()
    src = po.make_random_rays(512, verts.min(0), verts.max(0) + np.array([0, 30, 0], np.float32), seed=5)
    wa, wb = rtc.aligned_rayhits(512), rtc.aligned_rayhits(512)
    wa[:] = src
    wb[:] = src
    ga, gb = wa.copy(), wb.copy()
    dev2 = rtc.Device("tri_accel=bvh8.triangle4v")
    for sc_src, w in ((verts, wa), (verts + np.array([0, 30, 0], np.float32), wb)):
        s2 = rtc.Scene(dev2)
        s2.add_triangles(sc_src, tris)
        s2.commit()
        s2.intersect1M(w)
        s2.release()
    for i in range(512):
        a.intersect1(ga[i:i + 1])
        b.intersect1(gb[i:i + 1])
    assert ga.tobytes() == wa.tobytes() and gb.tobytes() == wb.tobytes()
    assert not np.array_equal(wa["geomID"], wb["geomID"]) or not np.array_equal(wa["tfar"], wb["tfar"])
    assert dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_SERVICE_CALLS) >= 1024
    a.release()
    b.release()
    dev.release()
    dev2.release()
