"""Row f2 of SURVEY.md section 8: rtcIntersect1 / rtcOccluded1 / short streams issued concurrently from many host
threads (the way embree harness threads call, section 8b "Threading") are combined into shared launches and still
return exactly what independent calls return."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_single_ray_calls_are_combined_and_exact(rtc, po, bomberman):
    verts, fs, fi = bomberman
    dev = rtc.Device("tri_accel=bvh8.triangle4v")
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    sc.commit()
    lo, hi = verts.min(0), verts.max(0)
    T, per = 32, 64
    src = po.make_random_rays(T * per, lo, hi, seed=77)
    want = rtc.aligned_rayhits(T * per)
    want[:] = src
    sc.intersect1M(want)  # one big batch = the reference answer for M independent calls
    got = rtc.aligned_rayhits(T * per)
    got[:] = src
    occ = rtc.aligned_rays(T * per)
    for f in occ.dtype.names:
        occ[f] = src[f]
    wocc = occ.copy()
    sc.occluded1M(wocc)
    launches0 = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES)
    calls0 = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS)
    errors = []

    def worker(t):
        try:
            for i in range(t * per, (t + 1) * per):
                if i % 8 == 7:  # a short stream now and then
                    continue
                sc.intersect1(got[i:i + 1])
                sc.occluded1(occ[i:i + 1])
            idx = np.arange(t * per + 7, (t + 1) * per, 8)
            for i in idx:
                sc.intersect1M(got[i:i + 1])
                sc.occluded1M(occ[i:i + 1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    assert got.tobytes() == want.tobytes()
    assert occ.tobytes() == wocc.tobytes()
    calls = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS) - calls0
    launches = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES) - launches0
    assert calls == 2 * T * per
    # the point of the combiner: fewer launches than calls.  Python threads serialise on the GIL between calls, so only
    # a few calls are ever pending together here (C harness threads queue up far deeper); intersect and occluded
    # records of one combined batch are two launches.  (How many calls meet depends on how long a launch takes: since small
    # batches are traced in place in pinned host memory a call is over sooner and fewer calls queue up behind it - 0.80 of the
    # calls were launches on one run, 0.65-0.75 before.  The C harness, tests/test_c_example.py, checks the depth of the combining.)
    assert launches < 0.95 * calls, (launches, calls)
    assert (got["geomID"] != 0xFFFFFFFF).sum() > 0
    sc.release()
    dev.release()


def test_held_leader_combines_all_pending_calls_into_one_launch(rtc, bomberman):
    """Deterministic check of the combiner: the leader is held (rtcamdDebugHoldCombiner) until T single-ray calls from T
    threads are pending, then released: the T calls must be traced as ONE batch - one kernel launch - and every caller gets its
    own result, identical to one rtcIntersect1M over the same rays."""
    import threading
    import time
    verts, fs, fi = bomberman
    dev = rtc.Device("tri_accel=bvh8.triangle4v")
    sc = rtc.Scene(dev)
    sc.add_triangles(verts, rtc.fan_triangulate(fs, fi))
    sc.commit()
    T = 12
    rg = __import__("importlib").import_module("embree-compressed_amd.raygen")
    src = rg.make_random_rays(T, verts.min(0), verts.max(0), seed=5).reshape(-1).view(rtc.RAYHIT_DTYPE)
    want = rtc.aligned_rayhits(T)
    want[:] = src
    sc.intersect1M(want)
    got = rtc.aligned_rayhits(T)
    got[:] = src
    calls0 = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS)
    launches0 = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES)
    lib = rtc.lib()
    lib.rtcamdDebugHoldCombiner(dev.handle, 1)
    errors = []

    def worker(i):
        try:
            sc.intersect1(got[i:i + 1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    for th in threads:
        th.start()
    deadline = time.time() + 30
    while dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS) - calls0 < T and time.time() < deadline:
        time.sleep(0.001)
    queued = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_COMBINED_CALLS) - calls0
    lib.rtcamdDebugHoldCombiner(dev.handle, 0)
    for th in threads:
        th.join()
    assert not errors, errors[0]
    assert queued == T
    launches = dev.get_property(rtc.RTCAMD_DEVICE_PROPERTY_TRACE_LAUNCHES) - launches0
    assert launches == 1, (launches, T)  # a call is counted once it is pending: all T were, when the leader was released
    assert got.tobytes() == want.tobytes()
    sc.release()
    dev.release()


def test_error_of_a_combined_call_reaches_its_caller(rtc):
    dev = rtc.Device("")
    sc = rtc.Scene(dev)
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    sc.add_triangles(v, np.array([[0, 1, 2]], np.uint32))
    # not committed: rtcIntersect1 must raise INVALID_OPERATION like scene.cpp:25,54
    rh = rtc.aligned_rayhits(1)
    with pytest.raises(rtc.RTCError) as e:
        sc.intersect1(rh)
    assert e.value.code == 3
    sc.commit()
    sc.intersect1(rh)
    sc.release()
    dev.release()
