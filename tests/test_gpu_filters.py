"""Filter callbacks (intersector_epilog.h:251-291, filter.h:27-130) through the C-ABI:
geometry intersection / occlusion filters and the context filter on triangle geometry.  The device finds candidates,
the host runs the callbacks, rejected candidates are excluded and the ray is traced again.  Checked against closed forms:
a stack of parallel unit quads pierced by rays along +z."""
import ctypes as C

import numpy as np
import pytest

from helpers import INVALID, fill_rays

pytestmark = pytest.mark.gpu

NQ = 5  # quads at z = 0, 1, 2, 3, 4 (geomID = index), two triangles each


def _stack(rtc, robust=True):
    dev = rtc.Device("tri_accel=bvh8.triangle4v" if robust else "")
    sc = rtc.Scene(dev)
    for z in range(NQ):
        v = np.array([[0, 0, z], [1, 0, z], [1, 1, z], [0, 1, z]], np.float32)
        assert sc.add_triangles(v, np.array([[0, 1, 2], [0, 2, 3]], np.uint32)) == z
    return dev, sc


def _rays(rtc, n, seed=1):
    rng = np.random.RandomState(seed)
    rh = rtc.aligned_rayhits(n)
    org = np.stack([rng.rand(n) * 0.9 + 0.05, rng.rand(n) * 0.9 + 0.05, -np.ones(n)], 1).astype(np.float32)
    fill_rays(rh, org, np.tile(np.array([0, 0, 1], np.float32), (n, 1)))
    return rh


def _ray_fields(args):
    ray = C.cast(args.contents.ray, C.POINTER(C.c_float * 12)).contents
    hit = C.cast(args.contents.hit, C.POINTER(C.c_uint * 8)).contents
    return ray, hit


@pytest.mark.parametrize("robust", [True, False])
def test_intersection_filter_rejects_until_an_accepted_candidate(rtc, robust):
    dev, sc = _stack(rtc, robust)
    calls = []

    @rtc.FILTER_FUNC
    def flt(args):  # "alpha test": quad g rejects rays whose x < 0.2 * (g + 1)
        ray, hit = _ray_fields(args)
        assert args.contents.N == 1 and args.contents.valid[0] == -1
        g = hit[6]
        calls.append((ray[0], g, ray[8]))
        assert abs(ray[8] - (g + 1.0)) < 1e-5  # tfar = candidate distance while the callback runs
        if ray[0] < 0.2 * (g + 1):
            args.contents.valid[0] = 0

    for g in range(NQ - 1):  # the last quad has no filter
        sc.set_filters(g, intersect=flt)
    sc.commit()
    assert dev.get_property(66) == 1  # RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED
    n = 4000
    rh = _rays(rtc, n)
    x = rh["org_x"].copy()
    sc.intersect1M(rh)
    # the accepted quad: the first g with x >= 0.2 (g+1), else the unfiltered last one
    want = np.array([next(g for g in range(NQ) if g == NQ - 1 or xi >= np.float32(0.2 * (g + 1))) for xi in x])
    assert np.array_equal(rh["geomID"], want.astype(np.uint32))
    assert np.allclose(rh["tfar"], want + 1.0, atol=1e-5)
    assert ((rh["primID"] == 0) | (rh["primID"] == 1)).all()
    # every candidate in front of the accepted one was offered exactly once, nearest first
    per_ray = {}
    for xo, g, t in calls:
        per_ray.setdefault(xo, []).append(g)
    for xi, w in zip(x, want):
        seq = per_ray[xi]
        assert seq == list(range(min(w, NQ - 2) + 1)), (xi, w, seq)
    sc.release()
    dev.release()


def test_always_rejecting_filter_collects_all_hits_and_leaves_the_ray_untouched(rtc):
    dev, sc = _stack(rtc)
    seen = []

    @rtc.FILTER_FUNC
    def collect(args):
        ray, hit = _ray_fields(args)
        seen.append((hit[6], hit[5], ray[8]))
        args.contents.valid[0] = 0

    for g in range(NQ):
        sc.set_filters(g, intersect=collect, occluded=collect)
    sc.commit()
    rh = _rays(rtc, 64, seed=3)
    before = rh.copy()
    sc.intersect1M(rh)
    assert rh.tobytes() == before.tobytes()           # all candidates rejected: a miss leaves the record untouched
    assert len(seen) == 64 * NQ                       # the "collect all hits" idiom sees every pierced triangle once
    occ = rtc.aligned_rays(64)
    for f in occ.dtype.names:
        occ[f] = before[f]
    seen.clear()
    sc.occluded1M(occ)
    assert np.isinf(occ["tfar"]).all() and (occ["tfar"] > 0).all() and len(seen) == 64 * NQ
    sc.release()
    dev.release()


def test_occlusion_and_context_filters(rtc):
    dev, sc = _stack(rtc)

    @rtc.FILTER_FUNC
    def only_far(args):  # geometry occlusion filter: quads 0..2 are transparent for shadow rays
        _, hit = _ray_fields(args)
        if hit[6] < 3:
            args.contents.valid[0] = 0

    for g in range(NQ):
        sc.set_filters(g, occluded=only_far)
    sc.commit()
    n = 512
    src = _rays(rtc, n, seed=5)
    occ = rtc.aligned_rays(n)
    for f in occ.dtype.names:
        occ[f] = src[f]
    occ["tfar"][: n // 2] = 3.5  # these rays end between quad 2 (t=3) and quad 3 (t=4): nothing opaque in range
    sc.occluded1M(occ)
    assert (occ["tfar"][: n // 2] == np.float32(3.5)).all()
    assert np.isneginf(occ["tfar"][n // 2:]).all()
    # context filter: reject primID 1 everywhere -> rays through the second triangle of every quad miss
    @rtc.FILTER_FUNC
    def no_prim1(args):
        _, hit = _ray_fields(args)
        if hit[5] == 1:
            args.contents.valid[0] = 0

    ctx = rtc.make_context()
    ctx.filter = C.cast(no_prim1, C.c_void_p)
    rh = _rays(rtc, n, seed=6)
    upper = rh["org_y"] > rh["org_x"]  # triangle (0,2,3) = primID 1 covers y > x
    sc.intersect1M(rh, ctx=ctx)
    assert (rh["geomID"][upper] == INVALID).all()
    assert (rh["geomID"][~upper] == 0).all() and (rh["primID"][~upper] == 0).all()
    sc.release()
    dev.release()
