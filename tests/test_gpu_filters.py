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


# ---- subdivision geometry (round 2) ------------------------------------------------------------------------------------
# Eager grid cells: GridSOAIntersector1 hands every triangle of a patch to Intersect1EpilogMU / Occluded1EpilogMU with the
# patch's geomID / primID (grid_soa_intersector1.h:61,83; intersector_epilog.h:460-600), which offers the candidates one by one.
# The fork's compressed modes never call a filter (compressed.h:454-756).

def _subdiv_stack(rtc, accel, level=3, with_tri_floor=False, bump=0.0):
    """NQ flat unit quads at z = 0..NQ-1 as one-face subdivision meshes with pinned corners (limit surface = the square),
    2^level x 2^level quads = 2 * 4^level triangles each, all reporting primID 0 of geometry z.  bump: lifts three corners (a
    saddle up to `bump` high) so that the fork's height-field leaves have a thickness."""
    dev = rtc.Device("subdiv_accel=" + accel)
    sc = rtc.Scene(dev)
    for z in range(NQ):
        v = np.array([[0, 0, z], [1, 0, z + 0.5 * bump], [1, 1, z + bump], [0, 1, z + 0.5 * bump]], np.float32)
        g = sc.add_subdiv(v, np.array([4], np.uint32), np.arange(4, dtype=np.uint32), vertex_creases=(np.arange(4, dtype=np.uint32), np.full(4, np.inf, np.float32)))
        assert g == z
    if with_tri_floor:
        v = np.array([[0, 0, NQ], [1, 0, NQ], [1, 1, NQ], [0, 1, NQ]], np.float32)
        assert sc.add_triangles(v, np.array([[0, 1, 2], [0, 2, 3]], np.uint32)) == NQ
    sc.set_levels(level, 2)
    return dev, sc


def test_subdiv_eager_intersection_filter_offers_the_triangles_of_a_patch_one_by_one(rtc):
    dev, sc = _subdiv_stack(rtc, "default")
    calls = []

    @rtc.FILTER_FUNC
    def flt(args):  # "alpha test": quad g rejects rays whose x < 0.2 * (g + 1)
        ray, hit = _ray_fields(args)
        assert args.contents.N == 1 and args.contents.valid[0] == -1
        g = hit[6]
        calls.append((ray[0], g, ray[8]))
        assert hit[5] == 0 and abs(ray[8] - (g + 1.0)) < 1e-5  # patch primID; tfar = candidate distance while the callback runs
        if ray[0] < 0.2 * (g + 1):
            args.contents.valid[0] = 0

    for g in range(NQ - 1):  # the last quad has no filter
        sc.set_filters(g, intersect=flt)
    sc.commit()
    n = 3000
    rh = _rays(rtc, n)
    x, y = rh["org_x"].copy(), rh["org_y"].copy()
    sc.intersect1M(rh)
    want = np.array([next(g for g in range(NQ) if g == NQ - 1 or xi >= np.float32(0.2 * (g + 1))) for xi in x])
    assert np.array_equal(rh["geomID"], want.astype(np.uint32)) and (rh["primID"] == 0).all()
    assert np.allclose(rh["tfar"], want + 1.0, atol=1e-5)
    assert np.allclose(rh["u"], x, atol=2e-3) and np.allclose(rh["v"], y, atol=2e-3)  # patch uv of the unit square (16-bit vertex uvs)
    per_ray = {}
    for xo, g, t in calls:
        per_ray.setdefault(xo, []).append(g)
    for xi, w in zip(x, want):  # every patch in front of the accepted one was offered exactly once, nearest first
        assert per_ray[xi] == list(range(min(w, NQ - 2) + 1)), (xi, w, per_ray[xi])
    sc.release()
    dev.release()


def test_subdiv_eager_reject_all_and_occlusion_filters(rtc):
    dev, sc = _subdiv_stack(rtc, "default", with_tri_floor=True)
    seen = []

    @rtc.FILTER_FUNC
    def collect(args):
        ray, hit = _ray_fields(args)
        seen.append((hit[6], ray[8]))
        args.contents.valid[0] = 0

    @rtc.FILTER_FUNC
    def transparent_below_3(args):  # occlusion filter: patches 0..2 let shadow rays through
        _, hit = _ray_fields(args)
        if hit[6] < 3:
            args.contents.valid[0] = 0

    for g in range(NQ + 1):
        sc.set_filters(g, intersect=collect, occluded=transparent_below_3)
    sc.commit()
    n = 64
    rh = _rays(rtc, n, seed=3)
    before = rh.copy()
    sc.intersect1M(rh)
    assert rh.tobytes() == before.tobytes()  # every candidate rejected, patches and the triangle floor alike: untouched
    assert len(seen) == n * (NQ + 1)         # one triangle of each patch + the floor per ray
    assert sorted(set(g for g, _ in seen)) == list(range(NQ + 1))
    occ = rtc.aligned_rays(n)
    for f in occ.dtype.names:
        occ[f] = before[f]
    occ["tfar"][: n // 2] = 3.5  # ends between patch 2 (t = 3) and patch 3 (t = 4)
    sc.occluded1M(occ)
    assert (occ["tfar"][: n // 2] == np.float32(3.5)).all() and np.isneginf(occ["tfar"][n // 2:]).all()
    sc.release()
    dev.release()


def test_subdiv_eager_distance_filter_equals_tnear(rtc, po, bomberman):
    """Size-independent property on the real scene: a context filter that rejects every candidate with t <= t0 must leave exactly
    the hit a ray with tnear = t0 finds (the triangle test's t does not depend on tnear / tfar; its range test is
    absDen * tnear < T, i.e. t > tnear), on every field: the rejected front layers of bomberman are excluded one candidate
    at a time, several triangles of the same patch included."""
    verts, fs, fi = bomberman
    dev = rtc.Device("subdiv_accel=default")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, fs, fi)
    sc.set_levels(4, 2)
    sc.commit()
    n = 6000
    lo, hi = verts.min(0), verts.max(0)
    src = po.make_random_rays(n, lo, hi, seed=5)
    plain = src.copy()
    sc.intersect1M(plain)
    hit = plain["geomID"] != INVALID
    assert hit.sum() > 500
    t0 = np.where(hit, plain["tfar"] * np.float32(1.0001), np.float32(0)).astype(np.float32)  # just behind the first hit
    idx_of = {}
    for i in range(n):
        idx_of[(src["org_x"][i], src["org_y"][i], src["dir_x"][i])] = i
    ncalls = [0]

    @rtc.FILTER_FUNC
    def behind_t0(args):
        ray, _ = _ray_fields(args)
        ncalls[0] += 1
        i = idx_of[(np.float32(ray[0]), np.float32(ray[1]), np.float32(ray[4]))]
        if not ray[8] > t0[i]:
            args.contents.valid[0] = 0

    ctx = rtc.make_context()
    ctx.filter = C.cast(behind_t0, C.c_void_p)
    got = src.copy()
    sc.intersect1M(got, ctx=ctx)
    want = src.copy()
    want["tnear"] = t0
    sc.intersect1M(want)
    want["tnear"] = src["tnear"]
    assert got.tobytes() == want.tobytes()
    second = want["geomID"] != INVALID
    assert 100 < (second & hit).sum() and ncalls[0] >= hit.sum() + (second & hit).sum()
    sc.release()
    dev.release()


@pytest.mark.parametrize("accel", ["bvh4.compressed.leaf", "bvh4.compressed.grid"])
def test_fork_accels_never_call_filters(rtc, accel):
    """CompressedBVHIntersector1 writes its hits itself and its occluded() is a stub (compressed.h:454-756): geometry and context
    filters are not called for subdivision geometry on these accels; filters of triangle geometry in the same scene still are."""
    dev, sc = _subdiv_stack(rtc, accel, with_tri_floor=True, bump=0.1)
    calls = []

    @rtc.FILTER_FUNC
    def reject(args):
        _, hit = _ray_fields(args)
        calls.append(hit[6])
        args.contents.valid[0] = 0

    for g in range(NQ + 1):
        sc.set_filters(g, intersect=reject, occluded=reject)
    sc.commit()
    n = 200
    rh = _rays(rtc, n, seed=9)
    sc.intersect1M(rh)
    assert (rh["geomID"] == 0).all() and np.all(np.abs(rh["tfar"] - 1.05) < 0.08) and calls == []  # first patch, no callback
    ctx = rtc.make_context()
    ctx.filter = C.cast(reject, C.c_void_p)
    rh2 = _rays(rtc, n, seed=9)
    sc.intersect1M(rh2, ctx=ctx)
    assert rh2.tobytes() == rh.tobytes() and calls == []
    # rays that start behind the patches only meet the triangle floor: its filter runs and rejects
    far = _rays(rtc, n, seed=9)
    far["org_z"] = NQ - 0.5
    before = far.copy()
    sc.intersect1M(far)
    assert far.tobytes() == before.tobytes() and calls == [NQ] * n
    occ = rtc.aligned_rays(n)
    for f in occ.dtype.names:
        occ[f] = before[f]
    calls.clear()
    sc.occluded1M(occ)
    assert (occ["tfar"] == before["tfar"]).all() and calls == [NQ] * n  # floor transparent, no blob reached
    occ2 = rtc.aligned_rays(n)
    src = _rays(rtc, n, seed=9)
    for f in occ2.dtype.names:
        occ2[f] = src[f]
    calls.clear()
    sc.occluded1M(occ2)
    assert np.isneginf(occ2["tfar"]).all() and calls == []  # the stub: a blob reached = occluded, unfiltered
    sc.release()
    dev.release()
