"""GPU parity tests of the triangle hot path (rows a1-a10, a18 of SURVEY.md section 8), through the C-ABI.

Bar: geomID / primID / instID bit-exact against the oracle, t/u/v within 1e-4 relative (north_star).
"""
import numpy as np
import pytest

from helpers import INVALID, compare_hits, fill_rays, random_rays_np, random_soup, unit_triangle_rays

pytestmark = pytest.mark.gpu

ULP = np.float32(1.1920929e-7)
MODES = [(0, "tri_accel=bvh8.triangle4v"), (1, "tri_accel=bvh8.triangle4")]


def _scene(rtc, cfg, verts, tris, flags=0):
    dev = rtc.Device(cfg)
    sc = rtc.Scene(dev, flags)
    sc.add_triangles(verts, tris)
    sc.commit()
    return dev, sc


@pytest.mark.parametrize("mode,cfg", MODES)
def test_triangle_hit_kat(rtc, mode, cfg):
    """TriangleHitTest (verify.cpp:2118-2205): closed-form answers, intersect and occluded, single and stream."""
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tris = np.array([[0, 1, 2]], np.uint32)
    dev, sc = _scene(rtc, cfg, verts, tris)
    frm, dirs, u, v = unit_triangle_rays(rtc, 256)
    rays = rtc.aligned_rayhits(256)
    fill_rays(rays, np.broadcast_to(frm, (256, 3)), dirs)
    sc.intersect1M(rays)
    tol = 16 * ULP
    assert np.all(rays["geomID"] == 0) and np.all(rays["primID"] == 0)
    assert np.all(np.abs(rays["u"] - u) <= tol) and np.all(np.abs(rays["v"] - v) <= tol)
    assert np.all(np.abs(rays["tfar"] - 1.0) <= tol)
    ht = frm[None, :] + rays["tfar"][:, None] * dirs
    huv = rays["u"][:, None] * np.array([1, 0, 0], np.float32) + rays["v"][:, None] * np.array([0, 1, 0], np.float32)
    assert np.abs(ht - huv).max() <= tol
    ng = np.stack([rays["Ng_x"], rays["Ng_y"], rays["Ng_z"]], 1)
    assert np.abs(ng - np.array([0, 0, 1], np.float32)).max() <= tol
    # single-ray entry point gives the same answer as the stream
    one = rtc.aligned_rayhits(1)
    fill_rays(one, frm[None, :], dirs[:1])
    sc.intersect1(one)
    assert one["geomID"][0] == 0 and one["tfar"][0] == rays["tfar"][0] and one["u"][0] == rays["u"][0]
    # occluded variants: tfar == -inf
    occ = rtc.aligned_rays(256)
    fill_rays(occ, np.broadcast_to(frm, (256, 3)), dirs)
    sc.occluded1M(occ)
    assert np.all(occ["tfar"] == -np.inf)
    o1 = rtc.aligned_rays(1)
    fill_rays(o1, frm[None, :], dirs[:1])
    sc.occluded1(o1)
    assert o1["tfar"][0] == -np.inf
    sc.release()
    dev.release()


@pytest.mark.parametrize("mode,cfg", MODES)
def test_bomberman_1m_parity(rtc, po, bomberman_tris, mode, cfg):
    """BASELINE config 2: 1 M incoherent rays vs bomberman triangles; also the SURVEY.md 8d anchors."""
    verts, tris = bomberman_tris
    dev, sc = _scene(rtc, cfg, verts, tris)
    lo, hi = verts.min(0), verts.max(0)
    want = po.make_random_rays(1_000_000, lo, hi, seed=0)
    got = want.copy()
    orc = po.TriangleScene(verts, tris, mode)
    orc.intersect1M(want, nthreads=8)
    sc.intersect1M(got)
    nh = compare_hits(got, want, what=f"bomberman mode {mode}")
    hit = got["geomID"] != INVALID
    assert nh == 227_188                                                   # reference output, SURVEY.md section 8d
    assert int(got["primID"][hit].astype(np.uint64).sum()) == 10_389_122   # same
    # occluded: a ray is occluded iff the closest-hit query hits
    occ = rtc.aligned_rays(1_000_000)
    src = po.make_random_rays(1_000_000, lo, hi, seed=0)
    for f in occ.dtype.names:
        occ[f] = src[f]
    wocc = occ.copy()
    sc.occluded1M(occ)
    orc.occluded1M(wocc, nthreads=8)
    assert np.array_equal(occ["tfar"], wocc["tfar"])
    assert np.array_equal(occ["tfar"] == -np.inf, hit)
    orc.free()
    sc.release()
    dev.release()


@pytest.mark.parametrize("mode,cfg", MODES)
@pytest.mark.parametrize("ntris,seed", [(1, 3), (5, 4), (37, 5), (5000, 6)])
def test_random_soup_multi_geometry(rtc, po, mode, cfg, ntris, seed):
    """Random triangle soups split over several geometries: IDs exact incl. geomID, ragged leaf sizes."""
    verts, tris = random_soup(ntris, seed)
    dev = rtc.Device(cfg)
    sc = rtc.Scene(dev)
    ngeo = min(3, ntris)
    bounds = np.linspace(0, ntris, ngeo + 1).astype(int)
    geom_ids, prim_ids = np.zeros(ntris, np.uint32), np.zeros(ntris, np.uint32)
    for g in range(ngeo):
        a, b = bounds[g], bounds[g + 1]
        gid = sc.add_triangles(verts, tris[a:b])
        geom_ids[a:b] = gid
        prim_ids[a:b] = np.arange(b - a)
    sc.commit()
    lo, hi = verts.min(0) - 1, verts.max(0) + 1
    org, d = random_rays_np(20000, lo, hi, seed + 100)
    want = rtc.aligned_rayhits(20000)
    fill_rays(want, org, d)
    got = want.copy()
    orc = po.TriangleScene(verts, tris, mode, geom_ids, prim_ids)
    orc.intersect1M(want, inst_id=7)
    sc.intersect1M(got, ctx=rtc.make_context(7))
    compare_hits(got, want, what=f"soup {ntris}")
    orc.free()
    sc.release()
    dev.release()


def test_stream_semantics(rtc, po, bomberman_tris):
    """Strided / 4-byte-aligned streams, tnear > tfar skipping, tnear/tfar windows, pre-set hits, M = 0."""
    verts, tris = bomberman_tris
    dev, sc = _scene(rtc, "tri_accel=bvh8.triangle4v", verts, tris)
    orc = po.TriangleScene(verts, tris, 0)
    lo, hi = verts.min(0), verts.max(0)
    m = 5000
    base = po.make_random_rays(m, lo, hi, seed=11)
    rng = np.random.RandomState(2)
    base["tnear"] = rng.rand(m).astype(np.float32) * 50
    base["tfar"] = np.where(rng.rand(m) < 0.5, np.inf, rng.rand(m) * 400).astype(np.float32)
    base["tnear"][::7] = 500.0
    base["tfar"][::7] = 100.0  # invalid: tnear > tfar -> skipped, record untouched
    base["geomID"][::11] = 5   # pre-set hit records stay unless a closer hit is found
    base["primID"][::11] = 9
    want = base.copy()
    orc.intersect1M(want)
    # (a) byteStride 96, base address only 4-byte aligned
    raw = np.zeros(m * 96 + 64, dtype=np.uint8)
    off = ((-raw.ctypes.data) % 16) + 4
    view = np.ndarray(shape=(m,), dtype=rtc.RAYHIT_DTYPE, buffer=raw.data, offset=off, strides=(96,))
    view[:] = base
    ctx = rtc.make_context()
    sc.lib.rtcIntersect1M(sc.handle, rtc.C.byref(ctx), view.ctypes.data, m, 96)
    dev.check("strided")
    got = np.array(view)
    assert np.array_equal(got.view(np.uint8).reshape(m, 80)[::7], base.view(np.uint8).reshape(m, 80)[::7])
    sel = np.ones(m, bool)
    sel[::11] = False
    compare_hits(got[sel], want[sel], what="strided stream")
    assert np.array_equal(got["geomID"][::11], want["geomID"][::11])
    assert np.array_equal(got["primID"][::11], want["primID"][::11])
    # (b) pointer stream
    recs = [rtc.aligned_rayhits(1) for _ in range(64)]
    for i, r in enumerate(recs):
        r[0] = base[i]
    arr = (rtc.C.c_void_p * 64)(*[r.ctypes.data for r in recs])
    sc.lib.rtcIntersect1Mp(sc.handle, rtc.C.byref(ctx), arr, 64)
    dev.check("1Mp")
    for i, r in enumerate(recs):
        assert r["geomID"][0] == want["geomID"][i] and r["primID"][0] == want["primID"][i]
    # (c) empty batch is a no-op
    sc.lib.rtcIntersect1M(sc.handle, rtc.C.byref(ctx), view.ctypes.data, 0, 96)
    dev.check("empty batch")
    orc.free()
    sc.release()
    dev.release()


def test_empty_scene_and_errors(rtc):
    dev = rtc.Device("")
    sc = rtc.Scene(dev)
    rays = rtc.aligned_rayhits(4)
    fill_rays(rays, np.zeros((4, 3), np.float32), np.tile(np.array([0, 0, 1], np.float32), (4, 1)))
    # tracing before commit: RTC_ERROR_INVALID_OPERATION "scene got not committed" (scene.cpp:25,54)
    sc.intersect1M(rays, check=False)
    assert dev.error() == rtc.RTC_ERROR_INVALID_OPERATION
    sc.commit()
    sc.intersect1M(rays)
    assert np.all(rays["geomID"] == INVALID) and np.all(np.isinf(rays["tfar"]))
    sc.release()
    dev.release()


def test_device_resident_stream(rtc, po, bomberman_tris):
    """Rays already in HBM (torch tensor): traced in place on the device's stream, no staging."""
    import torch
    verts, tris = bomberman_tris
    dev, sc = _scene(rtc, "tri_accel=bvh8.triangle4v", verts, tris)
    lo, hi = verts.min(0), verts.max(0)
    want = po.make_random_rays(200_000, lo, hi, seed=5)
    host = want.copy()
    orc = po.TriangleScene(verts, tris, 0)
    orc.intersect1M(want, nthreads=8)
    t = torch.from_numpy(host.view(np.uint8).reshape(-1, 80).copy()).cuda()
    torch.cuda.synchronize()
    sc.intersect1M(t)
    dev.synchronize()
    got = t.cpu().numpy().reshape(-1).view(rtc.RAYHIT_DTYPE)
    compare_hits(got, want, what="device-resident")
    cnt = sc.intersect1M_counted(torch.from_numpy(host.view(np.uint8).reshape(-1, 80).copy()).cuda())
    assert cnt["rays"] == 200_000 and cnt["hits"] == int((want["geomID"] != INVALID).sum())
    assert cnt["nodeVisits"] > 0 and cnt["primTests"] > 0
    orc.free()
    sc.release()
    dev.release()
