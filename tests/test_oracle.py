"""Pinning the oracle (CPU, no GPU): the reference's known-answer test, the reference outputs recorded in
SURVEY.md, the reference's own arithmetic primitives (oracle/_ref), and the committed golden fixtures."""
import os

import numpy as np
import pytest

from helpers import INVALID, compare_hits, fill_rays, unit_triangle_rays

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ULP = np.float32(1.1920929e-7)


def _rayhits(m):
    import importlib
    return importlib.import_module("embree-compressed_amd").rtc.aligned_rayhits(m)


@pytest.mark.parametrize("mode", [0, 1])
def test_triangle_hit_kat(po, mode):
    """TriangleHitTest, tutorials/verify/verify.cpp:2118-2205, asserted with the reference's own tolerances."""
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tris = np.array([[0, 1, 2]], np.uint32)
    sc = po.TriangleScene(verts, tris, mode)
    frm, dirs, u, v = unit_triangle_rays(None, 256)
    rays = _rayhits(256)
    fill_rays(rays, np.broadcast_to(frm, (256, 3)), dirs)
    occ = rays.copy()
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
    sc.occluded1M(occ)
    assert np.all(occ["tfar"] == -np.inf)
    sc.free()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("double_eval", [True, False])
def test_reference_outputs_on_bomberman(po, bomberman_tris, mode, double_eval):
    """Reference outputs recorded in SURVEY.md section 6/8d (measured on the real library): 1 M rays of the
    srand48(0) generator vs bomberman's 1454 fan triangles -> 227 188 hits, sum(primID) = 10 389 122, for
    bvh8.triangle4v, qbvh8.triangle4 and the default accel alike."""
    verts, tris = bomberman_tris
    sc = po.TriangleScene(verts, tris, mode)
    rays = po.make_random_rays(1_000_000, verts.min(0), verts.max(0), seed=0, double_eval=double_eval)
    sc.intersect1M(rays, nthreads=8)
    hit = rays["geomID"] != INVALID
    assert int(hit.sum()) == 227_188
    assert int(rays["primID"][hit].astype(np.uint64).sum()) == 10_389_122
    sc.free()


def test_primitives_match_reference_headers(po):
    """oracle/_ref = the reference's common/math + common/simd compiled in place: rcp, dot, cross,
    stable_triangle_normal, select_min must agree bit for bit (same FMA placement, same rcpps + Newton step)."""
    R = po.ref()
    if R is None:
        pytest.skip("oracle/_ref/libref_prims.so not built (no reference tree)")
    L = po.lib()
    rng = np.random.RandomState(7)
    for scale in (1e-3, 1.0, 1e4):
        a = (rng.randn(2000, 3, 4) * scale).astype(np.float32)
        b = (rng.randn(2000, 3, 4) * scale).astype(np.float32)
        c = (rng.randn(2000, 3, 4) * scale).astype(np.float32)
        for i in range(0, 2000, 7):
            A, B, Cc = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i]), np.ascontiguousarray(c[i])
            od, oc, on = np.zeros(4, np.float32), np.zeros((3, 4), np.float32), np.zeros((3, 4), np.float32)
            R.ref_dot4(A.ctypes.data, B.ctypes.data, od.ctypes.data)
            R.ref_cross4(A.ctypes.data, B.ctypes.data, oc.ctypes.data)
            R.ref_stable_triangle_normal4(A.ctypes.data, B.ctypes.data, Cc.ctypes.data, on.ctypes.data)
            for l in range(4):
                av, bv, cv = np.ascontiguousarray(A[:, l]), np.ascontiguousarray(B[:, l]), np.ascontiguousarray(Cc[:, l])
                assert L.orc_dot(av.ctypes.data, bv.ctypes.data) == od[l]
                o = np.zeros(3, np.float32)
                L.orc_cross(av.ctypes.data, bv.ctypes.data, o.ctypes.data)
                assert np.array_equal(o, oc[:, l])
                L.orc_stable_triangle_normal(av.ctypes.data, bv.ctypes.data, cv.ctypes.data, o.ctypes.data)
                assert np.array_equal(o, on[:, l])
    xs = np.concatenate([rng.randn(5000).astype(np.float32) * s for s in (1e-6, 1.0, 1e6)])
    for x in xs[::3]:
        assert L.orc_rcp(float(x)) == R.ref_rcp(float(x))
    # select_min: lowest lane among equal minima, masked lanes ignored
    for _ in range(500):
        v = rng.randint(0, 3, 4).astype(np.float32)
        mask = int(rng.randint(1, 16))
        want = R.ref_select_min4(mask, v.ctypes.data)
        vals = [(v[l], l) for l in range(4) if mask & (1 << l)]
        m = min(x for x, _ in vals)
        assert want == min(l for x, l in vals if x == m)


def test_fork_primitives_match_reference_headers(po):
    """The arithmetic the fork's cBVH traversal is built from (compressed.h:498-505,587; compressed_help.h:111,156-157,252;
    compressed_leaf.h:99-111): scalar rsqrt (rsqrtss + Newton step), Vec3fa dot (dpps) / length / normalize, Vec3f rcp_safe.
    oracle/subdiv_oracle.inc calls the orc_* restatements; here they are compared bit for bit with the reference's own
    headers compiled in place."""
    R = po.ref()
    if R is None:
        pytest.skip("oracle/_ref/libref_prims.so not built (no reference tree)")
    L = po.lib()
    rng = np.random.RandomState(11)
    o1, o2 = np.zeros(3, np.float32), np.zeros(4, np.float32)
    for i in range(6000):
        sc = 10.0 ** rng.uniform(-6, 6)
        a = (rng.randn(3) * sc).astype(np.float32)
        b = (rng.randn(3) * sc).astype(np.float32)
        if i % 50 == 0:
            a[rng.randint(3)] = np.float32(1e-20)  # below min_rcp_input: zero_fix path
        x = float(np.float32(abs(a[0]) + 1e-30))
        assert np.float32(L.orc_rsqrt(x)).tobytes() == np.float32(R.ref_rsqrt(x)).tobytes()
        assert np.float32(L.orc_dot3fa(a.ctypes.data, b.ctypes.data)).tobytes() == np.float32(R.ref_dot3fa(a.ctypes.data, b.ctypes.data)).tobytes()
        assert np.float32(L.orc_length3(a.ctypes.data)).tobytes() == np.float32(R.ref_length3(a.ctypes.data)).tobytes()
        L.orc_normalize3(a.ctypes.data, o1.ctypes.data)
        R.ref_normalize3(a.ctypes.data, o2.ctypes.data)
        assert o1.tobytes() == o2[:3].tobytes()
        R.ref_rcp_safe3f(a.ctypes.data, o2.ctypes.data)
        mine = np.array([L.orc_rcp(float(v) if abs(v) >= 1e-18 else 1e-18) for v in a], np.float32)
        assert mine.tobytes() == o2[:3].tobytes()
    F = po.ref_fork()
    if F is not None:  # getDelta() = rcp(16.f) of the fork's own header
        assert np.float32(F.ref_fork_leaf_delta()).tobytes() == np.float32(L.orc_rcp(16.0)).tobytes()


def test_block_tie_rules(po):
    """Two coplanar triangles in one Triangle4v block hit at the same t: the lowest lane wins (select_min,
    vfloat4_sse2.h:654-659); a later block with an equal t replaces the hit (T <= absDen*tfar, pluecker.h:117)."""
    L = po.lib()
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    v0 = np.zeros((3, 4), np.float32); v1 = np.zeros((3, 4), np.float32); v2 = np.zeros((3, 4), np.float32)
    for l in (1, 2):  # lanes 1 and 2 hold the same triangle, lanes 0 and 3 are empty (zero vertices)
        v0[:, l], v1[:, l], v2[:, l] = tri[0], tri[1], tri[2]
    org = np.array([0.25, 0.25, -1], np.float32); d = np.array([0, 0, 1], np.float32)
    out = np.zeros(6, np.float32)
    lane = L.orc_pluecker_block(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, org.ctypes.data, d.ctypes.data, 0.0, np.inf, out.ctypes.data)
    assert lane == 1 and abs(out[0] - 1.0) <= 16 * ULP
    # tfar == t still accepts (<=), tfar just below rejects
    assert L.orc_pluecker_block(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, org.ctypes.data, d.ctypes.data, 0.0, 1.0, out.ctypes.data) == 1
    assert L.orc_pluecker_block(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, org.ctypes.data, d.ctypes.data, 0.0, np.float32(0.9999999), out.ctypes.data) == -1
    # tnear is exclusive: absDen*tnear < T
    assert L.orc_pluecker_block(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, org.ctypes.data, d.ctypes.data, 1.0, np.inf, out.ctypes.data) == -1
    assert L.orc_moeller_block(v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, org.ctypes.data, d.ctypes.data, 0.0, np.inf, out.ctypes.data) == 1


@pytest.mark.parametrize("mode", [0, 1])
def test_golden_fixture(po, bomberman_tris, mode):
    """tests/golden/bomberman_tri_hits.npz (made by tests/golden/make_golden.py with this oracle after it had
    been pinned above): REGRESSION vectors - they guard the oracle itself against drift and travel to the GPU box; the parity
    anchors of the triangle path are the reference-held TriangleHitTest and the survey-recorded reference outputs above."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "bomberman_tri_hits.npz"))
    verts, tris = bomberman_tris
    m = int(g["count"])
    rays = po.make_random_rays(m, verts.min(0), verts.max(0), seed=int(g["seed"]))
    sc = po.TriangleScene(verts, tris, mode)
    sc.intersect1M(rays)
    assert np.array_equal(rays["geomID"], g["geomID"])
    assert np.array_equal(rays["primID"], g["primID"])
    hit = rays["geomID"] != INVALID
    for f in ("tfar", "u", "v"):
        a, b = rays[f][hit].astype(np.float64), g[f"{f}_{mode}"][hit].astype(np.float64)
        assert np.all(np.abs(a - b) <= 1e-5 * np.maximum(np.abs(b), 1e-3))  # rcpps differs between CPU vendors
    sc.free()


def test_raygen_matches_oracle_generator(po, bomberman_tris):
    import importlib
    rg = importlib.import_module("embree-compressed_amd.raygen")
    verts, _ = bomberman_tris
    a = rg.make_random_rays(50_000, verts.min(0), verts.max(0), seed=9)
    b = po.make_random_rays(50_000, verts.min(0), verts.max(0), seed=9)
    assert np.array_equal(a.reshape(-1), b.view(np.uint8))
    # first draws of drand48 after srand48(0) (glibc): 0.170828036..., 0.749901980...
    x = rg.lcg48_sequence(2, 0).astype(np.float64) / 2.0 ** 48
    assert abs(x[0] - 0.17082803610628972) < 1e-15 and abs(x[1] - 0.7499019804849638) < 1e-15
