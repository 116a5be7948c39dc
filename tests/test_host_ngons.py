"""Subdivision faces that are not quads (triangles, pentagons, ...) on a `gpu=none` device.

The reference splits an N-gon into N sub-patches after one Catmull-Clark step, tessellates each like a quad patch and
encodes the sub-patch number in the integer part of uv (patch_eval_grid.h:221-258, catmullclark_patch.h:442-483).  Checked
against an independent numpy Catmull-Clark implementation for arbitrary polygons and against the pure-quad code path."""
import numpy as np
import pytest

from test_host_interpolate import _torus


def _cc_step(V, faces):
    """one Catmull-Clark step on a closed polygon mesh (numpy, independent of the product): returns vertices, quad faces"""
    V = np.asarray(V, np.float64)
    nV = len(V)
    fp = np.array([V[list(f)].mean(0) for f in faces])
    edges = {}
    vfaces = [[] for _ in range(nV)]
    for fi, f in enumerate(faces):
        for k, a in enumerate(f):
            b = f[(k + 1) % len(f)]
            edges.setdefault((min(a, b), max(a, b)), []).append(fi)
            vfaces[a].append(fi)
    eidx = {e: i for i, e in enumerate(edges)}
    ep = np.array([(V[a] + V[b] + fp[fs[0]] + fp[fs[1]]) / 4 for (a, b), fs in edges.items()])
    vn = [[] for _ in range(nV)]
    for (a, b) in edges:
        vn[a].append(b)
        vn[b].append(a)
    newV = np.zeros_like(V)
    for v in range(nV):
        n = len(vn[v])
        Q = fp[vfaces[v]].mean(0)
        R = np.mean([(V[v] + V[w]) / 2 for w in vn[v]], 0)
        newV[v] = (Q + 2 * R + (n - 3) * V[v]) / n
    allV = np.concatenate([newV, ep, fp])
    quads = []
    for fi, f in enumerate(faces):
        N = len(f)
        for k in range(N):
            e_next = nV + eidx[(min(f[k], f[(k + 1) % N]), max(f[k], f[(k + 1) % N]))]
            e_prev = nV + eidx[(min(f[k - 1], f[k]), max(f[k - 1], f[k]))]
            quads.append((f[k], e_next, nV + len(ep) + fi, e_prev))
    return allV, quads


def _limit(V, quads, v):
    """limit position of interior vertex v of an all-quad mesh: (n^2 P + 4 sum E + sum F) / (n (n+5))"""
    E, F = set(), set()
    for q in quads:
        if v in q:
            k = q.index(v)
            E.update([q[(k + 1) % 4], q[(k + 3) % 4]])
            F.add(q[(k + 2) % 4])
    n = len(E)
    return (n * n * V[v] + 4 * V[list(E)].sum(0) + V[list(F)].sum(0)) / (n * (n + 5))


def _grids(sc, L):
    raw = sc.accel_data(4)
    w = 2 ** L + 1
    per = 12 + 12 * w * w
    out = []
    for p in range(len(raw) // per):
        h = raw[p * per: p * per + 12].view(np.uint32)
        out.append((int(h[1]), raw[p * per + 12: (p + 1) * per].view(np.float32).reshape(3, w, w)))
    return out


PRISM_V = np.array([[0, 0, 0], [2, 0, 0], [1, 1.7, 0], [0, 0, 3], [2, 0, 3], [1, 1.7, 3]], np.float32)
PRISM_F = [(0, 2, 1), (3, 4, 5), (0, 1, 4, 3), (1, 2, 5, 4), (2, 0, 3, 5)]  # 2 triangles + 3 quads, closed, consistently oriented


def _prism_scene(rtc, L, cfg="gpu=none,keep_grids=1"):
    dev = rtc.Device(cfg)
    sc = rtc.Scene(dev)
    fs = np.array([len(f) for f in PRISM_F], np.uint32)
    fi = np.concatenate([np.array(f, np.uint32) for f in PRISM_F])
    gid = sc.add_subdiv(PRISM_V, fs, fi)
    sc.set_levels(L, min(L, 2))
    sc.commit()
    return dev, sc, gid


def test_ngon_subpatches_hit_the_limit_points_of_an_independent_refinement(rtc):
    L = 3
    dev, sc, gid = _prism_scene(rtc, L)
    V1, Q1 = _cc_step(PRISM_V, PRISM_F)
    V2, Q2 = _cc_step(V1, Q1)  # limit stencils are evaluated one level further down, on an all-quad mesh with isolated extraordinary points
    grids = _grids(sc, L)
    # patch order: faces in order; a triangle gives 3 grids (sub-patch 0,1,2), a quad one grid
    expect = [(0, 3), (1, 3), (2, 1), (3, 1), (4, 1)]
    assert [p for p, _ in grids] == [p for p, k in expect for _ in range(k)]
    nV0 = len(PRISM_V)
    nE0 = 9
    gi = 0
    for prim, k in expect:
        f = PRISM_F[prim]
        if k == 3:
            for sub in range(3):
                g = grids[gi][1]
                gi += 1
                corner = _limit(V2, Q2, f[sub])                 # limit of control vertex `sub`
                centre = _limit(V2, Q2, nV0 + nE0 + prim)       # limit of the face point (valence 3)
                assert np.abs(g[:, 0, 0] - corner).max() < 2e-6 * 3
                assert np.abs(g[:, -1, -1] - centre).max() < 2e-6 * 3
                # (1,0) corner = limit of the edge point towards the next corner; (0,1) = towards the previous one
                nxt, prv = f[(sub + 1) % 3], f[sub - 1]
                e_keys = {}
                idx = 0
                for ff in PRISM_F:
                    for kk, a in enumerate(ff):
                        b = ff[(kk + 1) % len(ff)]
                        if (min(a, b), max(a, b)) not in e_keys:
                            e_keys[(min(a, b), max(a, b))] = idx
                            idx += 1
                en = nV0 + e_keys[(min(f[sub], nxt), max(f[sub], nxt))]
                epv = nV0 + e_keys[(min(prv, f[sub]), max(prv, f[sub]))]
                assert np.abs(g[:, 0, -1] - _limit(V2, Q2, en)).max() < 2e-6 * 3    # row 0 (v=0), last column (u=1)
                assert np.abs(g[:, -1, 0] - _limit(V2, Q2, epv)).max() < 2e-6 * 3
        else:
            g = grids[gi][1]
            gi += 1
            want = [_limit(V2, Q2, f[c]) for c in range(4)]
            got = [g[:, 0, 0], g[:, 0, -1], g[:, -1, -1], g[:, -1, 0]]  # uv corners (0,0),(1,0),(1,1),(0,1)
            for a, b in zip(got, want):
                assert np.abs(a - b).max() < 2e-6 * 3
            assert np.abs(g[:, 4, 4] - _limit(V2, Q2, nV0 + nE0 + prim)).max() < 2e-6 * 3  # face point at (1/2, 1/2)
    # uv window of the blobs: sub-patch k of a triangle lives in [2k + 0.5, 2k + 1.5] x [0.5, 1.5]
    sc.release()
    dev.release()


def test_quads_of_a_mixed_mesh_equal_the_pure_quad_path(rtc):
    verts, faces, nu, nv = _torus()
    L = 3
    dev = rtc.Device("gpu=none,keep_grids=1")
    sc = rtc.Scene(dev)
    sc.add_subdiv(verts, np.full(len(faces), 4, np.uint32), faces.ravel())
    sc.set_levels(L, 2)
    sc.commit()
    pure = _grids(sc, L)
    # the same torus plus a far-away tetrahedron of triangles in the same geometry
    tet = np.array([[100, 100, 100], [101, 100, 100], [100, 101, 100], [100, 100, 101]], np.float32)
    o = len(verts)
    tf = [(o, o + 2, o + 1), (o, o + 1, o + 3), (o + 1, o + 2, o + 3), (o + 2, o, o + 3)]
    v2 = np.concatenate([verts, tet])
    fs = np.concatenate([np.full(len(faces), 4, np.uint32), np.full(4, 3, np.uint32)])
    fi = np.concatenate([faces.ravel(), np.array(tf, np.uint32).ravel()])
    sc2 = rtc.Scene(dev)
    gid = sc2.add_subdiv(v2, fs, fi)
    sc2.set_levels(L, 2)
    sc2.commit()
    mixed = _grids(sc2, L)
    assert len(mixed) == len(pure) + 12
    scale = np.abs(verts).max()
    for (p0, g0), (p1, g1) in zip(pure, mixed):
        assert p0 == p1
        assert np.abs(g0 - g1).max() < 1e-6 * scale
    assert [p for p, _ in mixed[len(pure):]] == [len(faces) + k for k in range(4) for _ in range(3)]
    # rtcInterpolate: a quad face of the mixed mesh goes through four corner sub-quads and a chain rule; it must agree with
    # the pure mesh everywhere, derivatives included
    rng = np.random.RandomState(4)
    for _ in range(60):
        f = int(rng.randint(len(faces)))
        u, v = rng.rand(2)
        a = sc.interpolate(0, f, float(u), float(v))
        b = sc2.interpolate(gid, f, float(u), float(v))
        for x, y, tol in zip(a, b, (1e-6, 1e-5, 1e-5, 2e-4, 2e-4, 2e-4)):
            assert np.abs(x - y).max() <= tol * max(1.0, np.abs(x).max()), (f, u, v, x, y)
    # a triangle face: uv = (2*sub + 0.5 + s, 0.5 + t); the corners of the sub-patch are the tessellator's grid corners
    tri_grids = mixed[len(pure):]
    for k in range(12):
        prim, g = tri_grids[k]
        sub = k % 3
        for (s, t, J, I) in ((0, 0, 0, 0), (1, 0, 0, -1), (1, 1, -1, -1), (0, 1, -1, 0), (0.5, 0.25, 2, 4)):
            P = sc2.interpolate(gid, prim, 2 * sub + 0.5 + s, 0.5 + t, derivs=0)[0]
            assert np.abs(P - g[:, J, I]).max() < 2e-6 * 101, (prim, sub, s, t)
    with pytest.raises(rtc.RTCError):
        sc2.interpolate(gid, len(faces), 6.6, 0.7)  # sub-patch 3 of a triangle does not exist
    sc.release()
    sc2.release()
    dev.release()


def test_subpatch_uv_windows_in_the_leaf_records(rtc):
    dev, sc, gid = _prism_scene(rtc, 3, "gpu=none,subdiv_accel=bvh4.compressed.grid")
    raw = sc.accel_data(2)
    stride = sc.stats()["primBytes"]
    hdr = np.frombuffer(raw.tobytes(), dtype=np.dtype([("geomID", "<u4"), ("primID", "<u4"), ("uv0", "<f4", 2), ("uv1", "<f4", 2)]), count=len(raw) // stride,
                        offset=0) if False else None
    n = len(raw) // stride
    seen = {}
    for b in range(n):
        h = raw[b * stride + 128: b * stride + 152]  # ids + uv window: bytes 128.. of a blob (accel.h, CbvhMid)
        prim = int(h[4:8].view(np.uint32)[0])
        uv0 = h[8:16].view(np.float32)
        uv1 = h[16:24].view(np.float32)
        seen.setdefault(prim, []).append((float(uv0[0]), float(uv0[1]), float(uv0[0] + uv1[0]), float(uv0[1] + uv1[1])))
    for prim in (0, 1):  # triangles: 3 sub-patches x 4 blobs (L=3, C=2), windows inside [2k+.5, 2k+1.5] x [.5, 1.5]
        assert len(seen[prim]) == 12
        subs = sorted({int((a - 0.5) // 2) for a, _, _, _ in seen[prim]})
        assert subs == [0, 1, 2]
        for a, b, c, d in seen[prim]:
            k = int((a - 0.5) // 2)
            assert 2 * k + 0.5 <= a < c <= 2 * k + 1.5 and 0.5 <= b < d <= 1.5
    for prim in (2, 3, 4):
        assert len(seen[prim]) == 4 and all(0 <= a < c <= 1 and 0 <= b < d <= 1 for a, b, c, d in seen[prim])
    sc.release()
    dev.release()
