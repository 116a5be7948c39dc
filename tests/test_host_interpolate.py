"""rtcInterpolate / rtcInterpolateN on a `gpu=none` device.

Triangle meshes follow scene_triangle_mesh.cpp:214-270 exactly.  Subdivision meshes are checked against an independent
numpy evaluation of the uniform bicubic B-spline surface of the CONTROL mesh (what the limit surface of a regular region
is), against the tessellator's limit grid at dyadic parameters, and for the internal consistency of the derivatives."""
import numpy as np
import pytest


def _bspline(s):
    r = 1 - s
    B = np.array([r ** 3 / 6, (3 * s ** 3 - 6 * s ** 2 + 4) / 6, (-3 * s ** 3 + 3 * s ** 2 + 3 * s + 1) / 6, s ** 3 / 6])
    dB = np.array([-0.5 * r * r, 1.5 * s * s - 2 * s, -1.5 * s * s + s + 0.5, 0.5 * s * s])
    ddB = np.array([r, 3 * s - 2, -3 * s + 1, s])
    return B, dB, ddB


def _torus(nu=8, nv=6):
    """closed all-regular quad mesh (every vertex has valence 4)"""
    verts = np.zeros((nu * nv, 3), np.float32)
    for j in range(nv):
        for i in range(nu):
            a, b = 2 * np.pi * i / nu, 2 * np.pi * j / nv
            verts[j * nu + i] = [(3 + np.cos(b) * (1 + 0.2 * np.sin(3 * a))) * np.cos(a), (3 + np.cos(b)) * np.sin(a), np.sin(b) + 0.1 * np.cos(2 * a)]
    faces = [(j * nu + i, j * nu + (i + 1) % nu, ((j + 1) % nv) * nu + (i + 1) % nu, ((j + 1) % nv) * nu + i) for j in range(nv) for i in range(nu)]
    return verts, np.array(faces, np.uint32), nu, nv


def test_triangle_interpolation_matches_the_reference_formula(rtc):
    dev = rtc.Device("gpu=none")
    sc = rtc.Scene(dev)
    rng = np.random.RandomState(3)
    v = rng.rand(5, 3).astype(np.float32)
    t = np.array([[0, 1, 2], [2, 3, 4]], np.uint32)
    gid = sc.add_triangles(v, t)
    col = rng.rand(5, 4).astype(np.float32)
    sc.set_vertex_attribute(gid, 0, col)
    sc.commit()
    for prim in (0, 1):
        for u, w in ((0.25, 0.5), (0.0, 0.0), (1.0, 0.0), (0.1, 0.9)):
            P, du, dv, duu, dvv, duv = sc.interpolate(gid, prim, u, w)
            p0, p1, p2 = v[t[prim]]
            ww = np.float32(1.0) - np.float32(u) - np.float32(w)
            want = np.array([np.float32(np.float64(ww) * p0[k] + (np.float64(np.float32(u)) * p1[k] + np.float64(np.float32(np.float32(w) * p2[k])))) for k in range(3)], np.float32)
            assert np.allclose(P, want, rtol=0, atol=1e-6)
            assert np.array_equal(du, p1 - p0) and np.array_equal(dv, p2 - p0)
            assert not duu.any() and not dvv.any() and not duv.any()
            C4 = sc.interpolate(gid, prim, u, w, buffer_type=rtc.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, slot=0, count=4, derivs=1)
            c0, c1, c2 = col[t[prim]]
            assert np.allclose(C4[0], ww * c0 + np.float32(u) * c1 + np.float32(w) * c2, atol=1e-6)
            assert np.array_equal(C4[1], c1 - c0)
    Pn, dun, dvn = sc.interpolateN(gid, [0, 1, 1], [0.2, 0.3, 0.4], [0.1, 0.2, 0.3], valid=[-1, 0, -1])
    assert Pn.shape == (3, 3) and not Pn[:, 1].any()  # masked-out entry untouched, SoA [value][i] layout
    assert np.allclose(Pn[:, 2], sc.interpolate(gid, 1, 0.4, 0.3)[0])
    sc.release()
    dev.release()


def test_subdiv_interpolation_is_the_bicubic_bspline_surface_on_a_regular_mesh(rtc):
    verts, faces, nu, nv = _torus()
    dev = rtc.Device("gpu=none")
    sc = rtc.Scene(dev)
    gid = sc.add_subdiv(verts, np.full(len(faces), 4, np.uint32), faces.ravel())
    sc.commit()
    V = verts.astype(np.float64).reshape(nv, nu, 3)
    rng = np.random.RandomState(5)
    worst = 0.0
    for _ in range(200):
        f = rng.randint(len(faces))
        u, v = rng.rand(2)
        if rng.rand() < 0.2:
            u, v = rng.choice([0.0, 1.0, 0.5, 0.125]), rng.choice([0.0, 1.0, 0.375])
        fj, fi = divmod(f, nu)
        Bu, dBu, ddBu = _bspline(u)
        Bv, dBv, ddBv = _bspline(v)
        ctl = np.array([[V[(fj - 1 + r) % nv, (fi - 1 + c) % nu] for c in range(4)] for r in range(4)])  # [r][c][xyz]
        want = [np.einsum("r,c,rck->k", a, b, ctl) for a, b in ((Bv, Bu), (Bv, dBu), (dBv, Bu), (Bv, ddBu), (ddBv, Bu), (dBv, dBu))]
        got = sc.interpolate(gid, f, float(u), float(v))
        for g, w, tol in zip(got, want, (2e-6, 1e-5, 1e-5, 1e-4, 1e-4, 1e-4)):
            err = np.abs(g - w).max() / max(1.0, np.abs(w).max())
            worst = max(worst, err)
            assert err < tol, (f, u, v, g, w)
    assert worst > 0  # float32 outputs of a double evaluation
    sc.release()
    dev.release()


def test_subdiv_interpolation_agrees_with_the_tessellator_and_itself(rtc, bomberman):
    """bomberman has extraordinary vertices and boundaries: at the dyadic parameters of level 3 every face must give the
    tessellator's limit point (both evaluation branches are exact there); derivatives agree with central differences in
    regular regions; a copy of the vertex buffer bound as attribute interpolates identically."""
    verts, fs, fi = bomberman
    dev = rtc.Device("gpu=none,keep_grids=1")
    sc = rtc.Scene(dev)
    gid = sc.add_subdiv(verts, fs, fi)
    sc.set_vertex_attribute(gid, 1, np.concatenate([verts, verts[:, :1] * 2], 1))
    sc.set_levels(3, 2)
    sc.commit()
    raw = sc.accel_data(4)
    w = 9
    per = 12 + 12 * w * w
    grids = {int(raw[p * per: p * per + 12].view(np.uint32)[1]): raw[p * per + 12: (p + 1) * per].view(np.float32).reshape(3, w, w)
             for p in range(len(raw) // per)}
    scale = np.abs(verts).max()
    rng = np.random.RandomState(9)
    for prim in rng.choice(sorted(grids), 60, replace=False):
        g = grids[int(prim)]
        for (i, j) in ((0, 0), (8, 8), (3, 5), (8, 0), (4, 4), (1, 7)):
            P = sc.interpolate(gid, int(prim), i / 8.0, j / 8.0, derivs=0)[0]
            assert np.abs(P - g[:, j, i]).max() < 2e-6 * scale, (prim, i, j)
        u, v = 0.3 + 0.4 * rng.rand(2)
        P, du, dv, duu, dvv, duv = sc.interpolate(gid, int(prim), u, v)
        A = sc.interpolate(gid, int(prim), u, v, buffer_type=rtc.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, slot=1, count=4, derivs=1)
        assert np.allclose(A[0][:3], P, atol=1e-6 * scale) and abs(A[0][3] - 2 * P[0]) < 1e-5 * scale
        assert np.allclose(A[1][:3], du, atol=1e-5 * scale)
        h = 1e-3
        fdu = (sc.interpolate(gid, int(prim), u + h, v, derivs=0)[0].astype(np.float64) - sc.interpolate(gid, int(prim), u - h, v, derivs=0)[0]) / (2 * h)
        fdv = (sc.interpolate(gid, int(prim), u, v + h, derivs=0)[0].astype(np.float64) - sc.interpolate(gid, int(prim), u, v - h, derivs=0)[0]) / (2 * h)
        tol = 2e-2 * max(np.abs(du).max(), np.abs(dv).max(), 1e-3 * scale)  # float32 positions / 2h
        assert np.abs(fdu - du).max() < tol and np.abs(fdv - dv).max() < tol, (prim, u, v, fdu, du)
    with pytest.raises(rtc.RTCError):
        sc.interpolate(gid, 10 ** 6, 0.5, 0.5)
    sc.release()
    dev.release()


def test_face_varying_attribute_topology(rtc):
    """A vertex attribute with its own topology (rtcSetGeometryTopologyCount / rtcSetGeometryVertexAttributeTopology): every
    face owns four attribute vertices of its own (a texture atlas with a seam on every edge).  With PIN_ALL the attribute is
    the bilinear interpolation of the face's four values; with the position topology re-used it equals the plain attribute."""
    verts, faces, nu, nv = _torus()
    nf = len(faces)
    dev = rtc.Device("gpu=none")
    sc = rtc.Scene(dev)
    gid = sc.add_subdiv(verts, np.full(nf, 4, np.uint32), faces.ravel())
    rng = np.random.RandomState(12)
    per_face = rng.rand(nf * 4, 2).astype(np.float32)               # 4 private attribute vertices per face
    sc.set_vertex_attribute(gid, 0, per_face, topology_index=np.arange(nf * 4, dtype=np.uint32), mode=4)  # RTC_SUBDIVISION_MODE_PIN_ALL
    sc.commit()
    for _ in range(40):
        f = int(rng.randint(nf))
        u, v = rng.rand(2)
        c = per_face[4 * f: 4 * f + 4].astype(np.float64)            # corners in face order: (0,0) (1,0) (1,1) (0,1)
        want = (1 - u) * (1 - v) * c[0] + u * (1 - v) * c[1] + u * v * c[2] + (1 - u) * v * c[3]
        P, du, dv = sc.interpolate(gid, f, float(u), float(v), buffer_type=rtc.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, slot=0, count=2, derivs=1)[:3]
        assert np.abs(P - want).max() < 1e-6
        assert np.abs(du - ((1 - v) * (c[1] - c[0]) + v * (c[2] - c[3]))).max() < 1e-5
        assert np.abs(dv - ((1 - u) * (c[3] - c[0]) + u * (c[2] - c[1]))).max() < 1e-5
    # positions still use topology 0
    assert np.abs(sc.interpolate(gid, 3, 0.5, 0.5, derivs=0)[0]).max() > 0
    # second topology identical to the first: same as an ordinary attribute
    sc2 = rtc.Scene(dev)
    g2 = sc2.add_subdiv(verts, np.full(nf, 4, np.uint32), faces.ravel())
    col = rng.rand(len(verts), 3).astype(np.float32)
    sc2.set_vertex_attribute(g2, 0, col)
    sc2.set_vertex_attribute(g2, 1, col, topology_index=faces.ravel())
    sc2.commit()
    for _ in range(20):
        f = int(rng.randint(nf))
        u, v = (float(x) for x in rng.rand(2))
        a = sc2.interpolate(g2, f, u, v, buffer_type=rtc.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, slot=0)
        b = sc2.interpolate(g2, f, u, v, buffer_type=rtc.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, slot=1)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    sc.release()
    sc2.release()
    dev.release()


def test_subdiv_interpolation_next_to_extraordinary_vertices_is_feature_adaptive(rtc):
    """rtcInterpolate inside the cells that touch an extraordinary vertex: the reference subdivides such a patch adaptively
    down to depth 10 and evaluates regular B-spline sub-patches (feature_adaptive_eval.h:130-140, patch.h:40-42); only the
    last 2^-10 of the parameter range next to the vertex is a Gregory fill.  Checked against an INDEPENDENT numpy
    Catmull-Clark refinement (tests/test_host_ngons.py) of a cube (eight valence-3 vertices) and of a mesh with a valence-5
    vertex: following the corner child of a face for L steps gives the exact limit points of the surface at (0,0), (h,0),
    (h,h), (0,h), h = 2^-L - for L >= 4 these lie strictly inside the level-3 cell at the extraordinary corner, where a
    bilinear blend of the cell's corner limit points (the round-1 evaluation) is off by percents of the face size.
    Parity with the reference itself stays unpinned (no reference-held vector)."""
    from test_host_creases import CUBE_F, CUBE_V
    from test_host_ngons import _cc_step, _limit

    # valence 5: five quads around vertex 0, closed at the back by a second fan (double pyramid over a pentagon ring)
    ang = np.linspace(0, 2 * np.pi, 6)[:-1]
    ring = np.stack([np.cos(ang), np.sin(ang), 0 * ang], 1)
    mid = np.stack([np.cos(ang + np.pi / 5), np.sin(ang + np.pi / 5), 0 * ang], 1) * 1.3
    V5 = np.concatenate([[[0, 0, 1.0]], ring * 0.7 + [0, 0, 0.5], mid, [[0, 0, -1.0]]]).astype(np.float32)
    F5 = [(0, 1 + k, 6 + k, 1 + (k + 1) % 5) for k in range(5)] + [(11, 1 + (k + 1) % 5, 6 + k, 1 + k) for k in range(5)]

    for name, V, F in (("cube", CUBE_V, CUBE_F), ("valence5", V5, F5)):
        dev = rtc.Device("gpu=none")
        sc = rtc.Scene(dev)
        gid = sc.add_subdiv(V, np.full(len(F), 4, np.uint32), np.array(F, np.uint32).ravel())
        sc.commit()
        scale = float(np.abs(V).max())
        for face in (0, len(F) - 1):
            Vl, Q = np.asarray(V, np.float64), [tuple(f) for f in F]
            fi = face
            worst_old_style = 0.0
            for L in range(1, 7):
                Vl, Q = _cc_step(Vl, Q)
                fi = 4 * fi  # corner child at the face's first vertex: (v0, E01, F, E30), same (u, v) orientation
                h = 2.0 ** -L
                q = Q[fi]
                for (u, v), vid in (((0.0, 0.0), q[0]), ((h, 0.0), q[1]), ((h, h), q[2]), ((0.0, h), q[3])):
                    want = _limit(Vl, Q, vid)
                    got = sc.interpolate(gid, face, u, v, derivs=0)[0].astype(np.float64)
                    assert np.abs(got - want).max() < 3e-6 * scale, (name, face, L, u, v, got, want)
            # derivatives inside the extraordinary cell agree with central differences of the surface itself
            for (u, v) in ((0.03, 0.02), (0.06, 0.09), (0.004, 0.11), (0.0007, 0.0004)):
                P, du, dv = sc.interpolate(gid, face, u, v, derivs=1)[:3]
                e = 1e-4 * min(1.0, 50 * max(u, v))
                fdu = (sc.interpolate(gid, face, u + e, v, derivs=0)[0].astype(np.float64) - sc.interpolate(gid, face, max(u - e, 0.0), v, derivs=0)[0]) / (u + e - max(u - e, 0.0))
                fdv = (sc.interpolate(gid, face, u, v + e, derivs=0)[0].astype(np.float64) - sc.interpolate(gid, face, u, max(v - e, 0.0), derivs=0)[0]) / (v + e - max(v - e, 0.0))
                assert np.abs(du - fdu).max() < 2e-2 * max(1.0, np.abs(fdu).max()), (name, face, u, v, du, fdu)
                assert np.abs(dv - fdv).max() < 2e-2 * max(1.0, np.abs(fdv).max()), (name, face, u, v, dv, fdv)
        sc.release()
        dev.release()
